#!/usr/bin/env python3
"""Per-op GPU time of one vocoder decode (voc_debug_profile: HIP events around every op of the program), with the
op's FLOPs and activation bytes -> TFLOP/s and GB/s per op.  python scripts/voc_profile.py [--batch 32] [--exact 1]"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--exact", type=int, default=1)
    ap.add_argument("--full", type=int, default=1, help="0: the convolutional trunk alone")
    ap.add_argument("--wgs", type=int, default=-1, help="voc_set_max_workgroups (persistent tile loop); -1 = library default")
    ap.add_argument("--cache", default=os.environ.get("Q3_BENCH_CACHE", "/tmp/q3_bench_cache"))
    a = ap.parse_args()
    from qwen3_tts_axera_russian_amd import hiplib
    from qwen3_tts_axera_russian_amd import weights as W
    lib = hiplib.load()
    lib.voc_debug_profile.restype = ctypes.c_int
    lib.voc_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_int, hiplib.f32p, ctypes.c_int]
    vc = W.VocConfig() if a.full else W.trunk_voc_config()
    os.makedirs(a.cache, exist_ok=True)
    path = os.path.join(a.cache, f"voc_prof_full{a.full}.q3w")
    if not os.path.exists(path):
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(vc, seed=1234))
    prog, _ = W.voc_program(vc)
    lib.voc_set_exact_fp32(a.exact)
    if a.wgs >= 0:
        lib.voc_set_max_workgroups(a.wgs)
    h = lib.voc_load(path.encode(), 64, a.batch)
    assert h
    B = a.batch
    codes = np.random.default_rng(0).integers(0, 2048, size=(B, 64, 16)).astype(np.int64)
    out = np.empty((B, lib.voc_chunk_samples(h)), np.float32)
    for _ in range(2):
        assert lib.voc_decode(h, codes.ctypes.data_as(hiplib.i64p), B, hiplib.fptr(out)) == 0
    total_ms = lib.voc_last_decode_ms(h)
    ms = np.zeros(len(prog), np.float32)
    n = lib.voc_debug_profile(h, B, hiplib.fptr(ms), len(prog))
    assert n == len(prog)
    L, C = 64, 0
    print(f"B={B} exact={a.exact}: whole decode {total_ms:.2f} ms; per-op (profile mode, serialized): {ms.sum():.2f} ms")
    tot_fl = 0.0
    for i, row in enumerate(prog):
        op, p1, p2, p3, p4, fl = row[:6]
        flops = byt = 0.0
        name = {1: "rvq", 2: "conv", 3: "convT", 4: "dwconv", 5: "norm", 6: "attn", 7: "glu"}[op]
        if op == 1:
            C = p4
        elif op == 2:
            flops = 2.0 * p1 * p2 * p3 * L * B
            byt = 4.0 * (p1 + p2) * L * B + (4.0 * p2 * L * B if fl & 2 else 0)
            name += f" {p1}->{p2} k{p3} d{p4}" + (" +res" if fl & 2 else "") + (" snake" if fl & 1 else "")
            C = p2
        elif op == 3:
            flops = 2.0 * p1 * p2 * p3 * L * B          # k taps per input column
            byt = 4.0 * (p1 * L + p2 * L * p4) * B
            name += f" {p1}->{p2} k{p3} s{p4}"
            L = (L - 1) * p4 + p3 - row[6] - row[7]     # kept outputs after the table's trims
            C = p2
        else:
            byt = 8.0 * C * L * B
        tot_fl += flops
        t = float(ms[i])
        print(f"  op{i:3d} {name:32s} L={L:6d} {t:7.3f} ms  {flops / 1e9:8.1f} GF {flops / (t * 1e-3) / 1e12 if t > 0 else 0:6.1f} TF/s "
              f"{byt / 1e6:8.1f} MB {byt / (t * 1e-3) / 1e9 if t > 0 else 0:7.0f} GB/s")
    print(f"  total {tot_fl / 1e12:.2f} TFLOP -> {tot_fl / (total_ms * 1e-3) / 1e12:.1f} TF/s over the whole decode")
    lib.voc_free(h)


if __name__ == "__main__":
    main()
