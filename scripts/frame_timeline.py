#!/usr/bin/env python3
"""In-graph per-node timing of ONE replayed frame step (the hipGraph of csrc/q3_engine.hip), from the diagnostic
timeline build of the library (lib/libqwen3tts_tl.so: every kernel stamps the 100 MHz device wall clock per workgroup
at entry and exit; `python -m qwen3_tts_axera_russian_amd.build --timeline`).  rocprofv3's kernel trace cannot see
inside a graph replay on this stack, and an eager run is not the chain the benchmark times; this is.

    python scripts/frame_timeline.py [--batch 32] [--csv profiles/rNN_frame_nodes_b32.csv] [--json]

Per node: kind, workgroups, span (first workgroup entry -> last workgroup exit), gap (previous node's last exit ->
this node's first entry = the dispatch boundary).  Per kind: count, mean span, mean gap, and for the weight-streaming
linears the algorithmic bytes per launch (weights + fp16 activations in + outputs) over the mean span
-> GB/s.  --json prints one line for bench.py (which runs this as a child process, outside its timed region).
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TL = os.path.join(ROOT, "qwen3_tts_axera_russian_amd", "lib", "libqwen3tts_tl.so")
MAXB = 1024

KIND = {10: "linear f16/store (talker head)", 11: "linear f16/resid", 14: "linear norm/store", 16: "linear norm/swiglu (gate+up)",
        30: "attn fused (talker)", 31: "attn prep", 32: "attn attend", 33: "attn short (code predictor)",
        40: "final norm", 42: "talker sample", 43: "cp argmax + gather"}


def classify(kind, blocks):
    if kind == 14:
        return "linear norm/store (q|k|v)" if blocks >= 200 else "linear norm/store (cp head)"
    if kind == 11:
        return "linear f16/resid (o / down)"
    return KIND.get(kind, f"kind {kind}")


def algorithmic_bytes(name, rows):
    """weights + activations read + outputs written by one launch (fp16 weights / activations, f32 residual)."""
    H, F = 1024, 3072
    if "gate+up" in name:
        return 2 * F * H * 2 + rows * H * 2 + rows * F * 2
    if "q|k|v" in name:
        return 4096 * H * 2 + rows * H * 2 + rows * 4096 * 4
    if "cp head" in name:
        return 2048 * H * 2 + rows * H * 2 + rows * 2048 * 4
    if "talker head" in name:
        return 3072 * H * 2 + rows * H * 2 + rows * 3072 * 4
    return None      # o / down share a kind (K differs): priced together below


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--csv", default=None)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--beside-vocoder", type=int, default=None, metavar="WGS",
                    help="record the frame while 32-chunk vocoder decodes run on a second thread with voc_set_max_workgroups(WGS) "
                         "(-1 = one workgroup per CU, 0 = one per tile): where the frame step's time goes beside the decode")
    ap.add_argument("--cache", default=os.environ.get("Q3_BENCH_CACHE", "/tmp/q3_bench_cache"))
    a = ap.parse_args()
    if not os.path.exists(TL):
        raise SystemExit(f"{TL} missing: python -m qwen3_tts_axera_russian_amd.build --timeline")
    os.environ["QWEN3TTS_LIB"] = TL
    import bench
    from qwen3_tts_axera_russian_amd import hiplib
    from qwen3_tts_axera_russian_amd.engine import FrameEngine
    lib = hiplib.load()
    lib.q3t_tl2_begin.argtypes = [ctypes.c_int]
    lib.q3t_tl2_end.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    lib.q3t_tl2_kinds.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    path, cfg = bench.make_pack(a.cache, 1234, 0, lambda: None)
    B = a.batch
    prefixes, n_text, pad = bench.workload(B, 0, 1234)
    eng = FrameEngine(path, max_batch=B, n_ctx=128, max_frames=64)
    eng.set_pad_embed(pad)
    eng.start(prefixes, n_text, ignore_eos=True, max_frames=64)
    eng.run(24)                     # eager frame + capture + replays: caches and clocks settled
    cap = 1400
    eng.start(prefixes, n_text, ignore_eos=True, max_frames=64)
    lib.q3t_tl2_begin(cap)
    # numbering is assigned at launch / capture time: this run re-captures nothing, so number the nodes by one eager
    # frame + capture of a fresh engine state instead: destroy the graph by switching the frame budget
    stop, th, voc = [False], None, None
    if a.beside_vocoder is not None:
        import threading
        import time
        lib.voc_set_exact_fp32(1)
        lib.voc_set_max_workgroups(a.beside_vocoder)
        voc = bench.Vocoder(lib, bench.make_voc_pack(a.cache, 1234, 0, lambda: None), 32)
        codes = np.random.default_rng(0).integers(0, 2048, size=(64, 32, 16)).astype(np.int32)

        def churn():
            while not stop[0]:
                voc.decode(codes)
        th = threading.Thread(target=churn, daemon=True)
        th.start()
        time.sleep(1.0)             # the first decode (LDS attributes, clocks) is over, one is running now
    eng.start(prefixes, n_text, ignore_eos=True, max_frames=63)
    eng.run(12)                     # frame 0 eager (nodes 0..n-1), capture (nodes n..2n-1), 10 replays
    if th is not None:
        stop[0] = True
        th.join()
        voc.close()
        lib.voc_set_max_workgroups(0)
    buf = (ctypes.c_ulonglong * (cap * MAXB * 2))()
    n = lib.q3t_tl2_end(buf, cap)
    kinds = (ctypes.c_int * cap)()
    lib.q3t_tl2_kinds(kinds, cap)
    arr = np.frombuffer(buf, dtype=np.uint64).reshape(cap, MAXB, 2).astype(np.int64)
    # numbered launches: the prefill of the second start(), the eager frame, then the captured frame; every frame begins
    # with the talker's sampling kernel (kind 42), so the captured graph's nodes start at its last occurrence
    half = max(i for i in range(n) if int(kinds[i]) == 42)
    rows = []
    prev_end = None
    for i in range(half, n):        # the captured graph's nodes hold the stamps of the LAST replay
        st_all, en_all = arr[i, :, 0], arr[i, :, 1]
        m = st_all > 0
        st, en, nb = int(st_all[m].min()), int(en_all[m].max()), int(m.sum())
        gap = (st - prev_end) / 100.0 if prev_end is not None else 0.0
        rows.append((i - half, classify(int(kinds[i]), nb), nb, (en - st) / 100.0, gap))
        prev_end = en
    frame_us = (prev_end - int(arr[half, :, 0][arr[half, :, 0] > 0].min())) / 100.0
    if a.csv:
        with open(a.csv, "w") as f:
            f.write("node,kind,workgroups,span_us,gap_before_us\n")
            for r in rows:
                f.write(f"{r[0]},\"{r[1]}\",{r[2]},{r[3]:.2f},{r[4]:.2f}\n")
    summary = {}
    for _, name, nb, span, gap in rows:
        s = summary.setdefault(name, [0, 0.0, 0.0])
        s[0] += 1
        s[1] += span
        s[2] += gap
    out = {"batch": B, "nodes": len(rows), "frame_us": round(frame_us, 1), "kinds": {}}
    for name, (cnt, sp, gp) in sorted(summary.items(), key=lambda kv: -kv[1][1]):
        ent = {"n": cnt, "mean_span_us": round(sp / cnt, 3), "mean_gap_before_us": round(gp / cnt, 3),
               "share_of_frame": round((sp + gp) / frame_us, 4)}
        ab = algorithmic_bytes(name, B)
        if ab:
            ent["algorithmic_bytes"] = ab
            ent["GBps_over_span"] = round(ab / (sp / cnt) / 1e3, 1)
        out["kinds"][name] = ent
    if a.json:
        print(json.dumps(out), flush=True)
    else:
        print(f"B={B}: {len(rows)} nodes, frame span {frame_us:.1f} us")
        for name, e in out["kinds"].items():
            extra = f"  {e['algorithmic_bytes'] / 1e6:6.2f} MB -> {e['GBps_over_span']:7.1f} GB/s" if "GBps_over_span" in e else ""
            print(f"  {name:34s} n={e['n']:4d}  span {e['mean_span_us']:6.2f} us  gap {e['mean_gap_before_us']:5.2f} us  "
                  f"{100 * e['share_of_frame']:5.1f} % of frame{extra}")
    eng.destroy()


if __name__ == "__main__":
    main()
