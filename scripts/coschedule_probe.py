#!/usr/bin/env python3
"""Why the step time is frame loop + vocoder: the replayed frame graph beside a second stream of kernels.

Host-clock timelines of (a) synthetic neighbours whose only difference is the share of time their waves spend issuing
MFMAs (scripts/coschedule/nb.hip: 0 / 10 / 50 / 100 %, 256 resident workgroups of 4 waves or a refilling grid of
20 480), (b) the real vocoder (exact fp32 and split), each started together with a 64-frame run of the frame loop at
32 rows.  Prints when each finished and the serial sum; `frames while the neighbour ran` is the frame loop's progress
during the overlap.  Run on the GPU box:  python scripts/coschedule_probe.py  (builds libnb.so next to nb.hip)."""
import ctypes
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from qwen3_tts_axera_russian_amd import hiplib  # noqa: E402
from qwen3_tts_axera_russian_amd.engine import FrameEngine  # noqa: E402


def build_nb():
    d = os.path.join(ROOT, "scripts", "coschedule")
    so = os.path.join(d, "libnb.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(d, "nb.hip")):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(d, "nb.hip"),
                               "-o", so, "-Wno-unused-result", "-Wno-unused-value"])
    nb = ctypes.CDLL(so)
    nb.nb_init.argtypes = [ctypes.c_size_t, ctypes.c_int]
    nb.nb_run.restype = ctypes.c_float
    nb.nb_run.argtypes = [ctypes.c_int] * 9 + [ctypes.c_uint]
    nb.nb_launch.argtypes = [ctypes.c_int] * 9 + [ctypes.c_uint]
    nb.nb_wait.restype = ctypes.c_float
    nb.nb_wait.argtypes = [ctypes.c_int]
    assert nb.nb_init(1 << 28, 0) == 0
    return nb


def main():
    lib = hiplib.load()
    nb = build_nb()
    B, F = 32, 64
    prefixes, n_text, pad = bench.workload(B, 0, 1234, 1)
    path, _ = bench.make_pack(os.environ.get("Q3_BENCH_CACHE", "/tmp/q3_bench_cache"), 1234, 0, lambda: None)
    n_ctx = max(p.shape[0] for p in prefixes) + F + 8
    eng = FrameEngine(path, max_batch=B, n_ctx=n_ctx, max_frames=F)
    eng.set_pad_embed(pad)
    for _ in range(2):
        eng.start(prefixes, n_text, ignore_eos=True, max_frames=F)
        eng.run(F)
    base = eng.last_run_ms
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}; frame loop alone: {base:.0f} ms for {F} frames "
          f"({base / F:.3f} ms/frame)", flush=True)
    #        name                               NA grid   LDS    launches iters nmfma nlds nld store footprint
    cases = [("256 wg, LDS+VALU only (0 % MFMA)", 1, 256, 16384, 200, 40, 0, 256, 0, 0, 0),
             ("256 wg, 10 % MFMA", 1, 256, 16384, 200, 60, 32, 100, 0, 0, 0),
             ("256 wg, 50 % MFMA", 1, 256, 16384, 200, 300, 64, 20, 0, 0, 0),
             ("256 wg, 100 % MFMA, 1 ms kernels", 1, 256, 16384, 200, 1, 37500, 0, 0, 0, 0),
             ("20480 wg (3/CU by LDS), 100 % MFMA", 1, 20480, 51200, 67, 1, 1406, 0, 0, 0, 0),
             ("20480 wg, 132 VGPRs, 100 % MFMA", 4, 20480, 51200, 67, 1, 352, 0, 0, 0, 0),
             ("20480 wg, LDS+VALU only", 1, 20480, 51200, 67, 2, 0, 256, 0, 0, 0)]
    for c in cases:
        name, args = c[0], c[1:]
        alone = nb.nb_run(*args)
        eng.start(prefixes, n_text, ignore_eos=True, max_frames=F)
        t0 = time.perf_counter()
        nb.nb_launch(*args)          # asynchronous: nobody blocks in a HIP wait while the frame loop is submitted
        eng.run(F)
        f1 = (time.perf_counter() - t0) * 1e3
        ms = nb.nb_wait(0)
        n1 = (time.perf_counter() - t0) * 1e3
        during = max(0.0, F - max(0.0, f1 - ms) / (base / F))
        print(f"{name:38s}: neighbour alone {alone:4.0f} ms, beside {ms:4.0f} | frame loop done at {f1:4.0f} ms | both done "
              f"{max(f1, n1):4.0f} ms (serial sum {alone + base:4.0f}) | frames while the neighbour ran: {during:4.1f} "
              f"({during * base / F / max(ms, 1e-3) * 100:3.0f} % of full speed)", flush=True)
    voc_path = bench.make_voc_pack(os.environ.get("Q3_BENCH_CACHE", "/tmp/q3_bench_cache"), 1234, 0, lambda: None)
    codes = np.random.default_rng(0).integers(0, 2048, size=(F, B, 16)).astype(np.int32)
    for exact in (1, 0):
        lib.voc_set_exact_fp32(exact)
        voc = bench.Vocoder(lib, voc_path, B)
        voc.decode(codes)
        voc.decode(codes)
        alone = voc.ms[-1]
        eng.start(prefixes, n_text, ignore_eos=True, max_frames=F)
        r = {}
        t0 = time.perf_counter()

        def w():
            voc.decode(codes)
            voc.decode(codes)
            r["ms"] = voc.ms[-2:]
            r["t1"] = (time.perf_counter() - t0) * 1e3

        t = threading.Thread(target=w)
        t.start()
        time.sleep(0.005)
        eng.run(F)
        f1 = (time.perf_counter() - t0) * 1e3
        t.join()
        during = max(0.0, F - max(0.0, f1 - r["t1"]) / (base / F))
        print(f"vocoder {'exact fp32' if exact else 'split f16x2'} x 2 decodes     : alone {alone:5.1f} ms each, beside "
              f"{[round(x, 1) for x in r['ms']]} done at {r['t1']:4.0f} ms | frame loop done at {f1:4.0f} ms (serial sum "
              f"{base + 2 * alone:4.0f}) | frames while the vocoder ran: {during:4.1f}", flush=True)
        voc.close()


if __name__ == "__main__":
    main()
