#!/usr/bin/env python3
"""Host-side breakdown of bench.py's pipelined step (where the wall time of a step goes besides the frame loop's GPU span):
start() = prefix concatenation + upload + ragged prefill, run() = 64 graph replays, codes() = download, the wait for the previous
step's decode, the hand-over of this step's decode.  GPU box:  python scripts/step_breakdown.py [--voc-wgs -1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voc-wgs", type=int, default=-1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--cache", default=os.environ.get("Q3_BENCH_CACHE", "/tmp/q3_bench_cache"))
    a = ap.parse_args()
    from concurrent.futures import ThreadPoolExecutor
    from qwen3_tts_axera_russian_amd import hiplib
    from qwen3_tts_axera_russian_amd.engine import FrameEngine
    lib = hiplib.load()
    B, F = 32, 64
    prefixes, n_text, pad = bench.workload(B, 0, 1234)
    path, cfg = bench.make_pack(a.cache, 1234, 0, lambda: None)
    eng = FrameEngine(path, max_batch=B, n_ctx=max(p.shape[0] for p in prefixes) + F + 8, max_frames=F)
    eng.set_pad_embed(pad)
    lib.voc_set_max_workgroups(a.voc_wgs)
    lib.voc_set_exact_fp32(1)
    voc = bench.Vocoder(lib, bench.make_voc_pack(a.cache, 1234, 0, lambda: None), B)
    pool = ThreadPoolExecutor(max_workers=1)
    pending = None
    rows = []
    for s in range(a.steps + 2):
        t = [time.perf_counter()]
        eng.start(prefixes, n_text, ignore_eos=True, max_frames=F); t.append(time.perf_counter())
        eng.run(F); t.append(time.perf_counter())
        codes, _ = eng.codes(); t.append(time.perf_counter())
        if pending is not None:
            pending.result()
        t.append(time.perf_counter())
        pending = pool.submit(voc.decode, codes.copy()); t.append(time.perf_counter())
        if s >= 2:
            rows.append([1e3 * (t[i + 1] - t[i]) for i in range(5)] + [eng.last_run_ms, eng.last_prefill_ms])
    pending.result()
    m = np.mean(np.array(rows), axis=0)
    print("per step, ms: start %.2f | run %.2f | codes %.2f | wait for the previous decode %.2f | hand-over %.2f | sum %.2f"
          % (m[0], m[1], m[2], m[3], m[4], m[:5].sum()))
    print("GPU spans, ms: frame loop %.2f (%.3f per frame) | prefill %.2f | decode beside it %.2f" % (m[5], m[5] / F, m[6], np.mean(voc.ms[2:])))
    voc.close()
    eng.destroy()


if __name__ == "__main__":
    main()
