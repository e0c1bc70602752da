#!/usr/bin/env python3
"""A/B of the replayed frame graph: ms per frame step at B = 32 and B = 1 (bench.workload's prompts, 64 frames, EOS
suppressed), best of a few runs, for whatever environment knobs the caller exported (the library reads them at load:
one process per variant).  python scripts/ab_frame.py [--label text]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--label", default="")
    ap.add_argument("--cache", default=os.environ.get("Q3_BENCH_CACHE", "/tmp/q3_bench_cache"))
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    from qwen3_tts_axera_russian_amd.engine import FrameEngine
    path, cfg = bench.make_pack(a.cache, 1234, 0, lambda: None)
    res = []
    for B in (32, 1):
        prefixes, n_text, pad = bench.workload(32, 0, 1234)
        prefixes, n_text = prefixes[:B], n_text[:B]
        eng = FrameEngine(path, max_batch=B, n_ctx=max(p.shape[0] for p in prefixes) + 72, max_frames=64)
        eng.set_pad_embed(pad)
        ms = []
        for _ in range(a.reps):
            eng.start(prefixes, n_text, ignore_eos=True, max_frames=64)
            assert eng.run(64) == 64
            ms.append(eng.last_run_ms / 64)
        codes = eng.codes()[0]
        eng.destroy()
        res.append(f"B={B}: {min(ms[1:]):.4f} ms/frame (runs {' '.join(f'{x:.4f}' for x in ms)}) codes-sum {int(codes.sum())}")
    print(f"[{a.label}] " + "; ".join(res), flush=True)


if __name__ == "__main__":
    main()
