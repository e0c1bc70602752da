#!/usr/bin/env python3
"""Ragged prefill of the benchmark's 32 prompts (971 rows) alone on the chip: q3e_start's GPU milliseconds, best of a few
(for whatever environment knobs the caller exported).  python scripts/ab_prefill.py [--label text]"""
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
    a = ap.parse_args()
    from qwen3_tts_axera_russian_amd.engine import FrameEngine
    path, cfg = bench.make_pack(a.cache, 1234, 0, lambda: None)
    prefixes, n_text, pad = bench.workload(32, 0, 1234)
    eng = FrameEngine(path, max_batch=32, n_ctx=max(p.shape[0] for p in prefixes) + 16, max_frames=8)
    eng.set_pad_embed(pad)
    ms = []
    for _ in range(6):
        eng.start(prefixes, n_text, ignore_eos=True, max_frames=8)
        ms.append(eng.last_prefill_ms)
    eng.run(2)
    codes = eng.codes()[0]
    eng.destroy()
    print(f"[{a.label}] prefill of {sum(p.shape[0] for p in prefixes)} rows: {min(ms[1:]):.3f} ms (runs {' '.join(f'{x:.3f}' for x in ms)}) codes-sum {int(codes.sum())}", flush=True)


if __name__ == "__main__":
    main()
