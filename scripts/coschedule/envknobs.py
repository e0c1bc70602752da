#!/usr/bin/env python3
"""Frame loop alone (B=32 and B=1), ms per frame, under whatever HIP/ROCclr env knobs the caller set."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from qwen3_tts_axera_russian_amd import hiplib
from qwen3_tts_axera_russian_amd.engine import FrameEngine
lib = hiplib.load()
F = 64
out = []
for B in (32, 1):
    prefixes, n_text, pad = bench.workload(B, 0, 1234, 1)
    path, cfg = bench.make_pack("/tmp/q3_bench_cache", 1234, 0, lambda: None)
    n_ctx = max(p.shape[0] for p in prefixes) + F + 8
    eng = FrameEngine(path, max_batch=B, n_ctx=n_ctx, max_frames=F)
    eng.set_pad_embed(pad)
    best = 1e9
    for _ in range(4):
        eng.start(prefixes, n_text, ignore_eos=True, max_frames=F); eng.run(F)
        best = min(best, eng.last_run_ms / F)
    out.append(f"B={B}: {best:.3f} ms/frame, prefill {eng.last_prefill_ms:.2f} ms")
    eng.destroy()
print(os.environ.get("KNOBS", ""), "|", "; ".join(out), flush=True)
