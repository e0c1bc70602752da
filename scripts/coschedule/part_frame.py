#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from qwen3_tts_axera_russian_amd import hiplib
from qwen3_tts_axera_russian_amd.engine import FrameEngine
lib = hiplib.load()
B, F = 32, 64
prefixes, n_text, pad = bench.workload(B, 0, 1234, 1)
path, cfg = bench.make_pack("/tmp/q3_bench_cache", 1234, 0, lambda: None)
n_ctx = max(p.shape[0] for p in prefixes) + F + 8
eng = FrameEngine(path, max_batch=B, n_ctx=n_ctx, max_frames=F)
eng.set_pad_embed(pad)
for _ in range(2):
    eng.start(prefixes, n_text, ignore_eos=True, max_frames=F); eng.run(F)
alone = eng.last_run_ms / F
open("/tmp/frame_ready", "w").write("1")
while not os.path.exists("/tmp/voc_ready"): time.sleep(0.005)
time.sleep(0.05)
res = []
t0 = time.time()
for _ in range(int(sys.argv[1])):
    eng.start(prefixes, n_text, ignore_eos=True, max_frames=F); eng.run(F)
    res.append((round(eng.last_run_ms / F, 3), round(eng.last_prefill_ms, 1)))
print(f"FRAME mask={os.environ.get('ROC_GLOBAL_CU_MASK')}: alone {alone:.3f} ms/frame; beside the vocoder process (ms/frame, prefill ms): {res}; wall [{t0:.3f},{time.time():.3f}]", flush=True)
