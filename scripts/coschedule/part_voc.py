#!/usr/bin/env python3
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from qwen3_tts_axera_russian_amd import hiplib
lib = hiplib.load()
B, F = 32, 64
voc_path = bench.make_voc_pack("/tmp/q3_bench_cache", 1234, 0, lambda: None)
lib.voc_set_exact_fp32(1)
if len(sys.argv) > 2: lib.voc_set_max_workgroups(int(sys.argv[2]))
voc = bench.Vocoder(lib, voc_path, B)
codes = np.random.default_rng(0).integers(0, 2048, size=(F, B, 16)).astype(np.int32)
voc.decode(codes)
open("/tmp/voc_ready", "w").write("1")
while not os.path.exists("/tmp/frame_ready"): time.sleep(0.005)
n = int(sys.argv[1])
t0 = time.time()
for i in range(n):
    voc.decode(codes)
print(f"VOC wgs={sys.argv[2] if len(sys.argv)>2 else 0} mask={os.environ.get('ROC_GLOBAL_CU_MASK')}: {n} decodes, ms each: {[round(x) for x in voc.ms[1:]]}; wall [{t0:.3f},{time.time():.3f}]", flush=True)
