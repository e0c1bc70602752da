"""Workload of the r03 PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs) over the frame step's
weight-streaming kernels, through the kernel-level test hook q3t_bench_linear (96 timed launches per shape):
  * gate/up + SwiGLU (12.58 MB of fp16 weights, the dominant kernel by bytes) as the TALKER runs it -- 48 distinct weight
    copies = cold weights like the 28-layer walk, non-temporal weight loads (kernel name ...<2, 2, 4, 8, 1, 2, true>);
  * the same kernel as the CODE PREDICTOR runs it inside the frame graph -- 5 weight copies (its five layers: 63 MB that
    stay in the 256 MB Infinity Cache between the 15 passes of a frame), default-policy loads (...<2, 2, 4, 8, 1, 2, false>);
  * q|k|v at 32 rows, and the N = 1024 projections of round 3's own kernel (linear_narrow_kernel: o K = 2048, down K = 3072)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qwen3_tts_axera_russian_amd import hiplib  # noqa: E402

lib = hiplib.load_test()
for M in (32, 1):
    us = lib.q3t_bench_linear(M, 6144, 1024, 1, 2, 1, 48, 96)
    print(f"gateup talker (nt, 48 copies) M={M}: {us:.2f} us/launch", flush=True)
us = lib.q3t_bench_linear(32, 6144, 1024, 1, 2, 0, 5, 96)
print(f"gateup code predictor (cache-resident, 5 copies) M=32: {us:.2f} us/launch", flush=True)
us = lib.q3t_bench_linear(32, 4096, 1024, 1, 0, 1, 48, 96)
print(f"qkv M=32: {us:.2f} us/launch", flush=True)
us = lib.q3t_bench_linear(32, 1024, 2048, 0, 1, 1, 48, 96)
print(f"o (narrow) M=32: {us:.2f} us/launch", flush=True)
us = lib.q3t_bench_linear(32, 1024, 3072, 0, 1, 1, 48, 96)
print(f"down (narrow) M=32: {us:.2f} us/launch", flush=True)
