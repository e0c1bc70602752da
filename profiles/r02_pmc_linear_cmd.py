"""Workload of the r02 PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs): the dominant
weight-streaming kernel of the frame step -- gate/up + SwiGLU (12.58 MB of fp16 weights) -- and q/k/v, at 32 rows and
at 1 row, through the kernel-level test hook (48 distinct weight copies = cold weights like the layer walk,
non-temporal weight loads, 96 launches per shape)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qwen3_tts_axera_russian_amd import hiplib  # noqa: E402

lib = hiplib.load_test()
for M in (32, 1):
    us = lib.q3t_bench_linear(M, 6144, 1024, 1, 2, 1, 48, 96)
    print(f"gateup M={M}: {us:.2f} us/launch", flush=True)
us = lib.q3t_bench_linear(32, 4096, 1024, 1, 0, 1, 48, 96)
print(f"qkv M=32: {us:.2f} us/launch", flush=True)
