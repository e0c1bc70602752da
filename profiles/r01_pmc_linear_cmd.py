import sys, ctypes
sys.path.insert(0, '/root/repo')
from qwen3_tts_axera_russian_amd import hiplib
lib = hiplib.load()
lib.q3t_bench_linear.restype = ctypes.c_float
# gate/up (the largest weight stream of a layer) at 32 and 1 rows, 48 distinct weight copies (cold), nt loads
for M in (32, 1):
    us = lib.q3t_bench_linear(M, 6144, 1024, 1, 2, 1, 48, 96)
    print(f"gateup M={M}: {us:.2f} us/launch", flush=True)
us = lib.q3t_bench_linear(32, 4096, 1024, 1, 0, 1, 48, 96)
print(f"qkv M=32: {us:.2f} us/launch", flush=True)
