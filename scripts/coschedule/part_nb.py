#!/usr/bin/env python3
"""Synthetic neighbour in its own (CU-masked) process: runs case argv[1] for ~1 s in a loop, prints timing."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nb = ctypes.CDLL(os.path.join(ROOT, "scripts", "coschedule", "libnb.so"))
nb.nb_init.argtypes = [ctypes.c_size_t, ctypes.c_int]
nb.nb_run.restype = ctypes.c_float
nb.nb_run.argtypes = [ctypes.c_int] * 9 + [ctypes.c_uint]
assert nb.nb_init(1 << 30, 0) == 0
cases = {"A": (1, 128, 16384, 100, 1, 37500, 0, 0, 0, 0),          # 128 wg x 1 ms, dense MFMA (one per CU of a 128-CU half)
         "N2": (1, 128, 16384, 100, 300, 64, 20, 0, 0, 0),          # 50 % MFMA duty
         "C": (1, 10240, 51200, 34, 1, 1406, 0, 0, 0, 0),           # refilling grid, dense MFMA
         "F": (1, 128, 16384, 100, 73, 512, 0, 8, 0, 0),            # MFMA + global loads
         "W": (1, 128, 16384, 100, 73, 512, 0, 8, 1, 0),            # MFMA + global loads + stores
         "S": (1, 384, 16384, 100, 400, 0, 0, 8, 1, 0)}             # streaming loads + stores only (memory-bound)
args = cases[sys.argv[1]]
alone = nb.nb_run(*args)
open("/tmp/voc_ready", "w").write("1")
while not os.path.exists("/tmp/frame_ready"): time.sleep(0.005)
t0 = time.time()
ms = [nb.nb_run(*args) for _ in range(int(sys.argv[2]))]
print(f"NB {sys.argv[1]} mask={os.environ.get('ROC_GLOBAL_CU_MASK')}: first {alone:.0f} ms, then {[round(x) for x in ms]}; wall [{t0:.3f},{time.time():.3f}]", flush=True)
