#!/usr/bin/env python3
"""which pause in a dense MFMA stream lets another queue's kernel chain make progress?"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def load(name):
    nb = ctypes.CDLL(os.path.join(ROOT, "scripts", "coschedule", name))
    nb.nb_init.argtypes = [ctypes.c_size_t, ctypes.c_int]
    nb.nb_run.restype = ctypes.c_float
    nb.nb_run.argtypes = [ctypes.c_int] * 9 + [ctypes.c_uint]
    nb.nb_launch.argtypes = [ctypes.c_int] * 9 + [ctypes.c_uint]
    nb.nb_wait.restype = ctypes.c_float
    nb.nb_wait.argtypes = [ctypes.c_int]
    assert nb.nb_init(1 << 28, 0) == 0
    return nb
n1, n2 = load("libnb.so"), load("libnb2.so")
short_valu = (1, 256, 16384, 5000, 1, 0, 12, 0, 0, 0)
tv = n2.nb_run(*short_valu)
kinds = {0: "none", 1: "8 x s_nop 15 (128 cyc)", 2: "s_sleep 1", 3: "s_sleep 4", 4: "v_mov", 5: "LDS read", 6: "16 x s_nop 15 (256 cyc)"}
for grid in (256, 1024):
    for every, kind in ((0, 0), (16, 1), (16, 6), (16, 2), (16, 3), (64, 3), (16, 4), (16, 5), (4, 2), (4, 5)):
        n1.nb_set_gap(every, kind)
        nbr = (1, grid, 16384, 1, 100 if grid == 256 else 25, 37500, 0, 0, 0, 0)
        ta = n1.nb_run(*nbr)
        T0 = time.perf_counter()
        n1.nb_launch(*nbr); n2.nb_launch(*short_valu)
        eb = n2.nb_wait(0); wb = (time.perf_counter() - T0) * 1e3
        ea = n1.nb_wait(0); wa = (time.perf_counter() - T0) * 1e3
        print(f"neighbour {grid} wg dense MFMA, pause every {every:2d} MFMAs: {kinds[kind]:24s}: alone {ta:4.0f} ms | victim (5000 short kernels, {tv:.0f} ms alone) done at {wb:4.0f} ms, neighbour at {wa:4.0f} ms", flush=True)
