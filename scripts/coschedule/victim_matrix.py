#!/usr/bin/env python3
"""victim = chain of short kernels, neighbour = one long kernel; each with / without MFMA."""
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
#                      na grid  lds   launches iters nmfma nlds nld st fp
long_mfma   = (1, 256, 16384, 1, 200, 37500, 0, 0, 0, 0)       # one ~200 ms kernel, 100 % MFMA
long_valu   = (1, 256, 16384, 1, 110, 0, 256 * 40, 0, 0, 0)    # one long kernel, LDS + VALU only
long_mfma64 = (1, 64, 16384, 1, 200, 37500, 0, 0, 0, 0)        # 64 workgroups only (a quarter of the CUs)
short_mfma  = (1, 256, 16384, 5000, 1, 750, 0, 0, 0, 0)        # 5000 x ~20 us kernels, MFMA
short_valu  = (1, 256, 16384, 5000, 1, 0, 12, 0, 0, 0)         # 5000 x short kernels, LDS + VALU only
short_mfma1 = (1, 256, 16384, 5000, 1, 8, 12, 0, 0, 0)         # short kernels, LDS + VALU + 8 MFMAs per wave (like a GEMV)
for (kn, nbr), (kv, vic) in ((("long MFMA", long_mfma), ("short MFMA", short_mfma)),
                             (("long MFMA", long_mfma), ("short VALU", short_valu)),
                             (("long MFMA", long_mfma), ("short VALU+8 MFMA", short_mfma1)),
                             (("long VALU", long_valu), ("short MFMA", short_mfma)),
                             (("long VALU", long_valu), ("short VALU", short_valu)),
                             (("long MFMA 64wg", long_mfma64), ("short VALU+8 MFMA", short_mfma1)),
                             (("long MFMA 64wg", long_mfma64), ("short VALU", short_valu))):
    ta, tb = n1.nb_run(*nbr), n2.nb_run(*vic)
    T0 = time.perf_counter()
    n1.nb_launch(*nbr); n2.nb_launch(*vic)
    eb = n2.nb_wait(0); wb = (time.perf_counter() - T0) * 1e3
    ea = n1.nb_wait(0); wa = (time.perf_counter() - T0) * 1e3
    print(f"neighbour [{kn}] + victim [{kv}]: alone {ta:.0f} / {tb:.0f} ms; together: neighbour done {wa:.0f} ms, victim done {wb:.0f} ms (serial sum {ta+tb:.0f})", flush=True)
