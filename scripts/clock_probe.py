import ctypes, os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.getcwd())
from qwen3_tts_axera_russian_amd import hiplib, weights as W
lib = hiplib.load()
vc = W.VocConfig()
path = "/tmp/voc_clk.q3w"
W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(vc, seed=1234))
lib.voc_set_exact_fp32(1)
B = 32
h = lib.voc_load(path.encode(), 64, B)
codes = np.random.default_rng(0).integers(0, 2048, size=(B, 64, 16)).astype(np.int64)
out = np.empty((B, lib.voc_chunk_samples(h)), np.float32)
stop = False
samples = []
def poll():
    while not stop:
        r = subprocess.run(["rocm-smi", "-c", "-P", "--showperflevel"], capture_output=True, text=True)
        samples.append(r.stdout)
        time.sleep(0.3)
r0 = subprocess.run(["rocm-smi", "-c", "-P"], capture_output=True, text=True).stdout
print("IDLE:\n", "\n".join(l for l in r0.splitlines() if "sclk" in l or "Power" in l or "mclk" in l))
t = threading.Thread(target=poll); t.start()
t0 = time.time(); ms = []
while time.time() - t0 < 8:
    assert lib.voc_decode(h, codes.ctypes.data_as(hiplib.i64p), B, hiplib.fptr(out)) == 0
    ms.append(lib.voc_last_decode_ms(h))
stop = True; t.join()
print("decode ms first/median/last:", ms[0], float(np.median(ms)), ms[-1], len(ms))
for s in samples[2::6]:
    print(" | ".join(l.strip() for l in s.splitlines() if "sclk" in l or "Power" in l))
