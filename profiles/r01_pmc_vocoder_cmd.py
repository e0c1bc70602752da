"""Workload of the vocoder PMC passes: two decodes of 32 chunks with the default (split) arithmetic.
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python profiles/r01_pmc_vocoder_cmd.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qwen3_tts_axera_russian_amd import hiplib, weights as W

lib = hiplib.load()
os.makedirs("/tmp/q3", exist_ok=True)
path = "/tmp/q3/voc_full.q3w"
if not os.path.exists(path):
    W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(W.VocConfig(), seed=1234))
B = 32
h = lib.voc_load(path.encode(), 64, B)
codes = np.random.default_rng(0).integers(0, 2048, size=(B, 64, 16)).astype(np.int64)
out = np.empty((B, 64 * 1920), np.float32)
for _ in range(2):
    assert lib.voc_decode(h, codes.ctypes.data_as(hiplib.i64p), B, hiplib.fptr(out)) == 0
print("decode ms", lib.voc_last_decode_ms(h), flush=True)
lib.voc_free(h)
