#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over the vocoder: 3 decodes of 32 chunks, exact fp32, whole table.
   cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc <counters> --output-format csv -d <dir> -- python3 scripts/voc_pmc_cmd.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qwen3_tts_axera_russian_amd import hiplib, weights as W  # noqa: E402

lib = hiplib.load()
vc = W.VocConfig()
path = "/tmp/voc_pmc.q3w"
if not os.path.exists(path):
    W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(vc, seed=1234))
lib.voc_set_exact_fp32(int(os.environ.get("VOC_EXACT", "1")))
B = 32
h = lib.voc_load(path.encode(), 64, B)
assert h
codes = np.random.default_rng(0).integers(0, 2048, size=(B, 64, 16)).astype(np.int64)
out = np.empty((B, lib.voc_chunk_samples(h)), np.float32)
for _ in range(3):
    assert lib.voc_decode(h, codes.ctypes.data_as(hiplib.i64p), B, hiplib.fptr(out)) == 0
print("decode ms", lib.voc_last_decode_ms(h))
