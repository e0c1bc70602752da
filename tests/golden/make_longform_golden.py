#!/usr/bin/env python3
"""Generate tests/golden/longform_f768.npz: the CPU oracle's greedy trajectory of the benchmark's long-form leg --
BASELINE configs[4]: ONE utterance (utterance 0 of bench.workload(32, 0, 1234): 17 text tokens, 26 prefix rows), 768 frames
= 61.4 s of audio, EOS suppressed, at the real depth (28 talker + 5 code-predictor layers, synthetic weights seed 1234 =
bench.make_pack).  KV positions 26 ... 793: the talker attention far beyond the 64-frame regime of bench_b32_f64.npz.

    ids       int16   [768][16]   oracle/pipeline.py CpuPipeline.generate_batch (the loop of llamacpp_talker_server.py:254-293 /
                                  code_predictor_server.py:94-140 / tts_client.py:199-208 on oracle/q3_oracle.c)
    margins   float16 [768][16]   top-1 / top-2 gap of every decision's processed logits, clipped to 1.0
    inputs_sha  sha256 of the prefix, n_text and pad

Used by tests/test_gpu_longform.py: all 12 288 decisions graded teacher-forced under NEAR_TIE.  The first 64 frames must equal
utterance 0 of bench_b32_f64.npz (same inputs; checked here).

Usage:  python tests/golden/make_longform_golden.py      (CPU only; a few minutes on 8 cores)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle.pipeline import CpuPipeline  # noqa: E402
from tests.golden.make_bench_golden import inputs_sha  # noqa: E402
from tests.util import synthetic_pack  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "longform_f768.npz")
F, SEED = 768, 1234


def main():
    path, cfg, tensors = synthetic_pack(28, 5, seed=SEED)
    prefixes, n_text, pad = bench.workload(32, 0, SEED)
    prefixes, n_text = prefixes[:1], n_text[:1]
    cpu = CpuPipeline(cfg, tensors, n_ctx=prefixes[0].shape[0] + F + 1)
    t0 = time.time()
    frames, margins = cpu.generate_batch(prefixes, n_text, pad, F, ignore_eos=True)
    print(f"oracle: 1 utterance x {F} frames in {time.time() - t0:.0f} s", flush=True)
    assert len(frames[0]) == F
    ids = np.array(frames[0], np.int16)
    m = np.minimum(np.array(margins[0][:F], np.float64), 1.0).astype(np.float16)
    assert ids.shape == m.shape == (F, 16) and ids.min() >= 0 and ids.max() < 2048
    g = np.load(os.path.join(HERE, "bench_b32_f64.npz"))
    assert (g["ids"][0] == ids[:64]).all(), "the first 64 frames differ from utterance 0 of bench_b32_f64.npz"
    np.savez_compressed(OUT, ids=ids, margins=m, inputs_sha=np.frombuffer(inputs_sha(prefixes, n_text, pad).encode(), np.uint8),
                        seed=np.array(SEED), n_text=np.asarray(n_text, np.int32))
    print("wrote", OUT, os.path.getsize(OUT), "bytes; decisions with a gap < 5e-3:", int((m < 5e-3).sum()), "of", m.size)


if __name__ == "__main__":
    main()
