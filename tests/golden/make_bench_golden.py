#!/usr/bin/env python3
"""Generate tests/golden/bench_b32_f64.npz: the CPU oracle's greedy trajectory of the benchmark's own workload --
`bench.workload(32, 0, 1234)` (BASELINE configs[2]: 32 utterances, the fixed prompt-length set), 64 frames, EOS
suppressed, at the real depth (28 talker + 5 code-predictor layers, synthetic weights seed 1234 = bench.make_pack).

    ids       int16   [32][64][16]   oracle/pipeline.py CpuPipeline.generate_batch (the loop of
                                     llamacpp_talker_server.py:254-293 / code_predictor_server.py:94-140 /
                                     tts_client.py:199-208 on oracle/q3_oracle.c)
    margins   float16 [32][64][16]   top-1 / top-2 gap of every decision's processed logits, clipped to 1.0
                                     (only gaps below NEAR_TIE = 5e-3 matter; fp16 resolves 4e-6 there)
    inputs_sha  sha256 of the prefixes, n_text and pad the workload function returned (guards the regeneration)

Used by tests/test_gpu_engine.py (all 32 768 decisions graded teacher-forced; the free-running count asserted) and by
bench.py itself (`"verified"` in the JSON line: step 0's codes against this fixture up to each utterance's first near-tie).

Usage:  python tests/golden/make_bench_golden.py      (CPU only; ~20-30 min on 8 cores, 3 GB of memory)
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle.pipeline import CpuPipeline  # noqa: E402
from tests.util import synthetic_pack  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_b32_f64.npz")
B, F, SEED = 32, 64, 1234


def inputs_sha(prefixes, n_text, pad):
    h = hashlib.sha256()
    for p in prefixes:
        h.update(np.ascontiguousarray(p).tobytes())
    h.update(np.asarray(n_text, np.int32).tobytes())
    h.update(np.ascontiguousarray(pad).tobytes())
    return h.hexdigest()


def main():
    path, cfg, tensors = synthetic_pack(28, 5, seed=SEED)        # the tensors of bench.make_pack (same names, same seed)
    prefixes, n_text, pad = bench.workload(B, 0, SEED)
    cpu = CpuPipeline(cfg, tensors, n_ctx=max(p.shape[0] for p in prefixes) + F + 1)
    t0 = time.time()
    frames, margins = cpu.generate_batch(prefixes, n_text, pad, F, ignore_eos=True)
    print(f"oracle: {B} utterances x {F} frames in {time.time() - t0:.0f} s", flush=True)
    assert all(len(fr) == F for fr in frames)
    ids = np.array(frames, np.int16)                              # [B][F][16]
    m = np.minimum(np.array([[mm for mm in margins[b][:F]] for b in range(B)], np.float64), 1.0).astype(np.float16)
    assert ids.shape == m.shape == (B, F, 16) and ids.min() >= 0 and ids.max() < 2048
    np.savez_compressed(OUT, ids=ids, margins=m, inputs_sha=np.frombuffer(inputs_sha(prefixes, n_text, pad).encode(), np.uint8),
                        seed=np.array(SEED), n_text=np.asarray(n_text, np.int32))
    print("wrote", OUT, os.path.getsize(OUT), "bytes; decisions with a gap < 5e-3:", int((m < 5e-3).sum()), "of", m.size)


if __name__ == "__main__":
    main()
