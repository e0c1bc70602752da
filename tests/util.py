"""Shared helpers of the GPU parity tests: synthetic weight packs and oracle construction."""
import os

import numpy as np

from qwen3_tts_axera_russian_amd import weights as W

CACHE = os.environ.get("Q3_TEST_CACHE", "/tmp/q3_test_cache")


def synthetic_pack(talker_layers=2, cp_layers=2, seed=1234, parts=("talker", "cp")):
    """Path of a (cached) synthetic container + its config + memmapped tensors."""
    os.makedirs(CACHE, exist_ok=True)
    cfg = W.ModelConfig(talker_layers=talker_layers, cp_layers=cp_layers)
    path = os.path.join(CACHE, f"synth_t{talker_layers}_c{cp_layers}_s{seed}_{'-'.join(parts)}.q3w")
    if not os.path.exists(path):
        W.write_synthetic(path, cfg, seed=seed, parts=parts)
    meta, tensors = W.read_pack(path)
    return path, W.ModelConfig.from_meta(meta), tensors


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
