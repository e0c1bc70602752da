"""Shared by tests/golden/make_code2wav_golden.py (which runs transformers' Qwen3OmniMoeCode2Wav / Mimi split RVQ on
seeded weights and stores their outputs) and tests/test_code2wav_golden.py (which rebuilds the same weights from the
seeds, maps them through weights.state_to_voc and runs oracle/voc_ref.py and the HIP vocoder on them).

Nothing here is model code: the cases' hyper-parameters, the seeded tensor generator (driven by the (key, shape) list
the torch modules report; the list is stored in the fixture so the test needs no transformers) and the map from a
stage name to the number of table ops that produce it."""
from __future__ import annotations

import hashlib

import numpy as np

# ---- the cases ----------------------------------------------------------------------------------------------------
# "omni":  Qwen3OmniMoeCode2Wav.forward(codes) as it is (embedding-mean front, no pre-conv, no projections), default
#          rates (8,5,4,3) x (2,2) = 1920, grouped-query attention (4 heads on 2 k/v heads), window shorter than T.
# "omni_b": other rates / ratios / kernel counts, attention biases, window = 3.
# "tts":   the Qwen3-TTS-Tokenizer composition (recollection): Mimi split RVQ (1 semantic + 15 acoustic,
#          EMA codebooks, output projections) -> causal conv k3 -> Linear in -> the Omni pre-transformer at a narrower
#          width -> Linear out -> the Omni upsample + decoder stacks.  Every module is transformers' own; only the
#          order they are called in is ours.
CASES = {
    "omni": dict(kind="omni", T=8, seed=101,
                 c2w=dict(codebook_size=48, hidden_size=32, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                          intermediate_size=40, sliding_window=5, decoder_dim=128, num_quantizers=16,
                          upsample_rates=(8, 5, 4, 3), upsampling_ratios=(2, 2), rms_norm_eps=1e-5, attention_bias=False)),
    "omni_b": dict(kind="omni", T=11, seed=202,
                   c2w=dict(codebook_size=20, hidden_size=48, num_hidden_layers=3, num_attention_heads=6, num_key_value_heads=6,
                            intermediate_size=64, sliding_window=3, decoder_dim=64, num_quantizers=16,
                            upsample_rates=(4, 3, 2), upsampling_ratios=(3,), rms_norm_eps=1e-6, attention_bias=True)),
    "tts": dict(kind="tts", T=9, seed=303, rvq=dict(codebook_size=40, codebook_dim=8, hidden=16, num_quantizers=16, num_semantic=1),
                pre_kernel=3, tf_hidden=16,
                c2w=dict(codebook_size=40, hidden_size=32, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                         intermediate_size=24, sliding_window=4, decoder_dim=128, num_quantizers=16,
                         upsample_rates=(8, 5, 4, 3), upsampling_ratios=(2, 2), rms_norm_eps=1e-5, attention_bias=False)),
}


def _rng(seed: int, key: str) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    return np.random.default_rng(int.from_bytes(h[:8], "little"))


def seeded_tensor(seed: int, key: str, shape) -> np.ndarray:
    """One tensor of a decoder state dict.  Every vector that scales or shifts something is random (SnakeBeta alpha /
    beta, layer scales, ConvNeXt gamma, norm weights and biases, cluster usage) so that a misplaced one changes the
    waveform; matrices keep activations O(1)."""
    r = _rng(seed, key)
    shape = tuple(int(x) for x in shape)
    n = lambda *a: r.standard_normal(shape).astype(np.float32)
    if key.endswith("initialized"):
        return np.ones(shape, np.float32)
    if key.endswith("cluster_usage"):
        u = (0.5 + r.random(shape)).astype(np.float32)
        u.flat[3 % u.size] = 1e-7            # below the clamp (epsilon 1e-5): embed = embed_sum / 1e-5 there
        return u
    if key.endswith("embed_sum"):
        e = (0.4 * n()).astype(np.float32)
        e[3 % e.shape[0]] *= 1e-5            # the row whose usage is clamped stays O(1) after the division
        return e
    if key.endswith(("alpha", "beta")):
        return (0.3 * n()).astype(np.float32)
    if key.endswith("layer_scale.scale"):
        return (0.3 + 0.4 * r.random(shape)).astype(np.float32)
    if key.endswith("gamma"):
        return (0.5 + r.random(shape)).astype(np.float32)
    if key.endswith("code_embedding.weight"):
        return n()
    if key.endswith(("norm.weight", "layernorm.weight")):
        return (1.0 + 0.2 * n()).astype(np.float32)
    if key.endswith("bias"):
        return (0.1 * n() if "norm" in key else 0.05 * n()).astype(np.float32)
    if key.endswith("weight") and len(shape) >= 2:
        if len(shape) == 3 and ".block.1.conv" in key or ".0.conv.weight" in key and "upsample" in key:
            fan_in = shape[0] * 2 if ".block.1." in key else shape[0]          # ConvTranspose1d [cin, cout, k]: k / stride taps
        elif "dwconv" in key:
            fan_in = shape[2]
        else:
            fan_in = int(np.prod(shape[1:]))
        gain = 0.5 if ("conv2.conv" in key or "pwconv2" in key) else 0.9
        if ".block.1.conv" in key:
            gain = 0.6                         # the random stack stays contractive: O(1) activations down to the output
        if shape[0] == 1:
            gain = 0.1                         # the output conv: a waveform that mostly stays inside the clamp
        return (gain * n() / np.sqrt(fan_in)).astype(np.float32)
    raise KeyError(f"seeded_tensor: no rule for {key} {shape}")


def seeded_state(seed: int, key_shapes) -> dict:
    return {k: seeded_tensor(seed, k, shp) for k, shp in key_shapes}


def digest(state: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(np.ascontiguousarray(state[k]).tobytes())
    return h.hexdigest()


def seeded_codes(seed: int, T: int, size: int) -> np.ndarray:
    """codes int64 [1][T][16] (the layout vocoder_server.py:78-79 builds)"""
    return _rng(seed, f"codes{T}").integers(0, size, size=(1, T, 16)).astype(np.int64)


def stage_ops(vc, prog) -> dict:
    """stage name -> number of leading table ops whose output it is (weights.voc_program's order)."""
    from qwen3_tts_axera_russian_amd import weights as W
    st, i = {}, 1
    st["front"] = i
    if vc.pre_conv:
        i += 1
        st["pre_conv"] = i
    if vc.pre_transformer_layers:
        i += (1 if vc.tf_proj else 0) + 8 * vc.pre_transformer_layers + 1 + (1 if vc.tf_proj else 0)
        st["pre_transformer"] = i
    for u in range(len(vc.upsample_ratios)):
        i += 1 + (4 if vc.convnext else 0)
        st[f"upsample{u}"] = i
    i += 1
    st["dec_in"] = i
    for b in range(len(vc.rates)):
        i += 1 + 2 * len(vc.dilations)
        st[f"block{b}"] = i
    i += 1
    assert i == len(prog), (i, len(prog))
    for name, n in st.items():      # every stage ends on an op boundary the program knows
        assert 0 < n < len(prog)
    del W
    return st


def column_subset(L: int) -> np.ndarray:
    """Columns of a long activation that the fixture keeps: both ends + a sparse comb."""
    if L <= 96:
        return np.arange(L)
    return np.unique(np.concatenate([np.arange(40), np.arange(40, L - 40, 29), np.arange(L - 40, L)]))
