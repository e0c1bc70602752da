"""Seeded weights + inputs shared by tests/golden/make_hf_golden.py (which runs transformers' Qwen3Model on
them and stores its outputs) and tests/test_oracle_vs_hf.py (which runs oracle/q3_oracle.c on them).

The weights are NOT the all-ones-norm synthetic pack of the speed tests: every RMSNorm vector (input,
post-attention, per-head q/k, final) is random, so that a norm applied at the wrong place, to the wrong
operand, or before/after RoPE changes the result; projections are fp16-representable values (what the
device holds), so the HF model and the oracle see bit-identical weights."""
from __future__ import annotations

import hashlib

import numpy as np

from qwen3_tts_axera_russian_amd import weights as W

PARTS = ("input_ln", "q_proj", "k_proj", "v_proj", "o_proj", "q_norm", "k_norm", "post_ln", "gate_proj", "up_proj",
         "down_proj")


def hf_check_config(talker_layers=3, cp_layers=2) -> W.ModelConfig:
    return W.ModelConfig(talker_layers=talker_layers, cp_layers=cp_layers, cp_vocab=64)   # 64-entry CP tables suffice


def _stack(r, prefix, n_layers, cfg, ffn, t):
    H, D = cfg.hidden, cfg.head_dim
    NQ, NKV = cfg.n_heads * D, cfg.n_kv_heads * D

    def mat(n, k, std):
        return (std * r.standard_normal((n, k))).astype(np.float16).astype(np.float32)

    def gam(n):
        return (1.0 + 0.25 * r.standard_normal(n)).astype(np.float32)

    for i in range(n_layers):
        b = f"{prefix}.layers.{i}."
        t[b + "input_ln"] = gam(H)
        t[b + "q_proj"] = mat(NQ, H, 0.04)
        t[b + "k_proj"] = mat(NKV, H, 0.04)
        t[b + "v_proj"] = mat(NKV, H, 0.04)
        t[b + "o_proj"] = mat(H, NQ, 0.03)
        t[b + "q_norm"] = gam(D)
        t[b + "k_norm"] = gam(D)
        t[b + "post_ln"] = gam(H)
        t[b + "gate_proj"] = mat(ffn, H, 0.04)
        t[b + "up_proj"] = mat(ffn, H, 0.04)
        t[b + "down_proj"] = mat(H, ffn, 0.03)
    t[f"{prefix}.norm"] = gam(H)


def make_tensors(cfg: W.ModelConfig, seed=20261004) -> dict:
    """Tensor dict with the container's names (weights.py): talker + code-predictor stacks, tables, heads."""
    r = np.random.default_rng(seed)
    t: dict = {}
    _stack(r, "talker", cfg.talker_layers, cfg, cfg.talker_ffn, t)
    _stack(r, "cp", cfg.cp_layers, cfg, cfg.cp_ffn, t)
    H = cfg.hidden
    t["talker.codec_embedding"] = (0.05 * r.standard_normal((cfg.talker_vocab, H))).astype(np.float32)
    t["talker.codec_head"] = (0.05 * r.standard_normal((cfg.talker_vocab, H))).astype(np.float16).astype(np.float32)
    for g in range(cfg.cp_groups):
        t[f"cp.codec_emb.{g}"] = (0.05 * r.standard_normal((cfg.cp_vocab, H))).astype(np.float32)
        t[f"cp.lm_head.{g}"] = (0.05 * r.standard_normal((cfg.cp_vocab, H))).astype(np.float16).astype(np.float32)
    return t


def make_inputs(cfg: W.ModelConfig, seed=7, n_prefill=13, n_decode=5):
    r = np.random.default_rng(seed)
    H = cfg.hidden
    return dict(
        prefill=(0.6 * r.standard_normal((n_prefill, H))).astype(np.float32),
        decode=(0.6 * r.standard_normal((n_decode, H))).astype(np.float32),
        cp_hidden=(1.5 * r.standard_normal(H)).astype(np.float32),
        cp_code0=np.int32(1234),
        cp_forced=r.integers(0, cfg.cp_vocab, size=cfg.cp_groups).astype(np.int32),
    )


def digest(tensors: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(tensors):
        h.update(k.encode())
        h.update(np.ascontiguousarray(tensors[k]).tobytes())
    return h.hexdigest()
