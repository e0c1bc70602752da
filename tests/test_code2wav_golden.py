"""The vocoder's op table, converter and CPU oracle against the one importable implementation of the reference's
vocoder family: transformers' Qwen3OmniMoeCode2Wav (+ Mimi's split RVQ), whose outputs on seeded weights are stored in
tests/golden/code2wav_golden.npz by tests/golden/make_code2wav_golden.py (the reference itself holds no
vocoder code, only `decoder(codes)`: scripts/export_vocoder_traced.py:38-52).

CPU:  weights.state_to_voc(state_dict keys of those classes) -> oracle/voc_ref.py == the stored stages and waveform, 1e-5.
GPU:  tests/test_gpu_vocoder.py::test_code2wav_golden_* (voc_decode on the same tables, 2e-4)."""
import json
import os

import numpy as np
import pytest

from qwen3_tts_axera_russian_amd import weights as W
from tests import c2w_common as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "code2wav_golden.npz")
TOL = 1e-5    # fp32 round-off between two fp32 evaluations of the same graph (measured: <= 2e-6)


def load_case(name):
    """-> (case, VocConfig, table tensors, codes, golden arrays of the case)"""
    g = np.load(GOLD)
    case = C.CASES[name]
    key_shapes = [(k, tuple(s)) for k, s in json.loads(bytes(g[f"{name}.keys"]).decode())]
    state = C.seeded_state(case["seed"], key_shapes)
    assert C.digest(state) == bytes(g[f"{name}.sha"]).decode(), "seeded weights differ from the ones the fixture was made with"
    c2w = case["c2w"]
    cfgj = {"num_quantizers": c2w["num_quantizers"], "num_attention_heads": c2w["num_attention_heads"],
            "sliding_window": c2w["sliding_window"], "rms_norm_eps": c2w["rms_norm_eps"], "rope_theta": 10000,
            "dilations": [1, 3, 9]}
    vc, tens, report = W.state_to_voc(state, cfgj)
    gold = {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + ".")}
    return case, vc, tens, g[f"{name}.codes"], gold, report


@pytest.mark.parametrize("name", list(C.CASES))
def test_converter_reads_the_sizes_of_the_importable_classes(name):
    case, vc, tens, codes, gold, report = load_case(name)
    c2w = case["c2w"]
    assert vc.convt_trim == "both"                       # the default: Qwen3OmniMoeCausalTransConvNet's trim
    assert vc.rates == tuple(c2w["upsample_rates"]) and vc.upsample_ratios == tuple(c2w["upsampling_ratios"])
    assert vc.decoder_dim == c2w["decoder_dim"] and vc.latent == c2w["hidden_size"]
    assert vc.pre_transformer_layers == c2w["num_hidden_layers"] and vc.tf_heads == c2w["num_attention_heads"]
    assert vc.tf_ffn == c2w["intermediate_size"] and vc.tf_window == c2w["sliding_window"]
    assert vc.tf_attn_bias == c2w["attention_bias"] and vc.convnext and vc.n_q == 16
    if case["kind"] == "omni":
        assert vc.front == "embed" and not vc.pre_conv and not vc.tf_proj and vc.codebook_size == c2w["codebook_size"]
    else:
        assert vc.front == "rvq" and vc.pre_conv and vc.tf_proj and vc.tf_hidden == case["tf_hidden"]
        assert (vc.codebook_size, vc.codebook_dim, vc.rvq_out) == (case["rvq"]["codebook_size"], case["rvq"]["codebook_dim"],
                                                                   case["rvq"]["hidden"])
    # what one decode of T frames returns: (L - 1) * s per k = 2s block, not T * total_upsample
    assert W.voc_chunk_samples(vc, case["T"]) == gold["wav"].shape[0]
    assert W.voc_chunk_samples(vc, case["T"]) < case["T"] * W.voc_total_upsample(vc)
    assert "trimmed 'both'" in report[0]


@pytest.mark.parametrize("name", list(C.CASES))
def test_oracle_reproduces_every_stage_and_the_waveform(name):
    from oracle.voc_ref import voc_reference
    case, vc, tens, codes, gold, _ = load_case(name)
    prog = tens["voc.program"]
    worst = {}
    for stage, n_ops in C.stage_ops(vc, prog).items():
        act = voc_reference(tens, codes, n_ops=n_ops)[0]                     # [C][L]
        want = gold[stage]
        got = act[:, C.column_subset(act.shape[1])]
        assert got.shape == want.shape, (stage, act.shape, want.shape)
        err = float(np.abs(got - want).max() / max(1.0, float(np.abs(want).max())))
        worst[stage] = err
        assert err <= TOL, f"{name}: stage {stage} (after {n_ops} ops) differs by {err:.2e}"
    wav = voc_reference(tens, codes)[0]
    assert wav.shape == gold["wav"].shape
    err = float(np.abs(wav - gold["wav"]).max())
    assert err <= TOL, f"{name}: waveform differs by {err:.2e} (stages: {worst})"
    assert float(np.abs(gold["wav"]).max()) > 0.05 and float(np.abs(gold["wav"]).max()) <= 1.0   # a live, clamped signal


def test_right_trim_is_the_other_table_and_differs():
    """`convt_trim="right"` (rounds 1-2: strictly causal, L * s outputs) stays available as a table parameter; it is a
    different decoder: longer output, and already the first decoder block disagrees with the Omni class."""
    from oracle.voc_ref import voc_reference
    g = np.load(GOLD)
    case = C.CASES["omni"]
    key_shapes = [(k, tuple(s)) for k, s in json.loads(bytes(g["omni.keys"]).decode())]
    state = C.seeded_state(case["seed"], key_shapes)
    vc, tens, report = W.state_to_voc(state, {"num_quantizers": 16, "num_attention_heads": 4, "sliding_window": 5,
                                              "rms_norm_eps": 1e-5, "dilations": [1, 3, 9]}, convt_trim="right")
    assert vc.convt_trim == "right" and W.voc_chunk_samples(vc, case["T"]) == case["T"] * 1920
    wav = voc_reference(tens, g["omni.codes"])[0]
    assert wav.shape[0] == case["T"] * 1920 != g["omni.wav"].shape[0]
    rows = [r for r in tens["voc.program"] if r[0] == W.VOP_CONVT]
    assert all(r[6] == 0 and r[7] == r[3] - r[4] for r in rows)
    vb, tb, _ = W.state_to_voc(state, {"num_quantizers": 16, "num_attention_heads": 4, "sliding_window": 5,
                                       "rms_norm_eps": 1e-5, "dilations": [1, 3, 9]})
    assert all(r[6] == r[7] == r[3] - r[4] for r in tb["voc.program"] if r[0] == W.VOP_CONVT)
