"""Q3TTSW1 container round trip and the oracle on the CPU (no GPU)."""
import numpy as np

from qwen3_tts_axera_russian_amd import weights as W


def test_pack_roundtrip(tmp_path):
    t = {"a.b": np.arange(12, dtype=np.float32).reshape(3, 4), "c": np.array([1, 2, 3], np.float16),
         "d": np.arange(6, dtype=np.int32).reshape(2, 3), "e": np.array([7], np.int64)}
    p = str(tmp_path / "x.q3w")
    W.write_pack(p, {"hidden": 1024.0, "rms_eps": 1e-6}, t)
    meta, back = W.read_pack(p)
    assert meta["hidden"] == 1024.0 and abs(meta["rms_eps"] - 1e-6) < 1e-12
    for k in t:
        np.testing.assert_array_equal(np.asarray(back[k]), t[k])
        assert back[k].dtype == t[k].dtype


def test_config_meta_roundtrip_and_inventory():
    cfg = W.ModelConfig()
    c2 = W.ModelConfig.from_meta(cfg.meta())
    assert c2 == cfg
    names = W.talker_tensor_shapes(cfg)
    assert len(names) == 28 * 11 + 3 and names["talker.codec_head"] == (3072, 1024)
    cp = W.cp_tensor_shapes(cfg)
    assert len(cp) == 5 * 11 + 1 + 30 and cp["cp.lm_head.14"] == (2048, 1024)


def test_vocoder_program_shapes():
    vc = W.VocConfig()
    prog, shapes = W.voc_program(vc)
    assert W.voc_total_upsample(vc) == 1920
    assert prog[0][0] == W.VOP_RVQ and prog[-1][2] == 1 and prog[-1][5] & W.VF_CLAMP
    assert sum(1 for r in prog if r[0] == W.VOP_CONVT) == 6
    # residual units: 4 blocks x 3 dilations
    assert sum(1 for r in prog if r[0] == W.VOP_CONV and r[5] & W.VF_RES_SAVE) == 12
    # the default table is the whole decoder: 8 pre-transformer layers and a ConvNeXt block per x2 upsampler
    assert sum(1 for r in prog if r[0] == W.VOP_ATTN) == 8 and sum(1 for r in prog if r[0] == W.VOP_DWCONV) == 2
    trunk, _ = W.voc_program(W.trunk_voc_config())
    assert len(trunk) == 34 and not any(r[0] in (W.VOP_ATTN, W.VOP_DWCONV, W.VOP_NORM, W.VOP_GLU) for r in trunk)


def test_oracle_f16_rounding_matches_numpy():
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-6, 1e-3, 1.0, 300.0, 7e4)])
    got = orc.round_f16(x)
    want = np.clip(x, -65504, 65504).astype(np.float16).astype(np.float32)
    np.testing.assert_array_equal(got, want)


def test_oracle_cp_loop_equals_reference_shaped_python_loop():
    """oracle/q3_oracle.c's orc_cp_predict (C loop) == oracle/frontend.cp_predict_loop (pinned to the
    reference by tests/test_golden_frontend.py) driven by the same C transformer stack."""
    from oracle import frontend as fe
    from oracle import oracle as orc
    from tests.util import synthetic_pack
    path, cfg, tensors = synthetic_pack(2, 2)
    cp = orc.CpOracle(cfg, tensors)
    rng = np.random.default_rng(5)
    hidden = rng.standard_normal(1024).astype(np.float32)
    codes, _ = cp.predict(hidden, 321)
    st = orc.StackOracle(cfg, tensors, "cp", cfg.cp_layers, cfg.cp_ffn, 16)

    def step(h, positions):
        out = st.forward(h.reshape(-1, 1024), positions[0], all_rows=True)
        return orc.round_f16(out).reshape(1, -1, 1024)   # heads see the fp16-rounded hidden (numerics contract)

    heads = [np.asarray(a) for a in cp.heads]
    toks = fe.cp_predict_loop(step, hidden, 321, cp.talker_emb, cp.emb, heads)
    assert [int(x) for x in codes] == toks


def test_oracle_batched_pipeline_equals_per_utterance_pipeline():
    """oracle/pipeline.CpuPipeline.generate_batch (weights read once per pass for the whole batch -- what makes the
    full-depth 32-utterance GPU parity test affordable) is, per utterance, bit-identical to generate(): ids AND
    margins, including an utterance the adaptive EOS boost ends early."""
    from oracle.pipeline import CpuPipeline
    from tests.util import synthetic_pack
    path, cfg, tensors = synthetic_pack(2, 2)
    cpu = CpuPipeline(cfg, tensors, n_ctx=64)
    rng = np.random.default_rng(72)
    lens, n_text = [12, 21, 17, 9], [30, 12, 3, 25]
    prefixes = [(0.05 * rng.standard_normal((n, 1024))).astype(np.float32) for n in lens]
    pad = (0.05 * rng.standard_normal(1024)).astype(np.float32)
    frames_b, margins_b = cpu.generate_batch(prefixes, n_text, pad, 10)
    ended_early = 0
    for b in range(4):
        fr, mm = cpu.generate(prefixes[b], n_text[b], pad, 10, want_margins=True)
        assert fr == frames_b[b]
        assert len(mm) == len(margins_b[b])
        for x, y in zip(mm, margins_b[b]):
            np.testing.assert_array_equal(np.asarray(x, np.float64), np.asarray(y, np.float64))
        ended_early += len(fr) < 10
    assert ended_early >= 1
