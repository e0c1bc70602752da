"""weights.speech_tokenizer_to_voc: a speech_tokenizer/ directory (config.json + safetensors, the thing
scripts/export_vocoder_traced.py:74-79 points from_pretrained at) -> the vocoder's op table, with every size read from
tensor shapes.  The real checkpoint is not in the reference, so the directory is synthetic: the table's own tensors
written out under the VOC_NAMES layout (separate q/k/v, gate/up, 2-D Linear weights, layer scales / ConvNeXt gamma as
separate tensors, EMA-form codebook), at sizes that differ from every default so a hard-coded width would show."""
import numpy as np
import pytest

from qwen3_tts_axera_russian_amd import weights as W


def _odd_config():
    return W.VocConfig(n_q=5, codebook_size=96, codebook_dim=24, rvq_out=40, latent=48, pre_kernel=5, upsample_ratios=(2, 3),
                       decoder_dim=64, rates=(4, 2), dilations=(1, 2), kernel=5, pre_transformer_layers=3, tf_hidden=32,
                       tf_heads=2, tf_head_dim=8, tf_ffn=56, tf_window=11, tf_rope_theta=5000, tf_eps_e9=20000,
                       convnext=True, convnext_kernel=3)


@pytest.mark.parametrize("layer_scales", [True, False])
def test_round_trip_through_a_speech_tokenizer_directory(tmp_path, layer_scales):
    vc = _odd_config()
    t = W.make_synthetic_voc(vc, seed=11)
    W.export_speech_tokenizer_layout(t, vc, str(tmp_path / "speech_tokenizer"), layer_scales=layer_scales)
    vc2, t2, report = W.speech_tokenizer_to_voc(str(tmp_path / "speech_tokenizer"))
    for f in ("n_q", "codebook_size", "codebook_dim", "rvq_out", "latent", "pre_kernel", "upsample_ratios", "decoder_dim",
              "rates", "dilations", "kernel", "pre_transformer_layers", "tf_hidden", "tf_heads", "tf_head_dim", "tf_ffn",
              "tf_window", "tf_rope_theta", "tf_eps_e9", "convnext", "convnext_kernel"):
        assert getattr(vc2, f) == getattr(vc, f), f
    np.testing.assert_array_equal(t2["voc.program"], t["voc.program"])
    assert set(t2) == set(t)
    for k in t:
        if k != "voc.program":
            np.testing.assert_allclose(t2[k], t[k], rtol=2e-6, atol=1e-7, err_msg=k)   # scales divided out and folded back
    assert "ops:" in report[0] and W.voc_total_upsample(vc2) == 2 * 3 * 4 * 2


def test_trunk_only_directory_and_missing_tensor_message(tmp_path):
    vc = W.VocConfig(codebook_dim=16, rvq_out=24, latent=32, decoder_dim=64, rates=(2, 2), pre_transformer_layers=0, convnext=False)
    t = W.make_synthetic_voc(vc, seed=2)
    d = str(tmp_path / "st")
    W.export_speech_tokenizer_layout(t, vc, d)
    vc2, t2, _ = W.speech_tokenizer_to_voc(d)
    assert vc2.pre_transformer_layers == 0 and not vc2.convnext and vc2.rates == (2, 2)
    np.testing.assert_array_equal(t2["voc.program"], t["voc.program"])
    out = str(tmp_path / "voc.q3w")
    W.convert_speech_tokenizer(d, out)
    meta, back = W.read_pack(out)
    assert meta["voc_chunk"] == 64.0 and "voc.program" in back
    with pytest.raises(KeyError, match="VOC_NAMES"):
        W.speech_tokenizer_to_voc(d, names={"pre_conv": "decoder.renamed_pre_conv"})
