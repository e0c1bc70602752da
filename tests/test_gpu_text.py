"""Text front-end on the GPU (include/qwen3tts_text.h: fp16 text table in HBM, fc1 -> SiLU -> fc2 as MFMA GEMMs,
prefix assembly as a kernel) against the host front-end frontend.TextFrontEnd, whose arithmetic is pinned bit for bit
to the reference's own _embed_text / _build_prefix outputs (tests/test_host_frontend.py::test_prefix_matches_reference,
golden vectors from llamacpp_talker_server.py:115-161).

Tolerance: the device rounds the table, both weight matrices and the fc1 output (after SiLU) to fp16 and accumulates
in f32; the host is f32 throughout.  Stated bound: max |dev - host| <= 4e-3 x max |host| per call (measured ~1e-3),
and the codec-stream rows added on top are exact f32 adds on both sides."""
import os

import numpy as np
import pytest

from qwen3_tts_axera_russian_amd import frontend as fe
from qwen3_tts_axera_russian_amd import weights as W
from tests.util import CACHE

pytestmark = pytest.mark.gpu
TOL = 4e-3


@pytest.fixture(scope="module")
def pack():
    os.makedirs(CACHE, exist_ok=True)
    cfg = W.tiny_config(2, 2, text_vocab=640)           # real widths (2048 -> 2048 -> 1024), small vocabulary
    path = os.path.join(CACHE, "text_t2c2_v640.q3w")
    if not os.path.exists(path):
        W.write_synthetic(path, cfg, seed=77, parts=("talker", "text"))
    return path, cfg


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def test_prefix_and_embed_text_match_the_host_front_end(gpu_lib, pack):
    path, cfg = pack
    _, host = fe.load_text_front_end(path, cfg=cfg)
    dev = fe.DeviceTextFrontEnd(cfg, path, max_tokens=700)
    rng = np.random.default_rng(3)
    for n in (0, 1, 17, 40, 63, 130, 600):               # incl. empty text and a pass long enough for the tiled GEMM
        ids = rng.integers(0, cfg.text_vocab - 8, size=n).tolist()
        want, got = host.build_prefix(ids), dev.build_prefix(ids)
        assert got.shape == want.shape == (n + 9, 1024)
        e = _rel(got, want)
        assert e <= TOL, (n, e)
    ids = rng.integers(0, cfg.text_vocab, size=33)
    assert _rel(dev.embed_text(ids), host.embed_text(ids)) <= TOL
    for a, b in ((dev.tts_pad_embed, host.tts_pad_embed), (dev.tts_bos_embed, host.tts_bos_embed), (dev.tts_eos_embed, host.tts_eos_embed)):
        assert _rel(a, b) <= TOL
    pad = np.empty(1024, np.float32)
    from qwen3_tts_axera_russian_amd import hiplib
    assert gpu_lib.tfe_tts_pad_embed(dev.h, hiplib.fptr(pad)) == 0
    np.testing.assert_array_equal(pad, dev.tts_pad_embed)
    # layout facts that do not depend on precision: role rows carry no codec stream; the codec rows differ from the
    # text-only projection by exactly one codec-table row (f32 add on both sides)
    p = dev.build_prefix([5, 6, 7])
    codec = np.asarray(host.codec, np.float32)
    np.testing.assert_array_equal(p[0:3], dev.embed_text([cfg.im_start, cfg.assistant, cfg.newline]))
    np.testing.assert_array_equal(p[6], dev.tts_bos_embed + codec[cfg.codec_pad])
    np.testing.assert_array_equal(p[11], dev.tts_pad_embed + codec[cfg.codec_bos])
    dev.destroy()


def test_bad_ids_and_missing_tensors_fail_loudly(gpu_lib, pack, tmp_path):
    path, cfg = pack
    dev = fe.DeviceTextFrontEnd(cfg, path, max_tokens=64)
    with pytest.raises(RuntimeError):
        dev.build_prefix([cfg.text_vocab + 5])           # outside the table
    with pytest.raises(RuntimeError):
        dev.build_prefix(list(range(200)))               # longer than max_tokens
    dev.destroy()
    from tests.util import synthetic_pack
    no_text, _, _ = synthetic_pack(2, 2)                 # talker + cp only
    assert not gpu_lib.tfe_load(no_text.encode(), None, 64)


def test_loads_the_reference_embeddings_directory(gpu_lib, pack, tmp_path):
    """tfe_load on the reference's embeddings/ directory (the .npy files scripts/extract_embeddings.py:47-66 writes,
    parsed natively) gives the container's results."""
    path, cfg = pack
    _, t = W.read_pack(path)
    d = tmp_path / "embeddings"
    d.mkdir()
    np.save(d / "text_embedding.npy", np.asarray(t["text.embedding"], np.float32))
    for fc in ("fc1", "fc2"):
        np.save(d / f"text_projection_linear_{fc}_weight.npy", np.asarray(t[f"text.{fc}.weight"], np.float32))
        np.save(d / f"text_projection_linear_{fc}_bias.npy", np.asarray(t[f"text.{fc}.bias"], np.float32))
    np.save(d / "codec_embedding.npy", np.asarray(t["talker.codec_embedding"], np.float32))
    a = fe.DeviceTextFrontEnd(cfg, path)
    b = fe.DeviceTextFrontEnd(cfg, None, embeddings_dir=str(d))
    ids = [3, 100, 250, 9, 77]
    np.testing.assert_array_equal(a.build_prefix(ids), b.build_prefix(ids))
    a.destroy()
    b.destroy()


def _wide_tables(seed):
    """tests/golden/make_golden.py::wide_tables, regenerated from its seed"""
    r = np.random.default_rng(seed)
    return dict(text_embedding=(0.05 * r.standard_normal((640, 2048))).astype(np.float32),
                fc1_w=(0.02 * r.standard_normal((2048, 2048))).astype(np.float32),
                fc1_b=(0.02 * r.standard_normal(2048)).astype(np.float32),
                fc2_w=(0.02 * r.standard_normal((1024, 2048))).astype(np.float32),
                fc2_b=(0.02 * r.standard_normal(1024)).astype(np.float32),
                codec_embedding=(0.05 * r.standard_normal((3072, 1024))).astype(np.float32))


def test_device_prefix_against_the_references_own_output(gpu_lib, tmp_path):
    """tfe_build_prefix compared DIRECTLY with what the reference's `_build_prefix` returned
    (tests/golden/frontend_golden.npz: wprefix_*, llamacpp_talker_server.py:115-161 run by tests/golden/make_golden.py at
    the device path's widths: text dim 2048 -> 2048 -> hidden 1024).  The golden's text table is a 640-row table read at
    id % 640 (the 151 936-row one would be 1.2 GB); the device gets the same 640 rows and the special ids' residues.
    Tolerance as above: fp16 table / weights / SiLU output with f32 accumulation against f32 numpy, <= 4e-3 of the
    largest element (measured ~1e-3); rows whose text part is shared are identical where the reference's are."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_golden.npz"))
    t = _wide_tables(int(g["wprefix_seed"]))
    cfg = W.ModelConfig(talker_layers=2, cp_layers=2, text_vocab=640, tts_pad=151671 % 640, tts_bos=151672 % 640,
                        tts_eos=151673 % 640, im_start=151644 % 640, assistant=77091 % 640, newline=198 % 640)
    path = str(tmp_path / "wide_text.q3w")
    W.write_pack(path, cfg.meta(), {"text.embedding": t["text_embedding"], "text.fc1.weight": t["fc1_w"], "text.fc1.bias": t["fc1_b"],
                                    "text.fc2.weight": t["fc2_w"], "text.fc2.bias": t["fc2_b"],
                                    "talker.codec_embedding": t["codec_embedding"]})
    dev = fe.DeviceTextFrontEnd(cfg, path, max_tokens=64)
    host = fe.TextFrontEnd(cfg, t["text_embedding"], t["fc1_w"], t["fc1_b"], t["fc2_w"], t["fc2_b"], t["codec_embedding"])
    for i in range(2):
        ids = [int(x) for x in g[f"wprefix_{i}_ids"]]
        want = g[f"wprefix_{i}_out"]
        np.testing.assert_array_equal(host.build_prefix(ids), want)        # the host front-end: bit for bit
        got = dev.build_prefix(ids)
        assert got.shape == want.shape
        e = _rel(got, want)
        print(f"device prefix vs the reference's output, {len(ids)} text tokens: max rel err {e:.2e}")
        assert e <= TOL, (i, e)
        # structure the tolerance cannot hide: rows 3..5 differ from each other by codec rows only, exactly
        codec = t["codec_embedding"]
        np.testing.assert_allclose(got[4] - got[3], codec[cfg.codec_think_bos] - codec[cfg.codec_nothink], rtol=0, atol=2e-7)
    dev.destroy()
