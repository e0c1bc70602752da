"""The product's host-side front-end (frontend.py, protocol.py) against vectors produced by the
reference's own Python (tests/golden).  CPU only."""
import json
import struct

import numpy as np
import pytest

from qwen3_tts_axera_russian_amd import frontend as fe
from qwen3_tts_axera_russian_amd import protocol as P
from qwen3_tts_axera_russian_amd.weights import ModelConfig
from tests.test_golden_frontend import _talker_tables


@pytest.fixture(scope="module")
def tt(golden):
    H, TD, TV, CV = (int(x) for x in golden["talker_dims"])
    return _talker_tables(int(golden["talker_seed"]), H, TD, TV, CV)


def test_prefix_matches_reference(golden, tt):
    front = fe.TextFrontEnd(ModelConfig(), tt["text_embedding"], tt["fc1_w"], tt["fc1_b"], tt["fc2_w"], tt["fc2_b"],
                            tt["codec_embedding"])
    for i in range(3):
        ids = [int(x) for x in golden[f"prefix_{i}_ids"]]
        np.testing.assert_array_equal(front.build_prefix(ids), golden[f"prefix_{i}_out"])
    assert front.build_prefix([]).shape == (9, tt["codec_embedding"].shape[1])   # empty text: 9 rows


def test_prefix_matches_reference_at_the_real_widths(golden):
    """The wprefix_* goldens (reference `_build_prefix` at text dim 2048 -> hidden 1024 over a 640-row table read at
    id % 640): the host front-end reproduces them bit for bit with the special ids' residues."""
    r = np.random.default_rng(int(golden["wprefix_seed"]))
    emb = (0.05 * r.standard_normal((640, 2048))).astype(np.float32)
    fc1_w = (0.02 * r.standard_normal((2048, 2048))).astype(np.float32)
    fc1_b = (0.02 * r.standard_normal(2048)).astype(np.float32)
    fc2_w = (0.02 * r.standard_normal((1024, 2048))).astype(np.float32)
    fc2_b = (0.02 * r.standard_normal(1024)).astype(np.float32)
    codec = (0.05 * r.standard_normal((3072, 1024))).astype(np.float32)
    cfg = ModelConfig(text_vocab=640, tts_pad=151671 % 640, tts_bos=151672 % 640, tts_eos=151673 % 640,
                      im_start=151644 % 640, assistant=77091 % 640, newline=198 % 640)
    front = fe.TextFrontEnd(cfg, emb, fc1_w, fc1_b, fc2_w, fc2_b, codec)
    for i in range(2):
        ids = [int(x) for x in golden[f"wprefix_{i}_ids"]]
        np.testing.assert_array_equal(front.build_prefix(ids), golden[f"wprefix_{i}_out"])


def test_sampler_matches_reference(golden, tt):
    s = fe.TalkerSampler(temperature=0.0)
    for ci in range(int(golden["sample_n"])):
        past = [int(x) for x in golden[f"sample_{ci}_past"]]
        past_arg = past if ci % 5 else (past or None)
        logits = golden[f"sample_{ci}_hidden"] @ tt["codec_head"].T
        assert s.sample(logits, past_arg, int(golden[f"sample_{ci}_ntext"])) == int(golden[f"sample_{ci}_tok"]), ci
    np.random.seed(1234)
    st = fe.TalkerSampler(temperature=0.8, top_k=50, rng=np.random)
    toks = [st.sample(golden[f"sample_{ci}_hidden"] @ tt["codec_head"].T, [int(x) for x in golden[f"sample_{ci}_past"]],
                      int(golden[f"sample_{ci}_ntext"])) for ci in range(8)]
    np.testing.assert_array_equal(np.array(toks), golden["sample_stoch_toks"])


def test_feedback_matches_reference(golden):
    tr = np.random.default_rng(int(golden["client_seed"]))
    codec = (0.5 * tr.standard_normal((3072, 1024))).astype(np.float32)
    cp_emb = [(0.5 * tr.standard_normal((2048, 1024))).astype(np.float32) for _ in range(15)]
    pad = (0.5 * tr.standard_normal(1024)).astype(np.float32)
    for f in range(3):
        got = fe.feedback_embedding(int(golden["client_code0"][f]), [int(x) for x in golden["client_cp"][f]], codec, cp_emb, pad)
        np.testing.assert_array_equal(got, golden["client_feedback"][f])


def test_wire_messages_are_byte_identical_to_the_reference_client(golden):
    assert P.pack_talker_request("Привет", "russian") == golden["client_talker_request"].tobytes()
    assert P.pack_cp_request(golden["client_hidden"][0], int(golden["client_code0"][0])) == golden["client_cp_request_0"].tobytes()
    codes = np.concatenate([golden["client_code0"][:, None], golden["client_cp"]], axis=1)
    assert P.pack_voc_request(codes) == golden["client_voc_request"].tobytes()
    assert P.pack_cp_reply(range(15)) == struct.pack("<15i", *range(15))
    assert len(P.pack_talker_frame(7, np.zeros(1024, np.float32))) == 4100
    assert P.pack_sentinel(P.SENTINEL_DONE) == struct.pack("<i", -1)
    msg = json.loads(P.pack_talker_request("x", "english", [1, 2])[4:].decode())
    assert msg["token_ids"] == [1, 2]


def test_text_front_end_from_the_reference_embeddings_directory(tmp_path):
    """The talker server's tables from the reference's own embeddings/ directory of .npy files
    (scripts/extract_embeddings.py:47-66) give the same prefix rows as the container's text.* tensors."""
    import numpy as np
    from qwen3_tts_axera_russian_amd import weights as W
    from qwen3_tts_axera_russian_amd.frontend import load_text_front_end
    cfg = W.tiny_config(1, 1, text_vocab=300)
    cfg.text_dim = 48
    pack = str(tmp_path / "t.q3w")
    W.write_synthetic(pack, cfg, seed=5, parts=("talker", "text"))
    _, t = W.read_pack(pack)
    emb = tmp_path / "embeddings"
    emb.mkdir()
    np.save(emb / "text_embedding.npy", np.asarray(t["text.embedding"], np.float32))
    for k in ("fc1", "fc2"):
        np.save(emb / f"text_projection_linear_{k}_weight.npy", np.asarray(t[f"text.{k}.weight"], np.float32))
        np.save(emb / f"text_projection_linear_{k}_bias.npy", np.asarray(t[f"text.{k}.bias"], np.float32))
    np.save(emb / "codec_embedding.npy", np.asarray(t["talker.codec_embedding"], np.float32))
    _, a = load_text_front_end(pack, None)
    _, b = load_text_front_end(None, str(emb), cfg=cfg)
    ids = [5, 17, 200, 33, 41]
    np.testing.assert_array_equal(a.build_prefix(ids), b.build_prefix(ids))
    np.testing.assert_array_equal(a.tts_pad_embed, b.tts_pad_embed)


def test_client_tables_from_the_reference_directories(tmp_path):
    """The client's feedback tables from the reference's embeddings/ + code-predictor directories
    (tts_client.py:39-76) equal the container's."""
    import numpy as np
    from qwen3_tts_axera_russian_amd import weights as W
    from qwen3_tts_axera_russian_amd.tts_client import Qwen3TTSClient
    # the reference client hard-codes tts_pad = 151671: its tables at the real ids, with a sparse text table
    pack = str(tmp_path / "t.q3w")
    full = W.ModelConfig(talker_layers=1, cp_layers=1, text_dim=32)
    t = W.make_synthetic(full, seed=9, parts=("cp",))
    rng = np.random.default_rng(1)
    text = np.zeros((full.text_vocab, 32), np.float32)
    text[full.tts_pad:full.tts_eos + 1] = rng.standard_normal((3, 32)).astype(np.float32)
    for n in (full.im_start, full.assistant, full.newline):
        text[n] = rng.standard_normal(32).astype(np.float32)
    t["text.embedding"] = text
    t["text.fc1.weight"] = (0.2 * rng.standard_normal((32, 32))).astype(np.float32)
    t["text.fc1.bias"] = (0.1 * rng.standard_normal(32)).astype(np.float32)
    t["text.fc2.weight"] = (0.2 * rng.standard_normal((1024, 32))).astype(np.float32)
    t["text.fc2.bias"] = (0.1 * rng.standard_normal(1024)).astype(np.float32)
    t["talker.codec_embedding"] = (0.1 * rng.standard_normal((3072, 1024))).astype(np.float32)
    W.write_pack(pack, full.meta(), t)
    emb, cp = tmp_path / "embeddings", tmp_path / "code_predictor"
    emb.mkdir()
    cp.mkdir()
    np.save(emb / "text_embedding.npy", text)
    for k in ("fc1", "fc2"):
        np.save(emb / f"text_projection_linear_{k}_weight.npy", t[f"text.{k}.weight"])
        np.save(emb / f"text_projection_linear_{k}_bias.npy", t[f"text.{k}.bias"])
    np.save(emb / "codec_embedding.npy", t["talker.codec_embedding"])
    np.savez(cp / "code_predictor_weights.npz", **{f"codec_emb_{g}": np.asarray(t[f"cp.codec_emb.{g}"], np.float32) for g in range(15)})
    a = Qwen3TTSClient(weights=pack)
    b = Qwen3TTSClient(embeddings_dir=str(emb), cp_dir=str(cp))
    np.testing.assert_array_equal(a.tts_pad_embed, b.tts_pad_embed)
    np.testing.assert_array_equal(a.codec_embedding, b.codec_embedding)
    assert len(b.cp_codec_embeddings) == 15
    for x, y in zip(a.cp_codec_embeddings, b.cp_codec_embeddings):
        np.testing.assert_array_equal(x, y)
