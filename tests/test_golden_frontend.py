"""oracle/frontend.py (the numpy restatement) against vectors produced by the reference's own
Python front-end (tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import struct
import types

import numpy as np
import pytest

from oracle import frontend as fe
from qwen3_tts_axera_russian_amd.weights import ModelConfig


def _talker_tables(seed, H, TD, TV, CV):
    r = np.random.default_rng(seed)
    return dict(text_embedding=(0.5 * r.standard_normal((TV, TD))).astype(np.float32),
                codec_embedding=(0.5 * r.standard_normal((CV, H))).astype(np.float32),
                codec_head=(0.5 * r.standard_normal((CV, H))).astype(np.float32),
                fc1_w=(0.3 * r.standard_normal((TD, TD))).astype(np.float32),
                fc1_b=(0.1 * r.standard_normal(TD)).astype(np.float32),
                fc2_w=(0.3 * r.standard_normal((H, TD))).astype(np.float32),
                fc2_b=(0.1 * r.standard_normal(H)).astype(np.float32))


@pytest.fixture(scope="module")
def tt(golden):
    H, TD, TV, CV = (int(x) for x in golden["talker_dims"])
    return _talker_tables(int(golden["talker_seed"]), H, TD, TV, CV)


def test_build_prefix_matches_reference(golden, tt):
    cfg = ModelConfig()
    emb = lambda ids: fe.embed_text(ids, tt["text_embedding"], tt["fc1_w"], tt["fc1_b"], tt["fc2_w"], tt["fc2_b"])
    for i in range(3):
        ids = golden[f"prefix_{i}_ids"]
        out = fe.build_prefix(ids, cfg, tt["codec_embedding"], emb)
        ref = golden[f"prefix_{i}_out"]
        assert out.shape == ref.shape == (len(ids) + 9, tt["codec_embedding"].shape[1])
        assert out.dtype == np.float32
        np.testing.assert_array_equal(out, ref)


def test_talker_sampling_greedy_matches_reference(golden, tt):
    n = int(golden["sample_n"])
    seen_eos = 0
    for ci in range(n):
        hidden = golden[f"sample_{ci}_hidden"]
        past = [int(x) for x in golden[f"sample_{ci}_past"]]
        n_text = int(golden[f"sample_{ci}_ntext"])
        past_arg = past if ci % 5 else (past or None)
        logits = hidden @ tt["codec_head"].T
        tok = fe.sample_talker(logits, past_arg, n_text, temperature=0.0)
        assert tok == int(golden[f"sample_{ci}_tok"]), f"case {ci}"
        seen_eos += tok == 2150
    assert seen_eos >= 1  # the forced / boosted EOS regime is covered
    logits = golden["sample_rep_hidden"] @ tt["codec_head"].T
    assert fe.sample_talker(logits, [int(x) for x in golden["sample_rep_past"]], 50) == int(golden["sample_rep_tok"])


def test_talker_sampling_stochastic_matches_reference(golden, tt):
    np.random.seed(1234)
    toks = []
    for ci in range(8):
        hidden = golden[f"sample_{ci}_hidden"]
        past = [int(x) for x in golden[f"sample_{ci}_past"]]
        toks.append(fe.sample_talker(hidden @ tt["codec_head"].T, past, int(golden[f"sample_{ci}_ntext"]),
                                     temperature=0.8, top_k=50, rng=np.random))
    np.testing.assert_array_equal(np.array(toks), golden["sample_stoch_toks"])


def test_cp_sample_matches_reference(golden):
    lg = golden["cps_logits"]
    assert fe.sample_cp(lg, 0.0) == int(golden["cps_greedy"])
    np.random.seed(99)
    got = [fe.sample_cp(lg.copy(), 0.1, 50, np.random) for _ in range(6)]
    np.testing.assert_array_equal(np.array(got), golden["cps_stoch"])


def _cp_tables(seed, CPV=2048, CPH=1024, CV=3072):
    r = np.random.default_rng(seed)
    emb = [(0.5 * r.standard_normal((CPV, CPH))).astype(np.float32) for _ in range(15)]
    heads = [(0.5 * r.standard_normal((CPV, CPH))).astype(np.float32) for _ in range(15)]
    talker = (0.5 * r.standard_normal((CV, CPH))).astype(np.float32)
    return emb, heads, talker


@pytest.mark.parametrize("bp", [0, 1])
def test_cp_loop_schedule_matches_reference(golden, bp):
    emb, heads, talker = _cp_tables(int(golden["cp_seed"]))
    calls, kv_len = [], [0]

    def step(h, positions):
        calls.append([list(h.shape), [int(p) for p in positions], kv_len[0]])
        kv_len[0] += h.shape[1]
        return np.tanh(0.9 * h + 0.05 * np.asarray(positions, np.float32)[None, :, None]).astype(np.float32)

    toks = fe.cp_predict_loop(step, golden[f"cploop_{bp}_hidden"], 777, talker, emb, heads, batch_prefill=bool(bp))
    np.testing.assert_array_equal(np.array(toks), golden[f"cploop_{bp}_tokens"])
    ref_calls = json.loads(str(golden[f"cploop_{bp}_calls"]))
    assert calls == [[list(c[0]), c[1], c[2]] for c in ref_calls]
    assert len(calls) == (15 if bp else 16)


def test_feedback_and_wire_bytes_match_reference_client(golden):
    tr = np.random.default_rng(int(golden["client_seed"]))
    codec = (0.5 * tr.standard_normal((3072, 1024))).astype(np.float32)
    cp_emb = [(0.5 * tr.standard_normal((2048, 1024))).astype(np.float32) for _ in range(15)]
    pad = (0.5 * tr.standard_normal(1024)).astype(np.float32)
    fb = golden["client_feedback"]
    for f in range(fb.shape[0]):
        got = fe.feedback_embedding(int(golden["client_code0"][f]), [int(x) for x in golden["client_cp"][f]],
                                    codec, cp_emb, pad)
        np.testing.assert_array_equal(got, fb[f])
    # wire protocol (SURVEY.md "Wire protocol"): what the reference client actually sent
    req = golden["client_talker_request"].tobytes()
    n = struct.unpack("<I", req[:4])[0]
    msg = json.loads(req[4:4 + n].decode())
    assert msg == {"text": "Привет", "language": "russian"}
    cp_req = golden["client_cp_request_0"].tobytes()
    assert len(cp_req) == 4100
    np.testing.assert_array_equal(np.frombuffer(cp_req[:4096], np.float32), golden["client_hidden"][0])
    assert struct.unpack("<i", cp_req[4096:])[0] == int(golden["client_code0"][0])
    voc_req = golden["client_voc_request"].tobytes()
    nt = struct.unpack("<i", voc_req[:4])[0]
    codes = np.frombuffer(voc_req[4:], np.int64).reshape(nt, 16)
    assert nt == 3
    np.testing.assert_array_equal(codes[:, 0], golden["client_code0"])
    np.testing.assert_array_equal(codes[:, 1:], golden["client_cp"])
    assert list(golden["client_wav_params"]) == [1, 2, 24000, 3 * 1920]


def _stub_chunk(padded):
    c = padded[0].astype(np.float64)
    per_tok = (c @ (np.arange(16) + 1.0)) / (2048.0 * 136.0)
    t = np.arange(64 * 1920, dtype=np.float64)
    a = np.repeat(per_tok, 1920) * 0.8 + 0.1 * np.sin(t * 0.001) + 0.05 * per_tok.sum()
    return a.astype(np.float32)


def test_vocoder_chunking_matches_reference(golden):
    vrng = np.random.default_rng(int(golden["voc_seed"]))
    for n in (int(x) for x in golden["voc_ns"]):
        codes = vrng.integers(0, 2048, size=(n, 16)).astype(np.int64)
        out = fe.voc_synthesize(codes, _stub_chunk, 64)
        assert len(out) == int(golden[f"voc_{n}_len"]), n
        assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest() == str(golden[f"voc_{n}_sha"]), n
    # the reference's length quirk: n=150 -> 156 frames, n=97 -> 98
    assert int(golden["voc_150_len"]) == 156 * 1920
    assert int(golden["voc_97_len"]) == 98 * 1920
    assert int(golden["voc_96_len"]) == 96 * 1920


def test_vocoder_chunking_with_a_model_that_returns_short_chunks(golden):
    """`audio[:n * 1920]` of a chunk shorter than 64 x 1920 samples is as long as what is there (numpy slicing):
    the reference's own synthesize run on stubs returning 122 325 (the decoder family's 64-frame length),
    60 x 1920 + 7 and 15 x 1920 + 5 samples (vocoder_server.py:81,98-99,104-117)."""
    for cs in (int(x) for x in golden["vocshort_lens"]):
        srng = np.random.default_rng(53)
        for n in (int(x) for x in golden["vocshort_ns"]):
            codes = srng.integers(0, 2048, size=(n, 16)).astype(np.int64)
            out = fe.voc_synthesize(codes, lambda p: _stub_chunk(p)[:cs], 64)
            assert len(out) == int(golden[f"vocshort_{cs}_{n}_len"]), (cs, n)
            assert hashlib.sha256(np.ascontiguousarray(out, dtype=np.float32).tobytes()).hexdigest() == \
                str(golden[f"vocshort_{cs}_{n}_sha"]), (cs, n)
    assert int(golden["vocshort_122325_64_len"]) == 122325 and int(golden["vocshort_122325_63_len"]) == 63 * 1920
    assert int(golden["vocshort_28805_65_len"]) == 2 * 28805      # shorter than the overlap: plain concatenation


def test_int16_rule_matches_reference_server(golden):
    codes = golden["vocsrv_codes"]
    audio = fe.voc_synthesize(codes, lambda p: _stub_chunk(p) * 8.0 - 4.5, 64)
    got = fe.to_int16(audio)
    ref = golden["vocsrv_int16"]
    np.testing.assert_array_equal(got, ref)
    assert (got.max() == 32767 or got.min() == -32768) and got.min() < 0  # clipping + negative truncation exercised
