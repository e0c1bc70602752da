"""Code-predictor boundary (cp_* ABI) against the CPU oracle: greedy codec ids must be identical
(bit-exact integers); a mismatch is tolerated only where the oracle's own top-1/top-2 logit gap
is below 1e-4 (a float near-tie), and the per-position hidden must agree within 5e-3."""
import numpy as np
import pytest

from oracle import oracle as orc
from qwen3_tts_axera_russian_amd.llama_cpp_bindings import CodePredictor
from tests.util import rel_err, synthetic_pack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    path, cfg, tensors = synthetic_pack(2, 2)
    return path, cfg, tensors, orc.CpOracle(cfg, tensors)


def _check_codes(got, ref_codes, margins):
    got = np.asarray(got)
    if np.array_equal(got, ref_codes):
        return
    g = int(np.nonzero(got != ref_codes)[0][0])
    assert margins[g] < 1e-4, f"codes diverge at group {g} with oracle margin {margins[g]}: {got} vs {ref_codes}"


def test_cp_predict_greedy_matches_oracle(gpu_lib, setup):
    path, cfg, tensors, ref = setup
    cp = CodePredictor(path, max_batch=1)
    rng = np.random.default_rng(10)
    min_margin = 1e9
    for case in range(6):
        hidden = rng.standard_normal(1024).astype(np.float32)
        code0 = int(rng.integers(0, 2048))
        got = cp.predict(hidden, code0)
        codes, margins = ref.predict(hidden, code0)
        _check_codes(got, codes, margins)
        min_margin = min(min_margin, float(margins.min()))
        assert all(0 <= c < 2048 for c in got)
    print("cp min oracle margin:", min_margin)
    # the reference's warm-up input (code_predictor_server.cpp:510-518): hidden = 0.1, code_0 = 100
    w = np.full(1024, 0.1, np.float32)
    codes, margins = ref.predict(w, 100)
    _check_codes(cp.predict(w, 100), codes, margins)
    # out-of-range code_0 embeds as zeros (code_predictor_server.cpp:269-272)
    codes, margins = ref.predict(w, 5000)
    _check_codes(cp.predict(w, 5000), codes, margins)
    cp.destroy()


def test_cp_step_hidden_matches_oracle(gpu_lib, setup):
    path, cfg, tensors, ref = setup
    cp = CodePredictor(path, max_batch=1)
    rng = np.random.default_rng(11)
    hidden = rng.standard_normal(1024).astype(np.float32)
    code0 = 321
    codes, margins, hid = ref.predict(hidden, code0, want_hidden=True)
    cp.step(hidden, 0)
    emb = np.asarray(tensors["talker.codec_embedding"][code0], np.float32)
    errs = []
    for g in range(15):
        h = cp.step(emb, g + 1)
        errs.append(rel_err(h, hid[g]))
        lg = cp.lm_head(g, h)
        assert int(np.argmax(lg)) == codes[g] or margins[g] < 1e-4
        emb = np.asarray(tensors[f"cp.codec_emb.{g}"][codes[g]], np.float32) if g < 14 else None
    print("cp step rel errs:", ["%.1e" % e for e in errs])
    assert max(errs) < 5e-3
    cp.destroy()


def test_cp_batch_equals_single(gpu_lib, setup):
    path, cfg, tensors, ref = setup
    cpb = CodePredictor(path, max_batch=8)
    rng = np.random.default_rng(12)
    for R in (1, 5, 8):
        hidden = rng.standard_normal((R, 1024)).astype(np.float32)
        code0 = rng.integers(0, 2048, size=R).astype(np.int32)
        got = cpb.predict_batch(hidden, code0)
        assert got.shape == (R, 15)
        for r in range(R):
            codes, margins = ref.predict(hidden[r], int(code0[r]))
            _check_codes(got[r], codes, margins)
    with pytest.raises(RuntimeError):
        cpb.predict_batch(np.zeros((9, 1024), np.float32), np.zeros(9, np.int32))
    cpb.destroy()


def test_cp_stochastic_sampling_distribution(gpu_lib, setup):
    """temperature > 0: group-0 tokens drawn on the device follow the reference sampler's distribution
    (code_predictor_server.py:87-92: top-k 50, softmax((l-max)/T)); only the random stream differs."""
    path, cfg, tensors, ref = setup
    cp = CodePredictor(path, max_batch=32)
    rng = np.random.default_rng(21)
    hidden = rng.standard_normal(1024).astype(np.float32)
    code0 = 77
    # logits of group 0 from the oracle, then the reference's distribution
    codes, margins, hid = ref.predict(hidden, code0, want_hidden=True)
    from oracle import oracle as orc
    logits = orc.head_logits(ref.heads[0], hid[0])
    T, K = 0.5, 50
    top = np.argsort(logits)[-K:]
    p = np.exp((logits[top] - logits[top].max()) / T)
    p /= p.sum()
    want = dict(zip(top.tolist(), p.tolist()))
    counts, N = {}, 0
    H = np.tile(hidden, (32, 1))
    C0 = np.full(32, code0, np.int32)
    for seed in range(1, 65):
        got = cp.predict_batch(H, C0, temperature=T, top_k=K, seed=seed)
        for tok in got[:, 0]:
            counts[int(tok)] = counts.get(int(tok), 0) + 1
            N += 1
    assert set(counts) <= set(want), "a token outside the top-k was drawn"
    tv = 0.5 * sum(abs(counts.get(t, 0) / N - want[t]) for t in want)
    print("total variation distance over", N, "draws:", tv)
    assert tv < 0.08
    # reproducible per seed, different across seeds, and T -> 0 is the arg-max
    a = cp.predict_batch(H[:4], C0[:4], temperature=T, top_k=K, seed=5)
    b = cp.predict_batch(H[:4], C0[:4], temperature=T, top_k=K, seed=5)
    np.testing.assert_array_equal(a, b)
    g = cp.predict_batch(H[:1], C0[:1], temperature=0.0)
    assert list(g[0]) == [int(x) for x in codes] or margins.min() < 1e-4
    cp.destroy()


def test_cp_load_from_the_reference_directories(gpu_lib, setup, tmp_path):
    """cp_load on the reference's own files -- --model_dir/code_predictor_weights.npz (written by np.savez as
    scripts/export_code_predictor_weights.py:76 does, f32) + --embeddings_dir/codec_embedding.npy -- parsed natively
    (csrc/q3_formats.cpp) gives the codes of the container holding the same values."""
    path, cfg, tensors, ref = setup
    model_dir, emb_dir = tmp_path / "code_predictor", tmp_path / "embeddings"
    model_dir.mkdir()
    emb_dir.mkdir()
    w = {}
    for i in range(cfg.cp_layers):
        for p in ("input_ln", "q_proj", "k_proj", "v_proj", "o_proj", "q_norm", "k_norm", "post_ln", "gate_proj",
                  "up_proj", "down_proj"):
            w[f"layer_{i}_{p}"] = np.asarray(tensors[f"cp.layers.{i}.{p}"], dtype=np.float32)
    w["final_norm"] = np.asarray(tensors["cp.norm"], dtype=np.float32)
    for g in range(cfg.cp_groups):
        w[f"codec_emb_{g}"] = np.asarray(tensors[f"cp.codec_emb.{g}"], dtype=np.float32)
        w[f"lm_head_{g}"] = np.asarray(tensors[f"cp.lm_head.{g}"], dtype=np.float32)
    np.savez(model_dir / "code_predictor_weights.npz", **w)
    np.save(emb_dir / "codec_embedding.npy", np.asarray(tensors["talker.codec_embedding"], dtype=np.float32))
    a = CodePredictor(path, max_batch=1)
    b = CodePredictor(str(model_dir), str(emb_dir), max_batch=1)
    rng = np.random.default_rng(41)
    for _ in range(3):
        hidden = rng.standard_normal(1024).astype(np.float32)
        code0 = int(rng.integers(0, 2048))
        np.testing.assert_array_equal(np.asarray(b.predict(hidden, code0)), np.asarray(a.predict(hidden, code0)))
    a.destroy()
    b.destroy()
