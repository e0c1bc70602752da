"""Error behaviour of the C ABI (return codes, no exceptions, nothing written on refusal) and the
edge cases the reference handles at its call sites: bad sizes and positions at `wrapper_decode_embd`
(llama_wrapper.c:125-163 returns -1 when llama_decode fails), an out-of-range code_0 embedding as zeros
(code_predictor_server.cpp:374-380), request bounds of the vocoder (vocoder_server.py:143-166)."""
import ctypes

import numpy as np
import pytest

from oracle import oracle as orc
from qwen3_tts_axera_russian_amd import hiplib
from qwen3_tts_axera_russian_amd import weights as W
from qwen3_tts_axera_russian_amd.llama_cpp_bindings import CodePredictor
from tests.util import CACHE, synthetic_pack

pytestmark = pytest.mark.gpu
fp = hiplib.fptr


def test_talker_refuses_bad_arguments(gpu_lib):
    lib = gpu_lib
    path, cfg, _ = synthetic_pack(2, 2)
    assert not lib.wrapper_load_model(b"/nonexistent/model.q3w", 0)
    model = lib.wrapper_load_model(path.encode(), 99)
    assert model and lib.wrapper_model_n_embd(model) == 1024
    assert not lib.wrapper_create_context(None, 64, 64, 4, 1)
    ctx = lib.wrapper_create_context(model, 64, 64, 4, 1)
    assert ctx
    x = np.zeros((70, 1024), np.float32)
    out = np.full(1024, 7.0, np.float32)
    bad = [(70, 1024, 0),      # more rows than n_ctx / n_batch
           (8, 1024, 60),      # runs past n_ctx
           (8, 1000, 0),       # wrong n_embd
           (0, 1024, 0),       # empty
           (1, 1024, -1)]      # negative position
    for n, ne, pos in bad:
        assert lib.wrapper_decode_embd(ctx, fp(x), n, ne, pos, fp(out)) == -1, (n, ne, pos)
    assert lib.wrapper_decode_embd(None, fp(x), 1, 1024, 0, fp(out)) == -1
    assert lib.wrapper_decode_embd(ctx, None, 1, 1024, 0, fp(out)) == -1
    assert (out == 7.0).all()                      # nothing was written on refusal
    assert lib.wrapper_state_load_file(ctx, b"/nonexistent/state.bin") == -1
    assert lib.wrapper_decode_embd(ctx, fp(x), 8, 1024, 56, fp(out)) == 0      # arbitrary pos_start up to the edge
    assert np.isfinite(out).all()
    lib.wrapper_free_context(ctx)
    lib.wrapper_free_model(model)


def test_cp_out_of_range_code0_embeds_as_zeros(gpu_lib):
    path, cfg, tensors = synthetic_pack(2, 2)
    ref = orc.CpOracle(cfg, tensors)
    cp = CodePredictor(path, max_batch=2)
    rng = np.random.default_rng(31)
    hidden = rng.standard_normal(1024).astype(np.float32)
    for code0 in (-1, 3072, 100000):
        got = np.asarray(cp.predict(hidden, code0))
        want, margins = ref.predict(hidden, code0)
        if not np.array_equal(got, want):
            g = int(np.nonzero(got != want)[0][0])
            assert margins[g] < 1e-4, (code0, g, margins[g])
    a = np.asarray(cp.predict(hidden, -1))
    b = np.asarray(cp.predict(hidden, 999999))
    np.testing.assert_array_equal(a, b)            # both embed as zeros
    out = np.zeros(15, np.int32)
    assert gpu_lib.cp_predict(None, fp(hidden), 5, 0.0, 50, 0, hiplib.iptr(out)) == -1
    assert not gpu_lib.cp_load(b"/nonexistent_dir", None, 1)
    cp.destroy()


def test_vocoder_request_bounds(gpu_lib):
    import os
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "voc_tiny_s7b.q3w")
    if not os.path.exists(path):
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(W.tiny_voc_config(), seed=7))
    lib = gpu_lib
    assert not lib.voc_load(b"/nonexistent.q3w", 64, 1)
    talker_only, _, _ = synthetic_pack(2, 2)
    assert not lib.voc_load(talker_only.encode(), 64, 1)        # a container without a vocoder program
    h = lib.voc_load(path.encode(), 64, 2)
    assert h
    codes = np.zeros((3, 64, 16), np.int64)
    out = np.full((3, 64 * 1920), 5.0, np.float32)     # (>= 3 rows of voc_chunk_samples)
    assert lib.voc_decode(h, codes.ctypes.data_as(hiplib.i64p), 3, fp(out)) == -1     # beyond max_batch
    assert lib.voc_decode(h, codes.ctypes.data_as(hiplib.i64p), 0, fp(out)) == -1
    assert (out == 5.0).all()
    ns = np.zeros(1, np.int32)
    pcm = np.zeros(1000, np.int16)
    assert lib.voc_synthesize(h, codes.ctypes.data_as(hiplib.i64p), 0, pcm.ctypes.data_as(hiplib.i16p), hiplib.iptr(ns)) == -1
    assert lib.voc_synthesize_max_samples(h, 0) == 0
    # the batched walk: null arguments, an empty utterance, no utterances, a table that trims -> chunk_samples < 64 * 1920
    assert 0 < lib.voc_chunk_samples(h) <= 64 * 1920 and lib.voc_chunk_samples(None) == 0
    nn = np.array([3, 0], np.int32)
    off = np.zeros(3, np.int64)
    big = np.full(200000, 7, np.int16)
    c2 = np.zeros((3, 16), np.int64)
    i64 = lambda a: a.ctypes.data_as(hiplib.i64p)
    assert lib.voc_synthesize_batch(h, i64(c2), hiplib.iptr(nn), 2, big.ctypes.data_as(hiplib.i16p), len(big), i64(off)) == -1
    assert lib.voc_synthesize_batch(h, i64(c2), hiplib.iptr(nn), 0, big.ctypes.data_as(hiplib.i16p), len(big), i64(off)) == -1
    assert lib.voc_synthesize_batch(None, i64(c2), hiplib.iptr(nn), 1, big.ctypes.data_as(hiplib.i16p), len(big), i64(off)) == -1
    assert lib.voc_synthesize_batch(h, None, hiplib.iptr(nn), 1, big.ctypes.data_as(hiplib.i16p), len(big), i64(off)) == -1
    assert (big == 7).all()
    assert lib.voc_synthesize_batch_max_samples(h, hiplib.iptr(nn), 0) == 0
    assert lib.voc_synthesize_batch(h, i64(c2), hiplib.iptr(nn), 1, big.ctypes.data_as(hiplib.i16p), len(big), i64(off)) == 0
    assert list(off[:2]) == [0, 3 * 1920] and lib.voc_last_batch_chunks(h) == 1 and lib.voc_last_batch_ms(h) > 0
    lib.voc_free(h)
