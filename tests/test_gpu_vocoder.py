"""Vocoder library (voc_* ABI) against an independent torch fp32 reference of the same op table,
and its chunking/crossfade/int16 path against the restatement that is pinned to the reference's own
VocoderServer.synthesize (tests/test_golden_frontend.py).  Waveform tolerance: both arithmetic modes
(exact-fp32 MFMA; the default 2xfp16 split operands with f32 accumulation) differ from torch CPU fp32
only at the level of fp32 summation order: 2e-4 of full scale is asserted (observed < 1e-6), and
against a float64 evaluation the split path must be as close as an fp32 implementation is."""
import os

import numpy as np
import pytest

from oracle import frontend as fe
from qwen3_tts_axera_russian_amd import hiplib
from qwen3_tts_axera_russian_amd import weights as W
from tests.util import CACHE
from oracle.voc_ref import voc_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_voc():
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "voc_tiny_s7b.q3w")
    vc = W.tiny_voc_config()
    if not os.path.exists(path):
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(vc, seed=7))
    _, tensors = W.read_pack(path)
    return path, vc, tensors


class Voc:
    def __init__(self, lib, path, chunk=64, max_batch=4):
        self.lib = lib
        self.h = lib.voc_load(path.encode(), chunk, max_batch)
        assert self.h
        self.chunk, self.spt = lib.voc_chunk_tokens(self.h), lib.voc_samples_per_token(self.h)
        self.cs = lib.voc_chunk_samples(self.h)      # what a decode returns per chunk (<= chunk * spt)

    def decode(self, codes):
        codes = np.ascontiguousarray(codes, np.int64)
        B = codes.shape[0]
        out = np.empty((B, self.cs), np.float32)
        assert self.lib.voc_decode(self.h, codes.ctypes.data_as(hiplib.i64p), B, hiplib.fptr(out)) == 0
        return out

    def synth_f32(self, codes):
        codes = np.ascontiguousarray(codes, np.int64)
        n = codes.shape[0]
        out = np.empty(self.lib.voc_synthesize_max_samples(self.h, n), np.float32)
        ns = np.zeros(1, np.int32)
        assert self.lib.voc_synthesize_f32(self.h, codes.ctypes.data_as(hiplib.i64p), n, hiplib.fptr(out), hiplib.iptr(ns)) == 0
        return out[:ns[0]]

    def synth_i16(self, codes):
        codes = np.ascontiguousarray(codes, np.int64)
        n = codes.shape[0]
        out = np.empty(self.lib.voc_synthesize_max_samples(self.h, n), np.int16)
        ns = np.zeros(1, np.int32)
        assert self.lib.voc_synthesize(self.h, codes.ctypes.data_as(hiplib.i64p), n, out.ctypes.data_as(hiplib.i16p), hiplib.iptr(ns)) == 0
        return out[:ns[0]]

    def close(self):
        self.lib.voc_free(self.h)


@pytest.mark.parametrize("exact", [0, 1])
def test_decode_matches_torch_reference(gpu_lib, tiny_voc, exact):
    path, vc, tensors = tiny_voc
    gpu_lib.voc_set_exact_fp32(exact)
    v = Voc(gpu_lib, path)
    assert v.chunk == 64 and v.spt == 1920 == W.voc_total_upsample(vc)
    assert v.cs == W.voc_chunk_samples(vc, 64) == 122325      # the decoder family's trim: 64 frames -> 122 325 samples
    rng = np.random.default_rng(3)
    codes = rng.integers(0, 2048, size=(3, 64, 16)).astype(np.int64)
    codes[2, 40:] = 0                      # the server's zero padding of a short chunk
    codes[1, 5, 3] = 5000                  # out-of-range id embeds as zeros
    got = v.decode(codes)
    ref = voc_reference(tensors, codes)
    assert got.shape == ref.shape == (3, 122325)
    err = np.abs(got - ref).max()
    print("vocoder max abs err:", err, "ref max:", np.abs(ref).max(), "clamped frac:", float((np.abs(ref) >= 1).mean()))
    assert np.abs(ref).max() > 0.05        # a live signal, not silence
    assert err < 2e-4
    assert np.abs(got).max() <= 1.0        # final clamp
    v.close()
    gpu_lib.voc_set_exact_fp32(0)


def test_split_arithmetic_is_fp32_grade(gpu_lib, tiny_voc):
    """Error against a float64 evaluation of the same table: the default split path (two fp16 terms per
    operand, f32 accumulation) must be no further from it than fp32 implementations are (the exact-fp32
    MFMA path and torch CPU fp32), up to a factor 2 and a 1e-6 floor."""
    path, vc, tensors = tiny_voc
    codes = np.random.default_rng(5).integers(0, 2048, size=(2, 64, 16)).astype(np.int64)
    ref64 = voc_reference(tensors, codes, dtype=np.float64)
    ref32 = voc_reference(tensors, codes)
    live = np.abs(ref64) < 0.999           # clamped samples carry no information
    errs = {}
    for exact in (1, 0):
        gpu_lib.voc_set_exact_fp32(exact)
        v = Voc(gpu_lib, path)
        errs[exact] = float(np.abs(v.decode(codes) - ref64)[live].max())
        v.close()
    gpu_lib.voc_set_exact_fp32(0)
    e_torch = float(np.abs(ref32 - ref64)[live].max())
    print("max err vs float64: split", errs[0], "exact f32 MFMA", errs[1], "torch f32", e_torch)
    assert errs[0] <= max(2.0 * max(errs[1], e_torch), 1e-6)


def test_synthesize_chunk_walk_and_int16(gpu_lib, tiny_voc):
    path, vc, tensors = tiny_voc
    v = Voc(gpu_lib, path, max_batch=1)
    rng = np.random.default_rng(4)
    for n in (1, 10, 64, 65, 97, 150):     # incl. the reference's length quirk (150 -> 156 frames)
        codes = rng.integers(0, 2048, size=(n, 16)).astype(np.int64)
        got = v.synth_f32(codes)
        want = fe.voc_synthesize(codes, lambda padded: v.decode(padded)[0], 64)
        assert len(got) == len(want), n
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(v.synth_i16(codes), fe.to_int16(want))
    # the lengths the reference's own synthesize produces around a model that returns 122 325 samples per chunk
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_golden.npz"))
    for n in (int(x) for x in gold["vocshort_ns"]):
        assert len(v.synth_f32(np.zeros((n, 16), np.int64))) == int(gold[f"vocshort_122325_{n}_len"]), n
    v.close()


def test_strictly_causal_trim_is_a_table_parameter(gpu_lib, tmp_path):
    """`convt_trim="right"` (rounds 1-2): the transposed convs keep their first L * stride outputs -- 1920 samples per
    frame exactly; same kernels, other trims in the table rows.  Decode + chunk walk against the oracle."""
    vc = W.tiny_voc_config()
    vc.convt_trim = "right"
    t = W.make_synthetic_voc(vc, seed=7)
    path = str(tmp_path / "voc_right.q3w")
    W.write_pack(path, {"voc_chunk": 64.0}, t)
    for exact in (1, 0):
        gpu_lib.voc_set_exact_fp32(exact)
        v = Voc(gpu_lib, path, max_batch=2)
        assert v.cs == 64 * 1920 == W.voc_chunk_samples(vc, 64)
        codes = np.random.default_rng(8).integers(0, 2048, size=(2, 64, 16)).astype(np.int64)
        got, ref = v.decode(codes), voc_reference(t, codes)
        assert got.shape == ref.shape == (2, 122880) and np.abs(ref).max() > 0.05
        assert np.abs(got - ref).max() < 2e-4
        assert len(v.synth_f32(np.zeros((150, 16), np.int64))) == 156 * 1920      # the reference's length quirk, full chunks
        v.close()
    gpu_lib.voc_set_exact_fp32(0)


@pytest.mark.parametrize("name", ["omni", "omni_b", "tts"])
def test_code2wav_golden_on_the_gpu(gpu_lib, tmp_path, name):
    """voc_decode against the outputs of the importable implementation of the reference's vocoder family
    (transformers' Qwen3OmniMoeCode2Wav; Mimi split RVQ front in the "tts" case) stored by
    tests/golden/make_code2wav_golden.py: state-dict keys -> weights.state_to_voc -> container -> voc_load at the
    fixture's own frame count -> waveform, in both arithmetic modes, 2e-4 of full scale; stage activations through
    voc_debug_run.  (CPU twin: tests/test_code2wav_golden.py, 1e-5.)"""
    import ctypes
    from tests import c2w_common as C
    from tests.test_code2wav_golden import load_case
    case, vc, tens, codes, gold, _ = load_case(name)
    path = str(tmp_path / f"c2w_{name}.q3w")
    W.write_pack(path, {"voc_chunk": float(case["T"])}, tens)
    lib = gpu_lib
    lib.voc_debug_run.restype = ctypes.c_int
    lib.voc_debug_run.argtypes = [ctypes.c_void_p, hiplib.i64p, ctypes.c_int, ctypes.c_int, hiplib.f32p, hiplib.i32p, hiplib.i32p]
    for exact in (1, 0):
        lib.voc_set_exact_fp32(exact)
        v = Voc(lib, path, chunk=case["T"], max_batch=1)
        assert v.chunk == case["T"] and v.cs == gold["wav"].shape[0] and v.spt == W.voc_total_upsample(vc)
        wav = v.decode(codes)[0]
        err = float(np.abs(wav - gold["wav"]).max())
        print(f"{name} exact={exact}: waveform max abs err vs the golden {err:.2e} (signal max {np.abs(gold['wav']).max():.2f})")
        assert err < 2e-4
        for stage, n_ops in C.stage_ops(vc, tens["voc.program"]).items():
            want = gold[stage]
            Cc, L = np.zeros(1, np.int32), np.zeros(1, np.int32)
            buf = np.empty(want.shape[0] * 20000, np.float32)
            assert lib.voc_debug_run(v.h, codes.ctypes.data_as(hiplib.i64p), 1, n_ops, hiplib.fptr(buf), hiplib.iptr(Cc), hiplib.iptr(L)) == 0
            act = buf[: int(Cc[0]) * int(L[0])].reshape(int(Cc[0]), int(L[0]))
            got = act[:, C.column_subset(act.shape[1])]
            assert got.shape == want.shape, (stage, act.shape)
            e = float(np.abs(got - want).max() / max(1.0, float(np.abs(want).max())))
            assert e < 2e-4, (name, exact, stage, e)
        v.close()
    lib.voc_set_exact_fp32(0)


def test_full_size_table_is_causal_and_deterministic(gpu_lib, tmp_path_factory):
    """Size-independent properties at the full default table (the bench's vocoder): every op looks back only
    (but for the one-column look-ahead of the transposed convs), so the first 47 frames' samples must not change
    -- bit for bit -- when the last 16 frames' codes do (what chunked streaming with overlap relies on,
    vocoder_server.py:84-117);
    a chunk decodes to the same samples alone and inside a batch; and twice the same input gives the same bits."""
    path = os.path.join(CACHE, "voc_whole_s1234.q3w")
    os.makedirs(CACHE, exist_ok=True)
    if not os.path.exists(path):
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(W.VocConfig(), seed=1234))
    v = Voc(gpu_lib, path, max_batch=3)
    rng = np.random.default_rng(9)
    a = rng.integers(0, 2048, size=(64, 16)).astype(np.int64)
    b = a.copy()
    b[48:] = rng.integers(0, 2048, size=(16, 16))
    c = rng.integers(0, 2048, size=(64, 16)).astype(np.int64)
    out = v.decode(np.stack([a, b, c])).copy()
    assert np.abs(out).max() > 0.01 and np.isfinite(out).all()
    # (the family's transposed convs look ONE input column ahead per block -- a quarter frame in all -- so the first
    # 47 frames are the ones that cannot see frame 48)
    np.testing.assert_array_equal(out[0, :47 * 1920], out[1, :47 * 1920])
    assert not np.array_equal(out[0, 48 * 1920:], out[1, 48 * 1920:])
    np.testing.assert_array_equal(v.decode(a[None])[0], out[0])          # alone == inside a batch
    np.testing.assert_array_equal(v.decode(np.stack([a, b, c])), out)    # deterministic
    # coarse speed guard (a register spill in the conv kernel once cost 2.6x unnoticed by the parity checks):
    # 3 chunks = 0.92 TFLOP of fp32-equivalent work take ~7 ms on an MI355X; 30 ms means something broke
    ms = float(gpu_lib.voc_last_decode_ms(v.h))
    print("full table, 3 chunks:", ms, "ms")
    assert ms < 30.0
    v.close()


def test_capped_grid_decodes_to_the_same_bits(gpu_lib):
    """voc_set_max_workgroups(n): every kernel walks its tiles persistently with at most n workgroups (-1: one per compute unit,
    the setting that leaves room for the frame loop's workgroups when the two run side by side).  The tile a column belongs to
    and the order of its sums do not change: same bits as the one-workgroup-per-tile launch, at the full-size table, for a cap
    below and a cap above the smaller ops' tile counts and for the per-CU setting."""
    path = os.path.join(CACHE, "voc_whole_s1234.q3w")
    os.makedirs(CACHE, exist_ok=True)
    if not os.path.exists(path):
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(W.VocConfig(), seed=1234))
    v = Voc(gpu_lib, path, max_batch=3)
    codes = np.random.default_rng(23).integers(0, 2048, size=(3, 64, 16)).astype(np.int64)
    gpu_lib.voc_set_max_workgroups(0)
    want = v.decode(codes).copy()
    try:
        for cap in (64, 256, -1):
            assert gpu_lib.voc_set_max_workgroups(cap) == (cap if cap > 0 else gpu_lib.q3_device_compute_units())
            np.testing.assert_array_equal(v.decode(codes), want)
    finally:
        gpu_lib.voc_set_max_workgroups(0)
    v.close()


def test_short_chunks_decode_at_their_own_length_to_the_same_bits(gpu_lib):
    """The chunk walk decodes a chunk of n < 64 frames at n + 1 frames rounded up to 8 (voc_decode_frames) instead of the
    reference's zero-padded 64 (vocoder_server.py:78-81): the decoder is causal but for a quarter frame of look-ahead,
    so the samples the walk keeps (n * 1920) must be the SAME BITS as the padded 64-frame decode's -- at the full-size
    table, where the shorter activations would otherwise pick other kernel variants (other summation orders), for the
    single-utterance entry point and for the batched one (chunks of one decode length share a launch)."""
    path = os.path.join(CACHE, "voc_whole_s1234.q3w")
    os.makedirs(CACHE, exist_ok=True)
    if not os.path.exists(path):
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(W.VocConfig(), seed=1234))
    v = Voc(gpu_lib, path, max_batch=4)
    rng = np.random.default_rng(17)
    lens = [1, 6, 7, 8, 20, 33, 47, 55, 56, 63, 64, 70]
    utts = [rng.integers(0, 2048, size=(n, 16)).astype(np.int64) for n in lens]
    want = [fe.voc_synthesize(c, lambda padded: v.decode(padded)[0], 64) for c in utts]
    for c, w in zip(utts, want):
        np.testing.assert_array_equal(v.synth_f32(c), w)
    nn = np.array(lens, np.int32)
    cat = np.ascontiguousarray(np.concatenate(utts, axis=0))
    cap = int(gpu_lib.voc_synthesize_batch_max_samples(v.h, hiplib.iptr(nn), len(nn)))
    out = np.empty(cap, np.float32)
    off = np.zeros(len(lens) + 1, np.int64)
    assert gpu_lib.voc_synthesize_batch_f32(v.h, cat.ctypes.data_as(hiplib.i64p), hiplib.iptr(nn), len(nn), hiplib.fptr(out), cap,
                                            off.ctypes.data_as(hiplib.i64p)) == 0
    short_ms = float(gpu_lib.voc_last_batch_ms(v.h))
    for u, w in enumerate(want):
        np.testing.assert_array_equal(out[off[u]:off[u + 1]], w)
    v.close()
    print("12 utterances (13 chunks, 450 frames) at their own lengths:", short_ms, "ms")


def test_full_size_table_matches_torch_reference_in_both_arithmetic_modes(gpu_lib):
    """Numerics (not only properties) at the FULL default table -- the vocoder the benchmark times, with its
    96-row / 128- and 256-column tiles, XCD tile order and every kernel variant of the 1536 -> 96 channel trunk:
    one 64-frame chunk against oracle/voc_ref.py (torch CPU fp32, ~1 s) in the exact-fp32 mode and in the
    2 x fp16 split-operand mode, tolerance 2e-4 of full scale (the waveform is clamped to [-1, 1]); plus the
    second chunk of a batch of two, so the batch index of the tiling is covered."""
    path = os.path.join(CACHE, "voc_whole_s1234.q3w")
    os.makedirs(CACHE, exist_ok=True)
    if not os.path.exists(path):
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(W.VocConfig(), seed=1234))
    _, tensors = W.read_pack(path)
    rng = np.random.default_rng(21)
    codes = rng.integers(0, 2048, size=(2, 64, 16)).astype(np.int64)
    ref = voc_reference(tensors, codes)
    assert ref.shape == (2, 122325) and np.abs(ref).max() > 0.01
    for exact in (1, 0):
        gpu_lib.voc_set_exact_fp32(exact)
        v = Voc(gpu_lib, path, max_batch=2)
        out = v.decode(codes).copy()
        one = v.decode(codes[1:2]).copy()
        v.close()
        err = float(np.abs(out - ref).max())
        print(f"full-size table, exact={exact}: max abs err vs torch fp32 {err:.2e} (signal max {np.abs(ref).max():.3f})")
        assert err < 2e-4
        np.testing.assert_array_equal(one[0], out[1])
    gpu_lib.voc_set_exact_fp32(0)


def test_fused_residual_units_equal_the_per_conv_launches(gpu_lib):
    """The 96- / 192-channel residual units run as one launch on the exact path (voc_set_fused_units, default on):
    the dilated 7-tap conv's accumulators feed the 1x1 conv as MFMA operands in registers.  Same arithmetic except
    the order in which the 1x1 conv sums its channels: outputs of the whole default table agree to 2e-6 of full
    scale with the three-launch form, at a ragged batch (3 chunks) so the column tail of the tiling is covered."""
    path = os.path.join(CACHE, "voc_whole_s1234.q3w")
    os.makedirs(CACHE, exist_ok=True)
    if not os.path.exists(path):
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(W.VocConfig(), seed=1234))
    codes = np.random.default_rng(33).integers(0, 2048, size=(3, 64, 16)).astype(np.int64)
    gpu_lib.voc_set_exact_fp32(1)
    outs = []
    for fused in (1, 0):
        gpu_lib.voc_set_fused_units(fused)
        v = Voc(gpu_lib, path, max_batch=3)
        outs.append(v.decode(codes).copy())
        ms = gpu_lib.voc_last_decode_ms(v.h)
        v.close()
        print(f"fused={fused}: {ms:.2f} ms for 3 chunks")
    gpu_lib.voc_set_fused_units(1)
    gpu_lib.voc_set_exact_fp32(0)
    err = float(np.abs(outs[0] - outs[1]).max())
    print(f"fused vs per-conv launches: max abs diff {err:.2e}")
    assert np.abs(outs[1]).max() > 0.01 and err < 2e-6


def test_decode_from_a_converted_speech_tokenizer_directory(gpu_lib, tmp_path):
    """A speech_tokenizer/ directory (VOC_NAMES layout, layer scales as separate tensors) converted by
    weights.convert_speech_tokenizer decodes to the waveform of the table it was exported from (2e-6: the layer
    scales are divided out and folded back in f32)."""
    vc = W.tiny_full_voc_config()
    t = W.make_synthetic_voc(vc, seed=5)
    a = str(tmp_path / "direct.q3w")
    W.write_pack(a, {"voc_chunk": 64.0}, t)
    W.export_speech_tokenizer_layout(t, vc, str(tmp_path / "speech_tokenizer"))
    b = str(tmp_path / "converted.q3w")
    vc2, report = W.convert_speech_tokenizer(str(tmp_path / "speech_tokenizer"), b)
    print(report[0])
    codes = np.random.default_rng(6).integers(0, 2048, size=(1, 64, 16)).astype(np.int64)
    outs = []
    for path in (a, b):
        v = Voc(gpu_lib, path)
        outs.append(v.decode(codes).copy())
        v.close()
    assert np.abs(outs[0]).max() > 1e-3
    np.testing.assert_allclose(outs[1], outs[0], rtol=0, atol=2e-6)


@pytest.fixture(scope="module")
def tiny_full_voc():
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "voc_tiny_full_s7.q3w")
    vc = W.tiny_full_voc_config()
    if not os.path.exists(path):
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(vc, seed=7))
    _, tensors = W.read_pack(path)
    return path, vc, tensors


@pytest.mark.parametrize("exact", [0, 1])
def test_transformer_and_convnext_ops_match_torch_reference(gpu_lib, tiny_full_voc, exact):
    """The op kinds of the published decoder's other stages -- sliding-window RoPE attention, RMSNorm /
    LayerNorm over channels, gated MLP, causal depthwise conv, GELU -- stage by stage and end to end
    (window 24 < chunk 64: the window edge is exercised).  Hyper-parameters of the real model are not
    in the reference: this pins the op semantics, parity with the real decoder stays unpinned."""
    import ctypes
    path, vc, tensors = tiny_full_voc
    lib = gpu_lib
    lib.voc_set_exact_fp32(exact)
    v = Voc(lib, path, max_batch=2)
    rng = np.random.default_rng(13)
    codes = rng.integers(0, 2048, size=(2, 64, 16)).astype(np.int64)
    prog = np.asarray(tensors["voc.program"])
    lib.voc_debug_run.restype = ctypes.c_int
    lib.voc_debug_run.argtypes = [ctypes.c_void_p, hiplib.i64p, ctypes.c_int, ctypes.c_int, hiplib.f32p, hiplib.i32p, hiplib.i32p]
    kinds = {}
    for n in range(2, len(prog) + 1):
        op = int(prog[n - 1][0])
        if op in kinds and n != len(prog):
            continue                       # the first occurrence of every op kind, and the whole table
        kinds[op] = n
        ref = voc_reference(tensors, codes, n_ops=n)
        out = np.empty(ref.shape, np.float32)
        C, L = np.zeros(1, np.int32), np.zeros(1, np.int32)
        assert lib.voc_debug_run(v.h, codes.ctypes.data_as(hiplib.i64p), 2, n, hiplib.fptr(out), hiplib.iptr(C), hiplib.iptr(L)) == 0
        assert (2, int(C[0]), int(L[0])) == ref.shape, (n, C, L, ref.shape)
        err = float(np.abs(out - ref).max())
        print(f"after op {n - 1} (kind {op}): max abs err {err:.2e} (ref max {np.abs(ref).max():.2f})")
        assert err < 2e-4 * max(1.0, float(np.abs(ref).max())), (n, op, err)
    assert set(kinds) >= {W.VOP_CONV, W.VOP_CONVT, W.VOP_DWCONV, W.VOP_NORM, W.VOP_ATTN, W.VOP_GLU}
    got = v.decode(codes)
    ref = voc_reference(tensors, codes)
    assert np.abs(ref).max() > 0.05 and np.abs(got - ref).max() < 2e-4
    v.close()
    lib.voc_set_exact_fp32(0)


def test_split_arithmetic_falls_back_when_out_of_fp16_range(gpu_lib, tiny_voc, tmp_path):
    """Two fp16 terms cannot carry |x| > 65504: a call whose activations leave that range is redone on the
    exact-fp32 path (bit-identical to the exact mode), and an op with such a weight never leaves it."""
    path, vc, tensors = tiny_voc
    big = {k: np.array(v) for k, v in tensors.items()}
    big["voc.op0.codebook"] = (big["voc.op0.codebook"] * 3.0e5).astype(np.float32)     # RVQ output ~1e6
    big["voc.op1.weight"] = (big["voc.op1.weight"] * 1.0e-6).astype(np.float32)        # ... scaled back by the first conv
    p_act = str(tmp_path / "voc_big_act.q3w")
    W.write_pack(p_act, {"voc_chunk": 64.0}, big)
    bigw = {k: np.array(v) for k, v in tensors.items()}
    w2 = bigw["voc.op2.weight"].copy()
    w2.flat[0] = 1.0e5                                                                  # one weight beyond fp16
    bigw["voc.op2.weight"] = w2
    p_w = str(tmp_path / "voc_big_w.q3w")
    W.write_pack(p_w, {"voc_chunk": 64.0}, bigw)
    codes = np.random.default_rng(17).integers(0, 2048, size=(2, 64, 16)).astype(np.int64)
    for p in (p_act, p_w):
        outs = {}
        for exact in (1, 0):
            gpu_lib.voc_set_exact_fp32(exact)
            v = Voc(gpu_lib, p, max_batch=2)
            outs[exact] = v.decode(codes).copy()
            v.close()
        gpu_lib.voc_set_exact_fp32(0)
        assert np.isfinite(outs[0]).all()
        if p == p_act:
            np.testing.assert_array_equal(outs[0], outs[1])        # the whole call was redone exactly
        else:
            assert np.abs(outs[0] - outs[1]).max() < 2e-4          # that op exact, the others split


@pytest.mark.parametrize("trim", ["both", "right"])
def test_batched_chunk_walk_is_bit_identical_to_the_per_utterance_walk(gpu_lib, tmp_path, trim):
    """voc_synthesize_batch (BASELINE configs[2]: the overlap-crossfade vocoder at batch): utterances of ragged lengths --
    single chunk, exactly one chunk, the first blend, the appended short tail (n = 97 -> a 1-frame chunk; 150 -> 6 frames),
    a chunk shorter than the overlap -- with max_batch 5, so decode batches mix chunks of different utterances and split
    utterances across batches (the blend then folds into samples an EARLIER batch placed).  Per utterance the result is
    bit-identical to the restatement pinned to the reference's VocoderServer.synthesize (oracle/frontend.voc_synthesize
    around single-chunk decodes), as float32 and as int16, in both arithmetic modes and for both trims of the table."""
    vc = W.tiny_voc_config()
    vc.convt_trim = trim
    path = str(tmp_path / f"voc_{trim}.q3w")
    W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(vc, seed=7))
    rng = np.random.default_rng(41)
    lens = [1, 10, 64, 65, 97, 150, 33, 112, 80]
    utts = [rng.integers(0, 2048, size=(n, 16)).astype(np.int64) for n in lens]
    nn = np.array(lens, np.int32)
    cat = np.ascontiguousarray(np.concatenate(utts, axis=0))
    for exact in (1, 0):
        gpu_lib.voc_set_exact_fp32(exact)
        v = Voc(gpu_lib, path, max_batch=5)
        cap = int(gpu_lib.voc_synthesize_batch_max_samples(v.h, hiplib.iptr(nn), len(nn)))
        assert cap == sum(gpu_lib.voc_synthesize_max_samples(v.h, n) for n in lens)
        off = np.zeros(len(lens) + 1, np.int64)
        out = np.empty(cap, np.float32)
        assert gpu_lib.voc_synthesize_batch_f32(v.h, cat.ctypes.data_as(hiplib.i64p), hiplib.iptr(nn), len(nn), hiplib.fptr(out), cap,
                                                off.ctypes.data_as(hiplib.i64p)) == 0
        out16 = np.empty(cap, np.int16)
        off16 = np.zeros(len(lens) + 1, np.int64)
        assert gpu_lib.voc_synthesize_batch(v.h, cat.ctypes.data_as(hiplib.i64p), hiplib.iptr(nn), len(nn),
                                            out16.ctypes.data_as(hiplib.i16p), cap, off16.ctypes.data_as(hiplib.i64p)) == 0
        np.testing.assert_array_equal(off, off16)
        assert gpu_lib.voc_last_batch_chunks(v.h) == sum(1 if n <= 64 else len(range(0, n, 48)) for n in lens)
        for u, codes in enumerate(utts):
            want = fe.voc_synthesize(codes, lambda padded: v.decode(padded)[0], 64)
            got = out[off[u]:off[u + 1]]
            assert len(got) == len(want), (u, lens[u])
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(out16[off[u]:off[u + 1]], fe.to_int16(want))
            np.testing.assert_array_equal(got, v.synth_f32(codes))             # == the single-utterance entry point
        # too small a caller buffer is refused, nothing is written past it
        small = np.full(int(off[-1]) - 1, 7.0, np.float32)
        assert gpu_lib.voc_synthesize_batch_f32(v.h, cat.ctypes.data_as(hiplib.i64p), hiplib.iptr(nn), len(nn), hiplib.fptr(small),
                                                len(small), off.ctypes.data_as(hiplib.i64p)) == -1
        assert (small == 7.0).all()
        v.close()
    gpu_lib.voc_set_exact_fp32(0)
