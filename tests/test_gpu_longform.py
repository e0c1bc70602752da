"""BASELINE configs[4]: long-form (>= 60 s of audio) synthesis of one utterance -- 768 frames = 61.4 s -- through the
frame engine (n_ctx lifted beyond the reference's 512) and the vocoder's overlap-crossfade chunk walk.

A 768-frame oracle walk at full depth takes 8 minutes of CPU: it is a committed fixture (tests/golden/longform_f768.npz,
made by tests/golden/make_longform_golden.py) and test_long_form_every_decision_of_768_frames_graded teacher-forces the
device through all of it.  Besides: (1) decode steps ACROSS the reference's 512-position limit are graded decision by decision, teacher-forced,
from a 470-row prefix (cheap for the oracle: rows of a prefill share the weight reads) over 72 frames = positions
470..541; (2) the 768-frame run is checked on the device: frame f of a long run equals the same frame of a shorter
run from the same start (the loop has no length-dependent state), ids in range, deterministic; (3) its waveform -- one
request of 768 frames = 17 overlap-crossfade chunks -- is bit-exact against the restatement pinned to
VocoderServer.synthesize (vocoder_server.py:73-121); (4) the full-depth model runs 768 frames at the benchmark's
speed."""
import numpy as np
import pytest

from oracle import frontend as fe
from oracle.pipeline import CpuPipeline
from qwen3_tts_axera_russian_amd import hiplib
from qwen3_tts_axera_russian_amd import weights as W
from qwen3_tts_axera_russian_amd.engine import FrameEngine
from tests.test_gpu_engine import NEAR_TIE, _grade_teacher_forced
from tests.util import CACHE, synthetic_pack

pytestmark = pytest.mark.gpu
F = 768


def test_long_form_two_layer_pack(gpu_lib):
    path, cfg, tensors = synthetic_pack(2, 2)
    rng = np.random.default_rng(404)
    pad = (0.05 * rng.standard_normal(1024)).astype(np.float32)
    # (1) across position 512, graded against the oracle
    long_prefix = (0.05 * rng.standard_normal((470, 1024))).astype(np.float32)
    G = 72
    cpu = CpuPipeline(cfg, tensors, n_ctx=470 + G + 2)
    ref, margins = cpu.generate_batch([long_prefix], [400], pad, G, ignore_eos=True)
    assert len(ref[0]) == G
    eng = FrameEngine(path, max_batch=1, n_ctx=470 + G + 8, max_frames=G)
    eng.set_pad_embed(pad)
    eng.start([long_prefix], [400], ignore_eos=True, max_frames=G)
    eng.set_forced_codes(np.array([[ref[0][f]] for f in range(G)], np.int32))
    assert eng.run(G) == G
    dev, _ = eng.codes()
    n, same, flips = _grade_teacher_forced(dev, ref, margins)
    print(f"positions 470..541 (across the reference's n_ctx 512): {same}/{n} decisions identical, gaps of the others",
          sorted(round(m, 6) for *_, m in flips))
    assert n == G * 16 and all(m < NEAR_TIE for *_, m in flips) and len(flips) <= 0.02 * n
    eng.destroy()
    # (2) 768 frames = 61.4 s
    prefix = (0.05 * rng.standard_normal((26, 1024))).astype(np.float32)
    n_ctx = 26 + F + 8
    eng = FrameEngine(path, max_batch=1, n_ctx=n_ctx, max_frames=F)
    eng.set_pad_embed(pad)
    eng.start([prefix], [17], ignore_eos=True, max_frames=F)
    assert eng.run(F) == F
    dev, per = eng.codes()
    dev = dev.copy()
    assert int(per[0]) == F and ((dev >= 0) & (dev < 2048)).all()
    eng.start([prefix], [17], ignore_eos=True, max_frames=96)        # a short run from the same start: same first frames
    assert eng.run(96) == 96
    np.testing.assert_array_equal(eng.codes()[0][:96], dev[:96])
    eng.destroy()
    # the waveform: 768 frames in one request = the multi-chunk overlap-crossfade walk (17 chunks of 64 with 16 overlap)
    vpath = f"{CACHE}/voc_tiny_longform.q3w"
    import os
    if not os.path.exists(vpath):
        W.write_pack(vpath, {"voc_chunk": 64.0}, W.make_synthetic_voc(W.tiny_voc_config(), seed=7))
    h = gpu_lib.voc_load(vpath.encode(), 64, 1)
    assert h
    codes = np.ascontiguousarray(dev[:, 0, :].astype(np.int64))
    cap = gpu_lib.voc_synthesize_max_samples(h, F)
    out = np.empty(cap, np.float32)
    ns = np.zeros(1, np.int32)
    assert gpu_lib.voc_synthesize_f32(h, codes.ctypes.data_as(hiplib.i64p), F, hiplib.fptr(out), hiplib.iptr(ns)) == 0

    def chunk(padded):
        o = np.empty((1, gpu_lib.voc_chunk_samples(h)), np.float32)
        assert gpu_lib.voc_decode(h, np.ascontiguousarray(padded).ctypes.data_as(hiplib.i64p), 1, hiplib.fptr(o)) == 0
        return o[0]
    want = fe.voc_synthesize(codes, chunk, 64)
    # 768 % 48 == 0: no redundant tail chunk; 15 full chunks of 122 325 samples (the family's trim) + the last one's 48
    # frames, 15 overlaps of 30 720 blended away (the reference's own walk over such a model: frontend_golden.npz vocshort_*)
    assert int(ns[0]) == len(want) == 15 * 122325 + 48 * 1920 - 15 * 30720
    np.testing.assert_array_equal(out[:len(want)], want)
    assert int(ns[0]) / 24000.0 >= 60.0
    gpu_lib.voc_free(h)


def test_768_frames_full_depth_is_deterministic_and_in_range(gpu_lib):
    path, cfg, _ = synthetic_pack(28, 5)
    rng = np.random.default_rng(405)
    prefix = (0.03 * rng.standard_normal((26, 1024))).astype(np.float32)
    pad = (0.03 * rng.standard_normal(1024)).astype(np.float32)
    outs = []
    eng = FrameEngine(path, max_batch=1, n_ctx=26 + F + 8, max_frames=F)
    eng.set_pad_embed(pad)
    for _ in range(2):
        eng.start([prefix], [17], ignore_eos=True, max_frames=F)
        assert eng.run(F) == F
        codes, per = eng.codes()
        assert int(per[0]) == F
        outs.append(codes.copy())
    ms = eng.last_run_ms / F
    eng.destroy()
    np.testing.assert_array_equal(outs[0], outs[1])
    assert ((outs[0] >= 0) & (outs[0] < 2048)).all()
    print(f"long-form at full depth: {ms:.3f} ms per frame over {F} frames = RTF {ms / 80.0:.4f} for the frame loop")
    assert ms < 8.0


def test_long_form_every_decision_of_768_frames_graded(gpu_lib):
    """BASELINE configs[4] at the real depth (28 + 5 layers), the utterance the benchmark's long-form leg runs (utterance 0
    of bench.workload(32, 0, 1234)), against the committed oracle trajectory tests/golden/longform_f768.npz: teacher-forced
    with the oracle's ids, ALL 768 x 16 = 12 288 greedy decisions are graded under the one tolerance NEAR_TIE -- KV
    positions 26 ... 793, past the reference's 512-position context and 7x the benchmark regime's depth; then free-running,
    the stream is identical to the oracle's up to a decision whose oracle gap is a near-tie."""
    import os
    import bench
    from tests.golden.make_bench_golden import inputs_sha
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "longform_f768.npz"))
    ids, margins = g["ids"].astype(np.int32), g["margins"].astype(np.float32)
    prefixes, n_text, pad = bench.workload(32, 0, int(g["seed"]))
    prefixes, n_text = prefixes[:1], n_text[:1]
    assert inputs_sha(prefixes, n_text, pad) == bytes(g["inputs_sha"]).decode(), "bench.workload changed: regenerate the fixture"
    assert ids.shape == (F, 16)
    path, cfg, _ = synthetic_pack(28, 5)
    eng = FrameEngine(path, max_batch=1, n_ctx=prefixes[0].shape[0] + F + 8, max_frames=F)
    eng.set_pad_embed(pad)
    eng.start(prefixes, n_text, ignore_eos=True, max_frames=F)
    eng.set_forced_codes(np.ascontiguousarray(ids[:, None, :]))           # [F][1][16]
    assert eng.run(F) == F
    dev, per = eng.codes()
    n, same, flips = _grade_teacher_forced(dev, [[list(map(int, r)) for r in ids]], [[list(map(float, r)) for r in margins]])
    worst = max((m for *_, m in flips), default=0.0)
    by_q = np.zeros(4, int)
    for _, f, _, _ in flips:
        by_q[f * 4 // F] += 1
    print(f"long form, teacher-forced: {same}/{n} decisions identical, {len(flips)} differ (largest oracle gap among them "
          f"{worst:.2e}); flips per quarter of the run: {by_q.tolist()}")
    assert n == F * 16
    assert all(m < NEAR_TIE for *_, m in flips), [x for x in flips if x[3] >= NEAR_TIE][:5]
    assert len(flips) <= 0.02 * n
    # free-running (what the long-form leg of bench.py times): identical up to the first near-tie the device takes the other way
    eng.start(prefixes, n_text, ignore_eos=True, max_frames=F)
    assert eng.run(F) == F
    free, per = eng.codes()
    assert int(per[0]) == F
    eq = (free[:F, 0, :] == ids)
    if not eq.all():
        f = int(np.argmin(eq.all(axis=1)))
        gi = int(np.argmin(eq[f]))
        print(f"long form, free-running: identical for {f} frames, diverges at frame {f} group {gi} (oracle gap {margins[f, gi]:.2e})")
        assert margins[f, gi] < NEAR_TIE
    eng.destroy()
