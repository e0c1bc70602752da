#!/usr/bin/env python3
"""Benchmark of the MI355X Qwen3-TTS hot path (BASELINE.json metric: RTF + 12 Hz codec-tokens/s).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS a
rank; started plainly with --gpus N it first re-launches itself as N ranks through torch.distributed.run (before
anything touches a GPU) and relays rank 0's JSON line.

One "step" = one pass of the hot path over one batch of synthetic utterances: ragged prefill of B
prompts + F autoregressive frames (talker step, 15-group code predictor, feedback) for all of them
+ the fp32 vocoder chunk (F=64 frames -> 5.12 s of 24 kHz audio) of every utterance; the vocoder of
step i is submitted from a second thread on its own stream while the frame loop of step i+1 proceeds (the
streaming arrangement of the reference client, tts_client.py:188-197; the decode then runs on one persistent
workgroup per CU, voc_set_max_workgroups(-1), which is what lets the two share the chip: DESIGN.md section 4).  Weights and prefix embeddings are resident before the
timed region.  Utterances are independent:
the N*B utterances of a job (config 4: 256 = the 32 prompts x 8) are sorted by expected length (3 frames per
text token, the reference's own estimate: llamacpp_talker_server.py:174) and dealt round-robin to the N ranks; no
collective on the data path (weak scaling); the only torch.distributed traffic is the barrier and the
max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0.  Weights are random-init tensors of the Qwen3-TTS-0.6B architecture
(no checkpoint can exist here), decode is greedy with EOS suppressed so the frame count is fixed.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# One hardware queue for all of the process's streams: the library sets this when it is loaded (csrc/q3_common.cpp has
# the measurements); a rank started under torch.distributed initialises the HIP runtime before that, so say it here too.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak (exact-fp32 MFMA)
MFMA_F16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak
FRAME_SEC = 1920.0 / 24000.0      # one codec frame = 80 ms of audio (vocoder_server.py:29-30)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# token counts of the fixed mixed ru/en prompt set (5-40 tokens each; the first is the
# reference's default prompt, tts_client.py:291, which tokenises to ~17 Qwen tokens)
PROMPT_TOKENS = [17, 9, 24, 31, 12, 38, 7, 19, 27, 14, 35, 22, 5, 29, 16, 40,
                 11, 33, 8, 21, 26, 13, 37, 18, 6, 30, 15, 39, 10, 23, 28, 20]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (config 3/4: 32)")
    ap.add_argument("--frames", type=int, default=64, help="frames per utterance per step (one vocoder chunk)")
    ap.add_argument("--chains", type=int, default=0, help="parallel row groups per frame (0 = engine default)")
    ap.add_argument("--voc-wgs", type=int, default=-1,
                    help="workgroups per vocoder launch while it runs beside the frame loop: -1 = one per compute unit (default, the "
                         "measured optimum), 0 = one per tile (no cap), N = N")
    ap.add_argument("--no-vocoder", action="store_true", help="time the talker + code-predictor loop only")
    ap.add_argument("--no-b1", action="store_true", help="skip the batch-1 latency leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-longform", action="store_true", help="skip the long-form latency leg (configs[4])")
    ap.add_argument("--no-ragged", action="store_true", help="skip the natural-length leg (EOS on, continuous batching, batched vocoder)")
    ap.add_argument("--longform-frames", type=int, default=768, help="frames of the long-form leg (768 = 61.4 s of audio)")
    ap.add_argument("--no-timeline", action="store_true",
                    help="skip the in-graph timeline child process (use under rocprofv3: the profiler follows the child, "
                         "whose graph replay it cannot trace); the roofline then comes from the stand-alone launch loop")
    ap.add_argument("--cpu-frames", type=int, default=96, help="frames of the CPU baseline sample (~12 s at 8 frames/s)")
    ap.add_argument("--backend", default=os.environ.get("Q3_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="process-group backend (nccl = RCCL; gloo only for the CPU test of the rank logic)")
    ap.add_argument("--stub-engine", action="store_true", default=os.environ.get("Q3_BENCH_STUB", "") not in ("", "0"),
                    help="CPU test of the launch/rank/dealing logic: a stand-in engine that computes nothing "
                         "(the line is marked data=stub and is not a measurement)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--cache", default=os.environ.get("Q3_BENCH_CACHE", "/tmp/q3_bench_cache"))
    return ap.parse_args()


def make_pack(cache, seed, rank, barrier):
    from qwen3_tts_axera_russian_amd import weights as W
    os.makedirs(cache, exist_ok=True)
    cfg = W.ModelConfig()
    path = os.path.join(cache, f"qwen3tts06b_synth_s{seed}.q3w")
    if rank == 0 and not os.path.exists(path):
        t = time.time()
        W.write_synthetic(path, cfg, seed=seed, parts=("talker", "cp"))
        print(f"[bench] wrote synthetic Qwen3-TTS-0.6B weights ({os.path.getsize(path)/1e9:.2f} GB) in "
              f"{time.time()-t:.0f}s", file=sys.stderr, flush=True)
    barrier()
    return path, cfg


def make_voc_pack(cache, seed, rank, barrier, trunk_only=False):
    """The whole decoder table (split RVQ, 8-layer sliding-window pre-transformer, x2 x2 upsamplers with ConvNeXt
    blocks, BigVGAN-style trunk); trunk_only = round 1's timed table (the convolutional trunk alone)."""
    from qwen3_tts_axera_russian_amd import weights as W
    path = os.path.join(cache, f"qwen3tts_voc_{'trunk' if trunk_only else 'whole'}_s{seed}.q3w")
    if rank == 0 and not os.path.exists(path):
        vc = W.trunk_voc_config() if trunk_only else W.VocConfig()
        W.write_pack(path, {"voc_chunk": 64.0}, W.make_synthetic_voc(vc, seed=seed))
    barrier()
    return path


def deal(n_text_all, world):
    """Utterance indices of every rank: sort the job's utterances by expected length (3 frames per text token,
    llamacpp_talker_server.py:174; ties by index), deal round-robin.  Every rank gets the same
    count (+-1) and the same mix of short and long utterances; no rank waits on a batch of long ones."""
    order = sorted(range(len(n_text_all)), key=lambda i: (-3 * n_text_all[i], i))
    return [order[r::world] for r in range(world)]


def workload_indices(B, rank, world=1):
    """Job-wide indices (0 .. world*B-1) of the utterances in this rank's slots, in slot order."""
    return deal([PROMPT_TOKENS[i % len(PROMPT_TOKENS)] for i in range(world * B)], world)[rank]


def workload(B, rank, seed, world=1):
    """This rank's B utterances of the job's world*B (the fixed prompt set, repeated): prefix rows = n_text + 9
    (llamacpp_talker_server.py:121-161).  The job is the prompt set repeated (BASELINE configs[3]: 256 = the 32 prompts x 8):
    utterance i carries prompt i % 32 -- its length and its seeded embeddings -- whatever rank runs it, and the pad embedding
    is one vector for the job (tts_pad_embed is a model constant, llamacpp_talker_server.py:88-93), so a prompt's trajectory
    does not depend on the number of ranks or on the rank and slot it is dealt to (what the result check relies on)."""
    total = world * B
    n_text_all = [PROMPT_TOKENS[i % len(PROMPT_TOKENS)] for i in range(total)]
    mine = workload_indices(B, rank, world)
    n_text = [n_text_all[i] for i in mine]
    prefixes = [(0.03 * np.random.default_rng(seed + 7919 * (i % len(PROMPT_TOKENS))).standard_normal((n_text_all[i] + 9, 1024))).astype(np.float32)
                for i in mine]
    pad = (0.03 * np.random.default_rng(seed).standard_normal(1024)).astype(np.float32)
    return prefixes, n_text, pad


class StubEngine:
    """Stand-in for FrameEngine in the CPU test of the launch / rank / dealing logic (--stub-engine): same calls,
    no computation.  A line produced with it says data=stub."""
    last_prefill_ms, step_weight_bytes = 0.01, 3.467e9

    def __init__(self, *_, **kw):
        self.max_frames, self.B, self.last_run_ms = kw.get("max_frames", 64), 0, 0.0

    def set_pad_embed(self, pad): pass
    def set_chains(self, n): pass

    def start(self, prefixes, n_text, ignore_eos=False, max_frames=0):
        self.B = len(prefixes)

    def run(self, n):
        time.sleep(1e-4 * n)
        self.last_run_ms = 0.1 * n
        return n

    def codes(self):
        return np.zeros((self.max_frames, self.B, 16), np.int32), np.full(self.B, self.max_frames, np.int32)

    def destroy(self): pass


class Vocoder:
    """voc_* ABI wrapper; decode() is synchronous, so the bench runs it on a worker thread."""

    def __init__(self, lib, path, max_batch, chunk=64):
        from qwen3_tts_axera_russian_amd import hiplib
        self.lib, self.hl = lib, hiplib
        self.h = lib.voc_load(path.encode(), chunk, max_batch)
        if not self.h:
            raise SystemExit("bench.py: voc_load failed")
        self.cap = 0      # workgroups per launch while a frame loop runs beside the decode (run_leg re-arms it; 0 = one per tile)
        self.chunk, self.spt = lib.voc_chunk_tokens(self.h), lib.voc_samples_per_token(self.h)
        self.out = np.empty((max_batch, lib.voc_chunk_samples(self.h)), np.float32)   # the model's output rows (<= chunk * spt)
        self.ms = []

    def decode(self, codes_fb16):
        """codes [F][B][16] int32 from the engine -> waveform [B][voc_chunk_samples]."""
        c = np.ascontiguousarray(np.transpose(codes_fb16, (1, 0, 2)).astype(np.int64))
        B = c.shape[0]
        rc = self.lib.voc_decode(self.h, c.ctypes.data_as(self.hl.i64p), B, self.hl.fptr(self.out))
        assert rc == 0
        self.ms.append(self.lib.voc_last_decode_ms(self.h))
        return self.out[:B]

    def close(self):
        self.lib.voc_free(self.h)


def run_leg(eng, voc, prefixes, n_text, pad, frames, steps, warmup, sync_all):
    """-> (wall s for `steps` steps, mean GPU ms per frame step, mean prefill ms, mean vocoder ms)."""
    from concurrent.futures import ThreadPoolExecutor
    eng.set_pad_embed(pad)
    pool = ThreadPoolExecutor(max_workers=1)

    def one_step(pending, last=False):
        eng.start(prefixes, n_text, ignore_eos=True, max_frames=frames)
        ran = eng.run(frames)
        assert ran == frames
        codes, _ = eng.codes()
        if pending is not None:
            pending.result()
        if voc is None:
            return None
        if last:      # the queue drains: no frame loop runs beside the last decode, it takes the whole chip (one workgroup per tile)
            voc.lib.voc_set_max_workgroups(0)
        return pool.submit(voc.decode, codes.copy())

    pending = None
    if voc is not None:
        voc.lib.voc_set_max_workgroups(voc.cap)      # the grid of a decode that runs beside a frame loop (0 = one workgroup per tile)
    for _ in range(warmup):
        pending = one_step(pending)
    if pending is not None:
        pending.result()
    if voc is not None:
        voc.ms.clear()
    frame_ms, prefill_ms = [], []
    sync_all()
    t0 = time.perf_counter()
    pending = None
    for i in range(steps):
        pending = one_step(pending, last=(i == steps - 1))
        frame_ms.append(eng.last_run_ms / frames)
        prefill_ms.append(eng.last_prefill_ms)
    if pending is not None:
        pending.result()
    sync_all()
    dt = time.perf_counter() - t0
    pool.shutdown()
    run_leg.last_codes = eng.codes()[0].copy()      # what the LAST timed step computed (checked outside the timed region)
    return dt, float(np.mean(frame_ms)), float(np.mean(prefill_ms)), float(np.mean(voc.ms)) if voc is not None else 0.0


NEAR_TIE = 5e-3     # tests/test_gpu_engine.py: THE tolerance -- oracle top-1/top-2 gap below which two float pipelines may differ


def verify_first_utterance(codes_f16, seed, frames=64):
    """The batch-1 and long-form legs run the job's first utterance alone: its leading `frames` frames against the same fixture
    (utterance 0 of tests/golden/bench_b32_f64.npz), identical up to a decision whose oracle gap is a near-tie."""
    fx = os.path.join(ROOT, "tests", "golden", "bench_b32_f64.npz")
    if not os.path.exists(fx):
        return {"checked": False, "why": "fixture not found"}
    g = np.load(fx)
    if int(g["seed"]) != seed:
        return {"checked": False, "why": f"fixture is for seed {int(g['seed'])}"}
    ids, margins = g["ids"][0].astype(np.int32), g["margins"][0].astype(np.float32)
    n = min(frames, ids.shape[0], codes_f16.shape[0])
    eq = (codes_f16[:n] == ids[:n])
    if eq.all():
        return {"checked": True, "ok": True, "frames_compared": int(n), "identical_leading_frames": int(n)}
    f = int(np.argmin(eq.all(axis=1)))
    gap = float(margins[f, int(np.argmin(eq[f]))])
    return {"checked": True, "ok": bool(gap < NEAR_TIE), "frames_compared": int(n), "identical_leading_frames": f,
            "oracle_gap_at_the_divergence": round(gap, 6)}


def grade_slots(codes, F, slot_to_fixture, ids, margins):
    """Grade this rank's slots that the fixture covers: slot s against ids[slot_to_fixture[s]].  Free-running greedy
    streams of two float pipelines are identical up to a decision whose oracle top-1/top-2 gap is a float near-tie
    (< NEAR_TIE).  -> (leading identical frames per graded slot, decisions that diverge at no near-tie, largest gap at a
    divergence)."""
    lead, bad, worst = [], [], 0.0
    for s_, b in sorted(slot_to_fixture.items()):
        eq = (codes[:F, s_, :] == ids[b])
        if eq.all():
            lead.append(F)
            continue
        f = int(np.argmin(eq.all(axis=1)))
        gidx = int(np.argmin(eq[f]))
        gap = float(margins[b, f, gidx])
        worst = max(worst, gap)
        lead.append(f)
        if not gap < NEAR_TIE:
            bad.append((s_, f, gidx, gap))
    return lead, bad, worst


def fixture_slots(B, rank, world, n_fixture):
    """slot -> fixture row for this rank: the fixture holds the prompt set (n_fixture prompts) in the slot order of the
    single-GPU run (bench.workload(n_fixture, 0, seed)); utterance i of a larger job is prompt i % n_fixture."""
    if n_fixture != len(PROMPT_TOKENS):
        return {}
    row_of = {i: b for b, i in enumerate(workload_indices(n_fixture, 0, 1))}
    return {s_: row_of[i % n_fixture] for s_, i in enumerate(workload_indices(B, rank, world))}


def verify_against_fixture(codes, B, F, seed, world, prefixes, n_text, pad, rank=0, ranks=None):
    """Result check of the benchmark itself (the reference's client reports RTF for audio it really wrote,
    tts_client.py:268-271): the codec ids of the last TIMED step against the committed CPU-oracle trajectory of this
    exact workload (tests/golden/bench_b32_f64.npz, made by tests/golden/make_bench_golden.py) -- a data file, nothing
    under oracle/ is imported.  The fixture holds the 32 prompts; a job of N ranks is that set N times (a prompt's
    trajectory does not depend on its slot, its neighbours or its rank), every rank grades its 32 utterances and the counts
    are summed over the ranks (utterances = 32 N).  Anything but a near-tie fails the benchmark."""
    import hashlib
    fx = os.path.join(ROOT, "tests", "golden", "bench_b32_f64.npz")
    if not os.path.exists(fx):
        return {"checked": False, "why": "tests/golden/bench_b32_f64.npz is missing"}
    g = np.load(fx)
    ids, margins = g["ids"].astype(np.int32), g["margins"].astype(np.float32)
    NB = ids.shape[0]
    # the fixture's own inputs, regenerated: guards the workload function, then this rank's rows against them
    fp, fn, fpad = workload(NB, 0, int(g["seed"]), 1)
    h = hashlib.sha256()
    for p_ in fp:
        h.update(np.ascontiguousarray(p_).tobytes())
    h.update(np.asarray(fn, np.int32).tobytes())
    h.update(np.ascontiguousarray(fpad).tobytes())
    if (B, F, seed) != (NB, ids.shape[1], int(g["seed"])) or h.hexdigest() != bytes(g["inputs_sha"]).decode():
        return {"checked": False, "why": f"fixture is for batch {NB} x {ids.shape[1]} frames, seed {int(g['seed'])}"}
    slots = fixture_slots(B, rank, world, NB)
    same_inputs = np.array_equal(pad, fpad) and all(np.array_equal(prefixes[s_], fp[b]) and n_text[s_] == fn[b]
                                                    for s_, b in slots.items())
    lead, bad, worst = grade_slots(codes, F, slots, ids, margins) if same_inputs else ([], [], 0.0)
    tot = np.array([len(lead), sum(x == F for x in lead), sum(lead), len(bad), 0 if same_inputs else 1], np.float64)
    lead_min = float(min(lead)) if lead else float(F)
    if ranks is not None and world > 1:
        tot = ranks.sum_over_ranks(tot)
        worst = ranks.max_over_ranks(worst)
        lead_min = -ranks.max_over_ranks(-lead_min)
    n = int(tot[0])
    res = {"checked": bool(n > 0 and tot[4] == 0), "fixture": "tests/golden/bench_b32_f64.npz (CPU oracle, same weights and prompts)",
           "utterances": n, "frames": F, "rule": f"identical ids up to a decision whose oracle top-1/top-2 gap < {NEAR_TIE}",
           "ok": bool(n > 0 and tot[3] == 0 and tot[4] == 0), "utterances_identical_over_all_frames": int(tot[1]),
           "identical_leading_frames": {"min": int(lead_min), "total": int(tot[2]), "of": n * F},
           "largest_oracle_gap_at_a_divergence": round(worst, 6)}
    if world == 1:
        res["identical_leading_frames"]["median"] = int(np.median(lead)) if lead else 0
    else:
        res["graded_over_ranks"] = world
    if tot[4]:
        res["why"] = "a rank's inputs differ from the fixture's"
    if bad:
        print(f"[bench] RESULT CHECK FAILED (rank {rank}): codes diverge from the oracle at decisions that are no near-ties: "
              f"{bad[:5]}", file=sys.stderr, flush=True)
    return res


def longform_leg(lib, path, voc_path, prefix, n_text, pad, frames, seed=1234):
    """BASELINE configs[4]: ONE utterance of >= 60 s of audio in latency mode -- prefill, `frames` frame steps, and the
    vocoder's overlap-crossfade chunk walk over the whole utterance (voc_synthesize_f32 = VocoderServer.synthesize,
    vocoder_server.py:73-121: 64-frame chunks stepping by 48).  First audio = prefill + the first 64 frames + their
    chunk.  One untimed pass first (graph capture, LDS attributes)."""
    from qwen3_tts_axera_russian_amd import hiplib
    from qwen3_tts_axera_russian_amd.engine import FrameEngine
    eng = FrameEngine(path, max_batch=1, n_ctx=prefix.shape[0] + frames + 8, max_frames=frames)
    eng.set_pad_embed(pad)
    lib.voc_set_exact_fp32(1)
    h = lib.voc_load(voc_path.encode(), 64, 1)
    if not h:
        raise SystemExit("bench.py: voc_load failed (long-form leg)")
    cap = lib.voc_synthesize_max_samples(h, frames)
    wav = np.empty(cap, np.float32)
    first = np.empty((1, lib.voc_chunk_samples(h)), np.float32)
    ns = np.zeros(1, np.int32)
    res = None
    for timed in (False, True):
        t0 = time.perf_counter()
        eng.start([prefix], [n_text], ignore_eos=True, max_frames=frames)
        assert eng.run(64) == 64
        c0 = np.ascontiguousarray(eng.codes()[0][:64, 0, :].astype(np.int64))[None]
        assert lib.voc_decode(h, c0.ctypes.data_as(hiplib.i64p), 1, hiplib.fptr(first)) == 0
        t_first = time.perf_counter() - t0
        assert eng.run(frames - 64) == frames - 64
        codes = np.ascontiguousarray(eng.codes()[0][:frames, 0, :].astype(np.int64))
        assert lib.voc_synthesize_f32(h, codes.ctypes.data_as(hiplib.i64p), frames, hiplib.fptr(wav), hiplib.iptr(ns)) == 0
        wall = time.perf_counter() - t0
        if timed:
            audio = int(ns[0]) / 24000.0
            res = {"verified": verify_first_utterance(codes.astype(np.int32), seed),
                   "workload": f"configs[4]: one utterance, {frames} frames = {frames * FRAME_SEC:.1f} s of audio, latency mode "
                               "(prefill + frame loop + overlap-crossfade chunk walk, exact-fp32 vocoder)",
                   "frames": frames, "audio_s": round(audio, 2), "wall_s": round(wall, 4), "rtf": round(wall / audio, 5),
                   "first_audio_ms": round(t_first * 1e3, 2), "frame_loop_ms_per_frame": round(eng.last_run_ms / (frames - 64), 4)}
    lib.voc_free(h)
    eng.destroy()
    return res


def ragged_leg(lib, path, voc_path, B, seed, rounds=3, max_frames=256):
    """BASELINE configs[2] as written: mixed prompts at their NATURAL lengths -- EOS bookkeeping on (the adaptive EOS boost
    and the forced EOS of llamacpp_talker_server.py:167-189 end an utterance after ~3 x n_text frames), `rounds` x the 32
    prompts queued through B slots by continuous batching (q3e_refill keeps every slot busy), and the streaming
    overlap-crossfade vocoder at batch: finished utterances are handed to a worker that runs voc_synthesize_batch (the
    chunk walk of vocoder_server.py:84-117 for all of them in one call, int16 out).  Wall time covers prefills, refills,
    the frame loop and every vocoder call; one untimed pass first."""
    from concurrent.futures import ThreadPoolExecutor
    from qwen3_tts_axera_russian_amd import hiplib
    from qwen3_tts_axera_russian_amd.engine import FrameEngine
    total = rounds * len(PROMPT_TOKENS)
    n_text = [PROMPT_TOKENS[i % len(PROMPT_TOKENS)] for i in range(total)]
    order = sorted(range(total), key=lambda i: (-n_text[i], i))            # longest expected first (deal()'s rule)
    prefixes = [(0.03 * np.random.default_rng(seed + 7919 * i).standard_normal((n_text[i] + 9, 1024))).astype(np.float32)
                for i in order]
    n_text = [n_text[i] for i in order]
    pad = (0.03 * np.random.default_rng(seed).standard_normal(1024)).astype(np.float32)
    eng = FrameEngine(path, max_batch=B, n_ctx=max(p.shape[0] for p in prefixes) + max_frames + 8, max_frames=max_frames)
    eng.set_pad_embed(pad)
    lib.voc_set_exact_fp32(1)
    lib.voc_set_max_workgroups(0)     # (the per-CU grid loses here: 6.20 k against 6.91 k frames/s -- small ragged launches, long tails)
    h = lib.voc_load(voc_path.encode(), 64, B)
    if not h:
        raise SystemExit("bench.py: voc_load failed (ragged leg)")
    pool = ThreadPoolExecutor(max_workers=1)
    res = None
    for timed in (False, True):
        done_codes, futures, voc_ms, chunks, samples = [], [], [], [0], [0]

        def vocode(batch):
            n = np.array([len(c) for c in batch], np.int32)
            cat = np.ascontiguousarray(np.concatenate(batch, axis=0), np.int64)
            cap = int(lib.voc_synthesize_batch_max_samples(h, hiplib.iptr(n), len(n)))
            out = np.empty(cap, np.int16)
            off = np.zeros(len(n) + 1, np.int64)
            assert lib.voc_synthesize_batch(h, cat.ctypes.data_as(hiplib.i64p), hiplib.iptr(n), len(n), out.ctypes.data_as(hiplib.i16p),
                                            cap, off.ctypes.data_as(hiplib.i64p)) == 0
            voc_ms.append(float(lib.voc_last_batch_ms(h)))
            chunks[0] += int(lib.voc_last_batch_chunks(h))
            samples[0] += int(off[-1])

        def on_done(i, codes):
            if len(codes) == 0:                      # an utterance whose first decision was EOS: nothing to vocode
                return
            done_codes.append(codes)
            # stream: hand what has finished to the vocoder while the loop goes on (groups of 8 / 16 / 32 / all 96 at the end measure
            # 6.67 / 6.49 / 6.48 / 6.40 k frames/s: the two share the chip, the leg is the sum of their work whatever the grouping)
            if len(done_codes) >= int(os.environ.get("Q3_RAGGED_GROUP", B // 2)):
                futures.append(pool.submit(vocode, list(done_codes)))
                done_codes.clear()
        t0 = time.perf_counter()
        got = eng.generate_queue(prefixes, n_text, max_frames, ignore_eos=False, check_every=8, on_done=on_done)
        t_loop = time.perf_counter() - t0
        if done_codes:
            futures.append(pool.submit(vocode, list(done_codes)))
        for f in futures:
            f.result()
        wall = time.perf_counter() - t0
        if timed:
            frames = [len(g) for g in got]
            audio = samples[0] / 24000.0
            res = {"workload": f"configs[2] at natural lengths: {total} utterances ({rounds} x the 32 mixed prompts), EOS rule on, "
                               f"{B} slots kept busy by continuous batching, batched overlap-crossfade vocoder (exact fp32)",
                   "utterances": total, "frames_total": int(sum(frames)), "frames_min_median_max": [int(min(frames)), int(np.median(frames)),
                                                                                                   int(max(frames))],
                   "wall_s": round(wall, 4), "frame_loop_wall_s": round(t_loop, 4), "value": round(sum(frames) / wall, 1),
                   "unit": "codec_frames/s", "audio_s": round(audio, 2), "rtf_aggregate": round(wall / audio, 6),
                   "vocoder_calls": len(voc_ms), "vocoder_chunks": chunks[0], "vocoder_gpu_ms_total": round(float(sum(voc_ms)), 2)}
    pool.shutdown()
    lib.voc_free(h)
    eng.destroy()
    return res


def kv_bytes_per_step(n_text, frames, cfg):
    """Algorithmic KV reads of one frame step, averaged over the run: talker
    114688 B x T per utterance, CP <= 20480 B x 16 per position."""
    per_pos = cfg.talker_layers * 2 * cfg.n_kv_heads * cfg.head_dim * 2
    t_avg = [n + 9 + frames / 2.0 for n in n_text]
    cp = cfg.cp_layers * 2 * cfg.n_kv_heads * cfg.head_dim * 2 * sum(range(1, 17))
    return sum(per_pos * t for t in t_avg) + cp * len(n_text)


def pmc_traffic_gateup(rows):
    """HBM bytes per launch of the gate/up kernel from the committed PMC passes of the final build -- profiles/r03_pmc_linear.json,
    which scripts/pmc_linear_table.py derives from the raw rocprofv3 rows (r03_pmc_{fetch,write}_counter_collection.csv: FETCH_SIZE
    doubled as the gfx950 guide prescribes + WRITE_SIZE; tests/test_profiles.py re-derives it).  -> (talker variant: cold weights,
    nt loads; in-graph code-predictor variant: weights resident in the Infinity Cache), None where no pass covers the row count."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_pmc_linear.json")) as f:
            js = json.load(f)
    except OSError:
        return None, None
    mt = 2 if rows > 16 else 1
    t = js.get(f"linear_kernel<2, {mt}, 4, 8, 1, 2, true>")
    c = js.get(f"linear_kernel<2, {mt}, 4, 8, 1, 2, false>")
    ok = lambda e: e is not None and int(e["rows"]) == int(rows)
    return (float(t["hbm_bytes_per_launch"]) if ok(t) else None, float(c["hbm_bytes_per_launch"]) if ok(c) else None)


def dominant_kernel_roofline(rows, cache, timeline=True):
    """The dominant kernel of the path by bytes: the fused (RMSNorm-folded) gate/up GEMM -> SwiGLU launch (12.58 MB of
    fp16 weights, 36 % of a layer's stream).  Its launch duration is measured IN the replayed frame graph -- the chain
    the benchmark times, real activations -- by scripts/frame_timeline.py (child process, timeline build of the
    library: device-clock stamps per workgroup, first entry -> last exit, mean over the 103 gate/up nodes of a frame;
    the stamping lengthens a node by ~0.8 us, so the figure is conservative).  HIP events cannot bracket a node of a
    graph replay and rocprofv3's kernel trace does not see inside one; the eager-mode rocprofv3 summary of the same
    command is profiles/r02_bench_kernel_stats.csv.  Falls back to the stand-alone launch loop of the test library
    (q3t_bench_linear) when the timeline library is not built."""
    import subprocess
    N, K = 6144, 1024
    algo = N * K * 2 + rows * K * 2 + rows * (N // 2) * 2      # weights + fp16 activations in + fp16 out
    tl_lib = os.path.join(ROOT, "qwen3_tts_axera_russian_amd", "lib", "libqwen3tts_tl.so")
    us, how, table = None, None, None
    if timeline and os.path.exists(tl_lib):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "frame_timeline.py"), "--json", "--batch",
                                str(int(rows)), "--cache", cache], capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not lines:
                raise RuntimeError(f"no result line (rc {r.returncode}): {r.stderr[-400:]!r}")
            line = lines[-1]
            table = json.loads(line)
            us = float(table["kinds"]["linear norm/swiglu (gate+up)"]["mean_span_us"])
            how = "in-graph (replayed frame step), device-clock stamps per workgroup, timeline build; +~0.8 us stamping per node"
        except Exception as e:      # noqa: BLE001 -- the measurement must not take the benchmark down
            print(f"[bench] frame timeline failed ({e}); falling back to the stand-alone launch loop", file=sys.stderr)
    if us is None:
        from qwen3_tts_axera_russian_amd import hiplib
        us = float(hiplib.load_test().q3t_bench_linear(int(rows), N, K, 1, 2, 1, 48, 480))
        how = "stand-alone back-to-back launches over 48 weight copies (q3t_bench_linear), HIP events"
    ach = algo / (us * 1e-6) / 1e9
    t_talker, t_cp = pmc_traffic_gateup(rows)
    out = {"kernel": "linear_kernel<gate/up + SwiGLU> (RMSNorm folded, MFMA 16x16x32 f16, split-K over waves)",
           "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": t_talker, "traffic_cache_resident": t_cp,
           "traffic_source": "profiles/r03_pmc_linear.json (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE per launch = what the L2s request "
                             "from the fabric; talker variant = cold weights / nt loads, cache_resident = the in-graph code-predictor "
                             "variant, whose requests the Infinity Cache can serve: the counter does not tell)",
           "algorithmic_bytes_per_launch": int(algo), "avg_launch_us": round(us, 3), "rows": int(rows), "measured": how}
    return out, table


def cpu_baseline(path, cfg, prefix, n_text, pad, frames, voc_path=None):
    """The CPU restatement (oracle/, the 'port' baseline) on a bounded sample of the same workload: one
    utterance through prefill + `frames` frames (C/OpenMP), then one 64-frame vocoder chunk (torch CPU fp32)."""
    from oracle import oracle as orc
    from oracle.pipeline import CpuPipeline
    from qwen3_tts_axera_russian_amd import weights as W
    threads = orc.n_threads()
    _, tensors = W.read_pack(path)
    pipe = CpuPipeline(cfg, tensors, n_ctx=prefix.shape[0] + frames + 1)
    t0 = time.perf_counter()
    out = pipe.generate(prefix, n_text, pad, frames, ignore_eos=True)
    dt = time.perf_counter() - t0
    assert len(out) == frames
    res = {"value": round(frames / dt, 3), "unit": "codec_frames/s", "cores": threads, "kind": "port",
           "rtf": round(dt / (frames * FRAME_SEC), 3),
           "sample": f"1 utterance, prefill {prefix.shape[0]} rows + {frames} frames (talker+code predictor, "
                     f"fp32 C/OpenMP restatement of the same fp16-weight contract), {dt:.1f}s"}
    if voc_path is not None:
        from oracle.voc_ref import voc_reference
        _, vt = W.read_pack(voc_path)
        codes = np.asarray(out[:64] + [[0] * 16] * max(0, 64 - len(out)), dtype=np.int64)[None]
        t1 = time.perf_counter()
        wav = voc_reference(vt, codes)
        dv = time.perf_counter() - t1
        assert wav.shape[0] == 1 and wav.shape[1] <= 64 * 1920
        res["vocoder_s_per_chunk"] = round(dv, 2)
        res["rtf_with_vocoder"] = round((dt / frames * 64 + dv) / (64 * FRAME_SEC), 3)
        res["value_with_vocoder"] = round(64 / (dt / frames * 64 + dv), 3)
        res["sample"] += f"; + one 64-frame vocoder chunk (torch CPU fp32, {threads} threads), {dv:.1f}s"
    return res


class Ranks:
    """One process per GPU.  The data path has no collective (utterances are independent); the process
    group only carries the barrier and the max-over-ranks of the elapsed time.  backend "nccl" is RCCL
    on ROCm; "gloo" is used by the CPU tests of this logic."""

    def __init__(self, backend="nccl"):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend, self.dist = backend, None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend=backend)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def sync_all(self):
        if self.dist is not None:
            if self.backend == "nccl":
                import torch
                torch.cuda.synchronize()
            self.dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        """Element-wise sum of a small float64 vector over the ranks."""
        if self.dist is None:
            return np.asarray(x, np.float64)
        import torch
        t = torch.tensor(np.asarray(x, np.float64), dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def aggregate_value(world, B, frames, steps, dt_max):
    """Whole-job codec frames/s: every rank processed B*frames*steps frames in dt_max seconds."""
    return world * B * frames * steps / dt_max


def relaunch_as_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run
    BEFORE this process touches a GPU, relay their output (rank 0 prints the JSON line) and exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_as_ranks(a.gpus))
    R = Ranks(a.backend)
    rank, world, local_rank, dist = R.rank, R.world, R.local_rank, R.dist
    barrier = R.barrier
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)")

    sync_all = R.sync_all  # engine calls are synchronous (every q3e_run ends with a stream sync)
    B, F = a.batch, a.frames
    prefixes, n_text, pad = workload(B, rank, a.seed, world)
    if a.stub_engine:
        eng = StubEngine(max_frames=F)
        dt, frame_ms, _, _ = run_leg(eng, None, prefixes, n_text, pad, F, a.steps, a.warmup, sync_all)
        dt = R.max_over_ranks(dt)
        if rank == 0:
            print(json.dumps({"metric": "12Hz codec-tokens/s (codec frames/s, 16 codes each) + RTF, Qwen3-TTS-0.6B",
                              "value": round(aggregate_value(world, B, F, a.steps, dt), 1), "unit": "codec_frames/s",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                              "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "none", "data": "stub (no computation: launch-logic test)",
                              "config": {"workload": "stub", "batch_per_gpu": B, "frames_per_step": F,
                                         "n_text_rank0": n_text}}), flush=True)
        R.close()
        return

    from qwen3_tts_axera_russian_amd import hiplib
    from qwen3_tts_axera_russian_amd.engine import FrameEngine
    lib = hiplib.load()
    if lib.q3_device_count() <= 0:
        raise SystemExit("bench.py: no HIP device -- the HIP library is the only compute path")
    lib.q3_set_device(local_rank % lib.q3_device_count())

    path, cfg = make_pack(a.cache, a.seed, rank, barrier)
    n_ctx = max(p.shape[0] for p in prefixes) + F + 8
    eng = FrameEngine(path, max_batch=B, n_ctx=n_ctx, max_frames=F)
    if a.chains > 0:
        eng.set_chains(a.chains)
    voc = None
    if not a.no_vocoder:
        if F != 64:
            raise SystemExit("bench.py: the vocoder leg decodes 64-frame chunks; use --frames 64 or --no-vocoder")
        # the decode of step s runs beside the frame loop of step s + 1: one persistent vocoder workgroup per CU leaves the frame
        # loop's workgroups room (include/qwen3tts_voc.h; 244 -> 232 ms per step).  Same tiles, same bits.
        voc_cap = lib.voc_set_max_workgroups(a.voc_wgs)
        voc = Vocoder(lib, make_voc_pack(a.cache, a.seed, rank, barrier), B)
        voc.cap = voc_cap
        # headline arithmetic = exact fp32 (north_star: "fused fp32 HIP kernel"); the 2 x fp16 split-operand mode
        # (fp32-grade against float64) is reported beside it as an option
        lib.voc_set_exact_fp32(1)
    dt, frame_ms_step, prefill_ms, voc_ms = run_leg(eng, voc, prefixes, n_text, pad, F, a.steps, a.warmup, sync_all)
    verified = verify_against_fixture(run_leg.last_codes, B, F, a.seed, world, prefixes, n_text, pad, rank, R)
    # the frame graph alone on the chip (inside a step the previous step's vocoder chunk runs beside it and the two
    # share the machine; the kernel-quality figure is this one)
    eng.start(prefixes, n_text, ignore_eos=True, max_frames=F)
    assert eng.run(F) == F
    frame_ms = eng.last_run_ms / F
    voc_ms_step = voc_ms
    voc_ms_cap_alone = None
    if voc is not None:      # likewise one 32-chunk decode alone: with the co-run grid, then with one workgroup per tile (the kernels' own rate)
        codes_alone, _ = eng.codes()
        lib.voc_set_max_workgroups(voc_cap)
        voc.decode(codes_alone.copy())
        voc_ms_cap_alone = float(voc.ms[-1])
        lib.voc_set_max_workgroups(0)
        voc.decode(codes_alone.copy())
        voc_ms = float(voc.ms[-1])
    step_w_bytes = eng.step_weight_bytes
    dt = R.max_over_ranks(dt)
    value = aggregate_value(world, B, F, a.steps, dt)
    ms_per_step = dt / a.steps * 1e3
    algo_bytes = step_w_bytes + kv_bytes_per_step(n_text, F, cfg)
    achieved = algo_bytes / (frame_ms * 1e-3) / 1e9
    out = {
        "metric": "12Hz codec-tokens/s (codec frames/s, 16 codes each) + RTF, Qwen3-TTS-0.6B",
        "value": round(value, 1), "unit": "codec_frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 weights/KV, f32 accumulate (talker+CP); vocoder exact fp32 (f32 MFMA)", "data": "synthetic (random-init weights of the "
        "0.6B architecture, seeded prefix embeddings, greedy, EOS suppressed)",
        "config": {"workload": f"configs[2]/[3]: batch={B} mixed ru/en prompts per GPU, {F} frames/utterance/step "
                               f"(prefill + hipGraph decode loop), utterance-sharded DP over {world} GPU(s)",
                   "batch_per_gpu": B, "frames_per_step": F, "prompt_tokens": "5-40 (fixed set)"},
        "rtf": round((dt / a.steps) / (F * FRAME_SEC), 5),
        "rtf_aggregate": round((dt / a.steps) / (world * B * F * FRAME_SEC), 6),
        "prefill_ms": round(prefill_ms, 3), "vocoder_ms_per_step": round(voc_ms_step, 3),
        "verified": verified,
        "roofline": None,
        "roofline_step": {"kernel": "frame-step hipGraph (talker 28L + 15 code-predictor passes + heads, 553 nodes)",
                          "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                          "algorithmic_bytes_per_launch": int(algo_bytes), "avg_launch_ms": round(frame_ms, 4),
                          "avg_launch_ms_beside_vocoder": round(frame_ms_step, 4)},
    }
    if voc is not None:
        fl = float(lib.voc_decode_flops(voc.h, B))
        out["roofline_vocoder"] = {"kernel": "conv_kernel (exact-fp32 MFMA implicit-GEMM convs) over the whole decoder table: RVQ, 8-layer "
                                             "pre-transformer, x2 x2 upsamplers + ConvNeXt, BigVGAN trunk (hyper-parameters of the "
                                             "transformer / ConvNeXt stages: recollection)",
                                   "bound": "mfma", "achieved": round(fl / (voc_ms * 1e-3) / 1e12, 2),
                                   "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(fl / (voc_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                   "traffic": None, "flops_per_launch": fl, "avg_launch_ms": round(voc_ms, 3),
                                   "grid": "one workgroup per tile (the decode has the GPU to itself)",
                                   "co_run_grid_workgroups": int(voc_cap),
                                   "avg_launch_ms_co_run_grid_alone": round(voc_ms_cap_alone, 3),
                                   "avg_launch_ms_beside_frame_loop": round(voc_ms_step, 3)}
        # the optional arithmetic: every f32 operand as two fp16 terms, 3 fp16 MFMAs per product, f32 accumulate
        # (executed MFMA flops = 3 x table flops): one decode alone, and two whole steps (rank 0, N = 1)
        if world == 1:
            lib.voc_set_exact_fp32(0)
            lib.voc_set_max_workgroups(0)
            voc.decode(codes_alone.copy())
            sp_ms = float(voc.ms[-1])
            # (the co-run grid here too: 9.93 k against 9.48 k frames/s with one workgroup per tile)
            dt_sp, _, _, _ = run_leg(eng, voc, prefixes, n_text, pad, F, 2, 1, sync_all)
            lib.voc_set_exact_fp32(1)
            vt = Vocoder(lib, make_voc_pack(a.cache, a.seed, rank, barrier, trunk_only=True), B)
            vt.decode(codes_alone.copy())
            vt.decode(codes_alone.copy())
            out["roofline_vocoder"]["trunk_only_avg_launch_ms"] = round(float(vt.ms[-1]), 3)   # without pre-transformer / ConvNeXt
            vt.close()
            out["vocoder_split_f16x2"] = {"avg_launch_ms": round(sp_ms, 3),
                                          "achieved": round(3.0 * fl / (sp_ms * 1e-3) / 1e12, 2),
                                          "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": round(3.0 * fl / (sp_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                                          "fp32_equivalent_tflops": round(fl / (sp_ms * 1e-3) / 1e12, 2),
                                          "value_with_split_vocoder": round(aggregate_value(1, B, F, 2, dt_sp), 1),
                                          "ms_per_step_with_split_vocoder": round(dt_sp / 2 * 1e3, 3)}
        voc.close()
    eng.destroy()
    # the other legs run without a cap (measured with one workgroup per CU: one utterance 452.8 vs 448.0 frames/s, its chunk 11.2
    # instead of 9.3 ms; natural lengths 6.20 k vs 6.91 k; the long-form walk runs after its loop)
    lib.voc_set_max_workgroups(0)
    if world == 1 and not a.no_b1:
        eng1 = FrameEngine(path, max_batch=1, n_ctx=n_ctx, max_frames=F)
        voc1 = Vocoder(lib, make_voc_pack(a.cache, a.seed, rank, barrier), 1) if not a.no_vocoder else None
        dt1, frame_ms1, prefill_ms1, voc_ms1 = run_leg(eng1, voc1, prefixes[:1], n_text[:1], pad, F, a.steps, a.warmup,
                                                       sync_all)
        if voc1 is not None:
            voc1.close()
        ab1 = step_w_bytes + kv_bytes_per_step(n_text[:1], F, cfg)
        v1 = verify_first_utterance(run_leg.last_codes[:, 0, :], a.seed, F)
        if v1.get("checked") and not v1.get("ok"):
            verified = dict(verified, checked=True, ok=False, batch1_failed=True)
        out["batch1"] = {"workload": "configs[1]: batch=1, same engine", "verified": v1, "value": round(F * a.steps / dt1, 1),
                         "unit": "codec_frames/s", "rtf": round((dt1 / a.steps) / (F * FRAME_SEC), 5),
                         "ms_per_frame": round(frame_ms1, 4), "prefill_ms": round(prefill_ms1, 3),
                         "vocoder_ms_per_chunk": round(voc_ms1, 3),
                         "hbm_frac": round(ab1 / (frame_ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        eng1.destroy()
    if rank == 0 and world == 1 and not a.no_longform and not a.no_vocoder:
        out["longform"] = longform_leg(lib, path, make_voc_pack(a.cache, a.seed, rank, barrier), prefixes[0], n_text[0], pad,
                                       a.longform_frames, a.seed)
        if out["longform"]["verified"].get("checked") and not out["longform"]["verified"].get("ok"):
            verified = dict(verified, checked=True, ok=False, longform_failed=True)
    if rank == 0 and world == 1 and not a.no_ragged and not a.no_vocoder:
        out["ragged"] = ragged_leg(lib, path, make_voc_pack(a.cache, a.seed, rank, barrier), B, a.seed)
    if rank == 0 and world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(path, cfg, prefixes[0], n_text[0], pad, a.cpu_frames,
                                           None if a.no_vocoder else make_voc_pack(a.cache, a.seed, rank, barrier))
    if rank == 0:
        # the dominant kernel's in-graph launch duration (child process; this process's engines are gone by now)
        out["roofline"], table = dominant_kernel_roofline(B, a.cache, timeline=not a.no_timeline)
        if table is not None:
            out["frame_timeline"] = {k: {"n": v["n"], "span_us": v["mean_span_us"], "gap_us": v["mean_gap_before_us"]}
                                     for k, v in table["kinds"].items()}
        out["verified"] = verified
        print(json.dumps(out), flush=True)
    R.close()
    if verified.get("checked") and not verified.get("ok"):
        sys.exit(3)       # a fast step whose codes differ from the reference's is not a result


if __name__ == "__main__":
    main()
