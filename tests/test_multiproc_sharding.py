"""The N>1 path of bench.py on CPU (gloo, world size 2): rank bookkeeping, the length-sorted round-robin dealing of
utterances (no data-path collective), the max-over-ranks timing, and `python bench.py --gpus 2` starting its two
ranks itself."""
import json
import os
import subprocess
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    R = bench.Ranks("gloo")
    prefixes, n_text, pad = bench.workload(4, R.rank, 1234, R.world)
    R.barrier()
    dt = R.max_over_ranks(1.0 + R.rank)           # rank 1 is the slow one
    R.sync_all()
    q.put((R.rank, R.world, n_text, [p.shape for p in prefixes], dt, float(prefixes[0][0, 0])))
    R.close()


def test_two_ranks_shard_prompts_and_reduce_time():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, t0, s0, dt0, x0), (r1, w1, t1, s1, dt1, x1) = res
    assert (r0, r1, w0, w1) == (0, 1, 2, 2)
    assert dt0 == dt1 == 2.0                       # both ranks see the max
    sys.path.insert(0, ROOT)
    import bench
    # 8 utterances (prompts 0..7), sorted by 3 * n_text descending, dealt 0,1,0,1,...
    order = sorted(range(8), key=lambda i: (-bench.PROMPT_TOKENS[i], i))
    assert t0 == [bench.PROMPT_TOKENS[i] for i in order[0::2]]
    assert t1 == [bench.PROMPT_TOKENS[i] for i in order[1::2]]
    assert sorted(t0 + t1) == sorted(bench.PROMPT_TOKENS[:8])          # a partition: nothing twice, nothing lost
    assert all(s == (n + 9, 1024) for s, n in zip(s0, t0))
    assert x0 != x1                                # utterances carry their own seeds
    assert bench.aggregate_value(2, 32, 64, 6, 2.0) == 2 * 32 * 64 * 6 / 2.0


def test_dealing_balances_config4():
    """BASELINE config 4: 256 utterances over 8 ranks -> 32 each, identical length mix on every rank."""
    sys.path.insert(0, ROOT)
    import bench
    n_all = [bench.PROMPT_TOKENS[i % 32] for i in range(256)]
    d = bench.deal(n_all, 8)
    assert sorted(i for r in d for i in r) == list(range(256))
    assert all(len(r) == 32 for r in d)
    loads = [sum(n_all[i] for i in r) for r in d]
    assert max(loads) == min(loads)
    ragged = bench.deal([5, 40, 7, 33, 21, 9, 30], 3)                   # counts differ by at most one
    assert sorted(len(r) for r in ragged) == [2, 2, 3]


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher environment must run TWO ranks (round 1 ran one and printed
    n_gpus 1).  CPU: gloo backend and the stand-in engine; the JSON line must say n_gpus 2 and count both ranks'
    frames."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--batch", "4", "--frames", "8", "--stub-engine", "--backend", "gloo"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                                  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["data"].startswith("stub")
    assert abs(j["value"] - 2 * 4 * 8 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-3
    # a mismatch between --gpus and the launched world is an error, not a silent 1-GPU run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-engine", "--backend", "gloo"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0


def test_every_rank_runs_the_fixture_prompts_whatever_the_rank_count():
    """The result check at N ranks: the job is the 32-prompt set N times (BASELINE configs[3]), utterance i = prompt i % 32
    with that prompt's inputs, so every slot of every rank has a fixture row and each rank holds every prompt once."""
    sys.path.insert(0, ROOT)
    import bench
    row_of = {i: b for b, i in enumerate(bench.workload_indices(32, 0, 1))}
    assert sorted(row_of) == list(range(32)) and bench.fixture_slots(32, 0, 1, 32) == {s: s for s in range(32)}
    p1, n1, pad1 = bench.workload(32, 0, 1234, 1)
    for world in (2, 4, 8):
        for rank in (0, world - 1):
            mine = bench.workload_indices(32, rank, world)
            slots = bench.fixture_slots(32, rank, world, 32)
            assert sorted(slots) == list(range(32)) and sorted(slots.values()) == list(range(32))
            assert all(row == row_of[mine[slot] % 32] for slot, row in slots.items())
            pw, nw, padw = bench.workload(32, rank, 1234, world)
            assert (pad1 == padw).all()
            for slot, row in slots.items():
                assert nw[slot] == n1[row] and (pw[slot] == p1[row]).all()


def _verify_worker(rank, world, port, q, spoil):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    R = bench.Ranks("gloo")
    g = np.load(os.path.join(ROOT, "tests", "golden", "bench_b32_f64.npz"))
    ids = g["ids"].astype(np.int32)
    prefixes, n_text, pad = bench.workload(32, rank, 1234, world)
    codes = np.full((64, 32, 16), 5, np.int32)                    # slots the fixture does not cover: anything
    for slot, row in bench.fixture_slots(32, rank, world, 32).items():
        codes[:, slot, :] = ids[row]
    if spoil and rank == 1:                                       # one decision of one graded utterance, where the oracle's gap is wide
        slot, row = sorted(bench.fixture_slots(32, rank, world, 32).items())[0]
        m = g["margins"].astype(np.float32)[row]
        f, k = np.unravel_index(int(np.argmax(m)), m.shape)
        codes[f, slot, k] = (codes[f, slot, k] + 1) % 2048
    q.put((rank, bench.verify_against_fixture(codes, 32, 64, 1234, world, prefixes, n_text, pad, rank, R)))
    R.close()


def test_result_check_is_summed_over_two_ranks():
    """Both ranks report the same verdict: 64 utterances graded in all (the prompt set on either rank), all identical; one
    wrong id at a decision that is no near-tie on rank 1 fails the check on every rank."""
    ctx = mp.get_context("spawn")
    for spoil in (False, True):
        q = ctx.Queue()
        port = 31500 + os.getpid() % 2000 + int(spoil)
        procs = [ctx.Process(target=_verify_worker, args=(r, 2, port, q, spoil)) for r in range(2)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=180) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert res[0] == res[1]
        v = json.loads(json.dumps(res[0]))               # the line bench.py prints must serialise
        assert v["checked"] and v["utterances"] == 64 and v["graded_over_ranks"] == 2
        if spoil:
            assert not v["ok"] and v["utterances_identical_over_all_frames"] == 63
        else:
            assert v["ok"] and v["utterances_identical_over_all_frames"] == 64
            assert v["identical_leading_frames"] == {"min": 64, "total": 4096, "of": 4096}


def test_result_check_at_one_rank_serialises():
    """The single-GPU form of the check, on the fixture's own ids: 32 utterances identical, and the dict is plain JSON."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    g = np.load(os.path.join(ROOT, "tests", "golden", "bench_b32_f64.npz"))
    codes = np.ascontiguousarray(g["ids"].astype(np.int32).transpose(1, 0, 2))      # [frames][slots][16]
    prefixes, n_text, pad = bench.workload(32, 0, 1234, 1)
    v = json.loads(json.dumps(bench.verify_against_fixture(codes, 32, 64, 1234, 1, prefixes, n_text, pad)))
    assert v["checked"] is True and v["ok"] is True and v["utterances"] == 32
    assert v["identical_leading_frames"] == {"min": 64, "total": 2048, "of": 2048, "median": 64}
    # other inputs than the fixture's: reported as unchecked, never as a pass
    v = bench.verify_against_fixture(codes, 32, 64, 1235, 1, *bench.workload(32, 0, 1235, 1))
    assert v["checked"] is False and "ok" not in v
