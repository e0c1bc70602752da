"""The N>1 path of bench.py on CPU: two processes over gloo exercise the rank bookkeeping, the
utterance sharding (disjoint prompt slices, no data-path collective) and the max-over-ranks timing."""
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    R = bench.Ranks("gloo")
    prefixes, n_text, pad = bench.workload(4, R.rank, 1234)
    R.barrier()
    dt = R.max_over_ranks(1.0 + R.rank)           # rank 1 is the slow one
    R.sync_all()
    q.put((R.rank, R.world, n_text, [p.shape for p in prefixes], dt, float(pad[0])))
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
    (r0, w0, t0, s0, dt0, pad0), (r1, w1, t1, s1, dt1, pad1) = res
    assert (r0, r1, w0, w1) == (0, 1, 2, 2)
    assert dt0 == dt1 == 2.0                       # both ranks see the max
    sys.path.insert(0, ROOT)
    import bench
    assert t0 == bench.PROMPT_TOKENS[0:4] and t1 == bench.PROMPT_TOKENS[4:8]   # disjoint consecutive slices
    assert all(s == (n + 9, 1024) for s, n in zip(s0, t0))
    assert pad0 != pad1                            # per-rank seeds differ
    assert bench.aggregate_value(2, 32, 64, 6, 2.0) == 2 * 32 * 64 * 6 / 2.0
