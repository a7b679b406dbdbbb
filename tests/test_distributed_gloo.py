"""N > 1 path on CPU: world_size-2 gloo processes exercise the lane partition and the
reporting collective of carl_amd/distributed.py.  Each rank steps ITS lane range with the
CPU oracle (test infrastructure standing in for the HIP engine, which needs a GPU) using
the product's shard plan (global lane ids, sharded context rows), gathers episodic stats
through the product's all-gather, and rank 0 compares with the unsharded batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from carl_amd.distributed import (
    all_gather_episode_stats,
    lane_shard,
    reduce_episode_summary,
    shard_context_rows,
)
from oracle import oracle as O

N, T, SEED = 1003, 45, 17  # odd lane count: unequal shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    rng = np.random.default_rng(0)
    table = np.tile(O.default_row(O.MOUNTAINCAR), (N, 1))
    table[:, 5] = rng.uniform(5e-4, 2e-3, N)
    table[:, 3] = rng.uniform(0.3, 0.55, N)
    acts = rng.integers(0, 3, (T, N)).astype(np.int32)
    return table, acts


def _run(engine, acts):
    engine.reset()
    for t in range(acts.shape[0]):
        engine.step(acts[t])
    return engine


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table, acts = _problem()
        sh = lane_shard(N, rank, world)
        rows = shard_context_rows(table, sh, lane_to_context_identity=True)
        eng = O.Engine(O.MOUNTAINCAR, rows, sh.count, selector=O.SEL_STATIC, seed=SEED, lane_offset=sh.offset,
                       max_steps=20, ctx_idx0=np.arange(sh.count))
        _run(eng, acts[:, sh.slice])
        stats = {"last_return": torch.from_numpy(eng.last_return.copy()),
                 "last_length": torch.from_numpy(eng.last_length.copy()),
                 "episodes_done": torch.from_numpy(eng.episodes_done.copy())}
        g = all_gather_episode_stats(stats)
        summ = reduce_episode_summary(stats)
        state = torch.from_numpy(eng.state.copy())
        if rank == 0:
            q.put(({k: v.numpy() for k, v in g.items()}, summ, state.numpy(), (sh.offset, sh.count)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_lane_shard_plan():
    for total, world in ((10, 3), (65536, 8), (7, 8), (1003, 2)):
        parts = [lane_shard(total, r, world) for r in range(world)]
        assert parts[0].offset == 0 and sum(p.count for p in parts) == total
        for a, b in zip(parts, parts[1:]):
            assert a.offset + a.count == b.offset and a.count - b.count in (0, 1)
    with pytest.raises(ValueError):
        lane_shard(10, 3, 3)
    t = np.arange(20).reshape(10, 2)
    assert shard_context_rows(t, lane_shard(10, 1, 2), True).tolist() == t[5:].tolist()
    assert shard_context_rows(t, lane_shard(10, 1, 2), False) is t


def test_world2_gloo_sharded_equals_unsharded():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, summ, state0, (off, cnt) = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table, acts = _problem()
    full = _run(O.Engine(O.MOUNTAINCAR, table, N, selector=O.SEL_STATIC, seed=SEED, max_steps=20,
                         ctx_idx0=np.arange(N)), acts)
    np.testing.assert_array_equal(gathered["last_return"], full.last_return)
    np.testing.assert_array_equal(gathered["last_length"], full.last_length)
    np.testing.assert_array_equal(gathered["episodes_done"], full.episodes_done)
    np.testing.assert_array_equal(state0, full.state[off:off + cnt])  # bit-identical lanes
    fin = full.episodes_done > 0
    assert summ["lanes_finished"] == fin.sum() and summ["episodes"] == full.episodes_done.sum()
    assert summ["mean_return"] == pytest.approx(float(full.last_return[fin].mean()), rel=1e-6)


def test_default_lane_context_assignment_is_shard_aware():
    """10 lanes over 3 ranks (4 / 3 / 3).  With the context table sharded like the lanes (one row per
    lane) every rank's lanes start on their OWN rows 0..count-1; with the table replicated they start on
    ``global lane mod C``; an explicit context_offset describes any other slice.  (Round 1 took the modulo of the
    global id by the LOCAL row count: rank 1 -- offset 4, 3 rows -- read rows (4 + i) mod 3 = 1, 2, 0.)"""
    import numpy as np

    from carl_amd.distributed import default_context_index, lane_shard

    N = 10
    seen = []
    for r in range(3):
        sh = lane_shard(N, r, 3)
        idx = default_context_index(sh.count, sh.offset, sh.count)            # sharded table: identity
        np.testing.assert_array_equal(idx, np.arange(sh.count))
        rep = default_context_index(sh.count, sh.offset, 7)                   # replicated table of 7 contexts
        np.testing.assert_array_equal(rep, (np.arange(sh.count) + sh.offset) % 7)
        seen += list(sh.offset + idx)
        rr = default_context_index(sh.count, sh.offset, 7, round_robin=True, stride=2)
        np.testing.assert_array_equal((rr + 2) % 7, rep)                      # first reset lands on g mod C
    assert seen == list(range(N))                                             # every global context exactly once
    # an explicit offset: the rank holds rows [8, 16) of the global set and lanes [10, 14)
    np.testing.assert_array_equal(default_context_index(4, 10, 8, context_offset=8), [2, 3, 4, 5])


def _world1_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        stats = {"last_return": torch.arange(10, dtype=torch.float32), "last_length": torch.arange(10, dtype=torch.int32),
                 "episodes_done": torch.tensor([0, 1, 2, 0, 1, 0, 0, 3, 1, 1], dtype=torch.int32)}
        eq = all_gather_episode_stats(stats)                               # all_gather_into_tensor, one rank
        pad = all_gather_episode_stats(stats, counts=[10], padded=True)    # the padded list form
        summ = reduce_episode_summary(stats)                               # all_reduce, one rank
        q.put({"eq": {k: v.numpy() for k, v in eq.items()}, "pad": {k: v.numpy() for k, v in pad.items()}, "summ": summ})
    finally:
        dist.destroy_process_group()


def test_collectives_run_under_a_one_rank_group():
    """Round 3: the reporting collectives no longer short-circuit when world_size == 1 -- under an initialised group
    they RUN (that is how the one-GPU boxes exercise librccl: tests/test_gpu_mixed_and_multiproc.py does this with
    backend nccl on device tensors).  Here: gloo, one rank, both all-gather forms and the all-reduce are the identity."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_world1_worker, args=(0, 1, _free_port(), q))
    p.start()
    got = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    want = {"last_return": np.arange(10, dtype=np.float32), "last_length": np.arange(10, dtype=np.int32),
            "episodes_done": np.array([0, 1, 2, 0, 1, 0, 0, 3, 1, 1], dtype=np.int32)}
    for form in ("eq", "pad"):
        for k, v in want.items():
            np.testing.assert_array_equal(got[form][k], v)
    fin = want["episodes_done"] > 0
    assert got["summ"]["mean_return"] == pytest.approx(float(want["last_return"][fin].mean()))
    assert got["summ"]["episodes"] == float(want["episodes_done"].sum())
