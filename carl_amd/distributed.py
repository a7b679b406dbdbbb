"""Multi-GPU lane sharding (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no distributed code at all.  Lanes are independent
(carl/envs/carl_env.py:245-342 holds no cross-env state), so the partition is contiguous
ranges of GLOBAL lane ids with no data-path collective: rank r owns lanes
[offset_r, offset_r + n_r).  Philox streams and the initial lane<->context assignment are
functions of the global id, so any split of the same global batch produces bit-identical
transitions (tests/test_gpu_parity.py::test_lane_sharding_is_invariant).  The only
collective is the reporting one: an all-gather of per-lane episodic returns / lengths /
episode counts -- KBs, latency-bound, issued at reporting cadence, never per step.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class LaneShard:
    rank: int
    world_size: int
    total_lanes: int
    offset: int  # global id of this rank's first lane
    count: int   # lanes owned by this rank

    @property
    def slice(self) -> slice:
        return slice(self.offset, self.offset + self.count)


def lane_shard(total_lanes: int, rank: int, world_size: int) -> LaneShard:
    """Contiguous, balanced split: the first ``total % world`` ranks get one extra lane."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(int(total_lanes), int(world_size))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return LaneShard(rank, world_size, int(total_lanes), offset, count)


def shard_context_rows(values_2d, shard: LaneShard, lane_to_context_identity: bool):
    """Rows of the global context table a rank must hold: its own lanes' rows when lane i
    <-> context i (C == N, static selector), the whole table otherwise (C << N,
    replicated; round robin / random selectors can reach any context)."""
    return values_2d[shard.slice] if lane_to_context_identity else values_2d


def default_context_index(n_lanes: int, lane_offset: int, n_contexts: int, round_robin: bool = False, stride: int = 1,
                          context_offset: int | None = None):
    """Context-table row each of a rank's lanes holds before its first reset (``VecEngine.default_ctx_idx``; pure
    host arithmetic, int64 NumPy).  Global lane g starts on global context ``g mod C_global`` (round robin: one
    stride earlier, so that its first reset lands there).  The rank's table may be a SHARD of the global one
    (``shard_context_rows``): its row 0 has the global id ``context_offset``.  ``context_offset=None``: a table with
    exactly one row per lane is this rank's own slice of a lane <-> context identity (row 0 = context
    ``lane_offset``); any other table is the whole, replicated set (row 0 = context 0)."""
    import numpy as np

    off = context_offset
    if off is None:
        off = int(lane_offset) if int(n_contexts) == int(n_lanes) else 0
        if int(lane_offset) > 0 and int(n_contexts) == int(n_lanes):
            # a guess from the shapes: a REPLICATED table that happens to have one row per local lane would be
            # misread as this rank's shard.  The env layer passes context_offset explicitly; direct engine users
            # should too.
            import warnings

            warnings.warn("default_context_index: context_offset not given and the table has one row per local lane -- "
                          f"assuming it is this rank's shard of a lane <-> context identity (row 0 = context {lane_offset}); "
                          "pass context_offset=0 for a replicated table", RuntimeWarning, stacklevel=3)
    g = np.arange(int(n_lanes), dtype=np.int64) + int(lane_offset) - int(off)
    if round_robin:
        g = g - int(stride)
    return np.mod(g, int(n_contexts))


def _all_gather_1d(t: torch.Tensor, counts: list[int] | None = None, padded: bool = False) -> torch.Tensor:
    import torch.distributed as dist

    world = dist.get_world_size()
    if counts is None:
        n = torch.tensor([t.numel()], device=t.device, dtype=torch.int64)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        counts = [int(x.item()) for x in ns]
    if len(set(counts)) == 1 and not padded:
        out = torch.empty(world * counts[0], dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous())
        return out
    m = max(counts)
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def all_gather_episode_stats(engine_or_stats, counts: list[int] | None = None, padded: bool = False) -> dict[str, torch.Tensor]:
    """All ranks receive the global, lane-ordered ``last_return`` / ``last_length`` /
    ``episodes_done`` vectors (rank order == global lane order by construction).

    ``engine_or_stats`` is a ``VecEngine`` or a dict with those three tensors (the CPU
    tests pass plain tensors over gloo).  Equal shard sizes take one ``all_gather_into_tensor`` per vector;
    uneven ones (``counts``) -- or ``padded=True`` -- the padded list form.  Under an initialised process group
    the collective RUNS even with one rank (a world-1 RCCL communicator is how the one-GPU boxes exercise
    librccl); without a group it is the identity."""
    import torch.distributed as dist

    if not isinstance(engine_or_stats, dict):
        e = engine_or_stats
        engine_or_stats = {"last_return": e.last_return, "last_length": e.last_length,
                           "episodes_done": e.episodes_done}
    if not dist.is_available() or not dist.is_initialized():
        return {k: v.clone() for k, v in engine_or_stats.items()}
    return {k: _all_gather_1d(v, counts, padded) for k, v in engine_or_stats.items()}


def reduce_episode_summary(engine_or_stats) -> dict[str, float]:
    """Cheaper reporting form: one all-reduce of (sum return, sum length, n finished lanes)."""
    import torch.distributed as dist

    if not isinstance(engine_or_stats, dict):
        e = engine_or_stats
        engine_or_stats = {"last_return": e.last_return, "last_length": e.last_length,
                           "episodes_done": e.episodes_done}
    fin = engine_or_stats["episodes_done"] > 0
    v = torch.stack([
        (engine_or_stats["last_return"].double() * fin).sum(),
        (engine_or_stats["last_length"].double() * fin).sum(),
        fin.double().sum(),
        engine_or_stats["episodes_done"].double().sum(),
    ])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(v)
    n = max(float(v[2]), 1.0)
    return {"mean_return": float(v[0]) / n, "mean_length": float(v[1]) / n, "lanes_finished": float(v[2]),
            "episodes": float(v[3])}
