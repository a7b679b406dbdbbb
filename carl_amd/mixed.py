"""Mixed-family batches: several families' lanes on ONE device stepped as one object.

BASELINE config 3 ("CARLAcrobot + CARLMountainCar mixed batch, 131 072 contexts") and
config 5 ("CARLBraxHalfcheetah + CARLBraxHumanoid, 65 536 contexts") name batches whose
lanes belong to different env classes.  The reference has no such object -- every env is
its own Python object (carl/envs/carl_env.py:245-342 holds no cross-env state), so the lanes
stay independent and a mixed batch is a *partition* of the global lane range into one
contiguous piece per family.  SURVEY.md 8e: "split each family evenly across GPUs so every
GPU runs the same kernel sequence".

``MixedVecEngine`` owns one ``VecEngine`` / ``BraxVecEngine`` per family and

* enqueues every family's launch of a per-call ``step`` on its own HIP stream, forked from
  and joined back into the caller's stream with events, so the small per-call launches of
  different families overlap and the whole step is ONE stream-ordered operation for the
  caller (it also captures into a hipGraph as one fork/join); the launches of a fused
  ``rollout`` go back to back on the caller's stream -- each fills the chip by itself (two
  families' workgroups need more LDS than a CU has, so they could not co-reside) and the
  fork/join only added ~75 us of gaps per mixed launch (measured, round 2; round 3 tried it for the Brax families,
  whose small one-wavefront workgroups could co-reside: config 5 7.69 ms back to back vs 7.82-8.26 ms concurrent);
* re-homes the parts' episodic-return bookkeeping (``ep_return``, ``last_return``,
  ``last_length``, ``episodes_done``) and per-step ``reward`` / ``terminated`` /
  ``truncated`` into contiguous ``[N_total]`` buffers, so the reporting all-gather
  (``carl_amd.distributed.all_gather_episode_stats``) moves one vector per rank and the
  caller sees one ``[N]`` view; observations keep their per-family width.

Part k owns global lanes ``[lane_offset_k, lane_offset_k + n_k)``; results are bit-identical
to running the parts as separate engines (tests/test_gpu_parity.py).
"""
from __future__ import annotations

from typing import Sequence

import torch

from carl_amd.engine import VecEngine

_SHARED_1D = ("ep_return", "last_return", "last_length", "episodes_done", "reward", "terminated", "truncated", "done")


class MixedVecEngine:
    def __init__(self, parts: Sequence[VecEngine], names: Sequence[str] | None = None):
        if not parts:
            raise ValueError("a mixed batch needs at least one part")
        dev = parts[0].device
        if any(p.device != dev for p in parts):
            raise ValueError("all parts of a mixed batch live on one device")
        self.parts = list(parts)
        self.names = list(names) if names is not None else [f"part{k}" for k in range(len(parts))]
        self.device = dev
        self.sizes = [p.n for p in self.parts]
        self.n = sum(self.sizes)
        self.offsets = [sum(self.sizes[:k]) for k in range(len(self.parts))]
        # one contiguous [N_total] home for everything that is a per-lane scalar
        for name in _SHARED_1D:
            whole = torch.zeros(self.n, dtype=getattr(self.parts[0], name).dtype, device=dev)
            for p, off in zip(self.parts, self.offsets):
                piece = whole[off:off + p.n]
                piece.copy_(getattr(p, name))
                setattr(p, name, piece)
            setattr(self, name, whole)
        for p in self.parts:
            p._sync_pointers()
        # alternating priorities: two streams of one priority can be multiplexed onto ONE hardware queue by the runtime
        # (then their kernels serialise and `rollout(free_running=True)` overlaps nothing -- seen as 2.9e8 instead of
        # 4.2e8 env-steps/s for two Ant half-batches, depending on how many streams the process had made before)
        self._streams = [torch.cuda.Stream(device=dev, priority=(-1 if k % 2 else 0)) for k in range(len(self.parts))]
        self._fork = torch.cuda.Event()
        self._joins = [torch.cuda.Event() for _ in self.parts]
        self._pending = []      # outputs allocated by free-running launches that were not joined yet (see rollout / join)
        self._in_flight = False  # a free-running launch was enqueued since the last join()
        self.pair_launches = 0   # fused rollouts that went out as ONE heterogeneous launch (carl_rollout_pair)
        self._pair_ok = None     # False: this batch can never take the one-launch pair kernel (see _rollout_pair)

    # ------------------------------------------------------------------ fork / join
    def _each(self, fn):
        """Run ``fn(k, part)`` for every part on the part's stream, ordered after what the caller's
        stream holds now and before what it enqueues next."""
        cur = torch.cuda.current_stream(self.device)
        self._fork.record(cur)
        out = []
        for k, (p, s) in enumerate(zip(self.parts, self._streams)):
            s.wait_event(self._fork)
            with torch.cuda.stream(s):
                out.append(fn(k, p))
            self._joins[k].record(s)
        for j in self._joins:
            cur.wait_event(j)
        return out

    def _auto_join(self) -> None:
        """A call that touches engine state on the caller's stream while free-running launches are still in flight
        on the parts' streams would race with them: order the caller's stream after them first."""
        if self._in_flight:
            self.join()

    def part_slice(self, k: int) -> slice:
        return slice(self.offsets[k], self.offsets[k] + self.sizes[k])

    # ------------------------------------------------------------------ API (VecEngine-shaped)
    def seed(self, seed: int) -> None:
        for p in self.parts:
            p.seed(seed)

    def reset(self, mask: torch.Tensor | None = None) -> list[torch.Tensor]:
        if mask is not None and mask.numel() != self.n:
            raise ValueError("mask must have one entry per lane of the mixed batch")
        self._auto_join()
        return self._each(lambda k, p: p.reset(None if mask is None else mask[self.part_slice(k)]))

    def step(self, actions: Sequence):
        """One step of every lane of every family -> (obs per part, reward[N], terminated[N], truncated[N])."""
        if len(actions) != len(self.parts):
            raise ValueError(f"expected {len(self.parts)} action arrays (one per family)")
        self._auto_join()
        res = self._each(lambda k, p: p.step(actions[k]))
        return [r[0] for r in res], self.reward, self.terminated, self.truncated

    def alloc_rollout(self, n_steps: int, final_obs: bool = False) -> list[dict]:
        return [p.alloc_rollout(n_steps, final_obs) for p in self.parts]

    # a Brax part of at most this many envs leaves more than half of the chip's wavefront slots empty (4 096 envs at 16
    # lanes per env = 1 024 wavefronts = one per SIMD): two such launches side by side finish sooner than back to back
    SMALL_BRAX_PART = 8192

    def _small_brax_parts(self) -> bool:
        return len(self.parts) > 1 and all(hasattr(p, "sys") and p.n <= self.SMALL_BRAX_PART for p in self.parts)

    def rollout(self, actions: Sequence, outs: Sequence[dict] | None = None, *, free_running: bool = False,
                overlap: bool | None = None) -> list[dict]:
        """T fused steps of every family; ``actions[k]`` is part k's ``[T, n_k(, A_k)]``.

        ``overlap`` (default: decided per batch): run the parts' launches side by side on the parts' streams, forked
        from and joined back into the caller's stream inside this call -- still ONE stream-ordered operation for the
        caller.  It pays when every part under-fills the chip: the 8-GPU shard of BASELINE config 5 (Halfcheetah x 4 096
        + Humanoid x 4 096 per GPU) is two launches of ~1 000 wavefronts each on 1 024 SIMDs -- latency-bound at one
        wavefront per SIMD -- and takes max(a, b) x 1.2 side by side instead of a + b (measured: 1.60 -> 0.9 ms per
        20-step launch).  Full-size parts each fill the chip by themselves and go back to back (the fork / join costs
        more than it returns there: DESIGN.md appendix A).  ``overlap=False`` forces one launch per part, back to back
        (also for the Acrobot + x pair, which otherwise goes out as ONE heterogeneous launch -- ``_rollout_pair``).

        ``free_running=True``: part k's launch goes on part k's own stream, ordered after the caller's stream NOW, and
        is NOT joined back -- consecutive ``rollout`` calls of different parts then overlap on the device (part A's
        launch i + 1 runs in the tail of part B's launch i).  That is the double-buffered collector (half-batch A steps
        while the policy looks at half-batch B); the caller calls ``join()`` before it reads the outputs on its own
        stream.  Measured: CARLBraxAnt x 32 768 as two half-batches 3.5e8 -> 4.4e8 env-steps/s (the Brax workgroups live
        for a whole launch and leave the last round of SIMD slots half empty; bench.py `free_running_two_streams`)."""
        if len(actions) != len(self.parts):
            raise ValueError(f"expected {len(self.parts)} action arrays (one per family)")
        if free_running:
            cur = torch.cuda.current_stream(self.device)
            self._fork.record(cur)
            res = []
            for k, (p, s) in enumerate(zip(self.parts, self._streams)):
                s.wait_event(self._fork)
                a = actions[k]
                given = None if outs is None else outs[k]
                with torch.cuda.stream(s):
                    out_k = p.rollout(a, given)
                # The caching allocator hands a freed block back to the stream it was ALLOCATED on.  The actions (and
                # caller-provided output buffers) were made on the caller's stream and are used on the part's stream;
                # outputs allocated inside the call (``outs=None``) were made on the part's stream and are read by the
                # caller after ``join()``: each is recorded on the other stream, so dropping the tensor cannot hand its
                # block to a kernel that starts before the other stream's work on it has finished.  A
                # converted copy of the actions (dtype / device / layout) is made inside ``p.rollout`` on the part's
                # stream and consumed there.
                if torch.is_tensor(a) and a.is_cuda:
                    a.record_stream(s)
                if given is None:
                    self._pending.append(out_k)
                else:
                    for t in given.values():
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(s)
                self._in_flight = True
                res.append(out_k)
            return res
        self._auto_join()
        if self._pair_ok is not False and overlap is None:
            res = self._rollout_pair(actions, outs)
            if res is not None:
                return res
        if overlap is None:
            overlap = self._small_brax_parts()
        if overlap:
            return self._each(lambda k, p: p.rollout(actions[k], None if outs is None else outs[k]))
        return [p.rollout(actions[k], None if outs is None else outs[k]) for k, p in enumerate(self.parts)]

    def _rollout_pair(self, actions, outs):
        """Two classic-control families, one of them the float64 Acrobot, in the lean staged configuration: ONE launch
        for both (``carl_rollout_pair``: BASELINE config 3's Acrobot + MountainCar as a heterogeneous launch, the second
        family's wavefronts issuing in the gaps of Acrobot's RK4).  Returns ``None`` when the library declines
        (``CARL_ERR_UNSUPPORTED``: other families, int64 actions, terminal observations, moving selectors ...) -- the
        caller then launches the parts one after the other; results are bit-identical either way."""
        import ctypes as C

        from carl_amd import _lib

        if len(self.parts) != 2 or any(hasattr(p, "sys") for p in self.parts):
            self._pair_ok = False
            return None
        pa, pb = self.parts
        T = int(actions[0].shape[0])
        if int(actions[1].shape[0]) != T:
            return None
        if any(torch.is_tensor(a) and a.dtype in (torch.uint8, torch.float16, torch.bfloat16) for a in actions[:2]):
            return None  # the pair kernel reads int32 / float32 actions; narrow-format parts take their own launches
        # What carl_rollout_pair declines (carl_amd.hip: pair_part_ok) is decided HERE, before any tensor is touched: an
        # eligible family pair in a non-lean configuration (round-robin / random selector, int64 actions, terminal
        # observations, a finished-episode log, rows whose pitch is not a multiple of 16 lanes) used to convert both action
        # tensors and allocate full [T, N, ...] outputs on every call only to hear UNSUPPORTED and do it all again in the
        # per-part path.
        fams = {pa.family, pb.family}
        if fams - set(range(_lib.CARL_N_FAMILIES)) or _lib.ACROBOT not in fams or pa.family == pb.family or \
                any(p.family == _lib.ACROBOT and (p.b.flags & _lib.FLAG_ACROBOT_FP32) for p in self.parts):
            self._pair_ok = False  # never eligible: stop asking
            return None
        for k, p in enumerate(self.parts):
            a = actions[k]
            pitch = p._row_pitch() if outs is None else max(p.n, int(outs[k]["reward"].stride(0)))  # (engine.py: _rollout_io)
            lean = (p.b.selector in (_lib.SEL_STATIC, _lib.SEL_HOST) and not p._takes_direct_kernel(pitch) and p.fin_capacity == 0
                    and not (p.b.flags & _lib.FLAG_ROLLOUT_DIRECT)
                    and not (torch.is_tensor(a) and a.dtype == torch.int64)
                    and not (outs is not None and outs[k].get("final_obs") is not None))
            if not lean:
                return None
        aa, dta = pa._action_tensor(actions[0], (T,))
        ab, dtb = pb._action_tensor(actions[1], (T,))
        if outs is None:
            outs = [pa.alloc_rollout(T), pb.alloc_rollout(T)]
        ioa, iob = pa._rollout_io(aa, dta, outs[0], T), pb._rollout_io(ab, dtb, outs[1], T)
        with torch.cuda.device(self.device):
            code = pa.lib.carl_rollout_pair(C.byref(pa.b), C.byref(ioa), C.byref(pb.b), C.byref(iob), T, pa._stream())
        if code == _lib.ERR_UNSUPPORTED:
            if {pa.family, pb.family} - set(range(_lib.CARL_N_FAMILIES)) or _lib.ACROBOT not in (pa.family, pb.family):
                self._pair_ok = False  # never eligible: stop asking
            return None
        _lib.check(code)
        self.pair_launches += 1
        return list(outs)

    def join(self) -> None:
        """Order the caller's stream after everything the parts' streams hold (after ``rollout(free_running=True)``)."""
        cur = torch.cuda.current_stream(self.device)
        for j, s in zip(self._joins, self._streams):
            j.record(s)
            cur.wait_event(j)
        for out in self._pending:  # the caller reads these on ITS stream from here on
            for t in out.values():
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)
        self._pending.clear()
        self._in_flight = False

    def autotune(self, n_steps: int = 2) -> None:
        for p in self.parts:
            if hasattr(p, "autotune"):
                p.autotune(n_steps=n_steps)

    @property
    def ctx_idx(self) -> list[torch.Tensor]:
        return [p.ctx_idx for p in self.parts]
