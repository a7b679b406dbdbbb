"""Device-resident batch of env lanes: buffers (PyTorch-ROCm tensors) + C-ABI calls.

``VecEngine`` is the object the CARL-shaped envs wrap -- the replacement for the
gymnasium env (+ TimeLimit) that ``CARLGymnasiumEnv`` builds with ``gymnasium.make``
(reference: carl/envs/gymnasium/carl_gymnasium_env.py:63-64) -- for N lanes at once.
PyTorch only provides device memory and streams here; all arithmetic runs in the HIP
library behind include/carl_amd.h.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from carl_amd import _lib

_FAMILY_BY_NAME = {
    "cartpole": _lib.CARTPOLE, "pendulum": _lib.PENDULUM, "acrobot": _lib.ACROBOT,
    "mountaincar": _lib.MOUNTAINCAR, "mountaincar_cont": _lib.MOUNTAINCAR_CONT,
}


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


# raw handles straight from torch's C binding (the public wrappers build Stream / device objects: ~1.5 us per call,
# a fifth of an eager step at 65 536 lanes)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (
    lambda idx: torch.cuda.current_stream(idx).cuda_stream)
_current_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


class CapturedStep:
    """``VecEngine.capture_step``: a hipGraph of ``n_steps`` ``carl_step`` launches on a fixed action buffer."""

    def __init__(self, eng: "VecEngine", action_buffer: torch.Tensor, n_steps: int = 1):
        a, dt = eng._action_tensor(action_buffer, ())
        if a.data_ptr() != action_buffer.data_ptr():
            raise ValueError("capture_step needs the action buffer in its final form (device, dtype, contiguous): "
                             "the graph reads this exact address on every replay")
        self.eng, self.action, self.n_steps = eng, action_buffer, int(n_steps)
        self.graph = None
        if self.n_steps == 1:
            # ONE step per replay: a graph launch costs ~11 us where the prepared eager call costs ~4 (measured, 65 536
            # CartPole lanes) -- the replayable object then IS the eager fast path on the fixed buffer (same results, same
            # interface), and a capture only happens where it pays (n_steps >= 2).
            return
        dev = eng.device
        # the first launch of a kernel loads its code object, which is not capturable: run one real step on
        # the capture stream first, on a snapshot that is put back afterwards
        snap = eng.snapshot()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            eng.step(action_buffer)
            eng.restore(snap)
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                for _ in range(self.n_steps):
                    eng.step(action_buffer)
        torch.cuda.current_stream(dev).wait_stream(side)

    def replay(self):
        if self.graph is None:
            return self.eng.step(self.action)
        self.graph.replay()
        e = self.eng
        return e.obs, e.reward, e.terminated, e.truncated


class VecEngine:
    """N lanes of one env family on one device.

    Parameters
    ----------
    family : int | str
        ``carl_family_t`` (or its lower-case name).
    ctx_table : array [C, F]
        Context set in the family's feature order (defaults already filled).
    n_lanes : int
    device : torch.device | str
    selector : int
        ``carl_selector_t`` rule applied on every lane reset.
    ctx_idx0 : array [n_lanes] | None
        Context id each lane holds BEFORE its first reset.  Default: round robin
        ``(g - stride) mod C`` (so the first reset lands on ``g mod C``; with one lane
        this is the reference's 0, 1, 2, ...), otherwise ``g mod C`` (g = global lane id).
    lane_offset : int
        Global id of lane 0: lanes sharded over GPUs keep their global ids, so RNG
        streams and initial context assignment do not depend on the GPU count.
    """

    def __init__(self, family, ctx_table, n_lanes: int, device="cuda", *, selector: int = _lib.SEL_ROUND_ROBIN,
                 selector_stride: int = 1, auto_reset: bool = True, max_episode_steps: int | None = None,
                 seed: int = 0, lane_offset: int = 0, cartpole_recompute: bool = False,
                 ctx_obs_rows: Sequence[int] | None = None, ctx_idx0=None, fin_capacity: int = 0,
                 acrobot_fp32: bool = False, context_offset: int | None = None):
        self.lib = _lib.load()
        if isinstance(family, str):
            family = _FAMILY_BY_NAME[family]
        self.family = int(family)
        self.info = self._family_info()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.CarlHipError(
                f"VecEngine needs a ROCm device (got {self.device}); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n = int(n_lanes)
        S, D, F = self.info.state_dim, self.info.obs_dim, self.info.n_features
        self.S, self.D, self.F = S, D, F
        self._action_dim = int(self.info.action_dim)
        dev = self.device
        i32, f32 = torch.int32, torch.float32

        self.state = torch.zeros((S, self.n), dtype=f32, device=dev)
        self.elapsed = torch.zeros(self.n, dtype=i32, device=dev)
        self.episode = torch.zeros(self.n, dtype=i32, device=dev)  # uint32 bits
        self.n_calls = torch.zeros(self.n, dtype=i32, device=dev)
        self.ep_return = torch.zeros(self.n, dtype=f32, device=dev)
        self.last_return = torch.zeros(self.n, dtype=f32, device=dev)
        self.last_length = torch.zeros(self.n, dtype=i32, device=dev)
        self.episodes_done = torch.zeros(self.n, dtype=i32, device=dev)
        # step outputs (reused every call: returned tensors alias these buffers).  ONE allocation, the five outputs are
        # views of it: a scalar caller (num_envs = 1, the reference's return types) then reads a whole transition back with
        # one device-to-host copy instead of four (`read_transition`; per_call.dropin of the bench line: 84 -> ~40 us)
        def up(x):  # every view starts on a 256-byte boundary, like an allocation of its own
            return (x + 255) & ~255

        o_rew = up(4 * self.n * D)
        o_term = up(o_rew + 4 * self.n)
        o_trunc = up(o_term + self.n)
        o_done = up(o_trunc + self.n)
        self._out_offsets = (o_rew, o_term, o_trunc, o_done)
        self._out_slab = torch.zeros(up(o_done + self.n), dtype=torch.uint8, device=dev)
        self.obs = self._out_slab[: 4 * self.n * D].view(f32).view(self.n, D)
        self.reward = self._out_slab[o_rew: o_rew + 4 * self.n].view(f32)
        self.terminated = self._out_slab[o_term: o_term + self.n]
        self.truncated = self._out_slab[o_trunc: o_trunc + self.n]
        self.done = self._out_slab[o_done: o_done + self.n]  # terminated | truncated of the last step()
        self._out_host = None  # pinned mirror of the slab, made on first use (read_transition)
        self.final_obs = torch.zeros((self.n, D), dtype=f32, device=dev)
        # done-mask compaction buffers
        self.done_idx = torch.zeros(max(self.n, 1), dtype=i32, device=dev)
        self.done_count = torch.zeros(1, dtype=i32, device=dev)
        self._scratch = torch.zeros(int(self.lib.carl_done_compact_scratch_elems(self.n)), dtype=i32, device=dev)
        # finished-episode log
        self.fin_capacity = int(fin_capacity)
        if self.fin_capacity > 0:
            self.fin_count = torch.zeros(1, dtype=i32, device=dev)
            self.fin_lane = torch.zeros(self.fin_capacity, dtype=torch.int64, device=dev)
            self.fin_return = torch.zeros(self.fin_capacity, dtype=f32, device=dev)
            self.fin_length = torch.zeros(self.fin_capacity, dtype=i32, device=dev)
        else:
            self.fin_count = self.fin_lane = self.fin_return = self.fin_length = None

        self.b = _lib.Batch()
        self.b.family = self.family
        self.b.n_lanes = self.n
        self.b.max_episode_steps = self.info.max_episode_steps if max_episode_steps is None else int(max_episode_steps)
        self.b.selector = int(selector)
        self.b.selector_stride = int(selector_stride)
        self.b.flags = (_lib.FLAG_AUTORESET if auto_reset else 0) | (
            _lib.FLAG_CARTPOLE_RECOMPUTE if cartpole_recompute else 0) | (
            _lib.FLAG_ACROBOT_FP32 if acrobot_fp32 else 0)
        self.b.lane_offset = int(lane_offset)
        self.b.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        # global id of context-table row 0 (see default_ctx_idx); None: decided per table
        self.context_offset = None if context_offset is None else int(context_offset)

        self.ctx_obs_rows = list(range(F)) if ctx_obs_rows is None else [int(r) for r in ctx_obs_rows]
        if len(self.ctx_obs_rows) > _lib.CARL_MAX_CTX_OBS:
            raise ValueError(f"at most {_lib.CARL_MAX_CTX_OBS} observed context features")
        self.ctx_obs = torch.zeros((len(self.ctx_obs_rows), self.n), dtype=f32, device=dev)
        self.ctx_table = None
        self.ctx_idx = None
        self.set_contexts(ctx_table, ctx_idx0)
        self._io = _lib.StepIO()
        self._sync_pointers()
        # per-call fast path (see step()): everything the ctypes call needs, prepared once
        self._dev_index = int(self.device.index)
        self._b_ref, self._io_ref = C.byref(self.b), C.byref(self._io)
        self._step_fn = self.lib.carl_step
        self._last_action = None
        self._last_action_meta = None
        self._warned_direct = False

    # ------------------------------------------------------------------ contexts
    def default_ctx_idx(self, n_contexts: int) -> torch.Tensor:
        """Context row each lane holds before its first reset: global lane g starts on global context
        ``g mod C_global`` (round robin: one stride earlier, so that the first reset lands there).  The LOCAL
        table may be a shard of the global one (multi-GPU: ``distributed.shard_context_rows``); row 0 of it has
        the global id ``context_offset``.  ``context_offset=None`` (default): a table with exactly one row per
        lane is this rank's own slice of a lane <-> context identity (row 0 = context ``lane_offset``), any
        other table is the whole, replicated context set (row 0 = context 0).  With the modulo taken on the
        global id instead -- as round 1 did -- an uneven split (10 lanes over 3 ranks: offset 4, count 3) read
        row (4 + i) mod 3 and handed lanes their neighbours' contexts."""
        from carl_amd.distributed import default_context_index

        idx = default_context_index(self.n, int(self.b.lane_offset), n_contexts,
                                    self.b.selector == _lib.SEL_ROUND_ROBIN, int(self.b.selector_stride),
                                    self.context_offset)
        return torch.as_tensor(idx).to(torch.int32)

    def set_contexts(self, ctx_table, ctx_idx0=None) -> None:
        """Upload a context set ([C, F], reference feature order) and (re)assign lanes.

        Batched counterpart of the ``contexts`` setter (carl_env.py:122-137) plus
        ``_update_context`` (carl_gymnasium_env.py:75-77)."""
        t = torch.as_tensor(np.asarray(ctx_table, dtype=np.float64) if not torch.is_tensor(ctx_table) else ctx_table)
        if t.ndim != 2 or t.shape[1] != self.F:
            raise ValueError(f"ctx_table must be [C, {self.F}], got {tuple(t.shape)}")
        C_ = int(t.shape[0])
        if C_ < 1:
            raise ValueError("empty context set")
        # feature-major [F, C] fp32 on device
        self.ctx_table = t.to(torch.float32).t().contiguous().to(self.device)
        self.b.n_contexts = C_
        self.b.ctx_stride = C_
        if ctx_idx0 is None:
            idx = self.default_ctx_idx(C_)
        else:
            idx = torch.as_tensor(ctx_idx0).to(torch.int32).reshape(self.n)
            if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= C_):
                raise ValueError("ctx_idx0 out of range")
        self.ctx_idx = idx.to(self.device).contiguous()
        if hasattr(self, "_io"):
            self._sync_pointers()

    def set_contexts_device(self, table_fc, ctx_idx0=None) -> None:
        """Adopt a context table that already lives on this device as feature-major ``[F][C]``
        float32 (``carl_sample_contexts`` output, carl_amd/context/device_sampler.py): no copy."""
        t = table_fc
        if not torch.is_tensor(t) or t.ndim != 2 or t.shape[0] != self.F:
            raise ValueError(f"device context table must be a [{self.F}, C] tensor")
        if t.dtype != torch.float32 or t.device != self.device or not t.is_contiguous():
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
        C_ = int(t.shape[1])
        if C_ < 1:
            raise ValueError("empty context set")
        self.ctx_table = t
        self.b.n_contexts = C_
        self.b.ctx_stride = C_
        if ctx_idx0 is None:
            idx = self.default_ctx_idx(C_)
        else:
            idx = torch.as_tensor(ctx_idx0).to(torch.int32).reshape(self.n)
        self.ctx_idx = idx.to(self.device).contiguous()
        if hasattr(self, "_io"):
            self._sync_pointers()

    def set_ctx_idx(self, idx) -> None:
        """Host-driven context switch (``context_id`` setter, carl_env.py:139-157)."""
        idx = torch.as_tensor(idx).to(torch.int32).reshape(self.n)
        self.ctx_idx.copy_(idx.to(self.device))

    def refresh_ctx_obs(self) -> None:
        """ctx_obs[k, lane] = table[row_k, ctx_idx[lane]] for all lanes (the kernels only
        touch lanes they reset)."""
        if self.ctx_obs.shape[0]:
            rows = torch.as_tensor(self.ctx_obs_rows, device=self.device)
            self.ctx_obs.copy_(self.ctx_table[rows][:, self.ctx_idx.long()])

    # ------------------------------------------------------------------ family hooks
    def _family_info(self):
        return _lib.family_info(self.family)

    def _c_reset(self, mask_ptr) -> int:
        return self.lib.carl_reset(C.byref(self.b), mask_ptr, _ptr(self.obs), self._stream())

    def _c_step(self, io) -> int:
        return self.lib.carl_step(C.byref(self.b), C.byref(io), self._stream())

    def _c_step_fast(self, stream: int) -> int:  # self._io, prepared references
        return self._step_fn(self._b_ref, self._io_ref, stream)

    def _c_rollout(self, io, n_steps: int) -> int:
        return self.lib.carl_rollout(C.byref(self.b), C.byref(io), n_steps, self._stream())

    # ------------------------------------------------------------------ plumbing
    def _sync_pointers(self) -> None:
        b = self.b
        b.state, b.elapsed, b.ctx_idx = _ptr(self.state), _ptr(self.elapsed), _ptr(self.ctx_idx)
        b.episode, b.n_calls, b.ep_return = _ptr(self.episode), _ptr(self.n_calls), _ptr(self.ep_return)
        b.ctx_table = _ptr(self.ctx_table)
        b.n_ctx_obs = len(self.ctx_obs_rows)
        b.ctx_obs = _ptr(self.ctx_obs) if self.ctx_obs_rows else None
        for k, r in enumerate(self.ctx_obs_rows):
            b.ctx_obs_feat[k] = r
        b.last_return, b.last_length = _ptr(self.last_return), _ptr(self.last_length)
        b.episodes_done = _ptr(self.episodes_done)
        b.fin_capacity = self.fin_capacity
        b.fin_count, b.fin_lane = _ptr(self.fin_count), _ptr(self.fin_lane)
        b.fin_return, b.fin_length = _ptr(self.fin_return), _ptr(self.fin_length)
        b.goal_pos = _ptr(getattr(self, "goal_pos", None))
        b.success = _ptr(getattr(self, "success", None))
        b.first_state = _ptr(getattr(self, "first_state", None))
        io = getattr(self, "_io", None)
        if io is not None:  # the per-call outputs never move: step() only fills in the action
            io.obs, io.reward = _ptr(self.obs), _ptr(self.reward)
            io.terminated, io.truncated = _ptr(self.terminated), _ptr(self.truncated)
            io.final_obs = _ptr(self.final_obs)
            io.done = _ptr(self.done)
            io.branch_sig = _ptr(getattr(self, "branch_sig", None))

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    @property
    def auto_reset(self) -> bool:
        return bool(self.b.flags & _lib.FLAG_AUTORESET)

    @auto_reset.setter
    def auto_reset(self, on: bool) -> None:
        self.b.flags = (self.b.flags | _lib.FLAG_AUTORESET) if on else (self.b.flags & ~_lib.FLAG_AUTORESET)

    @property
    def n_contexts(self) -> int:
        return int(self.b.n_contexts)

    # the narrow rollout-only action formats (uint8 / float16 / bfloat16: include/carl_amd.h) exist in the classic-control
    # kernels' loader wave; the Brax entry points take float32 only -- BraxVecEngine turns this off, so that half-precision
    # actions (a policy under autocast) are widened once instead of being rejected by carl_brax_rollout
    _narrow_actions = True

    def _action_tensor(self, action, lead: tuple[int, ...], allow_narrow: bool = False) -> tuple[torch.Tensor, int]:
        allow_narrow = allow_narrow and self._narrow_actions
        a = action if torch.is_tensor(action) else torch.as_tensor(np.asarray(action))
        if self.info.action_is_discrete:
            if allow_narrow and a.dtype == torch.uint8:  # rollout-only input format (include/carl_amd.h: CARL_ACTION_U8)
                dt = _lib.ACTION_U8
                if a.is_contiguous() and a.data_ptr() % 4:  # (a view into a larger buffer: the kernel reads dwords)
                    a = a.clone()
            else:
                if a.dtype not in (torch.int32, torch.int64):
                    a = a.to(torch.int64)
                dt = _lib.ACTION_I32 if a.dtype == torch.int32 else _lib.ACTION_I64
        elif allow_narrow and a.dtype in (torch.float16, torch.bfloat16):  # (likewise: CARL_ACTION_F16 / BF16)
            dt = _lib.ACTION_F16 if a.dtype == torch.float16 else _lib.ACTION_BF16
            if a.is_contiguous() and a.data_ptr() % 8:
                a = a.clone()
        else:
            if a.dtype != torch.float32:
                a = a.to(torch.float32)
            dt = _lib.ACTION_F32
        if a.device != self.device:
            a = a.to(self.device)
        n_expected = self.n * self._action_dim
        for d in lead:
            n_expected *= d
        if a.numel() != n_expected:
            raise ValueError(f"action has {a.numel()} elements, expected {n_expected} ({lead} x {self.n} lanes)")
        return a.contiguous(), dt

    # ------------------------------------------------------------------ API
    def seed(self, seed: int) -> None:
        """``reset(seed=s)`` semantics: new RNG key, per-lane episode counters back to 0."""
        self.b.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.episode.zero_()

    def reset(self, mask: torch.Tensor | None = None) -> torch.Tensor:
        """Reset all lanes (or those with ``mask != 0``); returns the obs buffer."""
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if m.numel() != self.n:
                raise ValueError("mask must have one entry per lane")
        with torch.cuda.device(self.device):
            _lib.check(self._c_reset(_ptr(m)))
        return self.obs

    def done_compact(self) -> tuple[torch.Tensor, torch.Tensor]:
        """Ascending ids of lanes whose last step ended their episode -> (idx[n], count[1])."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.carl_done_compact(
                _ptr(self.terminated), _ptr(self.truncated), self.n, _ptr(self.done_idx),
                _ptr(self.done_count), _ptr(self._scratch), self._stream()))
        return self.done_idx, self.done_count

    def reset_indexed(self, idx: torch.Tensor, count: torch.Tensor) -> torch.Tensor:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.carl_reset_indexed(C.byref(self.b), _ptr(idx), _ptr(count), _ptr(self.obs),
                                                   self._stream()))
        return self.obs

    def reset_done(self) -> torch.Tensor:
        """Explicit-reset path: compact the done mask, reset exactly those lanes."""
        idx, count = self.done_compact()
        return self.reset_indexed(idx, count)

    def step(self, action):
        """One step of every lane -> (obs[N,D], reward[N], terminated[N] u8, truncated[N] u8).

        This is the per-call path (one launch per env step) a policy-in-the-loop caller sits on, so the host
        side is pared down to what a launch needs: the output pointers of ``self._io`` are set once
        (``_sync_pointers``); an action tensor that is the SAME object at the same address as in the previous
        call (the usual case: the policy writes into one buffer) skips dtype / shape / device validation; the
        raw stream handle comes from torch's C binding; the ctypes function and its two by-reference arguments
        are prepared once; the device guard is only entered when another device is current."""
        io = self._io
        if (action is self._last_action and action.data_ptr() == io.action
                and (action.dtype, action.numel()) == self._last_action_meta):
            pass  # (dtype and element count are re-checked: resize_ / set_ / .data = keep the object and its address)
        else:
            a, dt = self._action_tensor(action, ())
            io.action, io.action_dtype = a.data_ptr(), dt
            # only a tensor used as-is can take the fast path next time (a converted copy is a temporary)
            self._last_action = action if a is action else None
            self._last_action_meta = (a.dtype, a.numel())
        if _current_device() == self._dev_index:
            code = self._c_step_fast(_raw_stream(self._dev_index))
        else:
            with torch.cuda.device(self.device):
                code = self._c_step(io)
        if code != 0:
            _lib.check(code)
        return self.obs, self.reward, self.terminated, self.truncated

    def stage_scalar_action(self, action):
        """A host scalar / small array as THE device action tensor of this engine (pinned staging buffer, asynchronous
        copy on the launch stream): what a scalar caller passes to ``step`` -- the same tensor object every call, so
        ``step`` takes its validated-once fast path, and no blocking host-to-device copy per call."""
        if getattr(self, "_a_host", None) is None:
            dt = torch.int32 if self.info.action_is_discrete else torch.float32
            self._a_host = torch.empty(self.n * self._action_dim, dtype=dt).pin_memory()
            self._a_dev = torch.empty((self.n, self._action_dim) if self._action_dim > 1 or not self.info.action_is_discrete
                                      else (self.n,), dtype=dt, device=self.device)
            self._a_host_np = self._a_host.numpy()
            self._a_pending = False
        a = np.asarray(action).reshape(-1)
        if a.size != self._a_host_np.size:  # (NumPy would broadcast one value over every actuator silently)
            raise ValueError(f"action has {a.size} elements, this engine takes {self.n} x {self._action_dim}")
        if self._a_pending:
            # the previous asynchronous upload may still be reading the pinned buffer: wait for the stream before rewriting
            # it.  Never taken by the scalar step path -- ``read_transition`` synchronises the stream after every step and
            # clears the mark -- so the guard costs a flag test, not an event per call
            torch.cuda.current_stream(self.device).synchronize()
        self._a_host_np[:] = a
        self._a_dev.view(-1).copy_(self._a_host, non_blocking=True)
        self._a_pending = True
        return self._a_dev

    def read_transition(self):
        """The last step's outputs on the host with ONE device-to-host copy: (obs [N, D] float32, reward [N] float32,
        terminated [N] uint8, truncated [N] uint8) as NumPy views of a pinned buffer (valid until the next call).
        What a scalar caller of ``step`` needs (the reference's ``CARLEnv.step`` returns Python types,
        carl/envs/carl_env.py:321-342)."""
        if self._out_host is None:
            self._out_host = torch.empty(self._out_slab.shape, dtype=torch.uint8).pin_memory()
        self._out_host.copy_(self._out_slab, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        self._a_pending = False  # (everything on the stream is done: the staged action upload too)
        h = self._out_host.numpy()
        n, D = self.n, self.D
        o_rew, o_term, o_trunc, _ = self._out_offsets
        return (h[: 4 * n * D].view(np.float32).reshape(n, D), h[o_rew: o_rew + 4 * n].view(np.float32),
                h[o_term: o_term + n], h[o_trunc: o_trunc + n])

    # ------------------------------------------------------------------ replayable step (hipGraph)
    def snapshot(self) -> dict:
        """Copies of every buffer a launch can change (state, counters, bookkeeping, step outputs)."""
        names = ["state", "elapsed", "ctx_idx", "episode", "n_calls", "ep_return", "last_return", "last_length",
                 "episodes_done", "obs", "reward", "terminated", "truncated", "done", "final_obs", "ctx_obs"]
        names += [k for k in ("goal_pos", "success", "fin_count", "first_state") if getattr(self, k, None) is not None]
        return {k: getattr(self, k).clone() for k in names}

    def restore(self, snap: dict) -> None:
        for k, v in snap.items():
            getattr(self, k).copy_(v)

    def capture_step(self, action_buffer: torch.Tensor, n_steps: int = 1) -> "CapturedStep":
        """Capture ``n_steps`` per-call step launches that read ``action_buffer`` (a device tensor whose
        ADDRESS stays fixed: the policy writes the next action into it) into a hipGraph.  ``replay()`` then
        costs ONE graph launch for ``n_steps`` env steps.  Measured (65 536 CartPole lanes): a graph launch costs
        ~11 us, so a ONE-step graph (10.9 us per env step) would be SLOWER than the eager ``step`` (4.3 us: prepared
        ctypes call, raw stream handle) -- ``n_steps=1`` therefore returns a replayable object that makes the eager call
        on the fixed buffer (round 5; no graph); the captured form pays off from ~4 steps per replay (``n_steps=100``:
        3.7 us per env step, the same as a ``torch.cuda.graph`` around 100 ``step`` calls) and is meant for a policy that
        is captured into the same graph region or for open-loop replays.  Engine state is untouched by the capture."""
        return CapturedStep(self, action_buffer, n_steps)

    _has_direct_kernel = True  # (the classic-control families: staged / direct-store pair of rollout kernels)

    def _row_pitch(self) -> int:
        """Lanes per ROW of a rollout's action / output arrays (``carl_step_io_t.row_pitch``): the lane count rounded up
        to a multiple of 16, so that every row's 16-byte pieces stay aligned and ANY lane count -- 10, 65 537, the uneven
        shards of ``distributed.lane_shard`` -- takes the staged kernel (round 6; such batches used to fall back to the
        ~50 % slower direct-store kernel).  Dense rows under ``FLAG_ROLLOUT_DIRECT`` (the A/B switch)."""
        if self.b.flags & _lib.FLAG_ROLLOUT_DIRECT:
            return self.n
        return int(self.lib.carl_rollout_pitch(self.n))

    def alloc_rollout(self, n_steps: int, final_obs: bool = False) -> dict:
        """Output buffers of a ``rollout`` of ``n_steps``: ``obs [T, N, D]``, ``reward`` / ``terminated`` / ``truncated``
        ``[T, N]``.  For a lane count that is not a multiple of 16 they are VIEWS ``[:, :N]`` of arrays whose rows are
        ``_row_pitch()`` lanes long (non-contiguous; ``.contiguous()`` copies) -- the padding columns receive the
        records of the padding lanes."""
        dev, n, D, P = self.device, self.n, self.D, self._row_pitch()

        def rows(tail, dtype, zero=False):
            full = (torch.zeros if zero else torch.empty)((n_steps, P) + tail, dtype=dtype, device=dev)
            return full if P == n else full[:, :n]

        out = {
            "obs": rows((D,), torch.float32),
            "reward": rows((), torch.float32),
            "terminated": rows((), torch.uint8),
            "truncated": rows((), torch.uint8),
        }
        if final_obs:
            out["final_obs"] = rows((D,), torch.float32, zero=True)
        return out

    def rollout(self, actions, out: dict | None = None) -> dict:
        """T steps in one launch; ``actions`` is [T, N] (or [T, N, 1]).  Every step's full
        transition is written to ``out`` (see ``alloc_rollout``).  Discrete families: int32 / int64, or ``torch.uint8`` --
        one byte per lane-step instead of four on the launch's only per-step read stream (CartPole x 65 536: +15 %
        env-steps/s); Box families: float32, or ``torch.float16`` / ``torch.bfloat16`` (widened exactly).  Same transitions
        bit for bit as the wide launch fed the same values."""
        T = int(actions.shape[0])
        a, dt = self._action_tensor(actions, (T,), allow_narrow=True)
        if out is None:
            out = self.alloc_rollout(T)
        io = self._rollout_io(a, dt, out, T)
        if self._has_direct_kernel and not self._warned_direct and self._takes_direct_kernel(io.row_pitch or self.n):
            import warnings

            self._warned_direct = True
            warnings.warn(f"carl_rollout: {self.n} lanes in rows {io.row_pitch or self.n} lanes long (not a multiple of 16, or "
                          "a view into a wider array) -- this launch takes the direct-store kernel (~50 % slower than the "
                          "staged one; same results).  Use alloc_rollout()'s buffers (rows padded to a multiple of 16) for "
                          "the fast path.",
                          RuntimeWarning, stacklevel=2)
        with torch.cuda.device(self.device):
            code = self._c_rollout(io, T)
            if code == _lib.ERR_UNSUPPORTED and dt in (_lib.ACTION_U8, _lib.ACTION_F16, _lib.ACTION_BF16):
                # the narrow formats are read by the lean staged rollout only (moving selectors, the finished-episode
                # log, terminal observations and dense rows of an odd lane count take kernels that read int32 /
                # float32): widen once, same results
                a, dt = self._action_tensor(a.to(torch.int32 if dt == _lib.ACTION_U8 else torch.float32), (T,))
                io = self._rollout_io(a, dt, out, T)
                code = self._c_rollout(io, T)
            _lib.check(code)
        return out

    def _takes_direct_kernel(self, pitch: int) -> bool:
        """carl_amd.hip's rule (rollout_variant) for rows of this pitch, without the A/B flag"""
        n = self.n
        staged = pitch % 16 == 0 and (n % 16 == 0 or pitch == (n + 15) // 16 * 16)
        return not staged and not (self.b.flags & _lib.FLAG_ROLLOUT_DIRECT)

    def _rollout_io(self, a: torch.Tensor, dt: int, out: dict, T: int) -> "_lib.StepIO":
        """``carl_step_io_t`` of a fused rollout: validated actions + the caller's ``[T, ...]`` output buffers.  The row
        pitch is read off the buffers (``alloc_rollout`` pads rows to a multiple of 16 lanes; dense caller-made buffers
        keep working); actions of a padded layout are copied once into rows of the same pitch, the padding columns
        repeating the last lane's action (the padding lanes run as clones of that lane: valid numbers)."""
        if out["reward"].shape[0] < T:
            raise ValueError("rollout output buffers are shorter than the action sequence")
        n, D = self.n, self.D
        P = max(n, int(out["reward"].stride(0))) if out["reward"].dim() == 2 else n  # (a one-row buffer may carry any stride)
        for k, tail in (("obs", (D,)), ("reward", ()), ("terminated", ()), ("truncated", ()), ("final_obs", (D,))):
            t = out.get(k)
            if t is None:
                continue
            inner = int(np.prod(tail, dtype=np.int64)) if tail else 1
            want = (P * inner, inner, 1) if tail else (P, 1)
            if tuple(t.shape[1:]) != (n,) + tail or P < n or any(
                    sz > 1 and st != w for sz, st, w in zip(t.shape, t.stride(), want)):
                raise ValueError(f"rollout output '{k}': shape {tuple(t.shape)} / strides {tuple(t.stride())} do not form "
                                 f"[T, {n}{', ' + str(D) if tail else ''}] rows of one common pitch ({P} lanes)")
        io = _lib.StepIO()
        if P != n:
            a = self._pad_action_rows(a, T, P)
            io.row_pitch = P
        io.action, io.action_dtype = a.data_ptr(), dt
        io.obs, io.reward = _ptr(out["obs"]), _ptr(out["reward"])
        io.terminated, io.truncated = _ptr(out["terminated"]), _ptr(out["truncated"])
        io.final_obs = _ptr(out.get("final_obs"))
        io.branch_sig = _ptr(out.get("branch_sig"))
        self._rollout_actions = a  # (keeps a padded copy alive until the launch has been enqueued and beyond)
        return io

    def _pad_action_rows(self, a: torch.Tensor, T: int, P: int) -> torch.Tensor:
        key = (T, P, a.dtype)
        buf = self._act_pad.get(key) if getattr(self, "_act_pad", None) else None
        if buf is None:
            self._act_pad = {key: torch.empty((T, P), dtype=a.dtype, device=self.device)}  # (one size kept)
            buf = self._act_pad[key]
        rows = a.view(T, self.n)
        buf[:, : self.n] = rows
        buf[:, self.n:] = rows[:, self.n - 1:]
        return buf

    def rollout_variant(self) -> int:
        """Which kernel ``rollout`` launches for this batch into ``alloc_rollout``'s buffers: ``_lib.ROLLOUT_STAGED``
        (fast path: rows of a pitch that is a multiple of 16 -- any lane count since round 6), ``ROLLOUT_DIRECT_FLAG``
        (the A/B switch); ``ROLLOUT_DIRECT_SHAPE`` only for caller-made dense buffers of an odd lane count
        (carl_rollout_variant_io)."""
        io = _lib.StepIO()
        io.row_pitch = self._row_pitch()
        return int(self.lib.carl_rollout_variant_io(C.byref(self.b), C.byref(io)))

    def drain_finished(self):
        """Finished-episode log since the last drain -> (global lane ids, returns, lengths,
        n_dropped); resets the counter.  Synchronises."""
        if self.fin_count is None:
            raise RuntimeError("engine was built with fin_capacity=0")
        n = int(self.fin_count.item())
        k = min(n, self.fin_capacity)
        out = (self.fin_lane[:k].clone(), self.fin_return[:k].clone(), self.fin_length[:k].clone(), n - k)
        self.fin_count.zero_()
        return out
