"""Lane engine for the Brax-locomotion families: ``VecEngine`` with a model table
(``carl_brax_sys_t``) instead of a built-in family.  Replaces, for N lanes at once, what
``CARLBraxEnv`` builds with ``brax.envs.create(env_name, backend="spring", batch_size)`` +
``VectorGymWrapper`` (reference: carl/envs/brax/carl_brax_env.py:163-190,
carl/envs/brax/wrappers.py:93-158).  No CPU path."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from carl_amd import _lib
from carl_amd.engine import VecEngine, _ptr


@dataclass
class _BraxInfo:
    state_dim: int
    obs_dim: int
    n_features: int
    action_dim: int
    action_is_discrete: int
    n_actions: int
    max_episode_steps: int
    action_low: float
    action_high: float


class BraxVecEngine(VecEngine):
    def __init__(self, sys_table: _lib.BraxSys, n_features: int, ctx_table, n_lanes: int, device="cuda", **kw):
        self.sys = sys_table
        self._n_features = int(n_features)
        kw.pop("cartpole_recompute", None)
        self.goal_pos = self.success = None
        super().__init__(-1, ctx_table, n_lanes, device, **kw)
        if self.sys.goal_mode:  # BraxWalkerGoalWrapper state: integrated (x, y) + per-step success flag
            self.goal_pos = torch.zeros((2, self.n), dtype=torch.float32, device=self.device)
            self.success = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
            self._sync_pointers()
        # device copy of the model table (the kernels stage it into LDS once per workgroup)
        raw = bytes(self.sys)
        self.sys_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)

    def _family_info(self):
        s = self.sys
        return _BraxInfo(_lib.BRAX_LINK_STATE * s.n_links, s.obs_dim, self._n_features, s.n_act, 0, 0,
                         s.max_episode_steps, float(min(s.act_lo[: s.n_act])), float(max(s.act_hi[: s.n_act])))

    def _c_reset(self, mask_ptr) -> int:
        return self.lib.carl_brax_reset(C.byref(self.b), _ptr(self.sys_dev), C.byref(self.sys), mask_ptr,
                                        _ptr(self.obs), self._stream())

    def _c_step(self, io) -> int:
        return self.lib.carl_brax_step(C.byref(self.b), _ptr(self.sys_dev), C.byref(self.sys), C.byref(io),
                                       self._stream())

    def _c_rollout(self, io, n_steps: int) -> int:
        return self.lib.carl_brax_rollout(C.byref(self.b), _ptr(self.sys_dev), C.byref(self.sys), C.byref(io),
                                          n_steps, self._stream())

    def alloc_rollout(self, n_steps: int, final_obs: bool = False) -> dict:
        out = super().alloc_rollout(n_steps, final_obs)
        if self.sys.goal_mode:
            out["success"] = torch.zeros((n_steps, self.n), dtype=torch.uint8, device=self.device)
        return out

    def rollout(self, actions, out: dict | None = None) -> dict:
        if not self.sys.goal_mode:
            return super().rollout(actions, out)
        if out is None:
            out = self.alloc_rollout(int(actions.shape[0]))
        self.b.success = out["success"].data_ptr()  # [T][N] for the duration of this launch
        try:
            return super().rollout(actions, out)
        finally:
            self.b.success = self.success.data_ptr()

    def reset_indexed(self, idx, count):
        raise NotImplementedError("Brax families reset through a lane mask (reset(mask)) or in-kernel auto-reset")

    def reset_done(self):
        return self.reset((self.terminated | self.truncated))
