"""ctypes binding of oracle/brax_spring.c -- TEST INFRASTRUCTURE, not product code.

The model table is passed as the raw bytes of a ``carl_brax_sys_t`` (include/carl_amd.h)
so this module does not import the product.  PARITY UNPINNED (see brax_spring.c)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import oracle as O


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SysSnapshot:
    """The bytes and the handful of ints of a ``carl_brax_sys_t``, picklable: lets a worker PROCESS drive the restatement
    without importing the product package (whose ctypes binding imports torch -- 256 bench workers doing that cost the
    round-6 bench four minutes).  Accepted wherever a model table is (``Engine``, ``forward_kinematics`` ...)."""

    _INTS = ("n_links", "n_q", "n_dof", "n_act", "obs_dim", "max_episode_steps", "n_frames")

    def __init__(self, sys_struct):
        self.raw = bytes(sys_struct)
        for k in self._INTS:
            setattr(self, k, int(getattr(sys_struct, k)))
        self.act_lo = [float(sys_struct.act_lo[i]) for i in range(self.n_act)]
        self.act_hi = [float(sys_struct.act_hi[i]) for i in range(self.n_act)]

    def __bytes__(self):
        return self.raw


class _Sys:
    """holds the struct bytes and the handful of ints the binding needs"""

    def __init__(self, sys_struct):
        size = len(sys_struct.raw) if isinstance(sys_struct, SysSnapshot) else C.sizeof(sys_struct)
        self.buf = C.create_string_buffer(bytes(sys_struct), size)
        self.n_links, self.n_q, self.n_dof = sys_struct.n_links, sys_struct.n_q, sys_struct.n_dof
        self.n_act, self.obs_dim = sys_struct.n_act, sys_struct.obs_dim
        self.max_episode_steps = sys_struct.max_episode_steps

    @property
    def ptr(self):
        return C.cast(self.buf, C.c_void_p)


def forward_kinematics(sys_struct, q, qd) -> np.ndarray:
    s = _Sys(sys_struct)
    q = np.ascontiguousarray(q, dtype=np.float64)
    qd = np.ascontiguousarray(qd, dtype=np.float64)
    st = np.zeros(13 * s.n_links, dtype=np.float64)
    O.lib().obx_forward_kinematics(s.ptr, _p(q), _p(qd), _p(st))
    return st.reshape(s.n_links, 13)


def inverse_kinematics(sys_struct, state):
    s = _Sys(sys_struct)
    st = np.ascontiguousarray(state, dtype=np.float64).reshape(-1)
    q = np.zeros(s.n_q)
    qd = np.zeros(s.n_dof)
    O.lib().obx_inverse_kinematics(s.ptr, _p(st), _p(q), _p(qd))
    return q, qd


def goal_step(sys_struct, ctx_row, obs, pos):
    """one step of the goal-wrapper epilogue: (reward, success); ``pos`` (float64[2]) is advanced in place"""
    s = _Sys(sys_struct)
    row = np.ascontiguousarray(ctx_row, dtype=np.float64)
    o = np.ascontiguousarray(obs, dtype=np.float32)
    ok = C.c_int32(0)
    fn = O.lib().obx_goal_step
    fn.restype = C.c_double
    r = fn(s.ptr, _p(row), _p(o), _p(pos), C.byref(ok))
    return float(r), int(ok.value)


def substeps(sys_struct, ctx_row, tau, n_sub, state) -> np.ndarray:
    s = _Sys(sys_struct)
    st = np.ascontiguousarray(state, dtype=np.float64).reshape(-1).copy()
    row = np.ascontiguousarray(ctx_row, dtype=np.float64)
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    O.lib().obx_substeps(s.ptr, _p(row), _p(tau), C.c_int(n_sub), _p(st))
    return st.reshape(s.n_links, 13)


def joint_wrenches(sys_struct, ctx_row, tau, state):
    """net joint force / torque (about the COM, world frame) per link of one state: (F [L, 3], T [L, 3])"""
    s = _Sys(sys_struct)
    st = np.ascontiguousarray(state, dtype=np.float64).reshape(-1)
    row = np.ascontiguousarray(ctx_row, dtype=np.float64)
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    F = np.zeros((s.n_links, 3))
    T = np.zeros((s.n_links, 3))
    O.lib().obx_joint_wrenches(s.ptr, _p(row), _p(tau), _p(st), _p(F), _p(T))
    return F, T


class Engine:
    """Batched engine semantics for a Brax family (same contract as oracle.Engine)."""

    def __init__(self, sys_struct, ctx_table, n_lanes, *, selector=O.SEL_ROUND_ROBIN, selector_stride=1,
                 autoreset=True, max_steps=None, seed=0, lane_offset=0, ctx_idx0=None, autoreset_mode="redraw"):
        self.sys = _Sys(sys_struct)
        self.ctx = np.ascontiguousarray(ctx_table, dtype=np.float64)
        self.F = self.ctx.shape[1]
        n_ctx = self.ctx.shape[0]
        self.cfg = O._Cfg(0, n_lanes, n_ctx, self.sys.max_episode_steps if max_steps is None else max_steps,
                          selector, selector_stride, (2 if autoreset_mode == "first_state" else 1) * int(autoreset), 0,
                          lane_offset, seed)
        n, S, D = n_lanes, 13 * self.sys.n_links, self.sys.obs_dim
        # autoreset_mode "first_state": brax's AutoResetWrapper (the state of the last explicit reset)
        self.first_state = np.zeros((n_lanes, S), dtype=np.float64) if autoreset_mode == "first_state" else None
        self.n, self.S, self.D = n, S, D
        self.state = np.zeros((n, S), dtype=np.float64)
        self.elapsed = np.zeros(n, dtype=np.int32)
        g = lane_offset + np.arange(n, dtype=np.int64)
        if ctx_idx0 is not None:
            self.ctx_idx = np.ascontiguousarray(ctx_idx0, dtype=np.int32).copy()
        elif selector == O.SEL_ROUND_ROBIN:
            self.ctx_idx = ((g - selector_stride) % n_ctx).astype(np.int32)
        else:
            self.ctx_idx = (g % n_ctx).astype(np.int32)
        self.episode = np.zeros(n, dtype=np.uint32)
        self.n_calls = np.zeros(n, dtype=np.int32)
        self.ep_return = np.zeros(n, dtype=np.float64)
        self.obs = np.zeros((n, D), dtype=np.float32)
        self.last_return = np.zeros(n, dtype=np.float32)
        self.last_length = np.zeros(n, dtype=np.int32)
        self.episodes_done = np.zeros(n, dtype=np.int32)
        self.goal_pos = np.zeros((n, 2), dtype=np.float64)
        self.success = np.zeros(n, dtype=np.uint8)

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        O.lib().obx_engine_reset(self.sys.ptr, C.byref(self.cfg), _p(self.ctx), C.c_int(self.F), _p(m),
                                 _p(self.state), _p(self.elapsed), _p(self.ctx_idx), _p(self.episode),
                                 _p(self.n_calls), _p(self.ep_return), _p(self.obs), _p(self.goal_pos),
                                 _p(self.first_state))
        return self.obs.copy()

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float32).reshape(self.n, self.sys.n_act)
        rew = np.empty(self.n, dtype=np.float32)
        term = np.empty(self.n, dtype=np.uint8)
        trunc = np.empty(self.n, dtype=np.uint8)
        final_obs = np.full((self.n, self.D), np.nan, dtype=np.float32)
        self.branch_sig = np.zeros((self.n, 2), dtype=np.uint32)  # discrete decisions of this step (brax_spring.c: substep)
        O.lib().obx_engine_step(self.sys.ptr, C.byref(self.cfg), _p(self.ctx), C.c_int(self.F), _p(a),
                                _p(self.state), _p(self.elapsed), _p(self.ctx_idx), _p(self.episode),
                                _p(self.n_calls), _p(self.ep_return), _p(self.obs), _p(rew), _p(term), _p(trunc),
                                _p(final_obs), _p(self.last_return), _p(self.last_length), _p(self.episodes_done),
                                _p(self.goal_pos), _p(self.success), _p(self.first_state), _p(self.branch_sig))
        return O.StepOut(self.obs.copy(), rew, term, trunc, final_obs)
