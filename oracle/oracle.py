"""ctypes binding of oracle/carl_oracle.c -- TEST INFRASTRUCTURE, not product code.

Holds its own restatement of the reference's context-feature tables (names, order,
defaults) so that it does not depend on ``carl_amd``:
  carl/envs/gymnasium/classic_control/carl_cartpole.py:15-42
  carl/envs/gymnasium/classic_control/carl_pendulum.py:15-39
  carl/envs/gymnasium/classic_control/carl_acrobot.py:15-69
  carl/envs/gymnasium/classic_control/carl_mountaincar.py:15-51
  carl/envs/gymnasium/classic_control/carl_mountaincarcontinuous.py:15-48

PARITY UNPINNED for the step arithmetic (gymnasium is not importable here and the
reference's tests hold no step values) -- see carl_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcarl_oracle.so")

CARTPOLE, PENDULUM, ACROBOT, MOUNTAINCAR, MOUNTAINCAR_CONT = range(5)
FAMILY_NAMES = ["cartpole", "pendulum", "acrobot", "mountaincar", "mountaincar_cont"]
SEL_STATIC, SEL_ROUND_ROBIN, SEL_RANDOM = range(3)

# (name, default) in get_context_features() order
FEATURES = {
    CARTPOLE: [("gravity", 9.8), ("masscart", 1.0), ("masspole", 0.1), ("length", 0.5),
               ("force_mag", 10.0), ("tau", 0.02), ("initial_state_lower", -0.1),
               ("initial_state_upper", 0.1)],
    PENDULUM: [("gravity", 8.0), ("dt", 0.05), ("g", 10.0), ("m", 1.0), ("l", 1.0),
               ("initial_angle_max", math.pi), ("initial_velocity_max", 1.0)],
    ACROBOT: [("LINK_LENGTH_1", 1.0), ("LINK_LENGTH_2", 1.0), ("LINK_MASS_1", 1.0),
              ("LINK_MASS_2", 1.0), ("LINK_COM_POS_1", 0.5), ("LINK_COM_POS_2", 0.5),
              ("LINK_MOI", 1.0), ("MAX_VEL_1", 4 * math.pi), ("MAX_VEL_2", 9 * math.pi),
              ("torque_noise_max", 0.0), ("INITIAL_ANGLE_LOWER", -0.1),
              ("INITIAL_ANGLE_UPPER", 0.1), ("INITIAL_VELOCITY_LOWER", -0.1),
              ("INITIAL_VELOCITY_UPPER", 0.1)],
    MOUNTAINCAR: [("min_position", -1.2), ("max_position", 0.6), ("max_speed", 0.07),
                  ("goal_position", 0.45), ("goal_velocity", 0.0), ("force", 0.001),
                  ("gravity", 0.0025), ("min_position_start", -0.6),
                  ("max_position_start", -0.4), ("min_velocity_start", 0.0),
                  ("max_velocity_start", 0.0)],
    MOUNTAINCAR_CONT: [("min_position", -1.2), ("max_position", 0.6), ("max_speed", 0.07),
                       ("goal_position", 0.5), ("goal_velocity", 0.0), ("power", 0.0015),
                       ("min_position_start", -0.6), ("max_position_start", -0.4),
                       ("min_velocity_start", 0.0), ("max_velocity_start", 0.0)],
}
STATE_DIM = {CARTPOLE: 4, PENDULUM: 2, ACROBOT: 4, MOUNTAINCAR: 2, MOUNTAINCAR_CONT: 2}
OBS_DIM = {CARTPOLE: 4, PENDULUM: 3, ACROBOT: 6, MOUNTAINCAR: 2, MOUNTAINCAR_CONT: 2}
MAX_STEPS = {CARTPOLE: 500, PENDULUM: 200, ACROBOT: 500, MOUNTAINCAR: 200, MOUNTAINCAR_CONT: 999}
CONTINUOUS = {PENDULUM, MOUNTAINCAR_CONT}


def feature_names(family: int) -> list[str]:
    return [n for n, _ in FEATURES[family]]


def default_row(family: int) -> np.ndarray:
    return np.array([d for _, d in FEATURES[family]], dtype=np.float64)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("carl_oracle.c", "classic_control.inc", "brax_spring.c",
                                              "context_sampler.c", "Makefile")]
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs
    ):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


class _Cfg(C.Structure):
    _fields_ = [
        ("family", C.c_int32), ("n_lanes", C.c_int32), ("n_contexts", C.c_int32),
        ("max_steps", C.c_int32), ("selector", C.c_int32), ("selector_stride", C.c_int32),
        ("autoreset", C.c_int32), ("cartpole_recompute", C.c_int32),
        ("lane_offset", C.c_int64), ("seed", C.c_uint64),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_u01.restype = C.c_float
        _lib.oracle_u01.argtypes = [C.c_uint32]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def philox4x32_10(ctr, key) -> np.ndarray:
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().oracle_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def lane_words(seed: int, glane: int, episode: int, sub: int) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint32)
    lib().oracle_lane_words(C.c_uint64(seed), C.c_uint64(glane), C.c_uint32(episode),
                            C.c_uint32(sub), _p(out))
    return out


def u01(w: int) -> float:
    return float(lib().oracle_u01(C.c_uint32(int(w))))


def transitions(family: int, ctx_rows, state, action, *, precision: str = "f64",
                cartpole_recompute: bool = False):
    """Independent single transitions, row i <- (ctx_rows[i], state[i], action[i]).

    Returns (state', obs f32, reward, terminated u8)."""
    real = np.float64 if precision == "f64" else np.float32
    ctx_rows = np.ascontiguousarray(ctx_rows, dtype=np.float64)
    state = np.ascontiguousarray(state, dtype=real)
    n = state.shape[0]
    act = np.ascontiguousarray(action, dtype=np.float32 if family in CONTINUOUS else np.int32).reshape(n)
    s2 = np.empty_like(state)
    obs = np.empty((n, OBS_DIM[family]), dtype=np.float32)
    rew = np.empty(n, dtype=real)
    term = np.empty(n, dtype=np.uint8)
    fn = getattr(lib(), ("o64_" if precision == "f64" else "o32_") + "transitions")
    fn(C.c_int(family), C.c_int(n), C.c_int(int(cartpole_recompute)), _p(ctx_rows), _p(state),
       _p(act), _p(s2), _p(obs), _p(rew), _p(term))
    return s2, obs, rew, term


def done_compact(terminated, truncated) -> np.ndarray:
    t = np.ascontiguousarray(terminated, dtype=np.uint8)
    u = np.ascontiguousarray(truncated, dtype=np.uint8)
    out = np.empty(t.shape[0], dtype=np.int32)
    k = lib().oracle_done_compact(_p(t), _p(u), C.c_int(t.shape[0]), _p(out))
    return out[:k].copy()


@dataclass
class StepOut:
    obs: np.ndarray
    reward: np.ndarray
    terminated: np.ndarray
    truncated: np.ndarray
    final_obs: np.ndarray


class Engine:
    """Batched engine semantics of SURVEY.md 8(a), restated on the CPU.

    ``ctx_table`` is [C, F] in reference feature order.  Lane i's global id is
    ``lane_offset + i``; results depend on global ids only (multi-GPU invariance).
    """

    def __init__(self, family: int, ctx_table, n_lanes: int, *, selector: int = SEL_ROUND_ROBIN,
                 selector_stride: int = 1, autoreset: bool = True, max_steps: int | None = None,
                 seed: int = 0, lane_offset: int = 0, precision: str = "f64",
                 cartpole_recompute: bool = False, ctx_idx0=None):
        self.family = family
        self.precision = precision
        self.real = np.float64 if precision == "f64" else np.float32
        self.prefix = "o64_" if precision == "f64" else "o32_"
        self.ctx = np.ascontiguousarray(ctx_table, dtype=np.float64)
        assert self.ctx.ndim == 2 and self.ctx.shape[1] == len(FEATURES[family])
        n_ctx = self.ctx.shape[0]
        self.cfg = _Cfg(family, n_lanes, n_ctx, MAX_STEPS[family] if max_steps is None else max_steps,
                        selector, selector_stride, int(autoreset), int(cartpole_recompute),
                        lane_offset, seed)
        n, S, D = n_lanes, STATE_DIM[family], OBS_DIM[family]
        self.n, self.S, self.D = n, S, D
        self.state = np.zeros((n, S), dtype=self.real)
        self.elapsed = np.zeros(n, dtype=np.int32)
        g = lane_offset + np.arange(n, dtype=np.int64)
        if ctx_idx0 is not None:
            self.ctx_idx = np.ascontiguousarray(ctx_idx0, dtype=np.int32).copy()
        elif selector == SEL_ROUND_ROBIN:
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

    def seed(self, seed: int) -> None:
        self.cfg.seed = seed
        self.episode[:] = 0

    def reset(self, mask=None) -> np.ndarray:
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        getattr(lib(), self.prefix + "engine_reset")(
            C.byref(self.cfg), _p(self.ctx), _p(m), _p(self.state), _p(self.elapsed),
            _p(self.ctx_idx), _p(self.episode), _p(self.n_calls), _p(self.ep_return), _p(self.obs))
        return self.obs.copy()

    def step(self, action) -> StepOut:
        act = np.ascontiguousarray(
            action, dtype=np.float32 if self.family in CONTINUOUS else np.int32).reshape(self.n)
        rew = np.empty(self.n, dtype=np.float32)
        term = np.empty(self.n, dtype=np.uint8)
        trunc = np.empty(self.n, dtype=np.uint8)
        final_obs = np.full((self.n, self.D), np.nan, dtype=np.float32)
        getattr(lib(), self.prefix + "engine_step")(
            C.byref(self.cfg), _p(self.ctx), _p(act), _p(self.state), _p(self.elapsed),
            _p(self.ctx_idx), _p(self.episode), _p(self.n_calls), _p(self.ep_return), _p(self.obs),
            _p(rew), _p(term), _p(trunc), _p(final_obs), _p(self.last_return),
            _p(self.last_length), _p(self.episodes_done))
        return StepOut(self.obs.copy(), rew, term, trunc, final_obs)


# ---- device context sampler restatement (oracle/context_sampler.c) -------------------------
def sample_contexts(specs, n_contexts: int, seed: int, context_offset: int = 0) -> np.ndarray:
    """[F][C] float32 table from an array of carl_amd._lib.FeatureSpec (ctypes array)."""
    n_features = len(specs)
    out = np.empty((n_features, n_contexts), dtype=np.float32)
    fn = lib().oracle_sample_contexts
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint64, C.c_void_p]
    fn(C.addressof(specs), n_features, n_contexts, n_contexts, context_offset, seed & (2**64 - 1), _p(out))
    return out


def verify_contexts(specs, table: np.ndarray) -> int:
    table = np.ascontiguousarray(table, dtype=np.float32)
    fn = lib().oracle_verify_contexts
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return int(fn(C.addressof(specs), table.shape[0], table.shape[1], table.shape[1], _p(table)))
