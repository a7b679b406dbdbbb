"""The object the REFERENCE's ``CARLEnv`` wraps, backed by the MI355X lane engine.

SURVEY.md 8(b): the reference has no FFI on this path -- its boundary is a Python protocol, "the object
``CARLEnv`` wraps".  ``CARLGymnasiumEnv(env=...)`` accepts that object instead of calling ``gymnasium.make``
(carl/envs/gymnasium/carl_gymnasium_env.py:25,63-64), so a maintainer of the reference binds this engine with

    from carl.envs import CARLCartPole                      # the REFERENCE class, unchanged
    from carl_amd.dropin import Mi355xVecEnv
    env = CARLCartPole(env=Mi355xVecEnv("CartPole-v1"), contexts=contexts)

``Mi355xVecEnv`` implements exactly what the reference calls on ``env`` (file:line of each call site):

* ``gymnasium.Wrapper.__init__(env)`` stores it (carl_env.py:75); ``env.observation_space`` is read at :77,
  ``env.action_space`` through ``Wrapper.action_space``;
* ``reset(*, seed=None, options=None) -> (obs, info)`` (carl_env.py:271) and
  ``step(action) -> (obs, reward, terminated, truncated, info)`` (:339) with gymnasium's scalar return types
  (float32 ndarray, Python float, bool, bool, dict) for ``num_envs == 1``;
* ``env.unwrapped`` with the ``setattr`` protocol of ``_update_context`` (carl_gymnasium_env.py:75-77):
  ``setattr(env.unwrapped, feature, value)`` broadcasts the scalar into the feature's column of the engine's
  context table; unknown names become plain attributes (the reference's Pendulum reset writes ``last_u``);
* ``env.unwrapped.state`` read / write -- the reference's per-class ``reset`` overrides draw the init state on
  the host and assign it (carl_cartpole.py:51, carl_pendulum.py:60, carl_acrobot.py:100,
  carl_mountaincar.py:80, carl_mountaincarcontinuous.py:77): the write lands in the engine's lane state, the
  read returns it as float64 (gymnasium keeps float64 states; MountainCarContinuous float32);
* ``env.np_random`` (``numpy.random.Generator``; those overrides call ``.uniform`` on it), re-seeded by
  ``reset(seed=...)`` like ``gymnasium.Env.reset`` does;
* ``metadata`` / ``render_mode`` / ``spec`` / ``reward_range`` / ``close()``: what ``gymnasium.Wrapper``
  forwards.

The engine runs with ``CARL_SEL_HOST``: it never changes ``ctx_idx`` -- the reference's own selector object and
``_update_context`` keep deciding, exactly as today.  The TimeLimit that ``gymnasium.make`` wraps around the
env (``max_episode_steps`` of the registry) is the engine's ``elapsed`` / ``truncated``.

Batched use (``num_envs = N > 1``; no reference counterpart except ``VectorGymWrapper``,
carl/envs/brax/wrappers.py:93-158): ``set_contexts(table[C, F], ctx_idx[N])`` uploads a dense context set,
``reset`` / ``step`` take and return device tensors ``[N, ...]``, and done lanes are reset inside ``step``.

No CPU path: the engine raises without a ROCm device.  (``engine=`` injects a pre-built engine object with
the ``VecEngine`` surface; the CPU test suite passes an oracle-backed one to exercise this protocol layer.)
"""
from __future__ import annotations

from typing import Any

import numpy as np

from carl_amd import _lib, spaces

# gymnasium registry id (what the reference's classes carry as ``env_name``) / short family name
# -> (engine family, state dtype gymnasium keeps)
_FAMILIES = {
    "CartPole-v1": ("cartpole", _lib.CARTPOLE, np.float64),
    "Pendulum-v1": ("pendulum", _lib.PENDULUM, np.float64),
    "Acrobot-v1": ("acrobot", _lib.ACROBOT, np.float64),
    "MountainCar-v0": ("mountaincar", _lib.MOUNTAINCAR, np.float64),
    "MountainCarContinuous-v0": ("mountaincar_cont", _lib.MOUNTAINCAR_CONT, np.float32),
}
_BY_SHORT = {v[0]: k for k, v in _FAMILIES.items()}


def _family_class(env_id: str):
    """the mirror class that holds the reference's context-feature table (names in column order, defaults) and
    the gymnasium observation space of the family"""
    from carl_amd.envs.gymnasium import classic_control as cc

    return {"CartPole-v1": cc.CARLCartPole, "Pendulum-v1": cc.CARLPendulum, "Acrobot-v1": cc.CARLAcrobot,
            "MountainCar-v0": cc.CARLMountainCar, "MountainCarContinuous-v0": cc.CARLMountainCarContinuous}[env_id]


try:  # where gymnasium IS installed (the reference's own environment) the shims are gymnasium.Env subclasses, so that
    # `gymnasium.Wrapper.__init__` -- which newer gymnasium releases guard with `isinstance(env, Env)` -- accepts them
    import gymnasium as _gymnasium

    _EnvBase = _gymnasium.Env
except Exception:  # this image: no gymnasium; the protocol is duck-typed
    _EnvBase = object


class _Spec:
    """what ``gymnasium.Wrapper.spec`` forwards: the registry id and the TimeLimit"""

    def __init__(self, env_id: str, max_episode_steps: int):
        self.id, self.max_episode_steps = env_id, max_episode_steps


class Mi355xVecEnv(_EnvBase):
    metadata: dict = {"render_modes": []}
    render_mode = None
    reward_range = (-float("inf"), float("inf"))

    def __init__(self, family: str, num_envs: int = 1, device="cuda", *, seed: int = 0,
                 max_episode_steps: int | None = None, auto_reset: bool | None = None, derived: str = "stale",
                 engine=None):
        """``family``: a gymnasium id (``"CartPole-v1"`` -- the reference classes' ``env_name``) or the engine's
        short name (``"cartpole"``).  ``derived="stale"`` replicates the reference's CartPole (Quirk C1)."""
        env_id = _BY_SHORT.get(family, family)
        if env_id not in _FAMILIES:
            raise ValueError(f"unknown family {family!r}; one of {sorted(_FAMILIES) + sorted(_BY_SHORT)}")
        short, fam, state_dtype = _FAMILIES[env_id]
        cls = _family_class(env_id)
        feats = cls.get_context_features()
        d = object.__setattr__  # (this class overrides __setattr__: the context-push protocol)
        d(self, "_names", list(feats.keys()))
        d(self, "_defaults", [float(f.default_value) for f in feats.values()])
        d(self, "_state_dtype", state_dtype)
        d(self, "family", short)
        d(self, "num_envs", int(num_envs))
        if engine is None:
            from carl_amd.engine import VecEngine

            engine = VecEngine(fam, [self._defaults], self.num_envs, device, selector=_lib.SEL_HOST, seed=seed,
                               auto_reset=(self.num_envs > 1) if auto_reset is None else auto_reset,
                               max_episode_steps=max_episode_steps, cartpole_recompute=(derived == "recompute"))
        d(self, "eng", engine)
        info = engine.info
        d(self, "observation_space", cls._base_observation_space(None))
        if info.action_is_discrete:
            d(self, "action_space", spaces.Discrete(int(info.n_actions)))
        else:
            d(self, "action_space", spaces.Box(low=float(info.action_low), high=float(info.action_high), shape=(1,),
                                               dtype=np.float32))
        d(self, "single_observation_space", self.observation_space)
        d(self, "single_action_space", self.action_space)
        if self.num_envs > 1:  # the batched precedent: gymnasium.vector.utils.batch_space (wrappers.py:111-118)
            d(self, "observation_space", spaces.batch_space(self.single_observation_space, self.num_envs))
            d(self, "action_space", spaces.batch_space(self.single_action_space, self.num_envs))
        steps = info.max_episode_steps if max_episode_steps is None else int(max_episode_steps)
        d(self, "spec", _Spec(env_id, steps))
        d(self, "_np_random", np.random.default_rng(seed))

    # ------------------------------------------------------------------ gymnasium.Env surface
    @property
    def unwrapped(self):
        return self  # gymnasium.Env.unwrapped: the base env is its own unwrapped

    @property
    def np_random(self) -> np.random.Generator:
        return self._np_random

    @np_random.setter
    def np_random(self, rng: np.random.Generator) -> None:
        object.__setattr__(self, "_np_random", rng)

    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None):
        """``gymnasium.Env.reset``: re-seeds ``np_random`` (and the engine's Philox key) when a seed is given,
        draws the family's CARL init state on the device (the reference's override then overwrites it with its
        own host draw through ``unwrapped.state``), TimeLimit counter to 0."""
        if seed is not None:
            object.__setattr__(self, "_np_random", np.random.default_rng(seed))
            self.eng.seed(seed)
        obs = self.eng.reset()
        if self.num_envs == 1:
            return self._host(obs)[0].astype(np.float32), {}
        return obs, {}

    def step(self, action):
        eng = self.eng
        if self.num_envs == 1:
            a = eng.stage_scalar_action(action) if hasattr(eng, "stage_scalar_action") else np.asarray(action).reshape(1)
            obs, reward, term, trunc = eng.step(a)
            if hasattr(eng, "read_transition"):  # the lane engine: the whole transition in one device-to-host copy
                o, r, te, tr = eng.read_transition()
                return o[0].astype(np.float32), float(r[0]), bool(te[0]), bool(tr[0]), {}
            return (self._host(obs)[0].astype(np.float32), float(self._host(reward)[0]), bool(self._host(term)[0]),
                    bool(self._host(trunc)[0]), {})
        obs, reward, term, trunc = eng.step(action)
        info = {}
        if getattr(eng, "auto_reset", False):  # gymnasium-0.29 vector-env convention
            info = {"final_observation": eng.final_obs, "_final_observation": eng.done}
        return obs, reward, term, trunc, info

    def close(self) -> None:
        pass

    def render(self):
        return None

    # ------------------------------------------------------------------ context push + state access
    def __setattr__(self, name: str, value) -> None:
        """``setattr(env.unwrapped, k, v)`` (carl_gymnasium_env.py:75-77): a context feature's scalar goes into the
        whole column of the engine's table (every lane reads it from there); ``state`` is the lane state."""
        if name == "state":
            self._set_state(value)
        elif name in self._names:
            col = self.eng.ctx_table[self._names.index(name)]
            col[...] = float(value)
            if self.num_envs > 1 and hasattr(self.eng, "refresh_ctx_obs"):
                self.eng.refresh_ctx_obs()
        else:
            object.__setattr__(self, name, value)

    def __getattr__(self, name: str):  # only called when normal lookup fails
        if name == "state":
            return self._get_state()
        names = object.__getattribute__(self, "_names")
        if name in names:  # the parameter lane 0 currently runs with
            eng = object.__getattribute__(self, "eng")
            return float(eng.ctx_table[names.index(name)][int(eng.ctx_idx[0])])
        raise AttributeError(name)

    @staticmethod
    def _host(t) -> np.ndarray:
        return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)

    def _get_state(self):
        s = self.eng.state  # [S][N] struct-of-arrays
        if self.num_envs == 1:
            return self._host(s)[:, 0].astype(self._state_dtype)
        return s.t() if hasattr(s, "t") else s.T

    def _set_state(self, value) -> None:
        eng = self.eng
        v = np.asarray(self._host(value), dtype=np.float32).reshape(-1, eng.S)
        if v.shape[0] not in (1, self.num_envs):
            raise ValueError(f"state must be [{eng.S}] or [{self.num_envs}, {eng.S}]")
        full = np.array(np.broadcast_to(v, (self.num_envs, eng.S)).T, order="C", copy=True)  # -> [S][N]
        if hasattr(eng.state, "copy_"):
            import torch

            eng.state.copy_(torch.from_numpy(full).to(eng.state.device))
        else:
            eng.state[...] = full

    # ------------------------------------------------------------------ batched path
    def set_contexts(self, table, ctx_idx=None) -> None:
        """Dense context set ``[C, F]`` (the family's feature order) + the row each lane runs (default: lane i ->
        row i mod C): the batched counterpart of the setattr loop (SURVEY.md 8b "Context push")."""
        self.eng.set_contexts(table, ctx_idx)

    @property
    def feature_names(self) -> list[str]:
        return list(self._names)


# ======================================================================================================================
# Brax families: the object the REFERENCE's ``CARLBraxEnv`` wraps
# ======================================================================================================================
class Mi355xBraxVecEnv(_EnvBase):
    """What ``CARLBraxEnv(env=...)`` (carl/envs/brax/carl_brax_env.py:121,163-190) needs from ``env`` -- the surface of
    the reference's ``GymWrapper`` / ``VectorGymWrapper`` around ``brax.envs.create(env_name, backend="spring",
    batch_size)`` (carl/envs/brax/wrappers.py:32-158) -- answered by the lane engine:

    * ``observation_space`` / ``action_space``: ``Box(-inf, inf, [obs])`` and the actuators' control ranges
      (wrappers.py:46-51), batched for ``batch_size > 1`` (:111-118);
    * ``reset(*, seed=None, options=None) -> (obs, {})`` and ``step(action) -> (obs, reward, terminated, False, info)``
      with ``terminated = done`` (brax's EpisodeWrapper folds its 1 000-step truncation into ``done``) and
      ``truncated = False`` (wrappers.py:69-78, 136-145);
    * done envs return to the state of their last explicit ``reset()`` -- brax's ``AutoResetWrapper``, which
      ``brax.envs.create`` puts under both wrappers (``autoreset_mode="first_state"``);
    * ``env.unwrapped.sys = sys`` (carl_brax_env.py:292): the reference rebuilds a brax ``System`` from the context
      (``sys.replace(gravity=..., ang_damping=...)``, ``set_masses``, ``geom_friction.at[:, 0].set``,
      ``elasticity.at[:].set``: :272-290) and assigns it.  The setter reads exactly those fields back -- duck-typed:
      ``sys.gravity[2]``, ``sys.ang_damping``, ``sys.geom_friction[0][0]``, ``sys.elasticity[0]``,
      ``sys.link.inertia.mass[sys.link_names.index(link)]`` -- and writes them into the engine's context columns.
      In the reference that assignment lands on the gym shim and never reaches the jitted step (SURVEY Quirk B1); here it
      DOES move the physics, which is what ``_update_context`` intends;
    * ``env.context = ctx`` (:236, :302): a plain attribute.

    The engine runs with ``CARL_SEL_HOST``: the reference's selector decides.  Goal-directed variants
    (``BraxWalkerGoalWrapper``) are part of the mirror classes (``carl_amd.envs.CARLBrax<Family>``), whose kernels fuse
    the wrapper's arithmetic; this shim is the plain env.  ``brax`` itself is not needed by the shim, but the
    reference's ``_update_context`` imports it to build the ``System`` (``mjcf.load``)."""

    metadata: dict = {"render_modes": []}
    render_mode = None

    def __init__(self, env_name: str, batch_size: int = 1, device="cuda", *, seed: int = 0, engine=None):
        from carl_amd.envs import brax as BX
        from carl_amd.envs.brax import models

        classes = {c.env_name: c for name, c in vars(BX).items()
                   if isinstance(c, type) and getattr(c, "env_name", None) and not name.endswith("Stiffness")}
        if env_name not in classes:
            raise ValueError(f"unknown brax env {env_name!r}; one of {sorted(classes)}")
        feats = classes[env_name].get_context_features()
        self._names = list(feats.keys())
        self._defaults = [float(f.default_value) for f in feats.values()]
        self.env_name, self.num_envs = env_name, int(batch_size)
        if engine is None:
            from carl_amd.brax_engine import BraxVecEngine

            engine = BraxVecEngine(models.SYSTEMS[env_name](self._names), len(self._names), [self._defaults], self.num_envs,
                                   device, selector=_lib.SEL_HOST, auto_reset=True, autoreset_mode="first_state", seed=seed)
        self.eng = engine
        s = engine.sys
        obs = np.inf * np.ones(int(s.obs_dim), dtype=np.float32)
        self.single_observation_space = spaces.Box(-obs, obs, dtype=np.float32)
        self.single_action_space = spaces.Box(np.array(s.act_lo[: s.n_act], dtype=np.float32),
                                              np.array(s.act_hi[: s.n_act], dtype=np.float32), dtype=np.float32)
        batched = self.num_envs > 1
        self.observation_space = spaces.batch_space(self.single_observation_space, self.num_envs) if batched else self.single_observation_space
        self.action_space = spaces.batch_space(self.single_action_space, self.num_envs) if batched else self.single_action_space
        self.context = None
        self._sys = None

    @property
    def unwrapped(self):
        return self

    # ------------------------------------------------------------------ the reference's context push
    @property
    def sys(self):
        return self._sys

    @sys.setter
    def sys(self, system) -> None:
        """``self.env.unwrapped.sys = sys`` (carl_brax_env.py:292): read the fields ``_update_context`` wrote"""
        self._sys = system
        col = {n: i for i, n in enumerate(self._names)}
        table = self.eng.ctx_table

        def put(name, value):
            if name in col:
                table[col[name]][...] = float(value)

        if hasattr(system, "gravity"):
            put("gravity", np.asarray(system.gravity).reshape(-1)[2])
        if hasattr(system, "ang_damping"):
            put("ang_damping", np.asarray(system.ang_damping).reshape(-1)[0])
        if hasattr(system, "geom_friction"):
            put("friction", np.asarray(system.geom_friction).reshape(-1)[0])  # [:, 0] of every geom was set to the context's value
        if hasattr(system, "elasticity"):
            put("elasticity", np.asarray(system.elasticity).reshape(-1)[0])
        link = getattr(system, "link", None)
        names = list(getattr(system, "link_names", []))
        if link is not None and names:
            mass = np.asarray(link.inertia.mass).reshape(-1)
            for n in self._names:
                if n.startswith("mass_") and n.split("_", 1)[-1] in names:
                    put(n, mass[names.index(n.split("_", 1)[-1])])
        if self.num_envs > 1 and hasattr(self.eng, "refresh_ctx_obs"):
            self.eng.refresh_ctx_obs()

    # ------------------------------------------------------------------ GymWrapper / VectorGymWrapper surface
    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None):
        if seed is not None:
            self.eng.seed(seed)
        obs = self.eng.reset()
        if self.num_envs == 1:
            return Mi355xVecEnv._host(obs)[0].astype(np.float32), {}
        return obs, {}

    def step(self, action):
        h = Mi355xVecEnv._host
        if self.num_envs == 1:
            if hasattr(self.eng, "read_transition"):  # the lane engine: pinned action staging, one device-to-host copy
                self.eng.step(self.eng.stage_scalar_action(np.asarray(action, dtype=np.float32)))
                o, r, te, tr = self.eng.read_transition()
                return o[0].astype(np.float32), float(r[0]), bool(te[0]) or bool(tr[0]), False, {}
            a = np.asarray(action, dtype=np.float32).reshape(1, -1)
            obs, reward, term, trunc = self.eng.step(a)
            done = bool(h(term)[0]) or bool(h(trunc)[0])
            return h(obs)[0].astype(np.float32), float(h(reward)[0]), done, False, {}
        obs, reward, term, trunc = self.eng.step(action)
        return obs, reward, (term | trunc), (trunc & 0), {}  # terminated = done, truncated = False (wrappers.py:142-143)

    def close(self) -> None:
        pass

    def render(self):
        return None
