"""Test scaffolding for carl_amd/dropin.py (the object the REFERENCE's CARLEnv wraps).

* ``Wrapper`` + ``RefSequenceEnv``: a ~40-line stand-in that replays the reference's CALL SEQUENCE on the wrapped
  env -- which attribute it reads, which it sets and in what order -- citing the reference line of every call.  It is
  not the reference's code (gymnasium and ConfigSpace are not installable here, the reference classes cannot be
  imported): it exists so that the shim is driven the way ``CARLCartPole(env=shim)`` would drive it.
* ``OracleBackedEngine``: the ``VecEngine`` surface the shim uses, answered by the CPU oracle, so that the protocol
  layer (spaces, setattr broadcast, state write, return types) is tested without a GPU.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import numpy as np

from carl_amd import _lib
from oracle import oracle as O


class Wrapper:
    """gymnasium.Wrapper (0.29) as far as CARLEnv uses it: stores env, forwards"""

    def __init__(self, env):
        self.env = env

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def action_space(self):
        return self.env.action_space

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        return self.env.step(action)


class RefSequenceEnv(Wrapper):
    """The calls ``CARLCartPole`` / ``CARLPendulum`` make on ``env``, in the reference's order."""

    def __init__(self, env, contexts, selector_cls, family="cartpole"):
        super().__init__(env)                                    # carl_env.py:75
        self.base_observation_space = env.observation_space     # carl_env.py:77
        self.contexts, self.context, self.family = contexts, None, family
        self.context_selector = selector_cls(contexts=contexts)  # carl_env.py:91-103

    @property
    def context_id(self):
        return self.context_selector.context_id                 # carl_env.py:118-120

    def reset(self, *, seed=None, options=None):
        last = self.context_id                                   # carl_env.py:266
        self.context = self.context_selector.select()            # carl_env.py:243
        if self.context_id != last:                              # carl_env.py:268
            for k, v in self.context.items():                    # carl_gymnasium_env.py:75-77
                setattr(self.env.unwrapped, k, v)
        super().reset(seed=seed, options=options)                # carl_env.py:271 via carl_cartpole.py:50
        c = self.context
        if self.family == "cartpole":                            # carl_cartpole.py:51-62
            self.env.unwrapped.state = self.env.np_random.uniform(
                low=c["initial_state_lower"], high=c["initial_state_upper"], size=(4,))
            state = np.array(self.env.unwrapped.state, dtype=np.float32)
        else:                                                    # carl_pendulum.py:48-64
            theta = self.env.np_random.uniform(high=c["initial_angle_max"])
            thetadot = self.env.np_random.uniform(high=c["initial_velocity_max"])
            self.env.unwrapped.state = np.array([theta, thetadot], dtype=np.float32)
            self.env.unwrapped.last_u = None
            state = np.array([np.cos(theta), np.sin(theta), thetadot], dtype=np.float32)
        return {"obs": state, "context": dict(c)}, {"context_id": self.context_id}  # carl_env.py:272-274

    def step(self, action):
        state, reward, terminated, truncated, info = super().step(action)  # carl_env.py:339
        info["context_id"] = self.context_id                               # carl_env.py:341
        return {"obs": state, "context": dict(self.context)}, reward, terminated, truncated, info


class OracleBackedEngine:
    """``VecEngine``'s surface (as carl_amd/dropin.py uses it) on top of the CPU oracle."""

    def __init__(self, family: int, defaults, n: int, seed: int = 0, max_steps=None):
        self.family, self.n = family, n
        self.info = _lib.family_info(family)  # static facts come from the product library (no GPU needed)
        self._o = O.Engine(family, np.asarray([defaults], dtype=np.float64), n, selector=O.SEL_STATIC, autoreset=n > 1,  # (static = the ids stay where the host put them)
                           seed=seed, max_steps=max_steps, ctx_idx0=np.zeros(n, np.int32))
        self.S, self.D = self._o.S, self._o.D
        self.auto_reset = n > 1
        self.final_obs = np.zeros((n, self.D), np.float32)
        self.done = np.zeros(n, np.uint8)

    state = property(lambda self: self._o.state.T)       # [S][N] view of the oracle's [N][S]
    ctx_table = property(lambda self: self._o.ctx.T)     # [F][C] view of the oracle's [C][F]
    ctx_idx = property(lambda self: self._o.ctx_idx)

    def seed(self, seed):
        self._o.seed(seed)

    def reset(self, mask=None):
        return self._o.reset(mask)

    def step(self, action):
        out = self._o.step(action)
        self.final_obs, self.done = out.final_obs, (out.terminated | out.truncated)
        return out.obs, out.reward, out.terminated, out.truncated

    def set_contexts(self, table, ctx_idx=None):
        t = np.ascontiguousarray(table, dtype=np.float64)
        self._o.ctx = t
        self._o.cfg.n_contexts = t.shape[0]
        self._o.ctx_idx[:] = (np.arange(self.n) % t.shape[0]) if ctx_idx is None else np.asarray(ctx_idx)

    def refresh_ctx_obs(self):
        pass


# ---- Brax: the reference's call sequence on the wrapped env (carl/envs/brax/carl_brax_env.py) ----------------------
class FakeBraxSystem:
    """What the reference's ``_update_context`` leaves in the brax ``System`` it assigns (duck-typed: brax is not
    installable here): gravity vector, ang_damping (overwritten by viscosity: :276-279), per-geom friction /
    elasticity arrays, link masses by name (``set_masses``: :81-101)."""

    def __init__(self, context: dict, link_names: list[str], n_geoms: int = 21):
        from types import SimpleNamespace

        self.gravity = np.array([0.0, 0.0, context.get("gravity", -9.81)])                      # :272-273
        self.ang_damping = context.get("ang_damping", -0.05)                                     # :274-275
        if "viscosity" in context:
            self.ang_damping = context["viscosity"]                                              # :276-277 (Quirk B2)
        mass = np.ones(len(link_names))
        for k, v in context.items():                                                             # set_masses, :57-73
            if k.startswith("mass"):
                name = k.split("_", 1)[-1]
                if name not in link_names:
                    raise RuntimeError(f"Link {name} not in available link names {link_names}.")
                mass[link_names.index(name)] = v
        self.link = SimpleNamespace(inertia=SimpleNamespace(mass=mass))
        self.link_names = link_names
        self.geom_friction = np.ones((n_geoms, 3))
        if "friction" in context:
            self.geom_friction[:, 0] = context["friction"]                                       # :281-284
        self.elasticity = np.zeros(n_geoms)
        if "elasticity" in context:
            self.elasticity[:] = context["elasticity"]                                           # :285-288


class RefBraxSequenceEnv(Wrapper):
    """The calls ``CARLBraxEnv`` makes on ``env``, in the reference's order."""

    def __init__(self, env, contexts, selector_cls, link_names):
        super().__init__(env)                                    # carl_env.py:75
        self._brax_env = env.unwrapped                           # carl_brax_env.py:193
        self.base_observation_space = env.observation_space     # carl_env.py:77
        self.contexts, self.context, self.link_names = contexts, None, link_names
        self.context_selector = selector_cls(contexts=contexts)

    @property
    def context_id(self):
        return self.context_selector.context_id

    def _update_context(self):                                   # carl_brax_env.py:255-292
        self.env.unwrapped.sys = FakeBraxSystem(self.context, self.link_names)

    def reset(self, *, seed=None, options=None):                 # carl_brax_env.py:294-306
        last = self.context_id
        self.context = self.context_selector.select()
        if self.context_id != last:
            self._update_context()
        self.env.context = self.context
        state, info = self.env.reset(seed=seed, options=options)
        info["context_id"] = self.context_id
        return {"obs": state, "context": dict(self.context)}, info

    def step(self, action):
        state, reward, terminated, truncated, info = super().step(action)  # carl_env.py:339
        info["context_id"] = self.context_id
        return {"obs": state, "context": dict(self.context)}, reward, terminated, truncated, info
