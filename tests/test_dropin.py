"""carl_amd/dropin.py::Mi355xVecEnv -- the object the REFERENCE's ``CARLEnv`` wraps.

CPU half: the shim's protocol layer driven by the reference's call sequence (tests/dropin_util.py::RefSequenceEnv,
every call cited to carl_env.py / carl_gymnasium_env.py / carl_cartpole.py) on an oracle-backed engine, against an
independent scalar restatement of the same loop (oracle/ref_style.py).  The GPU half (tests/test_gpu_dropin.py) runs
the same sequence on the HIP engine and compares with the mirror class bit for bit."""
import os

import numpy as np
import pytest

from carl_amd import _lib, spaces
from carl_amd.context.selection import RoundRobinSelector
from carl_amd.dropin import Mi355xVecEnv
from dropin_util import OracleBackedEngine, RefSequenceEnv
from oracle import oracle as O


def _contexts(family):
    base = dict(O.FEATURES[family])
    if family == O.CARTPOLE:
        return {i: {**base, "gravity": g, "length": l, "tau": 0.02 + 0.01 * i}
                for i, (g, l) in enumerate([(9.8, 0.5), (15.0, 1.2), (5.0, 0.3)])}
    return {i: {**base, "g": g, "l": l} for i, (g, l) in enumerate([(10.0, 1.0), (4.2, 1.7), (19.0, 0.6)])}


def _shim(family, n=1, seed=0):
    name = O.FAMILY_NAMES[family]
    eng = OracleBackedEngine(family, O.default_row(family), n, seed=seed)
    return Mi355xVecEnv(name, n, engine=eng, seed=seed)


def test_surface_the_reference_reads():
    """carl_env.py:75-77 + gymnasium.Wrapper's forwards: spaces, unwrapped, np_random, spec"""
    env = _shim(O.CARTPOLE)
    assert env.unwrapped is env
    assert isinstance(env.observation_space, spaces.Box) and env.observation_space.shape == (4,)
    assert isinstance(env.action_space, spaces.Discrete) and env.action_space.n == 2
    assert isinstance(env.np_random, np.random.Generator)
    assert env.spec.id == "CartPole-v1" and env.spec.max_episode_steps == 500
    assert env.metadata == {"render_modes": []} and env.render_mode is None
    pen = _shim(O.PENDULUM)
    assert pen.observation_space.shape == (3,) and pen.action_space.shape == (1,)
    assert float(pen.action_space.low[0]) == -2.0 and float(pen.action_space.high[0]) == 2.0
    with pytest.raises(ValueError):
        Mi355xVecEnv("LunarLander-v2", engine=object())


def test_setattr_protocol_and_state_access():
    """carl_gymnasium_env.py:75-77 (setattr per feature) and carl_cartpole.py:51 (state write)"""
    env = _shim(O.CARTPOLE)
    u = env.unwrapped
    for k, v in {"gravity": 15.0, "length": 1.2, "tau": 0.05}.items():
        setattr(u, k, v)
    assert u.gravity == 15.0 and u.length == 1.2 and u.tau == 0.05 and u.masscart == 1.0
    assert env.eng.ctx_table[0, 0] == 15.0  # the engine's table row, not a Python attribute
    u.last_u = None                          # carl_pendulum.py writes a non-feature attribute: stays a plain one
    assert u.last_u is None
    env.reset(seed=3)
    u.state = np.array([0.01, -0.02, 0.03, 0.04])
    got = u.state
    assert got.dtype == np.float64 and got.shape == (4,)
    np.testing.assert_allclose(got, np.float32([0.01, -0.02, 0.03, 0.04]).astype(np.float64))
    with pytest.raises(AttributeError):
        u.no_such_feature


def test_reset_reseeds_np_random_like_gymnasium():
    env = _shim(O.CARTPOLE)
    env.reset(seed=7)
    a = env.np_random.uniform(size=3)
    env.reset(seed=7)
    np.testing.assert_array_equal(a, env.np_random.uniform(size=3))


@pytest.mark.parametrize("family", [O.CARTPOLE, O.PENDULUM])
def test_reference_call_sequence_matches_the_scalar_restatement(family):
    """The reference's sequence over the shim == the SURVEY 8c known-answer arithmetic: three contexts, round robin,
    the reference's host-side init-state draw written through ``unwrapped.state``, 60 steps per episode."""
    contexts = _contexts(family)
    name = O.FAMILY_NAMES[family]
    env = RefSequenceEnv(_shim(family, seed=11), contexts, RoundRobinSelector, name)
    rng = np.random.default_rng(5)
    for episode in range(4):
        obs, info = env.reset(seed=100 + episode)
        assert info["context_id"] == episode % 3 and obs["obs"].dtype == np.float32
        ctx = contexts[episode % 3]
        state = np.asarray(env.env.unwrapped.state, dtype=np.float64)[None]
        row = np.array([[ctx[k] for k in O.feature_names(family)]])
        for t in range(60):
            a = int(rng.integers(0, 2)) if family == O.CARTPOLE else np.float32([rng.uniform(-2, 2)])
            o, r, term, trunc, info = env.step(a)
            assert type(r) is float and type(term) is bool and type(trunc) is bool and info["context_id"] == episode % 3
            s2, o_want, r_want, t_want = O.transitions(family, row, state, np.asarray(a).reshape(1))
            np.testing.assert_allclose(o["obs"], o_want[0], rtol=1e-6, atol=1e-7)
            assert abs(r - float(r_want[0])) <= 1e-6 * (1 + abs(r)) and term == bool(t_want[0])
            # the engine carries its float32 state; re-anchor the restatement on it (per-step comparison)
            state = np.asarray(env.env.unwrapped.state, dtype=np.float64)[None]
            if term:
                break


def test_batched_path_has_the_vector_env_shape():
    """SURVEY 8b "existing batched precedent" (wrappers.py:93-145): num_envs, batched spaces, arrays in / out"""
    n = 8
    env = _shim(O.CARTPOLE, n=n)
    assert env.num_envs == n and env.observation_space.shape == (n, 4) and env.single_observation_space.shape == (4,)
    table = np.tile(O.default_row(O.CARTPOLE), (4, 1))
    table[:, 0] = [5.0, 9.8, 12.0, 15.0]
    env.set_contexts(table, np.arange(n) % 4)
    obs, info = env.reset(seed=1)
    assert obs.shape == (n, 4) and info == {}
    o, r, te, tr, info = env.step(np.ones(n, np.int32))
    assert o.shape == (n, 4) and r.shape == (n,) and te.shape == (n,) and "final_observation" in info
    env.unwrapped.gravity = 3.0  # scalar broadcast: every row of the column
    assert (env.eng.ctx_table[0] == 3.0).all()


def test_brax_shim_reads_the_system_the_reference_assigns():
    """carl_brax_env.py:292 (`self.env.unwrapped.sys = sys`): the fields `_update_context` wrote (:272-290) land in the
    engine's context columns -- including the reference's literal viscosity -> ang_damping overwrite (Quirk B2)"""
    from types import SimpleNamespace

    from carl_amd.dropin import Mi355xBraxVecEnv
    from carl_amd.envs import CARLBraxAnt
    from carl_amd.envs.brax import models
    from dropin_util import FakeBraxSystem

    names = list(CARLBraxAnt.get_context_features().keys())
    default = np.array([[float(f.default_value)] for f in CARLBraxAnt.get_context_features().values()], dtype=np.float32)
    eng = SimpleNamespace(sys=models.SYSTEMS["ant"](names), ctx_table=default.copy(), ctx_idx=np.zeros(1, np.int32), n=1)
    env = Mi355xBraxVecEnv("ant", 1, engine=eng)
    assert env.unwrapped is env and env.observation_space.shape == (27,) and env.action_space.shape == (8,)
    assert float(env.action_space.low[0]) == -1.0 and float(env.action_space.high[0]) == 1.0
    links = ["torso"] + [f"l{i}" for i in range(8)]
    ctx = {"gravity": -12.5, "friction": 0.7, "elasticity": 0.2, "ang_damping": -0.3, "mass_torso": 14.0, "viscosity": 0.0}
    env.unwrapped.sys = FakeBraxSystem(ctx, links)
    t = {n: float(eng.ctx_table[names.index(n), 0]) for n in names}
    assert t["gravity"] == pytest.approx(-12.5) and t["friction"] == pytest.approx(0.7) and t["elasticity"] == pytest.approx(0.2)
    assert t["mass_torso"] == pytest.approx(14.0)
    assert t["ang_damping"] == 0.0  # viscosity (0) was written into ang_damping AFTER the ang_damping line (:276-279)
    assert t["target_distance"] == 100.0  # untouched columns keep their defaults
    env.context = ctx  # carl_brax_env.py:302: a plain attribute
    assert env.context is ctx and env.sys.link_names == links
    with pytest.raises(RuntimeError):  # the reference's own error for an unknown link (set_masses)
        FakeBraxSystem({"mass_wing": 1.0}, links)
    with pytest.raises(ValueError):
        Mi355xBraxVecEnv("quadruped", engine=eng)


def test_shims_are_gymnasium_envs_where_gymnasium_exists():
    """In the reference's own environment gymnasium is installed and `gymnasium.Wrapper.__init__` may insist on
    `isinstance(env, gymnasium.Env)`: the shims then subclass it.  Simulated with a stand-in module that has
    gymnasium.Env's property surface (`unwrapped`, `np_random` with a setter) -- in a subprocess, this image has none."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import sys, types
        import numpy as np
        g = types.ModuleType("gymnasium")
        class Env:
            metadata = {"render_modes": []}
            render_mode = None
            spec = None
            _np_random = None
            @property
            def unwrapped(self):
                return self
            @property
            def np_random(self):
                if self._np_random is None:
                    self._np_random = np.random.default_rng()
                return self._np_random
            @np_random.setter
            def np_random(self, v):
                self._np_random = v
            def reset(self, *, seed=None, options=None):
                raise NotImplementedError
        g.Env = Env
        sys.modules["gymnasium"] = g
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from carl_amd.dropin import Mi355xVecEnv, Mi355xBraxVecEnv
        from dropin_util import OracleBackedEngine
        from oracle import oracle as O
        env = Mi355xVecEnv("CartPole-v1", 1, engine=OracleBackedEngine(O.CARTPOLE, O.default_row(O.CARTPOLE), 1))
        assert isinstance(env, Env) and issubclass(Mi355xBraxVecEnv, Env) and env.unwrapped is env
        env.reset(seed=3)
        a = env.np_random.uniform(size=2)
        env.reset(seed=3)
        assert (a == env.np_random.uniform(size=2)).all()
        env.unwrapped.gravity = 12.0
        assert env.unwrapped.gravity == 12.0
        env.unwrapped.state = np.array([0.01, 0.0, 0.02, 0.0])
        o, r, te, tr, info = env.step(1)
        assert o.shape == (4,) and r == 1.0 and te is False
        print("ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
