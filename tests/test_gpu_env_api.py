"""The CARLEnv API on the lane engine: the reference's own API-shape tests
(test/test_CARLEnv.py:9-28, test/test_context_selector.py:19-60,
test/test_gymnasium_envs.py:11-39, test/test_all_envs.py:9-22) plus the scalar-mode
parity of BASELINE configs[0] (CARLCartPole, one default context) against the
reference-style Python loop.  Needs an MI355X: the envs have no CPU path."""
import json
import os

import numpy as np
import pytest
import torch

import carl_amd
from carl_amd import envs as E
from carl_amd.context.context_space import NormalFloatContextFeature, UniformFloatContextFeature
from carl_amd.context.sampler import ContextSampler
from carl_amd.context.selection import (
    CustomSelector,
    RandomSelector,
    RoundRobinSelector,
    StaticSelector,
)
from oracle import oracle as O
from oracle import ref_style as R

pytestmark = pytest.mark.gpu

ALL = [E.CARLCartPole, E.CARLPendulum, E.CARLAcrobot, E.CARLMountainCar, E.CARLMountainCarContinuous]


def generate_contexts():
    context = {"dt": 0.03, "gravity": 10.0, "m": 1.0, "l": 1.8}
    return {k: context for k in "abc"}


# ---- test/test_CARLEnv.py ---------------------------------------------------------
def test_observation(device):
    env = E.CARLPendulum()
    context = E.CARLPendulum.get_default_context()
    obs, info = env.reset()
    assert type(obs) is dict and "obs" in obs and "context" in obs
    assert len(obs["context"]) == len(context)
    assert obs["obs"].dtype == np.float32 and obs["obs"].shape == (3,)
    assert info == {"context_id": 0}


def test_observation_emptycontext(device):
    env = E.CARLPendulum(obs_context_features=[])
    state, info = env.reset()
    assert len(state["context"]) == 0


def test_observation_reducedcontext(device):
    n = 3
    keys = list(E.CARLPendulum.get_default_context().keys())[:n]
    env = E.CARLPendulum(obs_context_features=keys)
    state, info = env.reset()
    assert len(state["context"]) == n


def test_observation_vector_mode_order(device):
    env = E.CARLPendulum(obs_context_features=["l", "g"], obs_context_as_dict=False, contexts={0: {"l": 1.5}})
    state, _ = env.reset()
    assert state["context"] == [1.5, 10.0]  # caller order (Quirk S5)
    assert env.observation_space["context"].shape == (2,)


# ---- test/test_context_selector.py --------------------------------------------------
def test_default_selector(device):
    env = E.CARLPendulum(contexts=generate_contexts())
    env.reset()
    assert type(env.context_selector) is RoundRobinSelector and env.context_selector.n_calls == 1
    env.reset()
    assert env.context_selector.n_calls == 2


def test_selector_init_forms(device):
    contexts = generate_contexts()
    env = E.CARLPendulum(contexts=contexts, context_selector=RoundRobinSelector(contexts=contexts))
    assert type(env.context_selector) is RoundRobinSelector
    env = E.CARLPendulum(contexts=contexts, context_selector=RandomSelector(contexts=contexts))
    assert type(env.context_selector) is RandomSelector
    env = E.CARLPendulum(contexts=contexts, context_selector=RandomSelector)
    assert type(env.context_selector) is RandomSelector
    with pytest.raises(ValueError):
        E.CARLPendulum(contexts=contexts, context_selector="bork")


def test_round_robin_ids_and_context_switch(device, golden_dir):
    """examples/sample_contexts_with_brax.ipynb cells 7-11: first reset id 0, second id 1,
    `env.context_id = 4` switches immediately; ids follow the reference's selector run"""
    gold = json.load(open(os.path.join(golden_dir, "selector_sequences.json")))["round_robin_5"]
    s = ContextSampler([NormalFloatContextFeature("g", mu=9.8, sigma=1, upper=50, lower=0)],
                       E.CARLPendulum.get_context_space(), seed=0)
    contexts = s.sample_contexts(5)
    env = E.CARLPendulum(contexts=contexts)
    ids = []
    for _ in range(12):
        obs, info = env.reset()
        ids.append(info["context_id"])
        assert obs["context"]["g"] == contexts[info["context_id"]]["g"]
        assert env.unwrapped.g == pytest.approx(contexts[info["context_id"]]["g"], rel=1e-6)
    assert ids == gold["context_id"]
    env.context_id = 4
    assert env.context_id == 4 and env.context == env.contexts[4]
    assert int(env.env.ctx_idx[0]) == 4
    with pytest.raises(AssertionError):
        env.context_id = 17


def test_custom_selector_scalar_mode(device):
    def fn(inst):
        cid = 1 if inst.n_calls == 0 else 0
        return inst.contexts[inst.contexts_keys[cid]], cid

    env = E.CARLPendulum(contexts=generate_contexts(), context_selector=CustomSelector,
                         context_selector_kwargs={"selector_function": fn})
    assert [env.reset()[1]["context_id"] for _ in range(3)] == [1, 0, 0]
    with pytest.raises(ValueError):
        E.CARLPendulum(contexts=generate_contexts(), num_envs=8, context_selector=CustomSelector,
                       context_selector_kwargs={"selector_function": fn})


# ---- test/test_gymnasium_envs.py, test/test_all_envs.py -------------------------------
@pytest.mark.parametrize("cls", ALL, ids=lambda c: c.__name__)
def test_envs_construct_progress_update_reset(cls, device):
    cls.get_context_features()
    env = cls()
    env._progress_instance()
    env._update_context()
    obs, info = env.reset()
    assert env.observation_space["obs"].contains(obs["obs"])
    a = env.action_space.sample()
    obs, r, term, trunc, info = env.step(a)
    assert isinstance(r, float) and isinstance(term, bool) and isinstance(trunc, bool)
    assert info["context_id"] == env.context_id


def test_registration_and_make(device):
    for cls in ALL:
        env_id = f"carl/{cls.__name__}-v0"
        assert env_id in carl_amd.registry
        assert isinstance(carl_amd.make(env_id), cls)
    with pytest.raises(KeyError):
        carl_amd.make("carl/Nope-v0")


def test_contexts_are_filled_with_defaults_and_validated(device):
    env = E.CARLCartPole(contexts={"x": {"gravity": 5.0}, "y": {}})
    assert env.contexts["x"]["gravity"] == 5.0 and env.contexts["x"]["tau"] == 0.02
    assert env.contexts["y"] == E.CARLCartPole.get_default_context()
    new = {0: {"length": 0.7}}
    env.contexts = new
    assert env.contexts[0]["length"] == 0.7 and env.env.n_contexts == 1


# ---- BASELINE configs[0]: CARLCartPole, 1 default context, scalar API vs ref-style loop ---
@pytest.mark.parametrize("fam,cls", list(zip(range(5), [E.CARLCartPole, E.CARLPendulum, E.CARLAcrobot,
                                                        E.CARLMountainCar, E.CARLMountainCarContinuous])),
                         ids=O.FAMILY_NAMES)
def test_scalar_api_matches_reference_style_loop(fam, cls, device):
    """same seeds -> same trajectory as the reference-style scalar Python stack (fp64),
    step by step through the public API; the loop is re-synchronised to the engine's fp32
    state each step so that 1e-5 is a per-transition statement"""
    rng = np.random.default_rng(fam)
    env = cls(seed=3)
    ref = R.RefStyleEnv(fam)
    for episode in range(2):
        obs, info = env.reset(seed=3 if episode == 0 else None)
        u = [O.u01(w) for w in O.lane_words(3, 0, episode, 0)]
        robs, rinfo = ref.reset(u=u)
        assert info == rinfo
        np.testing.assert_allclose(obs["obs"], robs["obs"], rtol=1e-6, atol=1e-7)
        assert obs["context"] == robs["context"]
        for t in range(60):
            s = env.unwrapped.state
            ref.env.unwrapped.state = tuple(s) if fam in (O.CARTPOLE, O.MOUNTAINCAR) else np.array(s)
            a = R.random_action(fam, rng)
            o1, r1, te1, tr1, i1 = env.step(a)
            o2, r2, te2, tr2, i2 = ref.step(a)
            np.testing.assert_allclose(o1["obs"], o2["obs"], rtol=1e-5, atol=1e-5)
            assert r1 == pytest.approx(r2, rel=1e-5, abs=1e-5)
            assert (te1, tr1, i1) == (te2, tr2, i2)
            if te1 or tr1:
                break


# ---- batched mode --------------------------------------------------------------------
def test_batched_api_shapes_and_semantics(device):
    n = 1000
    s = ContextSampler([UniformFloatContextFeature("g", 1, 20), UniformFloatContextFeature("l", 0.5, 2.0)],
                       E.CARLPendulum.get_context_space(), seed=0)
    table = s.sample_context_table(n)
    env = E.CARLPendulum(contexts=table, num_envs=n, context_selector=StaticSelector, seed=1, max_episode_steps=7)
    obs, info = env.reset(seed=1)
    assert obs["obs"].shape == (n, 3) and obs["obs"].is_cuda
    assert set(obs["context"]) == set(E.CARLPendulum.get_default_context())
    np.testing.assert_array_equal(obs["context"]["g"].cpu().numpy(), table.column("g").astype(np.float32))
    np.testing.assert_array_equal(info["context_id"].cpu().numpy(), np.arange(n))
    assert env.observation_space["obs"].shape == (n, 3) and env.action_space.shape == (n, 1)
    for t in range(1, 15):
        a = torch.rand(n, 1, device=device) * 4 - 2
        obs, rew, term, trunc, info = env.step(a)
        assert rew.shape == (n,) and term.dtype == torch.bool and trunc.dtype == torch.bool
        assert bool(trunc.all()) == (t % 7 == 0)
        if t % 7 == 0:
            assert bool(info["_final_observation"].all())
            assert torch.isfinite(info["final_observation"]).all()
    # vector-mode context
    env2 = E.CARLPendulum(contexts=table, num_envs=n, context_selector=StaticSelector, obs_context_as_dict=False,
                          obs_context_features=["l", "g"])
    obs, _ = env2.reset()
    assert obs["context"].shape == (n, 2)
    np.testing.assert_array_equal(obs["context"][:, 0].cpu().numpy(), table.column("l").astype(np.float32))


def test_batched_round_robin_lanes_advance_on_their_own_resets(device):
    n, n_ctx = 64, 5
    contexts = {i: {"g": float(5 + i)} for i in range(n_ctx)}
    env = E.CARLPendulum(contexts=contexts, num_envs=n, max_episode_steps=4)
    obs, info = env.reset()
    ids = info["context_id"].cpu().numpy().copy()
    np.testing.assert_array_equal(ids, np.arange(n) % n_ctx)
    for t in range(1, 9):
        obs, _, _, trunc, info = env.step(torch.zeros(n, 1, device=device))
        if t % 4 == 0:
            ids = (ids + 1) % n_ctx
        np.testing.assert_array_equal(info["context_id"].cpu().numpy(), ids)
        np.testing.assert_array_equal(obs["context"]["g"].cpu().numpy(), 5.0 + ids)


def test_flatten_observation_adapter(device):
    """the documented SB3 flow wraps CARL envs in FlattenObservation (examples/carl_with_sb3.py:22-36)"""
    from carl_amd.wrappers import FlattenObservation

    env = FlattenObservation(E.CARLPendulum(obs_context_features=["l", "g"]))
    obs, info = env.reset()
    assert obs.shape == (5,) and obs.dtype == np.float32 and env.observation_space.shape == (5,)
    np.testing.assert_allclose(obs[:2], [10.0, 1.0])  # context first, keys sorted: g, l
    o, r, te, tr, info = env.step(np.array([0.5], dtype=np.float32))
    assert o.shape == (5,)
    n = 128
    benv = FlattenObservation(E.CARLPendulum(num_envs=n, obs_context_features=["l", "g"], max_episode_steps=3))
    obs, _ = benv.reset()
    assert obs.shape == (n, 5) and obs.is_cuda and benv.observation_space.shape == (n, 5)
    for _ in range(3):
        obs, r, te, tr, info = benv.step(torch.zeros(n, 1, device=device))
    assert info["final_observation"].shape == (n, 5) and bool(tr.all())


def test_sb3_vecenv_facade_protocol(device):
    """the SB3 ``VecEnv`` protocol (NumPy obs / rewards / dones, per-env info
    dicts with ``terminal_observation`` and ``TimeLimit.truncated`` for finished envs, next-episode
    observation returned) on top of the auto-resetting batched env, flattened like the reference's
    ``FlattenObservation`` flow (examples/carl_with_sb3.py:22-36)."""
    from carl_amd.context.selection import StaticSelector
    from carl_amd.envs import CARLCartPole
    from carl_amd.wrappers import SB3VecEnv

    n = 512
    env = CARLCartPole(num_envs=n, device=device, context_selector=StaticSelector, seed=0, max_episode_steps=30)
    venv = SB3VecEnv(env)
    n_ctx = len(env.obs_context_features)
    obs = venv.reset()
    assert isinstance(obs, np.ndarray) and obs.shape == (n, n_ctx + 4) and obs.dtype == np.float32
    assert venv.observation_space.shape == (n_ctx + 4,) and venv.action_space.n == 2
    rng = np.random.default_rng(0)
    seen_term = seen_trunc = 0
    for t in range(40):
        prev = obs
        a = rng.integers(0, 2, n)
        if t >= 5:  # let some envs survive to the time limit
            a[: n // 8] = (prev[: n // 8, n_ctx + 2] > 0).astype(np.int64)
        obs, rew, dones, infos = venv.step(a)
        assert obs.shape == (n, n_ctx + 4) and rew.shape == (n,) and dones.dtype == np.bool_ and len(infos) == n
        for i in np.nonzero(dones)[0]:
            info = infos[i]
            term_obs = info["terminal_observation"]
            assert term_obs.shape == (n_ctx + 4,)
            if info["TimeLimit.truncated"]:
                seen_trunc += 1
            else:  # a CartPole termination: the terminal state is outside the bounds, the returned one is fresh
                seen_term += 1
                x, th = term_obs[n_ctx], term_obs[n_ctx + 2]
                assert abs(x) > 2.4 or abs(th) > 12 * 2 * np.pi / 360
            assert np.abs(obs[i, n_ctx:]).max() <= 0.1 + 1e-6  # reset draw (CARL: initial_state_lower/upper = -+0.1)
        assert all(infos[i] == {} for i in np.nonzero(~dones)[0][:16])
    assert seen_term > 100 and seen_trunc > 0
    assert venv.get_attr("num_envs")[0] == n and venv.env_is_wrapped(object) == [False] * n


@pytest.mark.parametrize("fam,n", [("cartpole", 1), ("pendulum", 1), ("acrobot", 7), ("mountaincar_cont", 33)])
def test_single_copy_readback_and_pinned_action_staging(device, fam, n):
    """The scalar step path of the shim and of the mirror classes: the engine's five step outputs are
    views of ONE allocation, `read_transition` brings them to the host with one copy, `stage_scalar_action` carries a host
    action through a pinned buffer into THE device action tensor.  Same transition as the tensors `step` returns, for a
    device action tensor built the ordinary way; the views start on 256-byte boundaries like separate allocations."""
    from carl_amd import _lib
    from carl_amd.engine import VecEngine

    family = {"cartpole": _lib.CARTPOLE, "pendulum": _lib.PENDULUM, "acrobot": _lib.ACROBOT,
              "mountaincar_cont": _lib.MOUNTAINCAR_CONT}[fam]
    cls = {"cartpole": E.CARLCartPole, "pendulum": E.CARLPendulum, "acrobot": E.CARLAcrobot,
           "mountaincar_cont": E.CARLMountainCarContinuous}[fam]
    row = [float(f.default_value) for f in cls.get_context_features().values()]
    kw = dict(selector=_lib.SEL_STATIC, seed=5, auto_reset=True)
    e1, e2 = VecEngine(family, [row], n, device, **kw), VecEngine(family, [row], n, device, **kw)
    for e in (e1, e2):
        e.reset()
        assert all(t.data_ptr() % 256 == 0 for t in (e.obs, e.reward, e.terminated, e.truncated, e.done))
    rng = np.random.default_rng(3)
    for t in range(30):
        if e1.info.action_is_discrete:
            a = rng.integers(0, int(e1.info.n_actions), n)
            ref = torch.as_tensor(a.astype(np.int32), device=device)
        else:
            a = rng.uniform(float(e1.info.action_low), float(e1.info.action_high), (n, 1)).astype(np.float32)
            ref = torch.as_tensor(a, device=device)
        obs, rew, term, trunc = e1.step(ref)
        e2.step(e2.stage_scalar_action(a))
        o, r, te, tr = e2.read_transition()
        np.testing.assert_array_equal(o, obs.cpu().numpy())
        np.testing.assert_array_equal(r, rew.cpu().numpy())
        np.testing.assert_array_equal(te, term.cpu().numpy())
        np.testing.assert_array_equal(tr, trunc.cpu().numpy())
    assert torch.equal(e1.state, e2.state)
    # staging again BEFORE reading back must not tear the upload in flight (an event guards the pinned buffer), and an
    # action of the wrong element count is refused as the validated path of `step` refuses it (no NumPy broadcasting)
    a1 = (rng.integers(0, int(e1.info.n_actions), n) if e1.info.action_is_discrete
          else rng.uniform(float(e1.info.action_low), float(e1.info.action_high), (n, 1)).astype(np.float32))
    a2 = a1[::-1].copy()
    first = e2.stage_scalar_action(a1).clone()
    second = e2.stage_scalar_action(a2).clone()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(first.cpu().numpy().reshape(-1), np.asarray(a1).reshape(-1).astype(first.cpu().numpy().dtype))
    np.testing.assert_array_equal(second.cpu().numpy().reshape(-1), np.asarray(a2).reshape(-1).astype(first.cpu().numpy().dtype))
    if n > 1:
        with pytest.raises(ValueError):
            e2.stage_scalar_action(np.asarray(a1).reshape(-1)[:1])
        with pytest.raises(ValueError):
            e2.stage_scalar_action(0)
