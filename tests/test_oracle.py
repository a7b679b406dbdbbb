"""The CPU oracle against everything that pins it (CPU-only).

Pinned: Philox4x32-10 vs the Random123 known-answer vectors; the float64 known-answer
table of SURVEY.md section 8(c) (self-derived; includes the well-known CartPole first
step 0.19512195 / -0.29268293); the committed transition fixtures; the second
independent restatement (oracle/ref_style.py).  NOT pinned: agreement with gymnasium
itself -- it is not importable here and the reference's tests hold no step values.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import ref_style as R


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert O.philox4x32_10([0, 0, 0, 0], [0, 0]).tolist() == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2).tolist() == [
        0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]).tolist() == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_u01_range():
    assert O.u01(0) == 0.0
    assert O.u01(0xFFFFFFFF) == 1.0 - 2.0**-24
    assert O.u01(0x80000000) == 0.5


KAT = [
    # family, ctx overrides {col: v}, state, action, recompute, next_state, reward, terminated
    (O.CARTPOLE, {}, [0, 0, 0, 0], 1, False, [0.0, 0.1951219512195122, 0.0, -0.2926829268292683], 1.0, 0),
    (O.CARTPOLE, {}, [0.01, -0.02, 0.03, 0.04], 0, False,
     [0.0096, -0.21553901710278936, 0.0308, 0.34199522377603914], 1.0, 0),
    (O.CARTPOLE, {0: 15, 1: 2, 2: 0.3, 3: 1.2, 4: 20, 5: 0.05}, [0.01, -0.02, 0.03, 0.04], 1, False,
     [0.009, 0.9207189357303092, 0.032, -0.6561274202723141], 1.0, 0),
    (O.CARTPOLE, {0: 15, 1: 2, 2: 0.3, 3: 1.2, 4: 20, 5: 0.05}, [0.01, -0.02, 0.03, 0.04], 1, True,
     [0.009, 0.45944305673608893, 0.032, -0.24545668649337735], 1.0, 0),
    (O.PENDULUM, {}, [1.0, 0.5], 1.5, False, [1.0678051619302962, 1.3561032386059226], -1.02725, 0),
    (O.PENDULUM, {2: 4.2, 3: 0.8, 4: 1.7}, [-2.5, -3.0], -3.0, False,
     [-2.6620325576672608, -3.240651153345214], -7.154, 0),
    (O.MOUNTAINCAR, {}, [-0.5, 0], 2, False, [-0.49917684300416926, 0.0008231569958307428], -1.0, 0),
    (O.MOUNTAINCAR, {}, [0.44, 0.02], 2, False, [0.46037956137086905, 0.02037956137086907], -1.0, 1),
    (O.MOUNTAINCAR_CONT, {}, [-0.5, 0], 0.7, False, [-0.49912684300416926, 0.0008731569958307427], -0.049, 0),
    (O.ACROBOT, {}, [0.05, -0.03, 0.02, 0.08], 2, False,
     [0.0340521385, 0.0275399703, -0.1741702501, 0.4824932354], -1.0, 0),
    (O.ACROBOT, {0: 1.5, 2: 0.7, 3: 1.3, 4: 0.4, 5: 0.6, 6: 1.2}, [2.9, 0.4, 3.0, -5.0], 0, False,
     [-2.7758713949, -0.6296191167, 3.1066454206, -5.1978390093], 0.0, 1),
]


@pytest.mark.parametrize("case", KAT, ids=lambda c: f"{O.FAMILY_NAMES[c[0]]}-{c[3]}")
def test_known_answer_table(case):
    fam, over, s, a, rec, s2_want, r_want, t_want = case
    ctx = O.default_row(fam)
    for k, v in over.items():
        ctx[k] = v
    s2, obs, r, t = O.transitions(fam, [ctx], [s], [a], precision="f64", cartpole_recompute=rec)
    tol = 5e-10 if fam == O.ACROBOT else 1e-12  # acrobot rows are quoted to 10 digits
    if fam == O.MOUNTAINCAR_CONT:
        tol = 1e-7  # that env stores its state as float32
    np.testing.assert_allclose(s2[0], s2_want, rtol=0, atol=tol)
    # continuous actions are float32 (0.7f = 0.69999999), hence the looser reward tolerance
    assert r[0] == pytest.approx(r_want, abs=1e-8 if fam in O.CONTINUOUS else 1e-9)
    assert int(t[0]) == t_want


def test_pendulum_obs_known_answer():
    _, obs, _, _ = O.transitions(O.PENDULUM, [O.default_row(O.PENDULUM)], [[1.0, 0.5]], [1.5])
    np.testing.assert_allclose(obs[0], [0.48204838409011624, 0.8761445973103457, 1.3561032386059226], rtol=1e-7)


@pytest.mark.parametrize("fam", range(5), ids=O.FAMILY_NAMES)
def test_fixture_matches_oracle(fam, golden_dir):
    """the committed vectors are reproduced by the oracle (drift guard), in f64 exactly
    and in f32 within fp32 rounding of a single transition"""
    g = np.load(os.path.join(golden_dir, f"transitions_{O.FAMILY_NAMES[fam]}.npz"))
    ctx, s, a = g["ctx"].astype(np.float64), g["state"].astype(np.float64), g["action"]
    s2, obs, r, t = O.transitions(fam, ctx, s, a, precision="f64")
    np.testing.assert_array_equal(s2, g["next_state"])
    np.testing.assert_array_equal(obs, g["obs"])
    np.testing.assert_array_equal(t, g["terminated"])
    np.testing.assert_array_equal(r, g["reward"])
    s2f, obsf, rf, tf = O.transitions(fam, ctx, s, a, precision="f32")
    scale = 1.0 + np.abs(g["next_state"])
    err = np.abs(s2f - g["next_state"]) / scale
    if fam == O.ACROBOT:
        # RK4 at dt = 0.2 from |w2| ~ 9 pi passes through stage values ~1e4: plain fp32
        # cancels catastrophically on ~1 % of these (deliberately extreme) rows.  This is
        # why the HIP Acrobot kernel evaluates _dsdt in fp64 by default.
        assert np.percentile(err.max(1), 95) < 1e-5 and err.max() < 5e-3
    else:
        assert err.max() < 2e-6
    assert np.max(np.abs(rf - g["reward"]) / (1.0 + np.abs(g["reward"]))) < 2e-5
    # flags may only differ on rows sitting within fp32 rounding of a threshold
    assert np.mean(tf != g["terminated"]) < 0.01


@pytest.mark.parametrize("fam", range(5), ids=O.FAMILY_NAMES)
def test_ref_style_loop_matches_c_oracle(fam):
    """two independent restatements (scalar Python loop with the reference's wrapper
    stack vs the C engine) produce the same trajectories, resets included"""
    rng = np.random.default_rng(fam)
    n_ctx = 3
    names = O.feature_names(fam)
    table = np.tile(O.default_row(fam), (n_ctx, 1))
    if fam == O.PENDULUM:
        table[:, 2] = [9.0, 10.0, 11.5]
    elif fam == O.CARTPOLE:
        table[:, 0] = [9.8, 12.0, 7.0]
    elif fam == O.ACROBOT:
        table[:, 2] = [1.0, 0.8, 1.3]
    else:
        table[:, 2] = [0.07, 0.06, 0.08]
    table = table.astype(np.float32).astype(np.float64)
    contexts = {i: dict(zip(names, table[i].tolist())) for i in range(n_ctx)}
    T = 40 if fam != O.MOUNTAINCAR else 250  # mountaincar: run past the 200-step TimeLimit
    eng = O.Engine(fam, table, 1, selector=O.SEL_ROUND_ROBIN, autoreset=False, seed=5, precision="f64")
    env = R.RefStyleEnv(fam, contexts)
    for episode in range(3):
        eng_obs = eng.reset()
        w = O.lane_words(5, 0, episode, 0)
        u = [O.u01(x) for x in w]
        obs, info = env.reset(u=u)
        assert info["context_id"] == int(eng.ctx_idx[0]) == episode % n_ctx
        np.testing.assert_allclose(obs["obs"], eng_obs[0], rtol=1e-6, atol=1e-7)
        assert list(obs["context"].keys()) == names
        for t in range(T):
            a = R.random_action(fam, rng)
            o, r, term, trunc, info = env.step(a)
            out = eng.step(np.asarray(a).reshape(1))
            np.testing.assert_allclose(o["obs"], out.obs[0], rtol=2e-6, atol=2e-6)
            assert r == pytest.approx(float(out.reward[0]), rel=1e-6, abs=1e-6)
            assert term == bool(out.terminated[0]) and trunc == bool(out.truncated[0])
            if term or trunc:
                break


def test_cartpole_reward_after_termination():
    """gymnasium: stepping a terminated CartPole again (no reset) earns 0.0"""
    eng = O.Engine(O.CARTPOLE, [O.default_row(O.CARTPOLE)], 1, autoreset=False)
    eng.reset()
    eng.state[0] = [2.39, 3.0, 0.0, 0.0]
    out = eng.step([1])
    assert out.terminated[0] == 1 and out.reward[0] == 1.0
    out = eng.step([1])
    assert out.terminated[0] == 1 and out.reward[0] == 0.0


def test_timelimit_truncation_and_autoreset():
    eng = O.Engine(O.PENDULUM, [O.default_row(O.PENDULUM)], 4, selector=O.SEL_STATIC, max_steps=5, seed=3)
    eng.reset()
    for t in range(1, 12):
        out = eng.step(np.zeros(4, np.float32))
        assert (out.truncated == (1 if t % 5 == 0 else 0)).all()
        assert (out.terminated == 0).all()
        if t % 5 == 0:
            assert (eng.elapsed == 0).all() and np.isfinite(out.final_obs).all()
        else:
            assert (eng.elapsed == t % 5).all()
    assert (eng.episodes_done == 2).all() and (eng.n_calls == 3).all()


def test_selector_rules_match_reference_sequences(golden_dir):
    """per-lane device rule (restated in the oracle) == ids produced by RUNNING the
    reference's selectors (tests/golden/selector_sequences.json)"""
    gold = json.load(open(os.path.join(golden_dir, "selector_sequences.json")))
    for n_ctx in (1, 2, 3, 5, 7):
        table = np.tile(O.default_row(O.PENDULUM), (n_ctx, 1))
        for rule, key in ((O.SEL_ROUND_ROBIN, "round_robin"), (O.SEL_STATIC, "static")):
            want = gold[f"{key}_{n_ctx}"]
            eng = O.Engine(O.PENDULUM, table, 1, selector=rule)
            ids, calls = [], []
            for _ in want["context_id"]:
                eng.reset()
                ids.append(int(eng.ctx_idx[0]))
                calls.append(int(eng.n_calls[0]))
            assert ids == want["context_id"] and calls == want["n_calls"]


def test_done_compact_oracle():
    term = np.array([0, 1, 0, 0, 1, 0], np.uint8)
    trunc = np.array([0, 0, 0, 1, 1, 0], np.uint8)
    assert O.done_compact(term, trunc).tolist() == [1, 3, 4]
