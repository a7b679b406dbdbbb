"""Brax spring-pipeline restatement (oracle/brax_spring.c): internal consistency and physical
invariants on CPU.  brax itself is not importable here and the reference's tests hold no Brax
step value (test/test_brax_env.py:8-23 is construct/reset only), so this is NOT parity with
brax -- it pins the specification the HIP kernel is then compared against."""
import ctypes as C
import os

import numpy as np
import pytest

from carl_amd import _lib
from carl_amd.envs.brax.models import ant_sys
from oracle import brax as B
from oracle import oracle as O

NAMES = ["gravity", "friction", "elasticity", "ang_damping", "mass_torso", "viscosity", "target_distance",
         "target_direction", "target_radius"]
DEFAULT = np.array([-9.8, 1.0, 0.0, -0.05, 10.0, 0.0, 100.0, 1.0, 5.0])


@pytest.fixture(scope="module")
def ant():
    return ant_sys(NAMES)


def test_model_table_shape(ant):
    assert (ant.n_links, ant.n_q, ant.n_dof, ant.n_act, ant.obs_dim) == (9, 15, 14, 8, 27)
    assert ant.n_frames * ant.dt == pytest.approx(0.05)  # brax Ant: dt 0.005 x 10 frames on "spring"
    assert list(ant.parent[:9]) == [-1, 0, 1, 0, 3, 0, 5, 0, 7]
    assert sorted(ant.act_dof[:8]) == list(range(6, 14))
    assert ant.ctx.gravity == 0 and ant.ctx.friction == 1 and ant.ctx.n_mass == 1 and ant.ctx.mass_row[0] == 4
    compat = ant_sys(NAMES, reference_compat=True)
    assert compat.ctx.gravity == -1 and compat.ctx.n_mass == 0
    with pytest.raises(RuntimeError):
        ant_sys(NAMES + ["mass_nonexistent"])


def test_brax_struct_layout_matches_c(tmp_path):
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fields = [f[0] for f in _lib.BraxSys._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{root}/include/carl_amd.h"', "int main(void){",
           'printf("%zu %zu\\n", sizeof(carl_brax_sys_t), sizeof(carl_brax_ctx_map_t));']
    src += [f'printf("%zu\\n", offsetof(carl_brax_sys_t, {f}));' for f in fields] + ["return 0;}"]
    (tmp_path / "l.c").write_text("\n".join(src))
    subprocess.run(["gcc", "-o", str(tmp_path / "l"), str(tmp_path / "l.c")], check=True)
    out = list(map(int, subprocess.run([str(tmp_path / "l")], capture_output=True, text=True, check=True).stdout.split()))
    assert out[:2] == [C.sizeof(_lib.BraxSys), C.sizeof(_lib.BraxCtxMap)]
    assert out[2:] == [getattr(_lib.BraxSys, f).offset for f in fields]


def test_forward_inverse_kinematics_roundtrip(ant):
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = np.array(ant.init_q[:15], dtype=np.float64) + rng.uniform(-0.2, 0.2, 15)
        q[3:7] /= np.linalg.norm(q[3:7])
        qd = rng.normal(0, 1.0, 14)
        st = B.forward_kinematics(ant, q, qd)
        q2, qd2 = B.inverse_kinematics(ant, st)
        # the model table is float32 (joint-frame quaternions are unit to ~1e-7 only)
        np.testing.assert_allclose(q2, q, atol=2e-6)
        np.testing.assert_allclose(qd2, qd, atol=5e-6)
        assert np.allclose(np.linalg.norm(st[:, 3:7], axis=1), 1.0)


def test_joint_anchors_coincide_after_forward_kinematics(ant):
    """with exact kinematics the joint springs are at rest: zero torque -> only gravity acts"""
    q = np.array(ant.init_q[:15], dtype=np.float64)
    st = B.forward_kinematics(ant, q, np.zeros(14))
    st[:, 2] += 5.0  # lift off the ground: no contacts
    row = DEFAULT.copy()
    out = B.substeps(ant, row, np.zeros(14), 1, st)
    # every link accelerates with g only (joint damping is zero at rest, limits inactive except
    # the ankles resting INSIDE their range at init_q)
    np.testing.assert_allclose(out[:, 9], -9.8 * ant.dt, atol=1e-9)
    np.testing.assert_allclose(out[:, 7:9], 0.0, atol=1e-9)


def test_free_flight_conserves_momentum(ant):
    """no gravity, no contact, no damping: internal joint forces are equal and opposite"""
    rng = np.random.default_rng(1)
    q = np.array(ant.init_q[:15], dtype=np.float64)
    qd = rng.normal(0, 1.0, 14)
    st = B.forward_kinematics(ant, q, qd)
    st[:, 2] += 10.0
    row = DEFAULT.copy()
    row[0] = -1e-6  # gravity ~ 0 (feature bound is < 0)
    row[3] = 0.0    # ang_damping off
    mass = np.ones(9)
    p0 = (mass[:, None] * st[:, 7:10]).sum(0)
    tau = rng.uniform(-50, 50, 14)
    tau[:6] = 0
    out = B.substeps(ant, row, tau, 200, st)
    p1 = (mass[:, None] * out[:, 7:10]).sum(0)
    np.testing.assert_allclose(p1[:2], p0[:2], atol=1e-9)
    assert abs(p1[2] - p0[2]) < 1e-3  # 200 * dt * 9 * 1e-6
    # and the joints hold: anchors stay within millimetres
    qq, _ = B.inverse_kinematics(ant, out)
    re = B.forward_kinematics(ant, qq, np.zeros(14))
    assert np.abs(re[:, :3] - out[:, :3]).max() < 0.02


def test_zero_action_ant_stands(ant):
    e = B.Engine(ant, DEFAULT[None], 8, selector=O.SEL_STATIC, seed=3)
    obs = e.reset()
    assert obs.shape == (8, 27) and (np.abs(obs[:, 0] - 0.55) < 0.11).all()
    for _ in range(200):
        out = e.step(np.zeros((8, 8), np.float32))
    assert not out.terminated.any() and (out.obs[:, 0] > 0.35).all() and (out.obs[:, 0] < 0.7).all()
    assert np.abs(out.obs[:, 13:]).max() < 0.5  # came to rest
    assert np.allclose(out.reward, 1.0, atol=0.1)  # healthy reward only


def test_contexts_change_the_dynamics(ant):
    rows = np.tile(DEFAULT, (4, 1))
    rows[1, 0] = -20.0   # stronger gravity
    rows[2, 4] = 40.0    # heavier torso
    rows[3, 1] = 0.05    # slippery floor
    e = B.Engine(ant, rows, 4, selector=O.SEL_STATIC, seed=9, ctx_idx0=np.arange(4))
    e.reset()
    e.state[1:] = e.state[0]  # same initial state, different contexts
    rng = np.random.default_rng(0)
    for _ in range(60):
        a = np.tile(rng.uniform(-1, 1, (1, 8)).astype(np.float32), (4, 1))
        out = e.step(a)
    for k in (1, 2, 3):
        assert np.abs(out.obs[k] - out.obs[0]).max() > 1e-2
    compat = ant_sys(NAMES, reference_compat=True)
    e2 = B.Engine(compat, rows, 4, selector=O.SEL_STATIC, seed=9, ctx_idx0=np.arange(4))
    e2.reset()
    e2.state[1:] = e2.state[0]
    for _ in range(20):
        out2 = e2.step(np.zeros((4, 8), np.float32))
    assert np.abs(out2.obs[1:] - out2.obs[0]).max() == 0.0  # reference_compat: contexts are observed-only


def test_random_policy_is_stable_and_episodes_end(ant):
    rng = np.random.default_rng(5)
    n = 32
    e = B.Engine(ant, DEFAULT[None], n, selector=O.SEL_STATIC, seed=1, max_steps=50)
    e.reset()
    for t in range(120):
        out = e.step(rng.uniform(-1, 1, (n, 8)).astype(np.float32))
        assert np.isfinite(out.obs).all() and np.abs(out.obs).max() < 50
        if (t + 1) % 50 == 0:
            assert (out.truncated | out.terminated).all()
    assert (e.episodes_done >= 2).all()


def test_reset_is_a_function_of_seed_lane_episode(ant):
    a = B.Engine(ant, DEFAULT[None], 6, seed=4, lane_offset=100)
    b = B.Engine(ant, DEFAULT[None], 3, seed=4, lane_offset=103)
    a.reset()
    b.reset()
    np.testing.assert_array_equal(a.state[3:], b.state)
    q, qd = B.inverse_kinematics(ant, a.state[0])
    assert np.abs(q[:3] - np.array(ant.init_q[:3])).max() <= 0.1 + 1e-9 and np.abs(qd).max() < 0.6


# ------------------------------------------------------------------ Halfcheetah
from carl_amd.envs.brax.models import halfcheetah_sys  # noqa: E402

HC_NAMES = ["gravity", "friction", "elasticity", "ang_damping", "viscosity", "mass_torso", "mass_bthigh",
            "mass_bshin", "mass_bfoot", "mass_fthigh", "mass_fshin", "mass_ffoot", "target_distance",
            "target_direction", "target_radius", "joint_stiffness"]
HC_DEFAULT = np.array([-9.8, 1.0, 0.0, -0.05, 0.0, 10.0, 1.5435146, 1.5874476, 1.0953975, 1.4380753, 1.2008368,
                       0.8845188, 100.0, 1.0, 5.0, 1.0])


@pytest.fixture(scope="module")
def cheetah():
    return halfcheetah_sys(HC_NAMES)


def test_halfcheetah_table_and_kinematics(cheetah):
    s = cheetah
    assert (s.n_links, s.n_q, s.n_dof, s.n_act, s.obs_dim) == (7, 9, 9, 6, 17)
    assert s.n_frames * s.dt == pytest.approx(0.05)
    assert list(s.parent[:7]) == [-1, 0, 1, 2, 0, 4, 5] and list(s.n_slide[:7]) == [2, 0, 0, 0, 0, 0, 0]
    assert s.ctx.n_mass == 7 and s.ctx.joint_stiffness_scale == 15
    rng = np.random.default_rng(0)
    for _ in range(10):
        q = rng.uniform(-0.3, 0.3, 9)
        qd = rng.normal(0, 1, 9)
        st = B.forward_kinematics(s, q, qd)
        q2, qd2 = B.inverse_kinematics(s, st)
        np.testing.assert_allclose(q2, q, atol=2e-6)
        np.testing.assert_allclose(qd2, qd, atol=5e-6)
        assert np.abs(st[:, 1]).max() < 1e-9  # planar: y = 0


def test_halfcheetah_stays_planar_and_stable(cheetah):
    rng = np.random.default_rng(2)
    n = 16
    e = B.Engine(cheetah, HC_DEFAULT[None], n, selector=O.SEL_STATIC, seed=2)
    obs = e.reset()
    assert obs.shape == (n, 17)
    for t in range(150):
        out = e.step(rng.uniform(-1, 1, (n, 6)).astype(np.float32))
        assert np.isfinite(out.obs).all() and np.abs(out.obs).max() < 60
        assert not out.terminated.any()  # Halfcheetah never terminates
    st = e.state.reshape(n, 7, 13)
    assert np.abs(st[:, :, 1]).max() < 1e-6      # y
    assert np.abs(st[:, :, 4]).max() < 1e-6 and np.abs(st[:, :, 6]).max() < 1e-6  # roll / yaw quaternion parts


def test_halfcheetah_joint_stiffness_context_acts(cheetah):
    rows = np.tile(HC_DEFAULT, (2, 1))
    rows[1, 15] = 0.2  # softer constraint springs: joints separate more under load
    e = B.Engine(cheetah, rows, 2, selector=O.SEL_STATIC, seed=3, ctx_idx0=np.arange(2))
    e.reset()
    e.state[1] = e.state[0]
    rng = np.random.default_rng(0)
    for _ in range(40):
        a = np.tile(rng.uniform(-1, 1, (1, 6)).astype(np.float32), (2, 1))
        out = e.step(a)
    assert np.abs(out.obs[1] - out.obs[0]).max() > 1e-3


# ------------------------------------------------------------------ Humanoid
from carl_amd.envs.brax.models import HUMANOID_MASSES, humanoid_sys  # noqa: E402

HU_NAMES = (["gravity", "friction", "elasticity", "ang_damping", "viscosity"] + list(HUMANOID_MASSES) +
            ["target_distance", "target_direction", "target_radius"])
HU_DEFAULT = np.array([-9.8, 1.0, 0.0, 0.0, 0.0] + list(HUMANOID_MASSES.values()) + [100.0, 1.0, 5.0])


@pytest.fixture(scope="module")
def humanoid():
    return humanoid_sys(HU_NAMES)


def test_humanoid_table_and_kinematics(humanoid):
    """brax Humanoid shape: q 24, qd 23, 17 motors, 244-dim obs; 2- and 3-dof joints round-trip
    through forward / inverse kinematics (Euler x-y-+-z decomposition)."""
    s = humanoid
    assert (s.n_links, s.n_q, s.n_dof, s.n_act, s.obs_dim) == (11, 24, 23, 17, 244)
    assert s.n_frames * s.dt == pytest.approx(0.015)
    assert list(s.parent[:11]) == [-1, 0, 1, 2, 3, 2, 5, 0, 7, 0, 9]
    assert list(s.n_link_dof[:11]) == [6, 2, 1, 3, 1, 3, 1, 2, 1, 2, 1]
    assert s.dof_sign3[3] == -1.0 and s.dof_sign3[5] == -1.0  # hips: x, z, y = x cross z reversed
    assert s.ctx.n_mass == 11 and list(s.ctx.mass_link[:11]) == list(range(11))
    assert sorted(s.act_dof[:17]) == list(range(6, 23))
    rng = np.random.default_rng(0)
    for _ in range(30):
        q = np.array(s.init_q[:24], dtype=np.float64)
        q[:7] += rng.uniform(-0.2, 0.2, 7)
        q[7:] += rng.uniform(-0.6, 0.6, 17)
        q[3:7] /= np.linalg.norm(q[3:7])
        qd = rng.normal(0, 1.0, 23)
        st = B.forward_kinematics(s, q, qd)
        q2, qd2 = B.inverse_kinematics(s, st)
        np.testing.assert_allclose(q2, q, atol=5e-6)
        np.testing.assert_allclose(qd2, qd, atol=1e-5)


def test_humanoid_consistent_state_has_no_constraint_force(humanoid):
    """At a kinematically consistent pose with zero velocity, zero gravity and in-range joints, the
    joint springs see zero anchor / alignment error: only the dof springs (stiffness * angle) act,
    so with those angles at zero the state is a fixed point of the substep."""
    s = humanoid
    q = np.array(s.init_q[:24], dtype=np.float64)
    q[13] = q[17] = -0.5  # knees inside their (-160, -2) deg range (no dof spring of their own: the asset gives none)
    q[12] = q[16] = -0.3  # hip y: dof stiffness 20 -- the only spring that acts
    st = B.forward_kinematics(s, q, np.zeros(23))
    row = HU_DEFAULT.copy()
    row[0] = -1e-6
    st2 = B.substeps(s, row, np.zeros(23), 1, st.copy())
    dv = np.abs(st2.reshape(11, 13)[:, 7:] - st.reshape(11, 13)[:, 7:])
    others = [i for i in range(11) if i not in (2, 3, 4, 5, 6)]  # pelvis / thighs (and the shins hanging on them) feel the hip springs
    assert dv[others].max() < 1e-6
    assert dv[[3, 5]].max() > 1e-5


def test_humanoid_stands_then_random_actions_stay_finite(humanoid):
    rng = np.random.default_rng(5)
    n = 8
    e = B.Engine(humanoid, HU_DEFAULT[None], n, selector=O.SEL_STATIC, seed=4)
    obs = e.reset()
    assert obs.shape == (n, 244)
    np.testing.assert_allclose(obs[:, 0], 1.4, atol=0.011)            # torso z = init + U(+-0.01)
    assert np.abs(obs[:, 22:45]).max() <= 0.0100001                    # qd ~ U(-0.01, 0.01)
    assert np.all(obs[:, -23:] == 0)                                   # qfrc_actuator of the reset obs
    np.testing.assert_allclose(obs[:, 45 + 9], 1.0)                    # com_inertia block: [.., mass] per link
    total = 0.0
    for t in range(40):
        a = rng.uniform(-0.4, 0.4, (n, 17)).astype(np.float32)
        out = e.step(a)
        assert np.isfinite(out.obs).all() and np.abs(out.obs).max() < 400
        total += out.reward.mean()
        # qfrc_actuator = gear * clipped action scattered to the dofs
        frc = out.final_obs[:, -23:] if out.terminated.any() else out.obs[:, -23:]
        live = ~(out.terminated | out.truncated).astype(bool)
        want = np.zeros((n, 23))
        for k in range(17):
            want[:, humanoid.act_dof[k]] += humanoid.act_gear[k] * a[:, k]
        np.testing.assert_allclose(frc[live], want[live], rtol=1e-6, atol=1e-5)
    assert total / 40 > 3.0  # healthy reward 5 dominates while upright


def test_humanoid_mass_context_changes_com_terms(humanoid):
    rows = np.tile(HU_DEFAULT, (2, 1))
    rows[1, HU_NAMES.index("mass_torso")] = 20.0
    e = B.Engine(humanoid, rows, 2, selector=O.SEL_STATIC, seed=3, ctx_idx0=np.arange(2))
    obs = e.reset()
    assert obs[0, 45 + 9] == pytest.approx(1.0) and obs[1, 45 + 9] == pytest.approx(2.0)  # effective mass, link 0
    assert obs[1, 45 + 19] == pytest.approx(1.0)                                           # link 1 untouched


# ------------------------------------------------------------------ Hopper / Walker2d / InvertedPendulum
from carl_amd.envs.brax.models import hopper_sys, inverted_pendulum_sys, walker2d_sys  # noqa: E402


def _features(cls_name):
    import importlib

    cls = getattr(importlib.import_module("carl_amd.envs.brax"), cls_name)
    feats = cls.get_context_features()
    return list(feats), np.array([float(f.default_value) for f in feats.values()])


@pytest.mark.parametrize("cls_name,fn,shape", [
    ("CARLBraxHopper", hopper_sys, (4, 6, 6, 3, 11)),
    ("CARLBraxWalker2d", walker2d_sys, (7, 9, 9, 6, 17)),
    ("CARLBraxInvertedPendulum", inverted_pendulum_sys, (2, 2, 2, 1, 4)),
])
def test_planar_family_tables_and_kinematics(cls_name, fn, shape):
    names, default = _features(cls_name)
    s = fn(names)
    assert (s.n_links, s.n_q, s.n_dof, s.n_act, s.obs_dim) == shape
    assert s.ctx.n_mass == sum(n.startswith("mass_") for n in names)
    rng = np.random.default_rng(1)
    for _ in range(10):
        q = np.array(s.init_q[:s.n_q], dtype=np.float64) + rng.uniform(-0.15, 0.15, s.n_q)
        qd = rng.normal(0, 1, s.n_dof)
        st = B.forward_kinematics(s, q, qd)
        q2, qd2 = B.inverse_kinematics(s, st)
        np.testing.assert_allclose(q2, q, atol=2e-6)
        np.testing.assert_allclose(qd2, qd, atol=5e-6)
        assert np.abs(st.reshape(s.n_links, 13)[:, 1]).max() < 1e-9  # planar: y = 0


def test_hopper_health_rule_and_velocity_clip():
    names, default = _features("CARLBraxHopper")
    s = hopper_sys(names)
    n = 4
    e = B.Engine(s, default[None], n, selector=O.SEL_STATIC, seed=3)
    obs = e.reset()
    np.testing.assert_allclose(obs[:, 0], 1.25, atol=6e-3)  # rootz carries ref = 1.25
    # pitch the torso beyond 0.2 rad: the step must terminate although z is still healthy
    st = e.state.reshape(n, 4, 13).copy()
    q = np.array(s.init_q[:6], dtype=np.float64)
    q[2] = 0.25
    qd = np.zeros(6)
    qd[0] = 50.0  # and a forward velocity beyond the +-10 clip of the observation
    e.state[0] = B.forward_kinematics(s, q, qd).reshape(-1)
    out = e.step(np.zeros((n, 3), np.float32))
    assert out.terminated[0] == 1 and not out.terminated[1:].any()
    assert out.final_obs[0, 0] > 0.7 and abs(out.final_obs[0, 1]) > 0.2
    assert out.final_obs[0, 5] == 10.0  # clipped qd[0]
    assert out.reward[0] > 40  # the reward uses the unclipped forward velocity


def test_inverted_pendulum_rules():
    names, default = _features("CARLBraxInvertedPendulum")
    s = inverted_pendulum_sys(names)
    n = 3
    e = B.Engine(s, default[None], n, selector=O.SEL_STATIC, seed=5)
    obs = e.reset()
    assert obs.shape == (n, 4) and np.abs(obs).max() <= 0.0100001
    # constant push: the slider's +-1 range holds the cart (limit spring), reward is 1 per step
    total, done_at = 0, None
    for t in range(120):
        out = e.step(np.full((n, 1), 3.0, np.float32))
        assert np.all(out.reward == 1.0) and np.isfinite(out.obs).all()
        x = np.where(out.terminated[:, None] != 0, out.final_obs, out.obs)[:, 0]
        assert np.all(x < 1.6)
        if out.terminated.any() and done_at is None:
            done_at = t
            ang = out.final_obs[out.terminated != 0, 1]
            assert np.all(np.abs(ang) > 0.2)
    assert done_at is not None and done_at < 60  # pushing the cart tips the pole over
    # the cart is carried by its joint against gravity: z stays 0, no rotation
    st = e.state.reshape(n, 2, 13)
    assert np.abs(st[:, 0, 2]).max() < 5e-3 and np.abs(st[:, 0, 4:7]).max() < 5e-3


def test_reference_recorded_ant_first_step_is_typical_for_this_restatement(golden_dir):
    """The ONE Brax transition the reference records (examples/brax_with_goals.ipynb cell 3: obs and
    reward of CARLBraxAnt after reset + one random step; fixture made by
    tests/golden/make_notebook_brax_golden.py).  Action and PRNG state are unknown, so this cannot
    pin the arithmetic -- it checks that the recorded vector is an ordinary draw of OUR reset + one
    step under random actions, dimension by dimension: observation layout (z, quaternion, joint
    angles, velocities), initial pose, reset-noise scale, velocity scale after one control step, and
    that the goal wrapper's reward 0 (no progress) is a common outcome."""
    import json
    import os

    g = json.load(open(os.path.join(golden_dir, "notebook_brax_ant_first_step.json")))
    names = list(g["context"])
    row = np.array([g["context"][k] for k in names], dtype=np.float64)
    s = ant_sys(names)
    s.goal_mode = 1
    n = 4096
    e = B.Engine(s, row[None], n, selector=O.SEL_STATIC, seed=0)
    e.reset()
    out = e.step(np.random.default_rng(0).uniform(-1, 1, (n, 8)).astype(np.float32))
    ref = np.array(g["obs"])
    assert ref.shape == (27,) == out.obs.shape[1:]
    z = (ref - out.obs.mean(0)) / out.obs.std(0)
    assert np.abs(z).max() < 3.0, z
    assert np.all(ref >= out.obs.min(0)) and np.all(ref <= out.obs.max(0))
    assert g["reward"] == 0.0 and 0.2 < (out.reward == 0).mean() < 0.8


def test_humanoidstandup_lies_down_and_rewards_height():
    from carl_amd.envs.brax.models import humanoidstandup_sys

    names, default = _features("CARLBraxHumanoidStandup")
    assert "target_distance" not in names and len(names) == 16
    s = humanoidstandup_sys(names)
    n = 4
    e = B.Engine(s, default[None], n, selector=O.SEL_STATIC, seed=0)
    obs = e.reset()
    assert obs.shape == (n, 244) and np.all(np.abs(obs[:, 0] - 0.105) <= 0.0101)
    np.testing.assert_allclose(np.abs(obs[:, 1]), np.sqrt(0.5), atol=0.02)  # root rotated -90 deg about y
    rng = np.random.default_rng(0)
    for t in range(50):
        a = rng.uniform(-0.4, 0.4, (n, 17)).astype(np.float32)
        out = e.step(a)
        assert not out.terminated.any() and np.isfinite(out.obs).all()
        z = out.obs[:, 0].astype(np.float64)
        want = z / 0.015 + 1.0 - 0.1 * (a.astype(np.float64) ** 2).sum(1)  # uph_cost + 1 - quad_ctrl_cost
        np.testing.assert_allclose(out.reward, want, rtol=1e-5, atol=1e-4)
    assert np.all(out.obs[:, 0] < 0.3)  # random torques do not stand it up


def test_inverted_double_pendulum_observation_and_tip_reward():
    from carl_amd.envs.brax.models import inverted_double_pendulum_sys

    names, default = _features("CARLBraxInvertedDoublePendulum")
    assert names == ["gravity", "friction", "elasticity", "mass_cart", "mass_pole", "mass_pole2", "ang_damping",
                     "viscosity"]
    s = inverted_double_pendulum_sys(names)
    assert (s.n_links, s.n_q, s.n_dof, s.n_act, s.obs_dim) == (3, 3, 3, 1, 8) and s.ctx.n_mass == 3
    n = 6
    e = B.Engine(s, default[None], n, selector=O.SEL_STATIC, seed=0)
    obs = e.reset()
    # cart x, sin(q1), sin(q2), cos(q1), cos(q2), qd: upright + U(+-0.01) noise
    assert np.abs(obs[:, 0:3]).max() <= 0.0101 and np.all(obs[:, 3:5] > 0.9999)
    np.testing.assert_allclose(obs[:, 1] ** 2 + obs[:, 3] ** 2, 1.0, atol=1e-6)
    rng = np.random.default_rng(0)
    seen_done = False
    for t in range(120):
        out = e.step(rng.uniform(-1, 1, (n, 1)).astype(np.float32))
        live = out.terminated == 0
        # live envs: the true tip from the stored state of link 2 (COM p, rotation r): origin + R (0, 0, 0.6)
        st = e.state.reshape(n, 3, 13)[live, 2]
        w, x_, y_, z_ = st[:, 3], st[:, 4], st[:, 5], st[:, 6]
        zx = 2 * (x_ * z_ + w * y_)                 # R e_z, x component
        zz = 1 - 2 * (x_ * x_ + y_ * y_)           # R e_z, z component
        tip_x = st[:, 0] + zx * (0.6 - 0.3)        # COM is at 0.3 along the pole
        tip_z = st[:, 2] + zz * (0.6 - 0.3)
        o = out.obs[live].astype(np.float64)
        unclipped = np.abs(o[:, 6:8]).max(1) < 10.0  # the penalty uses the unclipped rates
        want = 10.0 - (0.01 * tip_x**2 + (tip_z - 2.0) ** 2) - (1e-3 * o[:, 6] ** 2 + 5e-3 * o[:, 7] ** 2)
        np.testing.assert_allclose(out.reward[live][unclipped], want[unclipped], rtol=1e-5, atol=2e-4)
        assert np.all(tip_z > 1.0)
        # finished envs: the terminal observation shows a tip near or below the threshold (rigid-geometry
        # estimate from its own sines / cosines; the spring joints stretch by centimetres under the motor)
        f = out.final_obs[~live].astype(np.float64)
        zf = 0.6 * f[:, 3] + 0.6 * (f[:, 3] * f[:, 4] - f[:, 1] * f[:, 2])
        assert np.all(zf < 1.2)
        seen_done |= bool(out.terminated.any())
    assert seen_done and not out.truncated.any()


def test_reacher_goal_draw_observation_and_reward():
    """reach task (carl_brax_sys_t::target_link): the goal is drawn inside the 0.2 disc and stays put,
    the observation is cos ++ sin ++ goal ++ arm rates ++ (fingertip - goal) with the fingertip on the
    rigid-arm circle, reward = -distance - |a|^2, no termination before the time limit."""
    from carl_amd.envs.brax.models import reacher_sys

    names, default = _features("CARLBraxReacher")
    assert names == ["gravity", "friction", "elasticity", "ang_damping", "viscosity", "mass_body0", "mass_body1"]
    s = reacher_sys(names)
    assert (s.n_links, s.n_q, s.n_dof, s.n_act, s.obs_dim) == (3, 4, 4, 2, 11) and s.ctx.n_mass == 2
    n = 64
    e = B.Engine(s, default[None], n, selector=O.SEL_STATIC, seed=3)
    obs = e.reset().astype(np.float64)
    goal0 = obs[:, 4:6].copy()
    r = np.hypot(goal0[:, 0], goal0[:, 1])
    assert r.max() <= 0.2 and r.min() >= 0.0 and r.std() > 0.03          # uniform distance
    assert np.ptp(np.arctan2(goal0[:, 1], goal0[:, 0])) > 5.0             # all bearings
    np.testing.assert_allclose(obs[:, 0:2] ** 2 + obs[:, 2:4] ** 2, 1.0, atol=1e-6)
    th = np.arctan2(obs[:, 2:4], obs[:, 0:2])
    assert np.abs(th).max() <= 0.1001 and np.abs(obs[:, 6:8]).max() <= 0.0051
    # fingertip of the rigid arm: 0.1 (cos t0, sin t0) + 0.11 (cos(t0+t1), sin(t0+t1))
    tip = 0.1 * np.stack([np.cos(th[:, 0]), np.sin(th[:, 0])], 1) + \
        0.11 * np.stack([np.cos(th.sum(1)), np.sin(th.sum(1))], 1)
    np.testing.assert_allclose(obs[:, 8:10], tip - goal0, atol=1e-6)
    np.testing.assert_allclose(obs[:, 10], 0.0, atol=1e-6)                # arm and goal both at z = 0.01
    rng = np.random.default_rng(0)
    for t in range(150):
        a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
        out = e.step(a)
        o = out.obs.astype(np.float64)
        assert np.isfinite(o).all() and not out.terminated.any() and not out.truncated.any()
        np.testing.assert_allclose(o[:, 4:6], goal0, atol=2e-4)           # the marker does not move
        want = -np.linalg.norm(o[:, 8:11], axis=1) - (a.astype(np.float64) ** 2).sum(1)
        np.testing.assert_allclose(out.reward, want, rtol=1e-5, atol=1e-5)
        th = np.arctan2(o[:, 2:4], o[:, 0:2])
        tip = 0.1 * np.stack([np.cos(th[:, 0]), np.sin(th[:, 0])], 1) + \
            0.11 * np.stack([np.cos(th.sum(1)), np.sin(th.sum(1))], 1)
        # spring joints stretch by millimetres under the motors
        np.testing.assert_allclose(o[:, 8:10], tip - o[:, 4:6], atol=5e-3)
        assert np.abs(o[:, 10]).max() < 5e-3                              # gravity sag of the springs
    assert np.abs(th[:, 1]).max() <= 3.05                                  # elbow range +-3 rad
    assert np.abs(o[:, 6:8]).max() < 40.0                                  # gear 25 against damping 1


def pusher_contact_state(s, n_lanes, pan_rate=1.0):
    """[n_lanes, 13 * 8] states with the arm lowered to table height (shoulder lift 0.3909 rad, elbow
    straight: wrist at x = 0.767, z = -0.275) and swinging about the shoulder pan; even lanes hold the
    puck inside the fork (0.09 ahead of the cross bar, touching nothing yet), odd lanes 0.3 away."""
    q = np.zeros(9)
    q[1] = 0.3909
    qd = np.zeros(9)
    qd[0] = pan_rate
    out = np.zeros((n_lanes, 13 * 8))
    wrist_x = 0.1 + 0.721 * np.cos(0.3909)
    for i in range(n_lanes):
        qq = q.copy()
        px = wrist_x + (0.09 if i % 2 == 0 else 0.09 - 0.3)
        qq[7], qq[8] = px - 0.45, -0.6 - (-0.05)  # slides are offsets from the puck's MJCF position
        out[i] = B.forward_kinematics(s, qq, qd).reshape(-1)
    return out


def test_pusher_reset_observation_reward_and_gripper_contact():
    """push task (carl_brax_sys_t::push_link / n_pair): puck drawn in its box and kept 0.17 from the
    goal, observation arm q ++ arm qd ++ gripper / puck / goal positions, reward = -|puck - goal| -
    0.1 |a|^2 - 0.5 |puck - gripper|, and the fork pushes the puck it sweeps into (and only that one)."""
    from carl_amd.envs.brax.models import PUSHER_MASSES, pusher_sys

    names, default = _features("CARLBraxPusher")
    assert names[5:13] == list(PUSHER_MASSES) and names[13:] == ["goal_position_x", "goal_position_y", "goal_position_z"]
    s = pusher_sys(names)
    assert (s.n_links, s.n_q, s.n_dof, s.n_act, s.obs_dim) == (8, 9, 9, 7, 23) and s.ctx.n_mass == 8
    n = 128
    rows = np.tile(default, (n, 1))
    rows[:, names.index("gravity")] = -1e-6          # the MJCF has no gravity; the feature's upper bound
    rng = np.random.default_rng(5)
    rows[:, 13] = rng.uniform(0.2, 0.6, n)           # goals across the puck's box: some draws fall inside
    rows[:, 14] = rng.uniform(-0.2, 0.1, n)          # the 0.17 disc and are pushed out
    rows = rows.astype(np.float32).astype(np.float64)
    e = B.Engine(s, rows, n, selector=O.SEL_STATIC, seed=1, ctx_idx0=np.arange(n))
    obs = e.reset().astype(np.float64)
    np.testing.assert_array_equal(obs[:, 0:7], 0.0)
    assert np.abs(obs[:, 7:14]).max() <= 0.00501
    np.testing.assert_allclose(obs[:, 20:23], rows[:, 13:16], rtol=1e-6)
    puck, goal = obs[:, 17:20], obs[:, 20:23]
    np.testing.assert_allclose(puck[:, 2], -0.275, atol=1e-6)
    d = np.hypot(puck[:, 0] - goal[:, 0], puck[:, 1] - goal[:, 1])
    assert d.min() >= 0.17 - 1e-6 and np.isclose(d, 0.17, atol=1e-6).sum() >= 5
    free = d > 0.17 + 1e-6                            # untouched draws lie in the MJCF box
    assert np.all((puck[free, 0] >= 0.15 - 1e-6) & (puck[free, 0] <= 0.45 + 1e-6))
    assert np.all((puck[free, 1] >= -0.25 - 1e-6) & (puck[free, 1] <= 0.15 + 1e-6))
    # gripper link COM at init_q: wrist origin (0.821, -0.6, 0) + R com (0.03, 0, 0)
    np.testing.assert_allclose(obs[:, 14:17] - np.array([0.851, -0.6, 0.0]), 0.0, atol=1e-6)
    for t in range(20):
        a = rng.uniform(-2, 2, (n, 7)).astype(np.float32)
        out = e.step(a)
        o = out.obs.astype(np.float64)
        assert np.isfinite(o).all() and not out.terminated.any() and not out.truncated.any()
        want = (-np.linalg.norm(o[:, 17:20] - o[:, 20:23], axis=1) - 0.1 * (a.astype(np.float64) ** 2).sum(1)
                - 0.5 * np.linalg.norm(o[:, 17:20] - o[:, 14:17], axis=1))
        np.testing.assert_allclose(out.reward, want, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(o[:, 17:20], puck, atol=1e-4)  # nobody reaches the puck in one second
    # ---- the fork sweeps into the puck
    m = 8
    e2 = B.Engine(s, rows[:m], m, selector=O.SEL_STATIC, seed=1, ctx_idx0=np.arange(m))
    e2.reset()
    e2.state[:] = pusher_contact_state(s, m)
    p0 = e2.state.reshape(m, 8, 13)[:, 7, 0:3].copy()
    a = np.zeros((m, 7), np.float32)
    a[:, 0] = 2.0                                     # keep swinging
    for t in range(8):
        out = e2.step(a)
    p1 = e2.state.reshape(m, 8, 13)[:, 7, 0:3]
    grip = e2.state.reshape(m, 8, 13)[:, 6, 0:3]
    assert np.all(p1[0::2, 1] - p0[0::2, 1] > 0.08)                  # pushed along the sweep (+y)
    np.testing.assert_allclose(p1[1::2], p0[1::2], atol=1e-5)          # the far puck is not
    assert np.all(np.linalg.norm(p1[0::2, :2] - grip[0::2, :2], axis=1) < 0.2)  # and it stays in the fork
    np.testing.assert_allclose(p1[:, 2], -0.275, atol=1e-4)
    assert np.isfinite(out.obs).all()


def test_goal_epilogue_against_the_reference_wrapper_run(golden_dir):
    """The goal-reward epilogue (a14) against sequences produced by RUNNING the reference's
    BraxWalkerGoalWrapper.reset / .step (tests/golden/make_goal_wrapper_golden.py): observation indices
    of every walker family, the 16 compass directions, position integration, progress reward,
    success / termination inside the radius."""
    import json

    from carl_amd.envs.brax import models

    g = json.load(open(os.path.join(golden_dir, "goal_wrapper_sequences.json")))
    assert set(g["STATE_INDICES"]) == {"ant", "humanoid", "halfcheetah", "hopper", "walker2d"}
    saw_success = 0
    for case in g["cases"]:
        names, default = _features({"ant": "CARLBraxAnt", "humanoid": "CARLBraxHumanoid", "halfcheetah": "CARLBraxHalfcheetah",
                                    "hopper": "CARLBraxHopper", "walker2d": "CARLBraxWalker2d"}[case["env_name"]])
        s = models.SYSTEMS[case["env_name"]](names)
        assert [s.goal_obs_idx[0], s.goal_obs_idx[1]] == g["STATE_INDICES"][case["env_name"]] == case["obs_indices"]
        s.goal_mode, s.goal_dt = 1, case["dt"]
        row = default.copy()
        row[names.index("target_direction")] = case["target_direction"]
        row[names.index("target_distance")] = case["target_distance"]
        row[names.index("target_radius")] = case["target_radius"]
        row = row.astype(np.float32).astype(np.float64)  # the context table is float32 on the device
        pos = np.zeros(2)
        for t, (vx, vy) in enumerate(case["velocities"]):
            obs = np.zeros(s.obs_dim, np.float32)
            obs[s.goal_obs_idx[0]], obs[s.goal_obs_idx[1]] = vx, vy
            r, ok = B.goal_step(s, row, obs, pos)
            assert ok == case["success"][t] == int(case["terminated"][t])
            np.testing.assert_allclose(r, case["reward"][t], rtol=1e-5, atol=2e-7)
            np.testing.assert_allclose(pos, case["position"][t], rtol=1e-6, atol=1e-9)
            saw_success += ok
        d = np.asarray(g["direction_values"][str(case["target_direction"])]) * np.float32(case["target_distance"])
        np.testing.assert_allclose(d, case["goal_position"], rtol=1e-6)
    assert saw_success > 20


@pytest.mark.parametrize("model", ["ant", "halfcheetah", "humanoid", "hopper", "walker2d", "inverted_pendulum",
                                   "inverted_double_pendulum", "humanoidstandup", "reacher", "pusher"])
def test_joint_wrenches_agree_with_the_independent_numpy_restatement(model):
    """oracle/spring_ref.py: the joint pass written a second time, from the specification, with rotation matrices
    instead of quaternions (1-, 2-, 3-dof and locked joints, slides, limits, actuator torques, parent reactions).
    Random perturbed states of every model: both restatements give the same net force / torque per link."""
    from carl_amd import envs as E
    from carl_amd.envs.brax.models import SYSTEMS
    from oracle import spring_ref as R

    cls = {"ant": E.CARLBraxAnt, "halfcheetah": E.CARLBraxHalfcheetahStiffness, "humanoid": E.CARLBraxHumanoidStiffness,
           "hopper": E.CARLBraxHopper, "walker2d": E.CARLBraxWalker2d, "inverted_pendulum": E.CARLBraxInvertedPendulum,
           "inverted_double_pendulum": E.CARLBraxInvertedDoublePendulum, "humanoidstandup": E.CARLBraxHumanoidStandup,
           "reacher": E.CARLBraxReacher, "pusher": E.CARLBraxPusher}[model]
    feats = cls.get_context_features()
    names = list(feats)
    row = np.array([float(f.default_value) for f in feats.values()])
    scale = 1.0
    if "joint_stiffness" in names:
        scale = 1.7
        row[names.index("joint_stiffness")] = scale
    s = SYSTEMS[cls.env_name](names)
    rng = np.random.default_rng(sum(map(ord, model)))
    worst = 0.0
    for trial in range(12):
        # a consistent pose with joints pushed around (also beyond their ranges), then every body knocked off it
        q = np.array(s.init_q[: s.n_q], dtype=np.float64) + rng.uniform(-0.6, 0.6, s.n_q)
        if s.parent[0] < 0 and s.n_link_dof[0] == 6:
            q[3:7] = rng.normal(size=4)
            q[3:7] /= np.linalg.norm(q[3:7])
        qd = rng.normal(0, 2.0, s.n_dof)
        st = B.forward_kinematics(s, q, qd)
        st[:, 0:3] += rng.normal(0, 0.01, (s.n_links, 3))
        dq = rng.normal(0, 0.02, (s.n_links, 4))
        st[:, 3:7] += dq
        st[:, 3:7] /= np.linalg.norm(st[:, 3:7], axis=1, keepdims=True)
        st[:, 7:13] += rng.normal(0, 0.5, (s.n_links, 6))
        if s.n_pair > 0:
            st[s.push_link, 0:2] += 5.0  # the puck far from the gripper: no pair contact (not part of the joint pass)
        tau = rng.normal(0, 5.0, s.n_dof)
        F_c, T_c = B.joint_wrenches(s, row, tau, st)
        F_n, T_n = R.joint_wrenches(s, st, tau, stiffness_scale=float(np.float32(scale)))
        size = 1.0 + max(np.abs(F_c).max(), np.abs(T_c).max())
        worst = max(worst, np.abs(F_c - F_n).max() / size, np.abs(T_c - T_n).max() / size)
        assert np.abs(F_c).max() > 1.0  # a real load
    # the table's float32 quaternions are unit to ~3e-8 only; the two restatements treat that differently
    # (quaternion products as stored vs normalised bases): agreement to that level, relative to the wrench scale
    assert worst < 2e-6, worst


@pytest.mark.parametrize("fam", ["ant", "halfcheetah", "humanoid"])
def test_committed_brax_transitions_are_reproduced_by_the_oracle(fam, golden_dir):
    """tests/golden/transitions_brax_<family>.npz (made by make_brax_transition_golden.py FROM this oracle: self-derived,
    not brax output) pin the restatement against drift: one env step from each committed (context, state, action) row."""
    from carl_amd.envs.brax.models import SYSTEMS

    g = np.load(os.path.join(golden_dir, f"transitions_brax_{fam}.npz"))
    names = [str(x) for x in g["names"]]
    s = SYSTEMS[fam](names)
    rows = g["ctx"].astype(np.float64)
    n = len(rows)
    eng = B.Engine(s, rows, n, selector=O.SEL_STATIC, ctx_idx0=np.arange(n), autoreset=False, max_steps=1 << 30)
    eng.reset()
    eng.state[:] = g["state"]
    out = eng.step(g["action"])
    np.testing.assert_allclose(out.obs, g["obs"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out.reward, g["reward"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(out.terminated, g["terminated"])
    np.testing.assert_array_equal(eng.branch_sig, g["branch_sig"])


def test_mass_ratio_floor_clamps_the_effective_mass_per_env():
    """carl_brax_ctx_map_t::mass_ratio_floor / _multi (round 3): a mass context below the stability floor runs at the
    floor -- the single-feature floor when it is the env's only light link, the higher combined floor when two or
    more links are lighter than nominal.  Free fall of the torso under a constant joint-less push shows the mass."""
    from carl_amd.envs.brax.models import halfcheetah_sys
    from carl_amd.envs import CARLBraxHalfcheetah

    feats = CARLBraxHalfcheetah.get_context_features()
    names = list(feats)
    default = np.array([float(f.default_value) for f in feats.values()])
    s = halfcheetah_sys(names)
    cm = s.ctx
    k_torso = [k for k in range(cm.n_mass) if names[cm.mass_row[k]] == "mass_torso"][0]
    k_foot = [k for k in range(cm.n_mass) if names[cm.mass_row[k]] == "mass_bfoot"][0]
    for k in range(cm.n_mass):
        cm.mass_ratio_floor[k] = 0.5
        cm.mass_ratio_floor_multi[k] = 0.8
    st0 = B.forward_kinematics(s, np.array(s.init_q[: s.n_q], dtype=np.float64) + np.array([0, 5.0] + [0] * (s.n_q - 2)),
                               np.zeros(s.n_dof))  # lifted 5 m: no contacts, joints at rest

    def torso_dv(row):
        F0, _ = B.joint_wrenches(s, row, np.zeros(s.n_dof), st0)
        tau = np.zeros(s.n_dof)
        tau[0] = 100.0  # a force on the root's x slide
        out = B.substeps(s, row, tau, 1, st0)
        return out[0, 7] - st0[0, 7]

    base = torso_dv(default)                       # effective mass ratio 1
    row = default.copy()
    row[cm.mass_row[k_torso]] = 0.7 * default[cm.mass_row[k_torso]]
    assert torso_dv(row) == pytest.approx(base / 0.7, rel=1e-9)        # above the floor: the context value
    row[cm.mass_row[k_torso]] = 0.2 * default[cm.mass_row[k_torso]]
    assert torso_dv(row) == pytest.approx(base / 0.5, rel=1e-9)        # one light link: the single-feature floor
    row[cm.mass_row[k_foot]] = 0.9 * default[cm.mass_row[k_foot]]
    assert torso_dv(row) == pytest.approx(base / float(np.float32(0.8)), rel=1e-9)  # two light links: the combined floor


def test_relative_joint_rotation_is_linear_in_the_body_quaternions_product():
    """The algebra the round-5 joint phase rests on (carl_amd/csrc/brax_kernels.hip.h: LinkRec::G): with
    q1 = conj(r_parent) (x) r_child, the relative rotation of the two joint frames
        rel = conj(r_parent (x) rpl) (x) (r_child (x) joint_rot) = conj(rpl) (x) q1 (x) joint_rot = G q1,
    G = L(conj(rpl)) R(joint_rot) a constant 4 x 4 matrix whose column c is conj(rpl) (x) e_c (x) joint_rot -- exactly
    how expand_link builds it; and with an unrotated link frame (rpl = joint_rot) G is block diagonal:
    rel = (|j|^2 q1.w, M q1.xyz).  Checked against the three quaternion products the restatement forms."""
    rng = np.random.default_rng(11)

    def qmul(a, b):
        return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                         a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                         a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])

    def conj(a):
        return a * np.array([1.0, -1.0, -1.0, -1.0])

    for case in range(200):
        rp, rc = rng.normal(size=4), rng.normal(size=4)
        rp, rc = rp / np.linalg.norm(rp), rc / np.linalg.norm(rc)
        jrot = rng.normal(size=4).astype(np.float32).astype(np.float64)  # table values: float32, not exactly unit
        jrot /= np.linalg.norm(jrot) * (1.0 + 1e-7 * rng.normal())
        lrot = np.array([1.0, 0, 0, 0]) if case % 2 == 0 else rng.normal(size=4)
        lrot /= np.linalg.norm(lrot)
        rpl = qmul(lrot, jrot)
        want = qmul(conj(qmul(rp, rpl)), qmul(rc, jrot))
        G = np.stack([qmul(qmul(conj(rpl), e), jrot) for e in np.eye(4)], axis=1)
        q1 = qmul(conj(rp), rc)
        np.testing.assert_allclose(G @ q1, want, rtol=0, atol=5e-16)
        if case % 2 == 0:  # unrotated link frame: the lean kernels' 10 numbers
            assert np.abs(G[0, 1:]).max() < 1e-16 and np.abs(G[1:, 0]).max() < 1e-16
            np.testing.assert_allclose(np.concatenate([[G[0, 0] * q1[0]], G[1:, 1:] @ q1[1:]]), want, rtol=0, atol=5e-16)
            np.testing.assert_allclose(G[0, 0], jrot @ jrot, rtol=1e-15)
