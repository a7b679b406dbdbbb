"""Anisotropic effective inertia (carl_brax_sys_t::inv_inertia with three different principal moments).

No shipped model has it (spring_inertia_scale = 1 makes every effective inertia isotropic), but the C ABI's model table
allows it and oracle/brax_spring.c implements it (w += dt R diag(inv_inertia) R^T T, brax.spring.integrator / collisions).
Since round 6 the rotated-inertia code lives in the GENERAL kernels only (template parameter TASK; carl_brax.hip:
brax_is_task sends every model with an anisotropic link there) -- the lean and multi-hinge kernels compile isotropy in.
These tests pin that route: a closed form through both back ends, and the contact impulse of an anisotropic body on the
HIP kernel against the fp64 restatement (an implementation-vs-implementation check, like the parity tests).
"""
import numpy as np
import pytest

import brax_kat as K
from carl_amd.envs.brax.models import _set3

RUNNERS = [pytest.param(K.OracleRunner, id="oracle"),
           pytest.param(K.EngineRunner, id="hip", marks=pytest.mark.gpu)]


def _need_gpu(runner):
    if runner is K.EngineRunner:
        import torch

        if not torch.cuda.is_available():
            pytest.skip("no ROCm device")


@pytest.mark.parametrize("runner", RUNNERS)
@pytest.mark.parametrize("axis,k", [((1, 0, 0), 0), ((0, 1, 0), 1), ((0, 0, 1), 2)])
def test_hinge_torque_meets_the_principal_moment_of_its_axis(runner, axis, k):
    """A body hinged to the world about body axis k, link frame = world frame: the actuator torque is parallel to a principal
    axis at every angle (a rotation about e_k leaves e_k alone), so R diag(c) R^T tau = c_k tau EXACTLY -- the hinge speeds
    up with the k-th inverse moment and nothing else; the scheme's arithmetic series as in K5 (no damping here)."""
    _need_gpu(runner)
    dt, n = 0.001, 100
    inv_i = (4.0, 0.5, 1.5)
    gear = 2.0
    s = K.hinged_to_world(dt, n, axis=axis, k_pos=1000.0, k_vel=10.0, gear=gear)
    _set3(s.inv_inertia, 0, inv_i)
    act = np.array([[0.5], [-1.0], [0.25], [0.0]], dtype=np.float32)
    st0 = np.stack([K.body_state(p=(0, 0, 1.0))] * 4)[:, None, :]
    run = runner(s, K.ctx_rows(4, gravity=-1e-12), st0)
    st = run.step(act)[:, 0]
    dtf = float(np.float32(dt))
    tau = gear * act[:, 0].astype(np.float64)
    w, th = np.zeros(4), np.zeros(4)
    for _ in range(n):
        w = w + dtf * tau * inv_i[k]
        th = th + 2 * np.arctan(0.5 * dtf * w)
    exact = runner is K.OracleRunner
    # (the joint frame of a y / z hinge is a float32 quaternion with sqrt(1/2) entries: its x axis is e_k to 3e-8)
    np.testing.assert_allclose(st[:, 10 + k], w, rtol=2e-7 if exact else 5e-6, atol=1e-12)
    np.testing.assert_allclose([K.hinge_angle(r) for r in st], th, rtol=2e-7 if exact else 5e-6, atol=1e-12)
    others = [10 + j for j in range(3) if j != k]
    assert np.abs(st[:, others]).max() < (2e-7 if exact else 1e-6)


def _spinning_brick(dt, n_frames):
    """a free body with three different principal moments and two off-centre collision spheres"""
    s = K.free_body(dt, n_frames)
    s.mass[0] = 2.0
    _set3(s.inv_inertia, 0, (6.0, 1.5, 0.4))
    s.n_coll = 2
    s.coll_link[0], s.coll_radius[0] = 0, 0.10
    _set3(s.coll_pos, 0, (0.25, 0.05, -0.02))
    s.coll_link[1], s.coll_radius[1] = 0, 0.08
    _set3(s.coll_pos, 1, (-0.20, -0.10, 0.03))
    s.baumgarte_erp = 0.1
    return s


@pytest.mark.gpu
def test_anisotropic_contact_impulse_matches_the_fp64_restatement():
    """Tumbling bricks dropped on the plane with friction and restitution: the impulse's effective mass n . (I^-1 (r x n) x r),
    the friction direction's, and the angular velocity change I^-1 (r x J) all go through R diag(c) R^T with a rotated R.  One
    env step = one substep, the restatement restarted from the kernel's state every step (as the parity tests do), so only one
    substep's arithmetic is compared; lanes whose contact decision differs within rounding are set aside by comparing the two
    sides' velocity jumps (a fired contact changes v_z by >= 1e-3 here)."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    dt = 0.002
    s = _spinning_brick(dt, 1)
    n = 64
    rng = np.random.default_rng(11)
    rows = K.ctx_rows(n, gravity=-9.81, friction=rng.uniform(0.3, 1.2, n), elasticity=rng.uniform(0.0, 0.5, n))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    st0 = np.stack([K.body_state(p=(0, 0, rng.uniform(0.15, 0.45)), r=q[i], v=rng.uniform(-1, 1, 3) * (1, 1, 2),
                                 w=rng.uniform(-6, 6, 3)) for i in range(n)])[:, None, :]
    hip = K.EngineRunner(s, rows, st0)
    fired, worst = 0, 0.0
    prev = st0.copy()
    for _ in range(400):
        st = hip.step(0.0)
        ref = K.OracleRunner(s, rows, prev).step(0.0)
        # same discrete decisions: the kernel's and the restatement's v_z jumps agree in kind
        jump_h = np.abs(st[:, 0, 9] - (prev[:, 0, 9] + rows[:, 0] * float(np.float32(dt)))) > 1e-4
        jump_r = np.abs(ref[:, 0, 9] - (prev[:, 0, 9] + rows[:, 0] * float(np.float32(dt)))) > 1e-4
        same = jump_h == jump_r
        assert same.mean() > 0.95
        fired += int((jump_h & same).sum())
        err = np.abs(st[same] - ref[same]) / (1.0 + np.abs(ref[same]))
        worst = max(worst, float(err.max()))
        prev = st
    assert fired > 500, fired  # the bricks did hit the ground, many times
    assert worst < 1e-5, worst
    assert np.isfinite(prev).all()
