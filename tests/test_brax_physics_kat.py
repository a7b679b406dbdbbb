"""Analytic known-answer tests of the spring pipeline (tests/brax_kat.py has the cases and why they exist): the
fp64 C restatement on CPU, and the SAME cases on the HIP kernel under `-m gpu`.  Closed forms only -- nothing here
is compared with another implementation of the algorithm."""
import math

import numpy as np
import pytest

import brax_kat as K

RUNNERS = [pytest.param(K.OracleRunner, id="oracle"),
           pytest.param(K.EngineRunner, id="hip", marks=pytest.mark.gpu)]


def tol(runner, exact, f32):
    """the oracle is float64 throughout (exact-scheme cases hold to ~1e-12); the kernel keeps velocities / forces
    in float32"""
    return exact if runner is K.OracleRunner else f32


def _need_gpu(runner):
    if runner is K.EngineRunner:
        import torch

        if not torch.cuda.is_available():
            pytest.skip("no ROCm device")


@pytest.mark.parametrize("runner", RUNNERS)
def test_k1_free_fall_follows_the_scheme_and_the_gravity_context(runner):
    _need_gpu(runner)
    dt, n = 0.004, 50
    s = K.free_body(dt, n)
    g = np.array([-9.81, -3.7, -1.62, -24.8])  # per-lane gravity context
    rows = K.ctx_rows(4, gravity=g)
    g = rows[:, 0]
    v0 = np.array([[1.0, -2.0, 3.0], [0.0, 0.0, 0.0], [0.5, 0.5, -1.0], [0.0, 1.0, 10.0]])
    st0 = np.stack([K.body_state(p=(0, 0, 100.0), v=v) for v in v0])
    st = runner(s, rows, st0).step(0.0)[:, 0]
    t = n * dt
    dtf = float(np.float32(dt))  # the model table is float32
    t = n * dtf
    np.testing.assert_allclose(st[:, 7:9], v0[:, :2], rtol=tol(runner, 1e-13, 1e-6))
    np.testing.assert_allclose(st[:, 9], v0[:, 2] + g * t, rtol=tol(runner, 1e-12, 5e-6))  # 50 float32 additions
    np.testing.assert_allclose(st[:, 0:2], v0[:, :2] * t, rtol=tol(runner, 1e-12, 2e-6), atol=1e-12)
    # semi-implicit Euler: z_n = z0 + dt (n v0 + g dt n (n + 1) / 2)  -- NOT the continuous v0 t + g t^2 / 2
    z = 100.0 + dtf * (n * v0[:, 2] + g * dtf * n * (n + 1) / 2)
    np.testing.assert_allclose(st[:, 2], z, rtol=tol(runner, 1e-13, 1e-7))
    assert np.all(np.abs(st[:, 2] - (100.0 + v0[:, 2] * t + 0.5 * g * t * t)) > 1e-4 * np.abs(g))  # and it differs


@pytest.mark.parametrize("runner", RUNNERS)
def test_k2_bounce_reverses_the_normal_velocity_by_the_restitution(runner):
    _need_gpu(runner)
    dt, r, h = 0.001, 0.1, 0.5
    s = K.free_body(dt, 1, radius=r)  # one substep per env step: the trajectory is sampled every substep
    e = np.array([0.0, 0.3, 0.6, 0.9])
    rows = K.ctx_rows(4, gravity=-9.81, elasticity=e)
    e = rows[:, 2]
    st0 = np.stack([K.body_state(p=(0, 0, r + h))] * 4)
    run = runner(s, rows, st0)
    z, vz = [], []
    for _ in range(900):
        st = run.step(0.0)[:, 0]
        z.append(st[:, 2].copy())
        vz.append(st[:, 9].copy())
    z, vz = np.array(z), np.array(vz)
    for lane in range(4):
        k = int(np.argmax(vz[:, lane] > vz[0, lane]))  # first substep whose velocity went UP: the impact
        assert 300 < k < 340  # sqrt(2 h / g) = 0.319 s
        v_before = vz[k - 1, lane] + rows[lane, 0] * float(np.float32(dt))  # gravity acts before the collision pass
        # (erp = 0: no Baumgarte push) the contact impulse turns vn into -e vn -- exactly
        assert vz[k, lane] == pytest.approx(-e[lane] * v_before, rel=tol(runner, 1e-12, 2e-6), abs=tol(runner, 1e-12, 1e-6))
        if e[lane] > 0:
            apex = z[k:, lane].max() - r
            assert apex == pytest.approx(e[lane] ** 2 * h, rel=0.03)  # energy: h' = e^2 h (up to O(dt) of the scheme)


@pytest.mark.parametrize("runner", RUNNERS)
def test_k3_joint_constraint_spring_is_the_damped_oscillator_of_the_reduced_mass(runner):
    _need_gpu(runner)
    dt, n = 0.002, 40
    m0, m1, k, c = 3.0, 1.5, 800.0, 4.0
    s = K.two_bodies_on_a_joint(dt, n, m0=m0, m1=m1, k_pos=k, k_vel=c)
    rows = K.ctx_rows(3, gravity=-1e-12)
    delta = np.array([0.02, -0.05, 0.1])
    st0 = np.stack([np.stack([K.body_state(p=(0, 0, 5.0)), K.body_state(p=(0, d, 5.0))]) for d in delta])
    run = runner(s, rows, st0)
    mu = m0 * m1 / (m0 + m1)
    dtf = float(np.float32(dt))
    u, v = delta.copy(), np.zeros(3)
    for step in range(5):
        st = run.step(0.0)
        M = K.oscillator_matrix_power(k, c, mu, dtf, n)
        u, v = M[0, 0] * u + M[0, 1] * v, M[1, 0] * u + M[1, 1] * v
        rel_y = st[:, 1, 1] - st[:, 0, 1]
        rel_vy = st[:, 1, 8] - st[:, 0, 8]
        np.testing.assert_allclose(rel_y, u, rtol=tol(runner, 1e-10, 2e-5), atol=tol(runner, 1e-13, 2e-7))
        np.testing.assert_allclose(rel_vy, v, rtol=tol(runner, 1e-10, 2e-5), atol=tol(runner, 1e-12, 2e-6))
        # internal force: the pair's centre of mass does not move, nothing rotates
        com_y = (m0 * st[:, 0, 1] + m1 * st[:, 1, 1]) / (m0 + m1)
        np.testing.assert_allclose(com_y, m1 * delta / (m0 + m1), atol=tol(runner, 1e-13, 1e-7))
        assert np.abs(st[:, :, 10:13]).max() < tol(runner, 1e-12, 1e-6)
    # and it IS the textbook oscillator: frequency sqrt(k / mu), envelope exp(-c t / (2 mu))
    w = math.sqrt(k / mu - (c / (2 * mu)) ** 2)
    t = 5 * n * dtf
    cont = delta * math.exp(-c * t / (2 * mu)) * (math.cos(w * t) + c / (2 * mu * w) * math.sin(w * t))
    assert np.all(np.abs(u - cont) < 0.08 * np.abs(delta)), (u, cont)  # first-order scheme at w dt = 0.06


@pytest.mark.parametrize("runner", RUNNERS)
def test_k4_body_on_a_hinge_swings_with_the_physical_pendulum_period(runner):
    _need_gpu(runner)
    dt, frames = 2e-4, 50
    length, mass, inertia, g = 0.5, 2.0, 0.08, -9.81
    s = K.hinged_to_world(dt, frames, axis=(0, 1, 0), com=(0.0, 0.0, -length), mass=mass, inv_inertia=1.0 / inertia,
                          k_pos=2e5, k_vel=50.0)
    from oracle import brax as B

    theta0 = np.array([0.05, 0.1])
    st0 = np.stack([B.forward_kinematics(s, [th], [0.0]) for th in theta0])
    run = runner(s, K.ctx_rows(2, gravity=g), st0)
    ang = []
    for _ in range(260):  # 2.6 s: two periods
        st = run.step(0.0)[:, 0]
        ang.append([K.hinge_angle(st[k]) for k in range(2)])
    ang = np.array(ang)
    T_step = frames * float(np.float32(dt))
    for lane in range(2):
        a = ang[:, lane]
        down = [i for i in range(1, len(a)) if a[i - 1] > 0 >= a[i]]  # downward zero crossings, one per period
        assert len(down) >= 2
        t = [(i - 1 + a[i - 1] / (a[i - 1] - a[i])) * T_step for i in down]
        period = t[1] - t[0]
        want = 2 * math.pi * math.sqrt((inertia + mass * length ** 2) / (mass * abs(g) * length)) * (1 + theta0[lane] ** 2 / 16)
        assert period == pytest.approx(want, rel=3e-3), (period, want)
        assert np.abs(a).max() <= theta0[lane] * 1.001  # no energy gain


@pytest.mark.parametrize("runner", RUNNERS)
def test_k5_actuator_torque_accelerates_the_hinge_against_joint_and_angular_damping(runner):
    _need_gpu(runner)
    dt, n = 0.001, 100
    inertia, gear, d_joint, d_ang = 0.25, 3.0, 0.4, 0.7
    s = K.hinged_to_world(dt, n, axis=(1, 0, 0), inv_inertia=1.0 / inertia, k_pos=1000.0, k_vel=10.0, k_ang_damp=d_ang,
                          dof_damping=d_joint, gear=gear)
    act = np.array([[0.5], [-1.0], [2.5], [0.0]], dtype=np.float32)  # 2.5 clips to the control range's 1.0
    st0 = np.stack([K.body_state(p=(0, 0, 1.0))] * 4)[:, None, :]
    run = runner(s, K.ctx_rows(4, gravity=-1e-12), st0)
    st = run.step(act)[:, 0]
    dtf = float(np.float32(dt))
    tau = gear * np.clip(act[:, 0].astype(np.float64), -1.0, 1.0)
    w, th = np.zeros(4), np.zeros(4)
    for _ in range(n):  # w' = w + dt (tau - (d_joint + d_ang) w) / I ;  the quaternion turns by 2 atan(w' dt / 2)
        w = w + dtf * (tau - (float(np.float32(d_joint)) + float(np.float32(d_ang))) * w) / inertia
        th = th + 2 * np.arctan(0.5 * dtf * w)
    np.testing.assert_allclose(st[:, 10], w, rtol=tol(runner, 1e-9, 5e-6), atol=1e-12)
    np.testing.assert_allclose([K.hinge_angle(r) for r in st], th, rtol=tol(runner, 1e-9, 5e-6), atol=1e-12)
    # the geometric-series limit: w -> tau / d (1 - (1 - dt d / I)^n)
    d = float(np.float32(d_joint)) + float(np.float32(d_ang))
    np.testing.assert_allclose(w, tau / d * (1 - (1 - dtf * d / inertia) ** n), rtol=1e-9, atol=1e-12)
    assert np.abs(st[:, 11:13]).max() < 1e-9 and st[3, 10] == 0.0


@pytest.mark.parametrize("runner", RUNNERS)
def test_k6_limit_spring_holds_a_constant_torque_at_hi_plus_tau_over_k(runner):
    _need_gpu(runner)
    dt, n = 0.001, 500
    k_lim, gear, hi = 120.0, 6.0, 0.3
    s = K.hinged_to_world(dt, n, axis=(0, 0, 1), k_pos=1000.0, k_vel=10.0, k_limit=k_lim, dof_damping=18.0, lo=-0.2, hi=hi,
                          gear=gear)
    act = np.array([[1.0], [0.5], [-1.0], [0.02]], dtype=np.float32)
    st0 = np.stack([K.body_state(p=(0, 0, 1.0))] * 4)[:, None, :]
    run = runner(s, K.ctx_rows(4, gravity=-1e-12), st0)
    for _ in range(8):  # 4 s: settled (critical damping 2 sqrt(k I) = 21.9)
        st = run.step(act)[:, 0]
    th = np.array([K.hinge_angle(r) for r in st])
    tau = gear * act[:, 0].astype(np.float64)
    lo_f, hi_f = float(np.float32(-0.2)), float(np.float32(hi))
    np.testing.assert_allclose(th[0], hi_f + tau[0] / k_lim, rtol=tol(runner, 1e-9, 1e-6))
    np.testing.assert_allclose(th[1], hi_f + tau[1] / k_lim, rtol=tol(runner, 1e-9, 1e-6))
    np.testing.assert_allclose(th[2], lo_f + tau[2] / k_lim, rtol=tol(runner, 1e-9, 1e-6))
    assert np.abs(st[:3, 10:13]).max() < 1e-6  # at rest
    # a small torque INSIDE the range: nothing but the joint damping opposes it -- terminal rate tau / d (to the
    # float32 quantisation of the model table's joint-frame quaternion: |x_c| = 1 + 3e-8)
    assert 0.0 < th[3] < hi_f and st[3, 12] == pytest.approx(tau[3] / 18.0, rel=tol(runner, 1e-6, 1e-5))


@pytest.mark.parametrize("runner", RUNNERS)
def test_k7_coulomb_friction_stops_a_sliding_sphere_after_v0_squared_over_2_mu_g(runner):
    _need_gpu(runner)
    dt, r = 0.002, 0.1
    s = K.free_body(dt, 1, radius=r, inv_inertia=0.0)  # infinite rotational inertia: it slides, it does not roll
    mu = np.array([0.2, 0.5, 1.0, 0.5])
    g = np.array([-9.81, -9.81, -9.81, -3.0])
    v0 = np.array([1.0, 2.0, 1.5, 1.0])
    rows = K.ctx_rows(4, gravity=g, friction=mu)
    mu, g = rows[:, 1], rows[:, 0]
    st0 = np.stack([K.body_state(p=(0, 0, r - 1e-4), v=(v, 0, 0)) for v in v0])  # resting 0.1 mm inside the floor
    run = runner(s, rows, st0)
    dtf = float(np.float32(dt))
    n_stop = np.floor(v0 / (mu * np.abs(g) * dtf)).astype(int)  # substeps with the friction cone saturated
    xs, vs = [], []
    for _ in range(int(n_stop.max()) + 5):
        st = run.step(0.0)[:, 0]
        xs.append(st[:, 0].copy())
        vs.append(st[:, 7].copy())
    xs, vs = np.array(xs), np.array(vs)
    for lane in range(4):
        dec = mu[lane] * abs(g[lane]) * dtf
        k = np.arange(1, n_stop[lane] + 1)
        # every substep the normal impulse is m |g| dt, the friction impulse mu times that, until what is left of
        # the tangential velocity is smaller: then the sphere stops dead
        np.testing.assert_allclose(vs[: n_stop[lane], lane], v0[lane] - k * dec, rtol=tol(runner, 1e-10, 2e-5), atol=tol(runner, 1e-12, 5e-6))
        assert np.all(np.abs(vs[n_stop[lane] + 1:, lane]) <= tol(runner, 0.0, 1e-6))  # (v_rcp_f32 on the kernel: zero to an ulp)
        dist = dtf * np.sum(v0[lane] - k * dec)
        assert xs[-1, lane] == pytest.approx(dist, rel=tol(runner, 1e-10, 1e-5))
        assert xs[-1, lane] == pytest.approx(v0[lane] ** 2 / (2 * mu[lane] * abs(g[lane])), rel=2.5 * dec / v0[lane])
    assert np.abs(st[:, 2] - (r - 1e-4)).max() < tol(runner, 1e-12, 1e-7)  # and it never left the floor (erp = 0)


@pytest.mark.parametrize("runner", RUNNERS)
def test_k8_torque_free_spin_turns_by_two_atan_half_w_dt_per_substep(runner):
    _need_gpu(runner)
    dt, n = 0.005, 200
    s = K.free_body(dt, n)
    c = np.array([0.0, -0.05, -0.5, 0.0])  # ang_damping context (CARL's default is -0.05)
    W = np.array([3.0, 3.0, 10.0, 40.0])
    rows = K.ctx_rows(4, gravity=-1e-12, ang_damping=c)
    c = rows[:, 3]
    st0 = np.stack([K.body_state(p=(0, 0, 5.0), w=(0, 0, w)) for w in W])
    st = runner(s, rows, st0).step(0.0)[:, 0]
    dtf = float(np.float32(dt))
    k = np.arange(1, n + 1)[:, None]
    wk = W[None] * np.exp(c[None] * dtf * k)
    angle = np.sum(2 * np.arctan(0.5 * dtf * wk), axis=0)
    np.testing.assert_allclose(st[:, 12], wk[-1], rtol=tol(runner, 1e-10, 3e-5))
    got = np.array([K.hinge_angle(r) for r in st])
    d = (got - angle + math.pi) % (2 * math.pi) - math.pi
    assert np.abs(d).max() < tol(runner, 1e-9, 2e-4), d
    np.testing.assert_allclose(np.linalg.norm(st[:, 3:7], axis=1), 1.0, atol=1e-12)
    # the integrator under-rotates against the continuous w t by (w dt)^2 / 12 relative
    assert abs((angle[3] - W[3] * n * dtf) / (W[3] * n * dtf) + (W[3] * dtf) ** 2 / 12) < 2e-4


def _pusher_case(n):
    """CARLBraxPusher's model table on the four context columns of this file, a reset arm pose from the restatement's
    forward kinematics, and nothing damping the puck's slides but the table."""
    from carl_amd.envs.brax.models import pusher_sys
    from oracle import brax as B

    s = pusher_sys(K.CTX_NAMES)
    s.max_episode_steps = K.BIG
    s.dof_damping[7] = s.dof_damping[8] = 0.0  # (the MJCF's joint damping 0.5: off, Coulomb friction alone stops the puck)
    q = np.zeros(9)
    st = B.forward_kinematics(s, q, np.zeros(9)).reshape(8, 13)
    return s, np.tile(st[None], (n, 1, 1))


@pytest.mark.parametrize("runner", RUNNERS)
def test_k9_puck_slides_to_rest_on_the_table_after_v0_squared_over_2_mu_g(runner):
    """The push task's object on the table (carl_brax_sys_t::obj_support, ABI 8): under the context's gravity the puck
    carries the normal load m |g|, the table answers with Coulomb friction -- per substep the horizontal speed drops by
    mu |g| dt until what is left is smaller, then the puck stops dead; distance = the scheme's arithmetic series, and the
    textbook v0^2 / (2 mu g) to the discretisation.  With the MJCF's own zero gravity there is no load: it keeps sliding.
    (carl/envs/brax/carl_pusher.py:17-21: gravity is a context feature with default -9.8.)"""
    _need_gpu(runner)
    mu = np.array([0.5, 1.0, 0.3, 0.7, 0.5])
    g = np.array([-9.8, -9.8, -9.8, -3.0, 0.0])
    v0 = np.array([0.5, 1.0, 0.4, 0.3, 0.25])
    n = len(mu)
    s, st0 = _pusher_case(n)
    rows = K.ctx_rows(n, gravity=g, friction=mu)
    mu, g = rows[:, 1], rows[:, 0]
    ang = np.array([0.0, 0.5, 2.0, -1.0, 0.3])  # direction of travel: the friction is isotropic in the plane
    st0[:, 7, 0:2] = (-0.6, 0.8)                # far from the arm (which sags onto the table meanwhile)
    st0[:, 7, 7], st0[:, 7, 8] = v0 * np.cos(ang), v0 * np.sin(ang)
    run = runner(s, rows, st0.reshape(n, -1))
    dtf = float(np.float32(s.dt))
    dec = mu * np.abs(g) * dtf
    n_stop = np.where(dec > 0, np.floor(v0 / np.where(dec > 0, dec, 1.0)), 0).astype(int)
    steps = int(n_stop.max() // s.n_frames) + 2
    for _ in range(steps):
        st = run.step(0.0)
    d = np.hypot(st[:, 7, 0] + 0.6, st[:, 7, 1] - 0.8)
    for lane in range(n - 1):
        k = np.arange(1, n_stop[lane] + 1)
        dist = dtf * np.sum(v0[lane] - k * dec[lane])
        assert d[lane] == pytest.approx(dist, rel=tol(runner, 1e-9, 2e-5))
        assert d[lane] == pytest.approx(v0[lane] ** 2 / (2 * mu[lane] * abs(g[lane])), rel=2.5 * dec[lane] / v0[lane])
        assert np.hypot(st[lane, 7, 7], st[lane, 7, 8]) <= tol(runner, 0.0, 1e-6)  # at rest
        # along the initial direction, on the table's height all the way
        assert np.arctan2(st[lane, 7, 1] - 0.8, st[lane, 7, 0] + 0.6) == pytest.approx(ang[lane], abs=1e-5)
    # its height: held by the slide joint's constraint spring, which gives by m |g| / k_pos under the weight
    sag = np.array(s.mass[7]) * np.abs(g) / s.k_pos[7]
    np.testing.assert_allclose(st[:, 7, 2], -0.275 - sag, atol=2e-6)
    # no gravity, no load, no friction: still moving at v0
    assert np.hypot(st[-1, 7, 7], st[-1, 7, 8]) == pytest.approx(v0[-1], rel=tol(runner, 1e-12, 1e-6))
    assert d[-1] == pytest.approx(v0[-1] * dtf * s.n_frames * steps, rel=tol(runner, 1e-9, 1e-5))


@pytest.mark.parametrize("runner", RUNNERS)
def test_k10_the_fork_comes_to_rest_on_the_table_plane(runner):
    """`plane_z` (ABI 8): the push task's colliders -- the fork's seven spheres -- meet the plane z = -0.325 the puck lies
    on, not z = 0 (where every locomotion model's ground is).  Under gravity the unpowered arm sags until the fork rests
    on the table: its lowest sphere ends within the Baumgarte slack of plane_z + radius, and never goes through."""
    _need_gpu(runner)
    n = 3
    s, st0 = _pusher_case(n)
    assert s.plane_z == pytest.approx(-0.325) and s.n_coll == 7 and all(s.coll_link[k] == 6 for k in range(7))
    rows = K.ctx_rows(n, gravity=np.array([-9.8, -4.0, -9.8]), friction=np.array([1.0, 1.0, 0.2]))
    st0[:, 7, 0:2] = (-0.6, 0.8)  # the puck out of the way
    run = runner(s, rows, st0.reshape(n, -1))
    lowest = []
    for _ in range(40):  # 2 s
        st = run.step(0.0)
        r = st[:, 6, 3:7]
        R2 = np.stack([2 * (r[:, 1] * r[:, 3] - r[:, 0] * r[:, 2]), 2 * (r[:, 2] * r[:, 3] + r[:, 0] * r[:, 1]),
                       1 - 2 * (r[:, 1] ** 2 + r[:, 2] ** 2)], 1)  # third row of the rotation matrix
        z = [st[:, 6, 2] + R2 @ (np.array(s.coll_pos[k][:]) - np.array(s.com[6][:])) for k in range(7)]
        lowest.append(np.min(z, 0))
    lowest = np.array(lowest)
    assert (lowest.min(0) > s.plane_z + 0.02 - 0.012).all()   # never deeper than ~1 cm into the table
    assert (np.abs(lowest[-1] - (s.plane_z + 0.02)) < 0.004).all()  # and it rests ON it
    assert (lowest[0] > s.plane_z + 0.1).all()                # (it started well above)
