"""The classic-control restatements against CLOSED-FORM PHYSICS (none of it derived from either implementation).

gymnasium is not installable here, so E-CP / E-PD / E-AC are restated from its published source and
pinned by self-derived known answers -- which guard against transcription slips in OUR copies, not against a
misremembered equation.  These cases close that gap from the other side: each equation set must reproduce the textbook
behaviour of the mechanical system it claims to be (as tests/test_brax_physics_kat.py does for the spring pipeline).

K1  Pendulum-v1 is a uniform rod on a pivot: hanging, it swings with omega^2 = 3 g / (2 l)   (I = m l^2 / 3)
K2  Pendulum-v1: a constant torque u holds the rod at sin(theta*) = -2 u / (m g l): the static balance of a rod of mass m
K3  CartPole-v1 linearised about upright diverges like cosh(lambda t), lambda^2 = g / (l (4/3 - m_p / M))
K4  CartPole-v1: with the pole upright and at rest, the cart under the constant force F accelerates at F / M
K5  Acrobot-v1 (book dynamics) hanging: small oscillations are the two normal modes of M th'' + K th = 0
K6  MountainCar: the hill is the potential (g / 3) sin(3 p): force-free motion between two turning points of equal height
"""
import math

import numpy as np
import pytest

from oracle import oracle as O


def _step(fam, row, state, action):
    s2, obs, rew, term = O.transitions(fam, np.asarray([row], dtype=np.float64), np.asarray([state], dtype=np.float64),
                                       np.asarray([action]))
    return s2[0]


def _row(fam, **kv):
    names = O.feature_names(fam)
    row = O.default_row(fam).copy()
    for k, v in kv.items():
        row[names.index(k)] = v
    return row


@pytest.mark.parametrize("g,l", [(10.0, 1.0), (4.2, 1.7), (19.0, 0.6)])
def test_k1_pendulum_is_a_uniform_rod(g, l):
    dt = 1e-3
    row = _row(O.PENDULUM, g=g, l=l, dt=dt, m=0.8)
    s = np.array([math.pi + 0.01, 0.0])  # theta = 0 is upright in Pendulum-v1: pi hangs
    crossings, prev = [], s[0] - math.pi
    for k in range(20000):
        s = _step(O.PENDULUM, row, s, np.float32(0.0))
        cur = s[0] - math.pi
        if prev < 0 <= cur or prev > 0 >= cur:
            crossings.append((k + 1 - cur / (cur - prev)) * dt)
        prev = cur
        if len(crossings) >= 5:
            break
    period = (crossings[4] - crossings[0]) / 2
    assert period == pytest.approx(2 * math.pi * math.sqrt(2 * l / (3 * g)), rel=2e-3)


def test_k2_pendulum_static_balance_under_constant_torque():
    g, l = 10.0, 1.0
    # E-PD: thdot' = thdot + (3 g / (2 l) sin th + 3 / (m l^2) u) dt  ->  at rest where sin th* = -2 u / (m g l):
    # a rod of twice the mass needs twice the torque to hang at the same angle
    for m, u in ((1.0, 1.0), (2.0, 2.0), (0.5, 0.5)):
        th = math.asin(-2 * u / (m * g * l))
        assert th == pytest.approx(math.asin(-0.2))
        s = _step(O.PENDULUM, _row(O.PENDULUM, g=g, l=l, m=m, dt=0.05), np.array([th, 0.0]), np.float32(u))
        assert abs(s[1]) < 1e-12 and s[0] == pytest.approx(th, abs=1e-12)
    # the action is clipped to max_torque = 2: u = 3 acts like u = 2
    row = _row(O.PENDULUM, g=g, l=l, m=1.0, dt=0.05)
    a = _step(O.PENDULUM, row, np.array([0.3, 0.1]), np.float32(3.0))
    b = _step(O.PENDULUM, row, np.array([0.3, 0.1]), np.float32(2.0))
    np.testing.assert_array_equal(a, b)


def test_k3_cartpole_upright_is_unstable_at_the_textbook_rate():
    tau = 1e-3
    mp, mc, l, g = 0.1, 1.0, 0.5, 9.8  # gymnasium defaults: total_mass 1.1, polemass_length 0.05 (Quirk C1 is vacuous here)
    row = _row(O.CARTPOLE, tau=tau, force_mag=1e-9)  # bang-bang force switched off: free dynamics
    lam = math.sqrt(g / (l * (4.0 / 3.0 - mp / (mc + mp))))
    th0 = 1e-4
    s = np.array([0.0, 0.0, th0, 0.0])
    n = 400  # t = 0.4 s: theta stays inside the 12 degree bound
    for _ in range(n):
        s = _step(O.CARTPOLE, row, s, 1)
    assert s[2] / th0 == pytest.approx(math.cosh(lam * n * tau), rel=4e-3)
    assert s[3] / th0 == pytest.approx(lam * math.sinh(lam * n * tau), rel=4e-3)
    # the cart recoils: x'' = -(m_p l / M) theta'' to first order
    assert s[0] == pytest.approx(-(mp * l / (mc + mp)) * (s[2] - th0), rel=2e-2)


def test_k4_cartpole_cart_acceleration_with_the_pole_upright():
    F, M = 10.0, 1.1
    row = _row(O.CARTPOLE)
    s = _step(O.CARTPOLE, row, np.zeros(4), 1)
    # theta = 0: temp = F / M; thetaacc = -temp / (l (4/3 - m_p / M)); xacc = temp - m_p l thetaacc / M
    thacc = -(F / M) / (0.5 * (4.0 / 3.0 - 0.1 / M))
    assert s[3] == pytest.approx(thacc * 0.02, rel=1e-12)
    assert s[1] == pytest.approx((F / M - 0.05 * thacc / M) * 0.02, rel=1e-12)
    assert s[1] == pytest.approx(0.1951219512195122, rel=1e-12)  # the well-known first CartPole step


@pytest.mark.parametrize("ctx", [{}, {"LINK_LENGTH_1": 1.5, "LINK_MASS_1": 0.7, "LINK_MASS_2": 1.3, "LINK_COM_POS_1": 0.4,
                                      "LINK_COM_POS_2": 0.6, "LINK_MOI": 1.2}])
def test_k5_acrobot_normal_modes(ctx):
    names = O.feature_names(O.ACROBOT)
    row = _row(O.ACROBOT, **ctx)
    get = lambda k: row[names.index(k)]  # noqa: E731
    m1, m2, l1, lc1, lc2, I = get("LINK_MASS_1"), get("LINK_MASS_2"), get("LINK_LENGTH_1"), get("LINK_COM_POS_1"), get("LINK_COM_POS_2"), get("LINK_MOI")
    g = 9.8
    Mm = np.array([[m1 * lc1**2 + m2 * (l1**2 + lc2**2 + 2 * l1 * lc2) + 2 * I, m2 * (lc2**2 + l1 * lc2) + I],
                   [m2 * (lc2**2 + l1 * lc2) + I, m2 * lc2**2 + I]])
    K = np.array([[(m1 * lc1 + m2 * l1) * g + m2 * lc2 * g, m2 * lc2 * g], [m2 * lc2 * g, m2 * lc2 * g]])
    w2, V = np.linalg.eig(np.linalg.solve(Mm, K))
    assert (w2.real > 0).all() and np.abs(w2.imag).max() < 1e-12
    for k in range(2):  # start ON a mode: it must keep its shape and turn at its own frequency
        w, v = math.sqrt(w2[k].real), V[:, k].real
        amp = 1e-3
        s = np.array([amp * v[0], amp * v[1], 0.0, 0.0])
        n = 10  # 2 s of RK4 at dt = 0.2 (w dt < 1.3: RK4's phase error stays below a percent per step)
        for _ in range(n):
            s = _step(O.ACROBOT, row, s, 1)  # action 1 = zero torque
        want = amp * v * math.cos(w * n * 0.2)
        assert np.abs(s[:2] - want).max() < 0.03 * amp, (k, w, s[:2], want)


def test_k6_mountaincar_hill_is_a_potential():
    # v += -gravity cos(3 p) per unit step, p += v: the discrete flow of H = v^2 / 2 + (gravity / 3) sin(3 p); with a
    # small gravity the time step is small against the oscillation and the two turning points have equal height
    grav = 1e-5
    row = _row(O.MOUNTAINCAR, gravity=grav, force=0.0, max_speed=10.0, goal_position=5.0, min_position=-1.2, max_position=0.6)
    p0 = -0.9
    s = np.array([p0, 0.0])
    turning, prev_v = [], 0.0
    for _ in range(20000):
        s = _step(O.MOUNTAINCAR, row, s, 1)
        if prev_v > 0 >= s[1] or prev_v < 0 <= s[1]:
            turning.append(s[0])
            if len(turning) == 2:
                break
        prev_v = s[1]
    # released at rest at p0 on the left slope: the first turning point is on the other side of the valley bottom
    # (p = -pi / 6) at the same height sin(3 p), the second one back at p0
    assert math.sin(3 * turning[0]) == pytest.approx(math.sin(3 * p0), abs=2e-3) and turning[0] > -math.pi / 6
    assert turning[1] == pytest.approx(p0, abs=2e-3)
