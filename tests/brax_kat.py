"""Analytic known-answer cases for the spring pipeline -- closed forms that NEITHER implementation (the fp64 C
restatement oracle/brax_spring.c, the HIP kernel) was written from or tuned to.

brax 0.12.1 cannot be imported here and the reference's tests hold no Brax step value
(/root/reference test/test_brax_env.py:8-23), so the restatement cannot be pinned against brax.  What CAN be
pinned is that it is the physics it claims to be: each case below is a tiny custom `carl_brax_sys_t` whose motion
has a closed form -- either the textbook continuous one (pendulum period, Coulomb stopping distance, rebound
height) or the EXACT solution of the pipeline's integration scheme for a linear system (semi-implicit Euler is a
2 x 2 linear map on a spring-damper pair; a constant torque gives an arithmetic-geometric series).  Every case
runs through the same two back ends: `OracleRunner` (CPU, `-m "not gpu"`) and `EngineRunner` (the HIP kernel
through the C ABI, `-m gpu`).

Cases (tests/test_brax_physics_kat.py):
  K1 free fall            z(n), v(n) of the scheme; per-lane gravity context
  K2 bounce               v+ = -e v- at the impact substep, rebound apex ~ e^2 h
  K3 constraint spring    two free bodies on a joint: relative coordinate = M^n (delta, 0), M from (k_pos, k_vel, mu, dt)
  K4 pendulum             period 2 pi sqrt((I + m l^2) / (m g l)) (1 + theta0^2 / 16)
  K5 actuator             theta_dot(n) under gear x clip(action) with joint + angular damping, unit inertia
  K6 limit spring         static equilibrium theta* = hi + tau / k_limit
  K7 Coulomb friction     exact stopping substep and distance of the scheme; ~ v0^2 / (2 mu g)
  K8 torque-free spin     rotation angle sum_k 2 atan(w_k dt / 2), w_k = W exp(c dt k) (ang_damping context)
"""
import math

import numpy as np

from carl_amd import _lib
from carl_amd.envs.brax.models import _axis_quat, _set3, _wire_context

CTX_NAMES = ["gravity", "friction", "elasticity", "ang_damping"]
BIG = 1 << 30


def _blank(n_links, n_q, n_dof, dt, n_frames):
    s = _lib.BraxSys()
    s.env_kind = _lib.BRAX_ANT
    s.healthy_q_index = -1
    s.n_links, s.n_q, s.n_dof, s.n_act = n_links, n_q, n_dof, 1
    s.obs_dim = n_q + n_dof
    s.max_episode_steps = BIG
    s.terminate_when_unhealthy = 0
    s.exclude_current_positions = 0
    s.reset_vel_uniform = 1
    s.dt, s.n_frames = dt, n_frames
    s.gravity_z, s.vel_damping, s.ang_damping = -9.81, 0.0, 0.0
    s.baumgarte_erp, s.elasticity, s.friction = 0.0, 0.0, 1.0
    s.healthy_z_lo, s.healthy_z_hi = -1e9, 1e9
    s.healthy_reward, s.ctrl_cost_weight, s.forward_reward_weight = 0.0, 0.0, 0.0
    ident = (1.0, 0.0, 0.0, 0.0)
    for i in range(n_links):
        _set3(s.link_rot, i, ident)
        _set3(s.joint_rot, i, ident)
        s.dof_sign3[i] = 1.0
        s.mass[i] = 1.0
        _set3(s.inv_inertia, i, (1.0, 1.0, 1.0))
    for d in range(n_dof):
        s.dof_lo[d], s.dof_hi[d] = -1e9, 1e9
    s.act_dof[0], s.act_gear[0], s.act_lo[0], s.act_hi[0] = n_dof - 1, 0.0, -1.0, 1.0
    _wire_context(s, CTX_NAMES, False, {}, {})
    return s


def free_body(dt, n_frames, *, radius=0.0, mass=1.0, inv_inertia=1.0):
    """one free rigid body, optionally a collision sphere centred on its COM"""
    s = _blank(1, 7, 6, dt, n_frames)
    s.parent[0], s.n_link_dof[0], s.q_start[0], s.dof_start[0] = -1, 6, 0, 0
    s.mass[0] = mass
    _set3(s.inv_inertia, 0, (inv_inertia,) * 3)
    if radius > 0:
        s.n_coll = 1
        s.coll_link[0], s.coll_radius[0] = 0, radius
        _set3(s.coll_pos, 0, (0.0, 0.0, 0.0))
    return s


def two_bodies_on_a_joint(dt, n_frames, *, m0, m1, k_pos, k_vel):
    """free root + one link hinged (about x) to it; both anchors at the COMs, so the joint's linear spring acts
    through the centres of mass: pure relative translation"""
    s = _blank(2, 8, 7, dt, n_frames)
    s.parent[0], s.n_link_dof[0], s.q_start[0], s.dof_start[0] = -1, 6, 0, 0
    s.parent[1], s.n_link_dof[1], s.q_start[1], s.dof_start[1] = 0, 1, 7, 6
    s.mass[0], s.mass[1] = m0, m1
    s.k_pos[1], s.k_vel[1] = k_pos, k_vel
    return s


def hinged_to_world(dt, n_frames, *, axis=(0, 1, 0), com=(0.0, 0.0, 0.0), mass=1.0, inv_inertia=1.0, k_pos=0.0, k_vel=0.0,
                    k_limit=0.0, k_ang_damp=0.0, dof_damping=0.0, lo=-1e9, hi=1e9, gear=0.0, height=1.0):
    """one body on a hinge against the static world (parent -1, one rotational dof), anchor at its frame origin"""
    s = _blank(1, 1, 1, dt, n_frames)
    s.parent[0], s.n_link_dof[0], s.n_slide[0], s.q_start[0], s.dof_start[0] = -1, 1, 0, 0, 0
    _set3(s.link_pos, 0, (0.0, 0.0, height))
    _set3(s.joint_rot, 0, _axis_quat(axis))
    _set3(s.com, 0, com)
    s.mass[0] = mass
    _set3(s.inv_inertia, 0, (inv_inertia,) * 3)
    s.k_pos[0], s.k_vel[0], s.k_limit[0], s.k_ang_damp[0] = k_pos, k_vel, k_limit, k_ang_damp
    s.dof_damping[0] = dof_damping
    s.dof_lo[0], s.dof_hi[0] = lo, hi
    s.act_dof[0], s.act_gear[0], s.act_lo[0], s.act_hi[0] = 0, gear, -1.0, 1.0
    return s


def ctx_rows(n, gravity=-9.81, friction=1.0, elasticity=0.0, ang_damping=0.0):
    rows = np.zeros((n, 4))
    rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3] = gravity, friction, elasticity, ang_damping
    return rows.astype(np.float32).astype(np.float64)


def body_state(p=(0, 0, 0), r=(1, 0, 0, 0), v=(0, 0, 0), w=(0, 0, 0)):
    return np.array([*p, *r, *v, *w], dtype=np.float64)


# ------------------------------------------------------------------ back ends
class OracleRunner:
    """the fp64 C restatement, one env step (= sys.n_frames substeps) per call"""
    name = "oracle"

    def __init__(self, sys_t, rows, state0):
        from oracle import brax as B
        from oracle import oracle as O

        n = len(state0)
        self.eng = B.Engine(sys_t, rows, n, selector=O.SEL_STATIC, ctx_idx0=np.arange(n), autoreset=False, max_steps=BIG)
        self.eng.reset()
        self.eng.state[:] = np.asarray(state0, np.float64).reshape(n, -1)
        self.n_act = sys_t.n_act

    def step(self, action):
        a = np.broadcast_to(np.asarray(action, np.float32), (self.eng.n, self.n_act))
        self.eng.step(a)
        return self.eng.state.reshape(self.eng.n, -1, 13).copy()


class EngineRunner:
    """the HIP kernel through the C ABI"""
    name = "hip"

    def __init__(self, sys_t, rows, state0, device="cuda"):
        import torch

        from carl_amd.brax_engine import BraxVecEngine
        from oracle import oracle as O

        n = len(state0)
        self.torch = torch
        self.eng = BraxVecEngine(sys_t, len(CTX_NAMES), rows, n, device, selector=O.SEL_STATIC, ctx_idx0=np.arange(n),
                                 auto_reset=False, max_episode_steps=BIG)
        self.eng.reset()
        self.eng.set_state64(np.asarray(state0, np.float64).reshape(n, -1, 13))
        self.n_act = sys_t.n_act

    def step(self, action):
        a = np.broadcast_to(np.asarray(action, np.float32), (self.eng.n, self.n_act)).copy()
        self.eng.step(self.torch.as_tensor(a))
        return self.eng.state64().cpu().numpy()


# ------------------------------------------------------------------ closed forms
def oscillator_matrix_power(k, c, mu, dt, n):
    """semi-implicit Euler on mu u'' = -k u - c u': v' = v + dt (-k u - c v) / mu, u' = u + dt v'"""
    a = 1.0 - dt * c / mu
    M = np.array([[1.0 - dt * dt * k / mu, dt * a], [-dt * k / mu, a]])
    return np.linalg.matrix_power(M, n)


def hinge_angle(state_row):
    """rotation angle of a body about its hinge axis from its quaternion (w, x, y, z): 2 atan2(|u|, w), signed by
    the axis component that carries the rotation"""
    w, x, y, z = state_row[3:7]
    k = int(np.argmax(np.abs([x, y, z])))
    return 2.0 * math.atan2([x, y, z][k], w)
