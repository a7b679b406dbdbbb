"""spring_ref.py -- TEST INFRASTRUCTURE (oracle), not product code.

A SECOND, independently written restatement of the joint pass of the spring pipeline
(`brax.spring.joints.resolve` as reached from carl/envs/brax/carl_brax_env.py:117,163-167 -> n_frames x
brax.spring.pipeline.step; brax 0.12.1 is not in the reference tree [upstream-memory]), in NumPy, next to the C one
in oracle/brax_spring.c -- what oracle/ref_style.py is for the classic-control step functions.  The two were written
from the specification in brax_spring.c's header, not from each other: this one works with ROTATION MATRICES (joint
frames as 3 x 3 bases, the relative rotation as J_p^T J_c, hinge angles read off matrix entries, multi-hinge rates
from a 3 x 3 linear solve), the C one with quaternions (relative quaternion, closed-form rate formulas).  A slip in
either -- a sign, a frame, an anchor on the wrong body -- shows up as a disagreement in
tests/test_brax_oracle.py::test_joint_wrenches_agree_with_the_independent_numpy_restatement.

PARITY UNPINNED against brax itself, like the C file.
"""
import numpy as np


def _mat(q):
    """rotation matrix of a quaternion (w, x, y, z), normalised"""
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _mat_raw(q):
    """the same without normalising (the pipeline multiplies quaternions as stored: float32 table entries are unit
    only to ~1e-8, and the C restatement does not renormalise them either)"""
    w, x, y, z = np.asarray(q, dtype=np.float64)
    n = w * w + x * x + y * y + z * z
    # q v q^-1 for a non-unit q, written as v + 2 w (u x v) + 2 u x (u x v): equals (2 - n) I-part + ... ; build it
    # column by column from that formula so that it matches "rotate by the stored quaternion" exactly
    u = np.array([x, y, z])
    cols = []
    for e in np.eye(3):
        t = 2.0 * np.cross(u, e)
        cols.append(e + w * t + np.cross(u, t))
    del n
    return np.array(cols).T


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def joint_wrenches(sys_t, state, tau, stiffness_scale=1.0):
    """state [L, 13] (COM position 3, rotation w x y z, linear velocity 3, angular velocity 3), tau [n_dof] ->
    (F [L, 3], T [L, 3]) net joint force and torque about the COM per link, world frame.  Pair contacts of the
    push task are not part of the joint pass proper and are left out (the C export includes them: compare on models
    without `n_pair`, or with the gripper away from the puck)."""
    s = sys_t
    L = s.n_links
    st = np.asarray(state, dtype=np.float64).reshape(L, 13)
    F, T = np.zeros((L, 3)), np.zeros((L, 3))
    f3 = lambda a: np.array([a[0], a[1], a[2]], dtype=np.float64)  # noqa: E731
    for i in range(L):
        P = s.parent[i]
        if P < 0 and s.n_link_dof[i] == 6:
            continue  # free root: no joint
        pc, vc, wc = st[i, 0:3], st[i, 7:10], st[i, 10:13]
        Rc = _mat_raw(st[i, 3:7])
        if P >= 0:
            pp, vp, wp, Rp, com_p = st[P, 0:3], st[P, 7:10], st[P, 10:13], _mat_raw(st[P, 3:7]), f3(s.com[P])
        else:  # the static world
            pp, vp, wp, Rp, com_p = np.zeros(3), np.zeros(3), np.zeros(3), np.eye(3), np.zeros(3)
        Ml, Mj = _mat_raw(s.link_rot[i]), _mat_raw(s.joint_rot[i])
        a = f3(s.joint_pos[i])
        # anchors: on the child at joint_pos of its frame; on the parent where that point sits at zero joint
        # displacement (link_pos + link_rot a in the parent frame); both relative to the bodies' COMs
        arm_c = Rc @ (a - f3(s.com[i]))
        arm_p = Rp @ (f3(s.link_pos[i]) + Ml @ a - com_p)
        A_c, A_p = pc + arm_c, pp + arm_p
        dA = A_p - A_c
        dV = (vp + np.cross(wp, arm_p)) - (vc + np.cross(wc, arm_c))
        kp = float(s.k_pos[i]) * stiffness_scale
        ns = s.n_slide[i]
        d0 = s.dof_start[i]
        force = np.zeros(3)
        for k in range(ns):  # prismatic directions: carried by the parent, free of the constraint spring
            ax = Rp @ f3(s.slide_axis[i][k])
            x, xd = -(dA @ ax), -(dV @ ax)  # slide coordinate and rate (child relative to parent along ax)
            dA, dV = dA + x * ax, dV + xd * ax
            fa = tau[d0 + k] - s.dof_damping[d0 + k] * xd - s.dof_stiffness[d0 + k] * x
            fa += s.k_limit[i] * max(0.0, s.dof_lo[d0 + k] - x) - s.k_limit[i] * max(0.0, x - s.dof_hi[d0 + k])
            force += fa * ax
        force += kp * dA + s.k_vel[i] * dV
        # joint frames as bases (columns = x, y, z of the frame in the world)
        Jc, Jp = Rc @ Mj, Rp @ Ml @ Mj
        Rel = Jp.T @ Jc  # child frame seen from the parent-side frame
        w_rel = wc - wp
        nr = s.n_link_dof[i] - ns
        d = d0 + ns
        if nr == 1:
            xc, xp = Jc[:, 0], Jp[:, 0]
            # twist about x from the relative quaternion's (w, x): w = sqrt(1 + tr) / 2 >= 0, x = (R21 - R12) / (4 w)
            qw = 0.5 * np.sqrt(max(1.0 + np.trace(Rel), 0.0))
            qx = (Rel[2, 1] - Rel[1, 2]) / (4.0 * qw)
            theta = 2.0 * np.arctan2(qx, qw)
            rate = xc @ w_rel
            ta = tau[d] - s.dof_damping[d] * rate - s.dof_stiffness[d] * theta
            ta += s.k_limit[i] * max(0.0, s.dof_lo[d] - theta) - s.k_limit[i] * max(0.0, theta - s.dof_hi[d])
            torque = kp * np.cross(xc, xp) + ta * xc
        else:
            # Rel = Rx(al) Ry(be) Rz(ga)
            al = np.arctan2(-Rel[1, 2], Rel[2, 2])
            be = np.arcsin(np.clip(Rel[0, 2], -1.0, 1.0))
            ga = np.arctan2(-Rel[0, 1], Rel[0, 0])
            sg = float(s.dof_sign3[i]) if nr == 3 else 1.0
            ang = [al, be, sg * ga]
            B1 = Jp @ _rx(al)
            B2 = B1 @ _ry(be)
            axes = np.stack([Jp[:, 0], B1[:, 1], sg * B2[:, 2]], axis=1)  # columns: current hinge axes
            if nr == 3:
                rates = np.linalg.solve(axes, w_rel)  # w_rel = sum_k rate_k axis_k
            else:
                rates = axes.T @ w_rel  # free dofs about orthogonal axes; the rest is violation, damped below
            torque = np.zeros(3)
            for k in range(3):
                if k < nr:
                    dk = d + k
                    ta = tau[dk] - s.dof_damping[dk] * rates[k] - s.dof_stiffness[dk] * ang[k]
                    ta += s.k_limit[i] * max(0.0, s.dof_lo[dk] - ang[k]) - s.k_limit[i] * max(0.0, ang[k] - s.dof_hi[dk])
                else:
                    ta = -kp * ang[k]  # a missing hinge is locked by the constraint spring on its angle
                torque += ta * axes[:, k]
        torque -= s.k_ang_damp[i] * w_rel
        F[i] += force
        T[i] += np.cross(arm_c, force) + torque
        if P >= 0:
            F[P] -= force
            T[P] -= np.cross(arm_p, force) + torque
    return F, T
