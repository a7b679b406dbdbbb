/* brax_spring.c -- TEST INFRASTRUCTURE (oracle), not product code.
 *
 * CPU restatement, in float64, of the Brax "spring" path CARL's locomotion envs run:
 *   brax.envs.<env>.reset/step  ->  n_frames x brax.spring.pipeline.step
 *   (actuator.to_tau -> spring.joints.resolve -> semi-implicit Euler ->
 *    spring.collisions.resolve -> spring.integrator.integrate -> com.to_world ->
 *    kinematics.world_to_joint / inverse)
 * reached from carl/envs/brax/carl_brax_env.py:163-190 (backend = "spring", :117) through
 * carl/envs/brax/wrappers.py:54-78, with the context -> System mapping of
 * carl_brax_env.py:255-292 in its INTENDED form (SURVEY.md Quirk B1: in the reference the
 * rebuilt System never reaches the jitted step).
 *
 * brax==0.12.1 (pyproject.toml:62-65) is NOT vendored in /root/reference and not
 * installable here; neither are jax/mujoco.  Everything below is [upstream-memory]:
 * the structure follows SURVEY.md section 8a "Brax restatement"; coefficients and the exact
 * form of the joint constraint are this build's reconstruction.  PARITY UNPINNED against brax --
 * the reference's tests hold no Brax step value (test/test_brax_env.py:8-23 is a smoke test).
 * What this file pins is the HIP kernel against an independent fp64 implementation of the
 * same specification; what pins THIS file: ten analytic known-answer cases (closed forms of the
 * scheme and of the physics: free fall, restitution, the reduced-mass oscillator, the physical
 * pendulum's period, actuator + damping series, limit equilibrium, Coulomb stop, torque-free spin,
 * the push task's puck sliding to rest on the table after v0^2 / (2 mu g), its fork resting on the table plane:
 * tests/test_brax_physics_kat.py), a second, independently written NumPy restatement of the joint
 * pass (oracle/spring_ref.py) and physical invariants (tests/test_brax_oracle.py).
 *
 * Specification (per substep dt, all vectors in the world frame, state per link =
 * COM position p, rotation r, linear velocity v, angular velocity w):
 *  1. tau_k = gear_k * clip(act_k, ctrl range)                       (actuator.to_tau)
 *  2. per non-root link c with parent p (spring.joints.resolve):
 *       anchors A_c, A_p; F = k_pos (A_p - A_c) + k_vel (vA_p - vA_c) on c at A_c, -F on p at A_p
 *       hinge axis alignment: T = k_pos (x_c cross x_p)
 *       about the axis n: tau - dof_damping * thetadot - k_stiff * theta + limit spring
 *       relative angular damping: -k_ang_damp (w_c - w_p)
 *  3. v += dt (F / m + g);  w += dt R diag(inv_inertia) R^T T             (no gyroscopic term)
 *  4. contacts, sphere vs plane z = plane_z (0: the ground; the push task's table) (spring.collisions.resolve): for penetrating,
 *     approaching points an impulse with restitution `elasticity`, Baumgarte term
 *     erp * depth / dt and Coulomb friction (<= friction * normal impulse), averaged over the
 *     link's active contacts
 *  5. v, w damped by exp(vel_damping dt), exp(ang_damping dt), + contact deltas;
 *     p += dt v;  r = normalize(r + dt/2 * (0, w) (x) r)                   (integrator.integrate)
 *  6. q, qd by inverse kinematics (root pose/twist, hinge angles / rates)  (kinematics.inverse)
 * Env level (brax.envs.ant.Ant.step/reset): reward = forward_weight * dx/dt_env + healthy
 * - ctrl_cost_weight |a|^2, done when torso z leaves [healthy_z_lo, healthy_z_hi],
 * obs = q[2:] ++ qd, reset q = init_q + U(+-noise), qd = vel_scale * N(0,1);
 * brax EpisodeWrapper(1000) truncation.
 */
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/carl_amd.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
  int32_t family, n_lanes, n_contexts, max_steps, selector, selector_stride, autoreset, cartpole_recompute;
  int64_t lane_offset;
  uint64_t seed;
} oracle_cfg_t;

void oracle_lane_words(uint64_t seed, uint64_t glane, uint32_t episode, uint32_t sub, uint32_t out[4]);
float oracle_u01(uint32_t w);

#define L_MAX CARL_BRAX_MAX_LINKS

typedef struct { double x, y, z; } v3;
typedef struct { double w, x, y, z; } qt;

static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vscale(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static qt qmul(qt a, qt b) {
  qt r = {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
  return r;
}
static qt qconj(qt a) { qt r = {a.w, -a.x, -a.y, -a.z}; return r; }
static qt qnorm(qt a) {
  const double n = sqrt(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z);
  qt r = {a.w / n, a.x / n, a.y / n, a.z / n};
  return r;
}
static v3 qrot(qt q, v3 v) { /* v + 2 w (u x v) + 2 u x (u x v) */
  const v3 u = V(q.x, q.y, q.z);
  const v3 t = vscale(vcross(u, v), 2.0);
  return vadd(vadd(v, vscale(t, q.w)), vcross(u, t));
}
static qt qaxis(int k, double angle) { /* rotation about basis axis k */
  qt r = {cos(0.5 * angle), 0, 0, 0};
  const double s = sin(0.5 * angle);
  if (k == 0) r.x = s; else if (k == 1) r.y = s; else r.z = s;
  return r;
}
static v3 f3(const float* p) { return V(p[0], p[1], p[2]); }
static qt f4(const float* p) { qt r = {p[0], p[1], p[2], p[3]}; return r; }

/* per-lane context view: carl_brax_env.py:255-292 in its intended form */
typedef struct {
  double gravity_z, friction, elasticity, ang_damping, stiffness_scale;
  double mass[L_MAX];
  double goal[3]; /* push task */
} lane_ctx;

static lane_ctx make_ctx(const carl_brax_sys_t* s, const double* row) {
  lane_ctx c;
  const carl_brax_ctx_map_t* m = &s->ctx;
  c.gravity_z = (m->gravity >= 0) ? (double)(float)row[m->gravity] : s->gravity_z;
  c.friction = (m->friction >= 0) ? (double)(float)row[m->friction] : s->friction;
  c.elasticity = (m->elasticity >= 0) ? (double)(float)row[m->elasticity] : s->elasticity;
  c.ang_damping = (m->ang_damping >= 0) ? (double)(float)row[m->ang_damping] : s->ang_damping;
  c.stiffness_scale = (m->joint_stiffness_scale >= 0) ? (double)(float)row[m->joint_stiffness_scale] : 1.0;
  for (int i = 0; i < s->n_links; ++i) c.mass[i] = s->mass[i];
  int n_light = 0; /* stability clamp (carl_brax_ctx_map_t::mass_ratio_floor): the higher floor when >= 2 links are light */
  for (int k = 0; k < m->n_mass; ++k) n_light += ((float)row[m->mass_row[k]] / m->mass_nominal[k] < 0.999f) ? 1 : 0;
  for (int k = 0; k < m->n_mass; ++k)
    c.mass[m->mass_link[k]] = s->mass[m->mass_link[k]] *
                              fmax((double)(float)row[m->mass_row[k]] / m->mass_nominal[k],
                                   (double)(n_light >= 2 ? m->mass_ratio_floor_multi[k] : m->mass_ratio_floor[k]));
  for (int k = 0; k < 3; ++k)
    c.goal[k] = (s->push_link > 0 && m->goal_position[k] >= 0) ? (double)(float)row[m->goal_position[k]] : s->push_goal[k];
  return c;
}

typedef struct { v3 p; qt r; v3 v; v3 w; } body;

static void load_bodies(const carl_brax_sys_t* s, const double* st, body* b) {
  for (int i = 0; i < s->n_links; ++i) {
    const double* x = st + 13 * i;
    b[i].p = V(x[0], x[1], x[2]);
    b[i].r.w = x[3]; b[i].r.x = x[4]; b[i].r.y = x[5]; b[i].r.z = x[6];
    b[i].v = V(x[7], x[8], x[9]);
    b[i].w = V(x[10], x[11], x[12]);
  }
}
static void store_bodies(const carl_brax_sys_t* s, const body* b, double* st) {
  for (int i = 0; i < s->n_links; ++i) {
    double* x = st + 13 * i;
    x[0] = b[i].p.x; x[1] = b[i].p.y; x[2] = b[i].p.z;
    x[3] = b[i].r.w; x[4] = b[i].r.x; x[5] = b[i].r.y; x[6] = b[i].r.z;
    x[7] = b[i].v.x; x[8] = b[i].v.y; x[9] = b[i].v.z;
    x[10] = b[i].w.x; x[11] = b[i].w.y; x[12] = b[i].w.z;
  }
}

/* the static world as a parent body (planar roots are jointed to it) */
static const body WORLD = {{0, 0, 0}, {1, 0, 0, 0}, {0, 0, 0}, {0, 0, 0}};

/* kinematics.forward + com.from_world: (q, qd) -> per-link COM state */
static void forward_kinematics(const carl_brax_sys_t* s, const double* q, const double* qd, body* b) {
  v3 org[L_MAX], ovel[L_MAX];
  for (int i = 0; i < s->n_links; ++i) {
    const int P = s->parent[i];
    qt rot; v3 o, vel, ang;
    if (P < 0 && s->n_link_dof[i] == 6) { /* free joint: q = (pos, quat), qd = (vel, ang), world frame */
      const double* qi = q + s->q_start[i];
      const double* di = qd + s->dof_start[i];
      qt qq = {qi[3], qi[4], qi[5], qi[6]};
      rot = qnorm(qq);
      o = V(qi[0], qi[1], qi[2]);
      vel = V(di[0], di[1], di[2]);
      ang = V(di[3], di[4], di[5]);
    } else {
      const body bp = (P < 0) ? WORLD : b[P];
      const v3 o_p = (P < 0) ? V(0, 0, 0) : org[P];
      const v3 ov_p = (P < 0) ? V(0, 0, 0) : ovel[P];
      const int ns = s->n_slide[i], nr = s->n_link_dof[i] - ns;
      const qt jr = f4(s->joint_rot[i]);
      const qt lrot = f4(s->link_rot[i]);
      const qt rpj = qmul(qmul(bp.r, lrot), jr); /* parent-side joint frame in the world */
      /* hinges stack intrinsically about the joint frame's x, y, +-z */
      qt rj = {1, 0, 0, 0};
      v3 wj = V(0, 0, 0);
      for (int k = 0; k < nr; ++k) {
        const double sg = (k == 2) ? (double)s->dof_sign3[i] : 1.0;
        const v3 axis = vscale(qrot(qmul(rpj, rj), V(k == 0, k == 1, k == 2)), sg);
        wj = vadd(wj, vscale(axis, qd[s->dof_start[i] + ns + k]));
        rj = qmul(rj, qaxis(k, sg * q[s->q_start[i] + ns + k]));
      }
      const qt rl = qmul(qmul(jr, rj), qconj(jr)); /* joint rotation in child coordinates */
      const v3 a = f3(s->joint_pos[i]);
      v3 lpos = vadd(f3(s->link_pos[i]), qrot(lrot, vsub(a, qrot(rl, a)))); /* the anchor stays put */
      v3 slide_vel = V(0, 0, 0);
      for (int k = 0; k < ns; ++k) {
        const v3 ax = f3(s->slide_axis[i][k]);
        lpos = vadd(lpos, vscale(ax, q[s->q_start[i] + k]));
        slide_vel = vadd(slide_vel, vscale(qrot(bp.r, ax), qd[s->dof_start[i] + k]));
      }
      rot = qmul(bp.r, qmul(lrot, rl));
      o = vadd(o_p, qrot(bp.r, lpos));
      const v3 anchor_w = vadd(o, qrot(rot, a));
      ang = vadd(bp.w, wj);
      vel = vadd(vadd(ov_p, vcross(bp.w, vsub(o, o_p))), vadd(slide_vel, vcross(wj, vsub(o, anchor_w))));
    }
    org[i] = o; ovel[i] = vel;
    b[i].r = rot; b[i].w = ang;
    const v3 c = qrot(rot, f3(s->com[i]));
    b[i].p = vadd(o, c);
    b[i].v = vadd(vel, vcross(ang, c));
  }
}

/* joint geometry shared by inverse kinematics and joints.resolve */
typedef struct {
  v3 A_c, A_p, vA_c, vA_p, x_c, x_p, wrel;
  double theta, thetadot;        /* single hinge */
  v3 axis[3];                    /* multi-hinge joints: current world axes ... */
  double ang[3], rate[3];        /* ... Euler x-y-z angles (third signed) and their rates */
} joint_geom;

static joint_geom joint_geometry(const carl_brax_sys_t* s, int i, const body* bc, const body* bp) {
  joint_geom g;
  const int P = s->parent[i];
  const v3 a = f3(s->joint_pos[i]);
  const qt lrot = f4(s->link_rot[i]), jr = f4(s->joint_rot[i]);
  const v3 com_p = (P < 0) ? V(0, 0, 0) : f3(s->com[P]);
  const v3 o_c = vsub(bc->p, qrot(bc->r, f3(s->com[i])));
  const v3 o_p = vsub(bp->p, qrot(bp->r, com_p));
  g.A_c = vadd(o_c, qrot(bc->r, a));
  g.A_p = vadd(o_p, qrot(bp->r, vadd(f3(s->link_pos[i]), qrot(lrot, a)))); /* at zero slide */
  g.vA_c = vadd(bc->v, vcross(bc->w, vsub(g.A_c, bc->p)));
  g.vA_p = vadd(bp->v, vcross(bp->w, vsub(g.A_p, bp->p)));
  const qt rc = qmul(bc->r, jr);
  const qt rp = qmul(qmul(bp->r, lrot), jr);
  g.x_c = qrot(rc, V(1, 0, 0));
  g.x_p = qrot(rp, V(1, 0, 0));
  qt rel = qmul(qconj(rp), rc);
  if (rel.w < 0) { rel.w = -rel.w; rel.x = -rel.x; }
  g.theta = 2.0 * atan2(rel.x, rel.w); /* twist about the hinge (joint frame x) */
  g.wrel = vsub(bc->w, bp->w);
  g.thetadot = vdot(g.x_c, g.wrel);
  const int nr = s->n_link_dof[i] - s->n_slide[i];
  if (nr != 1) { /* rel = Rx(al) Ry(be) Rz(ga): decompose, ga = sign * theta_3 (nr = 0: all three locked) */
    const double R00 = 1 - 2 * (rel.y * rel.y + rel.z * rel.z), R01 = 2 * (rel.x * rel.y - rel.w * rel.z);
    double R02 = 2 * (rel.x * rel.z + rel.w * rel.y);
    const double R12 = 2 * (rel.y * rel.z - rel.w * rel.x), R22 = 1 - 2 * (rel.x * rel.x + rel.y * rel.y);
    if (R02 > 1) R02 = 1;
    if (R02 < -1) R02 = -1;
    const double al = atan2(-R12, R22), be = asin(R02), ga = atan2(-R01, R00);
    const double sg = (nr == 3) ? s->dof_sign3[i] : 1.0;
    g.ang[0] = al; g.ang[1] = be; g.ang[2] = sg * ga;
    g.axis[0] = g.x_p;
    const qt rx = qaxis(0, al);
    g.axis[1] = qrot(qmul(rp, rx), V(0, 1, 0));
    g.axis[2] = vscale(qrot(qmul(qmul(rp, rx), qaxis(1, be)), V(0, 0, 1)), sg);
    const double w0 = vdot(g.wrel, g.axis[0]), w1 = vdot(g.wrel, g.axis[1]), w2 = vdot(g.wrel, g.axis[2]);
    g.rate[1] = w1;
    if (nr == 3) { /* axis0 and axis2 are not orthogonal: axis0 . axis2 = sign * sin(be) */
      const double c = vdot(g.axis[0], g.axis[2]), den = 1 - c * c;
      g.rate[0] = (w0 - c * w2) / den;
      g.rate[2] = (w2 - c * w0) / den;
    } else {
      g.rate[0] = w0;
      g.rate[2] = w2; /* locked direction: any rate here is constraint violation, damped below */
    }
  }
  return g;
}

/* kinematics.world_to_joint + inverse: per-link COM state -> (q, qd) */
static void inverse_kinematics(const carl_brax_sys_t* s, const body* b, double* q, double* qd) {
  for (int i = 0; i < s->n_links; ++i) {
    const int P = s->parent[i];
    if (P < 0 && s->n_link_dof[i] == 6) {
      const v3 c = qrot(b[i].r, f3(s->com[i]));
      const v3 o = vsub(b[i].p, c);
      const v3 vel = vsub(b[i].v, vcross(b[i].w, c));
      double* qi = q + s->q_start[i];
      double* di = qd + s->dof_start[i];
      qi[0] = o.x; qi[1] = o.y; qi[2] = o.z;
      qi[3] = b[i].r.w; qi[4] = b[i].r.x; qi[5] = b[i].r.y; qi[6] = b[i].r.z;
      di[0] = vel.x; di[1] = vel.y; di[2] = vel.z;
      di[3] = b[i].w.x; di[4] = b[i].w.y; di[5] = b[i].w.z;
    } else {
      const body bp = (P < 0) ? WORLD : b[P];
      const joint_geom g = joint_geometry(s, i, &b[i], &bp);
      const int ns = s->n_slide[i];
      for (int k = 0; k < ns; ++k) {
        const v3 ax = qrot(bp.r, f3(s->slide_axis[i][k]));
        q[s->q_start[i] + k] = vdot(vsub(g.A_c, g.A_p), ax);
        qd[s->dof_start[i] + k] = vdot(vsub(g.vA_c, g.vA_p), ax);
      }
      const int nr = s->n_link_dof[i] - ns;
      if (nr == 1) {
        q[s->q_start[i] + ns] = g.theta;
        qd[s->dof_start[i] + ns] = g.thetadot;
      } else {
        for (int k = 0; k < nr; ++k) {
          q[s->q_start[i] + ns + k] = g.ang[k];
          qd[s->dof_start[i] + ns + k] = g.rate[k];
        }
      }
    }
  }
}

static v3 apply_inv_inertia(const carl_brax_sys_t* s, int i, qt r, v3 t) {
  const v3 l = qrot(qconj(r), t);
  return qrot(r, V(l.x * s->inv_inertia[i][0], l.y * s->inv_inertia[i][1], l.z * s->inv_inertia[i][2]));
}

/* one brax.spring.pipeline.step.  hc / hl (nullable, per link): running hashes h <- 33 h + bits of the DISCRETE
 * decisions taken -- hc: which of the link's collision spheres delivered an impulse (bit = ordinal of the sphere
 * among the link's spheres), hl: which range limits of the link's joint were active (slides: bits 0-3, hinges: bits
 * 4-9; below / above per dof).  A float32 implementation of the same arithmetic can only agree to rounding where
 * it took the same contact decisions (an impulse is discontinuous in the state): tests compare these hashes with
 * the kernel's (carl_step_io_t::branch_sig) and assert the tolerance on the agreeing lanes. */
/* spring.joints.resolve: net force F[i] and torque T[i] (about the COM, world frame) on every link from the joint
 * springs, dampers, limits, actuators (and the push task's pair contacts) */
static void joint_wrenches(const carl_brax_sys_t* s, const lane_ctx* c, const double* tau, const body* b, v3* F, v3* T,
                           uint32_t* hl, uint32_t* pair_fired) {
  const int L = s->n_links;
  if (pair_fired) *pair_fired = 0;
  for (int i = 0; i < L; ++i) { F[i] = V(0, 0, 0); T[i] = V(0, 0, 0); }
  for (int i = 0; i < L; ++i) {
    const int P = s->parent[i];
    if (P < 0 && s->n_link_dof[i] == 6) continue;
    const body bp = (P < 0) ? WORLD : b[P];
    const joint_geom g = joint_geometry(s, i, &b[i], &bp);
    const double kp = s->k_pos[i] * c->stiffness_scale;
    v3 e = vsub(g.A_p, g.A_c), ev = vsub(g.vA_p, g.vA_c);
    v3 f = V(0, 0, 0);
    uint32_t lim = 0;
    const int ns = s->n_slide[i], d0 = s->dof_start[i];
    for (int k = 0; k < ns; ++k) { /* prismatic dofs: free along the axis, own spring/damper/force */
      const v3 ax = qrot(bp.r, f3(s->slide_axis[i][k]));
      const double qk = -vdot(e, ax), qdk = -vdot(ev, ax);
      e = vadd(e, vscale(ax, qk));
      ev = vadd(ev, vscale(ax, qdk));
      double fa = tau[d0 + k] - s->dof_damping[d0 + k] * qdk - s->dof_stiffness[d0 + k] * qk;
      if (qk < s->dof_lo[d0 + k]) { fa += s->k_limit[i] * (s->dof_lo[d0 + k] - qk); lim |= 1u << (2 * k); } /* range of the slide */
      if (qk > s->dof_hi[d0 + k]) { fa -= s->k_limit[i] * (qk - s->dof_hi[d0 + k]); lim |= 2u << (2 * k); }
      f = vadd(f, vscale(ax, fa));
    }
    f = vadd(f, vadd(vscale(e, kp), vscale(ev, s->k_vel[i])));
    F[i] = vadd(F[i], f);
    T[i] = vadd(T[i], vcross(vsub(g.A_c, b[i].p), f));
    v3 t;
    const int d = d0 + ns, nr = s->n_link_dof[i] - ns;
    if (nr == 1) {
      t = vscale(vcross(g.x_c, g.x_p), kp); /* keep the hinge axes aligned */
      double ta = tau[d] - s->dof_damping[d] * g.thetadot - s->dof_stiffness[d] * g.theta;
      if (g.theta < s->dof_lo[d]) { ta += s->k_limit[i] * (s->dof_lo[d] - g.theta); lim |= 16u; }
      if (g.theta > s->dof_hi[d]) { ta -= s->k_limit[i] * (g.theta - s->dof_hi[d]); lim |= 32u; }
      t = vadd(t, vscale(g.x_c, ta));
    } else { /* 2 or 3 stacked hinges: per-dof torques about the current axes; a missing third
                dof is locked by the constraint spring on its Euler angle */
      t = V(0, 0, 0);
      for (int k = 0; k < 3; ++k) {
        double ta;
        if (k < nr) {
          const int dk = d + k;
          ta = tau[dk] - s->dof_damping[dk] * g.rate[k] - s->dof_stiffness[dk] * g.ang[k];
          if (g.ang[k] < s->dof_lo[dk]) { ta += s->k_limit[i] * (s->dof_lo[dk] - g.ang[k]); lim |= 16u << (2 * k); }
          if (g.ang[k] > s->dof_hi[dk]) { ta -= s->k_limit[i] * (g.ang[k] - s->dof_hi[dk]); lim |= 32u << (2 * k); }
        } else {
          ta = -kp * g.ang[k];
        }
        t = vadd(t, vscale(g.axis[k], ta));
      }
    }
    t = vsub(t, vscale(g.wrel, s->k_ang_damp[i]));
    if (hl) hl[i] = hl[i] * 33u + lim;
    T[i] = vadd(T[i], t);
    if (P >= 0) {
      F[P] = vsub(F[P], f);
      T[P] = vsub(T[P], vadd(vcross(vsub(g.A_p, b[P].p), f), t));
    }
    if (s->n_pair > 0 && i == s->push_link) { /* the gripper's spheres against the object (carl_brax_sys_t::n_pair) */
      const int a = s->pair_link;
      const v3 o = vsub(b[i].p, qrot(b[i].r, f3(s->com[i])));
      for (int k = 0; k < s->n_pair; ++k) {
        const v3 rel = qrot(b[a].r, vsub(f3(s->pair_pos[k]), f3(s->com[a])));
        const v3 cs = vadd(b[a].p, rel);
        const double rk = s->pair_radius[k];
        if (!(fabs(cs.z - o.z) < s->pair_obj_half + rk)) continue;
        const double dx = o.x - cs.x, dy = o.y - cs.y;
        const double dist = sqrt(dx * dx + dy * dy);
        const double depth = rk + s->pair_obj_radius - dist;
        if (!(depth > 0.0) || !(dist > 1e-9)) continue;
        const v3 n = V(dx / dist, dy / dist, 0.0);
        const v3 vs = vadd(b[a].v, vcross(b[a].w, rel));
        const double closing = vdot(vsub(vs, b[i].v), n);
        const double fm = s->pair_k * depth + s->pair_c * closing;
        if (!(fm > 0.0)) continue;
        /* a DISCRETE decision (the damper term makes the force jump where the contact opens or closes): recorded in the
         * object's contact hash (bits 16 + k), so that a parity check can set lanes aside where it flipped within rounding */
        if (pair_fired) *pair_fired |= 1u << (16 + (k & 15));
        v3 fc = vscale(n, fm);
        if (s->pair_ct > 0.0f) { /* Coulomb friction, regularised (carl_amd.h: pair_ct): the object is dragged along the
                                  * sphere's tangential velocity, never harder than friction x the normal force --
                                  * what brax's contact friction does between the gripper and puck geoms
                                  * [upstream-memory: brax/envs/assets/pusher.xml gives both geoms friction] */
          const v3 vr = vsub(vs, b[i].v);
          v3 vt = vsub(vr, vscale(n, closing));
          vt.z = 0.0; /* the object's free directions are horizontal: the table carries the vertical part */
          const double vt_len = sqrt(vdot(vt, vt));
          if (vt_len > 1e-9) {
            double ft = s->pair_ct * vt_len;
            const double cap = c->friction * fm;
            if (ft > cap) ft = cap;
            fc = vadd(fc, vscale(vt, ft / vt_len));
          }
        }
        F[i] = vadd(F[i], fc);
        F[a] = vsub(F[a], fc);
        T[a] = vsub(T[a], vcross(rel, fc));
      }
    }
  }
}

static void substep(const carl_brax_sys_t* s, const lane_ctx* c, const double* tau, body* b, uint32_t* hc, uint32_t* hl) {
  v3 F[L_MAX], T[L_MAX];
  const int L = s->n_links;
  uint32_t pair_fired = 0;
  joint_wrenches(s, c, tau, b, F, T, hl, &pair_fired);
  /* --- semi-implicit Euler: velocity update before the collision pass -------------- */
  for (int i = 0; i < L; ++i) {
    b[i].v = vadd(b[i].v, vscale(vadd(vscale(F[i], 1.0 / c->mass[i]), V(0, 0, c->gravity_z)), s->dt));
    b[i].w = vadd(b[i].w, vscale(apply_inv_inertia(s, i, b[i].r, T[i]), s->dt));
  }
  /* --- the push task's object on the table (carl_amd.h: obj_support): Coulomb friction under the normal load m |g|,
   * as an impulse like the plane contacts' -- the horizontal velocity shrinks by friction |g| dt, at most to zero.
   * [upstream-memory: the puck of brax's pusher lies on the table geom; its two slides leave it no vertical freedom, so
   * whatever holds it up carries m |g|; with the MJCF's own gravity 0 0 0 there is no load and no friction, with CARL's
   * context default gravity = -9.8 (carl_pusher.py:17-21) there is] */
  if (s->obj_support && s->push_link > 0) {
    const int i = s->push_link;
    const double load = c->gravity_z < 0.0 ? -c->gravity_z : 0.0;
    const double vh = sqrt(b[i].v.x * b[i].v.x + b[i].v.y * b[i].v.y);
    if (vh > 1e-9) {
      double cut = c->friction * load * s->dt;
      if (cut > vh) cut = vh;
      b[i].v.x -= b[i].v.x / vh * cut;
      b[i].v.y -= b[i].v.y / vh * cut;
    }
  }
  /* --- spring.collisions.resolve: spheres vs the plane z = plane_z (0: the ground; the push task's table) ------ */
  v3 dv[L_MAX], dw[L_MAX];
  int cnt[L_MAX], seen[L_MAX];
  uint32_t hit[L_MAX];
  for (int i = 0; i < L; ++i) { dv[i] = V(0, 0, 0); dw[i] = V(0, 0, 0); cnt[i] = 0; seen[i] = 0; hit[i] = 0; }
  if (s->n_pair > 0 && s->push_link > 0) hit[s->push_link] |= pair_fired; /* the gripper / object contacts that pushed */
  const v3 n = V(0, 0, 1);
  for (int k = 0; k < s->n_coll; ++k) {
    const int i = s->coll_link[k];
    const int ordinal = seen[i]++; /* of this sphere among its link's spheres */
    const v3 o = vsub(b[i].p, qrot(b[i].r, f3(s->com[i])));
    const v3 ctr = vadd(o, qrot(b[i].r, f3(s->coll_pos[k])));
    const double depth = s->coll_radius[k] - (ctr.z - (double)s->plane_z); /* > 0: penetrating */
    if (!(depth > 0)) continue;
    const v3 pos = V(ctr.x, ctr.y, ctr.z - s->coll_radius[k]); /* lowest point of the sphere */
    const v3 r = vsub(pos, b[i].p);
    const v3 rel = vadd(b[i].v, vcross(b[i].w, r));
    const double vn = vdot(n, rel);
    const double inv_m = 1.0 / c->mass[i];
    const double ang = vdot(n, vcross(apply_inv_inertia(s, i, b[i].r, vcross(r, n)), r));
    const double baum = s->baumgarte_erp * depth / s->dt;
    const double imp = (-(1.0 + c->elasticity) * vn + baum) / (inv_m + ang);
    if (!(imp > 0) || !(vn < 0)) continue; /* only approaching contacts push */
    hit[i] |= 1u << ordinal;
    v3 J = vscale(n, imp);
    const v3 vt = vsub(rel, vscale(n, vn));
    const double vt_len = sqrt(vdot(vt, vt));
    if (vt_len > 1e-9) {
      const v3 dir = vscale(vt, 1.0 / vt_len);
      const double ang_d = vdot(dir, vcross(apply_inv_inertia(s, i, b[i].r, vcross(r, dir)), r));
      double imp_d = vt_len / (inv_m + ang_d);
      const double cap = c->friction * imp;
      if (imp_d > cap) imp_d = cap;
      J = vsub(J, vscale(dir, imp_d));
    }
    dv[i] = vadd(dv[i], vscale(J, inv_m));
    dw[i] = vadd(dw[i], apply_inv_inertia(s, i, b[i].r, vcross(r, J)));
    cnt[i] += 1;
  }
  /* --- spring.integrator.integrate --------------------------------------------------- */
  const double dl = exp(s->vel_damping * s->dt), da = exp(c->ang_damping * s->dt);
  for (int i = 0; i < L; ++i) {
    if (hc) hc[i] = hc[i] * 33u + hit[i];
    b[i].v = vscale(b[i].v, dl);
    b[i].w = vscale(b[i].w, da);
    if (cnt[i] > 0) {
      b[i].v = vadd(b[i].v, vscale(dv[i], 1.0 / cnt[i]));
      b[i].w = vadd(b[i].w, vscale(dw[i], 1.0 / cnt[i]));
    }
    b[i].p = vadd(b[i].p, vscale(b[i].v, s->dt));
    const qt wq = {0, b[i].w.x, b[i].w.y, b[i].w.z};
    const qt dq = qmul(wq, b[i].r);
    qt r2 = {b[i].r.w + 0.5 * s->dt * dq.w, b[i].r.x + 0.5 * s->dt * dq.x, b[i].r.y + 0.5 * s->dt * dq.y,
             b[i].r.z + 0.5 * s->dt * dq.z};
    b[i].r = qnorm(r2);
  }
}

/* whole-body centre of mass (brax.envs.humanoid.Humanoid._com) */
static v3 system_com(const carl_brax_sys_t* s, const lane_ctx* c, const body* b, double* mass_sum) {
  v3 com = V(0, 0, 0);
  double M = 0;
  for (int i = 0; i < s->n_links; ++i) { com = vadd(com, vscale(b[i].p, c->mass[i])); M += c->mass[i]; }
  *mass_sum = M;
  return vscale(com, 1.0 / M);
}

/* brax.envs.<env>._get_obs: q[skip:] ++ qd, and for the humanoid ++ com_inertia (L x 10) ++
 * com_velocity (L x 6) ++ qfrc_actuator (n_dof) */
static void observe(const carl_brax_sys_t* s, const lane_ctx* c, const body* b, const double* tau, float* obs) {
  double q[CARL_BRAX_MAX_Q], qd[CARL_BRAX_MAX_DOF];
  inverse_kinematics(s, b, q, qd);
  int k = 0;
  if (s->target_link > 0) { /* brax.envs.reacher._get_obs: cos(theta) ++ sin(theta) ++ goal q ++ arm qd ++ (tip - goal) */
    const int tl = s->tip_link, g = s->target_link, tq = s->q_start[g], td = s->dof_start[g];
    for (int i = 0; i < tq; ++i) obs[k++] = (float)cos((double)(float)q[i]);
    for (int i = 0; i < tq; ++i) obs[k++] = (float)sin((double)(float)q[i]);
    for (int i = tq; i < s->n_q; ++i) obs[k++] = (float)q[i];
    for (int i = 0; i < td; ++i) obs[k++] = (float)qd[i];
    const v3 tip = vadd(vsub(b[tl].p, qrot(b[tl].r, f3(s->com[tl]))), qrot(b[tl].r, f3(s->tip_offset)));
    const v3 goal = vsub(b[g].p, qrot(b[g].r, f3(s->com[g])));
    const v3 d = vsub(tip, goal);
    obs[k++] = (float)d.x; obs[k++] = (float)d.y; obs[k++] = (float)d.z;
    return;
  }
  if (s->push_link > 0) { /* brax.envs.pusher._get_obs: arm q ++ arm qd ++ COM of the end effector, the object, the goal */
    const int na = s->q_start[s->push_link];
    for (int i = 0; i < na; ++i) obs[k++] = (float)q[i];
    for (int i = 0; i < na; ++i) obs[k++] = (float)qd[i];
    const v3 tp = b[s->tip_link].p, op = b[s->push_link].p;
    obs[k++] = (float)tp.x; obs[k++] = (float)tp.y; obs[k++] = (float)tp.z;
    obs[k++] = (float)op.x; obs[k++] = (float)op.y; obs[k++] = (float)op.z;
    for (int j = 0; j < 3; ++j) obs[k++] = (float)c->goal[j];
    return;
  }
  if (s->obs_trig_from > 0) { /* q[:from] ++ sin(q[from:]) ++ cos(q[from:]) */
    for (int i = s->exclude_current_positions; i < s->obs_trig_from; ++i) obs[k++] = (float)q[i];
    for (int i = s->obs_trig_from; i < s->n_q; ++i) obs[k++] = (float)sin((double)(float)q[i]);
    for (int i = s->obs_trig_from; i < s->n_q; ++i) obs[k++] = (float)cos((double)(float)q[i]);
  } else
  for (int i = s->exclude_current_positions; i < s->n_q; ++i) obs[k++] = (float)q[i];
  for (int i = 0; i < s->n_dof; ++i) {
    float v = (float)qd[i];
    if (s->obs_qd_clip > 0.0f) v = v > s->obs_qd_clip ? s->obs_qd_clip : (v < -s->obs_qd_clip ? -s->obs_qd_clip : v);
    obs[k++] = v;
  }
  if (!s->obs_extended) return;
  double M;
  const v3 com = system_com(s, c, b, &M);
  for (int i = 0; i < s->n_links; ++i) { /* inertia about the system com, world axes, row-major, then mass */
    const v3 d = vsub(b[i].p, com);
    const double m = c->mass[i], dd = vdot(d, d);
    const double Ib[3] = {1.0 / s->inv_inertia[i][0], 1.0 / s->inv_inertia[i][1], 1.0 / s->inv_inertia[i][2]};
    const v3 ex = qrot(b[i].r, V(1, 0, 0)), ey = qrot(b[i].r, V(0, 1, 0)), ez = qrot(b[i].r, V(0, 0, 1));
    const double e[3][3] = {{ex.x, ey.x, ez.x}, {ex.y, ey.y, ez.y}, {ex.z, ey.z, ez.z}}; /* R */
    const double dv[3] = {d.x, d.y, d.z};
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        double v = 0;
        for (int j = 0; j < 3; ++j) v += e[r][j] * Ib[j] * e[cc][j];
        v += m * ((r == cc ? dd : 0.0) - dv[r] * dv[cc]);
        obs[k++] = (float)v;
      }
    obs[k++] = (float)m;
  }
  for (int i = 0; i < s->n_links; ++i) {
    const double f = c->mass[i] / M;
    obs[k++] = (float)(f * b[i].v.x); obs[k++] = (float)(f * b[i].v.y); obs[k++] = (float)(f * b[i].v.z);
    obs[k++] = (float)b[i].w.x; obs[k++] = (float)b[i].w.y; obs[k++] = (float)b[i].w.z;
  }
  for (int i = 0; i < s->n_dof; ++i) obs[k++] = (float)(tau ? tau[i] : 0.0);
}

/* brax.envs.<env>.reset: q = init_q + U(-noise, noise), qd = vel_scale * N(0, 1).
 * Draw k of a lane's reset uses word (k mod 4) of Philox block k/4 on sub-stream
 * 0x80000000 | block; normals are Box-Muller pairs from two consecutive draws. */
static double draw_u(uint64_t seed, uint64_t g, uint32_t ep, int k) {
  uint32_t w[4];
  oracle_lane_words(seed, g, ep, 0x80000000u | (uint32_t)(k >> 2), w);
  return (double)oracle_u01(w[k & 3]);
}

static void reset_lane(const carl_brax_sys_t* s, const lane_ctx* c, uint64_t seed, uint64_t g, uint32_t ep, body* b) {
  double q[CARL_BRAX_MAX_Q], qd[CARL_BRAX_MAX_DOF];
  int k = 0;
  for (int i = 0; i < s->n_q; ++i, ++k)
    q[i] = (double)s->init_q[i] + (double)s->reset_noise_scale * (2.0 * draw_u(seed, g, ep, k) - 1.0);
  if (s->reset_vel_uniform) { /* brax.envs.humanoid: qvel = U(-scale, scale) */
    for (int i = 0; i < s->n_dof; ++i, ++k) qd[i] = (double)s->reset_vel_scale * (2.0 * draw_u(seed, g, ep, k) - 1.0);
  } else
  for (int i = 0; i < s->n_dof; i += 2, k += 2) {
    const double u1 = draw_u(seed, g, ep, k), u2 = draw_u(seed, g, ep, k + 1);
    const double rad = sqrt(-2.0 * log(1.0 - u1));
    qd[i] = s->reset_vel_scale * rad * cos(2.0 * M_PI * u2);
    if (i + 1 < s->n_dof) qd[i + 1] = s->reset_vel_scale * rad * sin(2.0 * M_PI * u2);
  }
  if (s->target_link > 0) { /* brax.envs.reacher._random_target: uniform distance and bearing; goal at rest */
    const int tq = s->q_start[s->target_link], td = s->dof_start[s->target_link];
    const double dist = (double)s->target_max_dist * draw_u(seed, g, ep, s->n_q + s->n_dof);
    const double ang = 2.0 * M_PI * draw_u(seed, g, ep, s->n_q + s->n_dof + 1);
    q[tq] = dist * cos(ang);
    q[tq + 1] = dist * sin(ang);
    for (int i = td; i < s->n_dof; ++i) qd[i] = 0.0;
  }
  if (s->push_link > 0) { /* brax.envs.pusher.reset: object placed in a box in front of the arm, pushed out
                             of the goal's push_min_dist disc; at rest */
    const int pl = s->push_link, tq = s->q_start[pl], td = s->dof_start[pl];
    const double cx = s->push_lo[0] + ((double)s->push_hi[0] - s->push_lo[0]) * draw_u(seed, g, ep, s->n_q + s->n_dof);
    const double cy = s->push_lo[1] + ((double)s->push_hi[1] - s->push_lo[1]) * draw_u(seed, g, ep, s->n_q + s->n_dof + 1);
    double dx = s->link_pos[pl][0] + cx - c->goal[0], dy = s->link_pos[pl][1] + cy - c->goal[1];
    const double nrm = sqrt(dx * dx + dy * dy);
    if (nrm < s->push_min_dist) { const double sc = s->push_min_dist / (nrm > 1e-12 ? nrm : 1e-12); dx *= sc; dy *= sc; }
    q[tq] = c->goal[0] + dx - s->link_pos[pl][0];
    q[tq + 1] = c->goal[1] + dy - s->link_pos[pl][1];
    for (int i = td; i < s->n_dof; ++i) qd[i] = 0.0;
  }
  forward_kinematics(s, q, qd, b);
}

static int32_t select_ctx(const oracle_cfg_t* cfg, int32_t idx, uint64_t g, uint32_t episode) {
  const int32_t C = cfg->n_contexts;
  if (cfg->selector == 1) { int64_t v = ((int64_t)idx + cfg->selector_stride) % C; return (int32_t)(v < 0 ? v + C : v); }
  if (cfg->selector == 2) { uint32_t w[4]; oracle_lane_words(cfg->seed, g, episode, 1u, w);
                            return (int32_t)(((uint64_t)w[0] * (uint64_t)(uint32_t)C) >> 32); }
  return idx;
}

/* BraxWalkerGoalWrapper (carl/envs/brax/brax_walker_goal_wrapper.py:69-140): compass code ->
 * unit direction (direction_values :69-106) */
static void goal_direction(int code, double* dx, double* dy) {
  const double c = cos(22.5 * M_PI / 180.0), sn = sin(22.5 * M_PI / 180.0), h = sqrt(0.5);
  switch (code) {
    case 3: *dx = 0; *dy = -1; break;
    case 1: *dx = 0; *dy = 1; break;
    case 2: *dx = 1; *dy = 0; break;
    case 4: *dx = -1; *dy = 0; break;
    case 34: *dx = -h; *dy = -h; break;
    case 14: *dx = -h; *dy = h; break;
    case 32: *dx = h; *dy = -h; break;
    case 12: *dx = h; *dy = h; break;
    case 334: *dx = -c; *dy = -sn; break;
    case 434: *dx = -sn; *dy = -c; break;
    case 114: *dx = -c; *dy = sn; break;
    case 414: *dx = -sn; *dy = c; break;
    case 332: *dx = c; *dy = -sn; break;
    case 232: *dx = sn; *dy = -c; break;
    case 112: *dx = c; *dy = sn; break;
    case 212: *dx = sn; *dy = c; break;
    default: *dx = 0; *dy = 0; break;
  }
}

/* step() of the goal wrapper (:124-140) on the transition's own observation: returns the
 * progress reward, sets *success; pos is the integrated (x, y) */
static double goal_epilogue(const carl_brax_sys_t* s, const double* ctx_row, const float* obs, double* pos,
                            int* success) {
  double dx, dy;
  goal_direction((int)lrint((double)(float)ctx_row[s->ctx.target_direction]), &dx, &dy);
  const double dist = (double)(float)ctx_row[s->ctx.target_distance];
  const double gx = dx * dist, gy = dy * dist;
  const double nx = pos[0] + (double)obs[s->goal_obs_idx[0]] * s->goal_dt;
  const double ny = pos[1] + (double)obs[s->goal_obs_idx[1]] * s->goal_dt;
  const double cur = sqrt((gx - nx) * (gx - nx) + (gy - ny) * (gy - ny));
  const double prev = sqrt((gx - pos[0]) * (gx - pos[0]) + (gy - pos[1]) * (gy - pos[1]));
  pos[0] = nx; pos[1] = ny;
  *success = cur <= (double)(float)ctx_row[s->ctx.target_radius];
  const double r = prev - cur;
  return r > 0 ? r : 0;
}

/* exported: pure helpers for tests */
/* one step of the goal wrapper on one observation: returns the reward, updates pos[2], sets *success */
double obx_goal_step(const carl_brax_sys_t* s, const double* ctx_row, const float* obs, double* pos, int32_t* success) {
  int ok = 0;
  const double r = goal_epilogue(s, ctx_row, obs, pos, &ok);
  *success = ok;
  return r;
}
void obx_forward_kinematics(const carl_brax_sys_t* s, const double* q, const double* qd, double* state) {
  body b[L_MAX];
  forward_kinematics(s, q, qd, b);
  store_bodies(s, b, state);
}
void obx_inverse_kinematics(const carl_brax_sys_t* s, const double* state, double* q, double* qd) {
  body b[L_MAX];
  load_bodies(s, state, b);
  inverse_kinematics(s, b, q, qd);
}
/* n_sub pipeline substeps on one lane's state with constant joint torques (tests) */
void obx_substeps(const carl_brax_sys_t* s, const double* ctx_row, const double* tau, int n_sub, double* state) {
  body b[L_MAX];
  const lane_ctx c = make_ctx(s, ctx_row);
  load_bodies(s, state, b);
  for (int k = 0; k < n_sub; ++k) substep(s, &c, tau, b, NULL, NULL);
  store_bodies(s, b, state);
}

/* the joint wrenches of one state (tests: compared with an independent NumPy restatement, oracle/spring_ref.py) */
void obx_joint_wrenches(const carl_brax_sys_t* s, const double* ctx_row, const double* tau, const double* state,
                        double* F_out, double* T_out) {
  body b[L_MAX];
  v3 F[L_MAX], T[L_MAX];
  const lane_ctx c = make_ctx(s, ctx_row);
  load_bodies(s, state, b);
  joint_wrenches(s, &c, tau, b, F, T, NULL, NULL);
  for (int i = 0; i < s->n_links; ++i) {
    F_out[3 * i] = F[i].x; F_out[3 * i + 1] = F[i].y; F_out[3 * i + 2] = F[i].z;
    T_out[3 * i] = T[i].x; T_out[3 * i + 1] = T[i].y; T_out[3 * i + 2] = T[i].z;
  }
}

void obx_engine_reset(const carl_brax_sys_t* s, const oracle_cfg_t* cfg, const double* ctx_table, int n_feat,
                      const uint8_t* mask, double* state, int32_t* elapsed, int32_t* ctx_idx, uint32_t* episode,
                      int32_t* n_calls, double* ep_return, float* obs, double* goal_pos, double* first_state) {
  const int S = 13 * s->n_links;
  for (int i = 0; i < cfg->n_lanes; ++i) {
    if (mask && !mask[i]) continue;
    const uint64_t g = (uint64_t)(cfg->lane_offset + i);
    ctx_idx[i] = select_ctx(cfg, ctx_idx[i], g, episode[i]);
    n_calls[i] += 1;
    body b[L_MAX];
    const lane_ctx c = make_ctx(s, ctx_table + (size_t)ctx_idx[i] * n_feat);
    reset_lane(s, &c, cfg->seed, g, episode[i], b);
    store_bodies(s, b, state + (size_t)i * S);
    if (first_state) memcpy(first_state + (size_t)i * S, state + (size_t)i * S, sizeof(double) * S);
    episode[i] += 1;
    elapsed[i] = 0;
    ep_return[i] = 0.0;
    if (goal_pos) { goal_pos[2 * i] = 0.0; goal_pos[2 * i + 1] = 0.0; } /* wrapper reset: position = (0, 0) */
    observe(s, &c, b, NULL, obs + (size_t)i * s->obs_dim); /* reset obs: qfrc_actuator of a zero action */
  }
}

/* one env step of every lane: n_frames substeps + reward/done/obs + EpisodeWrapper
 * truncation + (cfg->autoreset) in-step reset */
void obx_engine_step(const carl_brax_sys_t* s, const oracle_cfg_t* cfg, const double* ctx_table, int n_feat,
                     const float* action, double* state, int32_t* elapsed, int32_t* ctx_idx, uint32_t* episode,
                     int32_t* n_calls, double* ep_return, float* obs, float* reward, uint8_t* terminated,
                     uint8_t* truncated, float* final_obs, float* last_return, int32_t* last_length,
                     int32_t* episodes_done, double* goal_pos, uint8_t* success, const double* first_state,
                     uint32_t* branch_sig /* [n_lanes][2] or NULL: see substep() */) {
  const int S = 13 * s->n_links, D = s->obs_dim;
  for (int i = 0; i < cfg->n_lanes; ++i) {
    const uint64_t g = (uint64_t)(cfg->lane_offset + i);
    const lane_ctx c = make_ctx(s, ctx_table + (size_t)ctx_idx[i] * n_feat);
    body b[L_MAX];
    load_bodies(s, state + (size_t)i * S, b);
    const float* a = action + (size_t)i * s->n_act;
    double tau[CARL_BRAX_MAX_DOF];
    for (int d = 0; d < s->n_dof; ++d) tau[d] = 0.0;
    double ctrl = 0.0;
    for (int k = 0; k < s->n_act; ++k) {
      double u = a[k];
      ctrl += u * u;
      if (u < s->act_lo[k]) u = s->act_lo[k];
      if (u > s->act_hi[k]) u = s->act_hi[k];
      tau[s->act_dof[k]] += s->act_gear[k] * u;
    }
    const v3 c0 = qrot(b[0].r, f3(s->com[0]));
    double M;
    double x0 = s->reward_on_com ? system_com(s, &c, b, &M).x : b[0].p.x - c0.x;
    uint32_t hc[L_MAX], hl[L_MAX];
    for (int k = 0; k < s->n_links; ++k) hc[k] = hl[k] = 0;
    for (int k = 0; k < s->n_frames; ++k) substep(s, &c, tau, b, hc, hl);
    if (branch_sig) { /* the per-link hashes combined in link order */
      uint32_t a = 0, l = 0;
      for (int k = 0; k < s->n_links; ++k) { a = a * 1000003u + hc[k]; l = l * 1000003u + hl[k]; }
      branch_sig[2 * i] = a;
      branch_sig[2 * i + 1] = l;
    }
    const v3 c1 = qrot(b[0].r, f3(s->com[0]));
    const double x1 = s->reward_on_com ? system_com(s, &c, b, &M).x : b[0].p.x - c1.x;
    const double z1 = b[0].p.z - c1.z;
    const double dt_env = (double)s->dt * s->n_frames;
    int healthy = (z1 >= s->healthy_z_lo) && (z1 <= s->healthy_z_hi);
    if (s->healthy_q_index >= 0) { /* hopper / walker2d torso pitch, inverted-pendulum pole angle */
      double q[CARL_BRAX_MAX_Q], qd[CARL_BRAX_MAX_DOF];
      inverse_kinematics(s, b, q, qd);
      const float qa = (float)q[s->healthy_q_index]; /* the kernel checks the float32 observation entry */
      healthy = healthy && (qa >= s->healthy_q_lo) && (qa <= s->healthy_q_hi);
    }
    double r = s->forward_reward_weight * (s->reward_height ? z1 : (x1 - x0)) / dt_env +
               (s->terminate_when_unhealthy ? s->healthy_reward : s->healthy_reward * healthy) -
               s->ctrl_cost_weight * ctrl;
    int term = s->terminate_when_unhealthy ? !healthy : 0;
    if (s->push_link > 0) { /* brax.envs.pusher: -|object - goal| - w_ctrl |a|^2 - w_near |object - end effector| */
      const v3 op = b[s->push_link].p;
      const v3 d1 = vsub(op, b[s->tip_link].p), d2 = vsub(op, V(c.goal[0], c.goal[1], c.goal[2]));
      r = -sqrt(vdot(d2, d2)) - s->ctrl_cost_weight * ctrl - s->push_near_weight * sqrt(vdot(d1, d1));
      term = 0;
    } else if (s->target_link > 0) { /* brax.envs.reacher: -|tip - goal| - |a|^2 */
      const int tl = s->tip_link, gl = s->target_link;
      const v3 tip = vadd(vsub(b[tl].p, qrot(b[tl].r, f3(s->com[tl]))), qrot(b[tl].r, f3(s->tip_offset)));
      const v3 d = vsub(tip, vsub(b[gl].p, qrot(b[gl].r, f3(s->com[gl]))));
      r = -sqrt(vdot(d, d)) - s->ctrl_cost_weight * ctrl;
      term = 0;
    } else if (s->tip_link > 0) { /* brax.envs.inverted_double_pendulum: alive bonus - distance - velocity penalties */
      const int tl = s->tip_link;
      const v3 o = vsub(b[tl].p, qrot(b[tl].r, f3(s->com[tl])));
      const v3 tip = vadd(o, qrot(b[tl].r, f3(s->tip_offset)));
      double q[CARL_BRAX_MAX_Q], qd[CARL_BRAX_MAX_DOF];
      inverse_kinematics(s, b, q, qd);
      const double dist = s->tip_x_weight * tip.x * tip.x + (tip.z - s->tip_height) * (tip.z - s->tip_height);
      const double v0 = qd[s->tip_vel_dof[0]], v1 = qd[s->tip_vel_dof[1]];
      r = s->healthy_reward - dist - (s->tip_vel_weight[0] * v0 * v0 + s->tip_vel_weight[1] * v1 * v1);
      term = tip.z <= s->tip_min_height;
    }
    double r_out = r;
    elapsed[i] += 1;
    const int trunc = elapsed[i] >= cfg->max_steps;
    observe(s, &c, b, tau, obs + (size_t)i * D);
    if (s->goal_mode && goal_pos) { /* the goal wrapper REPLACES the reward and may terminate */
      int ok = 0;
      r_out = goal_epilogue(s, ctx_table + (size_t)ctx_idx[i] * n_feat, obs + (size_t)i * D, goal_pos + 2 * i, &ok);
      if (ok) term = 1;
      if (success) success[i] = (uint8_t)ok;
    }
    ep_return[i] += (double)(float)r_out;
    reward[i] = (float)r_out;
    terminated[i] = (uint8_t)term;
    truncated[i] = (uint8_t)trunc;
    if (term || trunc) {
      if (last_return) last_return[i] = (float)ep_return[i];
      if (last_length) last_length[i] = elapsed[i];
      if (episodes_done) episodes_done[i] += 1;
      if (cfg->autoreset == 2 && first_state) {
        /* brax.envs.wrappers.training.AutoResetWrapper (reached through carl/envs/brax/wrappers.py:54-78,
         * 121-145): the state of the last explicit reset, same context, nothing drawn */
        if (final_obs) memcpy(final_obs + (size_t)i * D, obs + (size_t)i * D, sizeof(float) * D);
        load_bodies(s, first_state + (size_t)i * S, b);
        elapsed[i] = 0;
        ep_return[i] = 0.0;
        if (goal_pos) { goal_pos[2 * i] = 0.0; goal_pos[2 * i + 1] = 0.0; }
        observe(s, &c, b, NULL, obs + (size_t)i * D);
      } else if (cfg->autoreset) {
        if (final_obs) memcpy(final_obs + (size_t)i * D, obs + (size_t)i * D, sizeof(float) * D);
        ctx_idx[i] = select_ctx(cfg, ctx_idx[i], g, episode[i]);
        n_calls[i] += 1;
        const lane_ctx c2 = make_ctx(s, ctx_table + (size_t)ctx_idx[i] * n_feat);
        reset_lane(s, &c2, cfg->seed, g, episode[i], b);
        episode[i] += 1;
        elapsed[i] = 0;
        ep_return[i] = 0.0;
        if (goal_pos) { goal_pos[2 * i] = 0.0; goal_pos[2 * i + 1] = 0.0; }
        observe(s, &c2, b, NULL, obs + (size_t)i * D);
      }
    }
    store_bodies(s, b, state + (size_t)i * S);
  }
}
