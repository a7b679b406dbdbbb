// brax_kernels.cuh -- Brax "spring" locomotion step on MI355X (Ant / Halfcheetah / Humanoid
// model tables; see include/carl_amd.h carl_brax_sys_t).
//
// Replaces, for N lanes at once: brax.envs.<env>.step/reset -> n_frames x
// brax.spring.pipeline.step as reached from carl/envs/brax/carl_brax_env.py:163-190 and
// carl/envs/brax/wrappers.py:54-78 [brax 0.12.1 is not in the reference tree: the
// specification implemented here is written out in oracle/brax_spring.c's header and
// DESIGN.md; PARITY UNPINNED].
//
// Mapping: one lane = one env.  A lane's whole maximal-coordinate state (13 floats per link)
// plus the per-link force/torque accumulators live in LDS for the duration of the launch
// (row-major [row][64 lanes]: a lane's column is bank-conflict free), so the n_frames
// substeps -- and, in the fused rollout, all T env steps -- never touch HBM for state.
// Link loops index LDS dynamically, which keeps the code compact and the VGPR count low
// (no 117-register unrolled state).  Actions and observations are lane-major records
// (A / O floats per lane): they are staged through LDS so that HBM sees contiguous
// 64-lane x A (or O) blocks instead of 64 strided dwords.  The model table is copied to LDS
// once per workgroup.  No MFMA: with spring_inertia_scale = 1 the world inverse inertia
// R diag(1/I) R^T is evaluated as rotate . scale . rotate^-1 on three floats per lane; packing
// 3x3 blocks of different lanes into MFMA tiles would cost more shuffles than it saves.
#pragma once

#include "carl_device.cuh"
#include "fast_math.cuh"

namespace carl {
namespace brax {

constexpr int kLanes = 64;  // lanes per workgroup = one wavefront
constexpr float kPiF = 3.14159265358979323846f;

struct v3 {
  float x, y, z;
};
struct qt {
  float w, x, y, z;
};
__device__ __forceinline__ v3 V(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ qt qmul(qt a, qt b) {
  return qt{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ qt qconj(qt a) { return qt{a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ qt qnormalize(qt a) {
  const float inv = rsqrtf(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z);
  return qt{a.w * inv, a.x * inv, a.y * inv, a.z * inv};
}
__device__ __forceinline__ v3 qrot(qt q, v3 v) {
  const v3 u = V(q.x, q.y, q.z);
  const v3 t = cross(u, v) * 2.0f;
  return v + t * q.w + cross(u, t);
}
__device__ __forceinline__ qt qaxis(int k, float angle) {
  float s, c;
  sincos_fast(0.5f * angle, s, c);
  return qt{c, k == 0 ? s : 0.0f, k == 1 ? s : 0.0f, k == 2 ? s : 0.0f};
}
__device__ __forceinline__ v3 f3(const float* p) { return V(p[0], p[1], p[2]); }
__device__ __forceinline__ qt f4(const float* p) { return qt{p[0], p[1], p[2], p[3]}; }

struct Body {
  v3 p;
  qt r;
  v3 v, w;
};

// per-lane context: carl_brax_env.py:255-292 in its intended form
struct LaneCtx {
  float gravity_z, friction, elasticity, ang_damping, stiffness_scale;
};

// LDS layout of a workgroup (floats, each row = kLanes consecutive floats)
struct Layout {
  int state;  // 13 * L rows
  int force;  // 6 * L rows  (F, T; reused for contact dv, dw)
  int count;  // L rows       (active contacts per link)
  int mass;   // L rows       (effective mass per link, context-scaled)
  int tau;    // n_dof rows
  int io;     // max(n_act, obs_dim) rows of staging for lane-major records
  int total;
  __host__ __device__ static Layout make(int L, int n_dof, int n_act, int obs_dim) {
    Layout l;
    l.state = 0;
    l.force = l.state + 13 * L;
    l.count = l.force + 6 * L;
    l.mass = l.count + L;
    l.tau = l.mass + L;
    l.io = l.tau + n_dof;
    l.total = l.io + (n_act > obs_dim ? n_act : obs_dim);
    return l;
  }
};

struct Lds {
  float* base;
  Layout lay;
  int tid;
  __device__ __forceinline__ float& at(int row) const { return base[row * kLanes + tid]; }
  __device__ __forceinline__ Body body(int i) const {
    const int r0 = lay.state + 13 * i;
    Body b;
    b.p = V(at(r0), at(r0 + 1), at(r0 + 2));
    b.r = qt{at(r0 + 3), at(r0 + 4), at(r0 + 5), at(r0 + 6)};
    b.v = V(at(r0 + 7), at(r0 + 8), at(r0 + 9));
    b.w = V(at(r0 + 10), at(r0 + 11), at(r0 + 12));
    return b;
  }
  __device__ __forceinline__ void put(int i, const Body& b) const {
    const int r0 = lay.state + 13 * i;
    at(r0) = b.p.x; at(r0 + 1) = b.p.y; at(r0 + 2) = b.p.z;
    at(r0 + 3) = b.r.w; at(r0 + 4) = b.r.x; at(r0 + 5) = b.r.y; at(r0 + 6) = b.r.z;
    at(r0 + 7) = b.v.x; at(r0 + 8) = b.v.y; at(r0 + 9) = b.v.z;
    at(r0 + 10) = b.w.x; at(r0 + 11) = b.w.y; at(r0 + 12) = b.w.z;
  }
  __device__ __forceinline__ void add3(int row, v3 a) const {
    at(row) += a.x; at(row + 1) += a.y; at(row + 2) += a.z;
  }
  __device__ __forceinline__ v3 get3(int row) const { return V(at(row), at(row + 1), at(row + 2)); }
};

__device__ __forceinline__ v3 apply_inv_inertia(const carl_brax_sys_t& s, int i, qt r, v3 t) {
  const v3 l = qrot(qconj(r), t);
  return qrot(r, V(l.x * s.inv_inertia[i][0], l.y * s.inv_inertia[i][1], l.z * s.inv_inertia[i][2]));
}

// the static world as a parent body (planar roots are jointed to it)
__device__ __forceinline__ Body world_body() {
  return Body{V(0, 0, 0), qt{1, 0, 0, 0}, V(0, 0, 0), V(0, 0, 0)};
}

// joint geometry shared by joints.resolve and inverse kinematics
struct JointGeom {
  v3 A_c, A_p, vA_c, vA_p, x_c, x_p, wrel;
  float theta, thetadot;  // single hinge
  v3 axis[3];             // 2-3 stacked hinges: current world axes ...
  float ang[3], rate[3];  // ... Euler x-y-z angles (third signed by dof_sign3) and their rates
};

__device__ __forceinline__ JointGeom joint_geometry(const carl_brax_sys_t& s, int i, const Body& bc, const Body& bp) {
  JointGeom g;
  const int P = s.parent[i];
  const v3 a = f3(s.joint_pos[i]);
  const qt lrot = f4(s.link_rot[i]), jr = f4(s.joint_rot[i]);
  const v3 com_p = (P < 0) ? V(0, 0, 0) : f3(s.com[P]);
  const v3 o_c = bc.p - qrot(bc.r, f3(s.com[i]));
  const v3 o_p = bp.p - qrot(bp.r, com_p);
  g.A_c = o_c + qrot(bc.r, a);
  g.A_p = o_p + qrot(bp.r, f3(s.link_pos[i]) + qrot(lrot, a));  // at zero slide
  g.vA_c = bc.v + cross(bc.w, g.A_c - bc.p);
  g.vA_p = bp.v + cross(bp.w, g.A_p - bp.p);
  const qt rc = qmul(bc.r, jr);
  const qt rp = qmul(qmul(bp.r, lrot), jr);
  g.x_c = qrot(rc, V(1, 0, 0));
  g.x_p = qrot(rp, V(1, 0, 0));
  qt rel = qmul(qconj(rp), rc);
  if (rel.w < 0.0f) { rel.w = -rel.w; rel.x = -rel.x; }
  g.theta = 2.0f * atan2f(rel.x, rel.w);  // twist about the hinge (joint frame x)
  g.wrel = bc.w - bp.w;
  g.thetadot = dot(g.x_c, g.wrel);
  const int nr = s.n_link_dof[i] - s.n_slide[i];
  if (nr >= 2) {  // rel = Rx(al) Ry(be) Rz(ga): decompose, ga = sign * theta_3 (wave-uniform branch)
    const float R00 = 1.0f - 2.0f * (rel.y * rel.y + rel.z * rel.z), R01 = 2.0f * (rel.x * rel.y - rel.w * rel.z);
    const float R02 = fminf(fmaxf(2.0f * (rel.x * rel.z + rel.w * rel.y), -1.0f), 1.0f);
    const float R12 = 2.0f * (rel.y * rel.z - rel.w * rel.x), R22 = 1.0f - 2.0f * (rel.x * rel.x + rel.y * rel.y);
    const float al = atan2f(-R12, R22), be = asinf(R02), ga = atan2f(-R01, R00);
    const float sg = s.dof_sign3[i];
    g.ang[0] = al; g.ang[1] = be; g.ang[2] = sg * ga;
    g.axis[0] = g.x_p;
    const qt rpx = qmul(rp, qaxis(0, al));
    g.axis[1] = qrot(rpx, V(0, 1, 0));
    g.axis[2] = qrot(qmul(rpx, qaxis(1, be)), V(0, 0, 1)) * sg;
    const float w0 = dot(g.wrel, g.axis[0]), w1 = dot(g.wrel, g.axis[1]), w2 = dot(g.wrel, g.axis[2]);
    g.rate[1] = w1;
    if (nr == 3) {  // axis0 and axis2 are not orthogonal: axis0 . axis2 = sign * sin(be)
      const float cc = dot(g.axis[0], g.axis[2]), den = 1.0f - cc * cc;
      g.rate[0] = (w0 - cc * w2) / den;
      g.rate[2] = (w2 - cc * w0) / den;
    } else {
      g.rate[0] = w0;
      g.rate[2] = w2;  // locked direction: damped by k_ang_damp
    }
  }
  return g;
}

// ---- one brax.spring.pipeline.step ---------------------------------------------------------
__device__ __forceinline__ void substep(const carl_brax_sys_t& s, const LaneCtx& c, const Lds& m) {
  const int L = s.n_links;
  for (int k = 0; k < 6 * L; ++k) m.at(m.lay.force + k) = 0.0f;
  // spring.joints.resolve
  for (int i = 0; i < L; ++i) {
    const int P = s.parent[i];
    if (P < 0 && s.n_link_dof[i] == 6) continue;
    const Body bc = m.body(i);
    const Body bp = (P < 0) ? world_body() : m.body(P);
    const JointGeom g = joint_geometry(s, i, bc, bp);
    const float kp = s.k_pos[i] * c.stiffness_scale;
    v3 e = g.A_p - g.A_c, ev = g.vA_p - g.vA_c;
    v3 f = V(0, 0, 0);
    const int ns = s.n_slide[i], d0 = s.dof_start[i];
    for (int k = 0; k < ns; ++k) {  // prismatic dofs: free along the axis, own spring/damper/force
      const v3 ax = qrot(bp.r, f3(s.slide_axis[i][k]));
      const float qk = -dot(e, ax), qdk = -dot(ev, ax);
      e = e + ax * qk;
      ev = ev + ax * qdk;
      f = f + ax * (m.at(m.lay.tau + d0 + k) - s.dof_damping[d0 + k] * qdk - s.dof_stiffness[d0 + k] * qk);
    }
    f = f + e * kp + ev * s.k_vel[i];
    v3 t;
    const int d = d0 + ns, nr = s.n_link_dof[i] - ns;
    if (nr == 1) {
      t = cross(g.x_c, g.x_p) * kp;  // keep the hinge axes aligned
      float ta = m.at(m.lay.tau + d) - s.dof_damping[d] * g.thetadot - s.dof_stiffness[d] * g.theta;
      if (g.theta < s.dof_lo[d]) ta += s.k_limit[i] * (s.dof_lo[d] - g.theta);
      if (g.theta > s.dof_hi[d]) ta -= s.k_limit[i] * (g.theta - s.dof_hi[d]);
      t = t + g.x_c * ta;
    } else {  // 2 or 3 stacked hinges: per-dof torques about the current axes; a missing third
              // dof is locked by the constraint spring on its Euler angle
      t = V(0, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float ta;
        if (k < nr) {
          const int dk = d + k;
          ta = m.at(m.lay.tau + dk) - s.dof_damping[dk] * g.rate[k] - s.dof_stiffness[dk] * g.ang[k];
          if (g.ang[k] < s.dof_lo[dk]) ta += s.k_limit[i] * (s.dof_lo[dk] - g.ang[k]);
          if (g.ang[k] > s.dof_hi[dk]) ta -= s.k_limit[i] * (g.ang[k] - s.dof_hi[dk]);
        } else {
          ta = -kp * g.ang[k];
        }
        t = t + g.axis[k] * ta;
      }
    }
    t = t - g.wrel * s.k_ang_damp[i];
    const int fc = m.lay.force + 6 * i;
    m.add3(fc, f);
    m.add3(fc + 3, cross(g.A_c - bc.p, f) + t);
    if (P >= 0) {
      const int fp = m.lay.force + 6 * P;
      m.add3(fp, f * -1.0f);
      m.add3(fp + 3, (cross(g.A_p - bp.p, f) + t) * -1.0f);
    }
  }
  // semi-implicit Euler: velocities first
  for (int i = 0; i < L; ++i) {
    Body b = m.body(i);
    const int fr = m.lay.force + 6 * i;
    const float inv_m = 1.0f / m.at(m.lay.mass + i);
    b.v = b.v + (m.get3(fr) * inv_m + V(0, 0, c.gravity_z)) * s.dt;
    b.w = b.w + apply_inv_inertia(s, i, b.r, m.get3(fr + 3)) * s.dt;
    m.put(i, b);
    for (int k = 0; k < 6; ++k) m.at(fr + k) = 0.0f;  // rows reused for contact deltas
    m.at(m.lay.count + i) = 0.0f;
  }
  // spring.collisions.resolve: spheres vs the plane z = 0
  const v3 n = V(0, 0, 1);
  for (int k = 0; k < s.n_coll; ++k) {
    const int i = s.coll_link[k];
    const Body b = m.body(i);
    const v3 ctr = b.p - qrot(b.r, f3(s.com[i])) + qrot(b.r, f3(s.coll_pos[k]));
    const float depth = s.coll_radius[k] - ctr.z;
    if (!(depth > 0.0f)) continue;
    const v3 r = V(ctr.x, ctr.y, ctr.z - s.coll_radius[k]) - b.p;
    const v3 rel = b.v + cross(b.w, r);
    const float vn = dot(n, rel);
    const float inv_m = 1.0f / m.at(m.lay.mass + i);
    const float ang = dot(n, cross(apply_inv_inertia(s, i, b.r, cross(r, n)), r));
    const float imp = (-(1.0f + c.elasticity) * vn + s.baumgarte_erp * depth / s.dt) / (inv_m + ang);
    if (!(imp > 0.0f) || !(vn < 0.0f)) continue;
    v3 J = n * imp;
    const v3 vt = rel - n * vn;
    const float vt_len = sqrtf(dot(vt, vt));
    if (vt_len > 1e-9f) {
      const v3 dir = vt * (1.0f / vt_len);
      const float ang_d = dot(dir, cross(apply_inv_inertia(s, i, b.r, cross(r, dir)), r));
      const float imp_d = fminf(vt_len / (inv_m + ang_d), c.friction * imp);
      J = J - dir * imp_d;
    }
    const int fr = m.lay.force + 6 * i;
    m.add3(fr, J * inv_m);
    m.add3(fr + 3, apply_inv_inertia(s, i, b.r, cross(r, J)));
    m.at(m.lay.count + i) += 1.0f;
  }
  // spring.integrator.integrate
  const float dl = __expf(s.vel_damping * s.dt), da = __expf(c.ang_damping * s.dt);
  for (int i = 0; i < L; ++i) {
    Body b = m.body(i);
    b.v = b.v * dl;
    b.w = b.w * da;
    const float cnt = m.at(m.lay.count + i);
    if (cnt > 0.0f) {
      const int fr = m.lay.force + 6 * i;
      const float ic = 1.0f / cnt;
      b.v = b.v + m.get3(fr) * ic;
      b.w = b.w + m.get3(fr + 3) * ic;
    }
    b.p = b.p + b.v * s.dt;
    const qt dq = qmul(qt{0.0f, b.w.x, b.w.y, b.w.z}, b.r);
    const float h = 0.5f * s.dt;
    b.r = qnormalize(qt{b.r.w + h * dq.w, b.r.x + h * dq.x, b.r.y + h * dq.y, b.r.z + h * dq.z});
    m.put(i, b);
  }
}

// whole-body centre of mass (brax.envs.humanoid.Humanoid._com); *mass_sum = total mass
__device__ __forceinline__ v3 system_com(const carl_brax_sys_t& s, const Lds& m, float* mass_sum) {
  v3 com = V(0, 0, 0);
  float M = 0.0f;
  for (int i = 0; i < s.n_links; ++i) {
    const float mi = m.at(m.lay.mass + i);
    const int r0 = m.lay.state + 13 * i;
    com = com + m.get3(r0) * mi;
    M += mi;
  }
  *mass_sum = M;
  return com * (1.0f / M);
}

// kinematics.world_to_joint + inverse -> observation rows (q[skip:] ++ qd) in the io staging;
// obs_extended (humanoid) appends com inertia (L x 10), com velocity (L x 6) and qfrc_actuator
// (the tau rows; `zero_frc`: reset observations see a zero action)
__device__ __forceinline__ void observe(const carl_brax_sys_t& s, const Lds& m, bool zero_frc = false) {
  const int skip = s.exclude_current_positions;
  const int qd0 = s.n_q - skip;  // first qd row in the observation
  for (int i = 0; i < s.n_links; ++i) {
    const int P = s.parent[i];
    const Body b = m.body(i);
    if (P < 0 && s.n_link_dof[i] == 6) {
      const v3 c = qrot(b.r, f3(s.com[i]));
      const v3 o = b.p - c;
      const v3 vel = b.v - cross(b.w, c);
      const float qv[7] = {o.x, o.y, o.z, b.r.w, b.r.x, b.r.y, b.r.z};
      for (int k = 0; k < 7; ++k)
        if (s.q_start[i] + k >= skip) m.at(m.lay.io + s.q_start[i] + k - skip) = qv[k];
      const float dv[6] = {vel.x, vel.y, vel.z, b.w.x, b.w.y, b.w.z};
      for (int k = 0; k < 6; ++k) m.at(m.lay.io + qd0 + s.dof_start[i] + k) = dv[k];
    } else {
      const Body bp = (P < 0) ? world_body() : m.body(P);
      const JointGeom g = joint_geometry(s, i, b, bp);
      const int ns = s.n_slide[i];
      for (int k = 0; k < ns; ++k) {
        const v3 ax = qrot(bp.r, f3(s.slide_axis[i][k]));
        if (s.q_start[i] + k >= skip) m.at(m.lay.io + s.q_start[i] + k - skip) = dot(g.A_c - g.A_p, ax);
        m.at(m.lay.io + qd0 + s.dof_start[i] + k) = dot(g.vA_c - g.vA_p, ax);
      }
      const int nr = s.n_link_dof[i] - ns;
      if (nr == 1) {
        if (s.q_start[i] + ns >= skip) m.at(m.lay.io + s.q_start[i] + ns - skip) = g.theta;
        m.at(m.lay.io + qd0 + s.dof_start[i] + ns) = g.thetadot;
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < nr) {
            if (s.q_start[i] + ns + k >= skip) m.at(m.lay.io + s.q_start[i] + ns + k - skip) = g.ang[k];
            m.at(m.lay.io + qd0 + s.dof_start[i] + ns + k) = g.rate[k];
          }
      }
    }
  }
  if (!s.obs_extended) return;
  int k = m.lay.io + qd0 + s.n_dof;
  float M;
  const v3 com = system_com(s, m, &M);
  for (int i = 0; i < s.n_links; ++i) {  // inertia about the system com, world axes, row-major, then mass
    const Body b = m.body(i);
    const v3 d = b.p - com;
    const float mi = m.at(m.lay.mass + i), dd = dot(d, d);
    const float I0 = 1.0f / s.inv_inertia[i][0], I1 = 1.0f / s.inv_inertia[i][1], I2 = 1.0f / s.inv_inertia[i][2];
    const v3 ex = qrot(b.r, V(1, 0, 0)), ey = qrot(b.r, V(0, 1, 0)), ez = qrot(b.r, V(0, 0, 1));
    const float e[3][3] = {{ex.x, ey.x, ez.x}, {ex.y, ey.y, ez.y}, {ex.z, ey.z, ez.z}};
    const float dv[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        float v = e[r][0] * I0 * e[cc][0];
        v += e[r][1] * I1 * e[cc][1];
        v += e[r][2] * I2 * e[cc][2];
        v += mi * ((r == cc ? dd : 0.0f) - dv[r] * dv[cc]);
        m.at(k++) = v;
      }
    m.at(k++) = mi;
  }
  for (int i = 0; i < s.n_links; ++i) {
    const Body b = m.body(i);
    const float f = m.at(m.lay.mass + i) / M;
    m.at(k++) = f * b.v.x; m.at(k++) = f * b.v.y; m.at(k++) = f * b.v.z;
    m.at(k++) = b.w.x; m.at(k++) = b.w.y; m.at(k++) = b.w.z;
  }
  for (int i = 0; i < s.n_dof; ++i) m.at(k++) = zero_frc ? 0.0f : m.at(m.lay.tau + i);
}

// kinematics.forward + com.from_world from (q, qd) held in the io staging rows
// (q at rows [0, n_q), qd at rows [n_q, n_q + n_dof)); writes the state rows.
__device__ __noinline__ void forward_kinematics(const carl_brax_sys_t& s, const Lds& m) {
  // link-frame origins and their velocities are kept in the force rows (free at this point)
  for (int i = 0; i < s.n_links; ++i) {
    const int P = s.parent[i];
    qt rot;
    v3 o, vel, ang;
    const int q0 = m.lay.io + s.q_start[i], d0 = m.lay.io + s.n_q + s.dof_start[i];
    if (P < 0 && s.n_link_dof[i] == 6) {
      rot = qnormalize(qt{m.at(q0 + 3), m.at(q0 + 4), m.at(q0 + 5), m.at(q0 + 6)});
      o = V(m.at(q0), m.at(q0 + 1), m.at(q0 + 2));
      vel = V(m.at(d0), m.at(d0 + 1), m.at(d0 + 2));
      ang = V(m.at(d0 + 3), m.at(d0 + 4), m.at(d0 + 5));
    } else {
      const Body bp = (P < 0) ? world_body() : m.body(P);
      const v3 o_p = (P < 0) ? V(0, 0, 0) : m.get3(m.lay.force + 6 * P);
      const v3 ov_p = (P < 0) ? V(0, 0, 0) : m.get3(m.lay.force + 6 * P + 3);
      const int ns = s.n_slide[i], nr = s.n_link_dof[i] - ns;
      const qt jr = f4(s.joint_rot[i]), lrot = f4(s.link_rot[i]);
      const qt rpj = qmul(qmul(bp.r, lrot), jr);  // parent-side joint frame in the world
      // hinges stack intrinsically about the joint frame's x, y, +-z
      qt rj{1.0f, 0.0f, 0.0f, 0.0f};
      v3 wj = V(0, 0, 0);
      for (int k = 0; k < nr; ++k) {
        const float sg = (k == 2) ? s.dof_sign3[i] : 1.0f;
        const v3 axis = qrot(qmul(rpj, rj), V(k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f)) * sg;
        wj = wj + axis * m.at(d0 + ns + k);
        rj = qmul(rj, qaxis(k, sg * m.at(q0 + ns + k)));
      }
      const qt rl = qmul(qmul(jr, rj), qconj(jr));  // joint rotation in child coordinates
      const v3 a = f3(s.joint_pos[i]);
      v3 lpos = f3(s.link_pos[i]) + qrot(lrot, a - qrot(rl, a));
      v3 slide_vel = V(0, 0, 0);
      for (int k = 0; k < ns; ++k) {
        const v3 ax = f3(s.slide_axis[i][k]);
        lpos = lpos + ax * m.at(q0 + k);
        slide_vel = slide_vel + qrot(bp.r, ax) * m.at(d0 + k);
      }
      rot = qmul(bp.r, qmul(lrot, rl));
      o = o_p + qrot(bp.r, lpos);
      const v3 anchor_w = o + qrot(rot, a);
      ang = bp.w + wj;
      vel = ov_p + cross(bp.w, o - o_p) + slide_vel + cross(wj, o - anchor_w);
    }
    const int fr = m.lay.force + 6 * i;
    m.at(fr) = o.x; m.at(fr + 1) = o.y; m.at(fr + 2) = o.z;
    m.at(fr + 3) = vel.x; m.at(fr + 4) = vel.y; m.at(fr + 5) = vel.z;
    const v3 c = qrot(rot, f3(s.com[i]));
    Body b;
    b.r = rot;
    b.w = ang;
    b.p = o + c;
    b.v = vel + cross(ang, c);
    m.put(i, b);
  }
}

// brax.envs.<env>.reset: q = init_q + U(-noise, noise), qd = vel_scale * N(0, 1).
// Draw k uses word (k mod 4) of Philox block k / 4 on sub-stream 0x80000000 | block.
__device__ __forceinline__ float draw_u(uint64_t seed, uint64_t g, uint32_t ep, int k) {
  const u32x4 w = lane_words(seed, g, ep, 0x80000000u | (uint32_t)(k >> 2));
  const uint32_t x = (k & 3) == 0 ? w.x : (k & 3) == 1 ? w.y : (k & 3) == 2 ? w.z : w.w;
  return u01(x);
}

__device__ __noinline__ void reset_state(const carl_brax_sys_t& s, const carl_batch_t& b, const Lds& m,
                                         uint64_t glane, uint32_t episode) {
  int k = 0;
  u32x4 w{};
  for (int i = 0; i < s.n_q; ++i, ++k) {
    if ((k & 3) == 0) w = lane_words(b.seed, glane, episode, 0x80000000u | (uint32_t)(k >> 2));
    const uint32_t x = (k & 3) == 0 ? w.x : (k & 3) == 1 ? w.y : (k & 3) == 2 ? w.z : w.w;
    m.at(m.lay.io + i) = s.init_q[i] + s.reset_noise_scale * (2.0f * u01(x) - 1.0f);
  }
  if (s.reset_vel_uniform) {  // brax.envs.humanoid: qvel = U(-scale, scale)
    for (int i = 0; i < s.n_dof; ++i, ++k)
      m.at(m.lay.io + s.n_q + i) = s.reset_vel_scale * (2.0f * draw_u(b.seed, glane, episode, k) - 1.0f);
  } else
  for (int i = 0; i < s.n_dof; i += 2, k += 2) {
    const float u1 = draw_u(b.seed, glane, episode, k), u2 = draw_u(b.seed, glane, episode, k + 1);
    const float rad = sqrtf(-2.0f * logf(1.0f - u1));
    float sn, cs;
    sincos_fast(2.0f * kPiF * u2, sn, cs);
    m.at(m.lay.io + s.n_q + i) = s.reset_vel_scale * rad * cs;
    if (i + 1 < s.n_dof) m.at(m.lay.io + s.n_q + i + 1) = s.reset_vel_scale * rad * sn;
  }
  forward_kinematics(s, m);
}

__device__ __forceinline__ LaneCtx load_ctx(const carl_brax_sys_t& s, const carl_batch_t& b, const Lds& m, int c) {
  const carl_brax_ctx_map_t& cm = s.ctx;
  auto get = [&](int row, float dflt) { return row >= 0 ? b.ctx_table[(size_t)row * b.ctx_stride + c] : dflt; };
  LaneCtx lc;
  lc.gravity_z = get(cm.gravity, s.gravity_z);
  lc.friction = get(cm.friction, s.friction);
  lc.elasticity = get(cm.elasticity, s.elasticity);
  lc.ang_damping = get(cm.ang_damping, s.ang_damping);
  lc.stiffness_scale = get(cm.joint_stiffness_scale, 1.0f);
  for (int i = 0; i < s.n_links; ++i) m.at(m.lay.mass + i) = s.mass[i];
  for (int k = 0; k < cm.n_mass; ++k)
    m.at(m.lay.mass + cm.mass_link[k]) =
        s.mass[cm.mass_link[k]] * (b.ctx_table[(size_t)cm.mass_row[k] * b.ctx_stride + c] / cm.mass_nominal[k]);
  return lc;
}

// lane-major records <-> LDS staging rows.  HBM side: 64 lanes x W floats contiguous.
__device__ __forceinline__ void stage_in(const float* __restrict__ src, size_t lane_base, int n_lanes, int W,
                                         const Lds& m) {
  // element e of the block (lane = e / W, k = e % W) -> LDS row k, column lane
  const int total = W * kLanes;
  for (int e = m.tid; e < total; e += kLanes) {
    const int lane = e / W, k = e - lane * W;
    if ((int)lane_base + lane < n_lanes) m.base[(m.lay.io + k) * kLanes + lane] = src[lane_base * W + e];
  }
  __syncthreads();
}
// `only_flagged`: copy only lanes whose flag (first `count` row, free outside substep) is set
__device__ __forceinline__ void stage_out(float* __restrict__ dst, size_t lane_base, int n_lanes, int W,
                                          const Lds& m, bool only_flagged = false) {
  __syncthreads();
  const int total = W * kLanes;
  for (int e = m.tid; e < total; e += kLanes) {
    const int lane = e / W, k = e - lane * W;
    if ((int)lane_base + lane < n_lanes && (!only_flagged || m.base[m.lay.count * kLanes + lane] != 0.0f))
      dst[lane_base * W + e] = m.base[(m.lay.io + k) * kLanes + lane];
  }
  __syncthreads();
}

struct LaneState {
  float ep_return;
  int elapsed, cidx, n_new_calls, n_new_episodes;
  uint32_t episode;
  LaneCtx ctx;
  float goal_x, goal_y, goal_radius, pos_x, pos_y;  // goal mode only
};

// BraxWalkerGoalWrapper (carl/envs/brax/brax_walker_goal_wrapper.py:69-121): compass code ->
// goal position = direction * target_distance; radius
__device__ __forceinline__ void load_goal(const carl_brax_sys_t& s, const carl_batch_t& b, int c, LaneState& r) {
  const carl_brax_ctx_map_t& cm = s.ctx;
  const int code = __float2int_rn(b.ctx_table[(size_t)cm.target_direction * b.ctx_stride + c]);
  const float dist = b.ctx_table[(size_t)cm.target_distance * b.ctx_stride + c];
  r.goal_radius = b.ctx_table[(size_t)cm.target_radius * b.ctx_stride + c];
  const float cc = 0.92387953251128674f, sn = 0.38268343236508977f, h = 0.70710678118654752f;  // 22.5 deg, sqrt(1/2)
  float dx = 0.0f, dy = 0.0f;
  switch (code) {
    case 3: dy = -1.0f; break;
    case 1: dy = 1.0f; break;
    case 2: dx = 1.0f; break;
    case 4: dx = -1.0f; break;
    case 34: dx = -h; dy = -h; break;
    case 14: dx = -h; dy = h; break;
    case 32: dx = h; dy = -h; break;
    case 12: dx = h; dy = h; break;
    case 334: dx = -cc; dy = -sn; break;
    case 434: dx = -sn; dy = -cc; break;
    case 114: dx = -cc; dy = sn; break;
    case 414: dx = -sn; dy = cc; break;
    case 332: dx = cc; dy = -sn; break;
    case 232: dx = sn; dy = -cc; break;
    case 112: dx = cc; dy = sn; break;
    case 212: dx = sn; dy = cc; break;
    default: break;
  }
  r.goal_x = dx * dist;
  r.goal_y = dy * dist;
}

// mode 0: reset (mask optional), mode 1: n_steps env steps (1 = per call, T = fused rollout)
template <int MODE>
__global__ void __launch_bounds__(kLanes) brax_kernel(const carl_batch_t b, const carl_brax_sys_t* __restrict__ sys_dev,
                                                      const carl_step_io_t io, const uint8_t* __restrict__ mask,
                                                      float* __restrict__ reset_obs, const int n_steps) {
  __shared__ carl_brax_sys_t s;
  extern __shared__ float lds_dyn[];
  {  // model table -> LDS, once per workgroup
    const uint32_t* src = reinterpret_cast<const uint32_t*>(sys_dev);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&s);
    for (int k = threadIdx.x; k < (int)(sizeof(carl_brax_sys_t) / 4); k += kLanes) dst[k] = src[k];
  }
  __syncthreads();
  Lds m{lds_dyn, Layout::make(s.n_links, s.n_dof, s.n_act, s.obs_dim > s.n_q + s.n_dof ? s.obs_dim : s.n_q + s.n_dof),
        (int)threadIdx.x};
  const size_t lane_base = (size_t)blockIdx.x * kLanes;
  const int lane = (int)lane_base + (int)threadIdx.x;
  const bool active = lane < b.n_lanes;
  const uint64_t glane = (uint64_t)(b.lane_offset + lane);
  const size_t n = (size_t)b.n_lanes;
  const int S = CARL_BRAX_LINK_STATE * s.n_links;
  LaneState r{};
  const bool goal = s.goal_mode != 0 && b.goal_pos != nullptr;
  if (active) {
    r.cidx = b.ctx_idx[lane];
    r.episode = b.episode[lane];
    r.elapsed = b.elapsed[lane];
    r.ep_return = b.ep_return[lane];
  }

  if constexpr (MODE == 0) {
    const bool go = active && (mask == nullptr || mask[lane] != 0);
    if (go) {
      r.cidx = select_context(b, r.cidx, glane, r.episode);
      reset_state(s, b, m, glane, r.episode);
      r.episode += 1u;
      for (int k = 0; k < S; ++k) b.state[(size_t)k * n + lane] = m.at(m.lay.state + k);
      b.elapsed[lane] = 0;
      b.ep_return[lane] = 0.0f;
      b.ctx_idx[lane] = r.cidx;
      b.episode[lane] = r.episode;
      b.n_calls[lane] += 1;
      if (goal) {  // wrapper reset: position = (0, 0)
        b.goal_pos[lane] = 0.0f;
        b.goal_pos[n + lane] = 0.0f;
      }
      if (b.ctx_obs != nullptr)
        for (int k = 0; k < b.n_ctx_obs; ++k)
          b.ctx_obs[(size_t)k * n + lane] = b.ctx_table[(size_t)b.ctx_obs_feat[k] * b.ctx_stride + r.cidx];
      if (s.obs_extended) load_ctx(s, b, m, r.cidx);  // com inertia / velocity use the lane's masses
      observe(s, m, true);
      if (reset_obs != nullptr)  // masked resets write only their own rows (no block staging)
        for (int k = 0; k < s.obs_dim; ++k) reset_obs[(size_t)lane * s.obs_dim + k] = m.at(m.lay.io + k);
    }
    return;
  } else {
    if (active) {
      for (int k = 0; k < S; ++k) m.at(m.lay.state + k) = b.state[(size_t)k * n + lane];
      r.ctx = load_ctx(s, b, m, r.cidx);
      if (goal) {
        load_goal(s, b, r.cidx, r);
        r.pos_x = b.goal_pos[lane];
        r.pos_y = b.goal_pos[n + lane];
      }
    }
    const float dt_env = s.dt * (float)s.n_frames;
    for (int t = 0; t < n_steps; ++t) {
      const size_t step_off = (size_t)t * n;
      stage_in(static_cast<const float*>(io.action) + step_off * s.n_act, lane_base, b.n_lanes, s.n_act, m);
      bool done = false, terminated = false, truncated = false;
      float reward = 0.0f;
      if (active) {
        for (int d = 0; d < s.n_dof; ++d) m.at(m.lay.tau + d) = 0.0f;
        float ctrl = 0.0f;
        for (int k = 0; k < s.n_act; ++k) {  // actuator.to_tau
          const float u = m.at(m.lay.io + k);
          ctrl += u * u;
          m.at(m.lay.tau + s.act_dof[k]) += s.act_gear[k] * fminf(fmaxf(u, s.act_lo[k]), s.act_hi[k]);
        }
        const Body b0 = m.body(0);
        float msum;
        const float x0 = s.reward_on_com ? system_com(s, m, &msum).x : b0.p.x - qrot(b0.r, f3(s.com[0])).x;
        for (int f = 0; f < s.n_frames; ++f) substep(s, r.ctx, m);
        const Body b1 = m.body(0);
        const v3 c1 = qrot(b1.r, f3(s.com[0]));
        const float x1 = s.reward_on_com ? system_com(s, m, &msum).x : b1.p.x - c1.x, z1 = b1.p.z - c1.z;
        const bool healthy = (z1 >= s.healthy_z_lo) && (z1 <= s.healthy_z_hi);
        reward = s.forward_reward_weight * (x1 - x0) / dt_env +
                 (s.terminate_when_unhealthy ? s.healthy_reward : (healthy ? s.healthy_reward : 0.0f)) -
                 s.ctrl_cost_weight * ctrl;
        terminated = s.terminate_when_unhealthy && !healthy;
        r.elapsed += 1;
        truncated = (b.max_episode_steps > 0) && (r.elapsed >= b.max_episode_steps);
        observe(s, m);
        if (goal) {  // brax_walker_goal_wrapper.py:124-140: progress reward replaces the env reward
          const float nx = r.pos_x + m.at(m.lay.io + s.goal_obs_idx[0]) * s.goal_dt;
          const float ny = r.pos_y + m.at(m.lay.io + s.goal_obs_idx[1]) * s.goal_dt;
          const float cur = sqrtf((r.goal_x - nx) * (r.goal_x - nx) + (r.goal_y - ny) * (r.goal_y - ny));
          const float prev = sqrtf((r.goal_x - r.pos_x) * (r.goal_x - r.pos_x) + (r.goal_y - r.pos_y) * (r.goal_y - r.pos_y));
          r.pos_x = nx;
          r.pos_y = ny;
          const bool ok = cur <= r.goal_radius;
          terminated = terminated | ok;
          reward = fmaxf(0.0f, prev - cur);
          if (b.success != nullptr) b.success[step_off + lane] = (uint8_t)ok;
        }
        r.ep_return += reward;
        done = terminated | truncated;
        io.reward[step_off + lane] = reward;
        io.terminated[step_off + lane] = (uint8_t)terminated;
        io.truncated[step_off + lane] = (uint8_t)truncated;
      }
      const unsigned long long any_done = __ballot(done);
      if (any_done != 0ull) {
        const float fin_ret = r.ep_return;
        const int fin_len = r.elapsed;
        if (done) {
          if (b.last_return) b.last_return[lane] = fin_ret;
          if (b.last_length) b.last_length[lane] = fin_len;
          r.n_new_episodes += 1;
        }
        log_finished(b, done, glane, fin_ret, fin_len);
        if (b.flags & CARL_FLAG_AUTORESET) {
          if (io.final_obs != nullptr) {  // terminal observation, done lanes only
            m.at(m.lay.count) = done ? 1.0f : 0.0f;
            stage_out(io.final_obs + step_off * s.obs_dim, lane_base, b.n_lanes, s.obs_dim, m, true);
          }
          if (done) {
            r.cidx = select_context(b, r.cidx, glane, r.episode);
            reset_state(s, b, m, glane, r.episode);
            r.episode += 1u;
            r.ctx = load_ctx(s, b, m, r.cidx);
            if (goal) {
              load_goal(s, b, r.cidx, r);
              r.pos_x = r.pos_y = 0.0f;
            }
            r.elapsed = 0;
            r.ep_return = 0.0f;
            r.n_new_calls += 1;
            if (b.ctx_obs != nullptr)
              for (int k = 0; k < b.n_ctx_obs; ++k)
                b.ctx_obs[(size_t)k * n + lane] = b.ctx_table[(size_t)b.ctx_obs_feat[k] * b.ctx_stride + r.cidx];
            observe(s, m, true);
          }
        }
      }
      stage_out(io.obs + step_off * s.obs_dim, lane_base, b.n_lanes, s.obs_dim, m);
    }
    if (active) {
      for (int k = 0; k < S; ++k) b.state[(size_t)k * n + lane] = m.at(m.lay.state + k);
      b.elapsed[lane] = r.elapsed;
      b.ep_return[lane] = r.ep_return;
      if (goal) {
        b.goal_pos[lane] = r.pos_x;
        b.goal_pos[n + lane] = r.pos_y;
      }
      if (r.n_new_episodes != 0 && b.episodes_done != nullptr) b.episodes_done[lane] += r.n_new_episodes;
      if (r.n_new_calls != 0) {
        b.ctx_idx[lane] = r.cidx;
        b.episode[lane] = r.episode;
        b.n_calls[lane] += r.n_new_calls;
      }
    }
  }
}

}  // namespace brax
}  // namespace carl
