// classic_control.hip.h -- per-lane physics of CARL's classic-control families.
//
// Each family is a traits struct the generic engine kernels (engine_kernels.hip.h) are
// instantiated with:
//   S, D, F            state columns, observation length, context-table rows
//   Action             int (Discrete) or float (Box)
//   Params             what the physics reads of a context, gathered (and pre-combined
//                      where the reference's own evaluation order allows) per lane
//   Aux                values derived from the CURRENT state that the next step and the
//                      observation share (e.g. sin/cos of the pendulum angle), kept in
//                      registers across the steps of a fused rollout
//   load(ctx, c, ..)   gather Params of context id c (feature rows in the order of the
//                      reference class's get_context_features())
//   prepare(s, aux)    aux of a freshly loaded / reset state
//   step(...)          one transition: the arithmetic of gymnasium 0.29.1's env.step
//                      [upstream; equations E-CP .. E-MCC of SURVEY.md section 8a];
//                      leaves aux describing the NEW state
//   observe(...)       what env.step / CARL's reset returns as "obs"
//   reset(...)         CARL's init-state distribution from 4 Philox words
//
// The reference computes in float64 on float32-representable state; this engine keeps
// fp32 state in HBM and computes in fp32 (north_star: transitions within 1e-5), except
// Acrobot's RK4, which is fp64 by default (see AcrobotT).
#pragma once

#include <type_traits>

#include "carl_device.hip.h"
#include "fast_math.hip.h"

namespace carl {

constexpr float kPi = 3.14159265358979323846f;

struct NoAux {};

// ================================ CartPole ========================================
// features: carl/envs/gymnasium/classic_control/carl_cartpole.py:15-42
struct CartPole {
  static constexpr int S = 4, D = 4, F = 8;
  using Action = int;
  using Aux = NoAux;
  enum { GRAVITY, MASSCART, MASSPOLE, LENGTH, FORCE_MAG, TAU, INIT_LO, INIT_HI };
  static constexpr bool kNeedsStepNoise = false;
  // the staged rollout's specialisations (engine_kernels.hip.h: init-state words drawn once per chunk, the PLAIN
  // done path, the LDS-resident context table).  Made for CartPole's short episodes; switched on for every
  // family because the smaller done path also frees the step loop's registers: Pendulum 279 -> 266,
  // MountainCar 292 -> 273, MountainCarContinuous 313 -> 276, Acrobot 2330 -> 2205 ns/step (A/B, one box)
  static constexpr bool kPredraw = true;
  static constexpr int kDeepBelowLanes = 65536;  // (engine_kernels.hip.h: deep_below_lanes_of)
  // short episodes under any policy: the done handling of the PLAIN staged rollout is straight-line selects on
  // every step instead of a wave-uniform branch that is taken ~95 % of the time (engine_kernels.hip.h: step_dense)
  static constexpr bool kDenseDone = true;

  struct Params {
    float gravity, masspole, length, force_mag, tau, inv_total_mass, polemass_length;
    float init_lo, init_hi;  // CARL reset distribution (kept with the physics so that a reset
                             // of a lane whose context does not change touches no memory)
  };

  template <class Ctx>
  __device__ static __forceinline__ Params load(const Ctx& ctx, int c, int flags) {
    Params p;
    p.gravity = ctx.get(GRAVITY, c);
    p.masspole = ctx.get(MASSPOLE, c);
    p.length = ctx.get(LENGTH, c);
    p.force_mag = ctx.get(FORCE_MAG, c);
    p.tau = ctx.get(TAU, c);
    float total_mass;
    if (flags & CARL_FLAG_CARTPOLE_RECOMPUTE) {
      total_mass = p.masspole + ctx.get(MASSCART, c);
      p.polemass_length = p.masspole * p.length;
    } else {
      // Quirk C1: gymnasium derives these once in __init__ from ITS
      // defaults (masspole 0.1 + masscart 1.0, masspole 0.1 * length 0.5); CARL's
      // setattr (carl_gymnasium_env.py:75-77) never refreshes them.
      total_mass = 0.1f + 1.0f;
      p.polemass_length = 0.1f * 0.5f;
    }
    p.inv_total_mass = 1.0f / total_mass;
    p.init_lo = ctx.get(INIT_LO, c);
    p.init_hi = ctx.get(INIT_HI, c);
    return p;
  }

  __device__ static __forceinline__ void prepare(const float (&)[S], Aux&) {}

  __device__ static __forceinline__ bool out_of_bounds(float x, float theta) {
    const float x_thr = 2.4f;
    const float th_thr = (float)(12.0 * 2.0 * 3.14159265358979323846 / 360.0);
    return (x < -x_thr) | (x > x_thr) | (theta < -th_thr) | (theta > th_thr);
  }

  // CartPoleEnv.step, kinematics_integrator == "euler"
  __device__ static __forceinline__ bool step(const Params& p, float (&s)[S], Aux&, int action, float /*noise*/,
                                              int elapsed, float& reward) {
    // Every rounding below is spelled out (explicit fma, no implicit contraction): left to -ffp-contract=fast the
    // compiler fuses `a * b - c * d` one way or the other depending on the code AROUND the expression, so the
    // per-call kernel and the unrolled fused rollout -- different surroundings -- came out one ulp apart; the
    // engine's contract is that they agree bit for bit.
#pragma clang fp contract(off)
    const float x = s[0], x_dot = s[1], theta = s[2], theta_dot = s[3];
    // steps_beyond_terminated: a lane stepped again after terminating (only reachable
    // with auto-reset off) earns 0.  Inferred instead of stored: the pre-step state is
    // out of bounds and this is not the first step since reset (a reset may legally
    // start out of bounds when the context widens initial_state_lower/upper; that
    // first terminating step still earns 1.0).
    const bool was_terminated = (elapsed > 0) && out_of_bounds(x, theta);
    const float force = (action == 1) ? p.force_mag : -p.force_mag;
    float sintheta, costheta;
    sincos_fast_smallarg(theta, sintheta, costheta);
    // temp = (force + polemass_length * theta_dot^2 * sin) / total_mass
    const float temp = __fmaf_rn(p.polemass_length * (theta_dot * theta_dot), sintheta, force) * p.inv_total_mass;
    // thetaacc = (g sin - cos temp) / (length (4/3 - masspole cos^2 / total_mass))
    const float num = __fmaf_rn(-costheta, temp, p.gravity * sintheta);
    const float den = p.length * __fmaf_rn(-(p.masspole * (costheta * costheta)), p.inv_total_mass, 4.0f / 3.0f);
    const float thetaacc = div_fast(num, den);
    // xacc = temp - polemass_length * thetaacc * cos / total_mass
    const float xacc = __fmaf_rn(-((p.polemass_length * thetaacc) * costheta), p.inv_total_mass, temp);
    s[0] = __fmaf_rn(p.tau, x_dot, x);
    s[1] = __fmaf_rn(p.tau, xacc, x_dot);
    s[2] = __fmaf_rn(p.tau, theta_dot, theta);
    s[3] = __fmaf_rn(p.tau, thetaacc, theta_dot);
    const bool terminated = out_of_bounds(s[0], s[2]);
    reward = (terminated && was_terminated) ? 0.0f : 1.0f;
    return terminated;
  }

  __device__ static __forceinline__ void observe(const float (&s)[S], const Aux&, float (&o)[D]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = s[i];
  }

  // carl_cartpole.py:51-61: state = U(initial_state_lower, initial_state_upper, 4)
  __device__ static __forceinline__ void reset(const Params& p, const u32x4& w, float (&s)[S]) {
    const float lo = p.init_lo, hi = p.init_hi;
    s[0] = uniform_between(lo, hi, w.x);
    s[1] = uniform_between(lo, hi, w.y);
    s[2] = uniform_between(lo, hi, w.z);
    s[3] = uniform_between(lo, hi, w.w);
  }
};

// ================================ Pendulum ========================================
// features: carl_pendulum.py:15-39.  Quirk P1: row 0 ("gravity", default 8.0) is never
// read by the physics; real gravity is "g".
struct Pendulum {
  static constexpr int S = 2, D = 3, F = 7;
  using Action = float;
  enum { GRAVITY_DEAD, DT, G, M, L, INIT_ANGLE_MAX, INIT_VEL_MAX };
  static constexpr bool kNeedsStepNoise = false;
  static constexpr bool kPredraw = true;
  static constexpr int kDeepBelowLanes = 32768;

  // `3*g/(2*l)*sin(th)` and `3.0/(m*l**2)*u` evaluate left to right, so the two
  // quotients are per-context constants with the reference's own rounding order
  struct Params {
    float c_sin;  // 3 g / (2 l)
    float c_u;    // 3 / (m l^2)
    float dt;
    float init_angle_max, init_vel_max;
  };
  // sin/cos of the current angle: the observation of step t and the torque term of
  // step t+1 need the same pair, so one sincos per step serves both
  struct Aux {
    float sn, cs;
  };

  template <class Ctx>
  __device__ static __forceinline__ Params load(const Ctx& ctx, int c, int) {
    const float g = ctx.get(G, c), m = ctx.get(M, c), l = ctx.get(L, c);
    return Params{3.0f * g / (2.0f * l), 3.0f / (m * (l * l)), ctx.get(DT, c), ctx.get(INIT_ANGLE_MAX, c),
                  ctx.get(INIT_VEL_MAX, c)};
  }

  __device__ static __forceinline__ void prepare(const float (&s)[S], Aux& a) { sincos_fast_pk(s[0], a.sn, a.cs); }

  // PendulumEnv.step; reward from the OLD (th, thdot); never terminates
  __device__ static __forceinline__ bool step(const Params& p, float (&s)[S], Aux& aux, float action,
                                              float /*noise*/, int /*elapsed*/, float& reward) {
    // (every rounding spelled out, no implicit contraction: see CartPole::step)
#pragma clang fp contract(off)
    const float max_speed = 8.0f, max_torque = 2.0f;
    const float th = s[0], thdot = s[1];
    const float u = fminf(fmaxf(action, -max_torque), max_torque);
    // angle_normalize(x) = ((x + pi) % (2 pi)) - pi with Python's floor-mod
    const float two_pi = 2.0f * kPi, inv_two_pi = 1.0f / (2.0f * kPi);
    const float y = th + kPi;
    // r may land a rounding error outside [0, 2 pi); the reference's own float64 mod has the
    // same ambiguity at the seam, and an^2 is continuous there ((-pi - e)^2 vs (pi - e)^2
    // differ by 4 pi e), so no fix-up is needed for the cost
    const float r = __fmaf_rn(-floorf(y * inv_two_pi), two_pi, y);
    const float an = r - kPi;
    // costs = an^2 + 0.1 thdot^2 + 0.001 u^2
    const float costs = __fmaf_rn(an, an, __fmaf_rn(0.1f, thdot * thdot, 0.001f * (u * u)));
    // newthdot = thdot + (3 g / (2 l) sin th + 3 / (m l^2) u) dt
    float newthdot = __fmaf_rn(__fmaf_rn(p.c_sin, aux.sn, p.c_u * u), p.dt, thdot);
    newthdot = fminf(fmaxf(newthdot, -max_speed), max_speed);
    s[0] = __fmaf_rn(newthdot, p.dt, th);
    s[1] = newthdot;
    sincos_fast_pk(s[0], aux.sn, aux.cs);
    reward = -costs;
    return false;
  }

  // _get_obs / carl_pendulum.py:61
  __device__ static __forceinline__ void observe(const float (&s)[S], const Aux& a, float (&o)[D]) {
    o[0] = a.cs;
    o[1] = a.sn;
    o[2] = s[1];
  }

  // carl_pendulum.py:44-60: theta = U(0, initial_angle_max), thdot = U(0, initial_velocity_max)
  // (one-sided: `low` defaults to 0 -- Quirk P2)
  __device__ static __forceinline__ void reset(const Params& p, const u32x4& w, float (&s)[S]) {
    s[0] = uniform_between(0.0f, p.init_angle_max, w.x);
    s[1] = uniform_between(0.0f, p.init_vel_max, w.y);
  }
};

// ---- fp64 sin/cos from an LDS table (Acrobot) -------------------------------------
// Acrobot's RK4 needs eight fp64 sin/cos pairs per step (the reference's float64 arithmetic is needed there,
// see AcrobotT), and with the polynomial sincos_fast they were two thirds of the step's 562 vector
// instructions (46 each: 3-term reduction, two degree-6 polynomials in z, and the quadrant swap / sign logic on
// 64-bit values).  Here x = i * (pi / 256) + r with |r| <= pi / 512: (sin, cos) of the grid point come from a
// 512-entry table of correctly rounded doubles (8 KiB of LDS, staged once per workgroup from a constant array;
// generated by tools/gen_sincos_table.py), sin r and cos r need two terms each, and
//   sin x = S cos r + C sin r,   cos x = C cos r - S sin r
// -- 13 vector instructions + the table read, no quadrant logic; max error < 1e-13 (tests/test_sincos_table.py: table
// entries against mpmath, the formula against long-double libm over +-40 rad).  Round 4 shortened the evaluation (magic-number
// rounding, one-term reduction, sin r without its r^5 term, cos r without its r^4 term: 6e-11) because Acrobot is
// vector-ALU-bound.  Round 6 took the cosine's r^4 / 24 term BACK (one fma per angle): inside the reference's
// declared context bounds (link masses / lengths / MOI x 3, MAX_VEL x 3, velocities up to those bounds) a transition
// amplifies an error of its stage trig by up to 1e8, and 6e-11 put 265 of 131 072 such rows outside the 1e-5 bar (worst
// 0.11; tools/fuzz_wide_contexts.py); with 7e-14 (the sine's dropped r^5 / 120) 5 are left, 4 of them rows
// whose float64 ORACLE moves by more than 1e-7 when its input moves by 1e-15.
#include "sincos_table.inc"
__device__ const double kSinCosTab[2 * CARL_SINCOS_TAB_N] = {CARL_SINCOS_TAB_VALUES};

struct SinCosTab {
  typedef double vd2 __attribute__((ext_vector_type(2)));
  __device__ static __forceinline__ vd2* lds() {
    __shared__ vd2 tab[CARL_SINCOS_TAB_N];
    return tab;
  }
  // every thread of the workgroup; the caller synchronises before the first lookup
  __device__ static __forceinline__ void stage() {
    vd2* t = lds();
    const vd2* g = reinterpret_cast<const vd2*>(kSinCosTab);
    for (int i = threadIdx.x; i < CARL_SINCOS_TAB_N; i += blockDim.x) t[i] = g[i];
  }
  // The same evaluation in two halves (Acrobot's RK4: the lookups of stage n + 1 are issued BEFORE stage n's algebra, the
  // polynomials and the combination follow it -- same operations, same values as sincos2).
  struct Pending {
    vd2 ea, eb;
    double ra, rb;
  };
  __device__ static __forceinline__ Pending lookup2(double xa, double xb) {
    const vd2* t = lds();
    const double inv = CARL_SINCOS_TAB_INV_STEP, hi = CARL_SINCOS_TAB_STEP_HI;
    const double magic = 0x1.8p52;
    const double ta = fma(xa, inv, magic), tb = fma(xb, inv, magic);
    const double ka = ta - magic, kb = tb - magic;
    Pending q;
    q.ea = t[__double2loint(ta) & (CARL_SINCOS_TAB_N - 1)];
    q.eb = t[__double2loint(tb) & (CARL_SINCOS_TAB_N - 1)];
    q.ra = fma(ka, -hi, xa);
    q.rb = fma(kb, -hi, xb);
    return q;
  }
  __device__ static __forceinline__ void finish2(const Pending& q, double& sna, double& csa, double& snb, double& csb) {
    const double ra = q.ra, rb = q.rb;
    const double za = ra * ra, zb = rb * rb;
    const double sra = fma(ra * za, -1.0 / 6.0, ra);
    const double srb = fma(rb * zb, -1.0 / 6.0, rb);
    const double cra = fma(za, fma(za, 1.0 / 24.0, -0.5), 1.0);
    const double crb = fma(zb, fma(zb, 1.0 / 24.0, -0.5), 1.0);
    sna = fma(q.ea.x, cra, q.ea.y * sra);
    csa = fma(q.ea.y, cra, -(q.ea.x * sra));
    snb = fma(q.eb.x, crb, q.eb.y * srb);
    csb = fma(q.eb.y, crb, -(q.eb.x * srb));
  }
  // two angles at once (an RK4 stage): both table reads are issued before the polynomials
  __device__ static __forceinline__ void sincos2(double xa, double xb, double& sna, double& csa, double& snb,
                                                 double& csb) {
    const vd2* t = lds();
    const double inv = CARL_SINCOS_TAB_INV_STEP, hi = CARL_SINCOS_TAB_STEP_HI;
    // k = rint(x / step) by the 1.5 * 2^52 trick: the sum's low mantissa bits ARE the integer (two's complement), so
    // the table index is the low dword of `ta` -- no v_rndne_f64 + v_cvt_i32_f64 (|x / step| < 2^31: |x| < 2.6e7 rad)
    const double magic = 0x1.8p52;
    const double ta = fma(xa, inv, magic), tb = fma(xb, inv, magic);
    const double ka = ta - magic, kb = tb - magic;
    const vd2 ea = t[__double2loint(ta) & (CARL_SINCOS_TAB_N - 1)], eb = t[__double2loint(tb) & (CARL_SINCOS_TAB_N - 1)];
    // r = x - k * step with step = hi alone: the dropped k * lo is < 2e-15 for |x| <= 40 rad (lo = 4.8e-19)
    const double ra = fma(ka, -hi, xa), rb = fma(kb, -hi, xb);
    const double za = ra * ra, zb = rb * rb;
    // sin r = r - r z / 6 (dropped r^5 / 120 <= 7.3e-14 at |r| <= pi / 512), cos r = 1 - z / 2 + z^2 / 24 (dropped z^3 / 720 <= 7e-17)
    const double sra = fma(ra * za, -1.0 / 6.0, ra);
    const double srb = fma(rb * zb, -1.0 / 6.0, rb);
    const double cra = fma(za, fma(za, 1.0 / 24.0, -0.5), 1.0);
    const double crb = fma(zb, fma(zb, 1.0 / 24.0, -0.5), 1.0);
    sna = fma(ea.x, cra, ea.y * sra);
    csa = fma(ea.y, cra, -(ea.x * sra));
    snb = fma(eb.x, crb, eb.y * srb);
    csb = fma(eb.y, crb, -(eb.x * srb));
  }
};

// ================================ Acrobot =========================================
// features: carl_acrobot.py:15-69
//
// Arithmetic type: RK4 at dt = 0.2 from |w2| near MAX_VEL_2 = 9 pi runs through stage
// accelerations ~1e3 and stage velocities ~1e2, whose squares cancel in phi1 / ddtheta2;
// plain fp32 loses up to 1e-3 relative on ~1 % of such states (measured with the CPU
// oracle's fp32 variant).  So _dsdt/rk4 are evaluated in Real = double by default
// (fp32 state in HBM, 1e-5 parity on every row) and in Real = float when the caller
// sets CARL_FLAG_ACROBOT_FP32 (about the error above, several times the throughput).
template <class Real>
struct AcrobotT {
  static constexpr int S = 4, D = 6, F = 14;
  using Action = int;
  enum { L1, L2, M1, M2, C1, C2, MOI, MAXV1, MAXV2, NOISE, IA_LO, IA_HI, IV_LO, IV_HI };
  static constexpr bool kNeedsStepNoise = true;
  static constexpr bool kPredraw = true;

  // Context-only combinations of _dsdt, formed once per launch / reset instead of in each of the four RK4
  // stages (they regroup the reference's left-to-right products; in float64 that moves results by ~1e-16
  // relative, nine orders below the parity bar):
  //   A = m2 l1 lc2            B = (m1 lc1 + m2 l1) g        C = m2 lc2 g
  //   D = m2 lc2^2 + I2        E = m1 lc1^2 + m2 (l1^2 + lc2^2) + I1 + I2
  struct Params {
    Real A, B, C, D, E;
    Real max_vel_1, max_vel_2;
    float noise_max;
    float ia_lo, ia_hi, iv_lo, iv_hi;
  };
  struct Aux {
    float c0, s0, c1, s1;  // cos/sin of theta1, theta2 of the unrounded new angles (observation)
    Real ks0, kc0, ks1, kc1;  // ... of the STORED (float32) angles: the next step's first RK4 stage
  };

  template <class Ctx>
  __device__ static __forceinline__ Params load(const Ctx& ctx, int c, int) {
    Params p;
    const Real m1 = ctx.get(M1, c), m2 = ctx.get(M2, c), l1 = ctx.get(L1, c);
    const Real lc1 = ctx.get(C1, c), lc2 = ctx.get(C2, c), moi = ctx.get(MOI, c);
    const Real g = (Real)9.8;  // AcrobotEnv._dsdt: literal
    p.A = m2 * l1 * lc2;
    p.B = (m1 * lc1 + m2 * l1) * g;
    p.C = m2 * lc2 * g;
    p.D = m2 * (lc2 * lc2) + moi;
    p.E = m1 * (lc1 * lc1) + m2 * (l1 * l1 + lc2 * lc2) + moi + moi;
    p.max_vel_1 = ctx.get(MAXV1, c);
    p.max_vel_2 = ctx.get(MAXV2, c);
    p.noise_max = ctx.get(NOISE, c);
    p.ia_lo = ctx.get(IA_LO, c);
    p.ia_hi = ctx.get(IA_HI, c);
    p.iv_lo = ctx.get(IV_LO, c);
    p.iv_hi = ctx.get(IV_HI, c);
    return p;
  }

  // the fp64 variant reads sin / cos of the table's grid points from LDS: every kernel instantiated with this
  // family stages the table first (engine_kernels.hip.h: stage_family_tables)
  static constexpr bool kUsesSinCosTab = std::is_same_v<Real, double>;
  __device__ static __forceinline__ void stage_tables() {
    if constexpr (kUsesSinCosTab) SinCosTab::stage();
  }

  __device__ static __forceinline__ void prepare(const float (&s)[S], Aux& a) {
    sincos_pair((Real)s[0], (Real)s[1], a.ks0, a.kc0, a.ks1, a.kc1);
    a.s0 = (float)a.ks0; a.c0 = (float)a.kc0;
    a.s1 = (float)a.ks1; a.c1 = (float)a.kc1;
  }

  struct Deriv {
    Real d0, d1, d2, d3;
  };

  // sin/cos of an RK4 stage angle or of the new angle: |x| <= pi + a few turns (velocities are clipped), so
  // the fp64 version runs without the library fallback for huge arguments (fast_math.hip.h)
  __device__ static __forceinline__ void sincos_stage(Real x, Real& sn, Real& cs) {
    if constexpr (std::is_same_v<Real, double>)
      sincos_fast<false>(x, sn, cs);
    else
      sincos_fast(x, sn, cs);
  }
  __device__ static __forceinline__ void sincos_pair(Real xa, Real xb, Real& sa, Real& ca, Real& sb, Real& cb) {
    if constexpr (std::is_same_v<Real, double>) {
      SinCosTab::sincos2(xa, xb, sa, ca, sb, cb);
    } else {
      sincos_fast(xa, sa, ca);
      sincos_fast(xb, sb, cb);
    }
  }
  __device__ static __forceinline__ Real recip(Real d) {
    if constexpr (std::is_same_v<Real, double>)
      return rcp_fast1(d);
    else
      return (Real)1.0 / d;
  }

  // AcrobotEnv._dsdt, book_or_nips == "book", g = 9.8 literal.
  // cos(theta1 + theta2 - pi/2) and cos(theta1 - pi/2) are sin(theta1 + theta2) and
  // sin(theta1) (differences ~ulp(pi/2)); with sin/cos of both angles in hand,
  // sin(theta1 + theta2) = s1 c2 + c1 s2, so a derivative costs two sincos instead of
  // one sincos + two cos.
  __device__ static __forceinline__ Deriv dsdt(const Params& p, Real theta1, Real theta2, Real dtheta1,
                                               Real dtheta2, Real a) {
    Real s1, c1, s2, c2;
    sincos_pair(theta1, theta2, s1, c1, s2, c2);
    return dsdt_trig(p, s1, c1, s2, c2, dtheta1, dtheta2, a);
  }

  __device__ static __forceinline__ Deriv dsdt_trig(const Params& p, Real s1, Real c1, Real s2, Real c2, Real dtheta1,
                                                    Real dtheta2, Real a) {
    // d1 = m1 lc1^2 + m2 (l1^2 + lc2^2 + 2 l1 lc2 cos t2) + I1 + I2;  d2 = m2 (lc2^2 + l1 lc2 cos t2) + I2
    // phi2 = m2 lc2 g sin(t1 + t2);  phi1 = -m2 l1 lc2 w2^2 sin t2 - 2 m2 l1 lc2 w2 w1 sin t2 + (m1 lc1 + m2 l1) g sin t1 + phi2
    // ddtheta2 = (a + d2 / d1 phi1 - m2 l1 lc2 w1^2 sin t2 - phi2) / (m2 lc2^2 + I2 - d2^2 / d1);  ddtheta1 = -(d2 ddtheta2 + phi1) / d1
    // The two lines are the 2x2 system  [d1 d2; d2 D] (ddtheta1, ddtheta2) = (-phi1, rhs2)  with
    // rhs2 = a - A sin t2 w1^2 - phi2, solved by Cramer's rule with ONE reciprocal (of det = d1 D - d2^2) instead of
    // the reference's two divisions: ddtheta2 = (d1 rhs2 + d2 phi1) / det, ddtheta1 = -(d2 rhs2 + D phi1) / det.
    const Real s12 = s1 * c2 + c1 * s2;
    const Real ac2 = p.A * c2;
    const Real d1 = p.E + (Real)2.0 * ac2;
    const Real d2 = p.D + ac2;
    const Real phi2 = p.C * s12;
    const Real as2 = p.A * s2;
    const Real phi1 = p.B * s1 + phi2 - (as2 * dtheta2) * (dtheta2 + (Real)2.0 * dtheta1);
    const Real rhs2 = a - (as2 * dtheta1) * dtheta1 - phi2;
    const Real rdet = recip(d1 * p.D - d2 * d2);
    const Real ddtheta2 = (d1 * rhs2 + d2 * phi1) * rdet;
    const Real ddtheta1 = -(d2 * rhs2 + p.D * phi1) * rdet;
    return Deriv{dtheta1, dtheta2, ddtheta1, ddtheta2};
  }

  // wrap(x, -pi, pi): while x > M: x -= 2pi; while x < m: x += 2pi (strict compares)
  __device__ static __forceinline__ Real wrap_pi(Real x) {
    const Real pi = (Real)3.14159265358979323846;
    const Real diff = pi - (-pi);
    // the reference's loops, bounded: a step moves an angle by at most a few turns (velocities are
    // clipped), so eight iterations reproduce them exactly; an absurd input (e.g. an action far
    // outside Discrete(3), which gymnasium would reject with an assert) must not spin a wavefront
    // for millions of iterations -- it is reduced in one go instead
    for (int it = 0; it < 8 && x > pi; ++it) x = x - diff;
    for (int it = 0; it < 8 && x < -pi; ++it) x = x + diff;
    if (!(x <= pi && x >= -pi) && x == x && (x - x) == (Real)0) x = x - diff * floor((x + pi) / diff);
    return x;
  }

  // Both angles of a step.  One turn in either direction is the common case (velocities are clipped at a few
  // rad per step), so it is done with selects -- exactly the first iteration of each of the reference's loops --
  // and the loops themselves run only when some lane of the wave is still out of range afterwards (four
  // exec-masked loop nests and ~90 scalar instructions per step otherwise).
  __device__ static __forceinline__ void wrap_pair(Real& x0, Real& x1) {
    const Real pi = (Real)3.14159265358979323846;
    const Real diff = pi - (-pi);
    if constexpr (std::is_same_v<Real, double>) {
      // Round 4: k = rint(x / 2 pi) by the 1.5 * 2^52 trick and ONE fma, y = x - k * diff -- for one turn exactly the
      // reference's `x - diff` / `x + diff` (one rounding of the same difference), for several turns its loop unrolled;
      // the strict compares at +-pi are reproduced except for the three doubles within one ulp of +-pi, where x / 2 pi
      // rounds onto 0.5 (never met: 1e-16 of the angle range; the observation's sin / cos are the same there anyway).
      // 4 instructions per angle instead of 8 + a wave-uniform "several turns" test and branch (Acrobot is
      // vector-ALU-bound: DESIGN 4.3).  Non-finite angles stay non-finite.
      const double inv = 0x1.45f306dc9c883p-3, magic = 0x1.8p52;  // 1 / (2 pi)
      const double k0 = fma(x0, inv, magic) - magic, k1 = fma(x1, inv, magic) - magic;
      x0 = fma(k0, -diff, x0);
      x1 = fma(k1, -diff, x1);
      return;
    }
    Real y0 = x0 > pi ? x0 - diff : x0;
    Real y1 = x1 > pi ? x1 - diff : x1;
    y0 = y0 < -pi ? y0 + diff : y0;
    y1 = y1 < -pi ? y1 + diff : y1;
    const bool more = !(y0 <= pi && y0 >= -pi && y1 <= pi && y1 >= -pi);  // several turns, or non-finite
    if (__builtin_expect(ballot(more) != 0ull, 0)) {
      y0 = wrap_pi(x0);
      y1 = wrap_pi(x1);
    }
    x0 = y0;
    x1 = y1;
  }

  // AcrobotEnv.step: rk4 over [0, dt = 0.2] on (state, torque), wrap, bound, _terminal
  __device__ static __forceinline__ bool step(const Params& p, float (&s)[S], Aux& aux, int action, float noise,
                                              int /*elapsed*/, float& reward) {
    const Real dt = (Real)0.2, dt2 = dt / (Real)2.0;
    const Real a = (Real)((float)(action - 1) + noise);
    const Real y0 = s[0], y1 = s[1], y2 = s[2], y3 = s[3];
    // first stage at the stored state: its trig was carried over from the previous step / reset.
    // A stage's ANGLES are the previous stage's velocities integrated -- (d0, d1) of a Deriv are its input velocities, not
    // its accelerations -- so the angles of stage n + 1 are known as soon as stage n - 1's accelerations are: stage 2's at
    // the step's start, stage 3's after stage 1's algebra, stage 4's after stage 2's.  Each stage's table lookups and
    // sin / cos are therefore written BEFORE the preceding stage's algebra (same expressions, same values): the LDS reads
    // are in flight and the independent float64 chain is there to issue while v_rcp_f64 and the reads complete -- a
    // lone wavefront per SIMD (65 536 lanes) has nothing else to hide them behind.
    Real s1b, c1b, s2b, c2b, s1c, c1c, s2c, c2c, s1d, c1d, s2d, c2d;
    Deriv k1, k2, k3;
    if constexpr (std::is_same_v<Real, double>) {
      // (the table form in its two halves: lookups, THEN the preceding stage's algebra, then polynomials + combination)
      const SinCosTab::Pending qb = SinCosTab::lookup2(y0 + dt2 * y2, y1 + dt2 * y3);  // stage 2: k1.d0 = y2, k1.d1 = y3
      __builtin_amdgcn_sched_barrier(0);
      k1 = dsdt_trig(p, aux.ks0, aux.kc0, aux.ks1, aux.kc1, y2, y3, a);
      SinCosTab::finish2(qb, s1b, c1b, s2b, c2b);
    } else {
      sincos_pair(y0 + dt2 * y2, y1 + dt2 * y3, s1b, c1b, s2b, c2b);
      k1 = dsdt_trig(p, aux.ks0, aux.kc0, aux.ks1, aux.kc1, y2, y3, a);
    }
    const Real w1b = y2 + dt2 * k1.d2, w2b = y3 + dt2 * k1.d3;  // stage 2's velocities = k2.d0, k2.d1
    if constexpr (std::is_same_v<Real, double>) {
      const SinCosTab::Pending qc = SinCosTab::lookup2(y0 + dt2 * w1b, y1 + dt2 * w2b);  // stage 3
      __builtin_amdgcn_sched_barrier(0);
      k2 = dsdt_trig(p, s1b, c1b, s2b, c2b, w1b, w2b, a);
      SinCosTab::finish2(qc, s1c, c1c, s2c, c2c);
    } else {
      sincos_pair(y0 + dt2 * w1b, y1 + dt2 * w2b, s1c, c1c, s2c, c2c);
      k2 = dsdt_trig(p, s1b, c1b, s2b, c2b, w1b, w2b, a);
    }
    const Real w1c = y2 + dt2 * k2.d2, w2c = y3 + dt2 * k2.d3;  // = k3.d0, k3.d1
    if constexpr (std::is_same_v<Real, double>) {
      const SinCosTab::Pending qd = SinCosTab::lookup2(y0 + dt * w1c, y1 + dt * w2c);  // stage 4
      __builtin_amdgcn_sched_barrier(0);
      k3 = dsdt_trig(p, s1c, c1c, s2c, c2c, w1c, w2c, a);
      SinCosTab::finish2(qd, s1d, c1d, s2d, c2d);
    } else {
      sincos_pair(y0 + dt * w1c, y1 + dt * w2c, s1d, c1d, s2d, c2d);
      k3 = dsdt_trig(p, s1c, c1c, s2c, c2c, w1c, w2c, a);
    }
    const Real w1d = y2 + dt * k3.d2, w2d = y3 + dt * k3.d3;  // stage 4's velocities = k4.d0, k4.d1
    // The NEW ANGLES need stage 4's velocities only (k4.d0 = w1d, k4.d1 = w2d), not its accelerations: they are formed,
    // wrapped and rounded BEFORE stage 4's algebra, and the step's one trig evaluation at the stored angles (below) has its
    // lookups in flight through that algebra.
    const Real two = (Real)2.0, six = (Real)6.0;
    Real n0 = y0 + dt / six * (k1.d0 + two * k2.d0 + two * k3.d0 + w1d);
    Real n1 = y1 + dt / six * (k1.d1 + two * k2.d1 + two * k3.d1 + w2d);
    wrap_pair(n0, n1);
    s[0] = (float)n0;
    s[1] = (float)n1;
    // One trig evaluation per step, at the STORED (float32) angles: it is exactly what the next step's first RK4 stage
    // needs, and it serves _terminal (-cos t1 - cos(t1 + t2) > 1, cos(t1 + t2) = c0 c1 - s0 s1) and the observation
    // as well -- the reference takes those from its unrounded float64 state, 2e-7 away at most (the float32 rounding
    // of an angle in [-pi, pi]): fifty times inside the 1e-5 bar, and a terminal decision can only differ for a state
    // within 2e-7 of the threshold.  (Round 4 evaluated the unrounded angles and corrected to the stored ones with
    // sin(x + d) = sin x + d cos x: ten float64 instructions more per step on a vector-ALU-bound kernel.)
    Real s0r, c0r, s1r, c1r;
    Deriv k4;
    if constexpr (std::is_same_v<Real, double>) {
      const SinCosTab::Pending qe = SinCosTab::lookup2((Real)s[0], (Real)s[1]);
      __builtin_amdgcn_sched_barrier(0);
      k4 = dsdt_trig(p, s1d, c1d, s2d, c2d, w1d, w2d, a);
      SinCosTab::finish2(qe, s0r, c0r, s1r, c1r);
    } else {
      k4 = dsdt_trig(p, s1d, c1d, s2d, c2d, w1d, w2d, a);
      sincos_pair((Real)s[0], (Real)s[1], s0r, c0r, s1r, c1r);
    }
    Real n2 = y2 + dt / six * (k1.d2 + two * k2.d2 + two * k3.d2 + k4.d2);
    Real n3 = y3 + dt / six * (k1.d3 + two * k2.d3 + two * k3.d3 + k4.d3);
    // bound(x, m, M) = min(max(x, m), M)
    n2 = fmin(fmax(n2, -p.max_vel_1), p.max_vel_1);
    n3 = fmin(fmax(n3, -p.max_vel_2), p.max_vel_2);
    s[2] = (float)n2;
    s[3] = (float)n3;
    const bool terminated = (-c0r - (c0r * c1r - s0r * s1r)) > (Real)1.0;
    aux = Aux{(float)c0r, (float)s0r, (float)c1r, (float)s1r, s0r, c0r, s1r, c1r};
    reward = terminated ? 0.0f : -1.0f;
    return terminated;
  }

  // torque noise: only when torque_noise_max > 0 (AcrobotEnv.step)
  __device__ static __forceinline__ float step_noise(const Params& p, const carl_batch_t& b, uint64_t glane,
                                                     uint32_t episode, int elapsed) {
    if (!(p.noise_max > 0.0f)) return 0.0f;
    const u32x4 w = lane_words(b.seed, glane, episode, kSubStep0 + (uint32_t)elapsed);
    return uniform_between(-p.noise_max, p.noise_max, w.x);
  }

  // carl_acrobot.py:101-111
  __device__ static __forceinline__ void observe(const float (&s)[S], const Aux& a, float (&o)[D]) {
    o[0] = a.c0;
    o[1] = a.s0;
    o[2] = a.c1;
    o[3] = a.s1;
    o[4] = s[2];
    o[5] = s[3];
  }

  // carl_acrobot.py:78-100
  __device__ static __forceinline__ void reset(const Params& p, const u32x4& w, float (&s)[S]) {
    const float alo = p.ia_lo, ahi = p.ia_hi;
    const float vlo = p.iv_lo, vhi = p.iv_hi;
    s[0] = uniform_between(alo, ahi, w.x);
    s[1] = uniform_between(alo, ahi, w.y);
    s[2] = uniform_between(vlo, vhi, w.z);
    s[3] = uniform_between(vlo, vhi, w.w);
  }
};
using Acrobot = AcrobotT<double>;
using AcrobotFast = AcrobotT<float>;

// ================================ MountainCar =====================================
// features: carl_mountaincar.py:15-51 (Quirk M1: CARL's goal_position default 0.45)
struct MountainCar {
  static constexpr int S = 2, D = 2, F = 11;
  using Action = int;
  using Aux = NoAux;
  enum { MIN_POS, MAX_POS, MAX_SPEED, GOAL_POS, GOAL_VEL, FORCE, GRAVITY, MINP_START, MAXP_START, MINV_START, MAXV_START };
  static constexpr bool kNeedsStepNoise = false;
  static constexpr bool kPredraw = true;
  static constexpr bool kDeepActionPrefetch = true;  // (48-instruction step: engine_kernels.hip.h, rollout_staged_body)

  struct Params {
    float min_position, max_position, max_speed, goal_position, goal_velocity, force, gravity;
    float minp_start, maxp_start, minv_start, maxv_start;
  };

  template <class Ctx>
  __device__ static __forceinline__ Params load(const Ctx& ctx, int c, int) {
    return Params{ctx.get(MIN_POS, c),    ctx.get(MAX_POS, c),    ctx.get(MAX_SPEED, c),  ctx.get(GOAL_POS, c),
                  ctx.get(GOAL_VEL, c),   ctx.get(FORCE, c),      ctx.get(GRAVITY, c),    ctx.get(MINP_START, c),
                  ctx.get(MAXP_START, c), ctx.get(MINV_START, c), ctx.get(MAXV_START, c)};
  }

  __device__ static __forceinline__ void prepare(const float (&)[S], Aux&) {}

  // MountainCarEnv.step
  __device__ static __forceinline__ bool step(const Params& p, float (&s)[S], Aux&, int action, float /*noise*/,
                                              int /*elapsed*/, float& reward) {
#pragma clang fp contract(off)
    float position = s[0], velocity = s[1];
    // velocity += (action - 1) force + cos(3 position) (-gravity)     (roundings spelled out: see CartPole::step)
    // cos(3 position) as cos_twice_fast(1.5 position): fl(1.5 p) is exactly fl(3 p) / 2 (a power-of-two scaling)
    velocity = __fmaf_rn(cos_twice_fast(1.5f * position), -p.gravity, __fmaf_rn((float)(action - 1), p.force, velocity));
    velocity = fminf(fmaxf(velocity, -p.max_speed), p.max_speed);
    position += velocity;
    position = fminf(fmaxf(position, p.min_position), p.max_position);
    if (position == p.min_position && velocity < 0.0f) velocity = 0.0f;
    s[0] = position;
    s[1] = velocity;
    reward = -1.0f;
    return (position >= p.goal_position) && (velocity >= p.goal_velocity);
  }

  __device__ static __forceinline__ void observe(const float (&s)[S], const Aux&, float (&o)[D]) {
    o[0] = s[0];
    o[1] = s[1];
  }

  // carl_mountaincar.py:60-80
  __device__ static __forceinline__ void reset(const Params& p, const u32x4& w, float (&s)[S]) {
    s[0] = uniform_between(p.minp_start, p.maxp_start, w.x);
    s[1] = uniform_between(p.minv_start, p.maxv_start, w.y);
  }
};

// ============================ MountainCarContinuous ===============================
// features: carl_mountaincarcontinuous.py:15-48 (goal_position default 0.5)
struct MountainCarCont {
  static constexpr int S = 2, D = 2, F = 10;
  using Action = float;
  using Aux = NoAux;
  enum { MIN_POS, MAX_POS, MAX_SPEED, GOAL_POS, GOAL_VEL, POWER, MINP_START, MAXP_START, MINV_START, MAXV_START };
  static constexpr bool kNeedsStepNoise = false;
  static constexpr bool kPredraw = true;
  static constexpr bool kDeepActionPrefetch = true;

  struct Params {
    float min_position, max_position, max_speed, goal_position, goal_velocity, power;
    float minp_start, maxp_start, minv_start, maxv_start;
  };

  template <class Ctx>
  __device__ static __forceinline__ Params load(const Ctx& ctx, int c, int) {
    return Params{ctx.get(MIN_POS, c),    ctx.get(MAX_POS, c),    ctx.get(MAX_SPEED, c),  ctx.get(GOAL_POS, c),
                  ctx.get(GOAL_VEL, c),   ctx.get(POWER, c),      ctx.get(MINP_START, c), ctx.get(MAXP_START, c),
                  ctx.get(MINV_START, c), ctx.get(MAXV_START, c)};
  }

  __device__ static __forceinline__ void prepare(const float (&)[S], Aux&) {}

  // Continuous_MountainCarEnv.step (gravity literal 0.0025; penalty on the UNclipped action)
  __device__ static __forceinline__ bool step(const Params& p, float (&s)[S], Aux&, float action, float /*noise*/,
                                              int /*elapsed*/, float& reward) {
#pragma clang fp contract(off)
    float position = s[0], velocity = s[1];
    const float force = fminf(fmaxf(action, -1.0f), 1.0f);
    // velocity += force power - 0.0025 cos(3 position)                (roundings spelled out: see CartPole::step)
    velocity = __fmaf_rn(-0.0025f, cos_twice_fast(1.5f * position), __fmaf_rn(force, p.power, velocity));
    velocity = (velocity > p.max_speed) ? p.max_speed : velocity;
    velocity = (velocity < -p.max_speed) ? -p.max_speed : velocity;
    position += velocity;
    position = (position > p.max_position) ? p.max_position : position;
    position = (position < p.min_position) ? p.min_position : position;
    if (position == p.min_position && velocity < 0.0f) velocity = 0.0f;
    const bool terminated = (position >= p.goal_position) && (velocity >= p.goal_velocity);
    reward = __fmaf_rn(-(action * action), 0.1f, terminated ? 100.0f : 0.0f);
    s[0] = position;
    s[1] = velocity;
    return terminated;
  }

  __device__ static __forceinline__ void observe(const float (&s)[S], const Aux&, float (&o)[D]) {
    o[0] = s[0];
    o[1] = s[1];
  }

  // carl_mountaincarcontinuous.py:57-77
  __device__ static __forceinline__ void reset(const Params& p, const u32x4& w, float (&s)[S]) {
    s[0] = uniform_between(p.minp_start, p.maxp_start, w.x);
    s[1] = uniform_between(p.minv_start, p.maxv_start, w.y);
  }
};

}  // namespace carl
