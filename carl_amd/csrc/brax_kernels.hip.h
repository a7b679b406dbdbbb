// brax_kernels.hip.h -- Brax "spring" locomotion step on MI355X (Ant / Halfcheetah / Humanoid
// model tables; see include/carl_amd.h carl_brax_sys_t).
//
// Replaces, for N envs at once: brax.envs.<env>.step/reset -> n_frames x
// brax.spring.pipeline.step as reached from carl/envs/brax/carl_brax_env.py:163-190 and
// carl/envs/brax/wrappers.py:54-78 [brax 0.12.1 is not in the reference tree: the
// specification implemented here is written out in oracle/brax_spring.c's header and
// DESIGN.md; PARITY UNPINNED].
//
// Mapping (round 5): one env = a group of kSub adjacent lanes, ONE LANE PER LINK (Ant 9, Halfcheetah and Walker2d 7,
// Humanoid 11, Hopper 4; 16 for small batches; never narrower than the model's link count); one wavefront =
// floor(64 / kSub) envs, spare lanes idle.  Lane `sub` of an env owns link sub: its joint (the one to its parent) in the
// joint phase, its body in the body phase.  The kernel is bound by the instruction stream a wavefront issues, so every
// phase is a single round.  32 768 Ant envs are 4 682 wavefronts = 4.6 per SIMD (the round-1 one-lane-per-env kernel:
// 512 wavefronts for 1 024 SIMDs, each walking all links serially through LDS).  An env's maximal-coordinate state
// (one 80-byte record per link: pose 7 doubles, velocity 6 floats), the joints' reactions, masses, torques and the
// action / observation staging ([row][envs] float rows) live in LDS for the whole launch; the n_frames substeps -- and
// in the fused rollout all T env steps of a fragment -- never touch HBM for state.  THROUGH the substeps of an env step
// a lane keeps its own body, its joint's wrench on the child, the hinge torques and the branch hashes in REGISTERS
// (StepRegs); LDS carries what crosses lanes.  A substep is two lockstep phases with a wavefront-local hand-over:
//   A  the lane's joint:  geometry (float64 for every difference of poses) -> spring / damper / limit / actuator wrench
//                         on the child (registers) and the reaction on the parent (a 48-byte LDS record, no atomics);
//   B  the lane's body:   own wrench + its children's reactions (ascending) -> semi-implicit Euler velocity update ->
//                         this body's sphere / plane contacts -> integrate -> the new pose to its LDS record (the
//                         children read it in the next phase A; the owner never reads it back).
// Sums run in the oracle's order (own joint, then children ascending; colliders ascending).
// Per-env scalars (reward, done, counters) are computed redundantly by the lanes of the env from the same LDS data, so
// they agree without any exchange; lane sub == 0 writes them; between reward and reward they wait in LDS rows (stash).
// Every per-link constant of the two phases sits in ONE LinkRec per link (static LDS, expanded on the device once per
// workgroup).  No MFMA: with spring_inertia_scale = 1 the world inverse inertia R diag(1/I) R^T is a scalar.
// DESIGN.md 5.2 has the structure and what was measured; -DCARL_BRAX_PROFILE adds region clocks (Prof below).
#pragma once

#include <type_traits>

#include "carl_device.hip.h"
#include "fast_math.hip.h"

// (The CARL_EXP_BRAX_* measurement-only switches of round 3 -- cost attribution by leaving a piece of the substep out or
// replacing it with its float32 form -- were removed in round 4; results in profiles/r03_brax_occupancy.txt, the tree
// that builds them is commit 3a3c7b2.)

namespace carl {
namespace brax {

constexpr int kLanes = 64;  // lanes per wavefront: one wavefront works on floor(64 / kSub) envs, on its own
// Wavefronts per workgroup.  The wavefronts of a workgroup are INDEPENDENT (own envs, own slice of the dynamic LDS,
// no workgroup barrier after the start-up copy); they only share the one LDS copy of the model table, topology and
// derived constants (5.5 KB).  With one wavefront per workgroup that copy made LDS, not registers, the occupancy
// limit: Ant 10.7 KB of rows + 5.5 KB static = 9 wavefronts per CU.
// The host picks the number of wavefronts per workgroup (<= kMaxWavesPerWg; 8 keeps the 256-VGPR budget under
// __launch_bounds__) that puts the most wavefronts on a CU's 160 KB (carl_brax.hip: launch_brax).
#ifndef CARL_BRAX_WAVES_PER_WG
#define CARL_BRAX_WAVES_PER_WG 8
#endif
constexpr int kMaxWavesPerWg = CARL_BRAX_WAVES_PER_WG;
constexpr int kMaxThreads = kLanes * kMaxWavesPerWg;
// The single-hinge kernels run THREE wavefronts per SIMD (168 VGPRs), so a workgroup of theirs may be a CU's whole
// 12 wavefronts -- what the balanced fragment schedule of a large batch wants (run(): the fewer, larger workgroups
// the groups are dealt to, the smaller the rounding loss between workgroups).
constexpr int kMaxWavesPerWg3 = 12;
__host__ __device__ constexpr int max_waves_per_wg(bool task) { return task ? kMaxWavesPerWg : kMaxWavesPerWg3; }
// Register budget.  Single-hinge models (MULTI = false: Ant, Halfcheetah, Hopper, Walker2d): THREE wavefronts per SIMD
// (168 VGPRs) -- the kernel is bound by the latency of its own dependent chains (one -> two wavefronts per SIMD: 1.7 x),
// the hot substep loop fits 168 registers without a spill (ISA checked) and what spills (62 dwords) sits in observe and
// in the done path, once per env step or rarer: Ant 2.98e8 -> 3.55e8 env-steps/s.  The reset path is INLINED there: with
// reset_state / forward_kinematics as non-inlined calls the 168-register Ant build faulted on the device
// (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in the spilled call path).  Multi-hinge / task models (Humanoid, ...):
// two wavefronts (256 VGPRs; they are LDS-bound at 2.5 and lose 4-28 % at 168 / 128).  profiles/r03_brax_occupancy.txt.
// Since late round 6 the multi-hinge kernels run three wavefronts per SIMD too: with the anisotropic-inertia code gone from
// them (general kernels only) their substep loop fits 168 registers without a spill (ISA checked: 336 B of scratch, all of
// it in the per-step code), and with the reaction records overlaying the observation staging (Layout::make) twelve
// Humanoid wavefronts fit a compute unit's LDS: Humanoid 2.565 -> see DESIGN 5.2.  NINE wavefronts per compute unit (what
// the LDS allowed before the overlay) were 8 % SLOWER than eight: the SIMD that holds three of them finishes its equal
// share of the groups last.  The general (TASK) kernels keep two wavefronts per SIMD (256 registers).
#ifndef CARL_BRAX_WAVES_PER_EU_F32  // the float32-substep kernels (CARL_FLAG_BRAX_FP32; never a general kernel)
#define CARL_BRAX_WAVES_PER_EU_F32(TASK) 3
#endif
#ifndef CARL_BRAX_WAVES_PER_EU
#define CARL_BRAX_WAVES_PER_EU(TASK) ((TASK) ? 2 : 3)
#endif
#ifndef CARL_BRAX_RESET_INLINE
#define CARL_BRAX_RESET_INLINE __forceinline__
#endif
// kSub = lanes per env is a template parameter of everything below (Group<kSub>): 2, 4, 7, 9, 11 or
// 16, chosen per model and batch size by the host (carl_amd.hip: brax_lanes_per_env)
constexpr float kPiF = 3.14159265358979323846f;

// Region clocks (measurement build only: -DCARL_BRAX_PROFILE, tools/brax_region_profile.py): s_memtime at the region
// boundaries of the step kernel, summed per wavefront in scalar registers and added to a device array at the end; the
// product build compiles every mark to nothing.  Results do not change (the marks only read the clock).
enum ProfRegion { kProfLoad = 0, kProfPrologue, kProfJoints, kProfBodies, kProfEpilogue, kProfObserve, kProfReward,
                  kProfDone, kProfOutput, kProfStore, kProfBodySum, kProfContacts, kProfRegions };
#ifdef CARL_BRAX_PROFILE
__device__ unsigned long long g_brax_prof[kProfRegions + 1];  // [kProfRegions]: wavefronts
struct Prof {
  uint64_t t, acc[kProfRegions];
  __device__ __forceinline__ void start() {
    for (int k = 0; k < kProfRegions; ++k) acc[k] = 0;
    t = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void mark(int k) {
    const uint64_t n = __builtin_amdgcn_s_memtime();
    acc[k] += n - t;
    t = n;
  }
  __device__ __forceinline__ void flush(bool lane0) {
    if (lane0) {
      for (int k = 0; k < kProfRegions; ++k) atomicAdd(&g_brax_prof[k], (unsigned long long)acc[k]);
      atomicAdd(&g_brax_prof[kProfRegions], 1ull);
    }
  }
};
#else
struct Prof {
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void flush(bool) {}
};
#endif

struct v3 {
  float x, y, z;
};
struct qt {
  float w, x, y, z;
};
__host__ __device__ __forceinline__ v3 V(float x, float y, float z) { return v3{x, y, z}; }
__host__ __device__ __forceinline__ v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ __forceinline__ v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ __forceinline__ v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__host__ __device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ __forceinline__ v3 cross(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__host__ __device__ __forceinline__ qt qmul(qt a, qt b) {
  return qt{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__host__ __device__ __forceinline__ qt qconj(qt a) { return qt{a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ qt qnormalize(qt a) {
  const float inv = rsqrtf(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z);
  return qt{a.w * inv, a.x * inv, a.y * inv, a.z * inv};
}
__host__ __device__ __forceinline__ v3 qrot(qt q, v3 v) {
  const v3 u = V(q.x, q.y, q.z);
  const v3 t = cross(u, v) * 2.0f;
  return v + t * q.w + cross(u, t);
}
// q rotating (0, y, z)
__host__ __device__ __forceinline__ v3 qrot_yz(qt q, float y, float z) {
  const v3 t = V(q.y * z - q.z * y, -(q.x * z), q.x * y) * 2.0f;  // 2 u x v
  return V(t.x * q.w + (q.y * t.z - q.z * t.y), y + t.y * q.w + (q.z * t.x - q.x * t.z), z + t.z * q.w + (q.x * t.y - q.y * t.x));
}
__device__ __forceinline__ qt qaxis(int k, float angle) {
  float s, c;
  sincos_fast(0.5f * angle, s, c);
  return qt{c, k == 0 ? s : 0.0f, k == 1 ? s : 0.0f, k == 2 ? s : 0.0f};
}
__host__ __device__ __forceinline__ v3 f3(const float* p) { return V(p[0], p[1], p[2]); }
__host__ __device__ __forceinline__ qt f4(const float* p) { return qt{p[0], p[1], p[2], p[3]}; }

// Pose (COM position, rotation) in FLOAT64, velocities in float32.  The pipeline is stiff: a constraint spring
// turns an ABSOLUTE error e in the relative position / orientation of two bodies into k dt e of velocity per
// substep (k dt / m = 20 for Ant, 30 for Humanoid, 47 for Halfcheetah), and a float32 pose carries e ~ 1e-7 from
// its own rounding between substeps -- round 2's residue against the float64 restatement (Humanoid: 11.7 % of
// observation entries beyond north_star's 1e-5).  So everything that is a DIFFERENCE OF POSES -- anchor
// separation, relative rotation and the joint angles read from it, contact depth, the step's forward progress --
// is formed in float64 from a float64 pose; forces, torques, impulses and velocities (relative errors, never
// amplified) stay float32.  Cost: at two wavefronts per SIMD a float64 instruction issues at 0.63 x the float32 rate
// (profiles/r03_fp64_rate.txt); about a third of the substep's vector instructions are float64, -10 % throughput
// against the all-float32 kernel of round 2 (profiles/r03_brax_occupancy.txt has the attribution).
struct v3d {
  double x, y, z;
};
struct qtd {
  double w, x, y, z;
};
__device__ __forceinline__ v3d D(double x, double y, double z) { return v3d{x, y, z}; }
__device__ __forceinline__ v3d operator+(v3d a, v3d b) { return D(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3d operator-(v3d a, v3d b) { return D(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3d operator*(v3d a, double s) { return D(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot(v3d a, v3d b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3d cross(v3d a, v3d b) {
  return D(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ qtd qmul(qtd a, qtd b) {
  return qtd{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ qtd qconj(qtd a) { return qtd{a.w, -a.x, -a.y, -a.z}; }
// (0, w) (x) b: the quaternion product with a pure-vector left factor, without the four multiplications by its zero
__device__ __forceinline__ qtd qmul_vec(v3d w, qtd b) {
  return qtd{-(w.x * b.x) - w.y * b.y - w.z * b.z, w.x * b.w + w.y * b.z - w.z * b.y,
             w.y * b.w - w.x * b.z + w.z * b.x, w.z * b.w + w.x * b.y - w.y * b.x};
}
__device__ __forceinline__ v3d qrot(qtd q, v3d v) {
  const v3d u = D(q.x, q.y, q.z);
  const v3d t = cross(u, v) * 2.0;
  return v + t * q.w + cross(u, t);
}
__device__ __forceinline__ v3 tof(v3d a) { return V((float)a.x, (float)a.y, (float)a.z); }
__device__ __forceinline__ qt tof(qtd a) { return qt{(float)a.w, (float)a.x, (float)a.y, (float)a.z}; }
__device__ __forceinline__ v3d tod(v3 a) { return D(a.x, a.y, a.z); }
__device__ __forceinline__ qtd tod(qt a) { return qtd{a.w, a.x, a.y, a.z}; }

// ---- the arithmetic type of the substep's POSE algebra: double (the product path: everything above) or float --
// CARL_FLAG_BRAX_FP32, opt-in: what brax itself computes in under JAX's default precision, reported beside the product
// figures with its measured deviation.  The float forms reuse the float32 vector types.
template <class R> struct PoseT;
template <> struct PoseT<double> {
  using V3 = v3d;
  using Q = qtd;
  static __device__ __forceinline__ v3d mk(double x, double y, double z) { return D(x, y, z); }
  static __device__ __forceinline__ v3d from(v3 a) { return tod(a); }
};
template <> struct PoseT<float> {
  using V3 = v3;
  using Q = qt;
  static __device__ __forceinline__ v3 mk(float x, float y, float z) { return V(x, y, z); }
  static __device__ __forceinline__ v3 from(v3 a) { return a; }
};
__device__ __forceinline__ v3 tof(v3 a) { return a; }
__device__ __forceinline__ qt tof(qt a) { return a; }
__device__ __forceinline__ qt qmul_vec(v3 w, qt b) {
  return qt{-(w.x * b.x) - w.y * b.y - w.z * b.z, w.x * b.w + w.y * b.z - w.z * b.y,
            w.y * b.w - w.x * b.z + w.z * b.x, w.z * b.w + w.x * b.y - w.y * b.x};
}
__device__ __forceinline__ double fma_r(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float fma_r(float a, float b, float c) { return fmaf(a, b, c); }

template <class R>
struct BodyT {
  typename PoseT<R>::V3 p;
  typename PoseT<R>::Q r;
  v3 v, w;
};
using Body = BodyT<double>;
__device__ __forceinline__ BodyT<float> narrow_body(const Body& b) { return BodyT<float>{tof(b.p), tof(b.r), b.v, b.w}; }
__device__ __forceinline__ Body widen_body(const BodyT<float>& b) { return Body{tod(b.p), tod(b.r), b.v, b.w}; }
__device__ __forceinline__ Body widen_body(const Body& b) { return b; }

// atan2 in float64 to ~3e-13 (one octant reduction, one division, degree-7 polynomial in t^2 on
// [0, tan^2(pi/8)], Chebyshev-node fit): a joint angle is multiplied by the limit / joint / locking stiffness
// (k dt / I up to 30 per substep), so the 2.8e-7 of atan2_fast would reach the velocities at 1e-5 within one env
// step.  ~36 instructions.  XPOS: the caller guarantees x >= 0 (a hinge's half angle: the relative rotation's scalar
// part after the sign flip) -- the quadrant fix-up for x < 0 is compiled out.
// The division: v_rcp_f64 (~26 bits) and ONE Newton step (4e-15, fast_math.hip.h: rcp_fast1).
template <bool XPOS = false>
__device__ __forceinline__ double atan2_f64(double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
  const double mx = fmax(ax, ay), mn = fmin(ax, ay);
  const bool mid = mn > 0.41421356237309503 * mx;  // atan(t) = pi/4 + atan((t - 1) / (t + 1))
  const double num = mid ? mn - mx : mn, den = mid ? mn + mx : mx;
  double t = num * rcp_fast1(den);
  t = (mx > 0.0) ? t : 0.0;
  const double z = t * t;
  double p = fma(z, -3.76549087472088720e-02, 6.97418621847181036e-02);
  p = fma(z, p, -8.99255026739464586e-02);
  p = fma(z, p, 1.11034566044549116e-01);
  p = fma(z, p, -1.42853865353561232e-01);
  p = fma(z, p, 1.99999930530023323e-01);
  p = fma(z, p, -3.33333332769153445e-01);
  p = fma(z, p, 9.99999999999244826e-01);
  double r = fma(t, p, mid ? 0.78539816339744831 : 0.0);
  r = (ay > ax) ? 1.5707963267948966 - r : r;
  if (!XPOS) r = (x < 0.0) ? 3.1415926535897932 - r : r;
  return copysign(r, y);
}
// sqrt(x), 0 <= x <= 1, to ~2e-16 relative: v_rsq_f32 seed (1 ulp of float32) and two coupled Newton steps -- the library
// sqrt is v_rsq_f64 (a sixteen-cycle instruction) plus scaling and a correctly rounded finish the angle does not need.
__device__ __forceinline__ double sqrt01_f64(double x) {
  const double y = (double)__builtin_amdgcn_rsqf((float)x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-g, h, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return (x > 1e-30) ? g : 0.0;  // (x = 0: the seed is inf)
}
// asin(x) = atan2(x, sqrt((1 - x)(1 + x))), |x| <= 1
__device__ __forceinline__ double asin_f64(double x) { return atan2_f64<true>(x, sqrt01_f64((1.0 - x) * (1.0 + x))); }
// the same three by the substep's arithmetic type (float: the observation pass's float32 forms)
template <bool XPOS = false>
__device__ __forceinline__ double atan2_r(double y, double x) { return atan2_f64<XPOS>(y, x); }
template <bool XPOS = false>
__device__ __forceinline__ float atan2_r(float y, float x) { return atan2_fast(y, x); }
__device__ __forceinline__ double asin_r(double x) { return asin_f64(x); }
__device__ __forceinline__ float asin_r(float x) {
  const float c2 = fmaxf((1.0f - x) * (1.0f + x), 0.0f);
  return atan2_fast(x, c2 * __builtin_amdgcn_rsqf(fmaxf(c2, 1e-30f)));
}

// per-env context scalars: carl_brax_env.py:255-292 in its intended form
struct LaneCtx {
  float gravity_z, friction, elasticity, ang_damping, stiffness_scale;
  float da;  // exp(ang_damping dt): the substep's angular velocity decay
};

// LDS layout of a wavefront's slice.  First the BODY RECORDS, one per (env, link), [env][link] x 80 bytes:
//   pose, 7 doubles (COM position 3, rotation w x y z) | linear, angular velocity, 6 floats
// read and written as five 16-byte accesses per body.  (Round 3 measured the [row][env] layout these replace:
// the LDS array was busy 74 % of the kernel's CU-cycles, 40 % of that bank conflicts (SQ_LDS_IDX_ACTIVE /
// SQ_LDS_BANK_CONFLICT, profiles/r03_brax_lds_counters.txt) -- the substep was bound by the LDS pipe, not by
// issue: a body took four 8-byte and three 4-byte two-address reads at 128 B/clk with 2-3-way conflicts (row stride
// of 7 or 9 envs); the 80-byte record is 5 x 4 banks wide and 5 is coprime to 16, so the 16 lanes of a
// ds_read_b128 group land on 16 different bank quads at 256 B/clk.)  Then the float rows: each row = kEnvs
// consecutive floats, one per env.
constexpr int kBodyBytes = 80;
// Behind a wavefront's body records: the WORLD as a body record (what a world-jointed link reads as its parent: origin,
// identity rotation, at rest; written once per launch), then one 32-byte CONTEXT record per env -- gravity_z, friction,
// elasticity, exp(ang_damping dt) | joint-stiffness scale -- which the phases read instead of holding five registers
// per lane through the substeps.
constexpr int kWorldRecBytes = kBodyBytes, kCtxRecBytes = 32;
// A joint's reaction on its parent, handed from the joint phase to the body phase: (force 3, torque 3) as two 16-byte
// pieces of a 48-byte record per (env, link) -- 48 = 3 x 4 banks, 3 coprime to 16: the lanes of a ds_read_b128 group land
// on different bank quads.  Record L of an env is kept at ZERO: what a link without a k-th child sums in the body phase
// (LinkB::children names it), so no select is needed.  The records OVERLAY the wrench rows, which forward kinematics and
// observe use as scratch between substep loops (the zero records are re-written before every substep loop).
constexpr int kReactBytes = 48;
constexpr int kLinkRecBytes = 464;  // sizeof(Group<>::LinkRec): the substep's per-link constants, static LDS
constexpr int kStashRows = 12;     // the env's episode scalars, parked in LDS while the substeps run (run(): stash)
struct Layout {
  int L;       // links: body records per env
  int wrench;  // 12 * (L + 1) rows: the substep's reaction records (48 bytes per (env, link)); scratch rows of FK / observe
  int mass;    // L rows (effective mass per link, context-scaled)
  int sig;     // 2 * L rows (uint32): per-link hash of the step's contact / limit branch decisions
  int goal;    // 3 rows: push task, the env's goal position (context or model default)
  int stash;   // kStashRows rows
  int tau;     // n_dof rows
  int io;      // staging of the env's action / observation record, and of (q, qd) in reset
  int total;   // float rows
  // `overlay_at` >= 0: the reaction records OVERLAY the staging rows from row `overlay_at` (rounded up to a multiple of 4: the
  // records are read in 16-byte pieces) instead of having rows of their own.  The two are never alive together where the
  // caller allows it (layout_of below): the staging holds the actions before the substeps (rows [0, n_act)), (q, qd) during
  // a reset's kinematics (rows [0, n_q + n_dof)) and the observation after the substeps; the records live through the
  // substeps, and as the kinematics' scratch during a reset.  For the extended observation of the Humanoid models (244 rows
  // against 47 of (q, qd)) that is 144 rows = 2.9 KB less per wavefront: twelve wavefronts fit a compute unit's LDS instead
  // of eight.
  __host__ __device__ static Layout make(int L, int n_dof, int io_rows, int overlay_at = -1) {
    Layout l;
    l.L = L;
    const int wr = 12 * (L + 1);
    int shared;  // rows of the staging / record region
    if (overlay_at >= 0) {
      l.io = 0;
      l.wrench = (overlay_at + 3) & ~3;
      shared = io_rows > l.wrench + wr ? io_rows : l.wrench + wr;
      l.mass = shared;
      l.sig = l.mass + L;
      l.goal = l.sig + 2 * L;
      l.stash = l.goal + 3;
      l.tau = l.stash + kStashRows;
      l.total = l.tau + n_dof;
    } else {
      l.wrench = 0;
      l.mass = l.wrench + wr;
      l.sig = l.mass + L;
      l.goal = l.sig + 2 * L;
      l.stash = l.goal + 3;
      l.tau = l.stash + kStashRows;
      l.io = l.tau + n_dof;
      l.total = l.io + io_rows;
    }
    return l;
  }
  // bytes of dynamic LDS for `envs` envs per workgroup
  __host__ __device__ size_t body_bytes(int envs) const { return (size_t)kBodyBytes * L * envs; }
  __host__ __device__ size_t rows_offset(int envs) const {  // multiple of 16
    return body_bytes(envs) + kWorldRecBytes + (size_t)kCtxRecBytes * envs;
  }
  __host__ __device__ size_t bytes(int envs) const {  // per wavefront, rounded up to 16 (the next wavefront's records)
    return (rows_offset(envs) + (size_t)total * 4 * envs + 15) & ~(size_t)15;
  }
};

// Persistent state record of one env in HBM (carl_batch_t::state, ::first_state): CARL_BRAX_LINK_RECORD = 20
// floats PER LINK, link-major (ABI 8), each link's record being
//   [0, 7)    pose, float32 head:  COM position 3, rotation 4 (w, x, y, z)
//   [7, 14)   pose, float32 tail:  pose = (double)head + (double)tail  (48 significant bits)
//   [14, 20)  linear velocity 3, angular velocity 3
__host__ __device__ inline int io_rows_of(const carl_brax_sys_t& s) {
  const int qrows = s.n_q + s.n_dof;
  int r = s.obs_dim > qrows ? s.obs_dim : qrows;
  return r > s.n_act ? r : s.n_act;
}

// The LDS layout of a model's wavefront slice (kernel and host).  The records overlay the staging rows where nothing else
// uses the record rows as scratch while the observation is being laid out: not the tip models (observe parks the unclipped
// rates there), not the reach / push task models (they park and re-order rows there) -- i.e. for the extended observation
// (the only staging much larger than (q, qd) anyway).
__host__ __device__ inline Layout layout_of(const carl_brax_sys_t& s) {
  const bool overlay = s.obs_extended != 0 && s.tip_link <= 0 && s.target_link <= 0 && s.push_link <= 0;
  return Layout::make(s.n_links, s.n_dof, io_rows_of(s), overlay ? s.n_q + s.n_dof : -1);
}

// tree topology derived from the model table once per workgroup
struct Topo {
  uint8_t child_begin[CARL_BRAX_MAX_LINKS + 1], child_idx[CARL_BRAX_MAX_LINKS];
  uint8_t coll_begin[CARL_BRAX_MAX_LINKS + 1], coll_idx[CARL_BRAX_MAX_COLL];
  uint8_t depth[CARL_BRAX_MAX_LINKS];
  int max_depth;
  int first_joint;  // 1 when link 0 is a free root (it has no joint), else 0
};

// Built ONCE PER LAUNCH ON THE HOST (a few hundred integer operations) and passed to the kernel by value
// with the derived constants below: building them in the kernel -- by one lane ~55 us, by the
// workgroup in three parallel steps ~5 us -- sat on the critical path of every workgroup of every
// launch, which is what a per-call carl_brax_step pays in full.
inline void build_topo_host(const carl_brax_sys_t& s, Topo& t) {
  const int L = s.n_links;
  int nc = 0, nk = 0, md = 0;
  for (int i = 0; i < L; ++i) {
    t.child_begin[i] = (uint8_t)nc;
    for (int c = i + 1; c < L; ++c)
      if (s.parent[c] == i) t.child_idx[nc++] = (uint8_t)c;  // ascending: the oracle's summation order
    // pair contact: the object (last link, jointed to the world) leaves the reaction on the gripper in
    // its parent-side wrench rows, which the gripper sums like a child's
    if (s.n_pair > 0 && s.pair_link == i) t.child_idx[nc++] = (uint8_t)s.push_link;
    t.coll_begin[i] = (uint8_t)nk;
    for (int k = 0; k < s.n_coll; ++k)
      if (s.coll_link[k] == i) t.coll_idx[nk++] = (uint8_t)k;
    const int d = s.parent[i] < 0 ? 0 : t.depth[s.parent[i]] + 1;
    t.depth[i] = (uint8_t)d;
    md = d > md ? d : md;
  }
  t.child_begin[L] = (uint8_t)nc;
  t.coll_begin[L] = (uint8_t)nk;
  t.max_depth = md;
  t.first_joint = (s.parent[0] < 0 && s.n_link_dof[0] == 6) ? 1 : 0;
}

// Per-link records laid out for the substep's two phases (built on the host, once per launch).  A phase used to pick its
// constants out of a dozen arrays of carl_brax_sys_t / Topo indexed one after the other -- parent -> body rows,
// n_slide -> dof_start -> tau row, child_begin -> child_idx -> wrench rows, coll_begin -> coll_idx -> sphere -- each a
// dependent LDS round trip the wavefront waits for; here a phase reads ONE 16-byte-aligned block per link with
// ds_read_b128s issued together, and every index it needs is a bit field of a word of that block.
typedef float vf4 __attribute__((ext_vector_type(4)));
struct alignas(16) LinkA {  // joint phase (spring.joints.resolve) and inverse kinematics: 24 words
  float ac[3];              // joint anchor relative to the child's COM (child frame)
  uint32_t word;            // parent + 1 (5 bits; 0: the world) | free root (1) | n_slide (2) | hinges (3) | dof_start (5)
  float ap[3];              // ... relative to the parent's COM (parent frame; world: origin), zero slide
  float k_pos;
  float rpl[4];             // parent-side joint frame in the parent frame: link_rot (x) joint_rot
  float jrot[4];            // child-side joint frame in the child frame: joint_rot
  float k_vel, k_limit, k_ang_damp;
  float damping;            // the link's FIRST hinge (dof_start + n_slide): dof_damping, dof_stiffness, dof_lo, dof_hi
  float stiffness, lo, hi;
  float axis_sign;          // planar models: +-1, the joint frame's x axis is +-y of the link frame (else 0)
};
constexpr uint32_t kWaFree = 1u << 5;
__host__ __device__ inline int wa_parent(uint32_t w) { return (int)(w & 31u) - 1; }
__host__ __device__ inline int wa_slides(uint32_t w) { return (int)((w >> 6) & 3u); }
__host__ __device__ inline int wa_hinges(uint32_t w) { return (int)((w >> 8) & 7u); }
__host__ __device__ inline int wa_dof(uint32_t w) { return (int)((w >> 11) & 31u); }
struct alignas(16) LinkB {  // body phase: 4 words
  uint32_t word;            // free root (1) | isotropic inertia (1) | children (4) | first sphere (6) | spheres (6)
  uint32_t children;        // the first 4 children, one byte each, ascending (the oracle's summation order); none: n_links
  float inv_i0;             // inv_inertia[0] (all there is to an isotropic inertia: R diag(c) R^T = c)
  float reach;              // max over the link's spheres of |centre - COM| + radius (-1: none)
};
constexpr uint32_t kWbFree = 1u, kWbIso = 2u;
__host__ __device__ inline int wb_children(uint32_t w) { return (int)((w >> 2) & 15u); }
__host__ __device__ inline int wb_first_sphere(uint32_t w) { return (int)((w >> 6) & 63u); }
__host__ __device__ inline int wb_spheres(uint32_t w) { return (int)((w >> 12) & 63u); }
struct alignas(16) Sphere {  // a link's spheres are consecutive, in Topo::coll_idx order
  float off[3];              // centre - COM of its link (link frame)
  float radius;
};
// What the body phase reads per sphere (static LDS, widened on the device once per workgroup from Packed::sph): the
// offset and radius as DOUBLES -- the operands of the float64 depth, which is evaluated for every sphere of every link near
// the plane in every substep (four v_cvt_f64_f32 per sphere and substep until round 6) -- and the float32 record the
// impulse reads.  One record = one base address.
struct alignas(16) SphRec {
  double off[3], radius;  //  0
  float f[4];             // 32: off x y z, radius
};
constexpr int kSphRecBytes = 48;
static_assert(sizeof(SphRec) == kSphRecBytes, "SphRec is read in 16-byte pieces at fixed offsets");
struct Packed {
  LinkA a[CARL_BRAX_MAX_LINKS];
  LinkB b[CARL_BRAX_MAX_LINKS];
  Sphere sph[CARL_BRAX_MAX_COLL];
  int max_children;  // over the links (bound of the body phase's wavefront-uniform child loop)
  int all_iso;       // every link: inv_inertia[0] == [1] == [2]
  int pad[2];
};

inline void build_packed_host(const carl_brax_sys_t& s, const Topo& t, Packed& pk) {
  int mc = 0;
  for (int i = 0; i < s.n_links; ++i) {
    const int P = s.parent[i];
    const v3 a = f3(s.joint_pos[i]);
    const qt lrot = f4(s.link_rot[i]);
    const v3 com_p = (P < 0) ? V(0, 0, 0) : f3(s.com[P]);
    const v3 ac = a - f3(s.com[i]);
    const v3 ap = f3(s.link_pos[i]) + qrot(lrot, a) - com_p;
    const qt rpl = qmul(lrot, f4(s.joint_rot[i]));
    LinkA& A = pk.a[i];
    A.ac[0] = ac.x; A.ac[1] = ac.y; A.ac[2] = ac.z;
    A.ap[0] = ap.x; A.ap[1] = ap.y; A.ap[2] = ap.z;
    A.rpl[0] = rpl.w; A.rpl[1] = rpl.x; A.rpl[2] = rpl.y; A.rpl[3] = rpl.z;
    for (int k = 0; k < 4; ++k) A.jrot[k] = s.joint_rot[i][k];
    const bool free_root = P < 0 && s.n_link_dof[i] == 6;
    const int ns = free_root ? 0 : s.n_slide[i], nr = free_root ? 0 : s.n_link_dof[i] - ns, d = s.dof_start[i] + ns;
    A.word = (uint32_t)(P + 1) | (free_root ? kWaFree : 0u) | ((uint32_t)ns << 6) | ((uint32_t)nr << 8) |
             ((uint32_t)s.dof_start[i] << 11);
    A.k_pos = s.k_pos[i]; A.k_vel = s.k_vel[i]; A.k_limit = s.k_limit[i]; A.k_ang_damp = s.k_ang_damp[i];
    const bool hinge = !free_root && nr >= 1 && d < CARL_BRAX_MAX_DOF;
    A.damping = hinge ? s.dof_damping[d] : 0.0f;
    A.stiffness = hinge ? s.dof_stiffness[d] : 0.0f;
    A.lo = hinge ? s.dof_lo[d] : 0.0f;
    A.hi = hinge ? s.dof_hi[d] : 0.0f;
    A.axis_sign = (s.joint_rot[i][1] == 0.0f && s.joint_rot[i][2] == 0.0f) ? (s.joint_rot[i][3] > 0.0f ? 1.0f : (s.joint_rot[i][3] < 0.0f ? -1.0f : 0.0f)) : 0.0f;
    LinkB& B = pk.b[i];
    const int nch = t.child_begin[i + 1] - t.child_begin[i], nsp = t.coll_begin[i + 1] - t.coll_begin[i];
    mc = nch > mc ? nch : mc;
    const bool iso = s.inv_inertia[i][0] == s.inv_inertia[i][1] && s.inv_inertia[i][1] == s.inv_inertia[i][2];
    B.word = (free_root ? kWbFree : 0u) | (iso ? kWbIso : 0u) | ((uint32_t)nch << 2) | ((uint32_t)t.coll_begin[i] << 6) |
             ((uint32_t)nsp << 12);
    B.children = 0u;
    for (int k = 0; k < 4; ++k)
      B.children |= (uint32_t)(k < nch ? t.child_idx[t.child_begin[i] + k] : s.n_links) << (8 * k);
    B.inv_i0 = s.inv_inertia[i][0];
    float reach = -1.0f;
    for (int kk = t.coll_begin[i]; kk < t.coll_begin[i + 1]; ++kk) {
      const int k = t.coll_idx[kk];
      const v3 c = f3(s.coll_pos[k]) - f3(s.com[i]);
      const float rk = sqrtf(dot(c, c)) + s.coll_radius[k];
      reach = rk > reach ? rk : reach;
      pk.sph[kk].off[0] = c.x; pk.sph[kk].off[1] = c.y; pk.sph[kk].off[2] = c.z;
      pk.sph[kk].radius = s.coll_radius[k];
    }
    B.reach = reach;
  }
  pk.max_children = mc;
  pk.all_iso = 1;
  for (int i = 0; i < s.n_links; ++i)
    if (!(pk.b[i].word & kWbIso)) pk.all_iso = 0;
}

// what the host precomputes per launch (kernel argument, ~2.6 KB)
struct Prepared {
  Topo topo;
  Packed packed;
};

// ---- the fragment schedule of a step / rollout launch (Group::run, MODE 1), as plain integer functions shared by the
// kernel and the host (carl_brax_fragment_plan: the CPU tests check coverage, order and hand-over on it).
// A group = one wavefront's worth of envs.  Workgroup `wg` of `n_wg` owns the groups [g_lo, g_lo + G).
struct WgShare {
  int g_lo, G;
};
__host__ __device__ inline WgShare wg_share(int n_groups, int n_wg, int wg) {
  const int lo = (int)(((long long)wg * n_groups) / n_wg);
  return WgShare{lo, (int)(((long long)(wg + 1) * n_groups) / n_wg) - lo};
}
// With no more groups than wavefronts each wavefront runs one group for all T steps.  With more, the workgroup's
// G x T group-steps, laid out group-major, are cut into one contiguous piece [p0, p1) per wavefront (G >= n_waves
// makes every piece at least T long): whole groups, plus at most the TAIL [s0, T) of group k0 at its start and the
// HEAD [0, s1) of group k1 at its end.
struct Piece {
  int k0, s0, k1, s1, ka, n_whole, n_frag;
};
__host__ __device__ inline Piece make_piece(int G, int T, int n_waves, int wave) {
  long long p0, p1;
  if (G <= n_waves) {
    p0 = (long long)wave * T;
    p1 = wave < G ? p0 + T : p0;
  } else {
    p0 = ((long long)wave * G * T) / n_waves;
    p1 = ((long long)(wave + 1) * G * T) / n_waves;
  }
  Piece p;
  p.k0 = (int)(p0 / T); p.s0 = (int)(p0 % T); p.k1 = (int)(p1 / T); p.s1 = (int)(p1 % T);
  p.ka = p.k0 + (p.s0 > 0 ? 1 : 0);
  p.n_whole = p.k1 > p.ka ? p.k1 - p.ka : 0;
  p.n_frag = (p.s1 > 0 ? 1 : 0) + p.n_whole + (p.s0 > 0 ? 1 : 0);
  return p;
}
// Fragment fi of a piece, in the order the wavefront runs them: the head fragment FIRST (the group is then stored as at
// the end of a launch and the wavefront's flag raised), the whole groups, the tail fragment LAST (after the previous
// wavefront's flag).  `grp` is relative to the workgroup's first group.
struct Fragment {
  int grp, t_lo, t_hi;
  bool wait_head, signal_head;
};
__host__ __device__ inline Fragment fragment_of(const Piece& p, int T, int fi) {
  Fragment f{0, 0, T, false, false};
  if (p.s1 > 0 && fi == 0) {
    f.grp = p.k1; f.t_hi = p.s1; f.signal_head = true;
    return f;
  }
  const int ff = fi - (p.s1 > 0 ? 1 : 0);
  if (ff < p.n_whole) {
    f.grp = p.ka + ff;
  } else {
    f.grp = p.k0; f.t_lo = p.s0; f.wait_head = true;
  }
  return f;
}

template <int kSub>
struct Group {
static constexpr int kEnvs = kLanes / kSub;  // envs per wavefront

struct Lds {
  char* rec;     // this wavefront's body records
  float* base;   // float rows
  Layout lay;
  int env;  // env within the wavefront (0..kEnvs-1)
  int sub;  // lane within the env (0..kSub-1)
  __device__ __forceinline__ float& at(int row) const { return base[row * kEnvs + env]; }
  __device__ __forceinline__ uint32_t& atu(int row) const { return reinterpret_cast<uint32_t*>(base)[row * kEnvs + env]; }
  __device__ __forceinline__ char* body_ptr(int i) const { return rec + (env * lay.L + i) * kBodyBytes; }
  __device__ __forceinline__ double& pd(int i, int c) const { return reinterpret_cast<double*>(body_ptr(i))[c]; }
  __device__ __forceinline__ float& vel(int i, int c) const { return reinterpret_cast<float*>(body_ptr(i) + 56)[c]; }
  // flat element k of the env's pose [7 L] / velocity [6 L] block (the record copies: once per env step)
  __device__ __forceinline__ double& pdk(int k) const { return pd(k / 7, k % 7); }
  __device__ __forceinline__ float& velk(int k) const { return vel(k / 6, k % 6); }
  __device__ __forceinline__ v3d pos(int i) const { return D(pd(i, 0), pd(i, 1), pd(i, 2)); }
  __device__ __forceinline__ qtd rot(int i) const { return qtd{pd(i, 3), pd(i, 4), pd(i, 5), pd(i, 6)}; }
  __device__ __forceinline__ int body_off(int i) const { return (env * lay.L + i) * kBodyBytes; }
  __device__ __forceinline__ int world_off() const { return kEnvs * lay.L * kBodyBytes; }  // the world record (Layout)
  static __device__ __forceinline__ Body body_of(const char* q) {
    typedef double vd2 __attribute__((ext_vector_type(2)));
    const vd2 a = *reinterpret_cast<const vd2*>(q), b2 = *reinterpret_cast<const vd2*>(q + 16),
              c2 = *reinterpret_cast<const vd2*>(q + 32);
    const vf4 d = *reinterpret_cast<const vf4*>(q + 48), e = *reinterpret_cast<const vf4*>(q + 64);
    typedef float vf2 __attribute__((ext_vector_type(2)));
    Body b;
    b.p = D(a.x, a.y, b2.x);
    b.r = qtd{b2.y, c2.x, c2.y, __builtin_bit_cast(double, vf2{d.x, d.y})};
    b.v = V(d.z, d.w, e.x);
    b.w = V(e.y, e.z, e.w);
    return b;
  }
  __device__ __forceinline__ Body body(int i) const { return body_of(rec + body_off(i)); }
  static __device__ __forceinline__ void put_body(char* q, const Body& b) {
    typedef double vd2 __attribute__((ext_vector_type(2)));
    typedef float vf2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<vd2*>(q) = vd2{b.p.x, b.p.y};
    *reinterpret_cast<vd2*>(q + 16) = vd2{b.p.z, b.r.w};
    *reinterpret_cast<vd2*>(q + 32) = vd2{b.r.x, b.r.y};
    const vf2 rz = __builtin_bit_cast(vf2, b.r.z);
    *reinterpret_cast<vf4*>(q + 48) = vf4{rz.x, rz.y, b.v.x, b.v.y};
    *reinterpret_cast<vf4*>(q + 64) = vf4{b.v.z, b.w.x, b.w.y, b.w.z};
  }
  __device__ __forceinline__ void put(int i, const Body& b) const { put_body(rec + body_off(i), b); }
  // The record while a CARL_FLAG_BRAX_FP32 launch runs an env step's substeps: the pose as seven floats in the first two
  // 16-byte pieces, the velocities where the float64 layout has them (pieces 3 and 4) -- converted from / back to the float64
  // layout at the env step's ends (run()), so everything outside the substeps reads what it always reads.
  static __device__ __forceinline__ BodyT<float> body_of(const char* q, float) {
    const vf4 a = *reinterpret_cast<const vf4*>(q), b2 = *reinterpret_cast<const vf4*>(q + 16);
    const vf4 d = *reinterpret_cast<const vf4*>(q + 48), e = *reinterpret_cast<const vf4*>(q + 64);
    BodyT<float> b;
    b.p = V(a.x, a.y, a.z);
    b.r = qt{a.w, b2.x, b2.y, b2.z};
    b.v = V(d.z, d.w, e.x);
    b.w = V(e.y, e.z, e.w);
    return b;
  }
  static __device__ __forceinline__ Body body_of(const char* q, double) { return body_of(q); }
  static __device__ __forceinline__ void put_body(char* q, const BodyT<float>& b) {
    *reinterpret_cast<vf4*>(q) = vf4{b.p.x, b.p.y, b.p.z, b.r.w};
    *reinterpret_cast<vf4*>(q + 16) = vf4{b.r.x, b.r.y, b.r.z, 0.0f};
    *reinterpret_cast<vf4*>(q + 48) = vf4{0.0f, 0.0f, b.v.x, b.v.y};
    *reinterpret_cast<vf4*>(q + 64) = vf4{b.v.z, b.w.x, b.w.y, b.w.z};
  }
  // reaction records (kReactBytes; byte offsets from the float rows)
  __device__ __forceinline__ int react_off(int i) const {
    return (env * (lay.L + 1) + i) * kReactBytes + lay.wrench * kEnvs * 4;  // (the record region starts at row lay.wrench)
  }
  // the env's context record (Layout): just below the float rows
  __device__ __forceinline__ const float* ctx_rec() const {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) - (kEnvs - env) * kCtxRecBytes);
  }
  static __device__ __forceinline__ void put_react(char* q, v3 f, v3 t) {
    *reinterpret_cast<vf4*>(q) = vf4{f.x, f.y, f.z, t.x};
    *reinterpret_cast<vf4*>(q + 16) = vf4{t.y, t.z, 0.0f, 0.0f};
  }
  // planar models: (force x, force z, torque y) in the record's first piece
  static __device__ __forceinline__ void put_react1(char* q, float fx, float fz, float ty) {
    *reinterpret_cast<vf4*>(q) = vf4{fx, fz, ty, 0.0f};
  }
  static __device__ __forceinline__ vf4 get_react1(const char* q) { return *reinterpret_cast<const vf4*>(q); }
  static __device__ __forceinline__ void get_react(const char* q, v3& f, v3& t) {
    const vf4 a = *reinterpret_cast<const vf4*>(q), b2 = *reinterpret_cast<const vf4*>(q + 16);
    f = V(a.x, a.y, a.z);
    t = V(a.w, b2.x, b2.y);
  }
  __device__ __forceinline__ void put3(int row, v3 a) const {
    at(row) = a.x; at(row + 1) = a.y; at(row + 2) = a.z;
  }
  __device__ __forceinline__ v3 get3(int row) const { return V(at(row), at(row + 1), at(row + 2)); }
};

// hand-over between lockstep phases: producer and consumer lanes are in the SAME wavefront, whose LDS instructions
// execute in program order -- so no s_barrier and no s_waitcnt are needed, only that the compiler keeps the order
// (from one thread's point of view the rows written before and read after are different addresses).  A
// wavefront-scope fence pair around a scheduling barrier is exactly that; the other wavefronts of the workgroup are
// not involved.
static __device__ __forceinline__ void phase_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// first column of the rotation matrix of q = q rotating e_x
static __device__ __forceinline__ v3 xaxis(qt q) {
  return V(1.0f - 2.0f * (q.y * q.y + q.z * q.z), 2.0f * (q.x * q.y + q.w * q.z), 2.0f * (q.x * q.z - q.w * q.y));
}

// the static world as a parent body (planar roots are jointed to it)
static __device__ __forceinline__ Body world_body() {
  return Body{D(0, 0, 0), qtd{1, 0, 0, 0}, V(0, 0, 0), V(0, 0, 0)};
}

static __device__ __forceinline__ bool is_free_root(const carl_brax_sys_t& s, int i) {
  return s.parent[i] < 0 && s.n_link_dof[i] == 6;
}

// joint geometry shared by joints.resolve and inverse kinematics
template <class R>
struct JointGeomT {
  v3 rc_off, rp_off;      // anchor relative to the child's / parent's COM, world frame
  typename PoseT<R>::V3 ed;  // A_p - A_c at zero slide, float64 (R = double: the product path)
  v3 vA_c, vA_p, x_c, x_p, wrel;
  v3 axx;                 // cross(x_c, x_p) from the float64 relative rotation
  float theta, thetadot;  // single hinge
  v3 axis[3];             // 2-3 stacked hinges: current world axes ...
  float ang[3], rate[3];  // ... Euler x-y-z angles (third signed by dof_sign3) and their rates
};
using JointGeom = JointGeomT<double>;

// Everything the substep's two phases need to know about a link, as ONE record per link in static LDS, expanded ON THE
// DEVICE once per workgroup (expand_link below; the host-built Packed travels as a kernel argument and stays compact).
// One record = one base address per lane: every field is a ds_read with an immediate offset.  (The phases used to pick
// their constants from a dozen arrays -- LinkA, LinkB, the float64 block, the model table's dof_* / slide_axis /
// inv_inertia / dof_sign3 rows -- and the compiler kept one hoisted address register per array and lane: the substep loop
// ran out of registers and reloaded them from scratch memory.)
//   ac, ap: joint anchor relative to the child's / parent's COM, widened (float64 operands of the anchor separation)
//   G: the relative rotation of the two joint frames is LINEAR in q1 = conj(r_parent) (x) r_child:
//        rel = conj(r_parent (x) rpl) (x) (r_child (x) joint_rot) = conj(rpl) (x) q1 (x) joint_rot = G q1
//      with G = L(conj(rpl)) R(joint_rot), a constant 4 x 4 matrix: one float64 quaternion product and one matrix-vector
//      product per joint and substep instead of three quaternion products (48 -> 32 float64 operations); where every
//      link frame is unrotated (link_rot = identity: every shipped single-hinge model) rpl = joint_rot and G is
//      block-diagonal, rel = (|j|^2 q1.w, M q1.xyz): 26.  Stored compact block first:
//        G00 | G11 G12 G13 | G21 G22 G23 | G31 G32 G33 || G01 G02 G03 | G10 G20 G30
//      built from the float32 table values widened to float64, i.e. the same numbers the float64 restatement multiplies.
//   axc: the hinge axis R(joint_rot) e_x in the child's frame (x_c = r_child rotating axc: float32 is enough for a direction)
//   dof[k]: damping, stiffness, lower, upper bound of the joint's k-th dof IN JOINT ORDER (slides first, then hinges)
struct alignas(16) LinkRec {
  double ac[3], ap[3];                        //   0
  double G[16];                               //  48
  float axc[3], k_pos;                        // 176
  float rpl[4];                               // 192: parent-side joint frame in the parent frame: link_rot (x) joint_rot
  float k_vel, k_limit, k_ang_damp, sign3;    // 208
  float dof[5][4];                            // 224
  float slide_axis[2][4];                     // 304: prismatic axes in the parent frame
  float inv_i[3], reach;                      // 336: body phase: inverse principal moments (link frame); sphere reach (-1: none)
  float axis_sign, ext[3];                    // 352: planar models: +-1, the joint frame's x axis is +-y of the link frame;
                                              //      ext: max over the link's spheres of |centre - COM| per link-frame axis, each
                                              //      + the largest radius (reach = the ball that holds every sphere; ext the box)
  float acf[4], apf[4], Gf[16];               // 368: ac, ap, G rounded to float32 (CARL_FLAG_BRAX_FP32 launches read these)
};
static_assert(sizeof(LinkRec) == kLinkRecBytes, "LinkRec is read in 16-byte pieces at fixed offsets");

// MULTI: the general kernels -- links with 0, 2 or 3 hinges (Humanoid), prismatic dofs, or rotated link frames; false
// compiles the Euler-angle path, the slides and the general G out (Ant: a free root and single hinges).
// Everything that is a difference of the two poses is formed in float64 and rounded ONCE: the anchor
// separation `ed`, the relative rotation of the joint frames, the axis-alignment term and the joint angles.
struct JointRec {
  qt rpl;
  uint32_t word;
  float sign3;
};
template <class R>
struct JointXrT {  // the float64 block of a LinkRec in registers (R = float: its float32 copy)
  typename PoseT<R>::V3 ac, ap;
  R G[16];
  v3 axc;
};
using JointXr = JointXrT<double>;
template <bool MULTI>
static __device__ __forceinline__ JointXrT<float> load_jointx_f32(const LinkRec& X) {
  JointXrT<float> r;
  const vf4 a = *reinterpret_cast<const vf4*>(&X.acf[0]), b = *reinterpret_cast<const vf4*>(&X.apf[0]);
  r.ac = V(a.x, a.y, a.z);
  r.ap = V(b.x, b.y, b.z);
  constexpr int kQuads = MULTI ? 4 : 3;  // (the lean kernels read G[0..9])
#pragma unroll
  for (int k = 0; k < kQuads; ++k) {
    const vf4 g = *reinterpret_cast<const vf4*>(&X.Gf[4 * k]);
    r.G[4 * k] = g.x; r.G[4 * k + 1] = g.y; r.G[4 * k + 2] = g.z; r.G[4 * k + 3] = g.w;
  }
  const vf4 ax = *reinterpret_cast<const vf4*>(&X.axc[0]);
  r.axc = V(ax.x, ax.y, ax.z);
  return r;
}
template <bool MULTI>
static __device__ __forceinline__ JointXr load_jointx(const LinkRec& X) {
  typedef double vd2 __attribute__((ext_vector_type(2)));
  const vd2* p = reinterpret_cast<const vd2*>(&X);
  JointXr r;
  const vd2 a0 = p[0], a1 = p[1], a2 = p[2];
  r.ac = D(a0.x, a0.y, a1.x);
  r.ap = D(a1.y, a2.x, a2.y);
  constexpr int kPairs = MULTI ? 8 : 5;
#pragma unroll
  for (int k = 0; k < kPairs; ++k) {
    const vd2 g = p[3 + k];
    r.G[2 * k] = g.x;
    r.G[2 * k + 1] = g.y;
  }
  const vf4 ax = *reinterpret_cast<const vf4*>(&X.axc[0]);
  r.axc = V(ax.x, ax.y, ax.z);
  return r;
}
template <bool MULTI>
static __device__ __forceinline__ JointXrT<float> load_jointx(const LinkRec& X, float) { return load_jointx_f32<MULTI>(X); }
template <bool MULTI>
static __device__ __forceinline__ JointXr load_jointx(const LinkRec& X, double) { return load_jointx<MULTI>(X); }
template <bool MULTI, class R>
static __device__ __forceinline__ JointGeomT<R> joint_geometry(const JointRec& la, const JointXrT<R>& X, const BodyT<R>& bc,
                                                               const BodyT<R>& bp) {
  using P3 = typename PoseT<R>::V3;
  using PQ = typename PoseT<R>::Q;
  JointGeomT<R> g;
  {
    const P3 rc_off = qrot(bc.r, X.ac), rp_off = qrot(bp.r, X.ap);
    g.ed = (bp.p - bc.p) + (rp_off - rc_off);  // A_p - A_c (at zero slide)
    g.rc_off = tof(rc_off);
    g.rp_off = tof(rp_off);
  }
  g.vA_c = bc.v + cross(bc.w, g.rc_off);
  g.vA_p = bp.v + cross(bp.w, g.rp_off);
  // relative rotation of the joint frames rc = bc.r (x) joint_rot, rp = bp.r (x) rpl: rel = G (conj(bp.r) (x) bc.r).
  // cross(x_c, x_p) = rp (x) cross(xaxis(rel), e_x) = rp (x) (0, a2, -a1): the SMALL components of xaxis(rel)
  // keep their relative accuracy.  The directions themselves (hinge axes, the frame that carries (0, a2, -a1) to the
  // world) are float32 work on the rounded rotations.
  PQ rel;
  const qt rp = qmul(tof(bp.r), la.rpl);
  {
    const PQ q1 = qmul(qconj(bp.r), bc.r);
    const R* G = X.G;
    if constexpr (MULTI) {
      rel.w = G[0] * q1.w + G[10] * q1.x + G[11] * q1.y + G[12] * q1.z;
      rel.x = G[13] * q1.w + G[1] * q1.x + G[2] * q1.y + G[3] * q1.z;
      rel.y = G[14] * q1.w + G[4] * q1.x + G[5] * q1.y + G[6] * q1.z;
      rel.z = G[15] * q1.w + G[7] * q1.x + G[8] * q1.y + G[9] * q1.z;
    } else {
      rel.w = G[0] * q1.w;
      rel.x = G[1] * q1.x + G[2] * q1.y + G[3] * q1.z;
      rel.y = G[4] * q1.x + G[5] * q1.y + G[6] * q1.z;
      rel.z = G[7] * q1.x + G[8] * q1.y + G[9] * q1.z;
    }
    const R a1 = (R)2.0 * (rel.x * rel.y + rel.w * rel.z), a2 = (R)2.0 * (rel.x * rel.z - rel.w * rel.y);
    g.x_c = qrot(tof(bc.r), X.axc);
    g.x_p = xaxis(rp);
    g.axx = qrot_yz(rp, (float)a2, (float)-a1);
  }
  if (rel.w < (R)0.0) { rel.w = -rel.w; rel.x = -rel.x; }
  const int nr = MULTI ? wa_hinges(la.word) : 1;
  // The first float64 arctangent serves both kinds of joint: the twist about the hinge (theta / 2, joint frame x) on a
  // single-hinge lane, the first Euler angle on a stacked-hinge lane.  A wavefront of a multi-hinge model holds both
  // kinds, so two separate calls under complementary lane masks cost it two evaluations (~30 float64 instructions each).
  [[maybe_unused]] R R00 = (R)0.0, R01 = (R)0.0, R02 = (R)0.0;
  R a_y = rel.x, a_x = rel.w;
  // STRAIGHT-LINE for the multi-hinge kernels: the Euler-angle path is evaluated on every lane and the single-hinge
  // lanes just do not use it.  A wavefront of such a model always holds both kinds of joint, so `if (nr != 1)` never
  // skipped anything -- it only cost the exec-mask bookkeeping and the register copies at the joins (31 saveexec, 30
  // branches, ~90 moves in the substep loop's ISA).
  const bool eul = MULTI && nr != 1;
  if (MULTI) {  // rel = Rx(al) Ry(be) Rz(ga): decompose, ga = sign * theta_3 (nr = 0: all locked)
    R00 = (R)1.0 - (R)2.0 * (rel.y * rel.y + rel.z * rel.z);
    R01 = (R)2.0 * (rel.x * rel.y - rel.w * rel.z);
    R02 = fmin(fmax((R)2.0 * (rel.x * rel.z + rel.w * rel.y), (R)-1.0), (R)1.0);
    const R m12 = -((R)2.0 * (rel.y * rel.z - rel.w * rel.x));  // -R12
    const R r22 = (R)1.0 - (R)2.0 * (rel.x * rel.x + rel.y * rel.y);  // R22
    a_y = eul ? m12 : a_y;
    a_x = eul ? r22 : a_x;
  }
  const R a1 = atan2_r<!MULTI>(a_y, a_x);  // (a single hinge: a_x = rel.w >= 0 after the flip)
  g.theta = (float)((R)2.0 * a1);  // (meaningful on single-hinge lanes)
  g.wrel = bc.w - bp.w;
  g.thetadot = dot(g.x_c, g.wrel);
  if (MULTI) {
    const float al = (float)a1, be = (float)asin_r(R02), ga = (float)atan2_r(-R01, R00);
    const float sg = (nr == 3) ? la.sign3 : 1.0f;
    g.ang[0] = al; g.ang[1] = be; g.ang[2] = sg * ga;
    g.axis[0] = g.x_p;
    // The second and third hinge axes, rp (x) Rx(al) e_y and rp (x) Rx(al) Ry(be) e_z = rp (x) (0, cos al, sin al) and
    // rp (x) (sin be, -sin al cos be, cos al cos be): the sines and cosines are entries of the relative rotation's
    // third column (R02, R12, R22) = (sin be, -sin al cos be, cos al cos be) -- no sincos of the angles just extracted,
    // no quaternion products (two sincos + two products + two rotations were ~155 instructions per joint: Humanoid
    // 3.51 -> 3.41 ms per 20-step launch, A/B on one box).
    // cos be = |(R12, R22)|; at the gimbal pole (cos be -> 0, outside every shipped joint range) al is arbitrary: 0.
    {
      const float sb = (float)R02, sacb = (float)a_y, cacb = (float)a_x;  // a_y = -R12, a_x = R22
      const float h2 = sacb * sacb + cacb * cacb;
      const bool pole = !(h2 > 1e-20f);
      const float ih = pole ? 0.0f : __builtin_amdgcn_rsqf(h2);
      const float ca = pole ? 1.0f : cacb * ih, sa = sacb * ih;
      g.axis[1] = qrot_yz(rp, ca, sa);
      g.axis[2] = qrot(rp, V(sb, -sacb, cacb)) * sg;
    }
    const float w0 = dot(g.wrel, g.axis[0]), w1 = dot(g.wrel, g.axis[1]), w2 = dot(g.wrel, g.axis[2]);
    g.rate[1] = w1;
    {  // three hinges: axis0 and axis2 are not orthogonal (axis0 . axis2 = sign * sin(be)); else the plain projections
       // (a locked direction is damped by k_ang_damp).  One v_rcp_f32 for the two quotients (1 ulp; an IEEE division is
       // ~10 instructions each).
      const float cc = dot(g.axis[0], g.axis[2]), iden = __builtin_amdgcn_rcpf(1.0f - cc * cc);
      const float r0 = (w0 - cc * w2) * iden, r2 = (w2 - cc * w0) * iden;
      g.rate[0] = (nr == 3) ? r0 : w0;
      g.rate[2] = (nr == 3) ? r2 : w2;
    }
  }
  return g;
}

// The same geometry for the OBSERVATION (kinematics.world_to_joint / inverse), in float32.  The dynamics need the joint
// angles and anchor separations to ~1e-10 -- they are multiplied by constraint stiffnesses -- and take them from
// joint_geometry; an observation entry is compared at 1e-5 (1 + |x|), and float32 quaternion algebra with atan2_fast
// delivers the angles to ~1e-6: the observation pass costs a third of the float64 one (one pass per env step: 4-9 % of
// a launch was observe).  Positions enter as the float64 difference of the two COMs, rounded once.
struct ObsGeom {
  v3 ed, vA_c, vA_p;
  float theta, thetadot;
  float ang[3], rate[3];
};
template <bool MULTI>
static __device__ __forceinline__ ObsGeom obs_geometry(const LinkA& rec, const float sign3, const Body& bc, const Body& bp) {
  ObsGeom g;
  const vf4 q0 = ld4(&rec.ac[0]), q1 = ld4(&rec.ap[0]), q2 = ld4(&rec.rpl[0]), q3 = ld4(&rec.jrot[0]);
  const uint32_t word = __float_as_uint(q0.w);
  const qt rc = tof(bc.r), rp = tof(bp.r);
  const v3 rc_off = qrot(rc, V(q0.x, q0.y, q0.z)), rp_off = qrot(rp, V(q1.x, q1.y, q1.z));
  g.ed = tof(bp.p - bc.p) + (rp_off - rc_off);
  g.vA_c = bc.v + cross(bc.w, rc_off);
  g.vA_p = bp.v + cross(bp.w, rp_off);
  const qt rcj = qmul(rc, qt{q3.x, q3.y, q3.z, q3.w}), rpj = qmul(rp, qt{q2.x, q2.y, q2.z, q2.w});
  qt rel = qmul(qconj(rpj), rcj);
  if (rel.w < 0.0f) { rel.w = -rel.w; rel.x = -rel.x; }
  const v3 x_c = xaxis(rcj), wrel = bc.w - bp.w;
  g.thetadot = dot(x_c, wrel);
  const int nr = MULTI ? wa_hinges(word) : 1;
  float a_y = rel.x, a_x = rel.w;
  [[maybe_unused]] float R00 = 0.0f, R01 = 0.0f, R02 = 0.0f;
  const bool eul = MULTI && nr != 1;
  if (MULTI) {
    R00 = 1.0f - 2.0f * (rel.y * rel.y + rel.z * rel.z);
    R01 = 2.0f * (rel.x * rel.y - rel.w * rel.z);
    R02 = fminf(fmaxf(2.0f * (rel.x * rel.z + rel.w * rel.y), -1.0f), 1.0f);
    const float m12 = -(2.0f * (rel.y * rel.z - rel.w * rel.x)), r22 = 1.0f - 2.0f * (rel.x * rel.x + rel.y * rel.y);
    a_y = eul ? m12 : a_y;
    a_x = eul ? r22 : a_x;
  }
  const float a1 = atan2_fast(a_y, a_x);
  g.theta = 2.0f * a1;
  if (MULTI) {
    const float h2 = a_y * a_y + a_x * a_x;  // (Euler lanes: cos^2 be)
    const bool pole = !(h2 > 1e-20f);
    const float ih = pole ? 0.0f : __builtin_amdgcn_rsqf(h2);
    const float be = atan2_fast(R02, h2 * ih), ga = atan2_fast(-R01, R00);
    const float sg = (nr == 3) ? sign3 : 1.0f;
    g.ang[0] = a1; g.ang[1] = be; g.ang[2] = sg * ga;
    const float ca = pole ? 1.0f : a_x * ih, sa = a_y * ih;
    const v3 ax0 = xaxis(rpj), ax1 = qrot_yz(rpj, ca, sa), ax2 = qrot(rpj, V(R02, -a_y, a_x)) * sg;
    const float w0 = dot(wrel, ax0), w1 = dot(wrel, ax1), w2 = dot(wrel, ax2);
    g.rate[1] = w1;
    const float cc = dot(ax0, ax2), iden = __builtin_amdgcn_rcpf(1.0f - cc * cc);
    g.rate[0] = (nr == 3) ? (w0 - cc * w2) * iden : w0;
    g.rate[2] = (nr == 3) ? (w2 - cc * w0) * iden : w2;
  }
  return g;
}

// link-pair contact of the push task (carl_brax_sys_t::n_pair): run by the object's joint lane.  Returns
// the push on the object and the reaction on the gripper link (force, torque about that link's COM).
// The object hangs on the world, so its parent-side wrench rows are free: the reaction is parked there,
// and Topo lists the object among the gripper's children, so phase B sums it in.  Everything goes in
// and out BY VALUE: reference parameters of a non-inlined call pin their variables to scratch memory in
// the caller for every model, not only for this one.  The penetration depth (times pair_k) is a pose
// difference: float64.
struct PairOut {
  v3 on_obj, on_a, t_a;
  uint32_t fired;  // bit 16 + k: pair contact k pushed (a discrete decision: part of the object's contact hash)
};
static __device__ __forceinline__ PairOut pair_contact(const carl_brax_sys_t& s, const Lds& m, const float friction) {
  const int a = s.pair_link;
  const Body ba = m.body(a), bo = m.body(s.push_link);
  const v3d o = bo.p - qrot(bo.r, tod(f3(s.com[s.push_link])));
  PairOut r{V(0, 0, 0), V(0, 0, 0), V(0, 0, 0), 0u};
  for (int k = 0; k < s.n_pair; ++k) {
    const v3d reld = qrot(ba.r, tod(f3(s.pair_pos[k]) - f3(s.com[a])));
    const v3d cs = ba.p + reld;
    const v3 rel = tof(reld);
    const float rk = s.pair_radius[k];
    if (!((float)fabs(cs.z - o.z) < s.pair_obj_half + rk)) continue;
    const double dxd = o.x - cs.x, dyd = o.y - cs.y;
    const double distd = sqrt(dxd * dxd + dyd * dyd);
    const float depth = (float)((double)rk + (double)s.pair_obj_radius - distd);
    const float dist = (float)distd;
    if (!(depth > 0.0f) || !(dist > 1e-9f)) continue;
    const v3 n = V((float)dxd / dist, (float)dyd / dist, 0.0f);
    const v3 vs = ba.v + cross(ba.w, rel);
    const v3 vr = vs - bo.v;
    const float closing = dot(vr, n);
    const float fm = s.pair_k * depth + s.pair_c * closing;
    if (!(fm > 0.0f)) continue;
    r.fired |= 1u << (16 + (k & 15));
    v3 fc = n * fm;
    if (s.pair_ct > 0.0f) {  // Coulomb friction, regularised (carl_amd.h: pair_ct): min(pair_ct |vt|, friction fm) along vt
      v3 vt = vr - n * closing;
      vt.z = 0.0f;  // (the object's free directions are horizontal: the table carries the vertical part)
      const float vt_len = sqrtf(dot(vt, vt));
      if (vt_len > 1e-9f) fc = fc + vt * (fminf(s.pair_ct * vt_len, friction * fm) / vt_len);
    }
    r.on_obj = r.on_obj + fc;
    r.on_a = r.on_a - fc;
    r.t_a = r.t_a - cross(rel, fc);
  }
  return r;
}

// 1 / sqrt(x), x near 1 (a quaternion's squared norm after one integration step): v_rsq_f64 + two Newton steps
static __device__ __forceinline__ float rsqrt_r(float x) {  // float32 pose (CARL_FLAG_BRAX_FP32): v_rsq_f32 + one Newton step
  const float y = __builtin_amdgcn_rsqf(x);
  return y * fmaf(-0.5f * x, y * y, 1.5f);
}
static __device__ __forceinline__ double rsqrt_f64(double x) {
  // seed: v_rsq_f32 of the rounded argument (1 ulp of float32, 6e-8); one Newton step squares the error: 5e-15, under the
  // 48 bits the pose record keeps.  (v_rsq_f64 + two steps: twice the dependent float64 chain at the end of the body
  // phase, the longest latency chain of the substep.)
  double y = (double)__builtin_amdgcn_rsqf((float)x);
  y = y * fma(-0.5 * x, y * y, 1.5);
  return y;
}

static __device__ __forceinline__ double rsqrt_r(double x) { return rsqrt_f64(x); }

// ---- one brax.spring.pipeline.step ---------------------------------------------------------
// Branch record: every DISCRETE decision of the substep that the float64 restatement also takes is hashed per
// link (h <- 33 h + bits): the contacts that delivered an impulse (bit = the sphere's ordinal on its link) and the
// range limits that were active on the link's joint (slides: bits 0-3, hinges: bits 4-9; below / above per dof).
// The step's combination of the per-link hashes is an optional output
// (carl_step_io_t::branch_sig): a parity check can then separate lanes that took the same branches as the
// reference arithmetic from lanes where a contact switched within rounding -- an impulse is discontinuous there.
// Launch-invariant scalars of the substep.  The model table sits in LDS and every phase hand-over is a fence, so the
// compiler may not carry a value it read from the table across a phase: left in the substep, `dt`, exp(damping dt),
// 1 / dt ... were re-read and re-derived in every one of the n_frames substeps.  Wave-uniform ones are pinned in
// scalar registers (readfirstlane), which also takes them out of the vector-register budget.
struct SubK {
  float dt, dl, inv_dt, erp, plane_z;
  int L, first_joint, max_children;
  bool all_iso;  // every link's effective inertia is isotropic (spring_inertia_scale = 1: every shipped model)
};
static __device__ __forceinline__ float uniform(float x) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
static __device__ __forceinline__ SubK make_subk(const carl_brax_sys_t& s, const Topo& tp, const Packed& pk) {
  SubK k;
  k.dt = uniform(s.dt);
  k.dl = uniform(__expf(s.vel_damping * s.dt));
  k.inv_dt = uniform(__builtin_amdgcn_rcpf(s.dt));
  k.erp = uniform(s.baumgarte_erp);
  k.plane_z = uniform(s.plane_z);
  k.L = __builtin_amdgcn_readfirstlane(s.n_links);
  k.first_joint = __builtin_amdgcn_readfirstlane(tp.first_joint);
  k.max_children = __builtin_amdgcn_readfirstlane(pk.max_children);
  k.all_iso = __builtin_amdgcn_readfirstlane(pk.all_iso) != 0;
  return k;
}
static __device__ __forceinline__ vf4 ld4(const void* p) { return *reinterpret_cast<const vf4*>(p); }

// ONE LANE PER LINK: lane `sub` of an env owns link sub -- its joint (the one to its parent) in the joint phase and
// its body in the body phase (kSub >= n_links; the host never launches a narrower group).  What a lane needs to find
// its data is fixed for the launch and kept in registers AS ADDRESSES: a phase starts with every address known, issues
// all its loads at once, and every access is base register + immediate offset.
struct LaneLink {
  int i;                  // the lane's link (idle lanes: clamped to L - 1, `body` false)
  bool body;              // the lane owns a body (sub < L)
  bool joint;             // ... and that link hangs on a joint (everything but a free root)
  uint32_t wa, wb;        // the index words of LinkA / LinkB
  uint32_t wch;           // LinkB::children: the first four children's links, a byte each (none: L, the env's zero record)
  const LinkRec* lr;      // the link's constants
  const SphRec* sph;      // the link's first sphere (widened records)
  char* own;              // the own body record
  const char* par;        // the parent's (the world record)
  char* react;            // the own joint's reaction record
  const char* react_env;  // the env's first reaction record
  const float* ctx;       // the env's context record
  const float* tau;       // the tau row of the joint's first dof (stride kEnvs floats per dof)
  __device__ __forceinline__ const char* child(int k) const {  // the k-th child's reaction record
    return react_env + ((wch >> (8 * k)) & 255u) * (uint32_t)kReactBytes;
  }
};

// What a lane carries in registers through the n_frames substeps of an env step: its body (the LDS record is rewritten
// at the end of every body phase -- the children read it in the next joint phase -- but the owner never reads it back),
// the joint's hinge torques, the branch hashes and 1 / mass.
template <bool MULTI, class R = double>
struct StepRegs {
  BodyT<R> b;
  float tau[MULTI ? 3 : 1];
  uint32_t sig_hit, sig_lim;
  float inv_m;
};

static __device__ __forceinline__ LaneLink make_lane_link(const Packed& pk, const LinkRec* lrec, const SphRec* sph_rec, const SubK& K,
                                                          const Lds& m) {
  LaneLink ll;
  const bool body = m.sub < K.L;
  const int i = body ? m.sub : K.L - 1;
  ll.i = i;
  ll.body = body;
  ll.wa = body ? pk.a[i].word : kWaFree;
  ll.wb = body ? pk.b[i].word : 0u;
  ll.wch = body ? pk.b[i].children : (uint32_t)K.L * 0x01010101u;
  ll.joint = body && (ll.wa & kWaFree) == 0u;
  const int P = wa_parent(ll.wa);
  ll.lr = lrec + i;
  ll.sph = sph_rec + wb_first_sphere(ll.wb);
  ll.own = m.rec + m.body_off(i);
  ll.par = m.rec + (P < 0 ? m.world_off() : m.body_off(P));
  ll.react = reinterpret_cast<char*>(m.base) + m.react_off(i);
  ll.react_env = reinterpret_cast<const char*>(m.base) + m.react_off(0);
  ll.ctx = m.ctx_rec();
  ll.tau = &m.at(m.lay.tau + wa_dof(ll.wa));
  return ll;
}

// LinkRec of link i from the model table and the host-built records (one lane per link, once per workgroup)
static __device__ __forceinline__ void expand_link(const carl_brax_sys_t& s, const Packed& pk, int i, LinkRec& out) {
  const LinkA& A = pk.a[i];
  // The float64 block is formed in FLOAT64 from the model table's float32 values -- the numbers the float64 restatement
  // combines (oracle/brax_spring.c: joint_geometry).  Rounds 1-5 widened the host's float32 results instead
  // (build_packed_host: ac = a - com, ap = link_pos + link_rot a - com_parent, rpl = link_rot (x) joint_rot, each rounded
  // to float32): a constant 6e-8 in the parent-side joint frame of every link whose frame is rotated against its
  // parent's (Humanoid's thighs, shins and arms) = 2e-7 of axis misalignment x k_pos = 1e-5 rad/s of angular velocity PER
  // SUBSTEP on those links (tools/diag_humanoid_reset_step.py), and 1.5e-8 m in the anchors x k_pos = 1e-6 m/s: the largest
  // single term of the Humanoid's parity error since round 2 -- found in round 6 when the stiffer humanoid.xml constants
  // pushed its worst entry from 7e-6 to 1.0e-5.
  const qtd lrot = tod(f4(s.link_rot[i])), j = tod(f4(s.joint_rot[i]));
  {
    const int P = wa_parent(A.word);
    const v3d a = tod(f3(s.joint_pos[i]));
    const v3d com_p = (P < 0) ? D(0, 0, 0) : tod(f3(s.com[P]));
    const v3d ac = a - tod(f3(s.com[i]));
    const v3d ap = (tod(f3(s.link_pos[i])) + qrot(lrot, a)) - com_p;
    out.ac[0] = ac.x; out.ac[1] = ac.y; out.ac[2] = ac.z;
    out.ap[0] = ap.x; out.ap[1] = ap.y; out.ap[2] = ap.z;
  }
  const qtd cr = qconj(qmul(lrot, j));
  double G[4][4];
#pragma unroll
  for (int cidx = 0; cidx < 4; ++cidx) {  // column cidx of G: conj(rpl) (x) e_cidx (x) joint_rot
    const qtd e{cidx == 0 ? 1.0 : 0.0, cidx == 1 ? 1.0 : 0.0, cidx == 2 ? 1.0 : 0.0, cidx == 3 ? 1.0 : 0.0};
    const qtd col = qmul(qmul(cr, e), j);
    G[0][cidx] = col.w; G[1][cidx] = col.x; G[2][cidx] = col.y; G[3][cidx] = col.z;
  }
  out.G[0] = G[0][0];
#pragma unroll
  for (int r = 1; r < 4; ++r) {
#pragma unroll
    for (int cidx = 1; cidx < 4; ++cidx) out.G[1 + 3 * (r - 1) + (cidx - 1)] = G[r][cidx];
    out.G[9 + r] = G[0][r];
    out.G[12 + r] = G[r][0];
  }
  const v3 ax = xaxis(f4(A.jrot));
  out.axc[0] = ax.x; out.axc[1] = ax.y; out.axc[2] = ax.z;
  out.k_pos = A.k_pos;
#pragma unroll
  for (int k = 0; k < 4; ++k) out.rpl[k] = A.rpl[k];
  out.k_vel = A.k_vel; out.k_limit = A.k_limit; out.k_ang_damp = A.k_ang_damp;
  out.sign3 = s.dof_sign3[i];
  const bool free_root = (A.word & kWaFree) != 0u;
  const int d0 = wa_dof(A.word), nd = free_root ? 0 : wa_slides(A.word) + wa_hinges(A.word);
  for (int k = 0; k < 5; ++k) {
    const int d = min(d0 + k, CARL_BRAX_MAX_DOF - 1);
    const bool on = k < nd;
    out.dof[k][0] = on ? s.dof_damping[d] : 0.0f;
    out.dof[k][1] = on ? s.dof_stiffness[d] : 0.0f;
    out.dof[k][2] = on ? s.dof_lo[d] : 0.0f;
    out.dof[k][3] = on ? s.dof_hi[d] : 0.0f;
  }
  for (int k = 0; k < 2; ++k) {
    out.slide_axis[k][0] = s.slide_axis[i][k][0]; out.slide_axis[k][1] = s.slide_axis[i][k][1];
    out.slide_axis[k][2] = s.slide_axis[i][k][2]; out.slide_axis[k][3] = 0.0f;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    out.acf[k] = (float)out.ac[k];
    out.apf[k] = (float)out.ap[k];
  }
  out.acf[3] = out.apf[3] = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) out.Gf[k] = (float)out.G[k];
  out.inv_i[0] = s.inv_inertia[i][0]; out.inv_i[1] = s.inv_inertia[i][1]; out.inv_i[2] = s.inv_inertia[i][2];
  out.reach = pk.b[i].reach;
  out.axis_sign = A.axis_sign;
  {
    float ex = 0.0f, ey = 0.0f, ez = 0.0f, rmax = 0.0f;
    const uint32_t wb = pk.b[i].word;
    for (int k = 0; k < wb_spheres(wb); ++k) {
      const Sphere& sp = pk.sph[wb_first_sphere(wb) + k];
      ex = fmaxf(ex, fabsf(sp.off[0])); ey = fmaxf(ey, fabsf(sp.off[1])); ez = fmaxf(ez, fabsf(sp.off[2]));
      rmax = fmaxf(rmax, sp.radius);
    }
    // the largest radius rides on EVERY axis: a sphere's lowest point is at most |R row 2| . (ex, ey, ez) + rmax below the
    // COM, and |R row 2| . (rmax, rmax, rmax) >= rmax because the row has unit length
    out.ext[0] = ex + rmax; out.ext[1] = ey + rmax; out.ext[2] = ez + rmax;
  }
}

// R diag(inv_i) R^T t; every shipped model has isotropic effective inertia (spring_inertia_scale = 1): inv_i[0] t
static __device__ __forceinline__ v3 apply_inv_inertia(const LinkRec* lr, qt r, v3 t, bool iso, float inv_i0) {
  if (iso) return t * inv_i0;
  const v3 l = qrot(qconj(r), t);
  return qrot(r, V(l.x * lr->inv_i[0], l.y * lr->inv_i[1], l.z * lr->inv_i[2]));
}

template <bool MULTI, bool TASK, class Real = double>
static __device__ __forceinline__ void substep(const carl_brax_sys_t& s, const Topo& tp, const SubK& K, const LaneLink& ll,
                                        const Lds& m, StepRegs<MULTI, Real>& R, Prof& prof) {
  using P3 = typename PoseT<Real>::V3;
  using PQ = typename PoseT<Real>::Q;
  BodyT<Real>& b = R.b;
  const LinkRec* const lr = ll.lr;
  // phase A -- spring.joints.resolve, the lane's own joint.  The wrench on the child stays in registers (the same lane
  // applies it in the body phase); only the reaction on the parent goes through LDS.
  v3 f = V(0, 0, 0), tc = V(0, 0, 0);
  [[maybe_unused]] uint32_t pair_fired = 0u;  // (task models: the object's lane carries it to its contact hash)
  if (ll.joint) {
    const uint32_t wa = ll.wa;
    const int ns = MULTI ? wa_slides(wa) : 0;
    const int nr = MULTI ? wa_hinges(wa) : 1;
    // every load of the phase, in one batch
    const JointXrT<Real> X = load_jointx<MULTI>(*lr, Real{});
    const vf4 q1 = ld4(&lr->axc[0]);  // .w k_pos
    const vf4 q2 = ld4(&lr->rpl[0]);
    const BodyT<Real> bp = Lds::body_of(ll.par, Real{});
    const vf4 q4 = ld4(&lr->k_vel);   // k_vel k_limit k_ang_damp sign3
    const JointRec la{qt{q2.x, q2.y, q2.z, q2.w}, wa, q4.w};
    const JointGeomT<Real> g = joint_geometry<MULTI, Real>(la, X, b, bp);
    const float k_limit = q4.y;
    const float kp = q1.w * ll.ctx[4];
    P3 ed = g.ed;
    v3 ev = g.vA_p - g.vA_c;
    uint32_t lim = 0u;
    if constexpr (MULTI) {
      for (int k = 0; k < ns; ++k) {  // prismatic dofs: free along the axis, own spring/damper/force
        const vf4 sa = ld4(&lr->slide_axis[k][0]), dk = ld4(&lr->dof[k][0]);  // dk: damping stiffness lo hi
        const P3 axd = qrot(bp.r, PoseT<Real>::from(V(sa.x, sa.y, sa.z)));
        const Real qkd = -dot(ed, axd);
        ed = ed + axd * qkd;  // a planar root's slide coordinate is its travelled distance: float64 projection
        const v3 ax = tof(axd);
        const float qk = (float)qkd, qdk = -dot(ev, ax);
        ev = ev + ax * qdk;
        float fa = ll.tau[k * kEnvs] - dk.x * qdk - dk.y * qk;
        if (qk < dk.z) { fa += k_limit * (dk.z - qk); lim |= 1u << (2 * k); }  // range of the slide
        if (qk > dk.w) { fa -= k_limit * (qk - dk.w); lim |= 2u << (2 * k); }
        f = f + ax * fa;
      }
    }
    f = f + tof(ed) * kp + ev * q4.x;
    v3 t;
    {  // single hinge: keep the hinge axes aligned + the hinge torque about the child-side axis
      uint32_t lim1 = 0u;
      const vf4 h0 = ld4(&lr->dof[MULTI ? min(ns, 2) : 0][0]);  // the first hinge: damping stiffness lo hi
      float ta = R.tau[0] - h0.x * g.thetadot - h0.y * g.theta;
      if (g.theta < h0.z) { ta += k_limit * (h0.z - g.theta); lim1 |= 16u; }
      if (g.theta > h0.w) { ta -= k_limit * (g.theta - h0.w); lim1 |= 32u; }
      t = g.axx * kp + g.x_c * ta;
      if constexpr (MULTI) {  // 2 or 3 stacked hinges (or none): per-dof torques about the current axes; a missing dof is
                              // locked by the constraint spring on its Euler angle.  Straight-line like joint_geometry:
                              // evaluated on every lane, selected by the lane's number of hinges.
        v3 t2 = V(0, 0, 0);
        uint32_t lim2 = 0u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const bool act = k < nr;
          const vf4 hk = ld4(&lr->dof[min(ns, 2) + k][0]);
          const float lo = hk.z, hi = hk.w;
          float tk = R.tau[k] - hk.x * g.rate[k] - hk.y * g.ang[k];
          const bool below = g.ang[k] < lo, above = g.ang[k] > hi;
          tk = below ? tk + k_limit * (lo - g.ang[k]) : tk;
          tk = above ? tk - k_limit * (g.ang[k] - hi) : tk;
          lim2 |= (act && below) ? (16u << (2 * k)) : 0u;
          lim2 |= (act && above) ? (32u << (2 * k)) : 0u;
          t2 = t2 + g.axis[k] * (act ? tk : -kp * g.ang[k]);
        }
        const bool single = nr == 1;
        t = V(single ? t.x : t2.x, single ? t.y : t2.y, single ? t.z : t2.z);
        lim1 = single ? lim1 : lim2;
      }
      lim |= lim1;
    }
    t = t - g.wrel * q4.z;
    R.sig_lim = R.sig_lim * 33u + lim;
    v3 pf = f * -1.0f, pt = (cross(g.rp_off, f) + t) * -1.0f;  // on the parent
    if (TASK && s.n_pair > 0 && ll.i == s.push_link) {
      const PairOut po = pair_contact(s, m, ll.ctx[1]);
      pair_fired = po.fired;
      f = f + po.on_obj;
      pf = po.on_a;
      pt = po.t_a;
    }
    tc = cross(g.rc_off, f) + t;
    Lds::put_react(ll.react, pf, pt);
  }
  phase_sync();
  prof.mark(kProfJoints);
  // phase B -- per body: wrench sum, semi-implicit Euler, its contacts, integrate
  if (ll.body) {
    const vf4 cx = ld4(ll.ctx);  // gravity_z, friction, elasticity, exp(ang_damping dt)
    const float dl = K.dl, da = cx.w, inv_dt = K.inv_dt, dt = K.dt;
    const uint32_t wb = ll.wb;
    const vf4 qb = ld4(&lr->inv_i[0]);  // .x inv_inertia[0], .w reach
    const float inv_i0 = qb.x;
    const bool iso = (wb & kWbIso) != 0u;
    // Anisotropic effective inertia only exists in the GENERAL kernels (TASK: the host sends every model with a link whose
    // principal moments differ there, carl_brax.hip: brax_is_general); everywhere else isotropy is a compile-time fact and
    // the rotated-inertia code is not in the instruction stream at all.
    const bool all_iso = !TASK || K.all_iso;
    const qt rf = tof(b.r);
    v3 F = f, T = tc;  // own joint's wrench (a free root has none)
    {  // children's reactions, ascending: a wavefront-uniform trip count and loads whose addresses come from registers
       // (a link with fewer children reads the env's zero record), so the pass is one batch of independent LDS reads
      if (K.max_children > 2) {  // (Ant's torso, Humanoid's torso: all four slots in ONE batch of loads)
        v3 fk[4], tk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) Lds::get_react(ll.child(k), fk[k], tk[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { F = F + fk[k]; T = T + tk[k]; }
      } else if (K.max_children > 0) {
        v3 fk[2], tk[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) Lds::get_react(ll.child(k), fk[k], tk[k]);
#pragma unroll
        for (int k = 0; k < 2; ++k) { F = F + fk[k]; T = T + tk[k]; }
      }
      if (K.max_children > 4)  // (no shipped model; wavefront-uniform)
        for (int cc = tp.child_begin[ll.i] + 4; cc < tp.child_begin[ll.i + 1]; ++cc) {
          v3 fk, tk;
          Lds::get_react(ll.react_env + tp.child_idx[cc] * kReactBytes, fk, tk);
          F = F + fk;
          T = T + tk;
        }
    }
    const float inv_m = R.inv_m;
    b.v = b.v + (F * inv_m + V(0, 0, cx.x)) * dt;
    b.w = b.w + (all_iso ? T * inv_i0 : apply_inv_inertia(lr, rf, T, iso, inv_i0)) * dt;
    if (TASK && s.obj_support != 0 && ll.i == s.push_link) {
      // the push task's object on the table (carl_amd.h: obj_support): Coulomb friction under the normal load m |g| as an
      // impulse -- the horizontal velocity shrinks by friction |g| dt, at most to zero
      const float load = cx.x < 0.0f ? -cx.x : 0.0f;
      const float vh = sqrtf(b.v.x * b.v.x + b.v.y * b.v.y);
      if (vh > 1e-9f) {
        const float cut = fminf(cx.y * load * dt, vh) / vh;
        b.v.x -= b.v.x * cut;
        b.v.y -= b.v.y * cut;
      }
    }
    prof.mark(kProfBodySum);
    // spring.collisions.resolve: this body's spheres vs the plane z = 0
    v3 cdv = V(0, 0, 0), cdw = V(0, 0, 0);
    float cnt = 0.0f;
    uint32_t hit = 0u;
    // (ONE definition of the eight accumulators: as plain constants the compiler re-materialised their zeros on every path
    //  into and around the sphere loop -- three sets of eight v_mov per substep)
    asm volatile("" : "+v"(cdv.x), "+v"(cdv.y), "+v"(cdv.z), "+v"(cdw.x), "+v"(cdw.y), "+v"(cdw.z), "+v"(cnt), "+v"(hit));
    // no sphere of this link can reach the plane while its COM is higher than the farthest sphere surface
    const Real pz = b.p.z - (Real)K.plane_z;  // height above the collision plane (the ground; the push task's table)
    const int n_sph = ((float)pz < qb.w) ? wb_spheres(wb) : 0;
    if (ballot(n_sph > 0) != 0ull) {  // (nothing at all while every link the wavefront holds is out of reach)
      // third row of the rotation matrix in float64: a sphere's height -- hence its depth, which the Baumgarte
      // term multiplies by erp / dt -- is a pose difference
      const Real R20 = (Real)2.0 * (b.r.x * b.r.z - b.r.w * b.r.y), R21 = (Real)2.0 * (b.r.y * b.r.z + b.r.w * b.r.x),
                 R22 = (Real)1.0 - (Real)2.0 * (b.r.x * b.r.x + b.r.y * b.r.y);
      // a sphere's operands in the substep's pose type: the widened record's doubles (float32 launches: its float part)
      struct SphOp { Real x, y, z, r; };
      auto ld_sph = [&](const SphRec* q) {
        if constexpr (std::is_same_v<Real, double>) {
          typedef double vd2 __attribute__((ext_vector_type(2)));
          const vd2 a = *reinterpret_cast<const vd2*>(q), c = *(reinterpret_cast<const vd2*>(q) + 1);
          return SphOp{a.x, a.y, c.x, c.y};
        } else {
          const vf4 a = ld4(&q->f[0]);
          return SphOp{a.x, a.y, a.z, a.w};
        }
      };
      auto depth_of = [&](const SphOp sp) { return (float)(sp.r - (pz + (R20 * sp.x + R21 * sp.y + R22 * sp.z))); };
      // the impulse of sphere j (ordinal on its link) at penetration `depth`.  ISO (wavefront-uniform: every link of the
      // model has isotropic effective inertia c) folds R diag(c) R^T = c into the formulas:
      //   n . (I^-1 (r x n) x r) = c (r.x^2 + r.y^2),   dir . (I^-1 (r x dir) x r) = c |r x dir|^2
      auto respond = [&](const vf4 sp, const float depth, const int j, auto iso_tag) {
        constexpr bool ISO = decltype(iso_tag)::value;
        // (the plane normal n = e_z is folded in by hand: `dot(n, x)`, `cross(r, n)`, `n * imp` spelled with n as a
        // vector leave the multiplications by its zeros in the instruction stream -- 0 * x is not 0 for IEEE)
        const float radius = sp.w;
        const v3 ro = qrot(rf, V(sp.x, sp.y, sp.z));
        const v3 r = V(ro.x, ro.y, ro.z - radius);
        const v3 rel = b.v + cross(b.w, r);
        const float vn = rel.z;  // n . rel
        float ang;
        if constexpr (ISO) {
          ang = inv_i0 * (r.x * r.x + r.y * r.y);
        } else {
          const v3 in = apply_inv_inertia(lr, rf, V(r.y, -r.x, 0.0f), iso, inv_i0);  // I^-1 (r x n)
          ang = in.x * r.y - in.y * r.x;                                              // n . (I^-1 (r x n) x r)
        }
        const float imp = div_fast(-(1.0f + cx.z) * vn + K.erp * depth * inv_dt, inv_m + ang);
        if (!(imp > 0.0f) || !(vn < 0.0f)) return;
        hit |= 1u << j;
        float Jx = 0.0f, Jy = 0.0f;  // J = n imp - dir imp_d, dir = the tangential velocity's direction (dir.z = 0)
        const float vt_len = sqrtf(rel.x * rel.x + rel.y * rel.y);
        if (vt_len > 1e-9f) {
          const float il = __builtin_amdgcn_rcpf(vt_len), dx = rel.x * il, dy = rel.y * il;
          const v3 rxd = V(-r.z * dy, r.z * dx, r.x * dy - r.y * dx);  // r x dir
          float ang_d;
          if constexpr (ISO) {
            ang_d = inv_i0 * dot(rxd, rxd);
          } else {
            const v3 id = apply_inv_inertia(lr, rf, rxd, iso, inv_i0);  // I^-1 (r x dir)
            const v3 c2 = cross(id, r);
            ang_d = dx * c2.x + dy * c2.y;
          }
          const float imp_d = fminf(div_fast(vt_len, inv_m + ang_d), cx.y * imp);
          Jx = -dx * imp_d;
          Jy = -dy * imp_d;
        }
        const v3 J = V(Jx, Jy, imp);
        cdv = cdv + J * inv_m;
        if constexpr (ISO) {
          cdw = cdw + cross(r, J) * inv_i0;
        } else {
          cdw = cdw + apply_inv_inertia(lr, rf, cross(r, J), iso, inv_i0);
        }
        cnt += 1.0f;
      };
      // The same impulse for isotropic models (every shipped one) as STRAIGHT-LINE code: every lane of the loop evaluates
      // it and the discrete decisions (penetrating, delivering an impulse, sliding) become selects on the impulse's
      // components -- a lane that does not fire adds exact zeros.  The nested per-lane branches of `respond` cost the
      // wavefront more than the arithmetic they skip: some foot of some env of a wavefront is on the ground in nearly
      // every substep, so the blocks ran anyway, with ~10 exec-mask / branch instructions and ~24 register copies at the
      // joins of the eight accumulators around ~60 of arithmetic (ISA of round 6).  The tangential speed's square root is
      // v_sqrt_f32 (1 ulp; sqrtf's correctly rounded finish is 14 instructions).  One wavefront-uniform branch remains:
      // nothing is evaluated while no sphere of the wavefront penetrates.
      auto respond_iso = [&](const vf4 sp, const float depth, const int j) {
        const float radius = sp.w;
        const v3 ro = qrot(rf, V(sp.x, sp.y, sp.z));
        const v3 r = V(ro.x, ro.y, ro.z - radius);
        const v3 rel = b.v + cross(b.w, r);
        const float vn = rel.z;
        const float ang = inv_i0 * (r.x * r.x + r.y * r.y);
        const float imp = div_fast(-(1.0f + cx.z) * vn + K.erp * depth * inv_dt, inv_m + ang);
        const bool fire = depth > 0.0f && imp > 0.0f && vn < 0.0f;
        const float vt_len = __builtin_amdgcn_sqrtf(rel.x * rel.x + rel.y * rel.y);
        const bool slip = fire && vt_len > 1e-9f;
        // (a lane that does not slide gets the zero direction: its tangential impulse is then an exact zero without a select
        //  per component; every other factor below is finite)
        const float il = slip ? __builtin_amdgcn_rcpf(vt_len) : 0.0f, dx = rel.x * il, dy = rel.y * il;
        const v3 rxd = V(-r.z * dy, r.z * dx, r.x * dy - r.y * dx);  // r x dir
        const float imp_d = fminf(div_fast(vt_len, inv_m + inv_i0 * dot(rxd, rxd)), cx.y * imp);
        const v3 J = V(-dx * imp_d, -dy * imp_d, fire ? imp : 0.0f);
        hit |= (uint32_t)fire << j;  // (the count of the impulses is the population count of these bits: below the loop)
        cdv = cdv + J * inv_m;
        // r x J accumulated as is, scaled by 1 / I once below the loop (the impulse runs ~2.5 times per Ant substep: every
        // instruction in it is 0.4 % of the launch)
        cdw.x = fmaf(r.y, J.z, cdw.x); cdw.x = fmaf(-r.z, J.y, cdw.x);
        cdw.y = fmaf(r.z, J.x, cdw.y); cdw.y = fmaf(-r.x, J.z, cdw.y);
        cdw.z = fmaf(r.x, J.y, cdw.z); cdw.z = fmaf(-r.y, J.x, cdw.z);
      };
      // (no software prefetch of the next record: the kernel is bound by instruction issue, not by this load's latency -- the
      //  other wavefronts of the SIMD fill the wait -- and rotating a prefetched record costs four 64-bit moves per sphere)
      if (all_iso) {  // (wavefront-uniform; two loops: one loop with both bodies copied the accumulators at its joins)
        // the record's doubles -- no conversions -- and the next record loaded into the registers the depth has just released:
        // no copies (Ant -1.7 % against the float32 record with a rotating prefetch; Humanoid unchanged within its +-0.4 %
        // run-to-run spread)
        SphOp sp = ld_sph(&ll.sph[0]);
        for (int j = 0; j < n_sph; ++j) {
          const float depth = depth_of(sp);
          __builtin_amdgcn_sched_barrier(0);
          sp = ld_sph(&ll.sph[j + 1]);
          if (ballot(depth > 0.0f) != 0ull) respond_iso(ld4(&ll.sph[j].f[0]), depth, j);
        }
        cnt = (float)__popc(hit);
        cdw = cdw * inv_i0;
      } else {
        for (int j = 0; j < n_sph; ++j) {
          const float depth = depth_of(ld_sph(&ll.sph[j]));
          if (depth > 0.0f) respond(ld4(&ll.sph[j].f[0]), depth, j, std::false_type{});
        }
      }
    }
    prof.mark(kProfContacts);
    if constexpr (TASK) hit |= pair_fired;
    R.sig_hit = R.sig_hit * 33u + hit;
    // spring.integrator.integrate
    b.v = b.v * dl;
    b.w = b.w * da;
    if (cnt > 0.0f) {
      const float ic = __builtin_amdgcn_rcpf(cnt);
      b.v = b.v + cdv * ic;
      b.w = b.w + cdw * ic;
    }
    const Real dtd = (Real)dt;
    b.p = PoseT<Real>::mk(fma_r((Real)b.v.x, dtd, b.p.x), fma_r((Real)b.v.y, dtd, b.p.y), fma_r((Real)b.v.z, dtd, b.p.z));
    const PQ dq = qmul_vec(PoseT<Real>::mk((Real)b.w.x, (Real)b.w.y, (Real)b.w.z), b.r);
    const Real h = (Real)0.5 * dtd;
    PQ r2 = PQ{fma_r(h, dq.w, b.r.w), fma_r(h, dq.x, b.r.x), fma_r(h, dq.y, b.r.y), fma_r(h, dq.z, b.r.z)};
    const Real inv = rsqrt_r(r2.w * r2.w + r2.x * r2.x + r2.y * r2.y + r2.z * r2.z);
    b.r = PQ{r2.w * inv, r2.x * inv, r2.y * inv, r2.z * inv};
    Lds::put_body(ll.own, b);
  }
  phase_sync();
  prof.mark(kProfBodies);
}

// ---- the same substep for PLANAR models (Halfcheetah, Hopper, Walker2d: a root on two world slides x, z and a hinge
// about y, every other link on a hinge about y, all geometry in the y = 0 plane; carl_brax.hip: brax_is_planar).
// Their state never leaves the plane: rotations are (w, 0, y, 0) -- the half-angle complex number u = w + i y --,
// angular velocities (0, omega, 0), nothing moves in y.  The general substep still pays for the full 3-D algebra:
// three float64 quaternion products and two float64 vector rotations per joint, 3-vector cross products everywhere.
// Here the SAME records, the SAME two phases and the SAME formulas are evaluated with the zero components left
// out: rotating (ax, ., az) by u is (c ax + s az, -s ax + c az) with c = w^2 - y^2, s = 2 w y; the relative joint
// rotation is conj(u_p) u_c (with link_rot = identity and joint_rot = "x -> y", the joint-frame products cancel:
// rel = (W, +-Y, 0, 0), the sign that of the hinge axis); a torque is its y component; the first-order quaternion update is w += -h omega y,
// y += h omega w.  Pose differences stay float64.  Results agree with the general substep to rounding
// (tests/test_gpu_brax.py: both paths against each other and against the float64 restatement of the 3-D pipeline).
template <bool TASK, class Real = double>
static __device__ __forceinline__ void substep_planar(const carl_brax_sys_t& s, const Topo& tp, const SubK& K, const LaneLink& ll,
                                               const Lds& m, StepRegs<false, Real>& R, Prof& prof) {
  static_assert(!TASK, "planar models are not task models");
  BodyT<Real>& b = R.b;
  const LinkRec* const lr = ll.lr;
  // phase A -- spring.joints.resolve (every link of a planar model has a joint: the root hangs on the world)
  float fx = 0.0f, fz = 0.0f, tcy = 0.0f;  // the wrench on the child: stays in registers
  if (ll.joint) {
    const uint32_t wa = ll.wa;
    const int ns = wa_slides(wa);
    Real acx, acz, apx, apz;
    if constexpr (std::is_same_v<Real, double>) {
      typedef double vd2 __attribute__((ext_vector_type(2)));
      const vd2* xp = reinterpret_cast<const vd2*>(lr);
      const vd2 x0 = xp[0], x1 = xp[1], x2 = xp[2];  // ac.x ac.y | ac.z ap.x | ap.y ap.z
      acx = x0.x; acz = x1.x; apx = x1.y; apz = x2.y;
    } else {
      const vf4 a4 = ld4(&lr->acf[0]), p4 = ld4(&lr->apf[0]);
      acx = a4.x; acz = a4.z; apx = p4.x; apz = p4.z;
    }
    const float k_pos = lr->k_pos;
    const BodyT<Real> bp = Lds::body_of(ll.par, Real{});
    const vf4 q4 = ld4(&lr->k_vel);                  // k_vel k_limit k_ang_damp .
    const vf4 q5 = ld4(&lr->dof[ns > 0 ? 2 : 0][0]);  // the hinge: damping stiffness lo hi
    const float sg = lr->axis_sign;                  // +-1: the hinge axis is +-y (joint_rot maps x to +-y)
    const float k_limit = q4.y, kp = k_pos * ll.ctx[4];
    // rotation of both bodies as (cos, sin) of the full angle, float64
    const Real cc = b.r.w * b.r.w - b.r.y * b.r.y, sc = (Real)2.0 * (b.r.w * b.r.y);
    const Real cp = bp.r.w * bp.r.w - bp.r.y * bp.r.y, sp = (Real)2.0 * (bp.r.w * bp.r.y);
    const Real rcx = cc * acx + sc * acz, rcz = cc * acz - sc * acx;  // anchor - COM, child side, world
    const Real rpx = cp * apx + sp * apz, rpz = cp * apz - sp * apx;  // ... parent side
    Real edx = (bp.p.x - b.p.x) + (rpx - rcx), edz = (bp.p.z - b.p.z) + (rpz - rcz);  // A_p - A_c
    const float rcxf = (float)rcx, rczf = (float)rcz, rpxf = (float)rpx, rpzf = (float)rpz;
    // anchor velocities v + omega x r, omega = (0, w.y, 0)
    const float vcx = b.v.x + b.w.y * rczf, vcz = b.v.z - b.w.y * rcxf;
    const float vpx = bp.v.x + bp.w.y * rpzf, vpz = bp.v.z - bp.w.y * rpxf;
    float evx = vpx - vcx, evz = vpz - vcz;
    // relative rotation conj(u_p) u_c: the hinge angle is twice its argument
    Real Wr = bp.r.w * b.r.w + bp.r.y * b.r.y, Yr = bp.r.w * b.r.y - bp.r.y * b.r.w;
    if (Wr < (Real)0.0) { Wr = -Wr; Yr = -Yr; }
    const float theta = sg * (float)((Real)2.0 * atan2_r<true>(Yr, Wr));
    const float wrel = b.w.y - bp.w.y, thetadot = sg * wrel;
    uint32_t lim = 0u;
    if (ns > 0) {  // the root: slides along world x and z (the parent is the world), unlimited
      const vf4 sx = ld4(&lr->dof[0][0]), sz = ld4(&lr->dof[1][0]);  // damping stiffness lo hi
      const float qx = (float)(-edx), qz = (float)(-edz), qdx = -evx, qdz = -evz;
      edx = (Real)0.0; edz = (Real)0.0; evx = 0.0f; evz = 0.0f;
      fx = ll.tau[0] - sx.x * qdx - sx.y * qx;
      fz = ll.tau[kEnvs] - sz.x * qdz - sz.y * qz;
      if (qx < sx.z) { fx += k_limit * (sx.z - qx); lim |= 1u; }
      if (qx > sx.w) { fx -= k_limit * (qx - sx.w); lim |= 2u; }
      if (qz < sz.z) { fz += k_limit * (sz.z - qz); lim |= 4u; }
      if (qz > sz.w) { fz -= k_limit * (qz - sz.w); lim |= 8u; }
    }
    fx = fx + (float)edx * kp + evx * q4.x;
    fz = fz + (float)edz * kp + evz * q4.x;
    float ta = R.tau[0] - q5.x * thetadot - q5.y * theta;
    if (theta < q5.z) { ta += k_limit * (q5.z - theta); lim |= 16u; }
    if (theta > q5.w) { ta -= k_limit * (theta - q5.w); lim |= 32u; }
    const float ty = sg * ta - wrel * q4.z;  // about y: hinge torque, angular damping (the axes are parallel: no alignment term)
    R.sig_lim = R.sig_lim * 33u + lim;
    // (a x f).y = a.z f.x - a.x f.z
    tcy = (rczf * fx - rcxf * fz) + ty;
    Lds::put_react1(ll.react, -fx, -fz, -((rpzf * fx - rpxf * fz) + ty));
  }
  phase_sync();
  prof.mark(kProfJoints);
  // phase B -- per body: wrench sum, semi-implicit Euler, its contacts, integrate
  if (ll.body) {
    const vf4 cx = ld4(ll.ctx);  // gravity_z, friction, elasticity, exp(ang_damping dt)
    const float dl = K.dl, da = cx.w, inv_dt = K.inv_dt, dt = K.dt;
    const uint32_t wb = ll.wb;
    const vf4 qb = ld4(&lr->inv_i[0]);  // .x inv_inertia[0], .w reach
    float Fx = fx, Fz = fz, Ty = tcy;
    {
      if (K.max_children > 2) {
        vf4 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = Lds::get_react1(ll.child(k));
#pragma unroll
        for (int k = 0; k < 4; ++k) { Fx += r[k].x; Fz += r[k].y; Ty += r[k].z; }
      } else if (K.max_children > 0) {
        const vf4 r0 = Lds::get_react1(ll.child(0)), r1 = Lds::get_react1(ll.child(1));
        Fx = (Fx + r0.x) + r1.x; Fz = (Fz + r0.y) + r1.y; Ty = (Ty + r0.z) + r1.z;
      }
      if (K.max_children > 4)  // (no shipped model; wavefront-uniform)
        for (int cc2 = tp.child_begin[ll.i] + 4; cc2 < tp.child_begin[ll.i + 1]; ++cc2) {
          const vf4 r = Lds::get_react1(ll.react_env + tp.child_idx[cc2] * kReactBytes);
          Fx += r.x; Fz += r.y; Ty += r.z;
        }
    }
    const float inv_m = R.inv_m, inv_i = qb.x;
    float vx = b.v.x + (Fx * inv_m) * dt, vz = b.v.z + (Fz * inv_m + cx.x) * dt;
    float om = b.w.y + (Ty * inv_i) * dt;
    prof.mark(kProfBodySum);
    // spring.collisions.resolve: this body's spheres vs the plane z = 0
    float cdvx = 0.0f, cdvz = 0.0f, cdw = 0.0f, cnt = 0.0f;
    uint32_t hit = 0u;
    asm volatile("" : "+v"(cdvx), "+v"(cdvz), "+v"(cdw), "+v"(cnt), "+v"(hit));  // (one definition: substep's note)
    const Real pz = b.p.z - (Real)K.plane_z;
    // Which links can touch the plane at all: the COM lower than the ball that holds every sphere (reach), and lower
    // than the link's box at its present pitch -- |sin| ext.x + |cos| ext.z: a level torso (Halfcheetah's: four
    // spheres along a 1.2 m rod, always inside the ball) stays out of the loop, whose trip count is the wavefront's
    // largest sphere count.  Float32 with a 1e-4 guard; the depths themselves stay float64.
    const vf4 qe = ld4(&lr->axis_sign);  // . ext.x ext.y ext.z
    const float cthf = 1.0f - 2.0f * ((float)b.r.y * (float)b.r.y), sthf = 2.0f * ((float)b.r.w * (float)b.r.y);
    const bool near = (float)pz < qb.w && (float)pz < fabsf(sthf) * qe.y + fabsf(cthf) * qe.w + 1e-4f;
    const int n_sph = near ? wb_spheres(wb) : 0;
    if (ballot(n_sph > 0) != 0ull) {
      const Real cth = (Real)1.0 - (Real)2.0 * (b.r.y * b.r.y), sth = (Real)2.0 * (b.r.w * b.r.y);  // R22, -R20 of the general form
      const float cf = (float)cth, sf = (float)sth;
      auto depth_of = [&](const vf4 sp) {
        return (float)((Real)sp.w - (pz + (cth * (Real)sp.z - sth * (Real)sp.x)));
      };
      // straight-line like the general substep's respond_iso: the decisions are selects on the impulse's components, the one
      // branch left is wavefront-uniform (no sphere of the wavefront penetrates)
      auto respond = [&](const vf4 sp, const float depth, const int j) {
        const float radius = sp.w;
        const float rx = cf * sp.x + sf * sp.z, rz = (cf * sp.z - sf * sp.x) - radius;  // sphere's lowest point - COM
        const float relx = vx + om * rz, relz = vz - om * rx;
        const float vn = relz;
        const float imp = div_fast(-(1.0f + cx.z) * vn + K.erp * depth * inv_dt, inv_m + inv_i * (rx * rx));
        const bool fire = depth > 0.0f && imp > 0.0f && vn < 0.0f;
        const float vt_len = fabsf(relx);
        const bool slip = fire && vt_len > 1e-9f;
        const float dx = relx > 0.0f ? 1.0f : -1.0f;
        const float imp_d = fminf(div_fast(vt_len, inv_m + inv_i * (rz * rz)), cx.y * imp);
        const float Jx = slip ? -dx * imp_d : 0.0f, Jz = fire ? imp : 0.0f;
        hit |= fire ? (1u << j) : 0u;
        cnt += fire ? 1.0f : 0.0f;
        cdvx += Jx * inv_m;
        cdvz += Jz * inv_m;
        cdw += inv_i * (rz * Jx - rx * Jz);  // (r x J).y
      };
      // (the float32 part of the widened record, the next one in flight while this one is evaluated: this short substep is
      //  not purely issue-bound -- without the prefetch Halfcheetah lost 2 %, measured; the general substep gained 1 %)
      vf4 nxt = ld4(&ll.sph[0].f[0]);
      for (int j = 0; j < n_sph; ++j) {
        const vf4 sp = nxt;
        nxt = ld4(&ll.sph[j + 1].f[0]);
        const float depth = depth_of(sp);
        if (ballot(depth > 0.0f) != 0ull) respond(sp, depth, j);
      }
    }
    prof.mark(kProfContacts);
    R.sig_hit = R.sig_hit * 33u + hit;
    vx *= dl; vz *= dl; om *= da;
    if (cnt > 0.0f) {
      const float ic = __builtin_amdgcn_rcpf(cnt);
      vx += cdvx * ic; vz += cdvz * ic; om += cdw * ic;
    }
    const Real dtd = (Real)dt, h = (Real)0.5 * dtd, omd = (Real)om;
    b.p.x = fma_r((Real)vx, dtd, b.p.x);
    b.p.z = fma_r((Real)vz, dtd, b.p.z);
    const Real w2 = fma_r(h, -(omd * b.r.y), b.r.w), y2 = fma_r(h, omd * b.r.w, b.r.y);  // r + h (0, omega) (x) r
    const Real inv = rsqrt_r(w2 * w2 + y2 * y2);
    b.r.w = w2 * inv; b.r.y = y2 * inv;
    b.v.x = vx; b.v.z = vz; b.w.y = om;
    Lds::put_body(ll.own, b);
  }
  phase_sync();
  prof.mark(kProfBodies);
}

// whole-body centre of mass (brax.envs.humanoid.Humanoid._com), float64 (the forward reward is the difference
// of two of these over one env step); *mass_sum = total mass
static __device__ __forceinline__ v3d system_com(const carl_brax_sys_t& s, const Lds& m, float* mass_sum) {
  v3d com = D(0, 0, 0);
  float M = 0.0f;
  for (int i = 0; i < s.n_links; ++i) {
    const float mi = m.at(m.lay.mass + i);
    com = com + m.pos(i) * (double)mi;
    M += mi;
  }
  *mass_sum = M;
  return com * (1.0 / (double)M);
}

// kinematics.world_to_joint + inverse -> observation rows (q[skip:] ++ qd) in the io staging, one
// link per lane; obs_extended (humanoid) appends com inertia (L x 10), com velocity (L x 6) and
// qfrc_actuator (the tau rows; `zero_frc`: reset observations see a zero action).  `go`: envs
// that take part (the calls are wavefront-uniform).  Ends with a phase_sync.
// SLIDES: the model may have prismatic dofs.  False for the step kernels of the lean non-planar models (the host sends a
// model with slides to the multi-hinge or the planar kernels, carl_brax.hip: brax_is_multi): the slide rows and the anchor
// separation / anchor velocities they alone read are then not in the instruction stream.
template <bool MULTI, bool TASK, bool SLIDES = true>
// `com_in` / `mass_in`: the whole-body centre of mass and total mass when the caller has just formed them at this very
// state (the step's forward reward of reward_on_com models: one 11-link pass less per Humanoid env step).
static __device__ __forceinline__ void observe(const carl_brax_sys_t& s, const Packed& pk, const Lds& m, bool go,
                                        bool zero_frc, const v3d* com_in = nullptr, float mass_in = 0.0f) {
  const int skip = s.exclude_current_positions;
  // q[from:] as sin ++ cos (inverted double pendulum): the raw angles are written to the sin rows and
  // converted in a second pass below; everything behind them moves back by n_trig rows
  const int n_trig = s.obs_trig_from > 0 ? s.n_q - s.obs_trig_from : 0;
  const int qd0 = s.n_q - skip + n_trig;  // first qd row in the observation
  const int L = s.n_links;
  const float clipv = s.obs_qd_clip > 0.0f ? s.obs_qd_clip : 3.0e38f;  // hopper / walker2d clip velocities
  const bool keep_raw = s.tip_link > 0 && !(TASK && (s.target_link > 0 || s.push_link > 0));  // the tip reward uses unclipped rates: parked in the wrench rows
  auto put_qd = [&](int dof, float v) {
    m.at(m.lay.io + qd0 + dof) = fminf(fmaxf(v, -clipv), clipv);
    if (keep_raw) m.at(m.lay.wrench + dof) = v;
  };
  float M = 1.0f;
  v3d com = D(0, 0, 0);
  const bool all_iso = __builtin_amdgcn_readfirstlane(pk.all_iso) != 0;
  if (MULTI && s.obs_extended) {
    if (com_in != nullptr) {
      com = *com_in;
      M = mass_in;
    } else if (go) {
      com = system_com(s, m, &M);
    }
  }
  for (int i = m.sub; i < L; i += kSub) {
    if (!go) continue;
    const int P = s.parent[i];
    const Body b = m.body(i);
    if (is_free_root(s, i)) {
      const v3d cd = qrot(b.r, tod(f3(s.com[i])));
      const v3 c = tof(cd), o = tof(b.p - cd);
      const v3 vl = b.v - cross(b.w, c);
      const float qv[7] = {o.x, o.y, o.z, (float)b.r.w, (float)b.r.x, (float)b.r.y, (float)b.r.z};
#pragma unroll
      for (int k = 0; k < 7; ++k)
        if (s.q_start[i] + k >= skip) m.at(m.lay.io + s.q_start[i] + k - skip) = qv[k];
      const float dvv[6] = {vl.x, vl.y, vl.z, b.w.x, b.w.y, b.w.z};
#pragma unroll
      for (int k = 0; k < 6; ++k) put_qd(s.dof_start[i] + k, dvv[k]);
    } else {
      const Body bp = (P < 0) ? world_body() : m.body(P);
      const ObsGeom g = obs_geometry<MULTI>(pk.a[i], s.dof_sign3[i], b, bp);
      const int ns = SLIDES ? s.n_slide[i] : 0;
      for (int k = 0; k < ns; ++k) {
        const v3 ax = qrot(tof(bp.r), f3(s.slide_axis[i][k]));
        if (s.q_start[i] + k >= skip) m.at(m.lay.io + s.q_start[i] + k - skip) = -dot(g.ed, ax);
        put_qd(s.dof_start[i] + k, dot(g.vA_c - g.vA_p, ax));
      }
      const int nr = MULTI ? s.n_link_dof[i] - ns : 1;
      if (!MULTI || nr == 1) {
        if (s.q_start[i] + ns >= skip) m.at(m.lay.io + s.q_start[i] + ns - skip) = g.theta;
        put_qd(s.dof_start[i] + ns, g.thetadot);
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < nr) {
            if (s.q_start[i] + ns + k >= skip) m.at(m.lay.io + s.q_start[i] + ns + k - skip) = g.ang[k];
            put_qd(s.dof_start[i] + ns + k, g.rate[k]);
          }
      }
    }
    if (MULTI && s.obs_extended) {  // inertia about the system com, world axes, row-major, then mass; com velocity
      const int e0 = m.lay.io + qd0 + s.n_dof;
      const v3 d = tof(b.p - com);
      const qt rf = tof(b.r);
      const float mi = m.at(m.lay.mass + i), dd = dot(d, d);
      // (v_rcp_f32, 1 ulp: an observation entry; an IEEE division is ~10 instructions)
      const float I0 = __builtin_amdgcn_rcpf(s.inv_inertia[i][0]), I1 = __builtin_amdgcn_rcpf(s.inv_inertia[i][1]),
                  I2 = __builtin_amdgcn_rcpf(s.inv_inertia[i][2]);
      const float dc[3] = {d.x, d.y, d.z};
      int k = e0 + 10 * i;
      if (all_iso) {  // R diag(c, c, c) R^T = c: no rotation needed (every shipped model; wavefront-uniform)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) m.at(k++) = (r == cc ? I0 : 0.0f) + mi * ((r == cc ? dd : 0.0f) - dc[r] * dc[cc]);
      } else {
        const v3 ex = qrot(rf, V(1, 0, 0)), ey = qrot(rf, V(0, 1, 0)), ez = qrot(rf, V(0, 0, 1));
        const float e[3][3] = {{ex.x, ey.x, ez.x}, {ex.y, ey.y, ez.y}, {ex.z, ey.z, ez.z}};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            float v = e[r][0] * I0 * e[cc][0];
            v += e[r][1] * I1 * e[cc][1];
            v += e[r][2] * I2 * e[cc][2];
            v += mi * ((r == cc ? dd : 0.0f) - dc[r] * dc[cc]);
            m.at(k++) = v;
          }
      }
      m.at(k) = mi;
      k = e0 + 10 * L + 6 * i;
      const float f = mi * __builtin_amdgcn_rcpf(M);
      m.at(k) = f * b.v.x; m.at(k + 1) = f * b.v.y; m.at(k + 2) = f * b.v.z;
      m.at(k + 3) = b.w.x; m.at(k + 4) = b.w.y; m.at(k + 5) = b.w.z;
    }
  }
  if (MULTI && s.obs_extended && go) {
    const int k0 = m.lay.io + qd0 + s.n_dof + 16 * L;
    for (int d = m.sub; d < s.n_dof; d += kSub) m.at(k0 + d) = zero_frc ? 0.0f : m.at(m.lay.tau + d);
  }
  phase_sync();
  if (n_trig > 0) {  // wavefront-uniform
    if (go)
      for (int i = s.obs_trig_from + m.sub; i < s.n_q; i += kSub) {
        float sn, cs;
        sincos_fast(m.at(m.lay.io + i - skip), sn, cs);
        m.at(m.lay.io + i - skip) = sn;
        m.at(m.lay.io + i - skip + n_trig) = cs;
      }
    phase_sync();
  }
  if (TASK && s.push_link > 0) {  // push task: arm q ++ arm qd ++ COM of the end effector, the object, the goal
    const int na = s.q_start[s.push_link], nq = s.n_q;
    if (go)
      for (int i = m.sub; i < na; i += kSub) m.at(m.lay.wrench + i) = m.at(m.lay.io + nq + i);
    phase_sync();
    if (go) {
      for (int i = m.sub; i < na; i += kSub) m.at(m.lay.io + na + i) = m.at(m.lay.wrench + i);
      for (int j = m.sub; j < 3; j += kSub) {
        const int k = m.lay.io + 2 * na + j;
        m.at(k) = (float)m.pd(s.tip_link, j);
        m.at(k + 3) = (float)m.pd(s.push_link, j);
        m.at(k + 6) = m.at(m.lay.goal + j);
      }
    }
    phase_sync();
  }
  if (TASK && s.target_link > 0) {  // reach task: the rows q ++ qd written above are parked in the wrench rows (free
                            // here) and laid out again as cos ++ sin ++ goal q ++ arm qd ++ (tip - goal)
    const int tq = s.q_start[s.target_link], td = s.dof_start[s.target_link], nq = s.n_q;
    if (go)
      for (int i = m.sub; i < nq + td; i += kSub) m.at(m.lay.wrench + i) = m.at(m.lay.io + i);
    phase_sync();
    if (go) {
      for (int i = m.sub; i < tq; i += kSub) {
        float sn, cs;
        sincos_fast(m.at(m.lay.wrench + i), sn, cs);
        m.at(m.lay.io + i) = cs;
        m.at(m.lay.io + tq + i) = sn;
      }
      for (int i = tq + m.sub; i < nq; i += kSub) m.at(m.lay.io + tq + i) = m.at(m.lay.wrench + i);
      for (int i = m.sub; i < td; i += kSub) m.at(m.lay.io + tq + nq + i) = m.at(m.lay.wrench + nq + i);
      if (m.sub == 0) {
        const Body bt = m.body(s.tip_link), bg = m.body(s.target_link);
        const v3d tip = bt.p + qrot(bt.r, tod(f3(s.tip_offset) - f3(s.com[s.tip_link])));
        const v3d d = tip - (bg.p - qrot(bg.r, tod(f3(s.com[s.target_link]))));
        m.put3(m.lay.io + tq + nq + td, tof(d));
      }
    }
    phase_sync();
  }
}

// kinematics.forward + com.from_world from (q, qd) held in the io staging rows (q at rows
// [0, n_q), qd at rows [n_q, n_q + n_dof)); writes the state rows.  The tree is walked level by
// level (links of one depth in parallel); link-frame origins and their velocities are kept in
// the wrench rows (free at this point).  Wavefront-uniform call; `go`: envs that take part.
static __device__ CARL_BRAX_RESET_INLINE void forward_kinematics(const carl_brax_sys_t& s, const Topo& tp, const Lds& m, bool go) {
  for (int lvl = 0; lvl <= tp.max_depth; ++lvl) {
    for (int i = m.sub; i < s.n_links; i += kSub) {
      if (!go || tp.depth[i] != lvl) continue;
      const int P = s.parent[i];
      qt rot;
      v3 o, vel, ang;
      const int q0 = m.lay.io + s.q_start[i], d0 = m.lay.io + s.n_q + s.dof_start[i];
      if (is_free_root(s, i)) {
        rot = qnormalize(qt{m.at(q0 + 3), m.at(q0 + 4), m.at(q0 + 5), m.at(q0 + 6)});
        o = V(m.at(q0), m.at(q0 + 1), m.at(q0 + 2));
        vel = V(m.at(d0), m.at(d0 + 1), m.at(d0 + 2));
        ang = V(m.at(d0 + 3), m.at(d0 + 4), m.at(d0 + 5));
      } else {
        const Body bpd = (P < 0) ? world_body() : m.body(P);
        struct { qt r; v3 w; } bp{tof(bpd.r), bpd.w};  // the reset pose is a draw: float32 kinematics, widened below
        const v3 o_p = (P < 0) ? V(0, 0, 0) : m.get3(m.lay.wrench + 12 * P);
        const v3 ov_p = (P < 0) ? V(0, 0, 0) : m.get3(m.lay.wrench + 12 * P + 3);
        const int ns = s.n_slide[i], nr = s.n_link_dof[i] - ns;
        const qt jr = f4(s.joint_rot[i]), lrot = f4(s.link_rot[i]);
        const qt rpj = qmul(qmul(bp.r, lrot), jr);  // parent-side joint frame in the world
        // hinges stack intrinsically about the joint frame's x, y, +-z
        qt rj{1.0f, 0.0f, 0.0f, 0.0f};
        v3 wj = V(0, 0, 0);
        for (int k = 0; k < nr; ++k) {
          const float sg = (k == 2) ? s.dof_sign3[i] : 1.0f;
          const v3 axis =
              qrot(qmul(rpj, rj), V(k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f)) * sg;
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
      m.put3(m.lay.wrench + 12 * i, o);
      m.put3(m.lay.wrench + 12 * i + 3, vel);
      const v3 c = qrot(rot, f3(s.com[i]));
      Body b;
      {
        // The float32 kinematics leave |rot|^2 = 1 +- 2e-7; the state record keeps 48 bits, so the rotation is made a
        // unit quaternion in float64 here, once.  Why it matters (round 6, tools/diag_humanoid_reset_step.py): rotation
        // formulas that agree for unit quaternions differ by (|q|^2 - 1) x the vector for others -- v + 2 w (u x v) +
        // 2 u x (u x v) keeps an unscaled v, brax's math.rotate (2 (u.v) u + (w^2 - u.u) v + 2 w (u x v)) scales the whole
        // result -- and the hinge-alignment term turns that 1e-7 into k_pos x 1e-7 = 5e-3 N m on every hinge that is
        // not parallel to e_x: 1.8e-5 rad/s per substep between this kernel (which forms the misalignment from the
        // relative rotation: scale-type) and the float64 restatement (which crosses two rotated axes) on the FIRST
        // steps after every reset -- the Humanoid's worst parity entries (1.0e-5) all sat there.  After the first
        // substep the integrator's own float64 normalisation has always made the question moot.
        // (rounded to what the state record holds -- float32 head + tail, 48 bits -- so that a fused launch continues after
        // an in-kernel reset from exactly the state a per-call step stores and reloads: rollout == repeated step)
        const qtd r64 = tod(rot);
        const double inv = 1.0 / sqrt(r64.w * r64.w + r64.x * r64.x + r64.y * r64.y + r64.z * r64.z);
        auto r48 = [](double d) {
          const float hi = (float)d;
          return (double)hi + (double)(float)(d - (double)hi);
        };
        b.r = qtd{r48(r64.w * inv), r48(r64.x * inv), r48(r64.y * inv), r48(r64.z * inv)};
      }
      b.w = ang;
      b.p = tod(o + c);
      b.v = vel + cross(ang, c);
      m.put(i, b);
    }
    phase_sync();
  }
}

// brax.envs.<env>.reset: q = init_q + U(-noise, noise), qd = vel_scale * N(0, 1) (humanoid:
// U(-scale, scale)).  Draw k uses word (k mod 4) of Philox block k / 4 on sub-stream
// 0x80000000 | block; normals are Box-Muller pairs from two consecutive draws.
static __device__ __forceinline__ float draw_u(uint64_t seed, uint64_t g, uint32_t ep, int k) {
  const u32x4 w = lane_words(seed, g, ep, 0x80000000u | (uint32_t)(k >> 2));
  const uint32_t x = (k & 3) == 0 ? w.x : (k & 3) == 1 ? w.y : (k & 3) == 2 ? w.z : w.w;
  return u01(x);
}

// push task: the env's goal position (context rows goal_position_*, else the model's) -> the goal rows
static __device__ __forceinline__ void put_goal(const carl_brax_sys_t& s, const carl_batch_t& b, const Lds& m, int c, bool go) {
  if (go)
    for (int j = m.sub; j < 3; j += kSub) {
      const int row = s.ctx.goal_position[j];
      m.at(m.lay.goal + j) = row >= 0 ? b.ctx_table[(size_t)row * b.ctx_stride + c] : s.push_goal[j];
    }
}

// wavefront-uniform call; `go`: envs that are reset.  Push task: the caller has put the envs' goal rows
// (put_goal + phase_sync) -- the batch descriptor stays out of this function's arguments (it was a non-inlined call
// until round 3: by reference the descriptor would have lived in scratch memory for the whole kernel).
template <bool TASK>
static __device__ CARL_BRAX_RESET_INLINE void reset_state(const carl_brax_sys_t& s, const Topo& tp, uint64_t seed, const Lds& m,
                                         uint64_t genv, uint32_t episode, bool go) {
  if (go) {
    const int tl = !TASK ? 0 : s.target_link > 0 ? s.target_link : s.push_link;  // reach / push task: the last link's
    const int tq = tl > 0 ? s.q_start[tl] : s.n_q;                    // coordinates / rates are set by the
    const int td = tl > 0 ? s.dof_start[tl] : s.n_dof;                // task, not by the noise
    for (int i = m.sub; i < s.n_q; i += kSub) {
      float v = s.init_q[i] + s.reset_noise_scale * (2.0f * draw_u(seed, genv, episode, i) - 1.0f);
      if (TASK && i >= tq) {
        const float u0 = draw_u(seed, genv, episode, s.n_q + s.n_dof);
        const float u1 = draw_u(seed, genv, episode, s.n_q + s.n_dof + 1);
        if (s.target_link > 0) {  // brax.envs.reacher._random_target: uniform distance and bearing
          float sn, cs;
          sincos_fast(2.0f * kPiF * u1, sn, cs);
          v = s.target_max_dist * u0 * (i == tq ? cs : sn);
        } else {  // brax.envs.pusher.reset: a box in front of the arm, pushed out of the goal's disc
          const float gx = m.at(m.lay.goal), gy = m.at(m.lay.goal + 1);
          float dx = s.link_pos[tl][0] + s.push_lo[0] + (s.push_hi[0] - s.push_lo[0]) * u0 - gx;
          float dy = s.link_pos[tl][1] + s.push_lo[1] + (s.push_hi[1] - s.push_lo[1]) * u1 - gy;
          const float nrm = sqrtf(dx * dx + dy * dy);
          if (nrm < s.push_min_dist) {
            const float sc = s.push_min_dist / fmaxf(nrm, 1e-12f);
            dx *= sc;
            dy *= sc;
          }
          v = (i == tq) ? gx + dx - s.link_pos[tl][0] : gy + dy - s.link_pos[tl][1];
        }
      }
      m.at(m.lay.io + i) = v;
    }
    if (s.reset_vel_uniform) {
      for (int i = m.sub; i < s.n_dof; i += kSub)
        m.at(m.lay.io + s.n_q + i) =
            (TASK && i >= td) ? 0.0f : s.reset_vel_scale * (2.0f * draw_u(seed, genv, episode, s.n_q + i) - 1.0f);
    } else {
      for (int i = 2 * m.sub; i < s.n_dof; i += 2 * kSub) {  // one Box-Muller pair per lane
        const int k = s.n_q + i;
        const float u1 = draw_u(seed, genv, episode, k), u2 = draw_u(seed, genv, episode, k + 1);
        const float rad = sqrtf(-2.0f * logf(1.0f - u1));
        float sn, cs;
        sincos_fast(2.0f * kPiF * u2, sn, cs);
        m.at(m.lay.io + s.n_q + i) = s.reset_vel_scale * rad * cs;
        if (i + 1 < s.n_dof) m.at(m.lay.io + s.n_q + i + 1) = s.reset_vel_scale * rad * sn;
      }
    }
  }
  phase_sync();
  forward_kinematics(s, tp, m, go);
}

// context scalars of the env + its mass rows (wavefront-uniform call; ends with a phase_sync)
template <bool TASK>
static __device__ __forceinline__ LaneCtx load_ctx(const carl_brax_sys_t& s, const carl_batch_t& b, const Lds& m, int c,
                                            bool go) {
  const carl_brax_ctx_map_t& cm = s.ctx;
  auto get = [&](int row, float dflt) { return row >= 0 ? b.ctx_table[(size_t)row * b.ctx_stride + c] : dflt; };
  LaneCtx lc{};
  if (go) {
    lc.gravity_z = get(cm.gravity, s.gravity_z);
    lc.friction = get(cm.friction, s.friction);
    lc.elasticity = get(cm.elasticity, s.elasticity);
    lc.ang_damping = get(cm.ang_damping, s.ang_damping);
    lc.stiffness_scale = get(cm.joint_stiffness_scale, 1.0f);
    lc.da = __expf(lc.ang_damping * s.dt);
    for (int i = m.sub; i < s.n_links; i += kSub) m.at(m.lay.mass + i) = s.mass[i];
    if (m.sub == 0) {  // the env's context record: what the substep's phases read
      float* cr = const_cast<float*>(m.ctx_rec());
      *reinterpret_cast<vf4*>(cr) = vf4{lc.gravity_z, lc.friction, lc.elasticity, lc.da};
      cr[4] = lc.stiffness_scale;
    }
  }
  if (TASK && s.push_link > 0) put_goal(s, b, m, c, go);
  phase_sync();
  if (go) {
    // stability clamp of the effective mass ratio (carl_brax_ctx_map_t::mass_ratio_floor): the higher floor when
    // two or more links of the env are lighter than nominal
    int n_light = 0;
    for (int k = 0; k < cm.n_mass; ++k)
      n_light += (b.ctx_table[(size_t)cm.mass_row[k] * b.ctx_stride + c] / cm.mass_nominal[k] < 0.999f) ? 1 : 0;
    for (int k = m.sub; k < cm.n_mass; k += kSub)
      m.at(m.lay.mass + cm.mass_link[k]) =
          s.mass[cm.mass_link[k]] * fmaxf(b.ctx_table[(size_t)cm.mass_row[k] * b.ctx_stride + c] / cm.mass_nominal[k],
                                          n_light >= 2 ? cm.mass_ratio_floor_multi[k] : cm.mass_ratio_floor[k]);
  }
  phase_sync();
  return lc;
}

// env-major records (W floats per env) <-> the env's io staging rows; each env's 8 lanes move
// their own record (32-byte segments per env; L2 merges them into full lines)
static __device__ __forceinline__ void record_in(const float* __restrict__ src, size_t env, int W, const Lds& m, bool go) {
  if (go)
    for (int k = m.sub; k < W; k += kSub) m.at(m.lay.io + k) = src[env * W + k];
  phase_sync();
}
static __device__ __forceinline__ void record_out(float* __restrict__ dst, size_t env, int W, const Lds& m, bool go) {
  if (go)
    for (int k = m.sub; k < W; k += kSub) dst[env * W + k] = m.at(m.lay.io + k);
}

// ---- the env's persistent record <-> LDS.  HBM holds one 80-byte record per (env, link) -- pose head 7 | pose tail 7 |
// velocities 6 floats (include/carl_amd.h: CARL_BRAX_LINK_RECORD) -- and the lane that owns the link moves it as five
// 16-byte pieces: a wavefront's envs x links are one contiguous run of whole 16-byte pieces in memory (round 4 kept
// three blocks per env -- heads, tails, velocities -- and moved them 4 bytes per lane, 36-byte pieces at 720-byte stride:
// the counter traffic of an Ant launch was 1.87 x its algorithmic bytes).
static __device__ __forceinline__ void record_load(const float* __restrict__ src, const Lds& m, int L, bool go) {
  if (!go || m.sub >= L) return;
  const vf4* p = reinterpret_cast<const vf4*>(src + CARL_BRAX_LINK_RECORD * m.sub);
  const vf4 a = p[0], b4 = p[1], c = p[2], d = p[3], e = p[4];
  // a: p.x p.y p.z r.w | b4: r.x r.y r.z tp.x | c: tp.y tp.z tr.w tr.x | d: tr.y tr.z v.x v.y | e: v.z w.x w.y w.z
  Body b;
  b.p = D((double)a.x + (double)b4.w, (double)a.y + (double)c.x, (double)a.z + (double)c.y);
  b.r = qtd{(double)a.w + (double)c.z, (double)b4.x + (double)c.w, (double)b4.y + (double)d.x, (double)b4.z + (double)d.y};
  b.v = V(d.z, d.w, e.x);
  b.w = V(e.y, e.z, e.w);
  m.put(m.sub, b);
}
static __device__ __forceinline__ void record_store(float* __restrict__ dst, const Lds& m, int L, bool go) {
  if (!go || m.sub >= L) return;
  const Body b = m.body(m.sub);
  const double pose[7] = {b.p.x, b.p.y, b.p.z, b.r.w, b.r.x, b.r.y, b.r.z};
  float hi[7], lo[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    hi[k] = (float)pose[k];
    lo[k] = (float)(pose[k] - (double)hi[k]);
  }
  vf4* p = reinterpret_cast<vf4*>(dst + CARL_BRAX_LINK_RECORD * m.sub);
  p[0] = vf4{hi[0], hi[1], hi[2], hi[3]};
  p[1] = vf4{hi[4], hi[5], hi[6], lo[0]};
  p[2] = vf4{lo[1], lo[2], lo[3], lo[4]};
  p[3] = vf4{lo[5], lo[6], b.v.x, b.v.y};
  p[4] = vf4{b.v.z, b.w.x, b.w.y, b.w.z};
}
// a pose coordinate as the state record holds it: float32 head + float32 tail (48 significant bits)
static __device__ __forceinline__ double round48(double d) {
  const float hi = (float)d;
  return (double)hi + (double)(float)(d - (double)hi);
}

struct LaneState {
  float ep_return;
  int elapsed, cidx, n_new_calls, n_new_episodes;
  uint32_t episode;
  float goal_x, goal_y, goal_radius, pos_x, pos_y;  // goal mode only
};

static __device__ __forceinline__ void stash_put(const Lds& m, const LaneState& r) {
  const int k = m.lay.stash;
  m.at(k) = r.ep_return; m.atu(k + 1) = (uint32_t)r.elapsed; m.atu(k + 2) = (uint32_t)r.cidx;
  m.atu(k + 3) = (uint32_t)r.n_new_calls; m.atu(k + 4) = (uint32_t)r.n_new_episodes; m.atu(k + 5) = r.episode;
  m.at(k + 6) = r.goal_x; m.at(k + 7) = r.goal_y; m.at(k + 8) = r.goal_radius; m.at(k + 9) = r.pos_x; m.at(k + 10) = r.pos_y;
}
static __device__ __forceinline__ void stash_get(const Lds& m, LaneState& r) {
  const int k = m.lay.stash;
  r.ep_return = m.at(k); r.elapsed = (int)m.atu(k + 1); r.cidx = (int)m.atu(k + 2);
  r.n_new_calls = (int)m.atu(k + 3); r.n_new_episodes = (int)m.atu(k + 4); r.episode = m.atu(k + 5);
  r.goal_x = m.at(k + 6); r.goal_y = m.at(k + 7); r.goal_radius = m.at(k + 8); r.pos_x = m.at(k + 9); r.pos_y = m.at(k + 10);
}

// BraxWalkerGoalWrapper (carl/envs/brax/brax_walker_goal_wrapper.py:69-121): compass code ->
// goal position = direction * target_distance; radius
static __device__ __forceinline__ void load_goal(const carl_brax_sys_t& s, const carl_batch_t& b, int c, LaneState& r) {
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

static __device__ __forceinline__ void write_ctx_obs(const carl_batch_t& b, const Lds& m, size_t n, int env, int cidx) {
  if (b.ctx_obs != nullptr)
    for (int k = m.sub; k < b.n_ctx_obs; k += kSub)
      b.ctx_obs[(size_t)k * n + env] = b.ctx_table[(size_t)b.ctx_obs_feat[k] * b.ctx_stride + cidx];
}

// mode 0: reset (mask optional), mode 1: n_steps env steps (1 = per call, T = fused rollout)
// F32 (MODE 1, CARL_FLAG_BRAX_FP32): the n_frames substeps of an env step run their pose algebra in float32.
template <int MODE, bool MULTI, bool TASK, bool PLANAR = false, bool F32 = false>
static __device__ __forceinline__ void run(const carl_batch_t& b, const carl_brax_sys_t* __restrict__ sys_dev,
                                           const Prepared& prep, const carl_step_io_t& io,
                                           const uint8_t* __restrict__ mask, float* __restrict__ reset_obs,
                                           const int n_steps) {
  __shared__ carl_brax_sys_t s;
  __shared__ Topo tp;
  __shared__ Packed pk;
  __shared__ LinkRec jx_lds[CARL_BRAX_MAX_LINKS];
  __shared__ vf4 dof_act_lds[CARL_BRAX_MAX_DOF];
  __shared__ SphRec sph_lds[CARL_BRAX_MAX_COLL + 1];
  __shared__ int head_done[kMaxWavesPerWg3];  // fragment hand-over flags (MODE 1, see below)
  if (threadIdx.x < kMaxWavesPerWg3) head_done[threadIdx.x] = 0;
  extern __shared__ vf4 lds_dyn[];  // per wavefront: body records, then the float rows (16-byte aligned slices)
  {  // model table -> LDS, once per workgroup: every load in flight before the first LDS write (a
     // load-store loop paid one HBM/L2 round trip per 256 bytes: ~15 us of a per-call step)
    constexpr int kWords = (int)(sizeof(carl_brax_sys_t) / 4);
    constexpr int kPer = (kWords + kLanes - 1) / kLanes;  // (sized for a one-wavefront workgroup)
    const int kThreads = (int)blockDim.x;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(sys_dev);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&s);
    uint32_t tmp[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int k = j * kThreads + (int)threadIdx.x;
      tmp[j] = (k < kWords) ? src[k] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int k = j * kThreads + (int)threadIdx.x;
      if (k < kWords) dst[k] = tmp[j];
    }
  }
  __syncthreads();
  {  // topology + derived constants: kernel argument -> LDS (per-lane indexed in the phases)
    constexpr int kWordsT = (int)(sizeof(Topo) / 4), kWordsD = (int)(sizeof(Packed) / 4);
    static_assert(sizeof(Topo) % 4 == 0 && sizeof(Packed) % 4 == 0, "word copies");
    const uint32_t* st = reinterpret_cast<const uint32_t*>(&prep.topo);
    const uint32_t* sd = reinterpret_cast<const uint32_t*>(&prep.packed);
    uint32_t* dt = reinterpret_cast<uint32_t*>(&tp);
    uint32_t* dd = reinterpret_cast<uint32_t*>(&pk);
    for (int k = (int)threadIdx.x; k < kWordsT; k += (int)blockDim.x) dt[k] = st[k];
    for (int k = (int)threadIdx.x; k < kWordsD; k += (int)blockDim.x) dd[k] = sd[k];
  }
  __syncthreads();
  if ((int)threadIdx.x < s.n_links) expand_link(s, pk, (int)threadIdx.x, jx_lds[threadIdx.x]);
  for (int k = (int)threadIdx.x; k < CARL_BRAX_MAX_COLL + 1; k += (int)blockDim.x) {
    const int kk = k < CARL_BRAX_MAX_COLL ? k : CARL_BRAX_MAX_COLL - 1;
    const Sphere sp = pk.sph[kk];
    SphRec r;
    r.off[0] = sp.off[0]; r.off[1] = sp.off[1]; r.off[2] = sp.off[2]; r.radius = sp.radius;
    r.f[0] = sp.off[0]; r.f[1] = sp.off[1]; r.f[2] = sp.off[2]; r.f[3] = sp.radius;
    sph_lds[k] = r;
  }
  if ((int)threadIdx.x < CARL_BRAX_MAX_DOF) {  // dof -> its actuator (gear, control range, index; -1: none), for the step's torque pass
    const int d = (int)threadIdx.x;
    int a = -1;
    for (int k = 0; k < s.n_act; ++k) a = (s.act_dof[k] == d) ? k : a;
    const int ac = a < 0 ? 0 : a;
    dof_act_lds[d] = vf4{s.act_gear[ac], s.act_lo[ac], s.act_hi[ac], __int_as_float(a)};
  }
  __syncthreads();
  const LinkRec* const jx = jx_lds;
  [[maybe_unused]] const vf4* const dof_act = dof_act_lds;
  // kSub need not divide 64 (one lane per link: 7, 9, 11): the wavefront's spare lanes idle -- they
  // point at the last env's column, own no link (sub beyond every loop bound) and are never active
  const int tid = (int)threadIdx.x & (kLanes - 1), wave = (int)threadIdx.x >> 6;
  const bool lane_ok = tid < kEnvs * kSub;
  const Layout lay = layout_of(s);
  char* const my_lds = reinterpret_cast<char*>(lds_dyn) + (size_t)wave * lay.bytes(kEnvs);  // this wavefront's slice
  const Lds m{my_lds, reinterpret_cast<float*>(my_lds + lay.rows_offset(kEnvs)), lay,
              lane_ok ? tid / kSub : kEnvs - 1, lane_ok ? tid % kSub : kLanes};
  const size_t n = (size_t)b.n_lanes;
  const int S = CARL_BRAX_LINK_RECORD * s.n_links;  // floats of the env's record in HBM
  const bool goal = s.goal_mode != 0 && b.goal_pos != nullptr;
  const int n_waves = (int)blockDim.x >> 6;

  if constexpr (MODE == 0) {
    const int gwave = (int)blockIdx.x * n_waves + wave;  // global wavefront = group of kEnvs envs
    const int env = gwave * kEnvs + m.env;
    if (gwave * kEnvs >= b.n_lanes) return;  // a wavefront past the end of the batch
    const bool active = lane_ok && env < b.n_lanes;
    const uint64_t genv = (uint64_t)(b.lane_offset + env);
    LaneState r{};
    if (active) {
      r.cidx = b.ctx_idx[env];
      r.episode = b.episode[env];
    }
    const bool go = active && (mask == nullptr || mask[env] != 0);
    if (ballot(go) == 0ull) return;
    if (go) r.cidx = select_context(b, r.cidx, genv, r.episode);
    if (TASK && s.push_link > 0) {  // the object is placed relative to the env's goal
      put_goal(s, b, m, r.cidx, go);
      phase_sync();
    }
    reset_state<TASK>(s, tp, b.seed, m, genv, r.episode, go);
    r.episode += 1u;
    if (go) {
      record_store(b.state + (size_t)env * S, m, s.n_links, true);
      if (b.first_state != nullptr)  // what AUTORESET_FIRST_STATE puts a done env back to
        record_store(b.first_state + (size_t)env * S, m, s.n_links, true);
      if (m.sub == 0) {
        b.elapsed[env] = 0;
        b.ep_return[env] = 0.0f;
        b.ctx_idx[env] = r.cidx;
        b.episode[env] = r.episode;
        b.n_calls[env] += 1;
        if (goal) {  // wrapper reset: position = (0, 0)
          b.goal_pos[env] = 0.0f;
          b.goal_pos[n + env] = 0.0f;
        }
      }
      write_ctx_obs(b, m, n, env, r.cidx);
    }
    if (s.obs_extended) load_ctx<TASK>(s, b, m, r.cidx, go);  // com inertia / velocity use the env's masses
    observe<MULTI, TASK>(s, pk, m, go, true);
    if (reset_obs != nullptr) record_out(reset_obs, (size_t)env, s.obs_dim, m, go);
    return;
  } else {
    // ---- which (group, step range) fragments this wavefront runs.  A group = kEnvs envs = one wavefront's worth.
    // The workgroup owns the groups [g_lo, g_hi); with no more groups than wavefronts each wavefront runs one group for
    // all n_steps steps.  With more (a batch larger than the chip holds at once: the host then launches exactly
    // the resident number of workgroups) every wavefront lives for a whole launch and its latency-bound step
    // chain sets the pace, so "1.52 groups per wavefront" used to cost TWO full rounds (Ant, 32 768 envs: 4 682 groups on
    // 3 072 resident wavefronts).  Here the workgroup's G x n_steps group-steps, laid out group-major, are cut into
    // one contiguous piece per wavefront: whole groups, plus at most the TAIL of one group at the piece's start and
    // the HEAD of one at its end.  A wavefront runs its head fragment FIRST (then stores the group exactly as at the
    // end of a launch and raises its LDS flag), its whole groups, and its tail fragment LAST (after the previous
    // wavefront's flag: that head was the first thing it did, n_steps <= piece length earlier).  A fragment
    // boundary is what a launch boundary is -- state record and episode scalars through HBM, pose rounded at every
    // env step -- so results are bit-identical to one launch per group; both wavefronts are in one workgroup,
    // i.e. co-resident on one CU, so the wait cannot deadlock.
    constexpr bool kSlides = MULTI || PLANAR;  // (observe: a lean non-planar step kernel never sees a prismatic dof)
    const int T = n_steps;
    const int n_groups = ((int)b.n_lanes + kEnvs - 1) / kEnvs;
    const WgShare share = wg_share(n_groups, (int)gridDim.x, (int)blockIdx.x);
    const int g_lo = share.g_lo;
    const Piece piece = make_piece(share.G, T, n_waves, wave);
    const int n_frag = piece.n_frag;
    const float dt_env = s.dt * (float)s.n_frames;
    const SubK K = make_subk(s, tp, pk);
    const int n_frames = __builtin_amdgcn_readfirstlane(s.n_frames);
    if (tid < kWorldRecBytes / 4) {  // the world record (the first phase_sync below orders it)
      // world: pose (0, 0, 0 | 1, 0, 0, 0) as doubles, velocities 0; double 1.0 = words (0, 0x3ff00000) at double index 3
      reinterpret_cast<uint32_t*>(m.rec + m.world_off())[tid] = (tid == 7) ? 0x3ff00000u : 0u;
    }
    Prof prof;
    prof.start();
    for (int fi = 0; fi < n_frag; ++fi) {
    const Fragment frag = fragment_of(piece, T, fi);
    const int grp = frag.grp, t_lo = frag.t_lo, t_hi = frag.t_hi;
    const bool wait_head = frag.wait_head, signal_head = frag.signal_head;
    const int gwave = g_lo + grp;
    const int env = gwave * kEnvs + m.env;
    const bool active = lane_ok && env < b.n_lanes;
    const bool lead = active && m.sub == 0;  // the lane that writes the env's scalars
    if (wait_head) {  // the head of this group: the previous wavefront's first fragment
      while (__hip_atomic_load(&head_done[wave - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
        __builtin_amdgcn_s_sleep(16);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    LaneState r{};
    if (active) {
      r.cidx = b.ctx_idx[env];
      r.episode = b.episode[env];
      r.elapsed = b.elapsed[env];
      r.ep_return = b.ep_return[env];
    }
    record_load(b.state + (size_t)env * S, m, s.n_links, active);
    load_ctx<TASK>(s, b, m, r.cidx, active);
    if (goal && active) {
      load_goal(s, b, r.cidx, r);
      r.pos_x = b.goal_pos[env];
      r.pos_y = b.goal_pos[n + env];
    }
    prof.mark(kProfLoad);
    // The env's action record of step t + 1 is fetched while step t computes (two registers per lane: every shipped
    // model has n_act <= 2 kSub) -- a load issued at the step's own start costs the wavefront one HBM round trip per env
    // step before anything can begin.  Models with more actuators per lane take the direct path.
    const int n_act = __builtin_amdgcn_readfirstlane(s.n_act);
    const bool act_prefetch = n_act <= 2 * kSub;
    const float* const act_env = static_cast<const float*>(io.action) + (size_t)env * n_act;
    const size_t act_step = n * (size_t)n_act;
    const int ak0 = m.sub, ak1 = m.sub + kSub;
    float a_next0 = 0.0f, a_next1 = 0.0f;
    if (act_prefetch && active) {
      if (ak0 < n_act) a_next0 = act_env[(size_t)t_lo * act_step + ak0];
      if (ak1 < n_act) a_next1 = act_env[(size_t)t_lo * act_step + ak1];
    }
    const Lds& m_launch = m;
    const int env_frag = env;
    for (int t = t_lo; t < t_hi; ++t) {
      // Every LDS address of the step is derived from an opaque copy of the lane's coordinates: as loop invariants the
      // compiler hoisted dozens of them (row addresses of the staging, tau, mass and stash rows ...) out of the step
      // loop, ran out of registers and reloaded them from scratch memory all through the step.
      Lds m_step = m_launch;
      int env_step = env_frag;  // (likewise the env's index: every per-env global address -- observation, reward, flag rows --
                                // is base + env x width, a 64-bit per-lane value each when hoisted)
      asm volatile("" : "+v"(m_step.sub), "+v"(m_step.env), "+v"(env_step));
      const Lds& m = m_step;
      const int env = env_step;
      const uint64_t genv = (uint64_t)(b.lane_offset + env);
      const size_t step_off = (size_t)t * n;
      // The env's episode scalars (identical in all its lanes) wait in LDS until the reward needs them: a dozen registers
      // per lane that the substeps and observe need more.
      if (lead) stash_put(m, r);
      if (act_prefetch) {
        if (active) {
          if (ak0 < n_act) m.at(m.lay.io + ak0) = a_next0;
          if (ak1 < n_act) m.at(m.lay.io + ak1) = a_next1;
          if (t + 1 < t_hi) {
            if (ak0 < n_act) a_next0 = act_env[(size_t)(t + 1) * act_step + ak0];
            if (ak1 < n_act) a_next1 = act_env[(size_t)(t + 1) * act_step + ak1];
          }
        }
        phase_sync();
      } else {
        record_in(static_cast<const float*>(io.action) + step_off * s.n_act, (size_t)env, s.n_act, m, active);
      }
      // the control cost's sum of squares, four actions per round trip (rows past n_act are read -- they exist: the staging
      // holds at least n_act rows, the index is clamped -- and masked; one LDS round trip per action made this loop
      // eight dependent waits per Ant env step)
      float ctrl = 0.0f;
      for (int k = 0; k < n_act; k += 4) {
        float u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = m.at(m.lay.io + min(k + j, n_act - 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) ctrl += (k + j < n_act) ? u[j] * u[j] : 0.0f;
      }
      // actuator.to_tau: every dof's torque in ONE pass from the dof -> actuator table (act_dof entries are distinct, checked
      // by the host: a dof has at most one actuator) -- tau = gear * clip(action), 0 for a dof without an actuator.  (Until
      // round 6: zero the rows, hand over, every actuator adds to its dof, hand over.)
      for (int d = m.sub; d < s.n_dof; d += kSub) {
        const vf4 da = dof_act[d];  // gear, lo, hi, actuator (int bits; -1: none)
        const int a = __float_as_int(da.w);
        m.at(m.lay.tau + d) = a >= 0 ? da.x * fminf(fmaxf(m.at(m.lay.io + max(a, 0)), da.y), da.z) : 0.0f;
      }
      phase_sync();
      float msum;
      // forward progress and root height: pose differences, float64
      // (the root's frame origin = its COM where com[0] = 0 -- Ant's torso: no float64 rotation of a zero vector)
      const bool root_com_off = s.com[0][0] != 0.0f || s.com[0][1] != 0.0f || s.com[0][2] != 0.0f;  // wavefront-uniform
      const double x0 = s.reward_on_com ? system_com(s, m, &msum).x
                                        : m.pos(0).x - (root_com_off ? qrot(m.rot(0), tod(f3(s.com[0]))).x : 0.0);
      prof.mark(kProfPrologue);
      if (active && m.sub == 1 % kSub) Lds::put_react(reinterpret_cast<char*>(m.base) + m.react_off(m.lay.L), V(0, 0, 0), V(0, 0, 0));  // the env's zero record
      {  // the n_frames substeps, the lane's body / torques / branch hashes in registers (StepRegs)
        // The lane's addresses are rebuilt per env step from an opaque copy of its lane coordinates: as launch
        // invariants the compiler kept all twelve in registers through observe / reward / the reset path as well, which
        // have none to spare (a scratch reload there waits for every store of the step before it).
        Lds ms = m;
        asm volatile("" : "+v"(ms.sub), "+v"(ms.env));
        const LaneLink ll = make_lane_link(pk, jx, sph_lds, K, ms);
        using Real = std::conditional_t<F32, float, double>;
        StepRegs<MULTI, Real> R;
        if constexpr (F32) {
          // the env's records -> the float32 layout the substeps read and write (Lds::put_body, float form); the world record
          // likewise (r.w = 1.0f is word 3).  Everything outside the substeps keeps reading float64 records: converted back below.
          static_assert(!TASK, "CARL_FLAG_BRAX_FP32: the task models' pair contact reads float64 records");
          const Body b64 = Lds::body_of(ll.own);
          R.b = narrow_body(b64);
          phase_sync();  // (every read of the float64 records above is done)
          if (ll.body) Lds::put_body(ll.own, R.b);
          if (tid < kWorldRecBytes / 4) reinterpret_cast<uint32_t*>(m.rec + m.world_off())[tid] = (tid == 3) ? 0x3f800000u : 0u;
          phase_sync();
        } else {
          R.b = Lds::body_of(ll.own);
        }
        {
          const int d = wa_dof(ll.wa) + wa_slides(ll.wa);  // the joint's first hinge dof
#pragma unroll
          for (int k = 0; k < (MULTI ? 3 : 1); ++k) R.tau[k] = m.at(m.lay.tau + min(d + k, s.n_dof - 1));
        }
        R.sig_hit = 0u;
        R.sig_lim = 0u;
        R.inv_m = __builtin_amdgcn_rcpf(m.at(m.lay.mass + ll.i));  // v_rcp_f32 (1 ulp): the phase is issue-bound
        if constexpr (PLANAR) {
          for (int f = 0; f < n_frames; ++f) substep_planar<TASK, Real>(s, tp, K, ll, m, R, prof);
        } else {
          for (int f = 0; f < n_frames; ++f) substep<MULTI, TASK, Real>(s, tp, K, ll, m, R, prof);
        }
        if constexpr (F32) {  // the world record back to its float64 form (observe reads it as a planar root's parent)
          if (tid < kWorldRecBytes / 4) reinterpret_cast<uint32_t*>(m.rec + m.world_off())[tid] = (tid == 7) ? 0x3ff00000u : 0u;
        }
        if (ll.body) {  // this step's branch record, per link
          m.atu(m.lay.sig + 2 * ll.i) = R.sig_hit;
          m.atu(m.lay.sig + 2 * ll.i + 1) = R.sig_lim;
          // Round the pose to what the state record holds (float32 head + tail, 48 bits) at the end of EVERY env step,
          // on the register copy: a fused rollout then continues from exactly the state a per-call step stores and
          // reloads -- rollout == repeated step, bit for bit -- and the step's observation, reward and health checks
          // read the rounded pose in both.
          Body bw = widen_body(R.b);  // (F32: a float is its own float32 head, tail 0 -- nothing to round)
          bw.p = D(round48(bw.p.x), round48(bw.p.y), round48(bw.p.z));
          bw.r = qtd{round48(bw.r.w), round48(bw.r.x), round48(bw.r.y), round48(bw.r.z)};
          Lds::put_body(ll.own, bw);
        }
        phase_sync();
      }
      const v3d c1 = root_com_off ? qrot(m.rot(0), tod(f3(s.com[0]))) : D(0, 0, 0);
      v3d com1 = D(0, 0, 0);
      if (s.reward_on_com) com1 = system_com(s, m, &msum);
      const double x1 = s.reward_on_com ? com1.x : m.pos(0).x - c1.x, z1d = m.pos(0).z - c1.z;
      const float z1 = (float)z1d;
      bool healthy = (z1d >= (double)s.healthy_z_lo) && (z1d <= (double)s.healthy_z_hi);
      if (lead && io.branch_sig != nullptr) {  // the step's branch record: the per-link hashes combined in link order
        uint32_t hc = 0u, hl = 0u;
        for (int i = 0; i < s.n_links; ++i) {
          hc = hc * 1000003u + m.atu(m.lay.sig + 2 * i);
          hl = hl * 1000003u + m.atu(m.lay.sig + 2 * i + 1);
        }
        io.branch_sig[(step_off + env) * 2] = hc;
        io.branch_sig[(step_off + env) * 2 + 1] = hl;
      }
      prof.mark(kProfEpilogue);
      observe<MULTI, TASK, kSlides>(s, pk, m, active, false, s.reward_on_com ? &com1 : nullptr, msum);
      prof.mark(kProfObserve);
      stash_get(m, r);
      r.elapsed += 1;
      const bool truncated = (b.max_episode_steps > 0) && (r.elapsed >= b.max_episode_steps);
      if (s.healthy_q_index >= 0) {  // torso pitch (hopper, walker2d) / pole angle: read from the observation
        const float qa = m.at(m.lay.io + s.healthy_q_index - s.exclude_current_positions);
        healthy = healthy && (qa >= s.healthy_q_lo) && (qa <= s.healthy_q_hi);
      }
      float reward = s.forward_reward_weight * (s.reward_height ? z1 : (float)(x1 - x0)) / dt_env +
                     (s.terminate_when_unhealthy ? s.healthy_reward : (healthy ? s.healthy_reward : 0.0f)) -
                     s.ctrl_cost_weight * ctrl;
      bool terminated = s.terminate_when_unhealthy && !healthy;
      if (TASK && s.push_link > 0) {  // brax.envs.pusher: the observation ends with end effector, object, goal
        const int k = m.lay.io + s.obs_dim - 9;
        const v3 tipc = m.get3(k), obj = m.get3(k + 3), gl = m.get3(k + 6);
        const v3 d1 = obj - tipc, d2 = obj - gl;
        reward = -sqrtf(dot(d2, d2)) - s.ctrl_cost_weight * ctrl - s.push_near_weight * sqrtf(dot(d1, d1));
        terminated = false;
      } else if (TASK && s.target_link > 0) {  // brax.envs.reacher: the observation ends with tip - goal
        const int k = m.lay.io + s.obs_dim - 3;
        const float dx = m.at(k), dy = m.at(k + 1), dz = m.at(k + 2);
        reward = -sqrtf(dx * dx + dy * dy + dz * dz) - s.ctrl_cost_weight * ctrl;
        terminated = false;
      } else if (s.tip_link > 0) {  // brax.envs.inverted_double_pendulum: alive bonus - distance - velocity penalties
        const v3d tipd = m.pos(s.tip_link) + qrot(m.rot(s.tip_link), tod(f3(s.tip_offset) - f3(s.com[s.tip_link])));
        const v3 tip = tof(tipd);
        const float dz = tip.z - s.tip_height;
        const float v0 = m.at(m.lay.wrench + s.tip_vel_dof[0]), v1 = m.at(m.lay.wrench + s.tip_vel_dof[1]);
        reward = s.healthy_reward - (s.tip_x_weight * tip.x * tip.x + dz * dz) -
                 (s.tip_vel_weight[0] * v0 * v0 + s.tip_vel_weight[1] * v1 * v1);
        terminated = tipd.z <= (double)s.tip_min_height;
      }
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
        if (lead && b.success != nullptr) b.success[step_off + env] = (uint8_t)ok;
      }
      r.ep_return += reward;
      const bool done = active && (terminated | truncated);
      if (lead) {
        io.reward[step_off + env] = reward;
        io.terminated[step_off + env] = (uint8_t)terminated;
        io.truncated[step_off + env] = (uint8_t)truncated;
        if (io.done != nullptr && n_steps == 1) io.done[env] = (uint8_t)(terminated | truncated);  // per-call step
      }
      prof.mark(kProfReward);
      if (__builtin_expect(ballot(done) != 0ull, 0)) {  // (cold: the register allocator may spill around it, not through the step)
        const float fin_ret = r.ep_return;
        const int fin_len = r.elapsed;
        if (done) {
          if (m.sub == 0) {
            if (b.last_return) b.last_return[env] = fin_ret;
            if (b.last_length) b.last_length[env] = fin_len;
          }
          r.n_new_episodes += 1;
        }
        log_finished(b, done && m.sub == 0, genv, fin_ret, fin_len);
        if (b.flags & CARL_FLAG_AUTORESET) {
          if (io.final_obs != nullptr)  // terminal observation, done envs only
            record_out(io.final_obs + step_off * s.obs_dim, (size_t)env, s.obs_dim, m, done);
          phase_sync();  // the io rows are rewritten below
          if ((b.flags & CARL_FLAG_AUTORESET_FIRST_STATE) && b.first_state != nullptr) {
            // brax AutoResetWrapper: the state of the last explicit reset, same context, nothing drawn
            record_load(b.first_state + (size_t)env * S, m, s.n_links, done);
            if (done) {
              r.elapsed = 0;
              r.ep_return = 0.0f;
              if (goal) r.pos_x = r.pos_y = 0.0f;
            }
            phase_sync();
          } else {
          if (done) r.cidx = select_context(b, r.cidx, genv, r.episode);
          if (TASK && s.push_link > 0) {
            put_goal(s, b, m, r.cidx, done);
            phase_sync();
          }
          reset_state<TASK>(s, tp, b.seed, m, genv, r.episode, done);
          load_ctx<TASK>(s, b, m, r.cidx, done);
          if (done) {
            r.episode += 1u;
            if (goal) {
              load_goal(s, b, r.cidx, r);
              r.pos_x = r.pos_y = 0.0f;
            }
            r.elapsed = 0;
            r.ep_return = 0.0f;
            r.n_new_calls += 1;
            write_ctx_obs(b, m, n, env, r.cidx);
          }
          }
          observe<MULTI, TASK, kSlides>(s, pk, m, done, true);
        }
      }
      prof.mark(kProfDone);
      record_out(io.obs + step_off * s.obs_dim, (size_t)env, s.obs_dim, m, active);
      phase_sync();  // the next step's actions overwrite the io rows
      prof.mark(kProfOutput);
    }
    if (active) {
      record_store(b.state + (size_t)env * S, m, s.n_links, true);
      if (m.sub == 0) {
        b.elapsed[env] = r.elapsed;
        b.ep_return[env] = r.ep_return;
        if (goal) {
          b.goal_pos[env] = r.pos_x;
          b.goal_pos[n + env] = r.pos_y;
        }
        if (r.n_new_episodes != 0 && b.episodes_done != nullptr) b.episodes_done[env] += r.n_new_episodes;
        if (r.n_new_calls != 0) {
          b.ctx_idx[env] = r.cidx;
          b.episode[env] = r.episode;
          b.n_calls[env] += r.n_new_calls;
        }
      }
    }
    if (signal_head) {  // the group continues in the next wavefront: everything above is in memory before the flag
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (tid == 0) __hip_atomic_store(&head_done[wave], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    phase_sync();  // the next fragment reloads this wavefront's rows
    prof.mark(kProfStore);
    }  // fragments
    prof.flush(tid == 0);
  }
}

};  // struct Group

// TASK: the model is a reach / push task (target_link / push_link / pair contacts); false compiles that code
// out, as MULTI = false does for the Euler-angle joints.  Task models always have a hinge-less last link,
// so TASK implies MULTI.
// PLANAR: substep_planar (step / rollout of a planar single-hinge model; the host checks the model, carl_brax.hip).
template <int MODE, bool MULTI, int K, bool TASK = false, bool PLANAR = false, bool F32 = false>
__global__ void __launch_bounds__(kLanes * max_waves_per_wg(TASK)) __attribute__((amdgpu_waves_per_eu(F32 ? CARL_BRAX_WAVES_PER_EU_F32(TASK) : CARL_BRAX_WAVES_PER_EU(TASK)))) brax_kernel(const carl_batch_t b, const carl_brax_sys_t* __restrict__ sys_dev,
                                                      const Prepared prep, const carl_step_io_t io,
                                                      const uint8_t* __restrict__ mask, float* __restrict__ reset_obs,
                                                      const int n_steps) {
  Group<K>::template run<MODE, MULTI, TASK, PLANAR, F32>(b, sys_dev, prep, io, mask, reset_obs, n_steps);
}

}  // namespace brax
}  // namespace carl
