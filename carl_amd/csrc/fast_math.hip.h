// fast_math.hip.h -- branch-free sin/cos for the step kernels.
//
// The fused rollout kernel runs one wavefront per SIMD at 65 536 lanes, so it is
// bound by the instruction stream of a single wave; the library sinf/cosf (Payne-Hanek
// large-argument path inlined twice) dominated that stream.  These versions are
//   k = rint(x * 2/pi);  r = x - k*(pi/2) with a 3-term Cody-Waite split (fma);
//   sin r, cos r by degree-9 / degree-8 polynomials on [-pi/4, pi/4];  quadrant fix-up.
// Measured against libm in extended precision (oracle-side C harness, 4e6 samples per
// range): max abs error 9.3e-8 (fp32, |x| <= 1e5) and 1.8e-16 (fp64, |x| <= 1e6), i.e.
// ~1 ulp -- far inside the 1e-5 parity budget.  Beyond those ranges the library
// functions are used (rare, wave-divergent branch; keeps huge angles correct).
#pragma once

#include <hip/hip_runtime.h>

#include "carl_device.hip.h"  // ballot()

namespace carl {

__device__ __forceinline__ void sincos_fast(float x, float& sn, float& cs) {
  const float two_over_pi = 0x1.45f306p-1f;
  const float hi = 0x1.921fb6p+0f, mid = -0x1.777a5cp-25f, lo = -0x1.ee59dap-50f;
  const float k = rintf(x * two_over_pi);
  float r = __fmaf_rn(k, -hi, x);
  r = __fmaf_rn(k, -mid, r);
  r = __fmaf_rn(k, -lo, r);
  const float z = r * r;
  float ps = __fmaf_rn(z, 2.7557314297e-06f, -1.9841270114e-04f);
  ps = __fmaf_rn(z, ps, 8.3333337680e-03f);
  ps = __fmaf_rn(z, ps, -1.6666667163e-01f);
  const float S = __fmaf_rn(r * z, ps, r);
  float pc = __fmaf_rn(z, -2.7557314297e-07f, 2.4801587642e-05f);
  pc = __fmaf_rn(z, pc, -1.3888889225e-03f);
  pc = __fmaf_rn(z, pc, 4.1666667908e-02f);
  const float C = __fmaf_rn(z * z, pc, __fmaf_rn(z, -0.5f, 1.0f));
  const int q = (int)k;
  const float s2 = (q & 1) ? C : S, c2 = (q & 1) ? S : C;
  sn = (q & 2) ? -s2 : s2;
  cs = ((q + 1) & 2) ? -c2 : c2;
  // huge / non-finite arguments: library path.  Tested with ONE wave-uniform branch (ballot)
  // after the unconditional fast path -- an exec-masked if/else around the fast path costs
  // ~8 scalar instructions per call, which matters when a single wave issues one
  // instruction per 4-cycle slot.
  const bool big = !(fabsf(x) <= 1.0e5f);  // also catches NaN/inf
  if (__builtin_expect(ballot(big) != 0ull, 0)) {
    if (big) sincosf(x, &sn, &cs);
  }
}

// ---- Pendulum's version: the same reduction constants and polynomials, arranged for the regime where ONE wavefront
// per SIMD issues the whole step (8 192 lanes per GPU: the 8-GPU split; ~90 instructions per step at one per ~5.5
// cycles), where every instruction is ~1 % of the step:
//  * k by the 1.5 * 2^23 trick: after t = fma(x, 2/pi, 1.5 * 2^23) the low mantissa bits of t ARE the integer
//    (two's complement, |k| < 2^22 since |x| <= 1e5), k = t - 1.5 * 2^23: no multiply + round + float->int
//    conversion.  (The product is not rounded before the nearest-integer step, so k differs from rint(fl(x 2/pi))
//    at exact ties only; r then sits a hair outside [-pi/4, pi/4], well inside the polynomials' accuracy.)
//  * the sine and cosine Horner chains side by side in the halves of v_pk_fma_f32 / v_pk_mul_f32 (the same IEEE
//    fma per half as the scalar instruction): 5 packed instructions for 10;
//  * quadrant signs by moving bit 1 of q (of q + 1) onto the sign bit with one shift / add and one v_bitop3_b32
//    each, instead of and + compare + select;
//  * the library path for huge arguments is a CALL (sincosf inlined is ~500 instructions that the register
//    allocator and scheduler of the hot loop otherwise have to work around).
// pendulum_8192_T1000 (tools/shard8_probe.py, one box, interleaved builds): 205 us with the plain version, 197 us
// with the compiler's SLP packing of it, 181 us with this one; no change at 65 536 lanes (store-bound there).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((noinline)) float2 sincosf_outlined(float x) {
  float s, c;
  sincosf(x, &s, &c);
  return make_float2(s, c);
}

__device__ __forceinline__ void sincos_fast_pk(float x, float& sn, float& cs) {
  const float two_over_pi = 0x1.45f306p-1f;
  const float hi = 0x1.921fb6p+0f, mid = -0x1.777a5cp-25f, lo = -0x1.ee59dap-50f;
  const float magic = 0x1.8p23f;
  const float t = __fmaf_rn(x, two_over_pi, magic);
  const float k = t - magic;
  float r = __fmaf_rn(k, -hi, x);
  r = __fmaf_rn(k, -mid, r);
  r = __fmaf_rn(k, -lo, r);
  const float z = r * r;
  const f32x2 zz = {z, z};
  f32x2 p = __builtin_elementwise_fma(zz, f32x2{2.7557314297e-06f, -2.7557314297e-07f},
                                      f32x2{-1.9841270114e-04f, 2.4801587642e-05f});
  p = __builtin_elementwise_fma(zz, p, f32x2{8.3333337680e-03f, -1.3888889225e-03f});
  p = __builtin_elementwise_fma(zz, p, f32x2{-1.6666667163e-01f, 4.1666667908e-02f});
  const f32x2 m = f32x2{r, z} * zz;  // {r z, z z}
  const f32x2 sc = __builtin_elementwise_fma(m, p, f32x2{r, __fmaf_rn(z, -0.5f, 1.0f)});  // {sin r, cos r}
  const unsigned q = __float_as_uint(t);
  const float s2 = (q & 1u) ? sc.y : sc.x, c2 = (q & 1u) ? sc.x : sc.y;
  sn = __uint_as_float(__float_as_uint(s2) ^ ((q << 30) & 0x80000000u));         // quadrants 2, 3
  cs = __uint_as_float(__float_as_uint(c2) ^ (((q + 1u) << 30) & 0x80000000u));  // quadrants 1, 2
  const bool big = !(fabsf(x) <= 1.0e5f);  // also catches NaN/inf
  if (__builtin_expect(ballot(big) != 0ull, 0)) {
    if (big) {
      const float2 lib = sincosf_outlined(x);
      sn = lib.x;
      cs = lib.y;
    }
  }
}

// For arguments that are small in practice (CartPole's pole angle: an episode ends at 0.21 rad): when EVERY lane
// of the wave has |x| <= 0.78 the reduction finds k = 0, r = x and quadrant 0, so the polynomials alone give the
// same bits as sincos_fast -- without the multiply / round / three-fma reduction and the quadrant swap and sign
// logic (14 of its 34 instructions); any larger |x| in the wave takes sincos_fast itself.  The polynomials come
// FIRST and unconditionally, the wave-uniform test after them: as `if (any large) {general; return;} polynomials`
// the structurizer gave the hot path five control-flow instructions per call (a flag move, two branches and their
// mask arithmetic) and cut the caller's step into three basic blocks; this way it is a compare and one
// branch-not-taken to an out-of-line block, and the polynomials schedule with the code around them.
__device__ __forceinline__ void sincos_fast_smallarg(float x, float& sn, float& cs) {
  const float z = x * x;
  float ps = __fmaf_rn(z, 2.7557314297e-06f, -1.9841270114e-04f);
  ps = __fmaf_rn(z, ps, 8.3333337680e-03f);
  ps = __fmaf_rn(z, ps, -1.6666667163e-01f);
  sn = __fmaf_rn(x * z, ps, x);
  float pc = __fmaf_rn(z, -2.7557314297e-07f, 2.4801587642e-05f);
  pc = __fmaf_rn(z, pc, -1.3888889225e-03f);
  pc = __fmaf_rn(z, pc, 4.1666667908e-02f);
  cs = __fmaf_rn(z * z, pc, __fmaf_rn(z, -0.5f, 1.0f));
  if (__builtin_expect(ballot(!(fabsf(x) <= 0.78f)) != 0ull, 0)) sincos_fast(x, sn, cs);
}

// A double constant pinned in a scalar register pair.  The fp64 Horner steps below are `p = fma(z, p, c)` with a
// CONSTANT addend; left alone, the compiler shrinks each to the two-address v_fmac_f64 and first copies the
// constant into the destination pair (v_mov_b64, plus v_mov_b32s to assemble pairs): 12 copies per sincos, a
// sixth of the Acrobot step's instruction stream (114 v_mov_b64 in its loop).  An SGPR addend cannot be a
// v_fmac destination, so the three-address v_fma_f64 with the constant on the scalar operand bus is what is left.
__device__ __forceinline__ double sconst(double c) {
  asm("" : "+s"(c));
  return c;
}

// 1 / d from v_rcp_f64 and two Newton steps (relative error ~1e-16, not correctly rounded): 5 instructions
// against the ~12 of the IEEE division sequence (div_scale x2, rcp, 5 fma, div_fmas, div_fixup)
__device__ __forceinline__ double rcp_fast(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  return fma(r, e, r);
}

// FALLBACK = false: the caller's arguments are bounded (Acrobot's RK4 stages: a wrapped angle plus at most a few
// turns), so the library path for |x| > 1e6 -- an exec-masked region with the Payne-Hanek reduction inlined,
// ~150 instructions and ~8 scalar instructions of mask bookkeeping per call site even when skipped -- is left
// out; beyond 1e6 the Cody-Waite reduction loses accuracy gradually, non-finite inputs give NaN.
template <bool FALLBACK = true>
__device__ __forceinline__ void sincos_fast(double x, double& sn, double& cs) {
  if constexpr (FALLBACK) {
    if (__builtin_expect(!(fabs(x) <= 1.0e6), 0)) {
      sincos(x, &sn, &cs);
      return;
    }
  }
  const double two_over_pi = 0x1.45f306dc9c883p-1;
  const double hi = 0x1.921fb54442d18p+0, mid = 0x1.1a62633145c07p-54, lo = -0x1.f1976b7ed8fbcp-110;
  const double k = rint(x * two_over_pi);
  double r = fma(k, -hi, x);
  r = fma(k, -mid, r);
  r = fma(k, -lo, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, sconst(-2.50507602534068634195e-08));
  ps = fma(z, ps, sconst(2.75573137070700676789e-06));
  ps = fma(z, ps, sconst(-1.98412698298579493134e-04));
  ps = fma(z, ps, sconst(8.33333333332248946124e-03));
  ps = fma(z, ps, sconst(-1.66666666666666324348e-01));
  const double S = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, sconst(2.08757232129817482790e-09));
  pc = fma(z, pc, sconst(-2.75573143513906633035e-07));
  pc = fma(z, pc, sconst(2.48015872894767294178e-05));
  pc = fma(z, pc, sconst(-1.38888888888741095749e-03));
  pc = fma(z, pc, sconst(4.16666666666666019037e-02));
  const double C = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)k;
  const double s2 = (q & 1) ? C : S, c2 = (q & 1) ? S : C;
  sn = (q & 2) ? -s2 : s2;
  cs = ((q + 1) & 2) ? -c2 : c2;
}

// Two angles at once, statement-interleaved: the four Horner chains (sin / cos of each angle) advance in lock
// step, so consecutive instructions are independent.  The compiler keeps source order here (it serialises the
// chains of sincos_fast one after the other), and with ONE wave per SIMD a dependent fp64 fma waits out its
// latency with nothing else to issue -- Acrobot's RK4 stages are two such angles each.  Same operations and
// constants as sincos_fast<false>: bit-identical results.
__device__ __forceinline__ void sincos2_fast(double xa, double xb, double& sna, double& csa, double& snb, double& csb) {
  const double two_over_pi = 0x1.45f306dc9c883p-1;
  const double hi = 0x1.921fb54442d18p+0, mid = 0x1.1a62633145c07p-54, lo = -0x1.f1976b7ed8fbcp-110;
  const double ka = rint(xa * two_over_pi), kb = rint(xb * two_over_pi);
  double ra = fma(ka, -hi, xa), rb = fma(kb, -hi, xb);
  ra = fma(ka, -mid, ra);
  rb = fma(kb, -mid, rb);
  ra = fma(ka, -lo, ra);
  rb = fma(kb, -lo, rb);
  const double za = ra * ra, zb = rb * rb;
  const double s5 = 1.58969099521155010221e-10, s4 = sconst(-2.50507602534068634195e-08),
               s3 = sconst(2.75573137070700676789e-06), s2 = sconst(-1.98412698298579493134e-04),
               s1 = sconst(8.33333333332248946124e-03), s0 = sconst(-1.66666666666666324348e-01);
  const double c5 = -1.13596475577881948265e-11, c4 = sconst(2.08757232129817482790e-09),
               c3 = sconst(-2.75573143513906633035e-07), c2 = sconst(2.48015872894767294178e-05),
               c1 = sconst(-1.38888888888741095749e-03), c0 = sconst(4.16666666666666019037e-02);
  double psa = fma(za, s5, s4), pca = fma(za, c5, c4), psb = fma(zb, s5, s4), pcb = fma(zb, c5, c4);
  const double rza = ra * za, rzb = rb * zb;
  psa = fma(za, psa, s3); pca = fma(za, pca, c3); psb = fma(zb, psb, s3); pcb = fma(zb, pcb, c3);
  const double zza = za * za, zzb = zb * zb;
  psa = fma(za, psa, s2); pca = fma(za, pca, c2); psb = fma(zb, psb, s2); pcb = fma(zb, pcb, c2);
  const double ha = fma(za, -0.5, 1.0), hb = fma(zb, -0.5, 1.0);
  psa = fma(za, psa, s1); pca = fma(za, pca, c1); psb = fma(zb, psb, s1); pcb = fma(zb, pcb, c1);
  const int qa = (int)ka, qb = (int)kb;
  psa = fma(za, psa, s0); pca = fma(za, pca, c0); psb = fma(zb, psb, s0); pcb = fma(zb, pcb, c0);
  const double Sa = fma(rza, psa, ra), Ca = fma(zza, pca, ha), Sb = fma(rzb, psb, rb), Cb = fma(zzb, pcb, hb);
  const double s2a = (qa & 1) ? Ca : Sa, c2a = (qa & 1) ? Sa : Ca;
  const double s2b = (qb & 1) ? Cb : Sb, c2b = (qb & 1) ? Sb : Cb;
  sna = (qa & 2) ? -s2a : s2a;
  csa = ((qa + 1) & 2) ? -c2a : c2a;
  snb = (qb & 2) ? -s2b : s2b;
  csb = ((qb + 1) & 2) ? -c2b : c2b;
}

// ... and with ONE Newton step: v_rcp_f64 is good to 2^29 ulp (2^-24 relative), one step squares that (~4e-15) -- what
// Acrobot's `_dsdt` uses for 1 / det: its result feeds float32 state at a 1e-5 parity bar (131 072 random transitions
// against the float64 oracle: tests/test_gpu_parity.py::test_random_transitions_100k), two dependent float64 fmas
// fewer per RK4 stage
__device__ __forceinline__ double rcp_fast1(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  return fma(r, fma(-d, r, 1.0), r);
}

__device__ __forceinline__ float cos_fast(float x) {
  float s, c;
  sincos_fast(x, s, c);
  return c;
}

__device__ __attribute__((noinline)) float cos_fast_outlined(float x) { return cos_fast(x); }

// cos(2h) for the MountainCar families' cos(3 position) (h = 1.5 position; position lives in [-1.2, 0.6] with
// the default contexts, so |h| <= 1.8): c = cos h by an even degree-10 polynomial fitted on |h| <= 1.86 (least
// squares on Chebyshev nodes, coefficients rounded to fp32), then the double-angle identity 2 c^2 - 1 -- no
// reduction, no sine polynomial, no quadrant logic: 9 vector instructions against the 34 of sincos_fast.  Max abs
// error against libm's double cos over |2h| <= 3.7 (2e6 samples, the fp32 roundings emulated:
// tests/test_sincos_table.py): 1.7e-7.  A lane outside the fitted range (contexts that move min_position /
// max_position) takes cos_fast(2h) itself -- selected PER LANE inside the rare wave-uniform branch, so a lane's
// result never depends on its wave mates -- and as a CALL: inlined, the range-reduced version and its library path
// sat in the middle of the step loop (of BOTH loops of the Acrobot + MountainCar pair kernel, whose Acrobot half then
// scheduled 1.4 % slower).
__device__ __forceinline__ float cos_twice_fast(float h) {
  const float z = h * h;
  float p = __fmaf_rn(z, -0x1.1173p-22f, 0x1.9ec0c8p-16f);
  p = __fmaf_rn(z, p, -0x1.6c0d24p-10f);
  p = __fmaf_rn(z, p, 0x1.555518p-5f);
  p = __fmaf_rn(z, p, -0x1.fffffep-2f);
  const float c = __fmaf_rn(z, p, 1.0f);
  float r = __fmaf_rn(c + c, c, -1.0f);
  const bool wide = !(fabsf(h) <= 1.85f);  // also catches NaN
  if (__builtin_expect(ballot(wide) != 0ull, 0)) {
    if (wide) r = cos_fast_outlined(h + h);
  }
  return r;
}

// atan2 with one reduction step and a degree-7 odd polynomial (Cephes atanf coefficients):
// max abs error 2.8e-7 over [-1, 1]^2 incl. tiny arguments (host harness against libm's double
// atan2, 2e7 samples) -- the rounding of the result itself near +-pi is 2.4e-7.  ~25
// instructions against ~70 for the library call.
__device__ __forceinline__ float atan2_fast(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float t = mn * __builtin_amdgcn_rcpf(mx);
  t = (mx > 0.0f) ? t : 0.0f;
  const bool mid = t > 0.41421356237f;  // tan(pi/8): atan(t) = pi/4 + atan((t - 1) / (t + 1))
  const float tr = mid ? (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f) : t;
  const float z = tr * tr;
  float p = __fmaf_rn(z, 8.05374449538e-2f, -1.38776856032e-1f);
  p = __fmaf_rn(z, p, 1.99777106478e-1f);
  p = __fmaf_rn(z, p, -3.33329491539e-1f);
  float r = __fmaf_rn(p * z, tr, tr) + (mid ? 0.78539816339f : 0.0f);
  r = (ay > ax) ? 1.57079632679f - r : r;
  r = (x < 0.0f) ? 3.14159265359f - r : r;
  return copysignf(r, y);
}

// a / b with one v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE sequence
__device__ __forceinline__ float div_fast(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

}  // namespace carl
