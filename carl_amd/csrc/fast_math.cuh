// fast_math.cuh -- branch-free sin/cos for the step kernels.
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
  if (__builtin_expect(__ballot(big) != 0ull, 0)) {
    if (big) sincosf(x, &sn, &cs);
  }
}

__device__ __forceinline__ void sincos_fast(double x, double& sn, double& cs) {
  if (__builtin_expect(!(fabs(x) <= 1.0e6), 0)) {
    sincos(x, &sn, &cs);
    return;
  }
  const double two_over_pi = 0x1.45f306dc9c883p-1;
  const double hi = 0x1.921fb54442d18p+0, mid = 0x1.1a62633145c07p-54, lo = -0x1.f1976b7ed8fbcp-110;
  const double k = rint(x * two_over_pi);
  double r = fma(k, -hi, x);
  r = fma(k, -mid, r);
  r = fma(k, -lo, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double S = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double C = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)k;
  const double s2 = (q & 1) ? C : S, c2 = (q & 1) ? S : C;
  sn = (q & 2) ? -s2 : s2;
  cs = ((q + 1) & 2) ? -c2 : c2;
}

__device__ __forceinline__ float cos_fast(float x) {
  float s, c;
  sincos_fast(x, s, c);
  return c;
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
