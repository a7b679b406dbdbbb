// context_kernels.hip.h -- dense context sets produced and checked on the device.
//
// Replaces, for C contexts at once, ContextSampler.sample_contexts (carl/context/sampler.py:45-61)
// + the default fill of the contexts setter (carl/envs/carl_env.py:135-137) and
// ContextSpace.verify_context (carl/context/context_space.py:54-59).  One thread = one context;
// a wavefront writes 256 contiguous bytes of each feature row ([F][C] table).  Philox counter =
// (global context id lo, hi, feature index, kSubSampler | attempt): see include/carl_amd.h.
#pragma once

#include "carl_device.hip.h"
#include "fast_math.hip.h"

namespace carl {

constexpr uint32_t kSubSampler = 0x40000000u;
constexpr int kNormalTries = 32;

__device__ __forceinline__ float sample_feature(const carl_feature_spec_t& sp, uint64_t seed, uint64_t gctx, uint32_t f) {
  if (sp.kind == CARL_FEAT_CONSTANT) return sp.value;
  const u32x4 w = lane_words(seed, gctx, f, kSubSampler);
  const float u = u01(w.x);
  if (sp.kind == CARL_FEAT_UNIFORM_FLOAT) {
    if (sp.log_scale) {
      const float lo = logf(sp.lower), hi = logf(sp.upper);
      return expf(__fmaf_rn(hi - lo, u, lo));
    }
    return __fmaf_rn(sp.upper - sp.lower, u, sp.lower);
  }
  if (sp.kind == CARL_FEAT_UNIFORM_INT) {
    const float span = sp.upper - sp.lower + 1.0f;
    return sp.lower + fminf(floorf(u * span), span - 1.0f);
  }
  if (sp.kind == CARL_FEAT_CATEGORICAL) {
    const int k = min((int)(u * (float)sp.n_choices), sp.n_choices - 1);
    return sp.choices[k];
  }
  // NORMAL_FLOAT: Box-Muller on (w.x, w.y); redraw with the next attempt's words while out of bounds
  float v = 0.0f;
  u32x4 ww = w;
  for (int attempt = 0; attempt < kNormalTries; ++attempt) {
    if (attempt > 0) ww = lane_words(seed, gctx, f, kSubSampler | (uint32_t)attempt);
    const float u1 = u01(ww.x), u2 = u01(ww.y);
    float sn, cs;
    sincos_fast(6.28318530717958647692f * u2, sn, cs);
    v = __fmaf_rn(sp.sigma, sqrtf(-2.0f * logf(1.0f - u1)) * cs, sp.mu);
    if (v >= sp.lower && v <= sp.upper) return v;
  }
  return fminf(fmaxf(v, sp.lower), sp.upper);
}

__global__ void __launch_bounds__(256) sample_contexts_kernel(const carl_feature_spec_t* __restrict__ specs,
                                                              int n_features, int n_contexts, int ctx_stride,
                                                              long long context_offset, uint64_t seed,
                                                              float* __restrict__ ctx_table) {
  extern __shared__ carl_feature_spec_t sp_lds[];
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(specs);
    uint32_t* dst = reinterpret_cast<uint32_t*>(sp_lds);
    const int words = n_features * (int)(sizeof(carl_feature_spec_t) / 4);
    for (int k = threadIdx.x; k < words; k += blockDim.x) dst[k] = src[k];
  }
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_contexts) return;
  const uint64_t gctx = (uint64_t)(context_offset + c);
  for (int f = 0; f < n_features; ++f)
    ctx_table[(size_t)f * ctx_stride + c] = sample_feature(sp_lds[f], seed, gctx, (uint32_t)f);
}

__global__ void __launch_bounds__(256) verify_contexts_kernel(const carl_feature_spec_t* __restrict__ specs,
                                                              int n_features, int n_contexts, int ctx_stride,
                                                              const float* __restrict__ ctx_table,
                                                              int32_t* __restrict__ n_bad) {
  extern __shared__ carl_feature_spec_t sp_lds[];
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(specs);
    uint32_t* dst = reinterpret_cast<uint32_t*>(sp_lds);
    const int words = n_features * (int)(sizeof(carl_feature_spec_t) / 4);
    for (int k = threadIdx.x; k < words; k += blockDim.x) dst[k] = src[k];
  }
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  int bad = 0;
  if (c < n_contexts)
    for (int f = 0; f < n_features; ++f) {
      const carl_feature_spec_t& sp = sp_lds[f];
      const float v = ctx_table[(size_t)f * ctx_stride + c];
      bool ok;
      if (sp.kind == CARL_FEAT_CATEGORICAL) {
        ok = false;
        for (int k = 0; k < sp.n_choices; ++k) ok |= (v == sp.choices[k]);
      } else {
        ok = (v >= sp.lower) && (v <= sp.upper);  // false for NaN
      }
      bad += ok ? 0 : 1;
    }
  // wave reduction, one atomic per wavefront
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) bad += __shfl_down(bad, off);
  if (lane_id() == 0 && bad != 0) atomicAdd(n_bad, bad);
}

}  // namespace carl
