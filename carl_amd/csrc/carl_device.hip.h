// carl_device.hip.h -- device-side building blocks shared by every family kernel
// (gfx950 / CDNA4 only: wave64, no portability shims).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/carl_amd.h"

namespace carl {

constexpr int kWave = 64;

// ---- Philox4x32-10 (Salmon et al. SC'11, Random123 constants) ------------------
// Counter-based: a lane's draw is a pure function of (seed, global lane id,
// episode, sub-stream), so resets need no RNG state in HBM and results do not
// depend on which GPU owns the lane.
struct u32x4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// sub-streams of a lane's episode
constexpr uint32_t kSubInit = 0u;      // init-state draws
constexpr uint32_t kSubSelector = 1u;  // random context selector
constexpr uint32_t kSubStep0 = 2u;     // + elapsed: per-step noise

__device__ __forceinline__ u32x4 lane_words(uint64_t seed, uint64_t glane, uint32_t episode,
                                            uint32_t sub) {
  return philox4x32_10(u32x4{(uint32_t)glane, (uint32_t)(glane >> 32), episode, sub},
                       (uint32_t)seed, (uint32_t)(seed >> 32));
}

// 24-bit uniform in [0, 1)
__device__ __forceinline__ float u01(uint32_t w) {
  return (float)(w >> 8) * (1.0f / 16777216.0f);
}
// numpy's uniform(low, high) = low + (high - low) * u, as ONE fma so the CPU
// oracle can reproduce the reset state bit for bit
__device__ __forceinline__ float uniform_between(float lo, float hi, uint32_t w) {
  return __fmaf_rn(hi - lo, u01(w), lo);
}

// ---- context table access ------------------------------------------------------
// Feature-major table: the gather of feature f by a wavefront whose lanes hold
// consecutive context ids (the lane<->context identity of "N sampled contexts")
// is one coalesced 256-byte request.
struct GlobalCtx {
  const float* __restrict__ table;
  int stride;
  __device__ __forceinline__ float get(int feat, int c) const {
    return table[(size_t)feat * stride + c];
  }
};
// Small context sets (C << lanes, e.g. 100 contexts x 65 536 lanes): the whole
// [F][C] table is staged in LDS once per workgroup and lanes index it there --
// random ctx ids then cost an LDS read instead of a scattered HBM/L2 gather.
struct LdsCtx {
  const float* lds;
  int n_contexts;
  __device__ __forceinline__ float get(int feat, int c) const {
    return lds[feat * n_contexts + c];
  }
};

template <int F>
__device__ __forceinline__ void stage_ctx_table(float* lds, const carl_batch_t& b) {
  const int total = F * b.n_contexts;
  for (int k = threadIdx.x; k < total; k += blockDim.x) {
    const int f = k / b.n_contexts, c = k - f * b.n_contexts;
    lds[k] = b.ctx_table[(size_t)f * b.ctx_stride + c];
  }
  __syncthreads();
}

// ---- selector ------------------------------------------------------------------
// carl/context/selection.py: round robin :116-122, static :131-136, random :103-107
__device__ __forceinline__ int select_context(const carl_batch_t& b, int idx, uint64_t glane,
                                              uint32_t episode) {
  if (b.selector == CARL_SEL_ROUND_ROBIN) {
    int v = (int)(((int64_t)idx + b.selector_stride) % b.n_contexts);
    return v < 0 ? v + b.n_contexts : v;
  }
  if (b.selector == CARL_SEL_RANDOM) {
    const u32x4 w = lane_words(b.seed, glane, episode, kSubSelector);
    return (int)__umulhi(w.x, (uint32_t)b.n_contexts);
  }
  return idx;  // static / host
}

// ---- wave-level helpers --------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// The wave's mask of a per-lane predicate.  HIP's `__ballot(int)` widens the predicate to an int and compares it
// with zero through an opaque intrinsic, so a predicate that already IS a lane mask (the OR of two compares) comes
// out as v_cndmask 0/1 + v_cmp_ne: two vector instructions per call in a loop that is bound by the instruction
// stream of one wave.  The ballot builtin takes the i1 and costs nothing.
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// number of set bits of `mask` below this lane (v_mbcnt_lo/hi)
__device__ __forceinline__ int prefix_popc(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Append this wave's finished episodes to the compact log: ballot -> one
// atomicAdd per wavefront -> each done lane writes at base + its rank.
__device__ __forceinline__ void log_finished(const carl_batch_t& b, bool done, uint64_t glane,
                                             float ret, int len) {
  if (b.fin_count == nullptr) return;
  const unsigned long long m = ballot(done);
  if (m == 0ull) return;
  int base = 0;
  if (lane_id() == __builtin_ctzll(m)) base = atomicAdd(b.fin_count, __popcll(m));
  base = __shfl(base, __builtin_ctzll(m));
  if (done) {
    const int pos = base + prefix_popc(m);
    if (pos < b.fin_capacity) {
      b.fin_lane[pos] = (int64_t)glane;
      b.fin_return[pos] = ret;
      b.fin_length[pos] = len;
    }
  }
}

// Make every 32-bit word of `v` an operand of an (empty) asm statement: the compiler
// must have the value in a register HERE, i.e. it has to wait for any load feeding it at
// this point instead of at a later first use.
template <class T>
__device__ __forceinline__ void settle(T& v) {
  if constexpr (sizeof(T) >= 4) {
    static_assert(sizeof(T) % 4 == 0, "settle() works on 32-bit words");
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) asm volatile("" : "+v"(w[i]));
    __builtin_memcpy(&v, w, sizeof(T));
  }
}

// ---- vector stores of one lane's observation -------------------------------------
// obs is lane-major [n][D]; consecutive lanes write consecutive D*4-byte records, so
// one wide store per lane keeps the wavefront's store contiguous in HBM.
template <int D>
__device__ __forceinline__ void store_obs(float* __restrict__ dst, size_t lane, const float (&o)[D]) {
  float* p = dst + lane * D;
  if constexpr (D == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  } else if constexpr (D == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
  } else if constexpr (D == 6) {
    float2* q = reinterpret_cast<float2*>(p);
    q[0] = make_float2(o[0], o[1]);
    q[1] = make_float2(o[2], o[3]);
    q[2] = make_float2(o[4], o[5]);
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) p[d] = o[d];
  }
}

template <class T>
__device__ __forceinline__ T load_action(const void* a, int dtype, size_t k);
template <>
__device__ __forceinline__ int load_action<int>(const void* a, int dtype, size_t k) {
  return dtype == CARL_ACTION_I64 ? (int)static_cast<const long long*>(a)[k]
                                  : static_cast<const int*>(a)[k];
}
template <>
__device__ __forceinline__ float load_action<float>(const void* a, int, size_t k) {
  return static_cast<const float*>(a)[k];
}

}  // namespace carl
