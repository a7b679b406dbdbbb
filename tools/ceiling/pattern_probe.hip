// pattern_probe.hip -- round 4: which part of the rollout's traffic pattern costs the HBM rate?
// Physics-free kernels with CartPole's shapes (per workgroup-step: 1 KiB actions read; 4 KiB obs + 1 KiB reward + 2 x 256 B
// flags written; 16-byte non-temporal accesses; 256 workgroups x 256 lanes; chunks of 8 steps like the staged kernel's
// storers), selected by `mode`:
//   0 all five streams         1 no reader              2 no flag rows (reader + obs + reward)
//   3 obs only (one stream)    4 reader + obs only      5 all five, temporal stores
//   6 plain fill of the obs buffer (grid-stride, 1 KiB per wave-instruction): the part's write peak
//   7 reader alone (8 rows in flight, wait, next 8)      8 reader alone, software-pipelined (next 8 issued before the wait)
//   9 all five streams with the pipelined reader
//   13 / 14 / 15 / 16 all five streams, reader's loads with cache policy: none / sc0 / sc1 / sc0 sc1 (nt is mode 0)
//   10 / 11 / 12 all five streams, reads in BURSTS of 32 / 64 / 128 rows, one burst ahead of the writers (the reader
//      polls the writers' progress in LDS and sleeps in between): fewer read / write turnarounds in the DRAM
// Built and timed by tools/ceiling/run_pattern_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float vf4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void put(vf4 v, void* p) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(p));
  else *reinterpret_cast<vf4*>(p) = v;
}

extern "C" __global__ void __launch_bounds__(512) probe_kernel(const int* __restrict__ act, float* __restrict__ obs,
                                                                float* __restrict__ rew, uint8_t* __restrict__ term,
                                                                uint8_t* __restrict__ trunc, float* __restrict__ sink, int n,
                                                                int T, int mode) {
  const int wave = threadIdx.x / 64, l = threadIdx.x % 64;
  const size_t lane_base = (size_t)blockIdx.x * 256;
  if (mode == 6) {  // fill: the obs buffer (16 B x lanes x T), grid-stride
    const size_t total16 = (size_t)T * n;
    const vf4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < total16; i += (size_t)gridDim.x * 512)
      __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(obs) + i);
    return;
  }
  const bool reader_on = mode == 0 || mode == 2 || mode == 4 || mode == 5 || mode >= 7;
  const bool obs_on = mode < 7 || mode >= 9, rew_on = mode == 0 || mode == 1 || mode == 2 || mode == 5 || mode >= 9;  // (17, 18: all five)
  const bool flags_on = mode == 0 || mode == 1 || mode == 5 || mode >= 9;
  __shared__ int progress;  // steps the writers have issued (storer 0 publishes per chunk)
  if (threadIdx.x == 0) progress = 0;
  __syncthreads();
  if (wave == 4) {  // reader (the loader wave): 8 rows in flight
    if (!reader_on) return;
    int acc = 0;
    if (mode == 19 || mode == 20) {  // 19: each workgroup re-reads ITS OWN first 8 rows (8 KiB, cacheable loads: L2 hits, spread over the channels); 20: the same with nt loads
      typedef int vi4 __attribute__((ext_vector_type(4)));
      for (int t0 = 0; t0 < T; t0 += 8) {
        vi4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const vi4* p = reinterpret_cast<const vi4*>(act + (size_t)u * n + lane_base) + l;
          r[u] = mode == 19 ? *p : __builtin_nontemporal_load(p);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += r[u].x + r[u].w;
        asm volatile("" ::: "memory");
      }
      if (acc == 123456789) sink[0] = (float)acc;
      return;
    }
    if (mode == 17 || mode == 18) {  // where do the reads cost?  17: every workgroup reads workgroup 0's piece of each row (L2 hits
                                     // after the first); 18: the same 8 rows over and over (L1 / L2 hits): no HBM reads at all
      typedef int vi4 __attribute__((ext_vector_type(4)));
      for (int t0 = 0; t0 < T; t0 += 8) {
        vi4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          r[u] = __builtin_nontemporal_load(reinterpret_cast<const vi4*>(act + (size_t)(mode == 18 ? u : min(t0 + u, T - 1)) * n) + l);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += r[u].x + r[u].w;
      }
      if (acc == 123456789) sink[0] = (float)acc;
      return;
    }
    if (mode >= 13) {
      typedef int vi4 __attribute__((ext_vector_type(4)));
      for (int t0 = 0; t0 < T; t0 += 8) {
        vi4 r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const vi4* p = reinterpret_cast<const vi4*>(act + (size_t)min(t0 + u, T - 1) * n + lane_base) + l;
          if (mode == 13) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[u]) : "v"(p) : "memory");
          else if (mode == 14) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(r[u]) : "v"(p) : "memory");
          else if (mode == 15) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r[u]) : "v"(p) : "memory");
          else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r[u]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += r[u].x;
      }
      if (acc == 123456789) sink[0] = (float)acc;
      return;
    }
    if (mode >= 10) {  // bursts of B rows, prefetched one burst ahead of the writers
      typedef int vi4 __attribute__((ext_vector_type(4)));
      const int B = mode == 10 ? 32 : mode == 11 ? 64 : 128;
      for (int tb = 0; tb < T; tb += B) {
        while (__hip_atomic_load(&progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < tb - B) __builtin_amdgcn_s_sleep(32);
        for (int t0 = tb; t0 < tb + B && t0 < T; t0 += 16) {
          vi4 r[16];
#pragma unroll
          for (int u = 0; u < 16; ++u)
            r[u] = __builtin_nontemporal_load(reinterpret_cast<const vi4*>(act + (size_t)min(t0 + u, T - 1) * n + lane_base) + l);
#pragma unroll
          for (int u = 0; u < 16; ++u) acc += r[u].x;
        }
      }
      if (acc == 123456789) sink[0] = (float)acc;
      return;
    }
    if (mode >= 8) {  // pipelined: the loads of batch k + 1 are in flight while batch k is consumed
      typedef int vi4 __attribute__((ext_vector_type(4)));
      vi4 a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7;
      auto ld = [&](int t) { return __builtin_nontemporal_load(reinterpret_cast<const vi4*>(act + (size_t)min(t, T - 1) * n + lane_base) + l); };
      a0 = ld(0); a1 = ld(1); a2 = ld(2); a3 = ld(3); a4 = ld(4); a5 = ld(5); a6 = ld(6); a7 = ld(7);
      for (int t0 = 0; t0 < T; t0 += 16) {
        b0 = ld(t0 + 8); b1 = ld(t0 + 9); b2 = ld(t0 + 10); b3 = ld(t0 + 11); b4 = ld(t0 + 12); b5 = ld(t0 + 13); b6 = ld(t0 + 14); b7 = ld(t0 + 15);
        acc += a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x;
        asm volatile("" ::: "memory");
        a0 = ld(t0 + 16); a1 = ld(t0 + 17); a2 = ld(t0 + 18); a3 = ld(t0 + 19); a4 = ld(t0 + 20); a5 = ld(t0 + 21); a6 = ld(t0 + 22); a7 = ld(t0 + 23);
        acc += b0.x + b1.x + b2.x + b3.x + b4.x + b5.x + b6.x + b7.x;
        asm volatile("" ::: "memory");
      }
      if (acc == 123456789) sink[0] = (float)acc;
      return;
    }
    for (int t0 = 0; t0 < T; t0 += 8) {
      typedef int vi4 __attribute__((ext_vector_type(4)));
      vi4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        r[u] = __builtin_nontemporal_load(reinterpret_cast<const vi4*>(act + (size_t)min(t0 + u, T - 1) * n + lane_base) + l);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += r[u].x + r[u].w;
    }
    if (acc == 123456789) sink[0] = (float)acc;
    return;
  }
  if (wave < 5 || !obs_on) return;  // waves 0..3 stand in for the compute waves: idle here
  const int which = wave - 5;  // three storers: every third step of a chunk, like drain_records
  for (int t0 = 0; t0 < T; t0 += 8) {
    if (which == 0 && l == 0) __hip_atomic_store(&progress, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int u = which; u < 8 && t0 + u < T; u += 3) {
      const size_t row = (size_t)(t0 + u) * n + lane_base;
      const float f = (float)(t0 + u + l);
      const vf4 v = {f, f + 1.0f, f + 2.0f, f + 3.0f};
      char* g_obs = reinterpret_cast<char*>(obs + row * 4);
      if (mode == 5) {
        if (obs_on) for (int off = 0; off < 4096; off += 1024) put<false>(v, g_obs + off + 16 * l);
        if (rew_on) put<false>(v, reinterpret_cast<char*>(rew + row) + 16 * l);
        if (flags_on && l < 32) put<false>(v, (l < 16) ? (void*)(term + row + 16 * l) : (void*)(trunc + row + 16 * (l - 16)));
      } else {
        if (obs_on) for (int off = 0; off < 4096; off += 1024) put<true>(v, g_obs + off + 16 * l);
        if (rew_on) put<true>(v, reinterpret_cast<char*>(rew + row) + 16 * l);
        if (flags_on && l < 32) put<true>(v, (l < 16) ? (void*)(term + row + 16 * l) : (void*)(trunc + row + 16 * (l - 16)));
      }
    }
  }
}

extern "C" int launch_probe(const int* act, float* obs, float* rew, uint8_t* term, uint8_t* trunc, float* sink, int n, int T,
                            int mode, void* stream) {
  const int grid = mode == 6 ? 2048 : n / 256;
  hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, act, obs, rew, term, trunc, sink, n, T, mode);
  return (int)hipGetLastError();
}
