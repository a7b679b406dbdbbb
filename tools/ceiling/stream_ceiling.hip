// stream_ceiling.hip -- how fast can the rollout's HBM traffic pattern go with NO physics?
// Same launch geometry and the same five streams as rollout_staged_kernel<Pendulum>: per step a
// workgroup (256 lanes) reads 1 KiB of actions and writes 3 KiB obs + 1 KiB reward + 256 B
// terminated + 256 B truncated, with 16-byte non-temporal accesses.  A reader wave keeps 8 action rows
// in flight; the writer waves store values computed from the loop index, so no wave ever waits on
// memory it does not have to: what is left is the memory system's throughput for this pattern.
// Built and timed by tools/ceiling/run_ceiling.py; the result is the practical ceiling quoted in
// DESIGN.md section 4 for the 18 % read / 82 % write mix.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float vf4 __attribute__((ext_vector_type(4)));
#ifndef LPB
#define LPB 256  // lanes per workgroup
#endif

extern "C" __global__ void __launch_bounds__(384) stream_kernel(const float* __restrict__ act, float* __restrict__ obs,
                                                                 float* __restrict__ rew, uint8_t* __restrict__ term,
                                                                 uint8_t* __restrict__ trunc, float* __restrict__ sink,
                                                                 int n, int T) {
  const int wave = threadIdx.x / 64, l = threadIdx.x % 64;
  const size_t lane_base = (size_t)blockIdx.x * LPB;
  if (l >= LPB / 4) return;  // a row piece of LPB floats = LPB / 4 sixteen-byte accesses
  if (wave == 5) {  // reader: 8 rows in flight, like the loader wave; the sum keeps the loads alive
    vf4 acc = {0, 0, 0, 0};
    for (int t0 = 0; t0 < T; t0 += 8) {
      vf4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#ifdef BLOCK_MAJOR
        r[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(act + ((size_t)blockIdx.x * T + t0 + u) * LPB) + l);
#else
        r[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(act + (size_t)(t0 + u) * n + lane_base) + l);
#endif
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += r[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;  // never true in practice
    return;
  }
  // writers never wait on memory: waves 0..2 obs rows, wave 3 reward, wave 4 flags (the storer split)
  for (int t = 0; t < T; ++t) {
#ifdef BLOCK_MAJOR  // [n/256][T][256] instead of [T][n]: a workgroup's records are one contiguous stream
    const size_t row = ((size_t)blockIdx.x * T + t) * LPB;
#else
    const size_t row = (size_t)t * n + lane_base;
#endif
    const float f = (float)(t + l);
    const vf4 v = {f, f + 1.0f, f + 2.0f, f + 3.0f};
    if (wave < 3) {
      __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(reinterpret_cast<char*>(obs + row * 3) + 4 * LPB * wave) + l);
    } else if (wave == 3) {
      __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(rew + row) + l);
    } else {
      if (l < LPB / 16) __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(term + row) + l);
      else if (l < LPB / 8) __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(trunc + row) + (l - LPB / 16));
    }
  }
}

extern "C" int launch_stream(const float* act, float* obs, float* rew, uint8_t* term, uint8_t* trunc, float* sink, int n,
                             int T, void* stream) {
  hipLaunchKernelGGL(stream_kernel, dim3(n / LPB), dim3(384), 0, (hipStream_t)stream, act, obs, rew, term, trunc, sink, n,
                     T);
  return (int)hipGetLastError();
}
