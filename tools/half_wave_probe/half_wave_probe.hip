// half_wave_probe.hip -- does gfx950 skip the passes of a wave64 vector instruction whose upper 32 lanes are switched off?
// for batches that leave SIMDs empty (the 8-GPU shard: 8 192 lanes = one wavefront on 128 of 1 024 SIMDs)
// splitting every wavefront into two 32-active-lane wavefronts doubles the wavefronts -- it shortens the launch only if a
// half-empty wavefront issues its instructions faster.  Each kernel runs ITERS x 8 independent chains of ONE instruction
// with `active` lanes enabled (the rest leave before the loop), W wavefronts per SIMD; printed: cycles per
// wavefront-instruction at the device's clock, and the time to push 64 lanes x ITERS x 8 instructions of work through
// one SIMD in each arrangement.
//   hipcc --offload-arch=gfx950 -O3 tools/half_wave_probe/half_wave_probe.hip -o tools/half_wave_probe/half_wave_probe
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int ITERS = 4096;

#define CHAIN8(OP, A, X, Y)                                                                                   \
  asm volatile(OP " %0, %8, %9, %0\n\t" OP " %1, %8, %9, %1\n\t" OP " %2, %8, %9, %2\n\t" OP " %3, %8, %9, %3\n\t" \
               OP " %4, %8, %9, %4\n\t" OP " %5, %8, %9, %5\n\t" OP " %6, %8, %9, %6\n\t" OP " %7, %8, %9, %7"     \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7])    \
               : "v"(X), "v"(Y))

template <int OP>
__global__ void __launch_bounds__(64) probe(float* out, float xf, float yf, int active) {
  float a[8];
  double d[8];
  const double xd = xf, yd = yf;
  for (int k = 0; k < 8; ++k) { a[k] = 0.001f * (threadIdx.x + k); d[k] = 0.001 * (threadIdx.x + k) + 1.0; }
  if ((int)threadIdx.x < active) {  // the other lanes are switched off for the whole loop (exec mask)
    for (int it = 0; it < ITERS; ++it) {
      if constexpr (OP == 0) CHAIN8("v_fma_f32", a, xf, yf);
      if constexpr (OP == 1) CHAIN8("v_fma_f64", d, xd, yd);
    }
  }
  float s = 0.0f;
  for (int k = 0; k < 8; ++k) s += a[k] + (float)d[k];
  if (s == 123.456f) out[0] = s;
}

template <int OP>
double run(int waves_per_simd, int active, float* out, double clock_hz) {
  const int grid = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<OP><<<grid, 64>>>(out, 0.999f, 0.001f, active);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) probe<OP><<<grid, 64>>>(out, 0.999f, 0.001f, active);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)waves_per_simd * ITERS * 8 * 5;
  return ms * 1e-3 * clock_hz / insts_per_simd;  // cycles per wavefront-instruction per SIMD
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double hz = p.clockRate * 1e3;
  float* out;
  hipMalloc(&out, 4);
  printf("device %s, %d CUs, clock %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  printf("%-12s %-28s %10s %34s\n", "instruction", "arrangement", "cycles/inst", "cycles per 64 lane-instructions of work");
  const char* names[2] = {"v_fma_f32", "v_fma_f64"};
  for (int op = 0; op < 2; ++op) {
    struct { int w, a; } cases[] = {{1, 64}, {1, 32}, {2, 32}, {2, 64}, {4, 32}, {1, 16}, {4, 16}};
    for (auto c : cases) {
      const double cyc = op == 0 ? run<0>(c.w, c.a, out, hz) : run<1>(c.w, c.a, out, hz);
      char arr[64];
      snprintf(arr, sizeof arr, "%d wavefront(s) x %d lanes", c.w, c.a);
      printf("%-12s %-28s %10.2f %34.2f\n", names[op], arr, cyc, cyc * 64.0 / c.a);
    }
  }
  return 0;
}
