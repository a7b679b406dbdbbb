// inertia_bench.hip -- north_star names "MFMA only for the dense 3x3 / 6x6 inertia-matrix contractions in the Brax
// articulated path".  The contraction is w' = R diag(1/I) R^T tau per link per substep (brax spring: world inverse
// inertia applied to the net torque; brax_kernels.hip.h apply_inv_inertia).  This standalone kernel pair measures it
// both ways on the batch shape of BASELINE config 5 (32 768 Humanoid envs x 11 links), K chained applications
// per item held in registers (as inside a substep loop: compute-bound, no HBM in the timed part):
//
//   VALU:  one lane = one link (the product kernel's mapping): two quaternion rotations + a scale, ~40 FMAs.
//   MFMA:  v_mfma_f32_4x4x1_16b_f32 -- the only MFMA shape that fits (16 independent 4x4 blocks per issue = "batched
//          small matrices").  Best layout for it: FOUR lanes per link; lane j holds row j and column j of R and
//          component j of the vectors; u = R^T tau and y = R v are three rank-1 updates each, with the broadcast
//          operand spread by DPP quad broadcasts.  (Feeding MFMA from the one-lane-per-link layout instead would
//          cost ~24 LDS transposition ops per link on top.)
//
// Build + run (on the GPU box): tools/mfma_inertia/run.sh.  Result recorded in profiles/r02_mfma_inertia.txt.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

struct v3 { float x, y, z; };
__device__ __forceinline__ v3 cross3(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ v3 qrot(float w, v3 u, v3 v) {
  v3 t = cross3(u, v); t = {2 * t.x, 2 * t.y, 2 * t.z};
  v3 c = cross3(u, t);
  return {v.x + w * t.x + c.x, v.y + w * t.y + c.y, v.z + w * t.z + c.z};
}

// one lane = one item; in: q[4][N], d[3][N], tau[3][N]; out y[3][N]
__global__ void __launch_bounds__(64) valu_kernel(const float* __restrict__ q, const float* __restrict__ d,
                                                  const float* __restrict__ tau, float* __restrict__ y, int n, int K) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float w = q[i];
  const v3 u = {q[n + i], q[2 * n + i], q[3 * n + i]}, nu = {-u.x, -u.y, -u.z};
  const v3 dd = {d[i], d[n + i], d[2 * n + i]};
  v3 t = {tau[i], tau[n + i], tau[2 * n + i]};
  for (int k = 0; k < K; ++k) {
    const v3 l = qrot(w, nu, t);                       // R^T t
    const v3 s = {l.x * dd.x, l.y * dd.y, l.z * dd.z};  // diag
    t = qrot(w, u, s);                                  // R s   (chained: the next application's torque)
  }
  y[i] = t.x; y[n + i] = t.y; y[2 * n + i] = t.z;
}

// four lanes = one item (lane j of the quad: row j / column j of R, component j); 16 items per wavefront
__global__ void __launch_bounds__(64) mfma_kernel(const float* __restrict__ q, const float* __restrict__ d,
                                                  const float* __restrict__ tau, float* __restrict__ y, int n, int K) {
  const int item = blockIdx.x * 16 + threadIdx.x / 4;
  const int j = threadIdx.x & 3;
  const int ii = item < n ? item : n - 1;
  const float w = q[ii], x = q[n + ii], yy = q[2 * n + ii], z = q[3 * n + ii];
  // rotation matrix of the unit quaternion
  const float R[3][3] = {{1 - 2 * (yy * yy + z * z), 2 * (x * yy - w * z), 2 * (x * z + w * yy)},
                         {2 * (x * yy + w * z), 1 - 2 * (x * x + z * z), 2 * (yy * z - w * x)},
                         {2 * (x * z - w * yy), 2 * (yy * z + w * x), 1 - 2 * (x * x + yy * yy)}};
  float row[3], col[3];  // lane j: R[j][:] and R[:][j]  (lane 3 of the quad: zeros)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    row[k] = j == 0 ? R[0][k] : j == 1 ? R[1][k] : j == 2 ? R[2][k] : 0.0f;
    col[k] = j == 0 ? R[k][0] : j == 1 ? R[k][1] : j == 2 ? R[k][2] : 0.0f;
  }
  const float dj = j < 3 ? d[j * n + ii] : 0.0f;
  float tj = j < 3 ? tau[j * n + ii] : 0.0f;
  for (int k = 0; k < K; ++k) {
    // u_j' = sum_k t_k R[k][j']:  rank-1 updates A = t_k (broadcast over the quad), B[j'] = R[k][j'] = col_{j'}[k]
    v4f acc = {0, 0, 0, 0};
    const float t0 = __shfl(tj, (threadIdx.x & ~3) + 0), t1 = __shfl(tj, (threadIdx.x & ~3) + 1),
                t2 = __shfl(tj, (threadIdx.x & ~3) + 2);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(t0, col[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(t1, col[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(t2, col[2], acc, 0, 0, 0);
    const float vj = acc.x * dj;  // every row of the block holds u_j' in lane j'
    // y_i = sum_k R[i][k] v_k:  A[i] = R[i][k] = row_i[k], B = v_k (broadcast)
    const float v0 = __shfl(vj, (threadIdx.x & ~3) + 0), v1 = __shfl(vj, (threadIdx.x & ~3) + 1),
                v2 = __shfl(vj, (threadIdx.x & ~3) + 2);
    v4f out = {0, 0, 0, 0};
    out = __builtin_amdgcn_mfma_f32_4x4x1f32(row[0], v0, out, 0, 0, 0);
    out = __builtin_amdgcn_mfma_f32_4x4x1f32(row[1], v1, out, 0, 0, 0);
    out = __builtin_amdgcn_mfma_f32_4x4x1f32(row[2], v2, out, 0, 0, 0);
    tj = j == 0 ? out.x : j == 1 ? out.y : j == 2 ? out.z : 0.0f;  // lane j keeps y_j
  }
  if (item < n && j < 3) y[j * n + item] = tj;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = 32768 * 11, K = argc > 1 ? atoi(argv[1]) : 200;
  std::vector<float> q(4 * n), d(3 * n), t(3 * n);
  srand(1);
  auto rnd = [] { return (float)(rand() % 2000001) / 2000000.0f * 2.0f - 1.0f; };
  for (int i = 0; i < n; ++i) {
    float a = rnd(), b = rnd(), c = rnd(), e = rnd(), nn = std::sqrt(a * a + b * b + c * c + e * e);
    q[i] = a / nn; q[n + i] = b / nn; q[2 * n + i] = c / nn; q[3 * n + i] = e / nn;
    for (int k = 0; k < 3; ++k) { d[k * n + i] = 0.9f + 0.1f * rnd(); t[k * n + i] = rnd(); }  // |eig| <= 1: chain stays bounded
  }
  float *dq, *dd, *dt, *y1, *y2;
  CK(hipMalloc(&dq, 4 * n * 4)); CK(hipMalloc(&dd, 3 * n * 4)); CK(hipMalloc(&dt, 3 * n * 4));
  CK(hipMalloc(&y1, 3 * n * 4)); CK(hipMalloc(&y2, 3 * n * 4));
  CK(hipMemcpy(dq, q.data(), 4 * n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dd, d.data(), 3 * n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dt, t.data(), 3 * n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_valu = 0, ms_mfma = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(valu_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, dq, dd, dt, y1, n, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_valu, e0, e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(mfma_kernel, dim3((n + 15) / 16), dim3(64), 0, 0, dq, dd, dt, y2, n, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_mfma, e0, e1));
  }
  std::vector<float> h1(3 * n), h2(3 * n);
  CK(hipMemcpy(h1.data(), y1, 3 * n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), y2, 3 * n * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int i = 0; i < 3 * n; ++i) worst = std::fmax(worst, std::fabs((double)h1[i] - h2[i]) / (1e-3 + std::fabs((double)h1[i])));
  const double apps = (double)n * K;
  printf("items %d (32768 envs x 11 links), %d chained applications of R diag(1/I) R^T per item\n", n, K);
  printf("VALU, one lane per link            : %8.3f ms per launch  %.3e applications/s\n", ms_valu / 10, apps / (ms_valu / 10 * 1e-3));
  printf("MFMA 4x4x1_16b, four lanes per link: %8.3f ms per launch  %.3e applications/s   (%.2fx the VALU time)\n",
         ms_mfma / 10, apps / (ms_mfma / 10 * 1e-3), ms_mfma / ms_valu);
  printf("max relative difference between the two after %d chained applications: %.2e\n", K, worst);
  return 0;
}
