// Where does the hardware put the workgroups of a launch that fits TWO workgroups per compute unit?
// (round 4: the order of the families inside the heterogeneous pair launch, engine_kernels.hip.h, depends on it)
//   hipcc --offload-arch=gfx950 -O2 tools/wg_placement/wg_placement.hip -o tools/wg_placement/wg_placement
//   tools/wg_placement/wg_placement [n_workgroups=512] [threads=512] [lds_bytes=77824]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    lds[0] = (char)hw;
  }
  // stay resident long enough for the whole grid to be placed
  long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (lds[0] == 77 && threadIdx.x == 1) out[0] = 0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512, threads = argc > 2 ? atoi(argv[2]) : 512;
  const int lds = argc > 3 ? atoi(argv[3]) : 77824;
  unsigned* d;
  hipMalloc(&d, sizeof(unsigned) * 2 * n);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(probe, dim3(n), dim3(threads), lds, 0, d, 2000000);
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * n, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  std::map<unsigned, std::vector<int>> by_cu;
  for (int i = 0; i < n; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(i);
  }
  printf("%d workgroups x %d threads, %d B LDS -> %zu distinct compute units\n", n, threads, lds, by_cu.size());
  int shown = 0;
  for (auto& kv : by_cu) {
    if (shown++ < 24 || shown > (int)by_cu.size() - 4) {
      printf("xcc %u se %u sh %u cu %2u :", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
      for (int i : kv.second) printf(" %d", i);
      printf("\n");
    }
  }
  // summary: how often do workgroups i and j share a CU, by (j - i)
  std::map<int, int> delta;
  for (auto& kv : by_cu)
    for (size_t a = 0; a < kv.second.size(); ++a)
      for (size_t b = a + 1; b < kv.second.size(); ++b) delta[kv.second[b] - kv.second[a]]++;
  printf("id distance between workgroups sharing a compute unit: ");
  for (auto& kv : delta) printf("%d x%d  ", kv.first, kv.second);
  printf("\nhalves: workgroups sharing a CU with one from the other half of the grid: ");
  int cross = 0, same = 0;
  for (auto& kv : by_cu)
    for (size_t a = 0; a < kv.second.size(); ++a)
      for (size_t b = a + 1; b < kv.second.size(); ++b) ((kv.second[a] < n / 2) != (kv.second[b] < n / 2) ? cross : same)++;
  printf("cross %d, same %d\n", cross, same);
  return 0;
}
