// How many workgroups of a given shape does a CU of gfx950 hold at once?  Each workgroup stamps the 100 MHz wall clock, spins ~6 us and stamps again;
// the host counts the workgroups alive at the launch's mid-life.  Shapes: threads per workgroup, dynamic LDS bytes, VGPRs (forced through an asm clobber), scratch.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/occ_probe tools/micro/occ_probe.hip && /tmp/occ_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

template <int THREADS, int VG, bool SCRATCH>
__global__ __launch_bounds__(THREADS) void spin(unsigned long long* t, int spin_ticks) {
  extern __shared__ float lds[];
  if (VG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  if (VG == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
  if (VG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
  if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  volatile int spill[4];
  if (SCRATCH) { spill[threadIdx.x & 3] = threadIdx.x; }
  lds[threadIdx.x] = 1.0f;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    t[blockIdx.x * 3] = t0;
    t[blockIdx.x * 3 + 1] = wall_clock64();
    t[blockIdx.x * 3 + 2] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
  }
  if (SCRATCH && spill[threadIdx.x & 3] == -1) t[0] = 0;
}

template <int THREADS, int VG, bool SCRATCH>
void run(int lds_bytes, int nwg) {
  unsigned long long* d;
  hipMalloc(&d, nwg * 24);
  hipMemset(d, 0, nwg * 24);
  hipFuncSetAttribute((const void*)spin<THREADS, VG, SCRATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  int occ = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin<THREADS, VG, SCRATCH>, THREADS, lds_bytes);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((spin<THREADS, VG, SCRATCH>), dim3(nwg), dim3(THREADS), lds_bytes, 0, d, 600);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(nwg * 3);
  hipMemcpy(h.data(), d, nwg * 24, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, h[i * 3]); t1 = std::max(t1, h[i * 3 + 1]); }
  // alive 3 us after the first start
  int alive = 0;
  for (int i = 0; i < nwg; ++i) alive += (h[i * 3] <= t0 + 300 && h[i * 3 + 1] > t0 + 300);
  printf("threads %4d lds %6d vgpr %3d scratch %d: API occupancy %d per CU; %4d workgroups alive 3 us in (= %.2f per CU); span %.1f us for %d workgroups\n", THREADS,
         lds_bytes, VG, (int)SCRATCH, occ, alive, alive / 256.0, (t1 - t0) / 100.0, nwg);
  hipFree(d);
}

int main() {
  const int n = 2048;
  run<576, 96, false>(38016, n);
  run<576, 96, true>(38016, n);
  run<576, 96, false>(4096, n);
  run<576, 80, false>(38016, n);
  run<576, 64, false>(38016, n);
  run<512, 96, false>(38016, n);
  run<512, 128, false>(38016, n);
  run<512, 96, true>(38016, n);
  run<256, 96, false>(38016, n);
  run<320, 96, false>(38016, n);
  run<192, 96, false>(38016, n);
  run<576, 96, false>(65536, n);
  run<576, 64, false>(20000, n);
  return 0;
}
