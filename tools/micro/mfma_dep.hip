// Cost of accumulator dependences between consecutive v_mfma_f32_32x32x16_f16 on gfx950, whole chip, one wave per SIMD:
//   0: eight accumulators round-robin   1: ONE accumulator (every MFMA depends on the previous one)   2: two accumulators alternating
//   3: the f16x3 order of one column tile: m l m l l l on (main, low), four tiles
//   4: the same 24 MFMAs ordered so that no two neighbours share an accumulator
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_dep.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, ACC, 0, 0, 0)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t, int iters) {
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a, b, c, d;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i - 3.f); c[i] = a[i] * (_Float16)0.5f; d[i] = b[i] + (_Float16)1.f; }
  __syncthreads();
  const unsigned long long w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) MF(acc[j], a, b);
    } else if (KIND == 1) {
#pragma unroll
      for (int r = 0; r < 24; ++r) MF(acc[0], a, b);
    } else if (KIND == 2) {
#pragma unroll
      for (int r = 0; r < 12; ++r) { MF(acc[0], a, b); MF(acc[1], c, d); }
    } else if (KIND == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        MF(acc[2 * j], a, b); MF(acc[2 * j + 1], a, d); MF(acc[2 * j], c, b);
        MF(acc[2 * j + 1], c, d); MF(acc[2 * j + 1], a, b); MF(acc[2 * j + 1], c, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        MF(acc[2 * j], a, b); MF(acc[2 * j + 1], a, d); MF(acc[2 * j + 2], a, b); MF(acc[2 * j + 3], a, d);
        MF(acc[2 * j], c, b); MF(acc[2 * j + 1], c, d); MF(acc[2 * j + 2], c, b); MF(acc[2 * j + 3], c, d);
        MF(acc[2 * j + 1], a, b); MF(acc[2 * j + 3], a, b); MF(acc[2 * j + 1], c, b); MF(acc[2 * j + 3], c, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = w1 - w0;
}
template <int KIND>
void run(float* out, unsigned long long* t, int nb, int iters) {
  unsigned long long h[512];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<KIND>, dim3(nb), dim3(256), 0, 0, out, t, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, t, nb * 8, hipMemcpyDeviceToHost);
  double wall = 0; for (int i = 0; i < nb; ++i) wall += h[i];
  wall /= nb;                                                           // 10 ns ticks
  const double ns = wall * 10.0 / ((double)iters * 24 * (nb > 256 ? nb / 256.0 : 1.0));
  printf("kind %d, %d WGs: %.2f ns per MFMA per SIMD = %.1f clk at 2.04 GHz\n", KIND, nb, ns, ns * 2.04);
}
int main() {
  float* out; unsigned long long* t;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&t, 512 * 8);
  for (int nb : {256, 512, 8}) {
    run<0>(out, t, nb, 2000); run<1>(out, t, nb, 2000); run<2>(out, t, nb, 2000); run<3>(out, t, nb, 2000); run<4>(out, t, nb, 2000);
  }
  return 0;
}
