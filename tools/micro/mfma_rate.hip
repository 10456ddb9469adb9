// Issue rate of the gfx950 32x32x16 MFMAs (f16 vs bf16), one to two waves per SIMD: cycles per MFMA from s_memtime, clock from
// s_memrealtime (100 MHz).   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* t, int iters) {
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a, b; bf16x8 c, d;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i - 3.f); c[i] = (__bf16)(float)a[i]; d[i] = (__bf16)(float)b[i]; }
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, d, acc[j], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
  float* out; unsigned long long* t; const int blocks = 512, iters = 4000;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&t, blocks * 16);
  unsigned long long h[2 * 512];
  for (int kind = 0; kind < 2; ++kind)
    for (int nb : {256, 512}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, out, t, iters);
        else hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, out, t, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, t, nb * 16, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0; for (int i = 0; i < nb; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
        cyc /= nb; wall /= nb;
        const double per_simd = (double)iters * 8 * (nb / 256);      // MFMAs per SIMD
        printf("%s %d WGs (%d wave/SIMD): %.3f ms, s_memtime %.0f ticks, wall %.1f us (100 MHz) -> %.1f us per MFMA-slot x1e-3, event-time cycles/MFMA at 2.4 GHz %.1f, TFLOP/s %.0f\n",
               kind ? "bf16" : "f16 ", nb, nb / 256, ms, cyc, wall / 100.0, 0.0, ms * 1e-3 * 2.4e9 / per_simd,
               2.0 * 32 * 32 * 16 * (double)iters * 8 * nb * 4 / (ms * 1e-3) / 1e12);
      }
    }
  return 0;
}
