// Do the f16 MFMAs honour subnormal f16 INPUTS on gfx950?  (The unscaled residual l = f16(x - h) of a small x is subnormal; rba_reduce_up4_mx_kernel
// relies on it for sigma and P in [0, 1], and any single-accumulator K6 / un-scaled P in K5 would.)  A = 2^-20 (subnormal in f16: min normal 2^-14),
// B = 2^10: every product is 2^-10, a 32x32x16 MFMA sums 16 of them = 2^-6 per output if the inputs are honoured, 0 if they are flushed.
// Also: a NORMAL product that underflows f16 but not f32 (2^-10 * 2^-10) must be exact in the fp32 accumulator.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_denorm.hip -o /tmp/mfma_denorm && /tmp/mfma_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float av, float bv) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)a[0]; }
}
int main() {
  float* o; hipMalloc(&o, 16);
  const float cases[3][2] = {{9.5367431640625e-07f /* 2^-20 */, 1024.f}, {5.9604644775390625e-08f /* 2^-24: smallest subnormal */, 1024.f}, {0.0009765625f, 0.0009765625f}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, cs[0], cs[1]);
    float h[3]; hipMemcpy(h, o, 12, hipMemcpyDeviceToHost);
    printf("a = %.10g (as f16 -> %.10g)  b = %.10g :  32x32x16 -> %.10g (16 a b = %.10g)   16x16x32 -> %.10g (32 a b = %.10g)\n", cs[0], h[2], cs[1], h[0], 16.0 * cs[0] * cs[1], h[1],
           32.0 * cs[0] * cs[1]);
  }
  return 0;
}
