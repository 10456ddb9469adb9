// v_mfma_f32_4x4x4_16B_f16 (sixteen 4 x 4 x 4 blocks): issue cost, and does it overlap with the VALU work of ANOTHER wave of the same SIMD?
// Same harness as mfma_valu_overlap.hip: 8 waves per CU, waves 0-3 matrix role, 4-7 vector role (wave w and w + 4 share a SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma4_overlap.hip -o tools/micro/bin/m4o && tools/micro/bin/m4o
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool matrix = wave < 4, vector = wave >= 4;
  if (MODE == 0 && !matrix) return;
  if (MODE == 1 && !vector) return;
  f32x4 acc[16];
  for (int j = 0; j < 16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x4 a, b;
  for (int i = 0; i < 4; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i - 1.5f); }
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = (f32x2){threadIdx.x * 1e-3f + i, 0.5f * i};
  const f32x2 c1 = {0.999f, 1.001f}, c2 = {1e-3f, -1e-3f};
  for (int it = 0; it < iters; ++it) {
    if (matrix && MODE != 1) {
#pragma unroll
      for (int j = 0; j < 32; ++j) acc[j & 15] = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, acc[j & 15], 0, 0, 0);
    }
    if (vector && MODE != 0) {
      for (int q = 0; q < 3; ++q) {
        if (KIND == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(c1.x), "v"(c2.x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].y) : "v"(c1.y), "v"(c2.y));
          }
        }
        if (KIND == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[i].x));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i].y));
          }
        }
        if (KIND == 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(v[i].y));
            asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0" : "+v"(v[i].y) : "v"(c1.x), "v"(c2.x));
          }
        }
        if (KIND == 3) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
        }
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int KIND>
void run(float* out, int iters, const char* what) {
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-64s %8.3f ms\n", what, best);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 4000;
  printf("per iteration: 32 v_mfma_f32_4x4x4_16B_f16 on 16 accumulators vs 3 x one vector group (4000 iterations)\n");
  run<0, 0>(out, iters, "matrix waves alone (128 000 MFMAs per wave)");
#define KIND(T, NAME)                                              \
  run<1, T>(out, iters, NAME ": vector waves alone");               \
  run<2, T>(out, iters, NAME ": matrix + vector, different waves of a SIMD");
  KIND(0, "16 v_fma_f32")
  KIND(1, "8 v_exp/v_rcp")
  KIND(2, "8 cvt_pk_f16 + 8 fma_mixlo")
  KIND(3, "8 v_pk_fma_f32")
  return 0;
}
