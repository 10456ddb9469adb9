// Do VALU instructions of one wave overlap with the MFMAs of ANOTHER wave of the same SIMD (and of the same wave) on gfx950?
// One workgroup of 8 waves per CU: waves 0-3 = "matrix" role, waves 4-7 = "vector" role (wave w and w + 4 share a SIMD).
//   mode 0: matrix waves only   mode 1: vector waves only   mode 2: both roles, different waves   mode 3: both roles in the SAME wave (4 waves)
// vector work = packed fp32 FMAs (+ optional transcendentals), matrix work = v_mfma_f32_32x32x16_f16 on 8 independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int TRANS>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* t, int iters, int vper) {
  const int wave = threadIdx.x >> 6;
  const bool matrix = MODE == 3 ? true : wave < 4, vector = MODE == 3 ? true : wave >= 4;
  if (MODE == 3 && wave >= 4) return;
  if (MODE == 0 && !matrix) return;
  if (MODE == 1 && !vector) return;
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i - 3.f); }
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = (f32x2){threadIdx.x * 1e-3f + i, 0.5f * i};
  const f32x2 c1 = {0.999f, 1.001f}, c2 = {1e-3f, -1e-3f};
  const unsigned long long cy0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (matrix && MODE != 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    if (vector && MODE != 0) {
      for (int q = 0; q < vper; ++q) {
        if (TRANS == 0 || TRANS == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = v[i] * c1 + c2;                       // 8 v_pk_fma_f32
        }
        if (TRANS == 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i) v[i].x = __builtin_amdgcn_rcpf(v[i].x) + __builtin_amdgcn_exp2f(v[i].y);
        }
        if (TRANS == 2) {                                                         // 16 scalar v_fma_f32 (the same flops, not packed)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(c1.x), "v"(c2.x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].y) : "v"(c1.y), "v"(c2.y));
          }
        }
        if (TRANS == 3) {                                                         // 8 transcendentals
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[i].x));
            asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i].y));
          }
        }
        if (TRANS == 4) {                                                         // 16 integer ops
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i].x) : "v"(c1.x));
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i].y) : "v"(c1.y));
          }
        }
        if (TRANS == 5) {                                                         // 16 v_mul_f32
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(c1.x));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i].y) : "v"(c1.y));
          }
        }
        if (TRANS == 6) {                                                         // 8 v_pk_mul_f32
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
        }
        if (TRANS == 7) {                                                         // 8 v_cvt_pk_f16_f32 + 8 v_fma_mixlo
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(v[i].y));
            asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0" : "+v"(v[i].y) : "v"(c1.x), "v"(c2.x));
          }
        }
      }
    }
  }
  const unsigned long long cy1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { t[2 * (blockIdx.x * 8 + wave)] = cy1 - cy0; t[2 * (blockIdx.x * 8 + wave) + 1] = w1 - w0; }
}

template <int MODE, int TRANS>
void run(float* out, unsigned long long* t, int iters, int vper, const char* what) {
  const int nb = 256;
  static unsigned long long h[2 * 8 * 256];
  float best = 1e9f; double cyc = 0, wall = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(t, 0, nb * 8 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, TRANS>), dim3(nb), dim3(512), 0, 0, out, t, iters, vper);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      hipMemcpy(h, t, nb * 8 * 16, hipMemcpyDeviceToHost);
      cyc = wall = 0; int n = 0;
      for (int i = 0; i < nb * 8; ++i) if (h[2 * i + 1]) { cyc += h[2 * i]; wall += h[2 * i + 1]; ++n; }
      cyc /= n; wall /= n;
    }
  }
  printf("%-58s %8.3f ms   shader clock %.2f GHz (cycles %.0f over %.1f us)\n", what, best, cyc / (wall * 10.0), cyc, wall / 100.0);
}

int main() {
  float* out; unsigned long long* t;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 8 * 16);
  const int iters = 4000;
  for (int vper : {3}) {
    printf("---- per iteration: 8 MFMA (256 clk of matrix pipe) vs %d x one vector group\n", vper);
    run<0, 0>(out, t, iters, vper, "matrix waves alone");
#define KIND(T, NAME)                                                                  \
    run<1, T>(out, t, iters, vper, NAME ": vector waves alone");                        \
    run<2, T>(out, t, iters, vper, NAME ": matrix + vector, different waves of a SIMD"); \
    run<3, T>(out, t, iters, vper, NAME ": matrix + vector, same wave");
    KIND(0, "8 v_pk_fma_f32")
    KIND(2, "16 v_fma_f32")
    KIND(5, "16 v_mul_f32")
    KIND(6, "8 v_pk_mul_f32")
    KIND(3, "8 v_exp/v_rcp")
    KIND(4, "16 int add/xor")
    KIND(7, "8 cvt_pk_f16 + 8 fma_mixlo")
  }
  return 0;
}
