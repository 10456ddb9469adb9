// Skinny linear layer for the transformer decoder's 100-query GEMMs (reference: the nn.Linear / in_proj / MLP calls of
// mask2former_transformer_decoder.py:25-212 on [100, B, 256] tensors):  out[m, n] = act( sum_k x[m, k] W[n, k] + bias[n] ),
// M <= 128 rows.  hipBLASLt launches 16 workgroups for the 100 x 2048 x 256 FFN GEMMs (88 us each in the 9-layer
// decoder profile); this is launch/latency-bound work, so the kernel maximises parallelism instead: one workgroup per
// 16 output columns, its 8 waves split K, every wave keeps all M/16 row tiles (exact-fp32 v_mfma_f32_16x16x4_f32,
// operands read as 32 contiguous bytes per lane with the k-index permutation d = 8*slot + step), partial tiles are
// reduced through LDS in a fixed order (deterministic), bias and ReLU fused.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

typedef float f32x4_s __attribute__((ext_vector_type(4)));
constexpr int SW = 8;                       // waves per workgroup = K splits

template <int MT>
__global__ __launch_bounds__(64 * SW) void skinny_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                int M, int N, int K, int relu) {
  extern __shared__ __attribute__((aligned(16))) float red[];          // [SW][MT*16][16]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kk = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int nrow = n0 + l15 < N ? n0 + l15 : N - 1;                     // clamped (results of padded columns are dropped)
  const float* wrow = W + (int64_t)nrow * K + kk * 8;
  f32x4_s acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (f32x4_s){0.f, 0.f, 0.f, 0.f};
  const int kblocks = K / 32;
  for (int kb = wave; kb < kblocks; kb += SW) {
    const float4 a0 = *reinterpret_cast<const float4*>(wrow + kb * 32), a1 = *reinterpret_cast<const float4*>(wrow + kb * 32 + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    float4 b0[MT], b1[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = t * 16 + l15 < M ? t * 16 + l15 : M - 1;
      const float* xr = x + (int64_t)m * K + kb * 32 + kk * 8;
      b0[t] = *reinterpret_cast<const float4*>(xr);
      b1[t] = *reinterpret_cast<const float4*>(xr + 4);
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const float bv[8] = {b0[t].x, b0[t].y, b0[t].z, b0[t].w, b1[t].x, b1[t].y, b1[t].z, b1[t].w};
#pragma unroll
      for (int s = 0; s < 8; ++s) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc[t], 0, 0, 0);
    }
  }
  // lane holds partial out[m = 16 t + l15][n = n0 + 4 kk + r] in acc[t][r]
#pragma unroll
  for (int t = 0; t < MT; ++t)
    *reinterpret_cast<f32x4_s*>(red + ((wave * MT * 16) + t * 16 + l15) * 16 + kk * 4) = acc[t];
  __syncthreads();
  for (int i = threadIdx.x; i < MT * 16 * 16; i += 64 * SW) {
    const int m = i >> 4, n = i & 15;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < SW; ++w) sum += red[(w * MT * 16 + m) * 16 + n];
    if (m < M && n0 + n < N) {
      if (bias) sum += bias[n0 + n];
      if (relu) sum = fmaxf(sum, 0.f);
      out[(int64_t)m * N + n0 + n] = sum;
    }
  }
}

template <int MT>
int launch(const float* x, const float* W, const float* bias, float* out, int M, int N, int K, int relu, hipStream_t st) {
  const size_t shm = (size_t)SW * MT * 16 * 16 * sizeof(float);
  hipLaunchKernelGGL(skinny_linear_kernel<MT>, dim3((N + 15) / 16), dim3(64 * SW), shm, st, x, W, bias, out, M, N, K, relu);
  return rba_launch_status();
}

}  // namespace

extern "C" int rba_skinny_linear_f32(const float* x, const float* weight, const float* bias, float* out, int M, int N, int K,
                                     int relu, void* stream) {
  RBA_CHECK_ARG(M >= 0 && M <= 128 && N >= 0 && K >= 32 && K % 32 == 0);
  if (M == 0 || N == 0) return 0;
  RBA_CHECK_ARG(x && weight && out && (((uintptr_t)x | (uintptr_t)weight) & 15) == 0);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const int mt = (M + 15) / 16;
  switch (mt) {
    case 1: return launch<1>(x, weight, bias, out, M, N, K, relu, st);
    case 2: return launch<2>(x, weight, bias, out, M, N, K, relu, st);
    case 3: return launch<3>(x, weight, bias, out, M, N, K, relu, st);
    case 4: return launch<4>(x, weight, bias, out, M, N, K, relu, st);
    case 5: return launch<5>(x, weight, bias, out, M, N, K, relu, st);
    case 6: return launch<6>(x, weight, bias, out, M, N, K, relu, st);
    case 7: return launch<7>(x, weight, bias, out, M, N, K, relu, st);
    default: return launch<8>(x, weight, bias, out, M, N, K, relu, st);
  }
}
