// Skinny linear layer for the transformer decoder's 100-query GEMMs (reference: the nn.Linear / in_proj / MLP calls of
// mask2former_transformer_decoder.py:25-212 on [100, B, 256] tensors):  out[m, n] = act( sum_k x[m, k] W[n, k] + bias[n] ),
// M <= 128 rows.  hipBLASLt launches 16 workgroups for the 100 x 2048 x 256 FFN GEMMs (88 us each in the 9-layer
// decoder profile); this is launch/latency-bound work, so the kernel maximises parallelism instead: one workgroup per
// 16 output columns, its 8 waves split K, every wave keeps all M/16 row tiles (exact-fp32 v_mfma_f32_16x16x4_f32,
// operands read as 32 contiguous bytes per lane with the k-index permutation d = 8*slot + step), partial tiles are
// reduced through LDS in a fixed order (deterministic), bias and ReLU fused.
#include "common.h"
#include "../../include/rba_hip.h"

// tools / tests only: 1 = always the round 1-2 decomposition (one workgroup per 16 columns, all row tiles), 2 = always the per-row-tile one
RBA_KNOB(rba_skinny_variant, 0);

namespace {

typedef float f32x4_s __attribute__((ext_vector_type(4)));
constexpr int SW = 8;                       // waves per workgroup = K splits

template <int MT>
__global__ __launch_bounds__(64 * SW) void skinny_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                int M, int N, int K, int relu, const float* __restrict__ x_add, int add_cols,
                                                                int seg_n) {
  extern __shared__ __attribute__((aligned(16))) float red[];          // [SW][MT*16][16]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kk = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const bool addx = x_add != nullptr && n0 < add_cols;                  // workgroup-uniform: these 16 output columns see x + x_add
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
      if (addx) {
        const float* ar = x_add + (int64_t)m * K + kb * 32 + kk * 8;
        const float4 c0 = *reinterpret_cast<const float4*>(ar), c1 = *reinterpret_cast<const float4*>(ar + 4);
        b0[t] = make_float4(b0[t].x + c0.x, b0[t].y + c0.y, b0[t].z + c0.z, b0[t].w + c0.w);
        b1[t] = make_float4(b1[t].x + c1.x, b1[t].y + c1.y, b1[t].z + c1.z, b1[t].w + c1.w);
      }
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
      sum = rba_clamp_below(sum, rba_relu_floor(relu));
      const int nn = n0 + n;
      out[seg_n ? ((int64_t)(nn / seg_n) * M + m) * seg_n + nn % seg_n : (int64_t)m * N + nn] = sum;
    }
  }
}

// Round 3: one workgroup per (16 output columns, 16-row tile) instead of per 16 columns with all row tiles.  The 9-layer decoder of BASELINE C5
// spends 1.2 ms per image in 112 of these launches (profiles/r03_c5_kernel_trace.md): with every row tile in one workgroup a wave walks its k
// blocks in a rolled loop of 16 loads + 56 MFMAs per trip and pays the memory latency once per trip -- the FFN's second Linear (K = 2048, N = 256:
// 16 workgroups on a 256-CU chip, 8 trips per wave) took 44 us.  Here a wave's trip is 4 loads + 8 MFMAs, four trips are requested at once, and
// the launch has M/16 times the workgroups.  k blocks are assigned to waves and reduced through LDS exactly as above: bit-identical results.
__global__ __launch_bounds__(64 * SW) void skinny_linear_tile_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                     const float* __restrict__ bias, float* __restrict__ out, int M, int N, int K,
                                                                     int relu, const float* __restrict__ x_add, int add_cols, int seg_n) {
  __shared__ __attribute__((aligned(16))) float red[SW * 16 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kk = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const bool addx = x_add != nullptr && n0 < add_cols;
  const int nrow = n0 + l15 < N ? n0 + l15 : N - 1;
  const int mrow = m0 + l15 < M ? m0 + l15 : M - 1;
  const float* wrow = W + (int64_t)nrow * K + kk * 8;
  const float* xrow = x + (int64_t)mrow * K + kk * 8;
  const float* arow = addx ? x_add + (int64_t)mrow * K + kk * 8 : nullptr;
  f32x4_s acc = {0.f, 0.f, 0.f, 0.f};
  const int kblocks = K / 32;
  constexpr int U = 4;
  for (int kb = wave; kb < kblocks; kb += SW * U) {
    float4 a[U][2], b[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kb + u * SW < kblocks ? kb + u * SW : kb;                      // clamped re-read, skipped below
      a[u][0] = *reinterpret_cast<const float4*>(wrow + k * 32);
      a[u][1] = *reinterpret_cast<const float4*>(wrow + k * 32 + 4);
      b[u][0] = *reinterpret_cast<const float4*>(xrow + k * 32);
      b[u][1] = *reinterpret_cast<const float4*>(xrow + k * 32 + 4);
      if (addx) {
        const float4 c0 = *reinterpret_cast<const float4*>(arow + k * 32), c1 = *reinterpret_cast<const float4*>(arow + k * 32 + 4);
        b[u][0] = make_float4(b[u][0].x + c0.x, b[u][0].y + c0.y, b[u][0].z + c0.z, b[u][0].w + c0.w);
        b[u][1] = make_float4(b[u][1].x + c1.x, b[u][1].y + c1.y, b[u][1].z + c1.z, b[u][1].w + c1.w);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (kb + u * SW < kblocks) {                                                 // wave-uniform
        const float av[8] = {a[u][0].x, a[u][0].y, a[u][0].z, a[u][0].w, a[u][1].x, a[u][1].y, a[u][1].z, a[u][1].w};
        const float bv[8] = {b[u][0].x, b[u][0].y, b[u][0].z, b[u][0].w, b[u][1].x, b[u][1].y, b[u][1].z, b[u][1].w};
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_], bv[s_], acc, 0, 0, 0);
      }
    }
  }
  *reinterpret_cast<f32x4_s*>(red + (wave * 16 + l15) * 16 + kk * 4) = acc;
  __syncthreads();
  if (threadIdx.x < 256) {
    const int m = threadIdx.x >> 4, n = threadIdx.x & 15;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < SW; ++w) sum += red[(w * 16 + m) * 16 + n];
    if (m0 + m < M && n0 + n < N) {
      if (bias) sum += bias[n0 + n];
      sum = rba_clamp_below(sum, rba_relu_floor(relu));
      const int nn = n0 + n, mm = m0 + m;
      out[seg_n ? ((int64_t)(nn / seg_n) * M + mm) * seg_n + nn % seg_n : (int64_t)mm * N + nn] = sum;
    }
  }
}

template <int MT>
int launch(const float* x, const float* W, const float* bias, float* out, int M, int N, int K, int relu, hipStream_t st, const float* x_add,
           int add_cols, int seg_n) {
  const size_t shm = (size_t)SW * MT * 16 * 16 * sizeof(float);
  hipLaunchKernelGGL(skinny_linear_kernel<MT>, dim3((N + 15) / 16), dim3(64 * SW), shm, st, x, W, bias, out, M, N, K, relu, x_add, add_cols, seg_n);
  return rba_launch_status();
}

int skinny_impl(const float* x, const float* x_add, int add_cols, const float* weight, const float* bias, float* out, int M, int N, int K, int relu,
                int seg_n, void* stream) {
  RBA_CHECK_ARG(M >= 0 && M <= 128 && N >= 0 && K >= 32 && K % 32 == 0);
  if (M == 0 || N == 0) return 0;
  RBA_CHECK_ARG(x && weight && out && (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)x_add) & 15) == 0);
  RBA_CHECK_ARG(seg_n == 0 || (seg_n > 0 && N % seg_n == 0));
  RBA_CHECK_ARG(!x_add || (add_cols >= 0 && add_cols % 16 == 0));
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const int mt = (M + 15) / 16;
  // the per-row-tile form where a wave of the column-block form would LOOP over k blocks (K > 256: the FFN's second Linear, 32 -> 20 us per call,
  // C5 +2 % images/s, +6 % single stream); with one k block per wave the column-block form's fewer, fatter workgroups launch faster
  if (rba_skinny_variant == 0 ? K > 32 * SW : rba_skinny_variant == 2) {
    hipLaunchKernelGGL(skinny_linear_tile_kernel, dim3((N + 15) / 16, mt), dim3(64 * SW), 0, st, x, weight, bias, out, M, N, K, relu, x_add, add_cols,
                       seg_n);
    return rba_launch_status();
  }
  switch (mt) {
    case 1: return launch<1>(x, weight, bias, out, M, N, K, relu, st, x_add, add_cols, seg_n);
    case 2: return launch<2>(x, weight, bias, out, M, N, K, relu, st, x_add, add_cols, seg_n);
    case 3: return launch<3>(x, weight, bias, out, M, N, K, relu, st, x_add, add_cols, seg_n);
    case 4: return launch<4>(x, weight, bias, out, M, N, K, relu, st, x_add, add_cols, seg_n);
    case 5: return launch<5>(x, weight, bias, out, M, N, K, relu, st, x_add, add_cols, seg_n);
    case 6: return launch<6>(x, weight, bias, out, M, N, K, relu, st, x_add, add_cols, seg_n);
    case 7: return launch<7>(x, weight, bias, out, M, N, K, relu, st, x_add, add_cols, seg_n);
    default: return launch<8>(x, weight, bias, out, M, N, K, relu, st, x_add, add_cols, seg_n);
  }
}

}  // namespace

extern "C" int rba_skinny_linear_f32(const float* x, const float* weight, const float* bias, float* out, int M, int N, int K,
                                     int relu, void* stream) {
  return skinny_impl(x, nullptr, 0, weight, bias, out, M, N, K, relu, 0, stream);
}

extern "C" int rba_skinny_linear_add_f32(const float* x, const float* x_add, int add_cols, const float* weight, const float* bias, float* out,
                                         int M, int N, int K, int relu, int seg_n, void* stream) {
  return skinny_impl(x, x_add, add_cols, weight, bias, out, M, N, K, relu, seg_n, stream);
}
