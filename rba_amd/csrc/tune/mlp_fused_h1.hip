// EXPERIMENT (tools only, librba_tune.so; measured in profiles/r04_mlp_one_accumulator.txt and NOT adopted -- result at the end of this comment).
// Fused Swin MLP for C = 128 on ONE accumulator per output (round 4, late): out = residual + fc2(GELU(fc1(y))) like mlp_fused_h3.h, with the f16x3 products
// summed into a single fp32 accumulator instead of a (main, 2^-11 low) pair -- which frees 96 of the kernel's 356 registers, so that TWO workgroups share a CU
// (two waves per SIMD: one wave's GELU arithmetic runs while the other's MFMAs do) where mlp_fused_h3_kernel leaves every SIMD a single wave.
// (reference: Mlp and the `x = x + mlp(norm2(x))` of backbone/swin.py:35-41, 293.)
//
// Arithmetic.  x w = h_x h_w + h_x l_w + l_x h_w + O(2^-22 |x w|) with UN-scaled residuals l = f16(v - f16(v)).  An un-scaled residual of a value below 2^-3 is a
// subnormal f16 (the matrix pipe honours subnormal inputs exactly: profiles/r04_mfma_denorm_probe.txt), so its ABSOLUTE error is bounded by 2^-25 whatever the value:
//   * weights: packed once per weight load as h = f16(s w), l = f16(s w - h) with ONE power-of-two scale s per matrix that puts max |w| just below 2^14
//     (rba_split_weight_f16x2_scaled): every weight down to 2^-16 of the largest keeps its 22 bits, smaller ones are off by at most 2^-39 max |w|; the accumulator holds
//     s x w and the epilogue multiplies by 1 / s (exact);
//   * activations (LayerNorm outputs, GELU outputs -- O(1) tensors): |x| >= 2^-3 keeps 22 bits, below that the error is at most 2^-25 ~ 3e-8 absolute per element,
//     i.e. <= 3e-8 |w| sqrt(K) ~ 1e-8 on an output of order one: below the fp32 accumulation rounding both forms share.
// tests/test_kernels_gpu.py::test_swin_mlp_fused_one_accumulator holds it to the two-accumulator kernel's fp64 bound (also on the heavy-tailed recipe's ranges).
// Domain as before: |x|, |s w| < 65504 (NaN beyond, never a wrong number).
//
// RESULT (tools/mlp_one_acc_ab.py, MI355X, 131 072 rows, hidden 512).  Numerics against fp64: rms 1.49e-7 / max 1.9e-6 (two accumulators: 1.06e-7 / 1.0e-6) on O(1) data;
// heavy-tailed ranges (|out| to 1.9e3) 1.5e-5 / 8.9e-4 vs 1.0e-5 / 6.5e-4; |x| ~ 1e-3 (every residual a subnormal) 3.8e-8 vs 3.6e-8 -- the arithmetic holds: 1.4x the
// two-accumulator rms, the price of a single rounding chain, inside the fp32-GEMM class.  Speed: 224 registers, two workgroups per CU as intended -- and 184 us against
// 171 us for the two-accumulator kernel at one workgroup per CU, packed or un-packed fp32 alike (bench: 116.8 vs 116.7 / 138.6 vs 138.9 images/s).  The second wave
// per SIMD buys nothing: both kernels keep the matrix pipe ~36 % busy, so the fused MLP is bound neither by its accumulator registers nor by occupancy (what the
// additive ablations of tools/mlp_fused_probe.py said in round 2).  The product keeps mlp_fused_h3_kernel.
#define RBA_H3_HELPERS_ONLY
#include "../split_linear_h3.h"

namespace {

// (a, b) -> packed f16 h and the packed UN-scaled residual l = f16(x - h)
__device__ __forceinline__ void split1x2(float a, float b, uint32_t& h, uint32_t& l) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const h2_t hv = {(_Float16)a, (_Float16)b};
  h = __builtin_bit_cast(uint32_t, hv);
  uint32_t r;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(-1.0f), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(-1.0f), "v"(b));
  l = r;
}
__device__ __forceinline__ void split1x8(const f32x4 u, const f32x4 v, f16x8_t& h, f16x8_t& l) {
  u32x4_t hp, lp;
  uint32_t a, r;
  split1x2(u.x, u.y, a, r); hp[0] = a; lp[0] = r;
  split1x2(u.z, u.w, a, r); hp[1] = a; lp[1] = r;
  split1x2(v.x, v.y, a, r); hp[2] = a; lp[2] = r;
  split1x2(v.z, v.w, a, r); hp[3] = a; lp[3] = r;
  h = __builtin_bit_cast(f16x8_t, hp);
  l = __builtin_bit_cast(f16x8_t, lp);
}

// The packed image of rba_split_weight_f16x2 (split_linear_h3.h) for s * weight with un-scaled residuals
__global__ void split_weight_f16x2_scaled_kernel(const float* __restrict__ w, u32x4_t* __restrict__ packed, int N, int K, float scale) {
  const int S = K >> 4;
  const int64_t total = (int64_t)((N + 127) >> 7) * S * 256;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int slot = (int)(i & 1), r = (int)((i >> 1) & 127);
    const int64_t ts = i >> 8;
    const int s = (int)(ts % S), nt = (int)(ts / S);
    const int h = slot ^ ((r >> 3) & 1);
    const int k0 = 32 * (s >> 1) + 16 * h + 8 * (s & 1);
    const int n = nt * 128 + r;
    f16x8_t p0, p1;
    if (n < N) {
      const f32x4* src = reinterpret_cast<const f32x4*>(w + (int64_t)n * K + k0);
      split1x8(src[0] * scale, src[1] * scale, p0, p1);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) p0[e] = p1[e] = (_Float16)0.f;
    }
    u32x4_t* dst = packed + ts * 512 + r * 2 + slot;
    dst[0] = __builtin_bit_cast(u32x4_t, p0);
    dst[256] = __builtin_bit_cast(u32x4_t, p1);
  }
}

template <int OCC>
__global__ __launch_bounds__(256, OCC) void mlp_fused_h1_kernel(const float* __restrict__ X, const u32x4_t* __restrict__ W1p, const float* __restrict__ b1,
                                                                const u32x4_t* __restrict__ W2p, const float* __restrict__ b2, const float* R, float* C, int M,
                                                                int HID, float inv_s1, float inv_s2) {
  constexpr int K1 = 128, NB1 = K1 / 32, N2 = 128, CT = 4;
  constexpr int W1U = 1024, W2U = 1024;                                            // 16-byte units per W1 chunk / W2 block
  __shared__ __attribute__((aligned(16))) u32x4_t lds[2 * (W1U + W2U)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 128;
  const int NJ = HID >> 5;
  const int S16_1 = K1 >> 4;

  // ---- x rows of this wave: 4 blocks x (h g0, l g0, h g1, l g1), resident, residuals un-scaled
  f16x8_t xh[NB1][2], xl[NB1][2];
  {
    int row = m0 + 32 * wave + l31;
    row = row < M ? row : M - 1;
    const char* xp = reinterpret_cast<const char*>(X + (int64_t)row * K1) + 64 * lh;
    f32x4 xr[NB1][4];
#pragma unroll
    for (int b = 0; b < NB1; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) xr[b][q] = *reinterpret_cast<const f32x4*>(xp + b * 128 + q * 16);
#pragma unroll
    for (int b = 0; b < NB1; ++b) {
      split1x8(xr[b][0], xr[b][1], xh[b][0], xl[b][0]);
      split1x8(xr[b][2], xr[b][3], xh[b][1], xl[b][1]);
    }
  }

  // ---- weight staging (as mlp_fused_h3_kernel): unit u = (W1 chunk u + 1, W2 k block u)
  u32x4_t wr1[4], wr2[4];
  const int last = NJ - 1;
  auto w1load = [&](int c) {
    const int cc = c < last ? c : last;
    const int tile = (cc * 32) >> 7, roff = (cc * 32) & 127;
    const u32x4_t* w1 = W1p + (int64_t)tile * S16_1 * 512 + roff * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;
      wr1[q] = w1[(idx >> 7) * 512 + ((idx >> 6) & 1) * 256 + (idx & 63)];
    }
  };
  auto wload = [&](int u) {
    w1load(u + 1);
    const u32x4_t* w2 = W2p + (int64_t)(u < last ? u : last) * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) wr2[q] = w2[tid + 256 * q];
  };
  auto wstore = [&](u32x4_t* buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      buf[tid + 256 * q] = wr1[q];
      buf[W1U + tid + 256 * q] = wr2[q];
    }
  };
  const int fb = l31 * 2 + (lh ^ ((l31 >> 3) & 1));

  f32x16_t acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // s S^T of one chunk (operands swapped: A = W1 fragment, B = x fragment), block by block: fragments of ONE block live at a time
  auto fc1 = [&](const u32x4_t* w1i, f32x16_t& s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int b = 0; b < NB1; ++b) {
      const f16x8_t wh0 = __builtin_bit_cast(f16x8_t, w1i[((2 * b) * 2 + 0) * 64 + fb]), wl0 = __builtin_bit_cast(f16x8_t, w1i[((2 * b) * 2 + 1) * 64 + fb]);
      const f16x8_t wh1 = __builtin_bit_cast(f16x8_t, w1i[((2 * b + 1) * 2 + 0) * 64 + fb]), wl1 = __builtin_bit_cast(f16x8_t, w1i[((2 * b + 1) * 2 + 1) * 64 + fb]);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xh[b][0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xh[b][1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl0, xh[b][0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xl[b][0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl1, xh[b][1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xl[b][1], s, 0, 0, 0);
    }
  };

  w1load(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) lds[(W1U + W2U) + tid + 256 * q] = wr1[q];
  wload(0);
  __syncthreads();
  f32x16_t sc;
  fc1(lds + (W1U + W2U), sc);
  wstore(lds);
  wload(1);
  __syncthreads();

  for (int j = 0; j < NJ; ++j) {
    const u32x4_t* w1i = lds + (j & 1) * (W1U + W2U);                               // W1 chunk j + 1
    const u32x4_t* w2i = w1i + W1U;                                                // W2 k block j
    uint32_t H[4][2], L[4][2];
    auto epi = [&](int q) {
      const int n = 32 * j + 8 * q + 4 * lh;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b1 + n);
      f32x2 y0 = (f32x2){sc[4 * q], sc[4 * q + 1]} * inv_s1 + (f32x2){bv.x, bv.y};
      f32x2 y1 = (f32x2){sc[4 * q + 2], sc[4 * q + 3]} * inv_s1 + (f32x2){bv.z, bv.w};
      y0 = gelu_erf2(y0);
      y1 = gelu_erf2(y1);
      split1x2(y0.x, y0.y, H[q][0], L[q][0]);
      split1x2(y1.x, y1.y, H[q][1], L[q][1]);
    };
    auto pair = [&](int g, f16x8_t& ah, f16x8_t& al) {
      const auto h0 = __builtin_amdgcn_permlane32_swap(H[g][0], H[g + 2][0], false, false), h1 = __builtin_amdgcn_permlane32_swap(H[g][1], H[g + 2][1], false, false);
      const auto l0 = __builtin_amdgcn_permlane32_swap(L[g][0], L[g + 2][0], false, false), l1 = __builtin_amdgcn_permlane32_swap(L[g][1], L[g + 2][1], false, false);
      ah = __builtin_bit_cast(f16x8_t, (u32x4_t){h0[0], h1[0], h0[1], h1[1]});
      al = __builtin_bit_cast(f16x8_t, (u32x4_t){l0[0], l1[0], l0[1], l1[1]});
    };
    // ---- phase A: fc1 of the NEXT chunk, then the GELU arithmetic of this one (the other wave of the SIMD fills the gaps)
    f32x16_t nx;
    fc1(w1i, nx);
    epi(0);
    epi(2);
    // ---- phase B: fc2's k-half g = 0
    f16x8_t ah0, al0, ah1, al1;
    pair(0, ah0, al0);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const f16x8_t bh0 = __builtin_bit_cast(f16x8_t, w2i[fb + 64 * t]), bl0 = __builtin_bit_cast(f16x8_t, w2i[fb + 2 * N2 + 64 * t]);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh0, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl0, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh0, acc[t], 0, 0, 0);
    }
    epi(1);
    epi(3);
    // ---- phase C: fc2's k-half g = 1, and the next unit's weights into the other LDS buffer
    pair(1, ah1, al1);
    wstore(lds + ((j + 1) & 1) * (W1U + W2U));
    wload(j + 2);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const f16x8_t bh1 = __builtin_bit_cast(f16x8_t, w2i[4 * N2 + fb + 64 * t]), bl1 = __builtin_bit_cast(f16x8_t, w2i[4 * N2 + fb + 2 * N2 + 64 * t]);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1, acc[t], 0, 0, 0);
    }
    __syncthreads();
    sc = nx;
  }
  // ---- out = (residual + acc / s2) + bias: lane holds D[row = 8 (r / 4) + 4 lh + r % 4][col = l31] of each 32-column tile
  const int rbase = m0 + 32 * wave + 4 * lh;
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int col = 32 * t + l31;
    const float bv = b2[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbase + 8 * (r >> 2) + (r & 3);
      if (row < M) {
        float y = R[(int64_t)row * N2 + col] + acc[t][r] * inv_s2;
        y = y + bv;
        C[(int64_t)row * N2 + col] = y;
      }
    }
  }
}

}  // namespace

// rba_split_weight_f16x2's packed image of scale * weight with UN-scaled residuals: h = f16(s w), l = f16(s w - h).  scale: a power of two (the caller picks
// the one that puts max |w| into [2^13, 2^14)); the consumer's epilogue multiplies by 1 / scale.
extern "C" int rba_split_weight_f16x2_scaled(const float* weight, void* packed, int N, int K, float scale, void* stream) {
  RBA_CHECK_ARG(N >= 0 && K >= 0 && (K % 32) == 0 && scale > 0.f);
  if (N == 0 || K == 0) return 0;
  RBA_CHECK_ARG(weight && packed && (((uintptr_t)weight | (uintptr_t)packed) & 15) == 0);
  rba_begin();
  const int64_t total = (int64_t)((N + 127) >> 7) * (K >> 4) * 256;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(split_weight_f16x2_scaled_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, weight, reinterpret_cast<u32x4_t*>(packed), N, K, scale);
  return rba_launch_status();
}

// The Mlp + residual of a Swin block with C = 128 in one kernel on ONE accumulator per output (see the head of this file): w1_packed / w2_packed =
// rba_split_weight_f16x2_scaled of fc1.weight [HID, 128] / fc2.weight [128, HID] with scales s1 / s2; inv_s1 = 1 / s1, inv_s2 = 1 / s2.  `out` may be `residual`.
extern "C" int rba_swin_mlp_fused_f16x3s_f32(const float* x, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                                             const float* residual, float* out, int64_t M, int C, int HID, float inv_s1, float inv_s2, void* stream) {
  RBA_CHECK_ARG(M >= 0 && C == 128 && HID >= 64 && (HID % 32) == 0 && inv_s1 > 0.f && inv_s2 > 0.f);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && w1_packed && b1 && w2_packed && b2 && residual && out && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1_packed | (uintptr_t)w2_packed | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)b1) & 15) == 0);
  rba_begin();
  const int64_t MT = (M + 127) / 128;
  hipLaunchKernelGGL((mlp_fused_h1_kernel<2>), dim3((unsigned)MT), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<const u32x4_t*>(w1_packed), b1,
                     reinterpret_cast<const u32x4_t*>(w2_packed), b2, residual, out, (int)M, HID, inv_s1, inv_s2);
  return rba_launch_status();
}
