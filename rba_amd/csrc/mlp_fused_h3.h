// Fused Swin MLP for C = 128 (backbone/swin.py:35-41 + the `x = x + mlp(norm2(x))` of :293): out = residual + fc2(GELU(fc1(y))) in ONE
// kernel, the [M, 4C] hidden tensor never written.  At stage 1 of Swin-B (131 072 tokens) that tensor is 268 MB: the unfused pair is
// bound by writing and re-reading it (fc1 147 us + fc2 74 us for 25 us of matrix work).
//
// Same f16x3 arithmetic, operand layouts and accumulation orders as split_linear_h3.h, so the result is bit-identical to
// rba_split_linear_f16x3_gelu_split_out followed by rba_split_linear_f16x3_frag_f32(residual):
//   * a workgroup owns 128 rows (four waves x 32); the wave's x rows (K = 128: four 32-wide blocks) are loaded ONCE, split once and
//     stay in 64 registers as the B operand of fc1;
//   * the hidden dimension is walked 32 columns at a time.  Chunk j: S^T = W1[32 j .. 32 j + 31, :] x^T with the MFMA operands swapped
//     (24 MFMAs), so that a lane holds four consecutive hidden channels of ONE row per register quad; bias + exact GELU + (h, l) split
//     in registers; v_permlane32_swap(piece q, piece q + 2) then leaves every lane with the two 8-channel pieces of ITS k-half --
//     which is exactly the A fragment of fc2's 32-wide k block j.  No LDS, no global round trip for the hidden activations;
//   * fc2: the four 32-column tiles of the 128 outputs accumulate A(j) W2[:, 32 j .. 32 j + 31]^T (24 MFMAs) into 128 accumulator registers;
//   * W1's chunk (16 KiB: eight sub-stages x two planes x 32 rows) and W2's k block (16 KiB) go global -> registers -> LDS one chunk
//     ahead, two buffers, one barrier per chunk; epilogue = the residual epilogue of the Linear kernels.
#pragma once
#include "split_linear_h3.h"

namespace {

// PROBE (tune builds only, results wrong): 1 no GELU arithmetic, 2 no fc1 MFMAs, 4 no fc2 MFMAs, 8 no weight staging / barrier in the loop
// LNIN (round 5): X holds the block's residual stream, not norm2's output: the wave normalises its rows itself (swin.py:293 `self.mlp(self.norm2(x))`; the
// arithmetic of add_layer_norm_kernel: mean, centred variance, (v - mean) * rstd * gamma + beta) -- a lane already holds half of a row, the other half is one
// xor-shuffle away -- so that neither the LayerNorm launch nor its output tensor exist (X is then also the residual R).
struct MlpNorm {
  const float* gamma;
  const float* beta;
  float eps;
};
template <int PROBE = 0, bool LNIN = false>
__global__ __launch_bounds__(256, 1) void mlp_fused_h3_kernel(const float* X, const u32x4_t* __restrict__ W1p,
                                                              const float* __restrict__ b1, const u32x4_t* __restrict__ W2p,
                                                              const float* __restrict__ b2, const float* R, float* C, int M, int HID,
                                                              MlpNorm ln = MlpNorm{nullptr, nullptr, 0.f}) {
  constexpr int K1 = 128, NB1 = K1 / 32, N2 = 128, CT = 4;
  constexpr int W1U = 1024, W2U = 1024;                                            // 16-byte units per W1 chunk / W2 block
  __shared__ __attribute__((aligned(16))) u32x4_t lds[2 * (W1U + W2U)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 128;
  const int NJ = HID >> 5;                                                         // hidden chunks = fc2's k blocks
  const int S16_1 = K1 >> 4;                                                       // sub-stages of W1's K

  // ---- x rows of this wave: 4 blocks x (h g0, l g0, h g1, l g1), resident
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
    if (LNIN) {                                                                    // lane: channels 32 b + 16 lh + 4 q .. + 3 of row l31
      // The sums run over (even, odd) channel PAIRS and are folded once at the end: written as (x + y) + (z + w) per quad the compiler emits the horizontal
      // packed add `v_pk_add_f32 v, v, v op_sel:[0,1] op_sel_hi:[1,0]` -- a cross select on source 1, the instruction form that went wrong in the
      // GroupNorm fold (split_linear_gnf.hip); tests/test_host_cpu.py keeps the library free of it.
      f32x2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NB1; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) sum2 += (f32x2){xr[b][q].x, xr[b][q].y} + (f32x2){xr[b][q].z, xr[b][q].w};
      float sum = sum2.x + sum2.y;
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum / (float)K1;
      f32x2 sq2 = {0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NB1; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 d = xr[b][q] - mean;
          const f32x2 d01 = {d.x, d.y}, d23 = {d.z, d.w};
          sq2 += d01 * d01 + d23 * d23;
        }
      float sq = sq2.x + sq2.y;
      sq += __shfl_xor(sq, 32, 64);
      const float rstd = 1.0f / sqrtf(sq / (float)K1 + ln.eps);
      const char* gp = reinterpret_cast<const char*>(ln.gamma) + 64 * lh;
      const char* bp = reinterpret_cast<const char*>(ln.beta) + 64 * lh;
#pragma unroll
      for (int b = 0; b < NB1; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(gp + b * 128 + q * 16), b4 = *reinterpret_cast<const f32x4*>(bp + b * 128 + q * 16);
          xr[b][q] = (xr[b][q] - mean) * rstd * g4 + b4;
        }
    }
#pragma unroll
    for (int b = 0; b < NB1; ++b) {
      split_h3(xr[b][0], xr[b][1], xh[b][0], xl[b][0]);
      split_h3(xr[b][2], xr[b][3], xh[b][1], xl[b][1]);
    }
  }

  // ---- weight staging.  Unit u = (W1 chunk u + 1, W2 k block u): what iteration u needs, because fc1 runs ONE chunk ahead of fc2 (its MFMAs
  // overlap the GELU arithmetic of the previous chunk).  W1 chunk c = rows 32 c .. 32 c + 31 of every (sub-stage, plane):
  // unit index (s, p, u) <- tile, s, p, (roff + u / 2), u % 2
  u32x4_t wr1[4], wr2[4];
  const int last = NJ - 1;
  auto w1load = [&](int c) {
    const int cc = c < last ? c : last;
    const int tile = (cc * 32) >> 7, roff = (cc * 32) & 127;
    const u32x4_t* w1 = W1p + (int64_t)tile * S16_1 * 512 + roff * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;                                               // (s, p, u) = (idx / 128, idx / 64 % 2, idx % 64)
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

  f32x16_t accm[CT], accl[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accm[t][r] = accl[t][r] = 0.f;

  // S^T of one chunk (operands swapped: A = W1 fragment, B = x fragment)
  auto fc1 = [&](const u32x4_t* w1i, f32x16_t& sm, f32x16_t& sl) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sm[r] = sl[r] = 0.f;
#pragma unroll
    for (int b = 0; b < NB1; ++b) {
      const f16x8_t wh0 = __builtin_bit_cast(f16x8_t, w1i[((2 * b) * 2 + 0) * 64 + fb]), wl0 = __builtin_bit_cast(f16x8_t, w1i[((2 * b) * 2 + 1) * 64 + fb]);
      const f16x8_t wh1 = __builtin_bit_cast(f16x8_t, w1i[((2 * b + 1) * 2 + 0) * 64 + fb]), wl1 = __builtin_bit_cast(f16x8_t, w1i[((2 * b + 1) * 2 + 1) * 64 + fb]);
      sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xh[b][0], sm, 0, 0, 0);
      sl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl0, xh[b][0], sl, 0, 0, 0);
      sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xh[b][1], sm, 0, 0, 0);
      sl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xl[b][0], sl, 0, 0, 0);
      sl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl1, xh[b][1], sl, 0, 0, 0);
      sl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xl[b][1], sl, 0, 0, 0);
    }
  };

  // prologue: W1 chunk 0 alone (into the W1 half of buffer 1), S(0); then unit 0 into buffer 0
  w1load(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) lds[(W1U + W2U) + tid + 256 * q] = wr1[q];
  wload(0);
  __syncthreads();
  f32x16_t sm, sl;
  fc1(lds + (W1U + W2U), sm, sl);
  wstore(lds);
  wload(1);
  __syncthreads();

  for (int j = 0; j < NJ; ++j) {
    const u32x4_t* w1i = lds + (j & 1) * (W1U + W2U);                               // W1 chunk j + 1
    const u32x4_t* w2i = w1i + W1U;                                                // W2 k block j
    // bias + GELU + split of piece q of chunk j (VALU only); the lane holds hidden channels 32 j + 8 q + 4 lh + (0..3) of row l31
    uint32_t H[4][2], L[4][2];
    auto epi = [&](int q) {
      const int n = 32 * j + 8 * q + 4 * lh;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b1 + n);                    // (no null test: a branch here would fence the MFMA / VALU interleaving)
      f32x2 y0 = (f32x2){sl[4 * q], sl[4 * q + 1]} * 0.00048828125f + (f32x2){sm[4 * q], sm[4 * q + 1]};
      f32x2 y1 = (f32x2){sl[4 * q + 2], sl[4 * q + 3]} * 0.00048828125f + (f32x2){sm[4 * q + 2], sm[4 * q + 3]};
      y0 = y0 + (f32x2){bv.x, bv.y};
      y1 = y1 + (f32x2){bv.z, bv.w};
      if (!(PROBE & 1)) {
        y0 = gelu_erf2(y0);
        y1 = gelu_erf2(y1);
      }
      rba_split_f16x2(y0.x, y0.y, H[q][0], L[q][0]);
      rba_split_f16x2(y1.x, y1.y, H[q][1], L[q][1]);
    };
    // v_permlane32_swap(piece g, piece g + 2): a low lane keeps its half of piece g and receives the partner's half of piece g; a high lane
    // receives the partner's half of piece g + 2 and keeps its own: every lane ends with piece 2 lh + g whole = the A fragment of its k-half
    auto pair = [&](int g, f16x8_t& ah, f16x8_t& al) {
      const auto h0 = __builtin_amdgcn_permlane32_swap(H[g][0], H[g + 2][0], false, false), h1 = __builtin_amdgcn_permlane32_swap(H[g][1], H[g + 2][1], false, false);
      const auto l0 = __builtin_amdgcn_permlane32_swap(L[g][0], L[g + 2][0], false, false), l1 = __builtin_amdgcn_permlane32_swap(L[g][1], L[g + 2][1], false, false);
      ah = __builtin_bit_cast(f16x8_t, (u32x4_t){h0[0], h1[0], h0[1], h1[1]});
      al = __builtin_bit_cast(f16x8_t, (u32x4_t){l0[0], l1[0], l0[1], l1[1]});
    };
    // ---- phase A: the 24 MFMAs of fc1 for the NEXT chunk under the GELU arithmetic of pieces 0 and 2 of this one
    u32x4_t f1[NB1][4], f2[CT][2];
#pragma unroll
    for (int b = 0; b < NB1; ++b) {
      f1[b][0] = w1i[((2 * b) * 2 + 0) * 64 + fb];
      f1[b][1] = w1i[((2 * b) * 2 + 1) * 64 + fb];
      f1[b][2] = w1i[((2 * b + 1) * 2 + 0) * 64 + fb];
      f1[b][3] = w1i[((2 * b + 1) * 2 + 1) * 64 + fb];
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {                                                 // fc2 fragments of k-half g = 0
      f2[t][0] = w2i[fb + 64 * t];
      f2[t][1] = w2i[fb + 2 * N2 + 64 * t];
    }
    f32x16_t nm, nl;
#pragma unroll
    for (int r = 0; r < 16; ++r) nm[r] = nl[r] = 0.f;
#pragma unroll
    for (int b = 0; b < ((PROBE & 2) ? 0 : NB1); ++b) {
      const f16x8_t wh0 = __builtin_bit_cast(f16x8_t, f1[b][0]), wl0 = __builtin_bit_cast(f16x8_t, f1[b][1]);
      const f16x8_t wh1 = __builtin_bit_cast(f16x8_t, f1[b][2]), wl1 = __builtin_bit_cast(f16x8_t, f1[b][3]);
      nm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xh[b][0], nm, 0, 0, 0);
      nl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl0, xh[b][0], nl, 0, 0, 0);
      nm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xh[b][1], nm, 0, 0, 0);
      nl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, xl[b][0], nl, 0, 0, 0);
      nl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl1, xh[b][1], nl, 0, 0, 0);
      nl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, xl[b][1], nl, 0, 0, 0);
    }
    epi(0);
    epi(2);
#pragma unroll
    for (int i_ = 0; i_ < 24; ++i_) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, 7, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B: fc2's k-half g = 0 (12 MFMAs: per tile main, low, low -- the accumulation order of the unfused kernel is g-major within each
    // accumulator, so this regrouping keeps every sum bit-identical) under the GELU arithmetic of pieces 1 and 3
    f16x8_t ah0, al0, ah1, al1;
    pair(0, ah0, al0);
    u32x4_t f3[CT][2];
#pragma unroll
    for (int t = 0; t < CT; ++t) {                                                 // fc2 fragments of k-half g = 1
      f3[t][0] = w2i[4 * N2 + fb + 64 * t];
      f3[t][1] = w2i[4 * N2 + fb + 2 * N2 + 64 * t];
    }
#pragma unroll
    for (int t = 0; t < ((PROBE & 4) ? 0 : CT); ++t) {
      const f16x8_t bh0 = __builtin_bit_cast(f16x8_t, f2[t][0]), bl0 = __builtin_bit_cast(f16x8_t, f2[t][1]);
      accm[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh0, accm[t], 0, 0, 0);
      accl[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl0, accl[t], 0, 0, 0);
      accl[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh0, accl[t], 0, 0, 0);
    }
    epi(1);
    epi(3);
#pragma unroll
    for (int i_ = 0; i_ < 12; ++i_) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, 14, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase C: fc2's k-half g = 1, and the next unit's weights into the other LDS buffer (last read in iteration j - 1)
    pair(1, ah1, al1);
    if (!(PROBE & 8)) {
      wstore(lds + ((j + 1) & 1) * (W1U + W2U));
      wload(j + 2);
    }
#pragma unroll
    for (int t = 0; t < ((PROBE & 4) ? 0 : CT); ++t) {
      const f16x8_t bh1 = __builtin_bit_cast(f16x8_t, f3[t][0]), bl1 = __builtin_bit_cast(f16x8_t, f3[t][1]);
      accm[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1, accm[t], 0, 0, 0);
      accl[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1, accl[t], 0, 0, 0);
      accl[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1, accl[t], 0, 0, 0);
    }
    if (!(PROBE & 8)) __syncthreads();
    sm = nm;
    sl = nl;
  }
  h3_epilogue<0, CT, 0, true>(accm, accl, b2, C, R, M, N2, m0, 0, 128, 128, wave, l31, lh);
}

inline int launch_mlp_fused(const float* x, const u32x4_t* w1p, const float* b1, const u32x4_t* w2p, const float* b2, const float* res, float* out,
                            int64_t M, int HID, hipStream_t st, const float* gamma = nullptr, const float* beta = nullptr, float eps = 0.f) {
  const int64_t MT = (M + 127) / 128;
  if (MT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  if (gamma)
    hipLaunchKernelGGL((mlp_fused_h3_kernel<0, true>), dim3((unsigned)MT), dim3(256), 0, st, x, w1p, b1, w2p, b2, res, out, (int)M, HID, MlpNorm{gamma, beta, eps});
  else
    hipLaunchKernelGGL((mlp_fused_h3_kernel<0>), dim3((unsigned)MT), dim3(256), 0, st, x, w1p, b1, w2p, b2, res, out, (int)M, HID, MlpNorm{nullptr, nullptr, 0.f});
  return 0;
}

}  // namespace
