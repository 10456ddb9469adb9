// K6, f16x3 arithmetic (split_linear_h3.h), fourth kernel form "h3q": 128 x 64 sub-tiles walked by a persistent workgroup with the
// EPILOGUE OF SUB-TILE s RIDING INSIDE THE K LOOP OF SUB-TILE s + 1.
// (reference: the nn.Linear calls of backbone/swin.py:35-41 (Mlp fc1 + GELU, fc2), :131-171 (qkv / proj), :284-293 (residual adds).)
//
// Why (profiles/r02_k6_h3p_ablation.txt, profiles/r03_k6_h3q.txt).  The pipelined 128 x 128 kernel (split_linear_h3p_kernel) issues MFMAs 84 %
// of the time INSIDE its k loop with two workgroups per CU -- but a launch is two rounds of workgroups that start together and therefore
// reach their epilogues together: while both workgroups of a CU run bias + GELU + split + stores (VALU-bound: ~10 us of a 32 us round
// for fc1 of Swin stage 3) the matrix pipe idles, and in the single-resident launches (stage-3 proj / fc2: 256 tiles) one wave per SIMD
// issues MFMAs 55-59 % of the time with nothing to overlap with.  Here
//   * a workgroup owns NS consecutive 64-column sub-tiles of one 128-row panel and runs them as ONE continuous stream of 32-wide k
//     blocks (weight blocks prefetched two steps ahead, A pieces one step ahead, straight across sub-tile boundaries: no drain, no
//     per-tile prologue);
//   * a finished sub-tile's accumulators are folded to one fp32 value per output (main + 2^-11 low: 32 registers per lane) and its
//     epilogue -- bias, residual, GELU, (h, l) split, stores -- is cut into eight 4-channel units that execute inside the first eight
//     k blocks of the NEXT sub-tile, between its MFMAs; only the last sub-tile of a workgroup has an epilogue of its own;
//   * 64-column sub-tiles need 64 accumulator registers, so 512 sub-tiles (stage-3 proj / fc2) give every SIMD two waves.
// Same products in the same order as split_linear_h3p_kernel (operand-swapped form, D^T = W x^T): results are bit-identical to it.
// A operand: the producer's split fragment image ("PRE", see split_linear_h3.h); K / 32 even and >= 8; N % 32 == 0.
#pragma once
#include "split_linear_h3.h"

namespace {

// what the deferred epilogue writes: fp32 rows (bias, optional activation) | fp32 rows with the residual add | the next Linear's split image
enum { H3Q_F32 = 0, H3Q_RES = 1, H3Q_SPLIT = 2 };

// PROBE (tools only, results wrong): 1 no A loads in the loop, 2 no weight loads / LDS stores, 4 no deferred epilogue, 8 no barrier, 16 no fragment reads
template <int MODE, int ACT, int PROBE = 0>
__global__ __launch_bounds__(256, 2) void split_linear_h3q_kernel(const char* __restrict__ Af, const u32x4_t* __restrict__ Wp,
                                                                 const float* __restrict__ bias, void* Cout, const float* R, int M, int N,
                                                                 int K, int NT64, int NS, int NCH, int NWG) {
  __shared__ __attribute__((aligned(16))) u32x4_t lds[2][512];                       // two 8 KiB weight blocks [g][plane][64 rows][2 slots]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  int bid = blockIdx.x;
  if ((NWG & 7) == 0) bid = (bid & 7) * (NWG >> 3) + (bid >> 3);                      // XCD-aware: one XCD, one run of chunks
  const int mt = bid / NCH, ch = bid - mt * NCH;
  const int nt0 = ch * NS;
  const int ns = min(NS, NT64 - nt0);
  const int m0 = mt * 128;
  const int NB = K >> 5, S16 = K >> 4;

  // ---- A pieces of this wave's 32 rows: contiguous 1 KiB wave loads straight into the operand registers
  const int rgrp = min((m0 >> 5) + wave, ((M + 31) >> 5) - 1);
  const char* fbase = Af + ((int64_t)rgrp * NB) * 4096 + lane * 16;
  const int row = m0 + 32 * wave + l31;
  const bool rowok = row < M;
  // ---- packed weight: sub-tile nt = 64-row half (nt & 1) of 128-row tile nt >> 1; LDS unit tid + 256 g <- sub-stage 2 b + g
  const uint32_t woff = (uint32_t)((tid >> 7) * 256 + (tid & 127)) * 16u;
  auto wtile = [&](int nt) { return reinterpret_cast<const char*>(Wp) + ((int64_t)(nt >> 1) * S16 * 512 + (nt & 1) * 128) * 16; };

  f32x16_t accm[2], accl[2], pend[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accm[j][r] = accl[j][r] = pend[j][r] = 0.f;
  u32x4_t wr[2];
  f16x8_t ah[2][2], al[2][2];                                                        // [block parity][g]
  u32x4_t bq[2][4];                                                                  // [column tile][h g0, l g0, h g1, l g1]

  // step t = (sub-tile, block); loads run ahead across sub-tile boundaries and are clamped (re-reads nobody uses) at the very end
  // (branch-free bookkeeping in scalar registers: a branch would cut the scheduling regions of the block below)
  int ws2 = 0, wb2 = 0;                                                              // (sub-tile, block) of the next weight load
  int ab1 = 0;                                                                       // block of the next A load
  const char* wptr = wtile(nt0);
  const int64_t tile_stride = (int64_t)S16 * 8192, loop_bytes = (int64_t)NB * 16384;
  auto wload = [&]() {
    const char* src = wptr + woff;
    wr[0] = *reinterpret_cast<const u32x4_t*>(src);
    wr[1] = *reinterpret_cast<const u32x4_t*>(src + 8192);
    ++wb2;
    const int wrap = wb2 == NB;                                                      // masks, not branches
    const int adv = wrap & (ws2 < ns - 1);                                           // past the last sub-tile: its blocks again (unused)
    // even half -> odd half of the same 128-row tile: + 2048 bytes; odd half -> next tile: + tile_stride - 2048
    const int64_t next_tile = 2048 + ((-(int64_t)((nt0 + ws2) & 1)) & (tile_stride - 4096));
    wptr += 16384 + (((-(int64_t)adv) & next_tile) - ((-(int64_t)wrap) & loop_bytes));
    ws2 += adv;
    wb2 &= wrap - 1;
  };
  auto xloadset = [&](int p) {
    const char* src = fbase + ab1 * 4096;
    ah[p][0] = *reinterpret_cast<const f16x8_t*>(src);
    al[p][0] = *reinterpret_cast<const f16x8_t*>(src + 1024);
    ah[p][1] = *reinterpret_cast<const f16x8_t*>(src + 2048);
    al[p][1] = *reinterpret_cast<const f16x8_t*>(src + 3072);
    ab1 = ab1 + 1 == NB ? 0 : ab1 + 1;
  };
  const int fb = l31 * 2 + (lh ^ ((l31 >> 3) & 1));
  auto bread = [&](const u32x4_t* img, int j, u32x4_t (&d)[4]) {
    d[0] = img[fb + 64 * j];
    d[1] = img[128 + fb + 64 * j];
    d[2] = img[256 + fb + 64 * j];
    d[3] = img[384 + fb + 64 * j];
  };

  wload();
  xloadset(0);
  lds[0][tid] = wr[0];
  lds[0][tid + 256] = wr[1];
  wload();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  bread(lds[0], 0, bq[0]);
  bread(lds[0], 1, bq[1]);

  // ---- the deferred epilogue of the pending sub-tile (columns pn0 ..), unit u = (column tile j, channel quad q): lane = row l31,
  // channels pn0 + 32 j + 8 q + 4 lh + (0..3) = registers 4 q .. 4 q + 3 of pend[j]
  int pn0 = 0;
  bool have = false;
  f32x4 ubias, ures;
  const int NBo = N >> 5;
  char* const obase = MODE == H3Q_SPLIT
                          ? reinterpret_cast<char*>(Cout) + ((int64_t)((m0 >> 5) + wave) * NBo) * 4096 + l31 * 16 + lh * 1024
                          : reinterpret_cast<char*>(Cout) + ((int64_t)(rowok ? row : M - 1) * N) * 4;
  const float* const bsafe = bias ? bias : reinterpret_cast<const float*>(Wp);
  uint32_t bmask = bias ? ~0u : 0u;
  asm volatile("" : "+v"(bmask));                                                    // keep the load unconditional (a branch would cut the block's scheduling region)
  const char* const rbase = reinterpret_cast<const char*>(R) + ((int64_t)(rowok ? row : M - 1) * N) * 4;
  auto unit_loads = [&](int u) {                                                     // issued at the head of a block, consumed at its end
    const int n = pn0 + 32 * (u >> 2) + 8 * (u & 3) + 4 * lh;
    const int nc = n < N ? n : 0;
    const u32x4_t bv = *reinterpret_cast<const u32x4_t*>(bsafe + nc);                // no bias: any readable address, masked to zero
    ubias = __builtin_bit_cast(f32x4, bv & bmask);
    if (MODE == H3Q_RES) ures = *reinterpret_cast<const f32x4*>(rbase + (int64_t)nc * 4);
  };
  f32x2 y0, y1;
  auto unit_math0 = [&](int u) {                                                     // first channel pair
    const int j = u >> 2, q = u & 3;
    y0 = (f32x2){pend[j][4 * q], pend[j][4 * q + 1]};
    if (MODE == H3Q_RES) y0 = (f32x2){ures.x, ures.y} + y0;
    y0 = y0 + (f32x2){ubias.x, ubias.y};
    if (ACT == 1) y0 = gelu_erf2(y0);
    if (ACT == 2) y0 = (f32x2){rba_relu(y0.x), rba_relu(y0.y)};
  };
  auto unit_math1 = [&](int u) {                                                     // second pair
    const int j = u >> 2, q = u & 3;
    y1 = (f32x2){pend[j][4 * q + 2], pend[j][4 * q + 3]};
    if (MODE == H3Q_RES) y1 = (f32x2){ures.z, ures.w} + y1;
    y1 = y1 + (f32x2){ubias.z, ubias.w};
    if (ACT == 1) y1 = gelu_erf2(y1);
    if (ACT == 2) y1 = (f32x2){rba_relu(y1.x), rba_relu(y1.y)};
  };
  auto unit_store = [&](int u) {
    const int j = u >> 2, q = u & 3;
    const int nj = pn0 + 32 * j;
    const bool ok = have && rowok && nj < N;
    if (MODE == H3Q_SPLIT) {
      uint32_t h0, l0, h1, l1;
      rba_split_f16x2(y0.x, y0.y, h0, l0);
      rba_split_f16x2(y1.x, y1.y, h1, l1);
      const auto p0 = __builtin_amdgcn_permlane32_swap(h0, l0, false, false), p1 = __builtin_amdgcn_permlane32_swap(h1, l1, false, false);
      const u32x4_t piece = {p0[0], p1[0], p0[1], p1[1]};
      if (ok) *reinterpret_cast<u32x4_t*>(obase + (int64_t)(nj >> 5) * 4096 + (q & 1) * 2048 + (q >> 1) * 512) = piece;
    } else {
      const f32x4 v = {y0.x, y0.y, y1.x, y1.y};
      if (ok) *reinterpret_cast<f32x4*>(obase + (int64_t)(nj + 8 * q + 4 * lh) * 4) = v;
    }
  };

  auto mm = [](const f16x8_t a, const f16x8_t b, const f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0); };
#define H3Q_MFMA6(J, P)                                                                                                   \
  {                                                                                                                       \
    const f16x8_t bh0 = __builtin_bit_cast(f16x8_t, bq[J][0]), bl0 = __builtin_bit_cast(f16x8_t, bq[J][1]);               \
    const f16x8_t bh1 = __builtin_bit_cast(f16x8_t, bq[J][2]), bl1 = __builtin_bit_cast(f16x8_t, bq[J][3]);               \
    accm[J] = mm(ah[P][0], bh0, accm[J]);                                                                                 \
    accl[J] = mm(ah[P][0], bl0, accl[J]);                                                                                 \
    accm[J] = mm(ah[P][1], bh1, accm[J]);                                                                                 \
    accl[J] = mm(al[P][0], bh0, accl[J]);                                                                                 \
    accl[J] = mm(ah[P][1], bl1, accl[J]);                                                                                 \
    accl[J] = mm(al[P][1], bh1, accl[J]);                                                                                 \
  }
// scheduling patterns: the epilogue unit's arithmetic (VALU incl. transcendentals: mask 0x402) and the next block's fragment reads
// (DS read: 0x100) go out between the MFMAs (0x008)
#define H3Q_SCHED_M_V(NV)                                                                                                 \
  _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);                                                                   \
  }
#define H3Q_SCHED_M_D_V(NV)                                                                                               \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);                                                                   \
  }                                                                                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);                                                                   \
  }
// one 32-wide block; P = parity of its A operand set and of its LDS buffer; EPI: carries unit U of the pending sub-tile's epilogue
#define H3Q_BLOCK(P, EPI, U)                                                                                              \
  {                                                                                                                       \
    if (!(PROBE & 2)) {                                                                                                   \
      lds[(P) ^ 1][tid] = wr[0];                                                                                          \
      lds[(P) ^ 1][tid + 256] = wr[1];                                                                                    \
    }                                                                                                                     \
    if (EPI) unit_loads(U);                                                                                               \
    if (!(PROBE & 2)) wload();                                                                                            \
    if (!(PROBE & 1)) xloadset((P) ^ 1);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    H3Q_MFMA6(0, P)                                                                                                       \
    if (EPI) {                                                                                                            \
      unit_math0(U);                                                                                                      \
      H3Q_SCHED_M_V(5)                                                                                                    \
    }                                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if (!(PROBE & 8)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    H3Q_MFMA6(1, P)                                                                                                       \
    if (!(PROBE & 16)) bread(lds[(P) ^ 1], 0, bq[0]);                                                                                        \
    if (EPI) {                                                                                                            \
      unit_math1(U);                                                                                                      \
      H3Q_SCHED_M_D_V(5)                                                                                                  \
    } else {                                                                                                              \
      H3Q_SCHED_M_D_V(0)                                                                                                  \
    }                                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    if (!(PROBE & 16)) bread(lds[(P) ^ 1], 1, bq[1]);                                                                     \
    if (EPI) unit_store(U);                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }

  for (int s = 0; s < ns; ++s) {
    if (have && !(PROBE & 4)) {
      H3Q_BLOCK(0, true, 0) H3Q_BLOCK(1, true, 1) H3Q_BLOCK(0, true, 2) H3Q_BLOCK(1, true, 3)
      H3Q_BLOCK(0, true, 4) H3Q_BLOCK(1, true, 5) H3Q_BLOCK(0, true, 6) H3Q_BLOCK(1, true, 7)
    } else {
      H3Q_BLOCK(0, false, 0) H3Q_BLOCK(1, false, 0) H3Q_BLOCK(0, false, 0) H3Q_BLOCK(1, false, 0)
      H3Q_BLOCK(0, false, 0) H3Q_BLOCK(1, false, 0) H3Q_BLOCK(0, false, 0) H3Q_BLOCK(1, false, 0)
    }
    for (int b = 8; b < NB; b += 2) {
      H3Q_BLOCK(0, false, 0)
      H3Q_BLOCK(1, false, 0)
    }
    // fold: value = main + 2^-11 low (the first operation of every epilogue form), accumulators restart at zero
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 y = (f32x2){accl[j][r], accl[j][r + 1]} * 0.00048828125f + (f32x2){accm[j][r], accm[j][r + 1]};
        pend[j][r] = y.x;
        pend[j][r + 1] = y.y;
        accm[j][r] = accm[j][r + 1] = accl[j][r] = accl[j][r + 1] = 0.f;
      }
    pn0 = (nt0 + s) * 64;
    have = true;
  }
#undef H3Q_BLOCK
#undef H3Q_SCHED_M_D_V
#undef H3Q_SCHED_M_V
#undef H3Q_MFMA6
  // ---- the last sub-tile's epilogue stands alone
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    unit_loads(u);
    unit_math0(u);
    unit_math1(u);
    unit_store(u);
  }
}

// sub-tiles per workgroup: minimise rounds x NS over the 512 workgroup slots of the chip (two per CU); on ties prefer the longer run (a
// larger share of the epilogues rides inside a k loop)
inline int h3q_pick_ns(int64_t MT, int NT64) {
  int best = 1;
  int64_t best_cost = INT64_MAX;
  for (int nsub = 1; nsub <= 8 && nsub <= NT64; ++nsub) {
    const int64_t wgs = MT * ((NT64 + nsub - 1) / nsub);
    const int64_t rounds = (wgs + 511) / 512;
    const int64_t cost = rounds * (nsub * 8 + 1);                                  // + 1: the stand-alone epilogue of a round
    if (cost <= best_cost) {
      best_cost = cost;
      best = nsub;
    }
  }
  return best;
}

inline bool h3q_supported(int64_t M, int N, int K) { return (K % 64) == 0 && K >= 256 && (N % 32) == 0 && M >= 1; }

template <int MODE, int ACT, int PROBE = 0>
int launch_h3q(const void* xf, const u32x4_t* wp, const float* bias, const float* res, void* out, int64_t M, int N, int K, hipStream_t st) {
  const int64_t MT = (M + 127) / 128;
  const int NT64 = (N + 63) / 64;
  const int NS = h3q_pick_ns(MT, NT64);
  const int NCH = (NT64 + NS - 1) / NS;
  const int64_t NWG = MT * NCH;
  if (NWG >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  // the deferred epilogue reads bias / residual and writes fp32 rows in 16-byte units (N % 32 == 0 keeps every unit inside its row)
  if ((((uintptr_t)bias | (uintptr_t)res | (uintptr_t)out | (uintptr_t)xf) & 15) != 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3q_kernel<MODE, ACT, PROBE>), dim3((unsigned)NWG), dim3(256), 0, st, reinterpret_cast<const char*>(xf), wp, bias, out,
                     res, (int)M, N, K, NT64, NS, NCH, (int)NWG);
  return 0;
}

}  // namespace
