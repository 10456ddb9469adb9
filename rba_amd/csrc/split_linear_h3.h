// K6, third form ("f16x3"): fp32-accurate Linear on the f16 matrix pipe with THREE products per fp32 product.
// (reference: the nn.Linear calls of backbone/swin.py:44-71 (Mlp), :131-171 (qkv / proj), :319-343 (PatchMerging).)
//
// Arithmetic.  Every fp32 x with |x| < 65504 is  x = h + 2^-11 l + e,  h = f16(x) (rne), l = f16((x - h) 2^11) (x - h is exact
// in fp32; scaled by 2^11 it has x's own magnitude, so it never falls into f16's subnormals unless x itself is below 6e-5, and
// even then the scaled residual keeps the bits h lost), |e| <= 2^-23 |x|.  With the weight split the same way,
//     x w = h_x h_w + 2^-11 (h_x l_w + l_x h_w) + O(2^-22 |x w|)
// (dropped: l_x l_w 2^-22 and the two e terms).  f16 x f16 products are exact in fp32, so three v_mfma_f32_32x32x16_f16 per k-step,
// the first into a "main" accumulator and the other two into a "low" accumulator that is added with weight 2^-11 in the epilogue,
// give the fp32 product to 2^-22 relative PER TERM with unbiased (rne) errors: against fp64 the result is as close as hipBLASLt's
// fp32 GEMM (whose own accumulation rounding, ~sqrt(K) 2^-24, dominates both) -- tests/test_kernels_gpu.py measures that on
// every Swin shape.  Domain: |x|, |w| < 65504 (f16's range; beyond it h is inf and the output row is NaN -- loud, never silently
// wrong; the bf16x6 form of split_linear_dma.h has fp32's full range and stays selectable).
//
// Why (profiles/r02_k6_*.txt): the bf16x6 kernel is pinned at the LDS-DMA fill rate of the chip (655 MB per launch through
// global_load_lds at ~7 TB/s = 93 us for Swin stage-3 fc1, whatever the tiling), and its six MFMAs per product cost 63 us on their
// own.  Here the matrix work halves (three MFMAs), the weight image shrinks from 6 to 4 bytes per element, and NOTHING goes
// through LDS-DMA: the activation rows a wave owns are loaded straight into its registers (each lane 64 contiguous bytes per
// 32-wide k block = one full 128-byte line per row and lane pair) and split there -- activations never touch LDS -- and the
// packed weight block is copied global -> registers -> LDS by the four waves together, one block ahead.
//
// Layout.  Packed weight (rba_split_weight_f16x2): [N/128][K/16][2 planes][128 rows][2 slots] x 16 B, plane 0 = h, plane 1 = l;
// sub-stage s = 2 b + g of 32-wide block b holds, for row r, slot h ^ ((r >> 3) & 1):  k = 32 b + 16 h + 8 g + (0..7)  -- lane
// group h = lane / 32 of the MFMA owns bytes [64 h, 64 h + 64) of its row's 128-byte block and uses them in two MFMAs (g = 0, 1).
#pragma once
#include "split_linear_dma.h"

// tools-only knobs (exported by split_linear_dma.hip; zero in the product): see the `stagger` parameter of split_linear_h3p_kernel
RBA_KNOB_EXTERN(rba_k6_stagger, 0);

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// 8 fp32 -> the two f16x8 MFMA operands: h = f16(x), l = f16((x - h) * 2^11) = f16(fma(h, -2^11, 2^11 x)) (the fma is exact).
// Four VALU per pair: v_cvt_pk_f16_f32, v_pk_mul_f32, and v_fma_mixlo/mixhi_f16 reading the f16 halves of h in place.
__device__ __forceinline__ void split_h3(const f32x4 u, const f32x4 v, f16x8_t& h, f16x8_t& l) {
  const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
  u32x4_t hp, lp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t a, r;
    rba_split_f16x2(x[2 * i], x[2 * i + 1], a, r);
    hp[i] = a;
    lp[i] = r;
  }
  h = __builtin_bit_cast(f16x8_t, hp);
  l = __builtin_bit_cast(f16x8_t, lp);
}

// Exact-form GELU (nn.GELU default, swin.py:51) on two values with packed fp32 arithmetic, erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7)
// like gelu_erf (split_linear_dma.h), re-arranged in round 4 for fewer instructions -- the fc1 epilogue is bound by VALU issue:
//   gelu(v) = 0.5 v (1 + erf(v / sqrt 2)) = max(v, 0) - |v| * (0.5 erfc(|v| / sqrt 2)),   0.5 erfc(|z|) = (b1 t + ... + b5 t^5) e^(-z^2),
//   t = 1 / (1 + p |z|), b_i = a_i / 2; with a = |v|:  1 + p |z| = a (p / sqrt 2) + 1  and  e^(-z^2) = exp2(-(a u0)^2), u0 = sqrt(log2(e) / 2)
// -- no select on the sign, no separate x = |v| / sqrt 2, the 0.5 and the log2(e) folded into constants: per PAIR 12 packed + 4 transcendental
// + 2 plain instructions (was 11 + 4 + 10).  Same polynomial, so the same 1.5e-7 bound; v = +inf gives NaN (inf * 0) where the old form gave
// inf -- the f16x3 kernels never produce an infinity (out-of-range values are NaN already).
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
  const f32x2 a = {fabsf(v.x), fabsf(v.y)};
  const f32x2 d = a * 0.23164189f + 1.0f;                                         // 0.3275911 / sqrt(2)
  const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2 p = t * 0.5307027145f + -0.7265760135f;
  p = p * t + 0.7107068705f;
  p = p * t + -0.142248368f;
  p = p * t + 0.127414796f;
  const f32x2 u = a * 0.8493218003f;                                              // sqrt(log2(e) / 2)
  const f32x2 s = -u * u;
  const f32x2 ex = {__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)};
  const f32x2 q = p * t * ex;
  const f32x2 r = (v + a) * 0.5f;                                                 // max(v, 0), exactly, in two packed instructions (fmaxf costs four plain ones per pair)
  return r - a * q;
}

#ifndef RBA_H3_HELPERS_ONLY   // (split_linear_ws.hip takes the arithmetic and epilogue helpers of this file, not its kernels)
// One thread per 16-byte unit of the packed image (see the layout above); rows N .. Np - 1 of the last 128-row tile are zero.
__global__ void split_weight_f16x2_kernel(const float* __restrict__ w, u32x4_t* __restrict__ packed, int N, int K) {
  const int S = K >> 4;
  const int64_t total = (int64_t)((N + 127) >> 7) * S * 256;                        // (tile, sub-stage, row, slot)
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
      split_h3(src[0], src[1], p0, p1);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) p0[e] = p1[e] = (_Float16)0.f;
    }
    u32x4_t* dst = packed + ts * 512 + r * 2 + slot;
    dst[0] = __builtin_bit_cast(u32x4_t, p0);
    dst[256] = __builtin_bit_cast(u32x4_t, p1);
  }
}

#endif  // RBA_H3_HELPERS_ONLY

// Epilogue shared by the three kernels: lane holds D[row = 8 (r / 4) + 4 (lane / 32) + r % 4][col = lane % 32] of each 32 x 32 tile of
// its wave's 32 rows; value = main + 2^-11 low.  RES: out = (residual + value) + bias -- the `x = x + proj(...)` / `x = x + fc2(...)`
// of a transformer block folded into the GEMM, in the fused add + LayerNorm kernel's own order of operations (bit-identical to it),
// `out` may be the residual tensor itself (every element is read and written by the same lane).
// GNM (round 4): the launch also leaves the GroupNorm moments of its OUTPUT -- per (128-row tile, group of cpg consecutive output channels) the triple
// (n, mean, M2) in the workspace layout of gn_stats_nhwc_kernel / gn_merge_kernel (group_norm.hip): ws[((b G + g) splits + s) * 3], splits = P / 128 row tiles per
// image -- so that the statistics pass over the convolution output (134 MB at the 1/4-resolution level) is never run: rba_group_norm_nhwc_merge_f32 turns them
// into (mean, rstd).  Every row of the tile is a valid row of ONE image (M % 128 == 0, P % 128 == 0: the launcher checks), cpg in {4, 8, 16, 32}, N % 128 == 0.
// Fixed summation order: 16 rows in a lane, lane halves, the cpg lanes of a group, the four waves (deterministic).
struct GnMoments {
  float* ws;
  int G, cpg, P;
};
template <int ACT, int CT, int PROBE, bool RES, bool GNM = false>
__device__ __forceinline__ void h3_epilogue(const f32x16_t (&accm)[CT], const f32x16_t (&accl)[CT], const float* __restrict__ bias,
                                            float* C, const float* R, int M, int N, int m0, int n0, int BM, int BN, int wave, int l31,
                                            int lh, GnMoments gm = GnMoments{nullptr, 1, 1, 128}, float* sh = nullptr) {
  const bool interior = m0 + BM <= M && n0 + BN <= N;
  if (GNM) __syncthreads();                                                          // `sh` overlays the operand ring: every wave has left the k loop
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = n0 + 32 * j + l31;
    const float bv = (bias && col < N) ? bias[col] : 0.f;
    const int rbase = m0 + 32 * wave + 4 * lh;
    float* dst = C + (int64_t)rbase * N + col;
    f32x16_t res;
    if (RES) {
      const float* src = R + (int64_t)rbase * N + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = 8 * (r >> 2) + (r & 3);
        res[r] = (interior || (col < N && rbase + ro < M)) ? src[(int64_t)ro * N] : 0.f;
      }
    }
    f32x16_t v;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f32x2 y = (f32x2){accl[j][r], accl[j][r + 1]} * 0.00048828125f + (f32x2){accm[j][r], accm[j][r + 1]};
      if (RES) y = (f32x2){res[r], res[r + 1]} + y;
      y = y + bv;
      if (ACT == 1) y = gelu_erf2(y);
      if (ACT == 2) y = (f32x2){rba_relu(y.x), rba_relu(y.y)};
      v[r] = y.x;
      v[r + 1] = y.y;
    }
    if (GNM) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sm += v[r];
        sq = fmaf(v[r], v[r], sq);
      }
      sm += __shfl_xor(sm, 32, 64);
      sq += __shfl_xor(sq, 32, 64);
      for (int o = 1; o < gm.cpg; o <<= 1) {
        sm += __shfl_xor(sm, o, 64);
        sq += __shfl_xor(sq, o, 64);
      }
      if (lh == 0 && (l31 & (gm.cpg - 1)) == 0) {
        float* d = sh + ((wave * (BN / gm.cpg)) + (32 * j + l31) / gm.cpg) * 2;
        d[0] = sm;
        d[1] = sq;
      }
    }
    if (PROBE & 4) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += v[r];
      if (sum == 1234.5f) dst[0] = sum;
    } else if (interior) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(int64_t)(8 * (r >> 2) + (r & 3)) * N] = v[r];
    } else if (col < N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = 8 * (r >> 2) + (r & 3);
        if (rbase + ro < M) dst[(int64_t)ro * N] = v[r];
      }
    }
  }
  if (GNM) {
    __syncthreads();
    const int gpt = BN / gm.cpg, t = 64 * wave + 32 * lh + l31;                       // thread of this 128-row half
    if (t < gpt && m0 < M) {                                                           // (a 256-row form's second half may start at M: nothing to report, and no slot to write)
      double ds = 0, dq = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        ds += (double)sh[(w * gpt + t) * 2];
        dq += (double)sh[(w * gpt + t) * 2 + 1];
      }
      const double n = 128.0 * gm.cpg, mean = ds / n, m2 = dq - ds * mean;
      const int b = m0 / gm.P, sp = (m0 - b * gm.P) >> 7, splits = gm.P >> 7, g = n0 / gm.cpg + t;
      float* o = gm.ws + (((int64_t)b * gm.G + g) * splits + sp) * 3;
      o[0] = (float)n;
      o[1] = (float)mean;
      o[2] = (float)(m2 > 0 ? m2 : 0);
    }
  }
}

#ifndef RBA_H3_HELPERS_ONLY
// The same accumulators written channel-major: out[b][n][p] for row m = b P + p (NHWC rows in, NCHW out: the mask-feature projection,
// pixel_decoder/msdeformattn.py:362, whose consumer K4 wants [C][pixels]).  A lane's register quad r = 4 q .. 4 q + 3 is four CONSECUTIVE rows
// (pixels) of one column (channel): one 16-byte store along the pixel axis when P % 4 == 0 (quads then never straddle two images).
template <int CT>
__device__ __forceinline__ void h3_epilogue_nchw(const f32x16_t (&accm)[CT], const f32x16_t (&accl)[CT], const float* __restrict__ bias, float* C,
                                                 int M, int N, int P, int m0, int n0, int wave, int l31, int lh) {
  const bool vec = (P & 3) == 0;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = n0 + 32 * j + l31;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = m0 + 32 * wave + 8 * q + 4 * lh;
      if (row >= M) continue;
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaf(accl[j][4 * q + i], 0.00048828125f, accm[j][4 * q + i]) + bv;
      if (vec && row + 3 < M) {
        const int b = row / P, p = row - b * P;
        *reinterpret_cast<f32x4*>(C + ((int64_t)b * N + col) * P + p) = v;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (row + i < M) {
            const int b = (row + i) / P, p = row + i - b * P;
            C[((int64_t)b * N + col) * P + p] = v[i];
          }
      }
    }
  }
}

// ACT: 0 none, 1 exact GELU, 2 ReLU.  CT: 32-column MFMA tiles per wave = tile width / 32 (BN = 32 CT; 128 must be a multiple).
// Tile 128 x BN, four waves, wave w owns rows 32 w .. 32 w + 31 x all BN columns; two workgroups per CU (256 registers a wave).
// PROBE (tune builds only, results wrong): bit 0 no weight loads in the loop, bit 1 no activation loads in the loop, bit 2 no epilogue,
// bit 3 weight fragments read once, bit 4 no activation split, bit 5 no weight ds_write in the loop, bit 6 no barrier in the loop
template <int ACT, int CT, int PROBE = 0, bool TIMING = false, bool RES = false>
__global__ __launch_bounds__(256, 2) void split_linear_h3_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                                const float* __restrict__ bias, float* C, int M, int N,
                                                                int K, int MT, int NT, unsigned long long* dbg = nullptr,
                                                                const float* R = nullptr) {
  unsigned long long tm[4];
  if (TIMING) tm[0] = wall_clock64();
  constexpr int BM = 128, BN = 32 * CT;
  constexpr int SUBW = 4 * BN;                                                     // 16-byte units of one sub-stage image
  constexpr int BLK = 2 * SUBW;                                                    // ... of one 32-wide k block
  constexpr int UPL = BLK / 256;                                                   // units per lane per block
  static_assert(BLK % 256 == 0 && 128 % BN == 0, "tile width");
  __shared__ __attribute__((aligned(16))) u32x4_t lds[2 * BLK];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);                     // XCD-aware: one XCD, one run of tiles (n fastest)
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int NB = K >> 5, S16 = K >> 4;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- weight copy: LDS unit u = tid + 256 q of the block image [g][plane][BN rows][2 slots] <- packed tile (n0 >> 7), sub-stage
  // 2 b + g, plane, rows (n0 & 127) ..  = one workgroup-uniform base (SGPRs) + a constant per q + ONE per-lane 32-bit offset
  constexpr int QSTEP = (CT == 4) ? 256 : 512;                                     // source units between consecutive q
  const char* wbase = reinterpret_cast<const char*>(Wp + (int64_t)(n0 >> 7) * S16 * 512 + (n0 & 127) * 2);
  uint32_t woff;
  {
    const int g = tid / SUBW, rem = tid - g * SUBW, p = rem / (2 * BN), rr = rem - p * (2 * BN);
    woff = (uint32_t)(g * 512 + p * 256 + rr) * 16u;
  }
  // ---- activation rows of this lane: uniform base of the row tile + per-lane 32-bit offset
  int row = m0 + 32 * wave + l31;
  row = (row < M ? row : M - 1) - m0;
  const char* xbase = reinterpret_cast<const char*>(A + (int64_t)m0 * K);
  const uint32_t xoff = ((uint32_t)row * (uint32_t)K + 16u * lh) * 4u;

  f32x16_t accm[CT], accl[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accm[j][r] = accl[j][r] = 0.f;

  // Software pipeline (branch-free: block indices are clamped, the surplus loads of the last blocks are re-reads nobody uses).
  // Weight block c: global -> registers issued during block c - 2, registers -> LDS buffer c & 1 at the head of block c - 1
  // (every wave left that buffer at the barrier that ended block c - 2), read during block c.  Activation block c: global ->
  // registers (set c & 1) issued during block c - 2, right after that set was split.
  u32x4_t wr[UPL];
  f32x4 xr[2][4];
  const int last = NB - 1;
  auto wload = [&](int c, u32x4_t (&r)[UPL]) {
    const char* src = wbase + (int64_t)(c < last ? c : last) * 16384;
#pragma unroll
    for (int q = 0; q < UPL; ++q) r[q] = *reinterpret_cast<const u32x4_t*>(src + q * (QSTEP * 16) + woff);
  };
  auto xload = [&](int c, f32x4 (&r)[4]) {
    const char* src = xbase + (c < last ? c : last) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const f32x4*>(src + q * 16 + xoff);
  };
  wload(0, wr);
  xload(0, xr[0]);
  xload(1, xr[1]);
#pragma unroll
  for (int q = 0; q < UPL; ++q) lds[tid + 256 * q] = wr[q];
  wload(1, wr);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  const int fb = l31 * 2 + (lh ^ ((l31 >> 3) & 1));                                // + g SUBW + (p BN + 32 j) 2
  if (TIMING) tm[1] = wall_clock64();
  auto block = [&](int b, const u32x4_t* img, u32x4_t* nxt, f32x4 (&xc)[4]) {
    // wr holds weight block b + 1: registers -> the other LDS buffer, then refill with block b + 2
    if (!(PROBE & 32)) {
#pragma unroll
      for (int q = 0; q < UPL; ++q) nxt[tid + 256 * q] = wr[q];
    }
    if (!(PROBE & 1)) wload(b + 2, wr);
    f16x8_t ah[2], al[2];
    if (PROBE & 16) {
      ah[0] = __builtin_bit_cast(f16x8_t, xc[0]); al[0] = __builtin_bit_cast(f16x8_t, xc[1]);
      ah[1] = __builtin_bit_cast(f16x8_t, xc[2]); al[1] = __builtin_bit_cast(f16x8_t, xc[3]);
    } else {
      split_h3(xc[0], xc[1], ah[0], al[0]);
      split_h3(xc[2], xc[3], ah[1], al[1]);
    }
    if (!(PROBE & 2)) xload(b + 2, xc);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      __builtin_amdgcn_sched_barrier(0);          // keep the fragment reads of g = 1 below the MFMAs of g = 0 (32 registers, not 64)
      f16x8_t bh[CT], bl[CT];
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const u32x4_t* im = (PROBE & 8) ? lds : img;
        bh[j] = __builtin_bit_cast(f16x8_t, im[g * SUBW + fb + 64 * j]);
        bl[j] = __builtin_bit_cast(f16x8_t, im[g * SUBW + fb + 2 * BN + 64 * j]);
      }
#pragma unroll
      for (int j = 0; j < CT; ++j) accm[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], bh[j], accm[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < CT; ++j) accl[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], bl[j], accl[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < CT; ++j) accl[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g], bh[j], accl[j], 0, 0, 0);
    }
    if (!(PROBE & 64)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  int b = 0;
  if (NB & 1) {                                                                    // odd block count: peel one, then pairs
    block(0, lds, lds + BLK, xr[0]);
    b = 1;
    for (; b < NB; b += 2) {
      block(b, lds + BLK, lds, xr[1]);
      block(b + 1, lds, lds + BLK, xr[0]);
    }
  } else {
    for (; b < NB; b += 2) {
      block(b, lds, lds + BLK, xr[0]);
      block(b + 1, lds + BLK, lds, xr[1]);
    }
  }

  // ---- epilogue: lane holds D[row = 8 (r / 4) + 4 (lane / 32) + r % 4][col = lane % 32] of each 32 x 32 tile
  if (TIMING) tm[2] = wall_clock64();
  h3_epilogue<ACT, CT, PROBE, RES>(accm, accl, bias, C, R, M, N, m0, n0, BM, BN, wave, l31, lh);
  if (TIMING && tid == 0) {
    tm[3] = wall_clock64();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
#pragma unroll
    for (int i = 0; i < 4; ++i) dbg[6 * blockIdx.x + i] = tm[i];
    dbg[6 * blockIdx.x + 4] = xcc;
    dbg[6 * blockIdx.x + 5] = hw;
  }
}

// ---- f16x3, activations staged through LDS.  Why: a lane of the MFMA A operand is a ROW, so straight-to-register loads touch a
// different 128-byte line in every lane and the vector L1 serves them one lane per clock -- the direct kernel above spends more
// cycles in the L1 than in the matrix pipe (ablation: 77 -> 60 us without the activation loads; deeper prefetch changes nothing).
// Here the 128 x 32 fp32 activation block is loaded like the weight block: 8 adjacent lanes read one row's whole 128-byte line,
// registers -> LDS (16-byte chunk c of row r at unit 8 r + (c ^ ((r >> 1) & 7)): the 8-lane row writes and the row-per-lane
// fragment reads are both conflict-free), and each MFMA wave reads its rows' fragments back (4 ds_read_b128 per 32-wide block)
// and splits them in registers.  LDS: 2 x (16 KB activations + 4 BN x 32 B weights).
// CONV: the activation operand is the implicit im2col matrix of a 3 x 3 / stride 1 / pad 1 convolution over NHWC x [B,H,W,Cin]:
// row m = output pixel, k = (tap, channel) with tap = 3 ky + kx, A[m][k] = x[b, y + ky - 1, x + kx - 1, c] or 0 outside the image.
// A 32-wide k block lies inside one tap (Cin % 32 == 0), so a block's loads are the usual 128-byte row reads at a
// workgroup-uniform offset, zeroed per row where the tap leaves the image.
struct ConvShape {
  int H, W, Cin;
};
// GNF (round 4): the A rows are a RAW convolution output whose GroupNorm (+ ReLU) is applied while the tile is stored to LDS -- the arithmetic of
// gn_apply_nhwc_kernel (group_norm.hip): a = gamma * rstd, b = fma(-mean, a, beta), y = fma(x, a, b), max(y, 0) -- so the normalised map is never
// written (pixel_decoder/msdeformattn.py:357-362: `y = output_conv(y)` feeds only `mask_features(y)`).  mr [B][G][2] = (mean, rstd) from
// rba_group_norm_nhwc_stats_f32; every 128-row tile lies inside one image (rows_per_image % 128 == 0, checked by the launcher).
struct GnFold {
  const float* mr;
  const float* gamma;
  const float* beta;
  int G, cpg, relu;
};
template <int ACT, int CT, int PROBE = 0, bool TIMING = false, bool CONV = false, bool RES = false, bool NCHW = false, bool GNF = false, bool GNM = false>
__global__ __launch_bounds__(256, 2) void split_linear_h3l_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                                 const float* __restrict__ bias, float* C, int M, int N,
                                                                 int K, int MT, int NT, unsigned long long* dbg = nullptr,
                                                                 ConvShape cs = ConvShape{0, 0, 0}, const float* R = nullptr, int rows_per_image = 0,
                                                                 GnFold gn = GnFold{nullptr, nullptr, nullptr, 1, 1, 0},
                                                                 GnMoments gm = GnMoments{nullptr, 1, 1, 128}) {
  unsigned long long tm[4];
  if (TIMING) tm[0] = wall_clock64();
  constexpr int BM = 128, BN = 32 * CT;
  constexpr int SUBW = 4 * BN;                                                     // 16-byte units of one weight sub-stage image
  constexpr int BLK = 2 * SUBW;                                                    // ... of one 32-wide k block
  constexpr int UPL = BLK / 256;                                                   // weight units per lane per block
  constexpr int XBLK = BM * 8;                                                     // activation units per block (128 rows x 128 B)
  constexpr int BUF = BLK + XBLK;
  static_assert(BLK % 256 == 0 && 128 % BN == 0, "tile width");
  __shared__ __attribute__((aligned(16))) u32x4_t lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);                     // XCD-aware: one XCD, one run of tiles (n fastest)
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int NB = K >> 5, S16 = K >> 4;
  const int l31 = lane & 31, lh = lane >> 5;

  constexpr int QSTEP = (CT == 4) ? 256 : 512;
  const char* wbase = reinterpret_cast<const char*>(Wp + (int64_t)(n0 >> 7) * S16 * 512 + (n0 & 127) * 2);
  uint32_t woff;
  {
    const int g = tid / SUBW, rem = tid - g * SUBW, p = rem / (2 * BN), rr = rem - p * (2 * BN);
    woff = (uint32_t)(g * 512 + p * 256 + rr) * 16u;
  }
  // activation copy: unit u = tid + 256 q -> row (tid >> 3) + 32 q, chunk tid & 7
  const char* xbase = reinterpret_cast<const char*>(CONV ? A : A + (int64_t)m0 * K);
  uint32_t xoff[4];
  int py[4], px[4];                                                                // CONV: the row's pixel (y, x)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int r = m0 + (tid >> 3) + 32 * q;
    r = r < M ? r : M - 1;
    if (CONV) {
      const int pix = r % (cs.H * cs.W);
      py[q] = pix / cs.W;
      px[q] = pix - py[q] * cs.W;
      xoff[q] = ((uint32_t)r * (uint32_t)cs.Cin + 4u * (tid & 7)) * 4u;
    } else {
      xoff[q] = ((uint32_t)(r - m0) * (uint32_t)K + 4u * (tid & 7)) * 4u;
    }
  }
  const int xdst = BLK + (tid >> 3) * 8 + ((tid & 7) ^ ((tid >> 4) & 7));            // + 256 q   (row = tid >> 3: (row >> 1) & 7)
  // fragment reads of this lane's row
  const int frow = 32 * wave + l31;
  const int fa = BLK + frow * 8, fsw = (frow >> 1) & 7;

  f32x16_t accm[CT], accl[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accm[j][r] = accl[j][r] = 0.f;

  u32x4_t wr[UPL];
  f32x4 xr[4];
  f32x4 gga = {0.f, 0.f, 0.f, 0.f}, gbe = {0.f, 0.f, 0.f, 0.f};                      // GNF: gamma / beta of this thread's four channels of the block in flight
  float gmean = 0.f, grstd = 0.f;
  const float* gmr = GNF ? gn.mr + (int64_t)(m0 / rows_per_image) * gn.G * 2 : nullptr;
  const int last = NB - 1;
  auto gload = [&](int c) {
    const int cc = c < last ? c : last;
    const char* ws = wbase + (int64_t)cc * 16384;
#pragma unroll
    for (int q = 0; q < UPL; ++q) wr[q] = *reinterpret_cast<const u32x4_t*>(ws + q * (QSTEP * 16) + woff);
    if (GNF) {
      const int ch = cc * 32 + 4 * (tid & 7);
      gga = *reinterpret_cast<const f32x4*>(gn.gamma + ch);
      gbe = *reinterpret_cast<const f32x4*>(gn.beta + ch);
      const int g = ch / gn.cpg;
      gmean = gmr[2 * g];
      grstd = gmr[2 * g + 1];
    }
    if (CONV) {
      const int k0 = cc * 32, tap = k0 / cs.Cin, ch0 = k0 - tap * cs.Cin;
      const int ky = tap / 3, dy = ky - 1, dx = tap - 3 * ky - 1;
      const int delta = ((dy * cs.W + dx) * cs.Cin + ch0) * 4;                      // bytes from the row's own pixel, channel 0
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = (unsigned)(py[q] + dy) < (unsigned)cs.H && (unsigned)(px[q] + dx) < (unsigned)cs.W;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xbase + (int64_t)(ok ? delta : ch0 * 4) + xoff[q]);
        xr[q] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    } else {
      const char* xs = xbase + cc * 128;
#pragma unroll
      for (int q = 0; q < 4; ++q) xr[q] = *reinterpret_cast<const f32x4*>(xs + xoff[q]);
    }
  };
  // GNF: the normalised rows (GroupNorm [+ ReLU] applied to the block in registers) go to LDS instead of the raw ones
  f32x4 xst[4];
#if defined(RBA_GNF_DEBUG)
  int gnf_blk = 0;
#endif
  auto lstore = [&](u32x4_t* buf) {
#pragma unroll
    for (int q = 0; q < UPL; ++q) buf[tid + 256 * q] = wr[q];
    if (GNF) {
      f32x4 a, b;
      // NaN must stay NaN through the ReLU (common.h: rba_relu): fmaxf(NaN, 0) is 0.  A NaN / inf anywhere in a group makes the group's statistics NaN,
      // so `poison` (0, or NaN when they are) added after the max keeps every value of such a group loud -- one v_add per value.
      // This arithmetic is compiled WITHOUT packed fp32 in the product (split_linear_gnf.hip): as `v_pk_mul_f32 a[0:1], gamma[0:1], (mean, rstd) op_sel:[0,1]`
      // the coefficient a came out 0 in lanes 48-63 now and then (profiles/r05_gnfold_select.txt).
      const float relu_floor = gn.relu ? 0.f : -INFINITY;
      const float poison = (grstd - grstd) + (gmean - gmean);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = gga[i] * grstd;
        b[i] = fmaf(-gmean, a[i], gbe[i]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if defined(RBA_GNF_FORM) && RBA_GNF_FORM == 1                                        // tools (tune/gnf_forms.hip): the form that goes wrong, for the probe
          const float y = fmaf(xr[q][i], a[i], b[i]);
          xst[q][i] = fmaxf(y, relu_floor) + (y - y);
#else
          xst[q][i] = fmaxf(fmaf(xr[q][i], a[i], b[i]), relu_floor) + poison;
#endif
        }
#if defined(RBA_GNF_DEBUG)
      if (dbg && gnf_blk < NB) {                                                       // [tile][block][thread][40]: xr (16), xst (16), a (4), b (4)
        float* d = reinterpret_cast<float*>(dbg) + (((int64_t)(mt * NT + nt) * NB + gnf_blk) * 256 + tid) * 40;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<f32x4*>(d + 4 * q) = xr[q];
          *reinterpret_cast<f32x4*>(d + 16 + 4 * q) = xst[q];
        }
        *reinterpret_cast<f32x4*>(d + 32) = a;
        *reinterpret_cast<f32x4*>(d + 36) = b;
      }
      ++gnf_blk;
#endif
#pragma unroll
      for (int q = 0; q < 4; ++q) buf[xdst + 256 * q] = __builtin_bit_cast(u32x4_t, xst[q]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) buf[xdst + 256 * q] = __builtin_bit_cast(u32x4_t, xr[q]);
    }
  };
  gload(0);
  lstore(lds);
  gload(1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  const int fb = l31 * 2 + (lh ^ ((l31 >> 3) & 1));
  if (TIMING) tm[1] = wall_clock64();
  for (int b = 0; b < NB; ++b) {
    const u32x4_t* img = lds + (b & 1) * BUF;
    // registers hold block b + 1: -> the other buffer (every wave left it at the barrier that ended block b - 1); refill with b + 2
    if (!(PROBE & 32)) lstore(lds + ((b + 1) & 1) * BUF);
    if (!(PROBE & 1)) gload(b + 2);
    f16x8_t ah[2], al[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 u = __builtin_bit_cast(f32x4, img[fa + ((4 * lh + 2 * g) ^ fsw)]);
      const f32x4 v = __builtin_bit_cast(f32x4, img[fa + ((4 * lh + 2 * g + 1) ^ fsw)]);
      if (PROBE & 16) {
        ah[g] = __builtin_bit_cast(f16x8_t, u);
        al[g] = __builtin_bit_cast(f16x8_t, v);
      } else {
        split_h3(u, v, ah[g], al[g]);
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (PROBE & 128) __builtin_amdgcn_sched_barrier(0);
      f16x8_t bh[CT], bl[CT];
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        bh[j] = __builtin_bit_cast(f16x8_t, img[g * SUBW + fb + 64 * j]);
        bl[j] = __builtin_bit_cast(f16x8_t, img[g * SUBW + fb + 2 * BN + 64 * j]);
      }
#pragma unroll
      for (int j = 0; j < CT; ++j) accm[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], bh[j], accm[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < CT; ++j) accl[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g], bl[j], accl[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < CT; ++j) accl[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g], bh[j], accl[j], 0, 0, 0);
    }
    if (!(PROBE & 64)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  if (TIMING) tm[2] = wall_clock64();
  if (NCHW) h3_epilogue_nchw<CT>(accm, accl, bias, C, M, N, rows_per_image, m0, n0, wave, l31, lh);
  else h3_epilogue<ACT, CT, PROBE, RES, GNM>(accm, accl, bias, C, R, M, N, m0, n0, BM, BN, wave, l31, lh, gm, reinterpret_cast<float*>(lds));
  if (TIMING && tid == 0) {
    tm[3] = wall_clock64();
#pragma unroll
    for (int i = 0; i < 4; ++i) dbg[6 * blockIdx.x + i] = tm[i];
    dbg[6 * blockIdx.x + 4] = dbg[6 * blockIdx.x + 5] = 0;
  }
}

#endif  // RBA_H3_HELPERS_ONLY

// Epilogue of the operand-swapped form (FOUT): the MFMAs ran with the weight fragment as A and the activation fragment as B, so a lane
// holds D^T: row = lane % 32 of its wave's 32 rows, columns n = 8 (r / 4) + 4 (lane / 32) + r % 4 of each 32-wide tile -- four
// consecutive output channels per register quad.  The output is not an fp32 tensor but the NEXT Linear's split A operand (see "PRE"
// in the pipelined kernel): value -> bias -> activation -> (h, l) split, one exchange with the lane that holds the other four
// channels of the 8-channel piece (lane ^ 32), one 16-byte store per piece, 512 contiguous bytes per 32 lanes.  N % 32 == 0.
template <int ACT, int CT>
__device__ __forceinline__ void h3_epilogue_split(const f32x16_t (&accm)[CT], const f32x16_t (&accl)[CT], const float* __restrict__ bias,
                                                  void* out_frag, int M, int N, int m0, int n0, int wave, int l31, int lh) {
  const int row = m0 + 32 * wave + l31;
  const bool full = m0 + 128 <= M;                                                 // uniform
  const int NBo = N >> 5;
  char* base = reinterpret_cast<char*>(out_frag) + ((int64_t)((m0 >> 5) + wave) * NBo) * 4096 + l31 * 16 + lh * 1024;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int nj = n0 + 32 * j;
    if (nj >= N) break;                                                            // uniform: whole 32-wide blocks only
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = nj + 8 * q + 4 * lh;
      const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x2 y0 = (f32x2){accl[j][4 * q], accl[j][4 * q + 1]} * 0.00048828125f + (f32x2){accm[j][4 * q], accm[j][4 * q + 1]};
      f32x2 y1 = (f32x2){accl[j][4 * q + 2], accl[j][4 * q + 3]} * 0.00048828125f + (f32x2){accm[j][4 * q + 2], accm[j][4 * q + 3]};
      y0 = y0 + (f32x2){bv.x, bv.y};
      y1 = y1 + (f32x2){bv.z, bv.w};
      if (ACT == 1) {
        y0 = gelu_erf2(y0);
        y1 = gelu_erf2(y1);
      }
      if (ACT == 2) {
        y0 = (f32x2){rba_relu(y0.x), rba_relu(y0.y)};
        y1 = (f32x2){rba_relu(y1.x), rba_relu(y1.y)};
      }
      uint32_t h0, l0, h1, l1;
      rba_split_f16x2(y0.x, y0.y, h0, l0);
      rba_split_f16x2(y1.x, y1.y, h1, l1);
      // v_permlane32_swap(h, l): lanes 32-63 of h <-> lanes 0-31 of l.  Afterwards a low lane holds (its h, the partner's h) and a high
      // lane (the partner's l, its l): exactly the piece each of them stores, in channel order
      const auto p0 = __builtin_amdgcn_permlane32_swap(h0, l0, false, false), p1 = __builtin_amdgcn_permlane32_swap(h1, l1, false, false);
      const u32x4_t piece = {p0[0], p1[0], p0[1], p1[1]};
      if (full || row < M) *reinterpret_cast<u32x4_t*>(base + (int64_t)((nj >> 5)) * 4096 + (q & 1) * 2048 + (q >> 1) * 512) = piece;
    }
  }
}

#ifndef RBA_H3_HELPERS_ONLY
// ---- f16x3, straight-to-register activations, SOFTWARE-PIPELINED fragment reads (the form used for K > 256 with 128-column
// tiles).  In the kernels above a wave issues its weight fragment reads, waits for LDS, issues 12 MFMAs, reads again, waits again:
// per 32-wide block ~2800 cycles for 768 cycles of MFMA (tools/gemm_h3_timing.py; the second workgroup of the CU fills some of
// it, never all).  Here the work of a block is cut into its four 32-column tiles: while the six MFMAs of column tile j run
// (192 cycles), the four fragment reads of tile j + 1 are in flight into the other half of a two-deep register buffer, the
// activation split of the NEXT block (into a second operand set) is interleaved with the MFMAs of tiles 0-1, and the workgroup
// barrier sits before tile 3 so that the first fragments of the next block are already being read from the other LDS buffer while
// tile 3 computes.  The interleaving is pinned with __builtin_amdgcn_sched_group_barrier (MFMA, DS read, 4 VALU, ...): left to
// itself the compiler re-serialises reads and MFMAs (83 us instead of 69 us on Swin stage-3 fc1).  Register budget: 128
// accumulators + 2 x 16 weight fragments + 2 x 16 activation operands + 16 raw activations + 16 weight staging.
// KS = 2 (single-resident launches: at most one 128 x 128 tile per CU): the workgroup has EIGHT waves, two per SIMD; waves 4-7 are a second
// copy of waves 0-3 that works on the odd 32-wide blocks of K (own weight ring in LDS, same barriers), so that every SIMD has a second
// wave to issue from while the first waits -- what a second workgroup does for the larger launches.  After the loop the odd half's
// accumulators are added to the even half's through LDS (two passes of 64 KiB, fixed order: deterministic) and waves 0-3 run the epilogue.
// Instantiated by the tune library only (cfg 6004 / 6104): fc2 of Swin stage 3 62.6 -> 59.3 us, proj unchanged (26.4 us), and the halves'
// summation order differs from the one-set kernel's -- not worth a second numerical form in the product.
// CONVP (with PRE, two workgroups per CU): the 3 x 3 convolution as an implicit GEMM over the producer's split image of the NHWC input
// (rows = pixels, K = 9 Cin, k = tap * Cin + channel): the four pieces of a lane's row for block (tap, channel block) are the pieces
// of the NEIGHBOUR pixel's row (row + dy W + dx) -- 32 lanes still read one or two contiguous runs -- zeroed where the tap leaves the image.
// RS = 2 (round 4, tune library only -- measured in profiles/r04_k6_rs2.txt): EIGHT waves, two per SIMD, as ONE workgroup of 256 x 128: waves 4-7 are
// a second copy of waves 0-3 that owns the NEXT 128 rows and walks the same k blocks through the SAME weight ring (staged once by all 512 threads), so
// the packed weight crosses the vector L1 once per 256 rows instead of once per 128: operand traffic per MFMA 42.7 -> 32 B/clk/CU at full matrix rate.
template <int ACT, int PROBE = 0, bool TIMING = false, bool RES = false, int OCC = 2, bool PRE = false, bool FOUT = false, int KS = 1,
          bool CONVP = false, int RS = 1, bool GNM = false>
__global__ __launch_bounds__(256 * KS * RS, (KS == 2 || RS == 2) ? 1 : OCC) void split_linear_h3p_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                                 const float* __restrict__ bias, float* C, int M, int N,
                                                                 int K, int MT, int NT, unsigned long long* dbg = nullptr,
                                                                 const float* R = nullptr, ConvShape cs = ConvShape{0, 0, 0}, int stagger = 0,
                                                                 GnMoments gm = GnMoments{nullptr, 1, 1, 128}) {
  static_assert(!GNM || (KS == 1 && !FOUT), "output moments: fp32-row epilogue, one wave set per K");
  unsigned long long tm[4];
  if (TIMING) tm[0] = wall_clock64();
  // experiment (tools): the second workgroup of every CU (the dispatcher hands out workgroups 256 .. 511 after every CU has its first)
  // starts `stagger` ticks of the 100 MHz clock late, so that the two workgroups of a CU reach their epilogues at different times
  if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)stagger) __builtin_amdgcn_s_sleep(8);
  }
  constexpr int CT = 4, BM = 128, BN = 128;
  constexpr int SUBW = 4 * BN, BLK = 2 * SUBW, UPL = BLK / 256;
  __shared__ __attribute__((aligned(16))) u32x4_t lds_all[2 * BLK * KS];

  static_assert(KS * RS <= 2, "K split and row split are alternatives");
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int ks = KS == 2 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;     // which half of the k blocks
  const int rs = RS == 2 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;     // which 128 rows of the 256-row tile
  const int wtid = RS == 2 ? (int)threadIdx.x : tid;                                  // weight staging: all threads that share the ring
  constexpr int UPLW = UPL / RS;
  u32x4_t* const lds = lds_all + ks * 2 * BLK;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = (mt * RS + rs) * BM, n0 = nt * BN;
  const int NB = (K >> 5) / KS, S16 = K >> 4;                                       // NB: blocks THIS wave set walks (K / 32 even when KS = 2)
  const int l31 = lane & 31, lh = lane >> 5;

  const char* wbase = reinterpret_cast<const char*>(Wp + (int64_t)(n0 >> 7) * S16 * 512);
  const uint32_t woff = (uint32_t)wtid * 16u;                                      // the block image is contiguous: unit wtid + 256 RS q
  int row = m0 + 32 * wave + l31;
  row = (row < M ? row : M - 1) - m0;
  const char* xbase = reinterpret_cast<const char*>(A + (int64_t)m0 * K);
  const uint32_t xoff = ((uint32_t)row * (uint32_t)K + 16u * lh) * 4u;

  f32x16_t accm[CT], accl[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accm[j][r] = accl[j][r] = 0.f;

  u32x4_t wr[UPLW];
  constexpr bool TWO = OCC == 2 || KS == 2 || RS == 2;                                        // two waves per SIMD: 256 registers each
  constexpr bool DEEP = !TWO;                                                      // registers to spare: activations two blocks ahead
  constexpr bool DIRECT = PRE && TWO;                                              // pieces straight into the next block's operand set
  f32x4 xr[DEEP ? 2 : 1][4];
  f16x8_t ah[2][2], al[2][2];                                                      // [block parity][g]
  u32x4_t bq[2][4];                                                                // [column-tile parity][h g0, l g0, h g1, l g1]
  const int last = NB - 1;
  auto bix = [&](int c) { return (c < last ? c : last) * KS + ks; };               // global 32-wide block of this set's step c (clamped)
  auto wload = [&](int c) {
    const char* src = wbase + (int64_t)bix(c) * 16384;
#pragma unroll
    for (int q = 0; q < UPLW; ++q) wr[q] = *reinterpret_cast<const u32x4_t*>(src + q * (4096 * RS) + woff);
  };
  // PRE: A is the producer's fragment-ordered, already split image (frag_layout.h): per (32-row group, 32-wide block) four 1 KiB
  // pieces [h g0 | l g0 | h g1 | l g1], each [lane][8 f16] -- one contiguous wave load per operand register quad, no arithmetic here
  const int rgrp = min((m0 >> 5) + wave, ((M + 31) >> 5) - 1);
  const char* fbase = reinterpret_cast<const char*>(A) + ((int64_t)rgrp * (K >> 5)) * 4096 + lane * 16;
  auto xload = [&](int c, f32x4 (&d)[4]) {
    if (PRE) {
      const char* src = fbase + bix(c) * 4096;
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = *reinterpret_cast<const f32x4*>(src + q * 1024);
      return;
    }
    if (PROBE & 2048) {                                                            // ablation: the same bytes as contiguous 1 KiB wave loads
      const char* src = xbase + (uint32_t)(32 * wave) * (uint32_t)K * 4u + bix(c) * 4096 + lane * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = *reinterpret_cast<const f32x4*>(src + q * 1024);
      return;
    }
    const char* src = xbase + bix(c) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] = *reinterpret_cast<const f32x4*>(src + q * 16 + xoff);
  };
  const int fb = l31 * 2 + (lh ^ ((l31 >> 3) & 1));                                // + g SUBW + (p BN + 32 j) 2
  auto bread = [&](const u32x4_t* img, int j, u32x4_t (&d)[4]) {
    d[0] = img[fb + 64 * j];
    d[1] = img[fb + 2 * BN + 64 * j];
    d[2] = img[SUBW + fb + 64 * j];
    d[3] = img[SUBW + fb + 2 * BN + 64 * j];
  };

  // CONVP: this lane's output pixel and the per-tap source row
  int cpy = 0, cpx = 0, crow = 0, cNB = 1, cinv = 0;
  if (CONVP) {
    crow = min(m0 + 32 * wave + l31, M - 1);
    const int pix = crow % (cs.H * cs.W);
    cpy = pix / cs.W;
    cpx = pix - cpy * cs.W;
    cNB = cs.Cin >> 5;
    cinv = (65536 + cNB - 1) / cNB;                                                // block / cNB == (block * cinv) >> 16 for block < 9 * cNB <= 576
  }
  auto xloadset = [&](int c, int p) {
    if (CONVP) {
      const int cc = bix(c);
      const int tap = (cc * cinv) >> 16, cb = cc - tap * cNB;
      const int ky = (tap * 11) >> 5, dy = ky - 1, dx = tap - 3 * ky - 1;              // tap / 3 for tap < 9
      const bool ok = (unsigned)(cpy + dy) < (unsigned)cs.H && (unsigned)(cpx + dx) < (unsigned)cs.W;
      const int rs = ok ? crow + dy * cs.W + dx : crow;
      const char* src = reinterpret_cast<const char*>(A) + ((int64_t)(rs >> 5) * cNB + cb) * 4096 + (lh * 32 + (rs & 31)) * 16;
      const f16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      const f16x8_t v0 = *reinterpret_cast<const f16x8_t*>(src), v1 = *reinterpret_cast<const f16x8_t*>(src + 1024);
      const f16x8_t v2 = *reinterpret_cast<const f16x8_t*>(src + 2048), v3 = *reinterpret_cast<const f16x8_t*>(src + 3072);
      ah[p][0] = ok ? v0 : z;
      al[p][0] = ok ? v1 : z;
      ah[p][1] = ok ? v2 : z;
      al[p][1] = ok ? v3 : z;
      return;
    }
    const char* src = fbase + bix(c) * 4096;
    ah[p][0] = *reinterpret_cast<const f16x8_t*>(src);
    al[p][0] = *reinterpret_cast<const f16x8_t*>(src + 1024);
    ah[p][1] = *reinterpret_cast<const f16x8_t*>(src + 2048);
    al[p][1] = *reinterpret_cast<const f16x8_t*>(src + 3072);
  };
  auto psplit = [&](const f32x4 u, const f32x4 v, f16x8_t& h, f16x8_t& l) {
    if (PRE || (PROBE & 512)) {                                                    // PRE: the pieces ARE the operands (PROBE 512: ablation)
      h = __builtin_bit_cast(f16x8_t, u);
      l = __builtin_bit_cast(f16x8_t, v);
    } else {
      split_h3(u, v, h, l);
    }
  };
  auto pbread = [&](const u32x4_t* img, int j, u32x4_t (&d)[4]) {
    if (!(PROBE & 1024)) bread(img, j, d);
  };
  wload(0);
  if (DIRECT) xloadset(0, 0);
  else xload(0, xr[0]);
#pragma unroll
  for (int q = 0; q < UPLW; ++q) lds[wtid + 256 * RS * q] = wr[q];
  if (!DIRECT) {
    psplit(xr[0][0], xr[0][1], ah[0][0], al[0][0]);
    psplit(xr[0][2], xr[0][3], ah[0][1], al[0][1]);
  }
  wload(1);
  if (DIRECT) {
  } else if (DEEP) {
    xload(1, xr[DEEP ? 1 : 0]);
    xload(2, xr[0]);
  } else {
    xload(1, xr[0]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  bread(lds, 0, bq[0]);
  bread(lds, 1, bq[1]);
  unsigned long long cyc = 0;
  if (TIMING) {
    tm[1] = wall_clock64();
    cyc = __builtin_readcyclecounter();
  }

  // FOUT: operands swapped (D^T = W x^T): same products, same order, the accumulators transposed for h3_epilogue_split
  auto mm = [](const f16x8_t a, const f16x8_t b, const f32x16_t c) {
    return FOUT ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  };
#define RBA_MFMA6(J, Q, P)                                                                                              \
  {                                                                                                                     \
    const f16x8_t bh0 = __builtin_bit_cast(f16x8_t, bq[Q][0]), bl0 = __builtin_bit_cast(f16x8_t, bq[Q][1]);             \
    const f16x8_t bh1 = __builtin_bit_cast(f16x8_t, bq[Q][2]), bl1 = __builtin_bit_cast(f16x8_t, bq[Q][3]);             \
    accm[J] = mm(ah[P][0], bh0, accm[J]);                                                                               \
    accl[J] = mm(ah[P][0], bl0, accl[J]);                                                                               \
    accm[J] = mm(ah[P][1], bh1, accm[J]);                                                                               \
    accl[J] = mm(al[P][0], bh0, accl[J]);                                                                               \
    accl[J] = mm(ah[P][1], bl1, accl[J]);                                                                               \
    accl[J] = mm(al[P][1], bh1, accl[J]);                                                                               \
  }
  // one 32-wide block with activation operands of parity P; cur / nxt = the LDS buffers of this / the next block
// scheduling patterns of one column-tile step: the four fragment reads of the next tile (and, in two of the steps, the split of half
// of the next block's activations: ~18 VALU) go out between the MFMAs
#define RBA_PIPE_MFMA_DSR                                                                                               \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                  \
  }                                                                                                                     \
  __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#define RBA_PIPE_MFMA_DSR_VALU                                                                                          \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                                                  \
  }                                                                                                                     \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                    \
  __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                                                    \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#define RBA_XI(P) (DEEP ? (P) ^ 1 : 0)
#define RBA_BLOCK(B, P, CUR, NXT)                                                                                       \
  {                                                                                                                     \
    if (!(PROBE & 256)) { _Pragma("unroll") for (int q = 0; q < UPLW; ++q)(NXT)[wtid + 256 * RS * q] = wr[q]; }        \
    if (!(PROBE & 1)) wload((B) + 2);                                                                                   \
    if (DIRECT) xloadset((B) + 1, (P) ^ 1);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    RBA_MFMA6(0, 0, P)                                                                                                  \
    pbread(CUR, 2, bq[0]);                                                                                              \
    if (!DIRECT) psplit(xr[RBA_XI(P)][0], xr[RBA_XI(P)][1], ah[(P) ^ 1][0], al[(P) ^ 1][0]);                                                           \
    RBA_PIPE_MFMA_DSR_VALU                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    RBA_MFMA6(1, 1, P)                                                                                                  \
    pbread(CUR, 3, bq[1]);                                                                                              \
    if (!DIRECT) psplit(xr[RBA_XI(P)][2], xr[RBA_XI(P)][3], ah[(P) ^ 1][1], al[(P) ^ 1][1]);                                                           \
    RBA_PIPE_MFMA_DSR_VALU                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (!DIRECT && !(PROBE & 2)) xload((B) + (DEEP ? 3 : 2), xr[RBA_XI(P)]);                                                                                 \
    RBA_MFMA6(2, 0, P)                                                                                                  \
    if (!(PROBE & 64)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    RBA_MFMA6(3, 1, P)                                                                                                  \
    pbread(NXT, 0, bq[0]);                                                                                              \
    RBA_PIPE_MFMA_DSR                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    pbread(NXT, 1, bq[1]);                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  }
  int b = 0;
  if (NB & 1) {
    RBA_BLOCK(0, 0, lds, lds + BLK)
    for (b = 1; b < NB; b += 2) {
      RBA_BLOCK(b, 1, lds + BLK, lds)
      RBA_BLOCK(b + 1, 0, lds, lds + BLK)
    }
  } else {
    for (; b < NB; b += 2) {
      RBA_BLOCK(b, 0, lds, lds + BLK)
      RBA_BLOCK(b + 1, 1, lds + BLK, lds)
    }
  }
#undef RBA_BLOCK
#undef RBA_XI
#undef RBA_PIPE_MFMA_DSR
#undef RBA_PIPE_MFMA_DSR_VALU
#undef RBA_MFMA6

  if (TIMING) {
    tm[2] = wall_clock64();
    cyc = __builtin_readcyclecounter() - cyc;
  }
  if (KS == 2) {
    // odd-block half -> even-block half, 16 units per thread per pass: [group of four registers][thread]
    f32x4* red = reinterpret_cast<f32x4*>(lds_all);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();                                                             // ring (pass 0) / previous pass read
      if (ks == 1) {
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x16_t& a = pass ? accl[j] : accm[j];
            red[(j * 4 + q) * 256 + tid] = (f32x4){a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
          }
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = red[(j * 4 + q) * 256 + tid];
            f32x16_t& a = pass ? accl[j] : accm[j];
            a[4 * q] += v.x;
            a[4 * q + 1] += v.y;
            a[4 * q + 2] += v.z;
            a[4 * q + 3] += v.w;
          }
      }
    }
    if (ks == 1) return;
  }
  if (FOUT) h3_epilogue_split<ACT, CT>(accm, accl, bias, C, M, N, m0, n0, wave, l31, lh);
  else h3_epilogue<ACT, CT, PROBE, RES, GNM>(accm, accl, bias, C, R, M, N, m0, n0, BM, BN, wave, l31, lh, gm, reinterpret_cast<float*>(lds_all) + rs * 512);
  if (TIMING && tid == 0) {
    tm[3] = wall_clock64();
#pragma unroll
    for (int i = 0; i < 4; ++i) dbg[6 * blockIdx.x + i] = tm[i];
    dbg[6 * blockIdx.x + 4] = cyc;                                                 // shader clocks spent in the k loop
    dbg[6 * blockIdx.x + 5] = 0;
  }
}

template <int ACT, int PROBE = 0, int OCC = 2, int KS = 1>
int launch_h3p(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t stream) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 127) / 128;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3p_kernel<ACT, PROBE, false, false, OCC, false, false, KS>), dim3((unsigned)(MT * NT)), dim3(256 * KS), 0, stream, x,
                     wp, bias, out, (int)M, N, K, (int)MT, NT, nullptr);
  return 0;
}

// At most one workgroup per CU (<= 256 tiles): the OCC = 1 build (no second workgroup to share the register file with, so the
// activations are prefetched two blocks ahead): -4 ... -9 % on Swin stage-3/4 proj and fc2 (profiles/r02_k6_h3p_ablation.txt).
// 2 (default since round 3): every launch runs the 256-register build, so that a single-resident launch (<= 256 tiles: stage-3 proj / fc2)
// leaves half of every SIMD's register file to ANOTHER stream's kernel; 1: such launches run the 364-register OCC = 1 build with its deeper
// prefetch -- 4-9 % faster alone, but it monopolises the CU: 128.8 -> 132.7 images/s with three streams, single stream unchanged
// (profiles/r03_k6_occ.txt)
RBA_KNOB_EXTERN(rba_k6_occ, 2);
inline bool h3p_single_resident(int64_t M, int N) { return rba_k6_occ == 1 && ((M + 127) / 128) * ((N + 127) / 128) <= 256; }

inline int launch_h3p_act(int act, const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t st) {
  if (h3p_single_resident(M, N)) {
    if (act == 1) return launch_h3p<1, 0, 1>(x, wp, bias, out, M, N, K, st);
    if (act == 2) return launch_h3p<2, 0, 1>(x, wp, bias, out, M, N, K, st);
    return launch_h3p<0, 0, 1>(x, wp, bias, out, M, N, K, st);
  }
  if (act == 1) return launch_h3p<1>(x, wp, bias, out, M, N, K, st);
  if (act == 2) return launch_h3p<2>(x, wp, bias, out, M, N, K, st);
  return launch_h3p<0>(x, wp, bias, out, M, N, K, st);
}

// Round 4: the 256 x 128 form (RS = 2: eight waves, one weight ring for two 128-row halves -- see the kernel) for launches whose tiles fill the
// chip evenly.  Measured (profiles/r04_k6_rs2.txt, isolated launches, bit-identical results): 1.10-1.15x where the 256 x 128 tiles make whole
// rounds of the 256 CUs or one round that is at least three quarters full (Swin-B stage-3 fc1 69 -> 62 us, stage-4 qkv / fc1 45 -> 40 / 55 -> 50 us,
// Swin-L stage-3 fc1 / fc2 130 -> 117 / 126 -> 110 us), 0.81-0.96x where they leave CUs idle (<= 128 tiles: stage-3 proj / fc2 of Swin-B) or end
// in a round that is less than three quarters full (1.5 rounds: Swin-B stage-3 qkv).  rba_k6_rs (tools): 0 = this rule, 1 = never, 2 = whenever there
// are at least 160 tiles of 256 x 128.  Only for K >= rba_k6_rs_min_k = 512: the short-K launches of Swin stages 1-2 (K = 128 / 256, 1 000-3 000 tiles) are
// bound by HBM and by their epilogues, and IN THE NETWORK (cold operands) two independent 128 x 128 workgroups per CU overlap one tile's stores with the
// other's loads better than one 8-wave workgroup: per-dispatch rocprofv3 times of one image, stage-1 qkv 72-77 -> 65 us, stage-2 qkv 52-57 -> 47-51,
// stage-2 fc1 83 -> 76-77 (profiles/r04_k6_rs2_min_k.txt; same-box bench A/B +0.7 % single stream, +0.2 % three streams).
RBA_KNOB_EXTERN(rba_k6_rs, 0);
RBA_KNOB_EXTERN(rba_k6_rs_min_k, 512);
// The library's one piece of caller-set state (rba_set_concurrent_streams, include/rba_hip.h): how many streams of this process launch forwards concurrently.
// Not exported; read by the two launch-geometry rules below, written by that entry point only.
extern "C" int rba_concurrent_streams_hint;
inline bool h3p_multi_stream() { return rba_k6_rs == 3 || (rba_k6_rs == 0 && rba_concurrent_streams_hint >= 2); }
inline bool h3p_use_rs2(int64_t M, int N, int K) {
  if (rba_k6_rs == 1 || K < rba_k6_rs_min_k) return false;
  const int64_t t = ((M + 255) / 256) * ((N + 127) / 128);
  if (h3p_multi_stream()) return t >= 64;                                 // tools: also the half-chip launches (128 tiles: stage-3 proj / fc2 of Swin-B)
  if (t < 160) return false;
  if (rba_k6_rs == 2) return true;
  const int64_t last = t % 256;                                       // workgroups in the last round of the 256 CUs (0 = full)
  return last == 0 || last >= 192;
}

// Round 4 (late): the K-split 8-wave form (KS = 2, see the kernel) for the single-resident launches with a LONG k loop -- exactly one 128 x 128 tile per CU
// (129 ... 256 tiles) and K >= 1024: Swin-B stage-3 fc2 (8192 x 512 x 2048) 60.6 -> 55.9 us (tools/k6_ks2_ab.py; proj, K = 512: no gain, stays).  Every SIMD gets
// a second wave without doubling the A traffic (64-column sub-tiles) or halving the workgroup count (256 x 128 tiles).  The two halves of K are summed in a
// fixed order (even blocks, odd blocks, then even + odd): deterministic, but not the one-set kernel's order -- results differ from it in the last bits
// (tests/test_kernels_gpu.py::test_split_linear_k_split_form holds both to the same fp64 bound).  Not under several concurrent streams (rba_k6_rs == 3: there
// these launches take whole CUs with the 256 x 128 form).  rba_k6_ks (tools / tests): 0 = this rule, 1 = never, 2 = wherever the form is legal.
RBA_KNOB_EXTERN(rba_k6_ks, 0);
inline bool h3p_use_ks2(int64_t M, int N, int K) {
  if (rba_k6_ks == 1 || (K & 63) || K < 128) return false;
  if (rba_k6_ks == 2) return true;
  const int64_t t = ((M + 127) / 128) * ((N + 127) / 128);
  return !h3p_multi_stream() && t > 128 && t <= 256 && K >= 1024;
}

// A operand = the producer's split fragment image (PRE); residual may be null
template <int ACT, bool RES, int OCC>
int launch_h3p_pre(const void* xf, const u32x4_t* wp, const float* bias, const float* res, float* out, int64_t M, int N, int K,
                   hipStream_t st) {
  const int NT = (N + 127) / 128;
  if (OCC == 2 && h3p_use_ks2(M, N, K)) {
    const int64_t MT1 = (M + 127) / 128;
    if (MT1 * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((split_linear_h3p_kernel<ACT, 0, false, RES, 2, true, false, 2>), dim3((unsigned)(MT1 * NT)), dim3(512), 0, st,
                       reinterpret_cast<const float*>(xf), wp, bias, out, (int)M, N, K, (int)MT1, NT, nullptr, res, ConvShape{0, 0, 0}, 0);
    return 0;
  }
  if (OCC == 2 && h3p_use_rs2(M, N, K)) {
    const int64_t MT2 = (M + 255) / 256;
    hipLaunchKernelGGL((split_linear_h3p_kernel<ACT, 0, false, RES, 2, true, false, 1, false, 2>), dim3((unsigned)(MT2 * NT)), dim3(512), 0, st,
                       reinterpret_cast<const float*>(xf), wp, bias, out, (int)M, N, K, (int)MT2, NT, nullptr, res, ConvShape{0, 0, 0}, 0);
    return 0;
  }
  const int64_t MT = (M + 127) / 128;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3p_kernel<ACT, 0, false, RES, OCC, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, st,
                     reinterpret_cast<const float*>(xf), wp, bias, out, (int)M, N, K, (int)MT, NT, nullptr, res, ConvShape{0, 0, 0}, rba_k6_stagger);
  return 0;
}
// out = the split fragment image of act(x W^T + bias) (FOUT); x either fp32 rows or a split image (PRE)
template <int ACT, bool PRE, int OCC>
int launch_h3p_fout(const void* x, const u32x4_t* wp, const float* bias, void* out_frag, int64_t M, int N, int K, hipStream_t st) {
  const int NT = (N + 127) / 128;
  if (OCC == 2 && PRE && h3p_use_rs2(M, N, K)) {
    const int64_t MT2 = (M + 255) / 256;
    hipLaunchKernelGGL((split_linear_h3p_kernel<ACT, 0, false, false, 2, true, true, 1, false, 2>), dim3((unsigned)(MT2 * NT)), dim3(512), 0, st,
                       reinterpret_cast<const float*>(x), wp, bias, reinterpret_cast<float*>(out_frag), (int)M, N, K, (int)MT2, NT, nullptr, nullptr,
                       ConvShape{0, 0, 0}, 0);
    return 0;
  }
  const int64_t MT = (M + 127) / 128;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3p_kernel<ACT, 0, false, false, OCC, PRE, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, st,
                     reinterpret_cast<const float*>(x), wp, bias, reinterpret_cast<float*>(out_frag), (int)M, N, K, (int)MT, NT, nullptr,
                     nullptr, ConvShape{0, 0, 0}, rba_k6_stagger);
  return 0;
}
// the same convolution leaving the GroupNorm moments of its output (GNM, see h3_epilogue)
inline int launch_h3p_conv_pre_gnm(const void* xf, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int H, int W, int Cin,
                                   const GnMoments gm, hipStream_t st) {
  const int NT = (N + 127) / 128;
  if ((M & 255) == 0 && h3p_use_rs2(M, N, 9 * Cin)) {                    // moments are per 128-row tile: the 256-row form only where both halves are whole tiles
    const int64_t MT2 = (M + 255) / 256;
    if (MT2 * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((split_linear_h3p_kernel<0, 0, false, false, 2, true, false, 1, true, 2, true>), dim3((unsigned)(MT2 * NT)), dim3(512), 0, st,
                       reinterpret_cast<const float*>(xf), wp, bias, out, (int)M, N, 9 * Cin, (int)MT2, NT, nullptr, nullptr, ConvShape{H, W, Cin}, 0, gm);
    return 0;
  }
  const int64_t MT = (M + 127) / 128;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3p_kernel<0, 0, false, false, 2, true, false, 1, true, 1, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, st,
                     reinterpret_cast<const float*>(xf), wp, bias, out, (int)M, N, 9 * Cin, (int)MT, NT, nullptr, nullptr, ConvShape{H, W, Cin}, 0, gm);
  return 0;
}
// fp32 rows in, fp32 rows out, no activation, 128-column tiles, plus the output's GroupNorm moments (the FPN's lateral 1 x 1 convolutions, K <= 256)
inline int launch_h3l_gnm(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, const GnMoments gm, hipStream_t stream) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 127) / 128;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3l_kernel<0, 4, 0, false, false, false, false, false, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, stream, x, wp, bias, out,
                     (int)M, N, K, (int)MT, NT, nullptr, ConvShape{0, 0, 0}, nullptr, 0, GnFold{nullptr, nullptr, nullptr, 1, 1, 0}, gm);
  return 0;
}

// 3 x 3 convolution (pad 1) over the split image of NHWC activations: M = B H W output pixels, K = 9 Cin (two workgroups per CU)
inline int launch_h3p_conv_pre(const void* xf, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int H, int W, int Cin,
                               hipStream_t st) {
  const int NT = (N + 127) / 128;
  if (h3p_use_rs2(M, N, 9 * Cin)) {
    const int64_t MT2 = (M + 255) / 256;
    if (MT2 * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((split_linear_h3p_kernel<0, 0, false, false, 2, true, false, 1, true, 2>), dim3((unsigned)(MT2 * NT)), dim3(512), 0, st,
                       reinterpret_cast<const float*>(xf), wp, bias, out, (int)M, N, 9 * Cin, (int)MT2, NT, nullptr, nullptr, ConvShape{H, W, Cin});
    return 0;
  }
  const int64_t MT = (M + 127) / 128;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3p_kernel<0, 0, false, false, 2, true, false, 1, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, st,
                     reinterpret_cast<const float*>(xf), wp, bias, out, (int)M, N, 9 * Cin, (int)MT, NT, nullptr, nullptr, ConvShape{H, W, Cin});
  return 0;
}
template <int OCC>
int launch_h3p_pre_act(int act, const void* xf, const u32x4_t* wp, const float* bias, const float* res, float* out, int64_t M, int N,
                       int K, hipStream_t st) {
  if (res) return launch_h3p_pre<0, true, OCC>(xf, wp, bias, res, out, M, N, K, st);
  if (act == 1) return launch_h3p_pre<1, false, OCC>(xf, wp, bias, nullptr, out, M, N, K, st);
  if (act == 2) return launch_h3p_pre<2, false, OCC>(xf, wp, bias, nullptr, out, M, N, K, st);
  return launch_h3p_pre<0, false, OCC>(xf, wp, bias, nullptr, out, M, N, K, st);
}

template <int ACT, int CT, int PROBE = 0>
int launch_h3l(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t stream) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 32 * CT - 1) / (32 * CT);
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3l_kernel<ACT, CT, PROBE>), dim3((unsigned)(MT * NT)), dim3(256), 0, stream, x, wp, bias, out, (int)M, N,
                     K, (int)MT, NT, nullptr);
  return 0;
}

// 3 x 3 convolution (pad 1) over NHWC activations as an implicit GEMM: M = B H W output pixels, K = 9 Cin
template <int CT>
int launch_h3l_conv(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int H, int W, int Cin,
                    hipStream_t stream) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 32 * CT - 1) / (32 * CT);
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3l_kernel<0, CT, 0, false, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, stream, x, wp, bias, out,
                     (int)M, N, 9 * Cin, (int)MT, NT, nullptr, ConvShape{H, W, Cin});
  return 0;
}

// x [B P, K] fp32 rows -> out [B, N, P] (no activation): the LDS-staged kernel with the channel-major epilogue
template <int CT>
int launch_h3l_nchw(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, int P, hipStream_t stream) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 32 * CT - 1) / (32 * CT);
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3l_kernel<0, CT, 0, false, false, false, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, stream, x, wp, bias, out,
                     (int)M, N, K, (int)MT, NT, nullptr, ConvShape{0, 0, 0}, nullptr, P);
  return 0;
}

// the same with the GroupNorm (+ ReLU) of the input rows folded into the A load (GNF)
template <int CT>
int launch_h3l_nchw_gn(const float* x, const GnFold gn, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, int P, hipStream_t stream) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 32 * CT - 1) / (32 * CT);
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3l_kernel<0, CT, 0, false, false, false, true, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, stream, x, wp, bias, out,
                     (int)M, N, K, (int)MT, NT, nullptr, ConvShape{0, 0, 0}, nullptr, P, gn);
  return 0;
}

template <int CT>
int launch_h3l_act(int act, const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t st) {
  if (act == 1) return launch_h3l<1, CT>(x, wp, bias, out, M, N, K, st);
  if (act == 2) return launch_h3l<2, CT>(x, wp, bias, out, M, N, K, st);
  return launch_h3l<0, CT>(x, wp, bias, out, M, N, K, st);
}

template <int ACT, int CT, int PROBE = 0>
int launch_h3(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t stream) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 32 * CT - 1) / (32 * CT);
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3_kernel<ACT, CT, PROBE>), dim3((unsigned)(MT * NT)), dim3(256), 0, stream, x, wp, bias, out, (int)M, N,
                     K, (int)MT, NT);
  return 0;
}

template <int CT>
int launch_h3_act(int act, const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t st) {
  if (act == 1) return launch_h3<1, CT>(x, wp, bias, out, M, N, K, st);
  if (act == 2) return launch_h3<2, CT>(x, wp, bias, out, M, N, K, st);
  return launch_h3<0, CT>(x, wp, bias, out, M, N, K, st);
}

// out = (residual + x W^T) + bias  (no activation): the residual forms of the three kernels
template <int CT>
int launch_h3_res(const float* x, const u32x4_t* wp, const float* bias, const float* res, float* out, int64_t M, int N, int K, hipStream_t st) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 32 * CT - 1) / (32 * CT);
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3_kernel<0, CT, 0, false, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, st, x, wp, bias, out, (int)M, N, K,
                     (int)MT, NT, nullptr, res);
  return 0;
}
template <int CT>
int launch_h3l_res(const float* x, const u32x4_t* wp, const float* bias, const float* res, float* out, int64_t M, int N, int K, hipStream_t st) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 32 * CT - 1) / (32 * CT);
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((split_linear_h3l_kernel<0, CT, 0, false, false, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, st, x, wp, bias, out, (int)M,
                     N, K, (int)MT, NT, nullptr, ConvShape{0, 0, 0}, res);
  return 0;
}
inline int launch_h3p_res(const float* x, const u32x4_t* wp, const float* bias, const float* res, float* out, int64_t M, int N, int K,
                          hipStream_t st) {
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 127) / 128;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  if (rba_k6_occ == 1 && MT * NT <= 256)
    hipLaunchKernelGGL((split_linear_h3p_kernel<0, 0, false, true, 1>), dim3((unsigned)(MT * NT)), dim3(256), 0, st, x, wp, bias, out, (int)M,
                       N, K, (int)MT, NT, nullptr, res);
  else
    hipLaunchKernelGGL((split_linear_h3p_kernel<0, 0, false, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, st, x, wp, bias, out, (int)M, N,
                       K, (int)MT, NT, nullptr, res);
  return 0;
}

#endif  // RBA_H3_HELPERS_ONLY

}  // namespace
