// fp32-accurate Linear on the bf16 matrix pipe ("bf16x6"), for the backbone's token GEMMs
// (reference: the nn.Linear calls of backbone/swin.py:44-71 (Mlp), :131-171 (qkv / proj), :319-343 (PatchMerging)).
//
// The exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) peaks at 157 TFLOP/s on MI355X; the bf16 MFMA at 2.5 PFLOP/s, 16x more.
// Every fp32 number is the exact sum of three bf16 numbers  a = a0 + a1 + a2  (8 + 8 + 8 significand bits,
// a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)), bf16 x bf16 products are exact in fp32, and
//     a * w = a0 w0 + (a0 w1 + a1 w0) + (a0 w2 + a2 w0 + a1 w1) + O(2^-24 |a w|)
// so six bf16 MFMAs accumulated in fp32 reproduce the fp32 product to below one fp32 rounding (the three dropped terms are
// <= 2^-25 relative; measured against fp64 the result is as accurate as the fp32 GEMM it replaces), at 16/6 = 2.7x the
// fp32-MFMA rate.  Weights are split once per weight load into three bf16 planes [3][N][K]; activations are split on the
// fly while the block stages its A tile into LDS (11 VALU per two elements, in the shadow of the MFMAs).
//
// Tiling (both kernels): workgroup = 4 waves, 128 x 128 output tile; wave = 64 x 64 = 2 x 2 tiles of
// v_mfma_f32_32x32x16_bf16 (24 MFMAs per 16-wide k step from 6 + 6 operand fragments: half the LDS operand traffic per MFMA
// of a plain bf16 GEMM of the same tile); blockIdx is remapped so that the workgroups sharing an A row-tile sit on one XCD's
// L2; two workgroups per CU.
//   split_linear_short_kernel (K <= 256): 32-wide k stages, one LDS stage (80-byte row stride), next stage's global loads
//     issued into registers before the current stage's MFMAs, two barriers per stage.  Fewest stages for the short-K GEMMs.
//   split_linear_pipe_kernel  (K > 256):  16-wide k stages, three swizzled LDS stage buffers, one mid-stage barrier, global
//     loads two stages ahead in two rotating register sets (all unconditional: see the note in the kernel).
// Measured (profiles/r01_split_linear.txt): 1.0-1.2x hipBLASLt's fp32 GEMM on the Swin token shapes, with the exact GELU fused
// into the epilogue (0.5 x (1 + erf(x / sqrt 2))) fc1 + GELU is 1.25x.  MFMAs alone would take 45 % of the kernel time:
// the rest is in-order issue of the staging work (LDS stores 22 %, VALU split 18 % of wave time) and barrier skew.
#include <stdlib.h>

#include "common.h"
#include "../../include/rba_hip.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));   // native vectors: arrays of HIP's uint4 class are not promoted to registers

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROWQ = 5;                       // uint4 (16 B) units per LDS row: 4 used (32 bf16) + 1 pad -> 80 B stride

__device__ __forceinline__ uint32_t pack_bf16(float x0, float x1) {           // rne; lowers to v_cvt_pk_bf16_f32
  bf16x2_t v = {(__bf16)x0, (__bf16)x1};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo_as_f32(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_as_f32(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }

// Exact-form GELU 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU default, swin.py:51) with erf from Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, the size of fp32 erff's own rounding in this expression): 14 VALU instead of ocml erff's two-branch
// ~45, which cost 20 % of the fc1 kernel when every lane evaluates 64 of them in the epilogue.
__device__ __forceinline__ float gelu_erf(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = p * t * __expf(-x * x);                       // 1 - erf(|x| / sqrt 2)
  const float one_plus_erf = v >= 0.f ? 2.0f - e : e;
  return 0.5f * v * one_plus_erf;
}

// x (4 floats) -> three planes of 4 bf16 (2 dwords each)
__device__ __forceinline__ void split4(const float4 x, uint2& p0, uint2& p1, uint2& p2) {
  const uint32_t a01 = pack_bf16(x.x, x.y), a23 = pack_bf16(x.z, x.w);
  const float r0 = x.x - lo_as_f32(a01), r1 = x.y - hi_as_f32(a01), r2 = x.z - lo_as_f32(a23), r3 = x.w - hi_as_f32(a23);
  const uint32_t b01 = pack_bf16(r0, r1), b23 = pack_bf16(r2, r3);
  const float s0 = r0 - lo_as_f32(b01), s1 = r1 - hi_as_f32(b01), s2 = r2 - lo_as_f32(b23), s3 = r3 - hi_as_f32(b23);
  p0 = make_uint2(a01, a23);
  p1 = make_uint2(b01, b23);
  p2 = make_uint2(pack_bf16(s0, s1), pack_bf16(s2, s3));
}

// Weight planes in global memory are packed as the LDS image of the pipelined kernels, one 12 KB block per (128-row n tile,
// 16-wide k stage):  [N/128][K/16][3 planes][128 rows][2 half-slots][8] bf16, the 8-element half h of row r in half-slot
// h ^ ((r >> 3) & 1).  A workgroup stages one block with three fully coalesced 1 KB-per-wave loads per thread and stores
// (or DMAs) it to LDS at the same linear offset; a row-major [N][K] layout costs 32 half-used cache lines per wave-load,
// which made the texture-address unit the bottleneck of the first version (2x slower).
// Epilogue of one wave's 64 x 64 tile: lane holds D[row = 8 (r / 4) + 4 (lane / 32) + r % 4][col = lane % 32] of each 32 x 32
// MFMA tile.  The 16 values of a tile are formed in 16 distinct registers and stored back to back (a wave-store covers two
// 128-byte row segments); interior tiles take a branch-free path.  (A store per `if (row < M)` block made the compiler put
// s_waitcnt vmcnt(0) between consecutive stores -- the value register was reused -- serialising 64 memory round trips.)
template <int ACT>
__device__ __forceinline__ void store_tile(const f32x16_t (&acc)[2][2], const float* __restrict__ bias, float* __restrict__ C,
                                           int M, int N, int row0, int col0, bool interior, int l31, int lh) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + 32 * j + l31;
    const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16_t v = acc[i][j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] += bv;
        if (ACT == 1) v[r] = gelu_erf(v[r]);
        if (ACT == 2) v[r] = fmaxf(v[r], 0.f);
      }
      const int rbase = row0 + 32 * i + 4 * lh;
      float* dst = C + (int64_t)rbase * N + col;
      if (interior) {
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
  }
}

// Transposed epilogue (TRANS kernels): the accumulator rows are weight rows n, the columns GEMM rows m = b * ldo + p;
// value (n, m) goes to C[(b * N + n) * ldo + p] (NCHW when the GEMM rows are NHWC pixels, ldo = H * W).
template <int ACT>
__device__ __forceinline__ void store_tile_transposed(const f32x16_t (&acc)[2][2], const float* __restrict__ bias,
                                                      float* __restrict__ C, int M, int N, int nrow0, int mcol0, int ldo, int l31,
                                                      int lh) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = mcol0 + 32 * j + l31;
    const int b = m / ldo, p = m - b * ldo;
    float* dst = C + (int64_t)b * N * ldo + p;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16_t v = acc[i][j];
      const int nbase = nrow0 + 32 * i + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nbase + 8 * (r >> 2) + (r & 3);
        v[r] += (bias && n < N) ? bias[n] : 0.f;
        if (ACT == 1) v[r] = gelu_erf(v[r]);
        if (ACT == 2) v[r] = fmaxf(v[r], 0.f);
      }
      if (m < M) {
        if (nrow0 + 64 <= N) {                                     // wave-uniform: branch-free stores (see store_tile)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[(int64_t)(nbase + 8 * (r >> 2) + (r & 3)) * ldo] = v[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = nbase + 8 * (r >> 2) + (r & 3);
            if (n < N) dst[(int64_t)n * ldo] = v[r];
          }
        }
      }
    }
  }
}

__global__ void split_weight_kernel(const float* __restrict__ w, u32x4_t* __restrict__ packed, int N, int K) {
  const int Kc = K >> 3, S = K >> 4;                              // 8-element chunks per row, stages
  const int Np = (N + BN - 1) / BN * BN;                          // rows N..Np-1 of the last tile are zero
  const int64_t total = (int64_t)Np * Kc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / Kc), c = (int)(i - (int64_t)n * Kc);
    uint2 p0 = make_uint2(0u, 0u), p1 = p0, p2 = p0, q0 = p0, q1 = p0, q2 = p0;
    if (n < N) {
      const float4* src = reinterpret_cast<const float4*>(w + (int64_t)n * K + c * 8);
      split4(src[0], p0, p1, p2);
      split4(src[1], q0, q1, q2);
    }
    const int nt = n >> 7, r = n & 127, st = c >> 1, slot = (c & 1) ^ ((r >> 3) & 1);
    u32x4_t* dst = packed + ((int64_t)nt * S + st) * 768 + r * 2 + slot;
    dst[0] = (u32x4_t){p0.x, p0.y, q0.x, q0.y};
    dst[256] = (u32x4_t){p1.x, p1.y, q1.x, q1.y};
    dst[512] = (u32x4_t){p2.x, p2.y, q2.x, q2.y};
  }
}

// act: 0 none, 1 exact GELU (0.5 x (1 + erf(x / sqrt 2)), nn.GELU default, swin.py:51), 2 ReLU
template <int ACT>
__global__ __launch_bounds__(256) void split_linear_short_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ C, int M, int N,
                                                           int K, int MT, int NT) {
  __shared__ u32x4_t As[3][BM][ROWQ];
  __shared__ u32x4_t Ws[3][BN][ROWQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: hardware deals consecutive workgroups round-robin to the 8 XCDs; give each XCD a contiguous run
  // of logical tiles (n fastest) so the NT column tiles that re-read one A row-tile hit the same L2.
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- staging maps
  const int a_c4 = tid & 7, a_r = tid >> 3;                       // A: rows a_r + 32 i (i < 4), floats 4 a_c4 .. +3
  const float* a_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = m0 + a_r + 32 * i;
    r = r < M ? r : M - 1;
    a_src[i] = A + (int64_t)r * K + a_c4 * 4;
  }
  // W: the 32-wide stage is two consecutive packed 12 KB blocks = 1536 x 16 B, six per thread, linear index tid + 256 j
  const u32x4_t* w_src = Wp + (int64_t)nt * (K >> 4) * 768 + tid;
  int w_dst[6];                                                   // element index into Ws viewed as [3][BN][ROWQ]
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int lin = tid + 256 * j, blk = lin / 768, within = lin - blk * 768;
    const int p = within >> 8, r = (within & 255) >> 1, slot = within & 1;
    w_dst[j] = (p * BN + r) * ROWQ + 2 * blk + (slot ^ ((r >> 3) & 1));
  }

  f32x4 pa[4];
  u32x4_t pw[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) pa[i] = *reinterpret_cast<const f32x4*>(a_src[i]);
#pragma unroll
  for (int j = 0; j < 6; ++j) pw[j] = w_src[256 * j];

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (k0) __syncthreads();                                     // previous stage's fragments are consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2 p0, p1, p2;
      split4(make_float4(pa[i].x, pa[i].y, pa[i].z, pa[i].w), p0, p1, p2);
      const int row = a_r + 32 * i;
      reinterpret_cast<uint2*>(&As[0][row][0])[a_c4] = p0;
      reinterpret_cast<uint2*>(&As[1][row][0])[a_c4] = p1;
      reinterpret_cast<uint2*>(&As[2][row][0])[a_c4] = p2;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) (&Ws[0][0][0])[w_dst[j]] = pw[j];
    __syncthreads();
    if (k0 + BK < K) {                             // prefetch the next stage into registers
      const int kn = k0 + BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) pa[i] = *reinterpret_cast<const f32x4*>(a_src[i] + kn);
#pragma unroll
      for (int j = 0; j < 6; ++j) pw[j] = w_src[(kn >> 4) * 768 + 256 * j];
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[2][3], b[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[t][p] = __builtin_bit_cast(bf16x8_t, As[p][64 * wm + 32 * t + l31][2 * ks + lh]);
          b[t][p] = __builtin_bit_cast(bf16x8_t, Ws[p][64 * wn + 32 * t + l31][2 * ks + lh]);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16_t c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);   // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  }

  // ---- epilogue: lane holds D[row = 8 (r / 4) + 4 (lane / 32) + r % 4][col = lane % 32] of each 32 x 32 tile
  store_tile<ACT>(acc, bias, C, M, N, m0 + 64 * wm, n0 + 64 * wn, m0 + BM <= M && n0 + BN <= N, l31, lh);
}


constexpr int BK2 = 16;                       // k extent of one pipelined stage

struct StageRegs {
  f32x4 a0, a1;                               // A[row + 64 i][4 c .. 4 c + 3], i = 0, 1
  u32x4_t w0, w1, w2;                         // packed W block, plane p, linear 16-B index t
  int ok;                                     // conv mode: bit i = the tap of this stage is inside the image for row i
};

// Implicit-GEMM geometry of a 3 x 3, stride 1, pad 1 convolution over NHWC activations x [B,H,W,C]:
// GEMM row m = output pixel (b, y, x), k = (tap, c) with tap = 3 ky + kx; A[m][k] = x[b, y + ky - 1, x + kx - 1, c] or 0.
struct ConvGeom {
  int H, W, C, cpt;                           // cpt = C / 16 = k stages per tap
};

// Staging maps of the 256 staging threads (t = 0..255) for one 16-wide stage:
//   A: row = (t >> 2) + 64 i, 4-float chunk c = t & 3: a 4-lane quad reads 64 contiguous bytes of one row;
//      its three 8-byte plane pieces go to row*32 + (((c >> 1) ^ ((row >> 3) & 1)) * 16 + (c & 1) * 8 of the plane image
//   W: 16-byte piece t of each plane block, stored to the same linear offset.
struct StageMap {
  const float* a_src;                         // + s * 16 floats; second row at + 64 K
  const u32x4_t* w_src;                       // + s * 768
  int64_t a_row2;                             // element offset of the second row (0 when clamped onto the same row)
  int a_dst0, a_dst1;                         // uint2 index into a plane image [128][4]
  int taps0, taps1;                           // conv mode: 9-bit masks of the taps that fall inside the image, rows 0 / 1
  int w_second;                               // 512-thread tiles: offset of this thread's second W piece relative to w_src
};
// lda = row stride of A in floats (K for a Linear, C for the conv's NHWC activations)
template <bool CONV, int BMT = 128>
__device__ __forceinline__ StageMap make_stage_map(const float* A, const u32x4_t* Wp, int t, int m0, int nt, int M, int K, int lda,
                                                   const ConvGeom& g) {
  StageMap m;
  const int c = t & 3, r0 = t >> 2, r1 = r0 + BMT / 2;           // 2 BMT threads x 2 rows x 4 chunks = BMT rows x 16 floats
  int g0 = m0 + r0, g1 = m0 + r1;
  g0 = g0 < M ? g0 : M - 1;
  g1 = g1 < M ? g1 : M - 1;
  m.a_src = A + (int64_t)g0 * lda + c * 4;
  m.a_row2 = (int64_t)(g1 - g0) * lda;
  m.w_src = Wp + (int64_t)nt * (K >> 4) * 768 + t;
  m.a_dst0 = r0 * 4 + (((c >> 1) ^ ((r0 >> 3) & 1)) << 1) + (c & 1);
  m.a_dst1 = r1 * 4 + (((c >> 1) ^ ((r1 >> 3) & 1)) << 1) + (c & 1);
  m.taps0 = m.taps1 = 0x1ff;
  m.w_second = 512 + (t & 255) - t;
  if (CONV) {
    const int p0 = g0 % (g.H * g.W), p1 = g1 % (g.H * g.W);
    const int y0 = p0 / g.W, x0 = p0 - y0 * g.W, y1 = p1 / g.W, x1 = p1 - y1 * g.W;
    m.taps0 = m.taps1 = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      m.taps0 |= (y0 + dy >= 0 && y0 + dy < g.H && x0 + dx >= 0 && x0 + dx < g.W) ? 1 << tap : 0;
      m.taps1 |= (y1 + dy >= 0 && y1 + dy < g.H && x1 + dx >= 0 && x1 + dx < g.W) ? 1 << tap : 0;
    }
  }
  return m;
}
template <bool CONV, int BMT = 128>
__device__ __forceinline__ void stage_load(StageRegs& r, const StageMap& m, int s, const ConvGeom& g) {
  if (CONV) {
    const int tap = s / g.cpt, c0 = (s - tap * g.cpt) * 16;          // wave-uniform
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    const int off = (dy * g.W + dx) * g.C;
    const int ok0 = (m.taps0 >> tap) & 1, ok1 = (m.taps1 >> tap) & 1;
    r.ok = ok0 | (ok1 << 1);
    // out-of-image taps read the pixel itself (always in bounds) and are zeroed when the data is stored to LDS
    r.a0 = *reinterpret_cast<const f32x4*>(m.a_src + c0 + (ok0 ? off : 0));
    r.a1 = *reinterpret_cast<const f32x4*>(m.a_src + m.a_row2 + c0 + (ok1 ? off : 0));
  } else {
    const float* ap = m.a_src + s * 16;
    r.a0 = *reinterpret_cast<const f32x4*>(ap);
    r.a1 = *reinterpret_cast<const f32x4*>(ap + m.a_row2);
    r.ok = 3;
  }
  const u32x4_t* wp = m.w_src + (int64_t)s * 768;
  r.w0 = wp[0];
  if (BMT == 128) {                                               // 256 threads: three 16-byte pieces each
    r.w1 = wp[256];
    r.w2 = wp[512];
  } else {                                                        // 512 threads: piece t and piece 512 + (t & 255) (loaded twice, stored once)
    r.w1 = wp[m.w_second];
    r.w2 = r.w1;
  }
}
// As3 / Ws3: the three plane images of one stage buffer, [3][BMT][2] / [3][128][2] x 16 B each
template <int BMT = 128>
__device__ __forceinline__ void stage_store(const StageRegs& r, const StageMap& m, u32x4_t* As3, u32x4_t* Ws3, int t) {
  uint2 p0, p1, p2, q0, q1, q2;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 v0 = (r.ok & 1) ? r.a0 : z, v1 = (r.ok & 2) ? r.a1 : z;
  split4(make_float4(v0.x, v0.y, v0.z, v0.w), p0, p1, p2);
  split4(make_float4(v1.x, v1.y, v1.z, v1.w), q0, q1, q2);
  uint2* a2 = reinterpret_cast<uint2*>(As3);
  constexpr int PL = BMT * 4;                                     // uint2 per A plane image
  a2[m.a_dst0] = p0;
  a2[PL + m.a_dst0] = p1;
  a2[2 * PL + m.a_dst0] = p2;
  a2[m.a_dst1] = q0;
  a2[PL + m.a_dst1] = q1;
  a2[2 * PL + m.a_dst1] = q2;
  Ws3[t] = r.w0;
  if (BMT == 128) {
    Ws3[256 + t] = r.w1;
    Ws3[512 + t] = r.w2;
  } else if (t < 256) {
    Ws3[512 + t] = r.w1;
  }
}

// ---- v3: three unpadded, swizzled LDS stage buffers (24 KB each) and ONE barrier per 16-wide k stage, placed mid-stage.
// Stage s computes from buffer s % 3; at its head the tile of stage s + 2 (global loads issued two stages earlier) is split
// and written to buffer (s + 2) % 3, so by the time any wave reads a buffer its writes are a full stage old, and the
// first operands of stage s + 1 are read before stage s ends: no LDS or HBM latency is exposed in steady state.
// Rows are 32 B (16 bf16); the 16-B half h of row r sits at half-slot h ^ ((r >> 3) & 1): conflict-free for ds_read_b128's
// lane groups without padding.
constexpr int STG = 3;

// CONV: A is an NHWC activation tensor and the GEMM is the implicit 3 x 3 convolution described at ConvGeom (K = 9 C).
// TRANS: the output is written transposed, out[(b, n, p)] for GEMM row m = b * ldo + p (NCHW from NHWC rows): the operand roles
// are swapped in the MFMAs (weights on the accumulator-row side) so that a wave-store still covers 128 contiguous bytes.
template <int ACT, bool CONV = false, bool TRANS = false, int BMT = 128>
__global__ __launch_bounds__(2 * BMT) void split_linear_pipe_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                              const float* __restrict__ bias, float* __restrict__ C, int M, int N,
                                                              int K, int MT, int NT, ConvGeom geom, int ldo) {
  __shared__ u32x4_t As[STG][3][BMT][2];
  __shared__ u32x4_t Ws[STG][3][BN][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BMT, n0 = nt * BN;

  const StageMap smap = make_stage_map<CONV, BMT>(A, Wp, tid, m0, nt, M, K, CONV ? geom.C : K, geom);
  const int S = K / BK2, SL = S - 1;
  auto gload = [&](StageRegs& r, int s) { stage_load<CONV, BMT>(r, smap, s < SL ? s : SL, geom); };
  auto stash = [&](const StageRegs& r, int buf) { stage_store<BMT>(r, smap, &As[buf][0][0][0], &Ws[buf][0][0][0], tid); };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int fa_row = 64 * wm + l31, fb_row = 64 * wn + l31;       // + 32 t; (row >> 3) & 1 is the same for t = 0, 1
  const int fa_slot = lh ^ ((fa_row >> 3) & 1), fb_slot = lh ^ ((fb_row >> 3) & 1);
  bf16x8_t a[2][3], b[2][3], na[2], nb0[2];
  auto rd_a = [&](int buf, int t, int p) {
    return __builtin_bit_cast(bf16x8_t, TRANS ? Ws[buf][p][fa_row + 32 * t][fa_slot] : As[buf][p][fa_row + 32 * t][fa_slot]);
  };
  auto rd_b = [&](int buf, int t, int p) {
    return __builtin_bit_cast(bf16x8_t, TRANS ? As[buf][p][fb_row + 32 * t][fb_slot] : Ws[buf][p][fb_row + 32 * t][fb_slot]);
  };
#define RBA_G(pa, pb)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)           \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);

  StageRegs rx, ry;
  gload(rx, 0);
  gload(ry, 1);
  stash(rx, 0);
  gload(rx, 2);
  stash(ry, 1);
  gload(ry, 3);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 2; ++t) { a[t][0] = rd_a(0, t, 0); b[t][0] = rd_b(0, t, 0); }

  int cur = 0;                                                    // s % 3
#define RBA_STAGE(SET, s)                                                            \
  {                                                                                  \
    const int nxt = cur == 2 ? 0 : cur + 1, wr = nxt == 2 ? 0 : nxt + 1;             \
    stash(SET, wr);                                                                  \
    gload(SET, (s) + 4);                                                             \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                  \
      a[t][1] = rd_a(cur, t, 1); b[t][1] = rd_b(cur, t, 1);                          \
      a[t][2] = rd_a(cur, t, 2); b[t][2] = rd_b(cur, t, 2);                          \
    }                                                                                \
    RBA_G(0, 0) RBA_G(0, 1) RBA_G(1, 0)                                              \
    __syncthreads();                                                                 \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) { na[t] = rd_a(nxt, t, 0); nb0[t] = rd_b(nxt, t, 0); } \
    RBA_G(1, 1) RBA_G(0, 2) RBA_G(2, 0)                                              \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) { a[t][0] = na[t]; b[t][0] = nb0[t]; } \
    cur = nxt;                                                                       \
  }
  int s = 0;
  for (; s + 2 <= S; s += 2) {
    RBA_STAGE(rx, s)
    RBA_STAGE(ry, s + 1)
  }
  if (s < S) RBA_STAGE(rx, s)
#undef RBA_STAGE
#undef RBA_G

  if (TRANS)
    store_tile_transposed<ACT>(acc, bias, C, M, N, n0 + 64 * wm, m0 + 64 * wn, ldo, l31, lh);
  else
    store_tile<ACT>(acc, bias, C, M, N, m0 + 64 * wm, n0 + 64 * wn, m0 + BMT <= M && n0 + BN <= N, l31, lh);
}


// ---- LDS-DMA variant (RBA_GEMM_VARIANT=8): the packed W tile of a stage is already the LDS image, so it is moved by
// global_load_lds_dwordx4 (three 1-KiB pieces per wave, no VGPRs, no ds_write); only the activations go through registers for
// the split.  With an LDS-DMA in flight hipcc would drain vmcnt(0) at every use of an ordinary global load and at
// __syncthreads(), so the A loads are inline asm and the VM counter is managed by hand.  Per thread and stage the VMEM issue order
// is [W(s+2) x3 DMA, A(s+4) x2]: A(s+2) has landed when at most 8 newer operations are outstanding (vmcnt(8), head of the stage),
// W(s+1) when at most 7 are (vmcnt(7), before the stage's barrier; stage s+1 is first read after that barrier).
extern __shared__ __attribute__((aligned(16))) u32x4_t lds_main[];
extern __shared__ __attribute__((aligned(16))) u32x4_t lds_alias[];

template <int ACT>
__global__ __launch_bounds__(256) void split_linear_dma_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                             const float* __restrict__ bias, float* __restrict__ C, int M, int N,
                                                             int K, int MT, int NT) {
  // One dynamic LDS block seen through two symbols: HIP places every `extern __shared__` array at the same base, and the
  // compiler treats them as distinct objects.  The DMA writes through `lds_dma`, everything else goes through `lds`; otherwise
  // hipcc orders every LDS read behind the pending LDS-DMAs with s_waitcnt vmcnt(0) and the prefetch collapses.  The ordering
  // that IS needed (W of stage s+1 landed before the barrier that precedes its first read) is the counted vmcnt(7) below.
  u32x4_t(*As)[3][BM][2] = reinterpret_cast<u32x4_t(*)[3][BM][2]>(lds_main);
  u32x4_t(*Ws)[3][BN][2] = reinterpret_cast<u32x4_t(*)[3][BN][2]>(lds_main + STG * 3 * BM * 2);
  u32x4_t* Wdma = lds_alias + STG * 3 * BM * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const ConvGeom nogeom = {0, 0, 0, 1};
  const StageMap smap = make_stage_map<false>(A, Wp, tid, m0, nt, M, K, K, nogeom);
  const int S = K / BK2, SL = S - 1;

  struct ARegs { f32x4 a0, a1; };
  auto aload = [&](ARegs& r, int s) {
    const float* ap = smap.a_src + (s < SL ? s : SL) * 16;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.a0) : "v"(ap) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r.a1) : "v"(ap + smap.a_row2) : "memory");
  };
  auto wdma = [&](int s, int buf) {                               // W tile of stage s -> Ws[buf]: lane's piece index = tid
    const u32x4_t* wp = smap.w_src + (int64_t)(s < SL ? s : SL) * 768;
#pragma unroll
    for (int p = 0; p < 3; ++p)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wp + 256 * p),
                                       (__attribute__((address_space(3))) void*)(Wdma + (buf * 3 + p) * (BN * 2) + wave * 64), 16, 0, 0);
  };
  // the A-plane stores are inline asm as well: hipcc puts s_waitcnt vmcnt(0) in front of every LDS store it can see while an
  // LDS-DMA is pending (write-after-write ordering it cannot disprove)
  const uint32_t as_base = (uint32_t)(uintptr_t)(&As[0][0][0][0]);
  const uint32_t ad0 = as_base + smap.a_dst0 * 8, ad1 = as_base + smap.a_dst1 * 8;
  auto astash = [&](const ARegs& r, int buf) {
    uint2 p0, p1, p2, q0, q1, q2;
    split4(make_float4(r.a0.x, r.a0.y, r.a0.z, r.a0.w), p0, p1, p2);
    split4(make_float4(r.a1.x, r.a1.y, r.a1.z, r.a1.w), q0, q1, q2);
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const uint32_t o = buf * (3 * BM * 2 * 16);
    const u2 P0 = {p0.x, p0.y}, P1 = {p1.x, p1.y}, P2 = {p2.x, p2.y}, Q0 = {q0.x, q0.y}, Q1 = {q1.x, q1.y}, Q2 = {q2.x, q2.y};
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:4096\n\tds_write_b64 %0, %3 offset:8192"
                 :: "v"(ad0 + o), "v"(P0), "v"(P1), "v"(P2) : "memory");
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:4096\n\tds_write_b64 %0, %3 offset:8192"
                 :: "v"(ad1 + o), "v"(Q0), "v"(Q1), "v"(Q2) : "memory");
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int l31 = lane & 31, lh = lane >> 5;
  const int fa_row = 64 * wm + l31, fb_row = 64 * wn + l31;
  const int fa_slot = lh ^ ((fa_row >> 3) & 1), fb_slot = lh ^ ((fb_row >> 3) & 1);
  bf16x8_t a[2][3], b[2][3], na[2], nb0[2];
  auto rd_a = [&](int buf, int t, int p) { return __builtin_bit_cast(bf16x8_t, As[buf][p][fa_row + 32 * t][fa_slot]); };
  auto rd_b = [&](int buf, int t, int p) { return __builtin_bit_cast(bf16x8_t, Ws[buf][p][fb_row + 32 * t][fb_slot]); };
#define RBA_G(pa, pb)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)           \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);

  ARegs ax, ay;
  aload(ax, 0);
  aload(ay, 1);
  wdma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ax.a0), "+v"(ax.a1), "+v"(ay.a0), "+v"(ay.a1)::"memory");
  astash(ax, 0);
  astash(ay, 1);
  // establish the steady-state queue [A(2) x2, W(1) x3, A(3) x2] the counted waits of stage 0 assume
  aload(ax, 2);
  wdma(1, 1);
  aload(ay, 3);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int t = 0; t < 2; ++t) { a[t][0] = rd_a(0, t, 0); b[t][0] = rd_b(0, t, 0); }

  int cur = 0;
  // at the head of stage s the outstanding VMEM operations are, oldest first: [A(s+2) x2] (set SET), W(s+1) x3, A(s+3) x2
#define RBA_STAGE(SET, s)                                                            \
  {                                                                                  \
    const int nxt = cur == 2 ? 0 : cur + 1, wr = nxt == 2 ? 0 : nxt + 1;             \
    wdma((s) + 2, wr);                                                               \
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(SET.a0), "+v"(SET.a1)::"memory");       \
    astash(SET, wr);                                                                 \
    aload(SET, (s) + 4);                                                             \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                  \
      a[t][1] = rd_a(cur, t, 1); b[t][1] = rd_b(cur, t, 1);                          \
      a[t][2] = rd_a(cur, t, 2); b[t][2] = rd_b(cur, t, 2);                          \
    }                                                                                \
    RBA_G(0, 0) RBA_G(0, 1) RBA_G(1, 0)                                              \
    asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");         \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) { na[t] = rd_a(nxt, t, 0); nb0[t] = rd_b(nxt, t, 0); } \
    RBA_G(1, 1) RBA_G(0, 2) RBA_G(2, 0)                                              \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) { a[t][0] = na[t]; b[t][0] = nb0[t]; } \
    cur = nxt;                                                                       \
  }
  int s = 0;
  for (; s + 2 <= S; s += 2) {
    RBA_STAGE(ax, s)
    RBA_STAGE(ay, s + 1)
  }
  if (s < S) RBA_STAGE(ax, s)
#undef RBA_STAGE
#undef RBA_G
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // drain the clamped surplus loads / DMAs before the epilogue
  store_tile<ACT>(acc, bias, C, M, N, m0 + 64 * wm, n0 + 64 * wn, m0 + BM <= M && n0 + BN <= N, l31, lh);
}

// ---- wave-specialised variant: 8 waves per workgroup.  Waves 0-3 (one per SIMD) only read operand fragments and issue
// MFMAs; waves 4-7 (their SIMD partners) only stage: global loads two stages ahead, the bf16 split, LDS stores.  A wave issues
// in order, so in the 4-wave kernels every staging instruction sits between two MFMAs of the same wave (counters: LDS stores
// 22 %, split VALU 18 % of wave time, the matrix pipe 39 % busy); here the staging stream runs beside an almost pure MFMA
// stream.  Same three swizzled LDS stage buffers and one barrier per stage as the pipe kernel.
template <int ACT>
__global__ __launch_bounds__(512) void split_linear_ws_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                              const float* __restrict__ bias, float* __restrict__ C, int M, int N,
                                                              int K, int MT, int NT) {
  __shared__ u32x4_t As[STG][3][BM][2];
  __shared__ u32x4_t Ws[STG][3][BN][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int S = K / BK2, SL = S - 1;

  if (wave >= 4) {                                                // ---------------- staging waves
    const int t = tid - 256;
    const ConvGeom nogeom = {0, 0, 0, 1};
    const StageMap smap = make_stage_map<false>(A, Wp, t, m0, nt, M, K, K, nogeom);
    auto gload = [&](StageRegs& r, int s) { stage_load<false>(r, smap, s < SL ? s : SL, nogeom); };
    auto stash = [&](const StageRegs& r, int buf) { stage_store(r, smap, &As[buf][0][0][0], &Ws[buf][0][0][0], t); };
    // NS rotating register sets: during stage s, set s % NS (stage s + 2, loaded NS stages ago) is split and stored, then
    // refilled with stage s + 2 + NS.  Loads are unconditional (clamped) so that the compiler's vmcnt waits stay exact.
    constexpr int NS = 2;
    StageRegs rs[NS];
    gload(rs[0], 0);
    gload(rs[1], 1);
    stash(rs[0], 0);
    stash(rs[1], 1);
#pragma unroll
    for (int j = 0; j < NS; ++j) gload(rs[j], 2 + j);
    __syncthreads();
    int wr = 2, s = 0;                                            // during stage s the tile of stage s + 2 goes to buffer (s + 2) % 3
    for (; s + NS <= S; s += NS) {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        stash(rs[j], wr);
        gload(rs[j], s + j + 2 + NS);
        __syncthreads();
        wr = wr == 2 ? 0 : wr + 1;
      }
    }
#pragma unroll
    for (int j = 0; j < NS - 1; ++j) {
      if (s + j < S) {
        stash(rs[j], wr);
        __syncthreads();
        wr = wr == 2 ? 0 : wr + 1;
      }
    }
    return;
  }

  // ---------------- MFMA waves
  const int wm = wave >> 1, wn = wave & 1;
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int l31 = lane & 31, lh = lane >> 5;
  const int fa_row = 64 * wm + l31, fb_row = 64 * wn + l31;
  const int fa_slot = lh ^ ((fa_row >> 3) & 1), fb_slot = lh ^ ((fb_row >> 3) & 1);
  bf16x8_t a[2][3], b[2][3], na[2], nb0[2];
  auto rd_a = [&](int buf, int t, int p) { return __builtin_bit_cast(bf16x8_t, As[buf][p][fa_row + 32 * t][fa_slot]); };
  auto rd_b = [&](int buf, int t, int p) { return __builtin_bit_cast(bf16x8_t, Ws[buf][p][fb_row + 32 * t][fb_slot]); };
#define RBA_G(pa, pb)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)           \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 2; ++t) { a[t][0] = rd_a(0, t, 0); b[t][0] = rd_b(0, t, 0); }
  int cur = 0;
  for (int s = 0; s < S; ++s) {
    const int nxt = cur == 2 ? 0 : cur + 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      a[t][1] = rd_a(cur, t, 1); b[t][1] = rd_b(cur, t, 1);
      a[t][2] = rd_a(cur, t, 2); b[t][2] = rd_b(cur, t, 2);
    }
    RBA_G(0, 0) RBA_G(0, 1) RBA_G(1, 0)
#pragma unroll
    for (int t = 0; t < 2; ++t) { na[t] = rd_a(nxt, t, 0); nb0[t] = rd_b(nxt, t, 0); }   // stage s + 1: written during stage s - 1
    RBA_G(1, 1) RBA_G(0, 2) RBA_G(2, 0)
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) { a[t][0] = na[t]; b[t][0] = nb0[t]; }
    cur = nxt;
  }
#undef RBA_G

  store_tile<ACT>(acc, bias, C, M, N, m0 + 64 * wm, n0 + 64 * wn, m0 + BM <= M && n0 + BN <= N, l31, lh);
}

}  // namespace

extern "C" int rba_split_weight_bf16x3(const float* weight, void* packed, int N, int K, void* stream) {
  RBA_CHECK_ARG(N >= 0 && K >= 0 && (K % BK) == 0);
  if (N == 0 || K == 0) return 0;
  RBA_CHECK_ARG(weight && packed && (((uintptr_t)weight | (uintptr_t)packed) & 15) == 0);
  rba_begin();
  const int64_t total = (int64_t)((N + BN - 1) / BN * BN) * (K >> 3);
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(split_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, weight,
                     reinterpret_cast<u32x4_t*>(packed), N, K);
  return rba_launch_status();
}

extern "C" int rba_split_linear_f32(const float* x, const void* weight_planes, const float* bias, float* out, int64_t M, int N,
                                    int K, int act, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 1 && K >= BK && (K % BK) == 0 && act >= 0 && act <= 2);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_planes && out);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_planes | (uintptr_t)out) & 15) == 0);
  const int64_t MT = (M + BM - 1) / BM;
  const int NT = (N + BN - 1) / BN;
  RBA_CHECK_ARG(MT * NT < (int64_t)1 << 31 && M < (int64_t)1 << 31);
  rba_begin();
  const dim3 grid((unsigned)(MT * NT)), block(256);
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_planes);
  static const int forced = getenv("RBA_GEMM_VARIANT") ? atoi(getenv("RBA_GEMM_VARIANT")) : 0;   // tuning hook (tools/gemm_sweep.py)
  const bool short_k = forced ? forced == 1 : K <= 256;
#define RBA_L(KERNEL, A) hipLaunchKernelGGL(KERNEL<A>, grid, block, 0, (hipStream_t)stream, x, wp, bias, out, (int)M, N, K, (int)MT, NT)
#define RBA_LP(A) hipLaunchKernelGGL((split_linear_pipe_kernel<A, false, false>), grid, block, 0, (hipStream_t)stream, x, wp, bias, out, (int)M, N, K, (int)MT, NT, ConvGeom{0, 0, 0, 1}, 0)
  // 256 x 128 tiles (8 waves, one workgroup per CU; RBA_GEMM_VARIANT=7) halve the W staging per MFMA.  Stand-alone they are up to
  // 14 % faster where the 256-row tiles fill the CUs in whole rounds (8192 x 2048 x 512: 100 vs 117 us) and slower elsewhere; inside
  // the network a shape rule that picks them made no difference (72.6 vs 73.1 images/s), so the 128-row tile stays the default.
  if (forced == 8) {
    constexpr size_t dyn = (size_t)STG * 3 * (BM + BN) * 2 * sizeof(u32x4_t);
#define RBA_LD(A)                                                                                                                    \
  {                                                                                                                                  \
    static const hipError_t attr = hipFuncSetAttribute((const void*)split_linear_dma_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
    (void)attr;                                                                                                                      \
    hipLaunchKernelGGL(split_linear_dma_kernel<A>, grid, block, dyn, (hipStream_t)stream, x, wp, bias, out, (int)M, N, K, (int)MT, NT); \
  }
    if (act == 1) RBA_LD(1) else if (act == 2) RBA_LD(2) else RBA_LD(0)
#undef RBA_LD
  } else if (forced == 7) {
    const int64_t MT2 = (M + 255) / 256;
    const dim3 grid2((unsigned)(MT2 * NT));
    if (act == 1)
      hipLaunchKernelGGL((split_linear_pipe_kernel<1, false, false, 256>), grid2, dim3(512), 0, (hipStream_t)stream, x, wp, bias, out, (int)M, N, K, (int)MT2, NT, ConvGeom{0, 0, 0, 1}, 0);
    else if (act == 2)
      hipLaunchKernelGGL((split_linear_pipe_kernel<2, false, false, 256>), grid2, dim3(512), 0, (hipStream_t)stream, x, wp, bias, out, (int)M, N, K, (int)MT2, NT, ConvGeom{0, 0, 0, 1}, 0);
    else
      hipLaunchKernelGGL((split_linear_pipe_kernel<0, false, false, 256>), grid2, dim3(512), 0, (hipStream_t)stream, x, wp, bias, out, (int)M, N, K, (int)MT2, NT, ConvGeom{0, 0, 0, 1}, 0);
  } else if (forced == 4) {
#define RBA_L8(A) hipLaunchKernelGGL((split_linear_ws_kernel<A>), grid, dim3(512), 0, (hipStream_t)stream, x, wp, bias, out, (int)M, N, K, (int)MT, NT)
    if (act == 1) RBA_L8(1); else if (act == 2) RBA_L8(2); else RBA_L8(0);
#undef RBA_L8
  } else if (short_k) {
    if (act == 1) RBA_L(split_linear_short_kernel, 1); else if (act == 2) RBA_L(split_linear_short_kernel, 2); else RBA_L(split_linear_short_kernel, 0);
  } else {
    if (act == 1) RBA_LP(1); else if (act == 2) RBA_LP(2); else RBA_LP(0);
  }
#undef RBA_L
#undef RBA_LP
  return rba_launch_status();
}

// out[(b, n, p)] = sum_k x[b * P + p, k] * weight[n, k] + bias[n]: a Linear over NHWC rows written as NCHW ([B,N,P], P = rows
// per image).  Used for the pixel decoder's mask-feature 1 x 1 convolution (msdeformattn.py:298-306 on NHWC activations).
extern "C" int rba_split_linear_nchw_out_f32(const float* x, const void* weight_packed, const float* bias, float* out, int64_t M,
                                             int N, int K, int rows_per_image, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 1 && K >= BK && (K % BK) == 0 && rows_per_image >= 1);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_packed && out && (M % rows_per_image) == 0);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_packed | (uintptr_t)out) & 15) == 0);
  const int64_t MT = (M + BM - 1) / BM;
  const int NT = (N + BN - 1) / BN;
  RBA_CHECK_ARG(MT * NT < (int64_t)1 << 31 && M < (int64_t)1 << 31);
  rba_begin();
  hipLaunchKernelGGL((split_linear_pipe_kernel<0, false, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<const u32x4_t*>(weight_packed), bias, out, (int)M, N, K, (int)MT, NT, ConvGeom{0, 0, 0, 1},
                     rows_per_image);
  return rba_launch_status();
}

// 3 x 3, stride 1, pad 1 convolution over NHWC activations as an implicit GEMM on the bf16x6 kernel:
// x [B,H,W,C] -> out [B,H,W,N]; weight_packed = rba_split_weight_bf16x3 of the [N, 9 C] matrix w[n][(3 ky + kx) * C + c]
// (= conv weight [N,C,3,3] permuted to [N,3,3,C]).  C % 16 == 0.  (msdeformattn.py:278-297 `layer_{j}` output convolutions.)
extern "C" int rba_conv3x3_nhwc_f32(const float* x, const void* weight_packed, const float* bias, float* out, int B, int H, int W,
                                    int C, int N, void* stream) {
  RBA_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && C >= 16 && (C % 16) == 0 && N >= 1 && ((9 * C) % BK) == 0);
  const int64_t M = (int64_t)B * H * W;
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_packed && out);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_packed | (uintptr_t)out) & 15) == 0);
  const int64_t MT = (M + BM - 1) / BM;
  const int NT = (N + BN - 1) / BN;
  RBA_CHECK_ARG(MT * NT < (int64_t)1 << 31 && M * C < (int64_t)1 << 31);
  rba_begin();
  hipLaunchKernelGGL((split_linear_pipe_kernel<0, true, false>), dim3((unsigned)(MT * NT)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<const u32x4_t*>(weight_packed), bias, out, (int)M, N, 9 * C, (int)MT, NT,
                     ConvGeom{H, W, C, C / 16}, 0);
  return rba_launch_status();
}
