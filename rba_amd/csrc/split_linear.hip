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
// This file: the weight split / packing, and split_linear_pipe_kernel -- 128 x 128 tile, 4 waves x (2 x 2 tiles of 32 x 32), 16-wide
// k stages staged through registers (activations split while staged) into three swizzled LDS stage buffers, one barrier per
// stage -- which now serves the implicit-GEMM 3 x 3 convolution (CONV) and the NCHW-output Linear (TRANS).  The plain token
// Linear (rba_split_linear_f32) moved to the all-LDS-DMA kernel of split_linear_dma.h / split_linear_dma.hip (round 2: 8-25 %
// faster; history in profiles/r01_split_linear.txt and profiles/r02_split_linear.txt).
#include <stdlib.h>

#include "common.h"
#include "../../include/rba_hip.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));   // native vectors: arrays of HIP's uint4 class are not promoted to registers

constexpr int BM = 128, BN = 128, BK = 32;

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
        if (ACT == 2) v[r] = rba_relu(v[r]);
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
        if (ACT == 2) v[r] = rba_relu(v[r]);
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
