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

__global__ void split_weight_kernel(const float* __restrict__ w, uint2* __restrict__ planes, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    uint2 p0, p1, p2;
    split4(reinterpret_cast<const float4*>(w)[i], p0, p1, p2);
    planes[i] = p0;
    planes[n4 + i] = p1;
    planes[2 * n4 + i] = p2;
  }
}

// act: 0 none, 1 exact GELU (0.5 x (1 + erf(x / sqrt 2)), nn.GELU default, swin.py:51)
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
  const int w_ch = tid & 3, w_r = tid >> 2;                       // W: plane p, rows w_r + 64 i (i < 2), 16 B chunk w_ch
  const int Kq = K >> 3;                                          // uint4 per row
  const u32x4_t* w_src[6];                                        // j = 2 p + i
#pragma unroll
  for (int j = 0; j < 6; ++j) w_src[j] = Wp + ((int64_t)(j >> 1) * N + n0 + w_r + 64 * (j & 1)) * Kq + w_ch;

  f32x4 pa[4];
  u32x4_t pw[6];
#pragma unroll
  for (int i = 0; i < 4; ++i) pa[i] = *reinterpret_cast<const f32x4*>(a_src[i]);
#pragma unroll
  for (int j = 0; j < 6; ++j) pw[j] = w_src[j][0];

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
    for (int j = 0; j < 6; ++j) Ws[j >> 1][w_r + 64 * (j & 1)][w_ch] = pw[j];
    __syncthreads();
    if (k0 + BK < K) {                             // prefetch the next stage into registers
      const int kn = k0 + BK;
#pragma unroll
      for (int i = 0; i < 4; ++i) pa[i] = *reinterpret_cast<const f32x4*>(a_src[i] + kn);
#pragma unroll
      for (int j = 0; j < 6; ++j) pw[j] = w_src[j][kn >> 3];
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
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 64 * wn + 32 * j + l31;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 64 * wm + 32 * i + 8 * (r >> 2) + 4 * lh + (r & 3);
        float v = acc[i][j][r] + bv;
        if (ACT == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (row < M) C[(int64_t)row * N + col] = v;
      }
  }
}


constexpr int BK2 = 16;                       // k extent of one pipelined stage

struct StageRegs {
  f32x4 a0, a1;                               // A[row][8 half .. 8 half + 7]
  u32x4_t w0, w1, w2;                         // W planes 0..2: [row][8 half .. +7] bf16
};

// ---- v3: three unpadded, swizzled LDS stage buffers (24 KB each) and ONE barrier per 16-wide k stage, placed mid-stage.
// Stage s computes from buffer s % 3; at its head the tile of stage s + 2 (global loads issued two stages earlier) is split
// and written to buffer (s + 2) % 3, so by the time any wave reads a buffer its writes are a full stage old, and the
// first operands of stage s + 1 are read before stage s ends: no LDS or HBM latency is exposed in steady state.
// Rows are 32 B (16 bf16); the 16-B half h of row r sits at half-slot h ^ ((r >> 3) & 1): conflict-free for ds_read_b128's
// lane groups without padding.
constexpr int STG = 3;

template <int ACT>
__global__ __launch_bounds__(256) void split_linear_pipe_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                              const float* __restrict__ bias, float* __restrict__ C, int M, int N,
                                                              int K, int MT, int NT) {
  __shared__ u32x4_t As[STG][3][BM][2];
  __shared__ u32x4_t Ws[STG][3][BN][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;

  const int s_row = tid >> 1, s_half = tid & 1;
  const int s_slot = s_half ^ ((s_row >> 3) & 1);
  int ar = m0 + s_row;
  ar = ar < M ? ar : M - 1;
  const float* a_src = A + (int64_t)ar * K + s_half * 8;
  const int Kq = K >> 3;
  const u32x4_t* w_src = Wp + ((int64_t)(n0 + s_row)) * Kq + s_half;
  const int64_t w_plane = (int64_t)N * Kq;
  const int S = K / BK2, SL = S - 1;

  auto gload = [&](StageRegs& r, int s) {
    s = s < SL ? s : SL;
    const float* ap = a_src + s * BK2;
    r.a0 = *reinterpret_cast<const f32x4*>(ap);
    r.a1 = *reinterpret_cast<const f32x4*>(ap + 4);
    const u32x4_t* wp = w_src + s * 2;
    r.w0 = wp[0];
    r.w1 = wp[w_plane];
    r.w2 = wp[2 * w_plane];
  };
  auto stash = [&](const StageRegs& r, int buf) {
    uint2 p0, p1, p2, q0, q1, q2;
    split4(make_float4(r.a0.x, r.a0.y, r.a0.z, r.a0.w), p0, p1, p2);
    split4(make_float4(r.a1.x, r.a1.y, r.a1.z, r.a1.w), q0, q1, q2);
    As[buf][0][s_row][s_slot] = (u32x4_t){p0.x, p0.y, q0.x, q0.y};
    As[buf][1][s_row][s_slot] = (u32x4_t){p1.x, p1.y, q1.x, q1.y};
    As[buf][2][s_row][s_slot] = (u32x4_t){p2.x, p2.y, q2.x, q2.y};
    Ws[buf][0][s_row][s_slot] = r.w0;
    Ws[buf][1][s_row][s_slot] = r.w1;
    Ws[buf][2][s_row][s_slot] = r.w2;
  };

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
  auto rd_a = [&](int buf, int t, int p) { return __builtin_bit_cast(bf16x8_t, As[buf][p][fa_row + 32 * t][fa_slot]); };
  auto rd_b = [&](int buf, int t, int p) { return __builtin_bit_cast(bf16x8_t, Ws[buf][p][fb_row + 32 * t][fb_slot]); };
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

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 64 * wn + 32 * j + l31;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 64 * wm + 32 * i + 8 * (r >> 2) + 4 * lh + (r & 3);
        float v = acc[i][j][r] + bv;
        if (ACT == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (row < M) C[(int64_t)row * N + col] = v;
      }
  }
}

}  // namespace

extern "C" int rba_split_weight_bf16x3(const float* weight, void* planes, int64_t elems, void* stream) {
  RBA_CHECK_ARG(elems >= 0 && (elems & 3) == 0);
  if (elems == 0) return 0;
  RBA_CHECK_ARG(weight && planes && (((uintptr_t)weight | (uintptr_t)planes) & 15) == 0);
  rba_begin();
  const int64_t n4 = elems >> 2;
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(split_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, weight,
                     reinterpret_cast<uint2*>(planes), n4);
  return rba_launch_status();
}

extern "C" int rba_split_linear_f32(const float* x, const void* weight_planes, const float* bias, float* out, int64_t M, int N,
                                    int K, int act, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= BN && (N % BN) == 0 && K >= BK && (K % BK) == 0 && (act == 0 || act == 1));
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_planes && out);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_planes | (uintptr_t)out) & 15) == 0);
  const int64_t MT = (M + BM - 1) / BM;
  const int NT = N / BN;
  RBA_CHECK_ARG(MT * NT < (int64_t)1 << 31 && M < (int64_t)1 << 31);
  rba_begin();
  const dim3 grid((unsigned)(MT * NT)), block(256);
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_planes);
  static const int forced = getenv("RBA_GEMM_VARIANT") ? atoi(getenv("RBA_GEMM_VARIANT")) : 0;   // tuning hook (tools/gemm_sweep.py)
  const bool short_k = forced ? forced == 1 : K <= 256;
#define RBA_L(KERNEL, A) hipLaunchKernelGGL(KERNEL<A>, grid, block, 0, (hipStream_t)stream, x, wp, bias, out, (int)M, N, K, (int)MT, NT)
  if (short_k) {
    if (act == 1) RBA_L(split_linear_short_kernel, 1); else RBA_L(split_linear_short_kernel, 0);
  } else {
    if (act == 1) RBA_L(split_linear_pipe_kernel, 1); else RBA_L(split_linear_pipe_kernel, 0);
  }
#undef RBA_L
  return rba_launch_status();
}
