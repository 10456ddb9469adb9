// K7 -- the attention half of a Swin block in ONE kernel (reference: backbone/swin.py:235-284 SwinTransformerBlock.forward up to
// `x = shortcut + drop_path(x)`, with WindowAttention.forward :131-171, the pad / roll / window_partition / window_reverse / un-roll / crop
// of :251-284 and the SW-MSA mask of BasicLayer.forward :413-440 as index arithmetic, and optionally norm2 of :284-293):
//
//     x <- x + proj(window_attention(qkv(norm1(x))))          [ + y2 = norm2(x) ]
//
// One workgroup = one 12 x 12 window (all heads), nine waves, wave w = the 16-token strip w of the window, and a strip's rows never leave
// their wave:
//   * norm1: the wave loads its 16 rows (lane = (token l15, 8-channel group kk of every 32-channel block)), two in-register passes +
//     xor-shuffles, and splits the normalised rows into the f16x3 operand pair (h, l) ONCE: they stay in registers as the B operand
//     (token = column) of every qkv MFMA of every head.  Tokens of the window padding are zero rows (the reference pads AFTER norm1).
//   * per head: q, k, v of the strip = W_head [96 x C] . x_strip^T on v_mfma_f32_16x16x32_f16 (three f16 MFMAs per fp32 product, main + low
//     accumulator: split_linear_h3.h), the weights as the A operand (feature = row) from an LDS image that the nine waves share.  The feature
//     order of the packed weight rows is chosen so that the lane (token l15, kk) ends up with head-dim channels 8 kk .. 8 kk + 7 of its
//     token: exactly the lane's piece of the Q^T operand of the attention (registers), of the K plane (one ds_write_b128) and of the V plane.
//   * attention of the strip against the window's 144 keys: K5's strip body (swin_window_attn_h3.h) unchanged -- S^T = K Q^T with the
//     relative-position bias as the accumulator's initial value, shift mask from region ids, softmax in registers, O^T = V^T P^T through
//     transposing LDS reads.
//   * proj: out_strip^T += W_proj[:, head] [C x 32] . O_head^T, O split in registers as the B operand (the k order of the packed proj weight
//     follows the attention's output layout), accumulated over the heads; the low accumulator is folded in after every head.
//   * epilogue: x = (x + out) + proj.bias in the lane layout of the first load (8 consecutive channels per lane: 32-byte stores), and,
//     if asked, y2 = norm2(x) from the same registers.
// Nothing but x (read twice, the second time from L2, written once) and y2 touches HBM: the qkv tensor, the attention output and both
// LayerNorm outputs of the unfused sequence (add_layer_norm -> K6 qkv -> K5 -> K6 proj (+ add_layer_norm)) never exist.
//
// Weights: rba_swin_attn_block_pack_f32 writes, per head, one contiguous chunk [qkv fragments | proj fragments] of 1 KiB MFMA A-operand
// fragments (64 lanes x 16 B, h and l planes); a chunk is copied to LDS with global_load_lds (no registers) while the previous head's
// attention runs.  Two barriers per head: (B) chunk landed + K / V planes free, (A) K / V planes written + qkv chunk free.
// LDS (C = 128): qkv chunk 48 KiB + two proj chunks 32 KiB + K / V planes 36 KiB + region ids = 117 KiB: one workgroup per CU.
// Arithmetic: f16x3 everywhere (|x| < 65504, NaN beyond, like K5 / K6).
#pragma once
#include <math.h>
#include "common.h"

// Timing build (tools only, csrc/tune/k7_timing.hip defines K7_TIMING before including this header): lane 0 of every wave stamps the shader clock
// (s_memtime) at its phase boundaries into dbg[(workgroup * 9 + wave) * 32 + i]; the product build compiles none of it.
#ifdef K7_TIMING
#define K7_DBG_PARAM , unsigned long long* __restrict__ dbg
#define K7_STAMP(i)                                                                                                                       \
  do {                                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
    if ((threadIdx.x & 63) == 0)                                                                                                          \
      dbg[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 9 + (threadIdx.x >> 6)) * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                                                                    \
  } while (0)
#else
#define K7_DBG_PARAM
#define K7_STAMP(i)
#endif

namespace {

typedef _Float16 k7_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 k7_f16x2 __attribute__((ext_vector_type(2)));
typedef __fp16 k7_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef uint32_t k7_u32x4 __attribute__((ext_vector_type(4)));
typedef float k7_f32x4 __attribute__((ext_vector_type(4)));

constexpr int K7_WS = 12, K7_N = 144, K7_NT = 9, K7_WAVES = 9, K7_PL = K7_N * 64;

__host__ __device__ constexpr int k7_wq_bytes(int C) { return 3 * (C / 32) * 4 * 1024; }       // (part, block, tile, plane) x 1 KiB
__host__ __device__ constexpr int k7_wp_bytes(int C) { return (C / 16) * 2 * 1024; }            // (tile, plane) x 1 KiB
__host__ __device__ constexpr int k7_head_bytes(int C) { return k7_wq_bytes(C) + k7_wp_bytes(C); }
__host__ __device__ constexpr int k7_lds_bytes(int C, bool proj = true) { return k7_wq_bytes(C) + (proj ? 2 * k7_wp_bytes(C) : 0) + 4 * K7_PL + K7_N * 4; }

// (a, b) -> packed f16 h and the packed f16 residual l = f16((x - h) * SC), SC = 1 (unscaled) or 2048 (swin_window_attn_h3.h)
template <bool SCALED>
__device__ __forceinline__ void k7_split2(float a, float b, uint32_t& h, uint32_t& l) {
  const k7_f16x2 hv = {(_Float16)a, (_Float16)b};
  h = __builtin_bit_cast(uint32_t, hv);
  const float m = SCALED ? -2048.0f : -1.0f;
  const float a2 = SCALED ? a * 2048.0f : a, b2 = SCALED ? b * 2048.0f : b;
  uint32_t r;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(m), "v"(a2));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(m), "v"(b2));
  l = r;
}
template <bool SCALED>
__device__ __forceinline__ void k7_split8(const float (&v)[8], k7_f16x8& h, k7_f16x8& l) {
  k7_u32x4 hp, lp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t a, r;
    k7_split2<SCALED>(v[2 * i], v[2 * i + 1], a, r);
    hp[i] = a;
    lp[i] = r;
  }
  h = __builtin_bit_cast(k7_f16x8, hp);
  l = __builtin_bit_cast(k7_f16x8, lp);
}

// ---- weight image.  Feature maps (i = MFMA A row 0..15, t / j = tile, e = element 0..7 of a lane's k piece, kk = lane / 16):
//   qkv  tile t of part p of head h: row i <-> W_qkv row  p C + 32 h + 8 (i / 4) + 4 t + i % 4,   k = 32 blk + 8 kk + e
//   proj tile j of head h:           row i <-> W_proj row 32 (j / 2) + 8 (i / 4) + 4 (j % 2) + i % 4,  k = 32 h + 16 (e / 4) + 4 kk + e % 4
// One thread per (fragment pair h | l, lane).
__global__ void swin_attn_block_pack_kernel(const float* __restrict__ wqkv, const float* __restrict__ wproj, unsigned char* __restrict__ img, int C) {
  const int NH = C / 32, NB = C / 32;
  const int qf = 3 * NB * 2, pf = C / 16;                      // fragment pairs per head
  const int64_t total = (int64_t)NH * (qf + pf) * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63), i = lane & 15, kk = lane >> 4;
    int f = (int)(idx >> 6);
    const int h = f / (qf + pf);
    f -= h * (qf + pf);
    float v[8];
    unsigned char* dst;
    if (f < qf) {
      const int t = f & 1, blk = (f >> 1) % NB, p = (f >> 1) / NB;
      const int row = p * C + 32 * h + 8 * (i >> 2) + 4 * t + (i & 3);
      const float* src = wqkv + (int64_t)row * C + 32 * blk + 8 * kk;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[e];
      dst = img + (int64_t)h * k7_head_bytes(C) + (int64_t)f * 2048 + lane * 16;
    } else {
      const int j = f - qf;
      const int row = 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
      const float* src = wproj + (int64_t)row * C + 32 * h + 4 * kk;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[16 * (e >> 2) + (e & 3)];
      dst = img + (int64_t)h * k7_head_bytes(C) + k7_wq_bytes(C) + (int64_t)j * 2048 + lane * 16;
    }
    k7_f16x8 hh, ll;
    k7_split8<true>(v, hh, ll);
    *reinterpret_cast<k7_u32x4*>(dst) = __builtin_bit_cast(k7_u32x4, hh);
    *reinterpret_cast<k7_u32x4*>(dst + 1024) = __builtin_bit_cast(k7_u32x4, ll);
  }
}

__device__ __forceinline__ k7_f16x8 k7_lds16(const unsigned char* p) {
  return __builtin_bit_cast(k7_f16x8, *reinterpret_cast<const k7_u32x4*>(p));
}

template <int C, bool LN2, bool PROJ = true>
__global__ __launch_bounds__(64 * K7_WAVES) void swin_attn_block_kernel(
    float* x, float* __restrict__ y2, const float* __restrict__ g1, const float* __restrict__ b1, float eps1,
    const unsigned char* __restrict__ img, const float* __restrict__ qkv_bias, const float* __restrict__ bias_frag,
    const float* __restrict__ proj_bias, const float* __restrict__ g2, const float* __restrict__ b2, float eps2, int H, int W, int Hp,
    int Wp, int shift, float scale, void* __restrict__ ofrag K7_DBG_PARAM) {
  K7_STAMP(0);
  constexpr int NH = C / 32, NB = C / 32, NJ = C / 16, NT = K7_NT, PL = K7_PL;
  constexpr int WQ = k7_wq_bytes(C), WP = k7_wp_bytes(C), HEADB = k7_head_bytes(C);
  extern __shared__ __attribute__((aligned(16))) unsigned char k7_lds[];
  unsigned char* const wq = k7_lds;
  unsigned char* const wp = k7_lds + WQ;                                          // two buffers (PROJ only)
  unsigned char* const Kh = wp + (PROJ ? 2 * WP : 0);                             // + PL: Kl; + 2 PL: Vh; + 3 PL: Vl
  int* const rid = reinterpret_cast<int*>(Kh + 4 * PL);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, kk = lane >> 4;
  const int wx = blockIdx.x, wy = blockIdx.y, b = blockIdx.z;

  // one head's chunk -> LDS: 1 KiB pieces round-robin over the waves (qkv pieces into wq, proj pieces into wp[h & 1])
  auto issue_dma = [&](int h) {
    const unsigned char* src = img + (int64_t)h * HEADB + lane * 16;
    unsigned char* pdst = wp + (h & 1) * WP;
    constexpr int NQ = WQ / 1024, NPIECE = PROJ ? HEADB / 1024 : NQ;           // (without PROJ the proj fragments of the image stay where they are)
#pragma unroll
    for (int p0 = 0; p0 < NPIECE; p0 += K7_WAVES) {
      const int p = p0 + wave;
      if (p < NPIECE) {
        unsigned char* d = p < NQ ? wq + p * 1024 : pdst + (p - NQ) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)p * 1024),
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
      }
    }
  };
  issue_dma(0);

  // ---- this lane's token: position in the shifted frame (r, c), source / destination pixel (rr, cc), region id of the shift mask
  const int qt = wave * 16 + l15;
  const int r = wy * K7_WS + qt / K7_WS, c = wx * K7_WS + qt % K7_WS;
  int rr = r + shift, cc = c + shift;
  rr = rr >= Hp ? rr - Hp : rr;
  cc = cc >= Wp ? cc - Wp : cc;
  const bool valid = rr < H && cc < W;
  const int64_t row = (int64_t)b * H * W + (valid ? (int64_t)rr * W + cc : 0);
  const bool need_mask = shift > 0 && (wy == (int)gridDim.y - 1 || wx == (int)gridDim.x - 1);
  if (kk == 0) {
    const int hid = r < Hp - K7_WS ? 0 : (r < Hp - shift ? 1 : 2);
    const int wid = c < Wp - K7_WS ? 0 : (c < Wp - shift ? 1 : 2);
    rid[qt] = hid * 3 + wid;
  }

  // ---- norm1 of the strip's rows, split once into the qkv GEMMs' B operand
  k7_f16x8 xh[NB], xl[NB];
  {
    float v[NB][8];
    const float* xr = x + row * C + 8 * kk;
    float sum = 0.f;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      k7_f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
      if (valid) {
        a0 = *reinterpret_cast<const k7_f32x4*>(xr + 32 * blk);
        a1 = *reinterpret_cast<const k7_f32x4*>(xr + 32 * blk + 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[blk][e] = a0[e]; v[blk][4 + e] = a1[e]; }
      sum += ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w));
    }
    sum += __shfl_xor(sum, 16, RBA_WAVE);
    sum += __shfl_xor(sum, 32, RBA_WAVE);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[blk][e] - mean;
        sq = fmaf(d, d, sq);
      }
    sq += __shfl_xor(sq, 16, RBA_WAVE);
    sq += __shfl_xor(sq, 32, RBA_WAVE);
    const float rstd = valid ? 1.0f / sqrtf(sq / (float)C + eps1) : 0.f;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const k7_f32x4 ga = *reinterpret_cast<const k7_f32x4*>(g1 + 32 * blk + 8 * kk), gb = *reinterpret_cast<const k7_f32x4*>(g1 + 32 * blk + 8 * kk + 4);
      const k7_f32x4 ba = *reinterpret_cast<const k7_f32x4*>(b1 + 32 * blk + 8 * kk), bb = *reinterpret_cast<const k7_f32x4*>(b1 + 32 * blk + 8 * kk + 4);
      float y[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = valid ? (v[blk][e] - mean) * rstd * ga[e] + ba[e] : 0.f;          // padding tokens: zero rows (swin.py:255-259 pads after norm1)
        y[4 + e] = valid ? (v[blk][4 + e] - mean) * rstd * gb[e] + bb[e] : 0.f;
      }
      k7_split8<true>(y, xh[blk], xl[blk]);
    }
  }

  K7_STAMP(1);                                                                   // norm1 + split done
  k7_f32x4 acc[PROJ ? NJ : 1];                                                   // proj output of the strip, all heads: tile j rows 4 kk + r
#pragma unroll
  for (int j = 0; j < (PROJ ? NJ : 1); ++j) acc[j] = (k7_f32x4){0.f, 0.f, 0.f, 0.f};

  // K fragment of key tile c: key c * 16 + l15, chunk kk; V transposing read of key tile c, d tile dt (swin_window_attn_h3.h)
  const int kperm = (0x1320 >> (4 * ((l15 >> 2) & 3))) & 3;                       // P = {0, 2, 3, 1}[(key >> 2) & 3]; key = 16 w + l15
  const int kfrag = l15 * 64 + ((kk ^ kperm) * 16);
  const int vfrag = (4 * kk + (l15 >> 2)) * 64 + (l15 & 3) * 8;
  const int kwr = qt * 64 + ((kk ^ kperm) * 16);                                   // where this lane's 8 channels of ITS token go
  const int vwr = qt * 64 + (((kk >> 1) ^ ((l15 >> 2) & 1)) * 32) + (kk & 1) * 16;

  for (int h = 0; h < NH; ++h) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                             // (B) chunk h landed; every wave has left head h - 1
    K7_STAMP(2 + 5 * h);
    // ---- q, k, v of the strip for head h
    k7_f16x8 qh, ql;
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
      const int p = pi == 0 ? 1 : (pi == 1 ? 2 : 0);                             // k, v, then q
      k7_f32x4 am[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, al[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        const unsigned char* f = wq + ((p * NB + blk) * 4) * 1024 + lane * 16;
        const k7_f16x8 w0h = k7_lds16(f), w0l = k7_lds16(f + 1024), w1h = k7_lds16(f + 2048), w1l = k7_lds16(f + 3072);
        am[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h, xh[blk], am[0], 0, 0, 0);
        am[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, xh[blk], am[1], 0, 0, 0);
        al[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0h, xl[blk], al[0], 0, 0, 0);
        al[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, xl[blk], al[1], 0, 0, 0);
        al[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0l, xh[blk], al[0], 0, 0, 0);
        al[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l, xh[blk], al[1], 0, 0, 0);
      }
      const float* bp = qkv_bias + p * C + 32 * h + 8 * kk;
      const k7_f32x4 bia = *reinterpret_cast<const k7_f32x4*>(bp), bib = *reinterpret_cast<const k7_f32x4*>(bp + 4);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaf(al[0][e], 0.00048828125f, am[0][e]) + bia[e];
        v[4 + e] = fmaf(al[1][e], 0.00048828125f, am[1][e]) + bib[e];
      }
      k7_f16x8 hh, ll;
      if (p == 1) {
        k7_split8<false>(v, hh, ll);
        *reinterpret_cast<k7_u32x4*>(Kh + kwr) = __builtin_bit_cast(k7_u32x4, hh);
        *reinterpret_cast<k7_u32x4*>(Kh + PL + kwr) = __builtin_bit_cast(k7_u32x4, ll);
      } else if (p == 2) {
        k7_split8<true>(v, hh, ll);
        *reinterpret_cast<k7_u32x4*>(Kh + 2 * PL + vwr) = __builtin_bit_cast(k7_u32x4, hh);
        *reinterpret_cast<k7_u32x4*>(Kh + 3 * PL + vwr) = __builtin_bit_cast(k7_u32x4, ll);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= scale;
        k7_split8<false>(v, qh, ql);
      }
    }
    K7_STAMP(3 + 5 * h);                                                         // q, k, v done
    __syncthreads();                                                             // (A) K / V planes complete; the qkv chunk is free
    K7_STAMP(4 + 5 * h);
    if (h + 1 < NH) issue_dma(h + 1);

    // ---- S^T tiles: lane holds S[key = c*16 + 4*kk + r][query = qt]; the relative-position bias is the accumulator's initial value
    const k7_f32x4* bfrag = reinterpret_cast<const k7_f32x4*>(bias_frag) + (((int64_t)h * NT + wave) * NT) * 64 + lane;
    k7_f32x4 S[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) S[ct] = bfrag[ct * 64];
    // two groups of key tiles (5 + 4: 20 fragment registers instead of 36); per accumulator the products arrive in K5's order kh.qh, kh.ql, kl.qh
#pragma unroll
    for (int g0 = 0; g0 < NT; g0 += 5) {
      constexpr int GN = 5;
      k7_f16x8 kf[GN];
#pragma unroll
      for (int i = 0; i < GN; ++i) if (g0 + i < NT) kf[i] = k7_lds16(Kh + (g0 + i) * 1024 + kfrag);
#pragma unroll
      for (int i = 0; i < GN; ++i) if (g0 + i < NT) S[g0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i], qh, S[g0 + i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < GN; ++i) if (g0 + i < NT) S[g0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i], ql, S[g0 + i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < GN; ++i) if (g0 + i < NT) kf[i] = k7_lds16(Kh + PL + (g0 + i) * 1024 + kfrag);
#pragma unroll
      for (int i = 0; i < GN; ++i) if (g0 + i < NT) S[g0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i], qh, S[g0 + i], 0, 0, 0);
    }
    float m = -INFINITY;
    if (need_mask) {
      const int myrid = rid[qt];
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const int4 kr4 = *reinterpret_cast<const int4*>(rid + ct * 16 + kk * 4);
        const int krid[4] = {kr4.x, kr4.y, kr4.z, kr4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (krid[e] != myrid) S[ct][e] += -100.0f;
      }
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, S[ct][e]);
    m = fmaxf(m, __shfl_xor(m, 16, RBA_WAVE));
    m = fmaxf(m, __shfl_xor(m, 32, RBA_WAVE));
    const float mneg = -m * 1.44269504088896340736f;
    f32x2 ls2 = {0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const f32x2 t = (f32x2){S[ct][e], S[ct][e + 1]} * 1.44269504088896340736f + mneg;
        const f32x2 pp = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        S[ct][e] = pp.x;
        S[ct][e + 1] = pp.y;
        ls2 += pp;
      }
    float lsum = ls2.x + ls2.y;
    lsum += __shfl_xor(lsum, 16, RBA_WAVE);
    lsum += __shfl_xor(lsum, 32, RBA_WAVE);
    // ---- O^T = V^T P^T: 32 keys per step, two 16-wide d tiles, main + low accumulators
    k7_f32x4 Om[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, Ol[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c0 = 0; c0 < NT; c0 += 2) {
      const bool two = c0 + 1 < NT;
      const int c1 = two ? c0 + 1 : c0;
      k7_u32x4 ph, pl;
      uint32_t x0, x1;
      k7_split2<true>(S[c0][0], S[c0][1], x0, x1); ph[0] = x0; pl[0] = x1;
      k7_split2<true>(S[c0][2], S[c0][3], x0, x1); ph[1] = x0; pl[1] = x1;
      if (two) {
        k7_split2<true>(S[c1][0], S[c1][1], x0, x1); ph[2] = x0; pl[2] = x1;
        k7_split2<true>(S[c1][2], S[c1][3], x0, x1); ph[3] = x0; pl[3] = x1;
      } else {
        ph[2] = ph[3] = pl[2] = pl[3] = 0u;
      }
      const k7_f16x8 pa = __builtin_bit_cast(k7_f16x8, ph), pb = __builtin_bit_cast(k7_f16x8, pl);
      k7_f16x8 vh[2], vl[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int half = (dt ^ (kk & 1)) * 32;
        const unsigned char* v0 = Kh + 2 * PL + c0 * 1024 + vfrag + half;
        const unsigned char* v1 = Kh + 2 * PL + c1 * 1024 + vfrag + half;
        const k7_h4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) k7_h4*)(v0));
        const k7_h4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) k7_h4*)(v1));
        const k7_h4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) k7_h4*)(v0 + PL));
        const k7_h4 r3 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) k7_h4*)(v1 + PL));
        const uint2 u0 = __builtin_bit_cast(uint2, r0), u1 = __builtin_bit_cast(uint2, r1);
        const uint2 u2 = __builtin_bit_cast(uint2, r2), u3 = __builtin_bit_cast(uint2, r3);
        vh[dt] = __builtin_bit_cast(k7_f16x8, (k7_u32x4){u0.x, u0.y, u1.x, u1.y});
        vl[dt] = __builtin_bit_cast(k7_f16x8, (k7_u32x4){u2.x, u2.y, u3.x, u3.y});
      }
      Om[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[0], pa, Om[0], 0, 0, 0);
      Om[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[1], pa, Om[1], 0, 0, 0);
      Ol[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[0], pa, Ol[0], 0, 0, 0);
      Ol[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[1], pa, Ol[1], 0, 0, 0);
      Ol[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[0], pb, Ol[0], 0, 0, 0);
      Ol[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[1], pb, Ol[1], 0, 0, 0);
    }
    K7_STAMP(5 + 5 * h);                                                         // attention done
    // ---- lane holds O[query qt][d = 16 dt + 4 kk + r]: split = the proj GEMM's B operand, k slot 8 kk + 4 dt + r
    const float inv = 1.0f / lsum;
    float o[8];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[4 * dt + e] = fmaf(Ol[dt][e], 0.00048828125f, Om[dt][e]) * inv;
    if (PROJ) {
      k7_f16x8 oh, ol;
      k7_split8<true>(o, oh, ol);
      const unsigned char* pf = wp + (h & 1) * WP + lane * 16;
#pragma unroll
      for (int j = 0; j < (PROJ ? NJ : 0); ++j) {
        const k7_f16x8 wh = k7_lds16(pf + j * 2048), wl = k7_lds16(pf + j * 2048 + 1024);
        k7_f32x4 lo = {0.f, 0.f, 0.f, 0.f};
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, oh, acc[j], 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ol, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, oh, lo, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(lo[e], 0.00048828125f, acc[j][e]);
      }
    } else {
      // attention output of (token, head) as the proj Linear's split A operand (split_linear_h3.h "PRE"; K5's split_out epilogue): channel
      // c = 32 h + 16 dt + 4 kk + r -> block h, piece (g = kk / 2, h | l), k-half dt; the 8-channel piece is completed by the lane of the
      // neighbouring kk (lane ^ 16): v_permlane16_swap(h, l) leaves an even-kk lane with (its h, the partner's h), an odd one with (the partner's l, its l)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        uint32_t h0, l0, h1, l1;
        rba_split_f16x2(o[4 * dt], o[4 * dt + 1], h0, l0);
        rba_split_f16x2(o[4 * dt + 2], o[4 * dt + 3], h1, l1);
        const auto p0 = __builtin_amdgcn_permlane16_swap(h0, l0, false, false), p1 = __builtin_amdgcn_permlane16_swap(h1, l1, false, false);
        const rba_u32x4 piece = {p0[0], p1[0], p0[1], p1[1]};
        if (valid) {
          char* dst = reinterpret_cast<char*>(ofrag) + ((row >> 5) * NH + h) * 4096 + ((kk >> 1) * 2 + (kk & 1)) * 1024 + (dt * 32 + (int)(row & 31)) * 16;
          *reinterpret_cast<rba_u32x4*>(dst) = piece;
        }
      }
    }
    K7_STAMP(6 + 5 * h);                                                         // proj done
  }

  if constexpr (PROJ) {
  // ---- x = (x + out) + proj.bias; tile j rows 4 kk + e <-> channel 32 (j / 2) + 8 kk + 4 (j % 2) + e: the layout of the first load
  float* xr = x + row * C + 8 * kk;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int ch = 32 * (j >> 1) + 4 * (j & 1);
    k7_f32x4 xo = {0.f, 0.f, 0.f, 0.f};
    if (valid) xo = *reinterpret_cast<const k7_f32x4*>(xr + ch);
    const k7_f32x4 pb = *reinterpret_cast<const k7_f32x4*>(proj_bias + ch + 8 * kk);
    acc[j] = (xo + acc[j]) + pb;
    if (valid) *reinterpret_cast<k7_f32x4*>(xr + ch) = acc[j];
    sum += (acc[j].x + acc[j].y) + (acc[j].z + acc[j].w);
  }
  if (LN2) {
    sum += __shfl_xor(sum, 16, RBA_WAVE);
    sum += __shfl_xor(sum, 32, RBA_WAVE);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = acc[j][e] - mean;
        sq = fmaf(d, d, sq);
      }
    sq += __shfl_xor(sq, 16, RBA_WAVE);
    sq += __shfl_xor(sq, 32, RBA_WAVE);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps2);
    float* yr = y2 + row * C + 8 * kk;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ch = 32 * (j >> 1) + 4 * (j & 1);
      const k7_f32x4 ga = *reinterpret_cast<const k7_f32x4*>(g2 + ch + 8 * kk), ba = *reinterpret_cast<const k7_f32x4*>(b2 + ch + 8 * kk);
      const k7_f32x4 yv = (acc[j] - mean) * rstd * ga + ba;
      if (valid) *reinterpret_cast<k7_f32x4*>(yr + ch) = yv;
    }
  }
  }  // PROJ
  K7_STAMP(31);
}

}  // namespace
