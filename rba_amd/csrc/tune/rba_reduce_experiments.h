// K1 experiments of round 2 that did NOT beat rba_reduce_pk_kernel (165 us, 63 % of 8 TB/s, tools/k1_sweep.py; counters in
// profiles/r02_k1_matrix_pipe.txt): two formulations of the class contraction on the matrix pipe that need NO LDS transposition.
//   mx (variant 200): bf16x6 MFMAs, 8 query planes x 4 pixels per lane = the 32x32x16 B fragment.   232-264 us (loads alone 198)
//   mf (variant 210): exact-fp32 MFMAs, v_permlane32_swap builds the B fragment of two pixel halves. 270-278 us (loads alone 204)
// Both are correct (max |d rba| 7.6e-6 at Q = 100, K = 19, 1024 x 2048) and both lose for the same reason: 64-128 accumulator
// registers leave two waves per SIMD, and with 8 waves per CU even their bare load streams reach only 4.1-4.3 TB/s, where the VALU
// kernel's 16 waves reach 6.6 TB/s.  Built only into librba_tune.so.
#pragma once
#include "../rba_reduce_kernels.h"

namespace rba_k1 {

// ---------------------------------------------------------------------------------------------------------------------------
// K1 on the matrix pipe without any transposition ("mx").  sem[k, p] = sum_q P[q, k] * sigmoid(mask[q, p]) is a GEMM with the
// classes on M (19 padded to 32), the pixels on N and the queries on the reduction axis.  The B operand of
// v_mfma_f32_32x32x16_bf16 wants, per lane, 8 CONSECUTIVE k (= queries) of ONE column (= pixel): lanes 0-31 hold k 0..7 and lanes
// 32-63 k 8..15 of column lane % 32.  A lane that issues eight 16-byte loads -- plane q0 + 8 (lane / 32) + j, j = 0..7, at pixels
// 4 (lane % 32) .. + 3 -- holds exactly that fragment for FOUR MFMAs (one per pixel of its quadruple): the loads stay 16 B per lane
// and fully coalesced (two 512-byte runs per wave instruction), nothing goes through LDS, and the 19 x 100 multiply-adds per
// pixel leave the VALU, which keeps the sigmoid (4 instructions) and the split of sigma into three bf16 (hi + mid + lo = sigma
// exactly, as in K6; the class probabilities are split once per workgroup into LDS): ~9.5 VALU per mask element instead of ~16.
// Six bf16 products per fp32 product keep fp32 accuracy (measured: |d sem| <= 2e-6 like the VALU kernel).
// One wave = 128 pixels x all queries; accumulators D[class][pixel]: lane holds pixel 4 (lane % 32) + i of MFMA i, classes
// 8 (r / 4) + 4 (lane / 32) + r % 4; the class reduction finishes with one cross-half shuffle.  Score only (no sem_seg / argmax
// output): the bench / rba_scores() path.  Tiles are handed out per WAVE, four at a time, by an atomic counter (no barrier).
typedef __bf16 mx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mx_f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t mx_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mx_pack(float x0, float x1) {
  mx_bf16x2 v = {(__bf16)x0, (__bf16)x1};
  return __builtin_bit_cast(uint32_t, v);
}
// two values at a time, everything that has a packed form as v_pk_*: sigmoid = v_pk_mul, 2 v_exp, v_pk_add, 2 v_rcp (raw
// v_exp_f32: exp2 of -x log2(e); an overflow to inf gives rcp(inf) = 0, an underflow to 0 gives 1 -- the limits of the sigmoid);
// split = v_cvt_pk_bf16_f32, two unpack ops, v_pk_add (exact remainder), twice, and a last v_cvt_pk: 15 VALU per pair.
__device__ __forceinline__ f32x2 mx_sigmoid2(f32x2 x) {
  const f32x2 t = x * (f32x2){-1.44269504088896340736f, -1.44269504088896340736f};
  const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + (f32x2){1.0f, 1.0f};
  return (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ void mx_split2(f32x2 v, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = mx_pack(v.x, v.y);
  const f32x2 r = v - (f32x2){__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  m = mx_pack(r.x, r.y);
  const f32x2 q = r - (f32x2){__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  l = mx_pack(q.x, q.y);
}
__device__ __forceinline__ void mx_split8(const float (&x)[8], mx_bf16x8& p0, mx_bf16x8& p1, mx_bf16x8& p2) {
  mx_u32x4 h, m, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t a, b, c;
    mx_split2((f32x2){x[2 * i], x[2 * i + 1]}, a, b, c);
    h[i] = a;
    m[i] = b;
    l[i] = c;
  }
  p0 = __builtin_bit_cast(mx_bf16x8, h);
  p1 = __builtin_bit_cast(mx_bf16x8, m);
  p2 = __builtin_bit_cast(mx_bf16x8, l);
}

constexpr int MX_CHUNK = 4;                                        // 128-pixel tiles per dequeue (even: the step loop is unrolled by 2)

// PROBE (tune builds; wrong results): 1 = loads only (the arithmetic replaced by one add per loaded register), 2 = arithmetic only
// (every step re-uses the first tile's loads)
template <int WPS, int PROBE = 0>
__global__ __launch_bounds__(256, WPS) void rba_reduce_mx_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                               float* __restrict__ rba, int Q, int K, int64_t HW, int ntiles, int mode,
                                                               unsigned int* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) mx_u32x4 mx_pfrag[];          // [G][3 planes][64 lanes]: A fragments of P
  const int G = (Q + 15) >> 4;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  for (int idx = tid; idx < G * 64; idx += 256) {
    const int g = idx >> 6, l = idx & 63, m = l & 31, h = l >> 5;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = 16 * g + 8 * h + i;
      x[i] = (q < Q && m < K) ? prob[q * K + m] : 0.f;
    }
    mx_bf16x8 p0, p1, p2;
    mx_split8(x, p0, p1, p2);
    mx_pfrag[(g * 3 + 0) * 64 + l] = __builtin_bit_cast(mx_u32x4, p0);
    mx_pfrag[(g * 3 + 1) * 64 + l] = __builtin_bit_cast(mx_u32x4, p1);
    mx_pfrag[(g * 3 + 2) * 64 + l] = __builtin_bit_cast(mx_u32x4, p2);
  }
  __syncthreads();

  const int nchunks = (ntiles + MX_CHUNK - 1) / MX_CHUNK;
  const int S = MX_CHUNK * G;                                      // (tile, query group) steps per chunk
  auto dequeue = [&]() -> int {
    unsigned int c = 0;
    if (lane == 0) c = atomicAdd(counters, 1u);
    return (int)__builtin_amdgcn_readfirstlane(c);
  };
  // loads of step (tile t, group g): plane q = 16 g + 8 lh + j (clamped: the P fragment is zero there), pixels 4 l31 .. + 3 of tile t
  auto load = [&](f32x4 (&buf)[8], int t, int g) {
    if (PROBE == 2) { t = 0; g = 0; }
    int64_t pix = (int64_t)t * 128 + 4 * l31;
    pix = pix < HW ? pix : HW - 4;
    const float* base = mask + pix;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int q = 16 * g + 8 * lh + j;
      q = q < Q ? q : Q - 1;
      buf[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (int64_t)q * HW));
    }
  };
  mx_f32x16 acc[4];
  auto compute = [&](const f32x4 (&buf)[8], int t, int g) {
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    if (PROBE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] += buf[j][i];
    } else {
    mx_bf16x8 a[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) a[p] = __builtin_bit_cast(mx_bf16x8, mx_pfrag[(g * 3 + p) * 64 + lane]);
    // sigma and its three bf16 planes for all four pixels first, then the 24 MFMAs round-robin over the four accumulators: a
    // dependent MFMA would otherwise stall the in-order wave (and every VALU instruction behind it) for its full latency
    mx_bf16x8 b[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mx_u32x4 h, m, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t ha, ma, la;
        mx_split2(mx_sigmoid2((f32x2){buf[2 * j][i], buf[2 * j + 1][i]}), ha, ma, la);
        h[j] = ha;
        m[j] = ma;
        l[j] = la;
      }
      b[i][0] = __builtin_bit_cast(mx_bf16x8, h);
      b[i][1] = __builtin_bit_cast(mx_bf16x8, m);
      b[i][2] = __builtin_bit_cast(mx_bf16x8, l);
    }
#define RBA_MX(pa, pb) \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[i][pb], acc[i], 0, 0, 0);
    RBA_MX(2, 0) RBA_MX(0, 2) RBA_MX(1, 1) RBA_MX(1, 0) RBA_MX(0, 1) RBA_MX(0, 0)      // smallest terms first
#undef RBA_MX
    }
    if (g == G - 1) {                                              // ---- epilogue of tile t
      // classes of this lane: 8 (r / 4) + 4 lh + r % 4; rows >= K are exactly 0 (zero P fragment): tanh(0) = 0 and 0 add nothing
      float out[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float part = 0.f;
        if (mode == 1) {
          float mx = -3.0e38f;
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, 8 * (r >> 2) + 4 * lh + (r & 3) < K ? acc[i][r] : -3.0e38f);
          mx = fmaxf(mx, __shfl_xor(mx, 32, RBA_WAVE));
#pragma unroll
          for (int r = 0; r < 16; ++r) part += 8 * (r >> 2) + 4 * lh + (r & 3) < K ? __expf(acc[i][r] - mx) : 0.f;
          part += __shfl_xor(part, 32, RBA_WAVE);
          out[i] = -(mx + __logf(part));
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) part += mode == 0 ? rba_tanh(acc[i][r]) : acc[i][r];
          part += __shfl_xor(part, 32, RBA_WAVE);
          out[i] = -part;
        }
      }
      const int64_t pix = (int64_t)t * 128 + 4 * l31;
      if (lh == 0 && pix < HW) *reinterpret_cast<f32x4*>(rba + pix) = (f32x4){out[0], out[1], out[2], out[3]};
    }
  };

  f32x4 bufA[8], bufB[8];
  int chunk = dequeue();
  while (chunk < nchunks) {
    const int nxt_chunk = dequeue();                               // one chunk ahead: its first loads overlap this chunk's last step
    const int t0 = chunk * MX_CHUNK;
    int t = t0, g = 0;
    load(bufA, t < ntiles ? t : ntiles - 1, 0);
    for (int s = 0; s < S; s += 2) {
      int t1 = t, g1 = g + 1;
      if (g1 == G) { g1 = 0; ++t1; }
      load(bufB, t1 < ntiles ? t1 : ntiles - 1, g1);
      if (t < ntiles) compute(bufA, t, g);
      int t2 = t1, g2 = g1 + 1;
      if (g2 == G) { g2 = 0; ++t2; }
      if (s + 2 < S) load(bufA, t2 < ntiles ? t2 : ntiles - 1, g2);
      if (t1 < ntiles) compute(bufB, t1, g1);
      t = t2;
      g = g2;
    }
    chunk = nxt_chunk;
  }
  __syncthreads();
  if (tid == 0) {
    const unsigned int done = atomicAdd(counters + 1, 1u);
    if (done == gridDim.x - 1) {                                   // every wave of every workgroup has fetched its last chunk
      atomicExch(counters, 0u);
      atomicExch(counters + 1, 0u);
    }
  }
}

template <int WPS, int PROBE = 0>
int launch_reduce_mx(const float* mask, const float* prob, float* rba, int Q, int K, int64_t HW, int mode, unsigned int* counters,
                     hipStream_t st) {
  const int64_t tiles = (HW + 127) / 128;
  if (tiles > 0x7fffffffLL || Q > 1024) return (int)hipErrorInvalidValue;
  const int G = (Q + 15) / 16;
  const size_t dyn = (size_t)G * 3 * 64 * 16;
  const int64_t chunks = (tiles + MX_CHUNK - 1) / MX_CHUNK;
  int64_t grid = 256 * WPS;
  grid = chunks < grid * 4 ? (chunks + 3) / 4 : grid;
  grid = grid < 1 ? 1 : grid;
  hipLaunchKernelGGL((rba_reduce_mx_kernel<WPS, PROBE>), dim3((unsigned)grid), dim3(256), dyn, st, mask, prob, rba, Q, K, HW, (int)tiles, mode,
                     counters);
  return rba_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------------
// K1 on the f16 matrix pipe, register-light ("h3"): the mx idea (no transposition: the 32x32x16 B fragment of a lane IS 8 query
// planes of one pixel) with what made mx lose removed --
//   * THREE f16 MFMAs per product instead of six bf16 ones, and a two-piece split (sigma = h + l, h = f16(sigma), l = f16(sigma - h):
//     sigma and the class probabilities live in (0, 1), so the unscaled residual's f16 underflow is an ABSOLUTE error <= 2^-25 per
//     term, 3e-6 on a 100-query sum at the very worst, far inside the 1e-4 score tolerance): 4 VALU per pair instead of ~10;
//   * 64 pixels per wave step (two pixels per lane, 8-byte loads) instead of 128: 32 accumulator registers instead of 64, ~100
//     registers in all -> four waves per SIMD like the VALU kernel, each with 16-32 plane loads in flight (the VALU kernel: 2).
// Step (tile, g): lane (l31, lh) loads planes q = 16 g + 8 lh + i, i < 8, at pixels 2 l31, 2 l31 + 1 of the tile (256 contiguous bytes
// per plane and half-wave); sigmoid; pack (q = 2 j, 2 j + 1) pairs per pixel -> the f16x8 B operands of the two 32-pixel MFMA tiles;
// A = the split class-probability fragment of query group g from LDS (built once per workgroup).
template <typename KernelT>
static inline int ensure_dynamic_lds_k1(KernelT kernel, size_t bytes, unsigned char (&done)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
  if (!done[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    done[dev] = 1;
  }
  return 0;
}

typedef _Float16 h3k_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h3k_f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void h3k_split2(float a, float b, uint32_t& h, uint32_t& l) {          // (a, b) -> packed f16 h, l = f16(x - h)
  const h3k_f16x2 hv = {(_Float16)a, (_Float16)b};
  h = __builtin_bit_cast(uint32_t, hv);
  uint32_t r;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(-1.0f), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(-1.0f), "v"(b));
  l = r;
}

// PROBE: 1 = loads only, 2 = arithmetic only (every step re-uses the first tile's planes)
template <int WPS, int PROBE = 0>
__global__ __launch_bounds__(256, WPS) void rba_reduce_h3_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                               float* __restrict__ rba, int Q, int K, int64_t HW, int ntiles,
                                                               unsigned int* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) mx_u32x4 h3k_pfrag[];         // [G][2 planes][64 lanes]: A fragments of P^T
  const int G = (Q + 15) >> 4;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  for (int idx = tid; idx < G * 64; idx += 256) {
    const int g = idx >> 6, l = idx & 63, m = l & 31, hh = l >> 5;
    mx_u32x4 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q0 = 16 * g + 8 * hh + 2 * j;
      const float a = (q0 < Q && m < K) ? prob[q0 * K + m] : 0.f;
      const float b = (q0 + 1 < Q && m < K) ? prob[(q0 + 1) * K + m] : 0.f;
      uint32_t h, lo;
      h3k_split2(a, b, h, lo);
      ph[j] = h;
      pl[j] = lo;
    }
    h3k_pfrag[(g * 2 + 0) * 64 + l] = ph;
    h3k_pfrag[(g * 2 + 1) * 64 + l] = pl;
  }
  __syncthreads();

  // a wave works on 64-pixel tiles; the four waves of a workgroup take consecutive tiles of a 256-pixel chunk
  const int wave = tid >> 6;
  const int nchunks = (ntiles + 3) >> 2;
  auto load = [&](f32x2 (&buf)[8], int64_t tile, int g) {
    if (PROBE == 2) { tile = wave; g = 0; }
    int64_t pix = tile * 64 + 2 * l31;
    pix = pix < HW ? pix : HW - 2;
    const float* base = mask + pix;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int q = 16 * g + 8 * lh + i;
      q = q < Q ? q : Q - 1;
      buf[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(base + (int64_t)q * HW));
    }
  };
  mx_f32x16 acc[2];
  auto compute = [&](const f32x2 (&buf)[8], int g) {
    if (g == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
    if (PROBE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc[0][i] += buf[i].x; acc[1][i] += buf[i].y; }
      return;
    }
    const h3k_f16x8 ah = __builtin_bit_cast(h3k_f16x8, h3k_pfrag[(g * 2 + 0) * 64 + lane]);
    const h3k_f16x8 al = __builtin_bit_cast(h3k_f16x8, h3k_pfrag[(g * 2 + 1) * 64 + lane]);
    f32x2 sg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sg[i] = rba_sigmoid2(buf[i]);
    mx_u32x4 bh[2], bl[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t h0, l0, h1, l1;
      h3k_split2(sg[2 * j].x, sg[2 * j + 1].x, h0, l0);
      h3k_split2(sg[2 * j].y, sg[2 * j + 1].y, h1, l1);
      bh[0][j] = h0; bl[0][j] = l0; bh[1][j] = h1; bl[1][j] = l1;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(h3k_f16x8, bh[t]), acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(h3k_f16x8, bl[t]), acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(h3k_f16x8, bh[t]), acc[t], 0, 0, 0);
  };
  auto finish = [&](int64_t tile) {
    // lane holds sem[class = 8 (r / 4) + 4 lh + r % 4][pixel 2 l31 + t of the tile]
    float out[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc += (8 * (r >> 2) + 4 * lh + (r & 3) < K) ? rba_tanh(acc[t][r]) : 0.f;
      sacc += __shfl_xor(sacc, 32, RBA_WAVE);
      out[t] = -sacc;
    }
    const int64_t pix = tile * 64 + 2 * l31;
    if (lh == 0 && pix + 1 < HW) *reinterpret_cast<f32x2*>(rba + pix) = (f32x2){out[0], out[1]};
    else if (lh == 0 && pix < HW) rba[pix] = out[0];
  };

  auto dequeue = [&]() -> int {
    __syncthreads();
    __shared__ unsigned int sh_chunk;
    if (tid == 0) sh_chunk = atomicAdd(counters, 1u);
    __syncthreads();
    return (int)sh_chunk;
  };
  f32x2 bufA[8], bufB[8];
  for (int chunk = dequeue(); chunk < nchunks; chunk = dequeue()) {
    const int64_t tile = (int64_t)chunk * 4 + wave;
    if (tile >= ntiles) continue;
    load(bufA, tile, 0);
    for (int g = 0; g < G; g += 2) {                                   // two query groups per trip: the buffers alternate statically
      if (g + 1 < G) load(bufB, tile, g + 1);
      compute(bufA, g);
      if (g + 2 < G) load(bufA, tile, g + 2);
      if (g + 1 < G) compute(bufB, g + 1);
    }
    finish(tile);
  }
  if (tid == 0) {
    const unsigned int done = atomicAdd(counters + 1, 1u);
    if (done == gridDim.x - 1) {
      atomicExch(counters, 0u);
      atomicExch(counters + 1, 0u);
    }
  }
}

template <int WPS, int PROBE = 0>
int launch_reduce_h3(const float* mask, const float* prob, float* rba, int Q, int K, int64_t HW, unsigned int* counters, hipStream_t st) {
  const int64_t tiles = (HW + 63) / 64;
  if (tiles > 0x7fffffffLL || Q > 1024 || K > 32) return (int)hipErrorInvalidValue;
  const int G = (Q + 15) / 16;
  const size_t dyn = (size_t)G * 2 * 64 * 16;
  const int64_t chunks = (tiles + 3) / 4;
  int64_t grid = 256 * WPS;
  grid = chunks < grid ? chunks : grid;
  grid = grid < 1 ? 1 : grid;
  hipLaunchKernelGGL((rba_reduce_h3_kernel<WPS, PROBE>), dim3((unsigned)grid), dim3(256), dyn, st, mask, prob, rba, Q, K, HW, (int)tiles, counters);
  return rba_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------------
// K1 on the f16 matrix pipe with the VALU kernel's LOAD PATTERN ("tr"): every wave instruction reads 1 KiB of ONE plane (lane = 4
// consecutive pixels), sigma is computed and split (h, l) where it was loaded, written to LDS as f16 rows [query][pixel], and the MFMA
// B operand (8 consecutive queries of one pixel per lane) is fetched with the TRANSPOSING LDS read ds_read_b64_tr_b16: a 16-lane group
// reads a [4 queries][16 pixels] block and every lane receives its pixel's column (probe: tools/micro/tr_probe).
//   workgroup tile 256 pixels; super-step = 32 queries: wave w loads planes 32 g + 8 w + i (i < 8) for all 256 pixels, then computes
//   its own 64 pixels (4 pixel tiles x 2 class tiles x 3 products = 24 v_mfma_f32_16x16x32_f16) over all 32 queries.
//   LDS sigma image: [h | l][16 pixel tiles] x 1056 B, a tile = [half 2][kg 4][4 queries][16 pixels] f16 (+32 B pad: the 8-byte
//   writes of 16 consecutive lanes hit 4 tiles x 32 B -> distinct banks); the two tr reads of a lane: tile + half 512 + lane 8.
typedef __fp16 trk_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float trk_f32x4 __attribute__((ext_vector_type(4)));

template <int WPS, int PROBE = 0>
__global__ __launch_bounds__(256, WPS) void rba_reduce_tr_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                               float* __restrict__ rba, int Q, int K, int64_t HW, int ntiles,
                                                               unsigned int* __restrict__ counters) {
  constexpr int TS = 1056, PS = 16 * TS;                                       // bytes per pixel tile / per plane part (h or l)
  extern __shared__ __attribute__((aligned(16))) unsigned char trk_lds[];
  const int G = (Q + 31) >> 5;
  unsigned char* sig = trk_lds;                                                // 2 PS bytes
  mx_u32x4* pfrag = reinterpret_cast<mx_u32x4*>(trk_lds + 2 * PS);             // [G][mt 2][part 2][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int idx = tid; idx < G * 2 * 64; idx += 256) {
    const int l = idx & 63, mt = (idx >> 6) & 1, g = idx >> 7;
    const int m = (l & 15) + 16 * mt, kg = l >> 4;
    mx_u32x4 ph, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q0 = 32 * g + 8 * kg + 2 * j;
      const float a = (q0 < Q && m < K) ? prob[q0 * K + m] : 0.f;
      const float b = (q0 + 1 < Q && m < K) ? prob[(q0 + 1) * K + m] : 0.f;
      uint32_t h, lo;
      h3k_split2(a, b, h, lo);
      ph[j] = h;
      pl[j] = lo;
    }
    pfrag[((g * 2 + mt) * 2 + 0) * 64 + l] = ph;
    pfrag[((g * 2 + mt) * 2 + 1) * 64 + l] = pl;
  }

  // this thread's slot in the sigma image: pixel tile lane / 4, column group lane % 4, kg = wave; + half 512 + row 32 per plane
  const int wslot = (lane >> 2) * TS + wave * 128 + (lane & 3) * 8;
  const int rslot = lane * 8;                                                  // + (4 wave + t) TS + half 512 + part PS
  auto load = [&](f32x4 (&buf)[8], int64_t tile, int g) {
    if (PROBE == 2) { tile = 0; g = 0; }
    int64_t pix = tile * 256 + 4 * lane;
    pix = pix < HW ? pix : HW - 4;
    const float* base = mask + pix;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int q = 32 * g + 8 * wave + i;
      q = q < Q ? q : Q - 1;
      buf[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (int64_t)q * HW));
    }
  };
  trk_f32x4 acc[4][2];
  auto stage = [&](const f32x4 (&buf)[8]) {                                    // sigmoid + split + LDS write of this thread's 8 planes
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x2 s01 = rba_sigmoid2((f32x2){buf[i].x, buf[i].y}), s23 = rba_sigmoid2((f32x2){buf[i].z, buf[i].w});
      uint32_t h0, l0, h1, l1;
      h3k_split2(s01.x, s01.y, h0, l0);
      h3k_split2(s23.x, s23.y, h1, l1);
      const int off = wslot + (i >> 2) * 512 + (i & 3) * 32;
      *reinterpret_cast<uint2*>(sig + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(sig + PS + off) = make_uint2(l0, l1);
    }
  };
  auto compute = [&](int g) {
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    f16x8v ah[2], al[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      ah[mt] = __builtin_bit_cast(f16x8v, pfrag[((g * 2 + mt) * 2 + 0) * 64 + lane]);
      al[mt] = __builtin_bit_cast(f16x8v, pfrag[((g * 2 + mt) * 2 + 1) * 64 + lane]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned char* tb = sig + (4 * wave + t) * TS + rslot;
      trk_h4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trk_h4*)(tb));
      trk_h4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trk_h4*)(tb + 512));
      trk_h4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trk_h4*)(tb + PS));
      trk_h4 r3 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) trk_h4*)(tb + PS + 512));
      const uint2 u0 = __builtin_bit_cast(uint2, r0), u1 = __builtin_bit_cast(uint2, r1);
      const uint2 u2 = __builtin_bit_cast(uint2, r2), u3 = __builtin_bit_cast(uint2, r3);
      const f16x8v bh = __builtin_bit_cast(f16x8v, (mx_u32x4){u0.x, u0.y, u1.x, u1.y});
      const f16x8v bl = __builtin_bit_cast(f16x8v, (mx_u32x4){u2.x, u2.y, u3.x, u3.y});
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, acc[t][mt], 0, 0, 0);
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, acc[t][mt], 0, 0, 0);
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh, acc[t][mt], 0, 0, 0);
      }
    }
  };
  auto finish = [&](int64_t tile) {
    // lane holds sem[class = 16 mt + 4 (lane >> 4) + r][pixel 64 wave + 16 t + (lane & 15)]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float sacc = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc += (16 * mt + 4 * (lane >> 4) + r < K) ? rba_tanh(acc[t][mt][r]) : 0.f;
      sacc += __shfl_xor(sacc, 16, RBA_WAVE);
      sacc += __shfl_xor(sacc, 32, RBA_WAVE);
      const int64_t pix = tile * 256 + 64 * wave + 16 * t + (lane & 15);
      if (lane < 16 && pix < HW) rba[pix] = -sacc;
    }
  };

  __shared__ unsigned int sh_tile;
  auto dequeue = [&]() -> int {
    __syncthreads();
    if (tid == 0) sh_tile = atomicAdd(counters, 1u);
    __syncthreads();
    return (int)sh_tile;
  };
  f32x4 buf[8];
  for (int tile = dequeue(); tile < ntiles; tile = dequeue()) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[t][mt] = (trk_f32x4){0.f, 0.f, 0.f, 0.f};
    load(buf, tile, 0);
    for (int g = 0; g < G; ++g) {
      if (PROBE != 1) stage(buf);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i & 3][i >> 2] += (trk_f32x4){buf[i].x, buf[i].y, buf[i].z, buf[i].w};
      }
      if (g + 1 < G) load(buf, tile, g + 1);
      __syncthreads();
      if (PROBE != 1) compute(g);
      __syncthreads();
    }
    finish(tile);
  }
  if (tid == 0) {
    const unsigned int done = atomicAdd(counters + 1, 1u);
    if (done == gridDim.x - 1) {
      atomicExch(counters, 0u);
      atomicExch(counters + 1, 0u);
    }
  }
}

template <int WPS, int PROBE = 0>
int launch_reduce_tr(const float* mask, const float* prob, float* rba, int Q, int K, int64_t HW, unsigned int* counters, hipStream_t st) {
  const int64_t tiles = (HW + 255) / 256;
  if (tiles > 0x7fffffffLL || Q > 256 || K > 32 || (HW & 3)) return (int)hipErrorInvalidValue;
  const int G = (Q + 31) / 32;
  const size_t dyn = (size_t)2 * 16 * 1056 + (size_t)G * 4 * 64 * 16;
  static unsigned char done_attr[64] = {0};
  int rc = ensure_dynamic_lds_k1(rba_reduce_tr_kernel<WPS, PROBE>, dyn, done_attr);
  if (rc) return rc;
  int64_t grid = 256 * WPS;
  grid = tiles < grid ? tiles : grid;
  grid = grid < 1 ? 1 : grid;
  hipLaunchKernelGGL((rba_reduce_tr_kernel<WPS, PROBE>), dim3((unsigned)grid), dim3(256), dyn, st, mask, prob, rba, Q, K, HW, (int)tiles, counters);
  return rba_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------------
// K1 with the class contraction on the EXACT-fp32 matrix pipe ("mf"): v_mfma_f32_32x32x2_f32, D[class][pixel] += P[q][class] *
// sigma[q][pixel] for TWO queries per instruction, bitwise a k-ordered fmaf chain -- the same ascending-q order as the VALU kernels,
// no operand splitting.  Loads keep the VALU kernels' pattern (lane l = pixels 4 l .. 4 l + 3 of ONE plane, 1 KiB contiguous per
// wave instruction).  The MFMA B operand wants lanes 0-31 = query k0 and lanes 32-63 = query k1 of the SAME 32 pixels: one
// v_permlane32_swap of the sigma registers of planes q and q + 1 produces exactly that for BOTH pixel halves of the wave's 256-pixel
// tile (upper half of the first register <-> lower half of the second), so nothing goes through LDS and the VALU keeps only the
// sigmoid (4 instructions per element) and one swap per dword.  Matrix-pipe time at Q = 100, K = 19 (padded to 32 rows): 93 us;
// VALU ~35 us; both hide behind the 128 us the load pattern needs.  8 accumulators x 16 registers (AGPRs): 2 waves per SIMD,
// latency covered by a 4-pair (8 KiB per wave) register ring.  Score only (no sem_seg / argmax output).
typedef float mf_f32x16 __attribute__((ext_vector_type(16)));
constexpr int MF_RING = 4;                                          // plane pairs in flight per wave
constexpr int MF_CHUNK = 2;                                         // 256-pixel tiles per dequeue

// Inline asm: with the builtin (ROCm 7.2) hipcc used the FIRST result register for both outputs (both MFMAs of a pair got the same
// operand).  The two v_nop are the VALU-write -> permlane-read wait states, the trailing s_nop the VALU-write -> MFMA-operand ones.
__device__ __forceinline__ void mf_swap(float& lo, float& hi) {     // lanes 32-63 of lo <-> lanes 0-31 of hi
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
}
__device__ __forceinline__ float mf_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

template <int WPS, int PROBE = 0>
__global__ __launch_bounds__(256, WPS) void rba_reduce_mf_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                               float* __restrict__ rba, int Q, int K, int64_t HW, int ntiles, int mode,
                                                               unsigned int* __restrict__ counters) {
  extern __shared__ float mf_afrag[];                               // [NPpad][64]: A operand of pair p, lane l: P[2 p + l / 32][l % 32]
  const int NP = (Q + 1) >> 1, NPpad = (NP + MF_RING - 1) / MF_RING * MF_RING;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int idx = tid; idx < NPpad * 64; idx += 256) {
    const int p = idx >> 6, l = idx & 63, q = 2 * p + (l >> 5), m = l & 31;
    mf_afrag[idx] = (q < Q && m < K) ? prob[q * K + m] : 0.f;
  }
  __syncthreads();
  const int nchunks = (ntiles + MF_CHUNK - 1) / MF_CHUNK;
  auto dequeue = [&]() -> int {
    unsigned int c = 0;
    if (lane == 0) c = atomicAdd(counters, 1u);
    return (int)__builtin_amdgcn_readfirstlane(c);
  };
  auto load = [&](f32x4 (&r)[2], int tile, int pair) {
    if (PROBE == 2) { tile = 0; pair = 0; }
    tile = tile < ntiles ? tile : ntiles - 1;
    int64_t pix = (int64_t)tile * 256 + 4 * lane;
    pix = pix < HW ? pix : HW - 4;
    const int q0 = 2 * pair < Q ? 2 * pair : Q - 1, q1 = 2 * pair + 1 < Q ? 2 * pair + 1 : Q - 1;    // clamped planes meet a zero A operand
    r[0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mask + (int64_t)q0 * HW + pix));
    r[1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mask + (int64_t)q1 * HW + pix));
  };
  mf_f32x16 accA[4], accB[4];
  auto compute = [&](const f32x4 (&r)[2], int tile, int pair) {
    if (pair == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) accA[c][i] = accB[c][i] = 0.f;
    }
    if (PROBE == 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { accA[c][0] += r[0][c]; accB[c][0] += r[1][c]; }
    } else {
      const float a = mf_afrag[pair * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s0 = mf_sigmoid(r[0][c]), s1 = mf_sigmoid(r[1][c]);
        mf_swap(s0, s1);
        accA[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s0, accA[c], 0, 0, 0);
        accB[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s1, accB[c], 0, 0, 0);
      }
    }
    if (pair == NPpad - 1 && tile < ntiles) {                       // ---- epilogue: rows (classes) >= K are exactly zero
      f32x4 out;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float pa = 0.f, pb = 0.f;
        if (mode == 1) {
          float ma = -3.0e38f, mb = -3.0e38f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const bool ok = 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3) < K;
            ma = fmaxf(ma, ok ? accA[c][i] : -3.0e38f);
            mb = fmaxf(mb, ok ? accB[c][i] : -3.0e38f);
          }
          mf_swap(ma, mb);
          const float mx = fmaxf(ma, mb);                           // lanes 0-31: max over all classes of set A, lanes 32-63: set B
          float mxa = mx, mxb = mx;
          mf_swap(mxa, mxb);                                        // now every lane has set A's max in mxa and set B's in mxb
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const bool ok = 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3) < K;
            pa += ok ? __expf(accA[c][i] - mxa) : 0.f;
            pb += ok ? __expf(accB[c][i] - mxb) : 0.f;
          }
          mf_swap(pa, pb);
          out[c] = -(mx + __logf(pa + pb));
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            pa += mode == 0 ? rba_tanh(accA[c][i]) : accA[c][i];
            pb += mode == 0 ? rba_tanh(accB[c][i]) : accB[c][i];
          }
          mf_swap(pa, pb);                                          // lanes 0-31: both halves' set-A parts, lanes 32-63: set-B parts
          out[c] = -(pa + pb);
        }
      }
      const int64_t pix = (int64_t)tile * 256 + 4 * lane;
      if (pix < HW) *reinterpret_cast<f32x4*>(rba + pix) = out;
    }
  };

  f32x4 ring[MF_RING][2];
  int chunk = dequeue();
  int lt = chunk * MF_CHUNK, lp = 0;                                // loader cursor (tile, pair), MF_RING steps ahead of the compute cursor
  int nxt_chunk = dequeue();
#pragma unroll
  for (int k = 0; k < MF_RING; ++k) load(ring[k], lt, lp++);        // NPpad >= MF_RING: still inside the first tile
  while (chunk < nchunks) {
    int ct = chunk * MF_CHUNK, cp = 0;
    const int chunk_end = (chunk + 1) * MF_CHUNK;
    const int S = MF_CHUNK * NPpad;
    for (int s = 0; s < S; s += MF_RING) {
#pragma unroll
      for (int k = 0; k < MF_RING; ++k) {
        compute(ring[k], ct, cp);
        if (++cp == NPpad) { cp = 0; ++ct; }
        // the loader cursor crosses into the next chunk MF_RING steps before this chunk ends
        load(ring[k], lt, lp);
        if (++lp == NPpad) {
          lp = 0;
          if (++lt == chunk_end) lt = nxt_chunk * MF_CHUNK;         // run on into the next chunk (load() clamps when there is none)
        }
      }
    }
    chunk = nxt_chunk;
    nxt_chunk = dequeue();
  }
  __syncthreads();
  if (tid == 0) {
    const unsigned int done = atomicAdd(counters + 1, 1u);
    if (done == gridDim.x - 1) {
      atomicExch(counters, 0u);
      atomicExch(counters + 1, 0u);
    }
  }
}

template <int WPS, int PROBE = 0>
int launch_reduce_mf(const float* mask, const float* prob, float* rba, int Q, int K, int64_t HW, int mode, unsigned int* counters,
                     hipStream_t st) {
  const int64_t tiles = (HW + 255) / 256;
  if (tiles > 0x3fffffffLL || Q > 2048 || K > 32) return (int)hipErrorInvalidValue;
  const int NP = (Q + 1) / 2, NPpad = (NP + MF_RING - 1) / MF_RING * MF_RING;
  const size_t dyn = (size_t)NPpad * 64 * 4;
  const int64_t chunks = (tiles + MF_CHUNK - 1) / MF_CHUNK;
  int64_t grid = 256 * WPS;
  grid = chunks < grid * 4 ? (chunks + 3) / 4 : grid;
  grid = grid < 1 ? 1 : grid;
  hipLaunchKernelGGL((rba_reduce_mf_kernel<WPS, PROBE>), dim3((unsigned)grid), dim3(256), dyn, st, mask, prob, rba, Q, K, HW, (int)tiles,
                     mode, counters);
  return rba_launch_status();
}


// ------------------------------------------------------------------------------------------------------------
// K1 with the class contraction on the matrix pipe and NO transposition ("m4", round 3): v_mfma_f32_4x4x4_16B_f16 is sixteen independent
// 4 x 4 x 4 products, block b = lane / 4.  Its B operand wants, in lane (b, j), the four k values of column j of block b -- with k = four
// consecutive query planes and "column" = the lane's OWN pixel that is exactly what a lane has after loading 16 B of each of four planes
// (the VALU kernel's load pattern: one wave instruction = 1 KiB of one plane).  The A operand (class probabilities, rows = four classes,
// k = the same four queries) is the same in every block: lane (b, i) reads row i of a small LDS table.  D lands as
// acc[class tile][pixel][r] = sem[4 tile + r][the lane's pixel]: every lane ends up with all K classes of its four pixels, the layout of the
// VALU kernel, so the epilogue (tanh-sum / sem_seg / argmax) is shared.  K accumulators per pixel as before (80 registers for K <= 20):
// occupancy stays at four workgroups per CU, which the 32 x 32 / 16 x 16 MFMA forms of rounds 1-2 could not keep (r02_k1_matrix_pipe.txt).
// Arithmetic: sigma = h + l and P = h + l with f16 pairs, l = f16(x - h) UNSCALED (both lie in [0, 1]: |l| <= 2^-12 is an f16 subnormal
// or small normal, absolute error <= 2^-25 -- the matrix pipe does not flush subnormal inputs), three products h.h + h.l + l.h into ONE
// fp32 accumulator; max |d sem| vs the fp32 fma chain ~1e-6 (tests pin the score to the oracle at the 1e-4 budget like every K1 form).
// VALU work per lane and four planes: 16 sigmoids + 24 split instructions instead of + 152 packed FMAs.
// RESULT (tools/k1_sweep.py 121,400-405, profiles/r03_k1_m4.txt): correct (max |d rba| 5.7e-6) and SLOWER, 184-194 us against 158-164: the
// 4 x 4 x 4 MFMA does not run beside the VALU work of the SIMD's other waves -- tools/micro/mfma4_overlap.hip: 4.5 ns per instruction and
// the times of matrix waves and vector waves ADD (0.57 + 0.44 -> 0.86 ms), unlike the 32 x 32 / 16 x 16 forms -- so its 60 instructions per
// four planes cost more VALU-pipe time (~540 cycles) than the 76 packed FMAs they replace (365).  Without the MFMAs the kernel takes
// 150 us, without the sigmoids 155.  Built only into librba_tune.so.
typedef _Float16 k1_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 k1_f16x2 __attribute__((ext_vector_type(2)));

// {h(a), h(b)} and {l(a), l(b)} with l = f16(x - h): v_cvt_pk_f16_f32 + two v_fma_mix (x - f32(h) is exact in fp32)
__device__ __forceinline__ void k1_split_pair(float a, float b, uint32_t& h, uint32_t& l) {
  const k1_f16x2 hh = {(_Float16)a, (_Float16)b};
  const uint32_t ap = __builtin_bit_cast(uint32_t, hh);
  uint32_t r;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(ap), "v"(-1.0f), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(ap), "v"(-1.0f), "v"(b));
  h = ap;
  l = r;
}

template <int K, bool SEM, bool ARG, int WPS, bool DYN, int ABL>
__global__ __launch_bounds__(256, WPS) void rba_reduce_m4_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                                float* __restrict__ rba, float* __restrict__ sem,
                                                                int32_t* __restrict__ argmax, int Q, int64_t HW, int tiles, int mode,
                                                                unsigned int* __restrict__ counters) {
  constexpr int MT = (K + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char k1lds[];   // [ceil(Q / 4)][MT][4] x {h01, h23, l01, l23}
  __shared__ unsigned int sh_tile;
  const int QS = (Q + 3) >> 2;
  for (int e = threadIdx.x; e < QS * MT * 4; e += 256) {
    const int i = e & 3, mt = (e >> 2) % MT, ks = (e >> 2) / MT;
    const int c = 4 * mt + i;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (4 * ks + k < Q && c < K) ? prob[(4 * ks + k) * K + c] : 0.f;
    rba_u32x4 t;
    uint32_t h, l;
    k1_split_pair(v[0], v[1], h, l); t.x = h; t.z = l;
    k1_split_pair(v[2], v[3], h, l); t.y = h; t.w = l;
    *reinterpret_cast<rba_u32x4*>(k1lds + (size_t)e * 16) = t;
  }
  __syncthreads();
  const int li = threadIdx.x & 3;
  auto next_tile = [&](int prev) -> int {
    if (!DYN) return prev < 0 ? (int)blockIdx.x : prev + (int)gridDim.x;
    __syncthreads();
    if (threadIdx.x == 0) sh_tile = atomicAdd(counters, 1u);
    __syncthreads();
    return (int)sh_tile;
  };
  for (int tile = next_tile(-1); tile < tiles; tile = next_tile(tile)) {
    const int64_t p0 = ((int64_t)tile * 256 + threadIdx.x) * 4;
    if (p0 < HW) {
      f32x4 acc4[MT][4];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc4[t][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* mp = mask + p0;
      f32x4 buf[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)(u < Q ? u : Q - 1) * HW));
      for (int ks = 0; ks < QS; ++ks) {
        f32x4 sg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (ABL == 2) { sg[u] = buf[u]; }
          else {
            const f32x2 s01 = rba_sigmoid2((f32x2){buf[u].x, buf[u].y});
            const f32x2 s23 = rba_sigmoid2((f32x2){buf[u].z, buf[u].w});
            sg[u] = (f32x4){s01.x, s01.y, s23.x, s23.y};
          }
          const int qn = 4 * ks + 4 + u < Q ? 4 * ks + 4 + u : Q - 1;   // planes beyond Q meet zero probabilities in the table
          buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)qn * HW));
        }
        k1_f16x4 bh[4], bl[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          uint32_t h0, l0, h1, l1;
          k1_split_pair(sg[0][p], sg[1][p], h0, l0);
          k1_split_pair(sg[2][p], sg[3][p], h1, l1);
          bh[p] = __builtin_bit_cast(k1_f16x4, (uint2){h0, h1});
          bl[p] = __builtin_bit_cast(k1_f16x4, (uint2){l0, l1});
        }
        if (ABL != 1) {
          const unsigned char* tp = k1lds + ((size_t)ks * MT * 4 + li) * 16;
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const rba_u32x4 a = *reinterpret_cast<const rba_u32x4*>(tp + t * 64);
            const k1_f16x4 ah = __builtin_bit_cast(k1_f16x4, (uint2){a.x, a.y}), al = __builtin_bit_cast(k1_f16x4, (uint2){a.z, a.w});
#pragma unroll
            for (int p = 0; p < 4; ++p) acc4[t][p] = __builtin_amdgcn_mfma_f32_4x4x4f16(ah, bh[p], acc4[t][p], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc4[t][p] = __builtin_amdgcn_mfma_f32_4x4x4f16(ah, bl[p], acc4[t][p], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc4[t][p] = __builtin_amdgcn_mfma_f32_4x4x4f16(al, bh[p], acc4[t][p], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p) acc4[0][p][0] += __builtin_bit_cast(float, __builtin_bit_cast(uint2, bh[p]).x ^ __builtin_bit_cast(uint2, bl[p]).y);
        }
      }
      float acc[K][4];
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[k][p] = acc4[k >> 2][p][k & 3];
      rba_epilogue<K, 4, SEM, ARG>(acc, K, mode, rba, sem, argmax, p0, HW);
    }
  }
  if (DYN && threadIdx.x == 0) {
    const unsigned int done = atomicAdd(counters + 1, 1u);
    if (done == gridDim.x - 1) {
      atomicExch(counters, 0u);
      atomicExch(counters + 1, 0u);
    }
  }
}

template <int K, int WPS, int ABL = 0>
int launch_reduce_m4(const float* mask, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int64_t HW, int mode,
                     unsigned int* counters, hipStream_t st) {
  const int64_t tiles = (HW + 1023) / 1024;
  if (tiles > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  const size_t shm = (size_t)((Q + 3) / 4) * ((K + 3) / 4) * 64;
  if (shm > 32 * 1024) return (int)hipErrorInvalidValue;
  const int64_t cap = 256 * WPS;
  int64_t grid = tiles;
  if (tiles <= cap) counters = nullptr;                             // one tile per workgroup: the static split (see launch_reduce_pk)
  if (tiles > cap) {
    const int64_t rounds = (tiles + cap - 1) / cap;
    grid = counters ? cap : (tiles + rounds - 1) / rounds;
  }
#define RBA_L(S, A, D) \
  hipLaunchKernelGGL((rba_reduce_m4_kernel<K, S, A, WPS, D, ABL>), dim3((unsigned)grid), dim3(256), shm, st, mask, prob, rba, sem, argmax, Q, HW, (int)tiles, mode, counters)
  if (counters) {
    if (sem && argmax) RBA_L(true, true, true); else if (sem) RBA_L(true, false, true); else if (argmax) RBA_L(false, true, true); else RBA_L(false, false, true);
  } else {
    if (sem && argmax) RBA_L(true, true, false); else if (sem) RBA_L(true, false, false); else if (argmax) RBA_L(false, true, false); else RBA_L(false, false, false);
  }
#undef RBA_L
  return rba_launch_status();
}


}  // namespace rba_k1
