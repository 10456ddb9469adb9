// K4 -- mask logits: out[b,q,n] = sum_c embed[b,q,c] * feat[b,c,n]  (reference: torch.einsum("bqc,bchw->bqhw"),
// mask2former_transformer_decoder.py:479).  Small-M contraction (Q = 100) against a long N = H*W/16 axis.
//
// v1 (fp32 VALU): one thread owns VEC consecutive columns n and QT query rows; feat is streamed once per
// query tile with coalesced loads, the embedding tile sits in LDS transposed ([c][q]) so that the QT values of
// one c are read with wave-uniform (broadcast) ds_read_b128.
#include <stdlib.h>
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

template <int QT, int VEC>
__global__ __launch_bounds__(256) void mask_logits_kernel(const float* __restrict__ embed, const float* __restrict__ feat,
                                                          float* __restrict__ out, int Q, int C, int64_t N) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [C][QT]
  const int b = blockIdx.z;
  const int q0 = blockIdx.y * QT;
  const int64_t n0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  const float* eb = embed + ((int64_t)b * Q) * C;
  for (int i = threadIdx.x; i < C * QT; i += blockDim.x) {
    const int c = i / QT, qq = i % QT;
    lds[i] = (q0 + qq < Q) ? eb[(int64_t)(q0 + qq) * C + c] : 0.f;
  }
  __syncthreads();
  if (n0 >= N) return;
  float acc[QT][VEC];
#pragma unroll
  for (int i = 0; i < QT; ++i)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[i][j] = 0.f;
  const float* fb = feat + (int64_t)b * C * N + n0;
#pragma unroll 2
  for (int c = 0; c < C; ++c) {
    float f[VEC];
    if (VEC == 2) {
      const float2 t = *reinterpret_cast<const float2*>(fb + (int64_t)c * N);
      f[0] = t.x; f[VEC - 1] = t.y;
    } else {
      f[0] = fb[(int64_t)c * N];
    }
    const float4* e4 = reinterpret_cast<const float4*>(lds + c * QT);
#pragma unroll
    for (int i = 0; i < QT / 4; ++i) {
      const float4 e = e4[i];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        acc[4 * i][j] = fmaf(e.x, f[j], acc[4 * i][j]);
        acc[4 * i + 1][j] = fmaf(e.y, f[j], acc[4 * i + 1][j]);
        acc[4 * i + 2][j] = fmaf(e.z, f[j], acc[4 * i + 2][j]);
        acc[4 * i + 3][j] = fmaf(e.w, f[j], acc[4 * i + 3][j]);
      }
    }
  }
  float* ob = out + ((int64_t)b * Q + q0) * N + n0;
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    if (q0 + i < Q) {
      if (VEC == 2) *reinterpret_cast<float2*>(ob + (int64_t)i * N) = make_float2(acc[i][0], acc[i][VEC - 1]);
      else ob[(int64_t)i * N] = acc[i][0];
    }
  }
}

template <int QT, int VEC>
int launch(const float* embed, const float* feat, float* out, int B, int Q, int C, int64_t N, hipStream_t st) {
  const int threads = 256;
  const int64_t per = (int64_t)threads * VEC;
  dim3 grid((unsigned)((N + per - 1) / per), (Q + QT - 1) / QT, B);
  const size_t shm = (size_t)C * QT * sizeof(float);
  if (shm > 160 * 1024) return (int)hipErrorInvalidValue;
  auto kern = mask_logits_kernel<QT, VEC>;
  if (shm > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, grid, dim3(threads), shm, st, embed, feat, out, Q, C, N);
  return rba_launch_status();
}


// ------------------------------------------------------------------------------------------------------------
// Matrix-pipe version (Q <= 112).  out[q, n] = sum_c E[q, c] F[c, n] on v_mfma_f32_16x16x4_f32 (exact fp32 fma chain):
// M = queries (7 tiles of 16, 100/112 used), N = pixel columns, K = channels.  One wave owns 64 columns: per k-step it
// loads ONE float4 per lane -- lane (kk = lane/16, j = lane%16) reads F[c0+kk][n0+4j .. 4j+3], 256 B contiguous per
// channel row -- which is the B operand of four column tiles at once (tile t = columns {n0 + 4j + t}).  The A operand
// E[q][c] is the same for every column tile and comes from an LDS copy of the whole embedding matrix laid out [c][112]
// (conflict-free ds_read_b32: 16 consecutive floats per 16-lane group, groups 112 floats = 16 banks apart).
// 7 x 4 accumulator tiles = 112 VGPRs; F loads are software-prefetched PF k-steps ahead.
typedef float f32x4_k4 __attribute__((ext_vector_type(4)));

template <int WAVES, int PF>
__global__ __launch_bounds__(64 * WAVES) void mask_logits_mfma_kernel(const float* __restrict__ embed, const float* __restrict__ feat,
                                                                      float* __restrict__ out, int Q, int C, int64_t N) {
  // LDS image [C][QP] with an ODD row stride: the fill reads E coalesced along c and writes transposed (bank = 17 c + q:
  // conflict-free), the A-operand reads (16 consecutive q per 16-lane group, groups one row apart) stay conflict-free too
  constexpr int QP = 113, QT = 7;
  extern __shared__ __attribute__((aligned(16))) float lds[];        // [C][QP]
  const int b = blockIdx.y;
  const float* eb = embed + (int64_t)b * Q * C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kk = lane >> 4;
  {
    // all of a wave's rows are requested before the first LDS write (one L2 round trip instead of one per row)
    constexpr int ROWS = (112 + WAVES - 1) / WAVES;
    for (int c4 = lane; c4 < C / 4; c4 += 64) {
      f32x4 tmp[ROWS];
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {
        const int q = wave + j * WAVES;
        tmp[j] = q < Q ? *reinterpret_cast<const f32x4*>(eb + (int64_t)q * C + 4 * c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {
        const int q = wave + j * WAVES;
        if (q < 112) {
#pragma unroll
          for (int i = 0; i < 4; ++i) lds[(4 * c4 + i) * QP + q] = tmp[j][i];
        }
      }
    }
  }
  __syncthreads();
  const int64_t n0 = ((int64_t)blockIdx.x * WAVES + wave) * 64;
  if (n0 >= N) return;
  const int64_t ncol = n0 + 4 * l15;                                  // this lane's 4 columns (N % 4 == 0)
  const bool cvalid = ncol < N;
  const float* fb = feat + (int64_t)b * C * N + (cvalid ? ncol : 0);
  f32x4_k4 acc[QT][4];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4_k4){0.f, 0.f, 0.f, 0.f};
  const int steps = C / 4;
  f32x4 fbuf[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) fbuf[u] = *reinterpret_cast<const f32x4*>(fb + (int64_t)((u < steps ? u : steps - 1) * 4 + kk) * N);
  // A operands are double-buffered in registers: the 7 LDS reads of step s+1 are issued before the 28 MFMAs of step s,
  // otherwise the compiler reads each one right before its use and waits lgkmcnt(0) four times per step.
  float a_nxt[QT];
  {
    const float* ea = lds + kk * QP + l15;
#pragma unroll
    for (int t = 0; t < QT; ++t) a_nxt[t] = ea[t * 16];
  }
  for (int s0 = 0; s0 < steps; s0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int st = s0 + u;
      if (st < steps) {
        const f32x4 f4 = fbuf[u];
        const int sn = st + PF < steps ? st + PF : steps - 1;
        fbuf[u] = *reinterpret_cast<const f32x4*>(fb + (int64_t)(sn * 4 + kk) * N);
        float a_cur[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) a_cur[t] = a_nxt[t];
        const int s1 = st + 1 < steps ? st + 1 : st;
        const float* ea = lds + (s1 * 4 + kk) * QP + l15;
#pragma unroll
        for (int t = 0; t < QT; ++t) a_nxt[t] = ea[t * 16];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t], f4[i], acc[t][i], 0, 0, 0);
      }
    }
  }
  // lane holds out[q = 16 t + 4 kk + r][n = ncol + i] in acc[t][i][r]
  if (!cvalid) return;
  float* ob = out + (int64_t)b * Q * N + ncol;
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = 16 * t + 4 * kk + r;
      if (q < Q) *reinterpret_cast<f32x4*>(ob + (int64_t)q * N) = (f32x4){acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
    }
}

template <int WAVES, int PF>
int launch_mfma(const float* embed, const float* feat, float* out, int B, int Q, int C, int64_t N, hipStream_t st) {
  const size_t shm = (size_t)C * 113 * sizeof(float);
  if (shm > 160 * 1024) return (int)hipErrorInvalidValue;
  auto kern = mask_logits_mfma_kernel<WAVES, PF>;
  static size_t shm_enabled[64];                      // raise the dynamic-LDS cap once per instantiation AND device, not per launch
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
  if (shm > 64 * 1024 && shm > shm_enabled[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return (int)e;
    shm_enabled[dev] = shm;
  }
  const int64_t per_block = 64LL * WAVES;
  dim3 grid((unsigned)((N + per_block - 1) / per_block), B);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES), shm, st, embed, feat, out, Q, C, N);
  return rba_launch_status();
}


// ------------------------------------------------------------------------------------------------------------
// f16x3 version (round 3; Q <= 112, C % 32 == 0, C <= 256).  The exact-fp32 MFMA above is bound by that instruction's rate (7.5 GFLOP padded
// at 157 TFLOP/s = 48 us of the 80 us it takes at 256 x 512 columns; HBM needs 23 us for the 186 MB).  Here every fp32 value is
// h + 2^-11 l with two f16 numbers (common.h: rba_split_f16x2, the arithmetic of split_linear_h3.h) and the product takes three
// v_mfma_f32_16x16x32_f16 -- h.h into a MAIN accumulator, h.l and l.h into a LOW one added with weight 2^-11 -- at 16x the fp32 MFMA rate.
// M = queries in 7 tiles of 16 (the A operand: the embedding matrix, split ONCE per workgroup into an LDS image in fragment order,
// [k-step of 32 channels][query tile][h | l][lane] x 16 B, read back with lane-linear ds_read_b128), N = pixel columns (the B operand: a lane
// (j = lane % 16, kb = lane / 16) loads channels 32 s + 8 kb .. + 7 of its CT consecutive columns -- 16 CT bytes per lane, 16 lanes
// contiguous per channel row -- and splits them in registers: the CT column tiles are the columns {n0 + CT j + i}).  Eight waves per workgroup
// (56 CT accumulator registers each, 256-register budget: everything stays in VGPRs), one persistent workgroup per CU (the 112 KB image is
// built once), F loads PF k-steps ahead.
typedef _Float16 k4_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t k4_u32x4 __attribute__((ext_vector_type(4)));
template <int CT> struct K4Vec;
template <> struct K4Vec<4> { using type = f32x4; };
template <> struct K4Vec<2> { using type = f32x2; };
template <> struct K4Vec<1> { using type = float; };
template <int CT>
__device__ __forceinline__ float k4_get(const typename K4Vec<CT>::type& v, int i) {
  if constexpr (CT == 1) return v; else return v[i];
}

template <int CT, int PF, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : 2) void mask_logits_h3_kernel(
    const float* __restrict__ embed, const float* __restrict__ feat, float* __restrict__ out, int Q, int C, int64_t N, int wave_tiles) {
  constexpr int QT = 7;
  using V = typename K4Vec<CT>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char k4lds[];   // [C / 32][QT][2][64] x 16 B
  const int b = blockIdx.y;
  const float* eb = embed + (int64_t)b * Q * C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kb = lane >> 4;
  const int steps = C / 32;
  // one thread = one 16-byte slot of the image: 8 consecutive channels of one query row, consecutive threads -> consecutive slots
  // (conflict-free writes; the four lane groups of a slot block read the four 32-byte pieces of 16 rows' 128-byte lines); all of a
  // thread's loads are in flight before the first is split -- the fill is a chain of L2 round trips otherwise
  constexpr int FB = WAVES == 8 ? 4 : 7;                               // slots per thread and batch: two batches at C = 256
  for (int base = 0; base < steps * QT * 64; base += 64 * WAVES * FB) {
    f32x4 x0[FB], x1[FB];
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int e = base + i * 64 * WAVES + threadIdx.x;
      const int blk = e >> 6, li = e & 63, s = blk / QT, t = blk - s * QT;
      const int q = 16 * t + (li & 15), c = 32 * s + 8 * (li >> 4);
      x0[i] = x1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (q < Q && s < steps) {
        x0[i] = *reinterpret_cast<const f32x4*>(eb + (int64_t)q * C + c);
        x1[i] = *reinterpret_cast<const f32x4*>(eb + (int64_t)q * C + c + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int e = base + i * 64 * WAVES + threadIdx.x;
      k4_u32x4 h, l;
      uint32_t hh, ll;
      rba_split_f16x2(x0[i].x, x0[i].y, hh, ll); h.x = hh; l.x = ll;
      rba_split_f16x2(x0[i].z, x0[i].w, hh, ll); h.y = hh; l.y = ll;
      rba_split_f16x2(x1[i].x, x1[i].y, hh, ll); h.z = hh; l.z = ll;
      rba_split_f16x2(x1[i].z, x1[i].w, hh, ll); h.w = hh; l.w = ll;
      if (e < steps * QT * 64) {
        unsigned char* dst = k4lds + (size_t)(e >> 6) * 2048 + (e & 63) * 16;
        *reinterpret_cast<k4_u32x4*>(dst) = h;
        *reinterpret_cast<k4_u32x4*>(dst + 1024) = l;
      }
    }
  }
  __syncthreads();
  for (int tile = blockIdx.x * WAVES + wave; tile < wave_tiles; tile += gridDim.x * WAVES) {
    const int64_t ncol = (int64_t)tile * (16 * CT) + CT * j;          // this lane's CT columns (N % CT == 0)
    const bool cvalid = ncol < N;
    const float* fb = feat + (int64_t)b * C * N + (cvalid ? ncol : 0) + (int64_t)(8 * kb) * N;
    f32x4 accm[QT][CT], accl[QT][CT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int i = 0; i < CT; ++i) accm[t][i] = accl[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    V fbuf[PF][8];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int r = 0; r < 8; ++r) fbuf[u][r] = *reinterpret_cast<const V*>(fb + (int64_t)((u < steps ? u : steps - 1) * 32 + r) * N);
    for (int s0 = 0; s0 < steps; s0 += PF) {                            // steps % PF == 0 (launch_h3)
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int s = s0 + u;
        {
          k4_f16x8 bh[CT], bl[CT];
#pragma unroll
          for (int i = 0; i < CT; ++i) {
            k4_u32x4 h, l;
            uint32_t hh, ll;
            rba_split_f16x2(k4_get<CT>(fbuf[u][0], i), k4_get<CT>(fbuf[u][1], i), hh, ll); h.x = hh; l.x = ll;
            rba_split_f16x2(k4_get<CT>(fbuf[u][2], i), k4_get<CT>(fbuf[u][3], i), hh, ll); h.y = hh; l.y = ll;
            rba_split_f16x2(k4_get<CT>(fbuf[u][4], i), k4_get<CT>(fbuf[u][5], i), hh, ll); h.z = hh; l.z = ll;
            rba_split_f16x2(k4_get<CT>(fbuf[u][6], i), k4_get<CT>(fbuf[u][7], i), hh, ll); h.w = hh; l.w = ll;
            bh[i] = __builtin_bit_cast(k4_f16x8, h);
            bl[i] = __builtin_bit_cast(k4_f16x8, l);
          }
          const int sn = s + PF < steps ? s + PF : steps - 1;
#pragma unroll
          for (int r = 0; r < 8; ++r) fbuf[u][r] = *reinterpret_cast<const V*>(fb + (int64_t)(sn * 32 + r) * N);
          const unsigned char* ea = k4lds + (size_t)s * QT * 2048 + lane * 16;
#pragma unroll
          for (int t = 0; t < QT; ++t) {
            const k4_f16x8 ah = __builtin_bit_cast(k4_f16x8, *reinterpret_cast<const k4_u32x4*>(ea + t * 2048));
            const k4_f16x8 al = __builtin_bit_cast(k4_f16x8, *reinterpret_cast<const k4_u32x4*>(ea + t * 2048 + 1024));
            // consecutive MFMAs write different accumulators; per accumulator the order is h.h | h.l, l.h
#pragma unroll
            for (int i = 0; i < CT; ++i) accm[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[i], accm[t][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < CT; ++i) accl[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[i], accl[t][i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < CT; ++i) accl[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[i], accl[t][i], 0, 0, 0);
          }
        }
      }
    }
    // lane holds out[q = 16 t + 4 kb + r][n = ncol + i] in acc[t][i][r]
    if (cvalid) {
      float* ob = out + (int64_t)b * Q * N + ncol;
      asm volatile("" : "+v"(ob));                                      // keeps the 28 row addresses out of the k loop's registers
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * t + 4 * kb + r;
          if (q < Q) {
            V o;
            if constexpr (CT == 1) o = fmaf(accl[t][0][r], 0.00048828125f, accm[t][0][r]);
            else {
#pragma unroll
              for (int i = 0; i < CT; ++i) o[i] = fmaf(accl[t][i][r], 0.00048828125f, accm[t][i][r]);
            }
            *reinterpret_cast<V*>(ob + (int64_t)q * N) = o;
          }
        }
    }
  }
}

template <int CT, int PF, int WAVES>
int launch_h3(const float* embed, const float* feat, float* out, int B, int Q, int C, int64_t N, hipStream_t st) {
  const size_t shm = (size_t)(C / 32) * 7 * 2048;
  auto kern = mask_logits_h3_kernel<CT, PF, WAVES>;
  static size_t shm_enabled[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
  if (shm > 64 * 1024 && shm > shm_enabled[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return (int)e;
    shm_enabled[dev] = shm;
  }
  const int64_t wave_tiles = (N + 16 * CT - 1) / (16 * CT);
  if (wave_tiles > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  int64_t wgs = (wave_tiles + WAVES - 1) / WAVES;
  const int64_t cap = B >= 256 ? 1 : 256 / B;                       // one workgroup per CU (LDS): persistent beyond that
  if (wgs > cap) { const int64_t rounds = (wgs + cap - 1) / cap; wgs = (wgs + rounds - 1) / rounds; }
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs, B), dim3(64 * WAVES), shm, st, embed, feat, out, Q, C, N, (int)wave_tiles);
  return rba_launch_status();
}

}  // namespace

extern "C" int rba_mask_logits_f32(const float* embed, const float* feat, float* out, int B, int Q, int C, int64_t N,
                                   void* stream) {
  RBA_CHECK_ARG(B >= 0 && Q >= 0 && C >= 1 && N >= 0 && B <= 65535);
  if (B == 0 || Q == 0 || N == 0) return 0;
  RBA_CHECK_ARG(embed && feat && out);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  if (Q <= 112 && C % 4 == 0 && C <= 360 && N % 4 == 0 && ((((uintptr_t)feat | (uintptr_t)out | (uintptr_t)embed) & 15) == 0)) {
    // 8 waves (512 columns) per block when that still gives every CU a block, else 4 waves
    if ((N + 511) / 512 * B >= 256) return launch_mfma<8, 2>(embed, feat, out, B, Q, C, N, st);
    return launch_mfma<4, 4>(embed, feat, out, B, Q, C, N, st);
  }
  const bool vec2 = (N % 2 == 0) && ((((uintptr_t)feat | (uintptr_t)out) & 7) == 0);
  if (vec2) return launch<52, 2>(embed, feat, out, B, Q, C, N, st);
  return launch<52, 1>(embed, feat, out, B, Q, C, N, st);
}

// tools / tests only: 0 = columns per lane chosen from N, 1 / 2 = forced; waves per workgroup 8 (product) or 4
RBA_KNOB(rba_k4_variant, 0);
RBA_KNOB(rba_k4_waves, 8);

// The same contraction in f16x3 arithmetic (domain |x| < 65504 like every f16x3 entry point; beyond it the result is NaN, never a wrong number).
// Shapes outside Q <= 112, C % 32 == 0, C <= 256 take the exact-fp32 path above.
extern "C" int rba_mask_logits_f16x3_f32(const float* embed, const float* feat, float* out, int B, int Q, int C, int64_t N,
                                         void* stream) {
  RBA_CHECK_ARG(B >= 0 && Q >= 0 && C >= 1 && N >= 0 && B <= 65535);
  if (B == 0 || Q == 0 || N == 0) return 0;
  RBA_CHECK_ARG(embed && feat && out);
  const bool a16 = ((((uintptr_t)feat | (uintptr_t)out | (uintptr_t)embed) & 15) == 0);
  if (!(Q <= 112 && C % 32 == 0 && C <= 256 && a16)) return rba_mask_logits_f32(embed, feat, out, B, Q, C, N, stream);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  // two columns per lane (8-byte loads) once that still leaves every SIMD of the chip a wave tile; measured (tools/k4_sweep.py, 256 x 512 / 180 x 320
  // / 4 x 32 x 64 columns): 66 -> 41 us, 43 -> 22 us, 37 -> 11 us against the exact-fp32 kernel.  Four columns per lane need 224 accumulator
  // registers, which the compiler splits between VGPRs and AGPRs with two copies per MFMA (64 us): not built.
  int ct = (N % 2 == 0 && N * B >= 32768) ? 2 : 1;
  if (rba_k4_variant == 1) ct = 1;
  if (rba_k4_variant == 2 && N % 2 == 0) ct = 2;
  const bool even = (C / 32) % 2 == 0;                               // prefetch depth 2 needs an even number of 32-channel steps
#define RBA_K4(CTV, W) (even ? launch_h3<CTV, 2, W>(embed, feat, out, B, Q, C, N, st) : launch_h3<CTV, 1, W>(embed, feat, out, B, Q, C, N, st))
  const bool w8 = rba_k4_waves == 8;
  if (ct == 2) return w8 ? RBA_K4(2, 8) : RBA_K4(2, 4);
  return w8 ? RBA_K4(1, 8) : RBA_K4(1, 4);
#undef RBA_K4
}
