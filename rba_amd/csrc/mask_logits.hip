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
