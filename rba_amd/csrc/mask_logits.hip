// K4 -- mask logits: out[b,q,n] = sum_c embed[b,q,c] * feat[b,c,n]  (reference: torch.einsum("bqc,bchw->bqhw"),
// mask2former_transformer_decoder.py:479).  Small-M contraction (Q = 100) against a long N = H*W/16 axis.
//
// v1 (fp32 VALU): one thread owns VEC consecutive columns n and QT query rows; feat is streamed once per
// query tile with coalesced loads, the embedding tile sits in LDS transposed ([c][q]) so that the QT values of
// one c are read with wave-uniform (broadcast) ds_read_b128.
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

}  // namespace

extern "C" int rba_mask_logits_f32(const float* embed, const float* feat, float* out, int B, int Q, int C, int64_t N,
                                   void* stream) {
  RBA_CHECK_ARG(B >= 0 && Q >= 0 && C >= 1 && N >= 0 && B <= 65535);
  if (B == 0 || Q == 0 || N == 0) return 0;
  RBA_CHECK_ARG(embed && feat && out);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const bool vec2 = (N % 2 == 0) && ((((uintptr_t)feat | (uintptr_t)out) & 7) == 0);
  if (vec2) return launch<52, 2>(embed, feat, out, B, Q, C, N, st);
  return launch<52, 1>(embed, feat, out, B, Q, C, N, st);
}
