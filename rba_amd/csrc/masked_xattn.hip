// K3 -- masked multi-head cross attention core of the transformer decoder (reference:
// mask2former_transformer_decoder.py:106-118 nn.MultiheadAttention with bool attn_mask, the all-masked-row
// fix :433, and the threshold sigmoid(mask) < 0.5 of :483-487 which is fused here).
//
// 100 queries x 8 heads against up to ~15k keys, head_dim 32: tiny FLOPs, latency/launch bound.  One
// 256-thread workgroup per (batch, head, query); thread t walks keys t, t+256, ... with a private online
// softmax state (m, l, acc[32]); blocked keys cost one mask load and nothing else; the 256 states are merged
// by wave shuffles and a 4-entry LDS exchange.  K/V rows are 128 B contiguous per (key, head).
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

constexpr int HD = 32;

__device__ __forceinline__ void merge_state(float& m, float& l, float (&acc)[HD], float m2, float l2, const float (&acc2)[HD]) {
  const float mn = fmaxf(m, m2);
  const float a = (m == -INFINITY) ? 0.f : __expf(m - mn);
  const float b = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
  l = l * a + l2 * b;
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = acc[d] * a + acc2[d] * b;
  m = mn;
}

__global__ __launch_bounds__(256) void masked_xattn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ mlog,
                                                           float* __restrict__ out, int Q, int S, int nH) {
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float sh_any[4];
  __shared__ float sh_m[4], sh_l[4], sh_acc[4][HD];

  const float* mrow = mlog ? mlog + ((int64_t)b * Q + qi) * S : nullptr;
  // does the row have any un-blocked key?  (blocked iff sigmoid(x) < 0.5)
  bool use_mask = false;
  if (mrow) {
    float any = 0.f;
    for (int s = tid; s < S; s += 256) any = fmaxf(any, rba_sigmoid(mrow[s]) < 0.5f ? 0.f : 1.f);
    any = wave_reduce_max(any);
    if (lane == 0) sh_any[wave] = any;
    __syncthreads();
    use_mask = fmaxf(fmaxf(sh_any[0], sh_any[1]), fmaxf(sh_any[2], sh_any[3])) > 0.f;
  }

  const float scale = 0.17677669529663687f;  // 32^-0.5
  float qv[HD];
  const float* qp = q + (((int64_t)b * Q + qi) * nH + h) * HD;   // wave-uniform -> scalar loads
#pragma unroll
  for (int d = 0; d < HD; ++d) qv[d] = qp[d] * scale;

  float m = -INFINITY, l = 0.f, acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;

  const int64_t rs = (int64_t)nH * HD;
  const float* kb = k + (int64_t)b * S * rs + h * HD;
  const float* vb = v + (int64_t)b * S * rs + h * HD;
  for (int s = tid; s < S; s += 256) {
    if (use_mask && rba_sigmoid(mrow[s]) < 0.5f) continue;
    const float4* kr = reinterpret_cast<const float4*>(kb + s * rs);
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
      const float4 kk = kr[i];
      sc = fmaf(qv[4 * i], kk.x, sc); sc = fmaf(qv[4 * i + 1], kk.y, sc);
      sc = fmaf(qv[4 * i + 2], kk.z, sc); sc = fmaf(qv[4 * i + 3], kk.w, sc);
    }
    if (sc > m) {
      const float a = __expf(m - sc);   // m = -inf -> 0
      l *= a;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] *= a;
      m = sc;
    }
    const float p = __expf(sc - m);
    l += p;
    const float4* vr = reinterpret_cast<const float4*>(vb + s * rs);
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
      const float4 vv = vr[i];
      acc[4 * i] = fmaf(p, vv.x, acc[4 * i]); acc[4 * i + 1] = fmaf(p, vv.y, acc[4 * i + 1]);
      acc[4 * i + 2] = fmaf(p, vv.z, acc[4 * i + 2]); acc[4 * i + 3] = fmaf(p, vv.w, acc[4 * i + 3]);
    }
  }
  // merge across the wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, RBA_WAVE), l2 = __shfl_xor(l, o, RBA_WAVE);
    float acc2[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc2[d] = __shfl_xor(acc[d], o, RBA_WAVE);
    merge_state(m, l, acc, m2, l2, acc2);
  }
  if (lane == 0) {
    sh_m[wave] = m; sh_l[wave] = l;
#pragma unroll
    for (int d = 0; d < HD; ++d) sh_acc[wave][d] = acc[d];
  }
  __syncthreads();
  if (tid < HD) {
    float M = fmaxf(fmaxf(sh_m[0], sh_m[1]), fmaxf(sh_m[2], sh_m[3]));
    float L = 0.f, A = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = sh_m[w] == -INFINITY ? 0.f : __expf(sh_m[w] - M);
      L += sh_l[w] * f;
      A += sh_acc[w][tid] * f;
    }
    out[(((int64_t)b * Q + qi) * nH + h) * HD + tid] = A / L;
  }
}

}  // namespace

extern "C" int rba_masked_xattn_f32(const float* q, const float* k, const float* v, const float* mask_logits, float* out,
                                    int B, int Q, int S, int nH, int hd, void* stream) {
  RBA_CHECK_ARG(B >= 0 && Q >= 0 && S >= 1 && nH >= 1 && hd == HD && nH <= 65535 && B <= 65535);
  if (B == 0 || Q == 0) return 0;
  RBA_CHECK_ARG(q && k && v && out && (((uintptr_t)k | (uintptr_t)v) & 15) == 0);
  rba_begin();
  hipLaunchKernelGGL(masked_xattn_kernel, dim3(Q, nH, B), dim3(256), 0, (hipStream_t)stream, q, k, v, mask_logits, out, Q, S, nH);
  return rba_launch_status();
}
