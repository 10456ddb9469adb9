// K2 -- multi-scale deformable attention forward (replaces the reference's only native op,
// pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 behind ms_deform_attn_forward, vision.cpp:19).
//
// Gather-bound.  Layout choice for wave64: the D channels of one (query, head) are contiguous in `value`
// ([N,S,M,D]), so a group of D/4 lanes reads one 16 B-per-lane, 128 B-contiguous (D=32) segment per bilinear
// tap and the wave covers 64/(D/4) = 8 (query, head) pairs; consecutive heads of a query are adjacent in both
// sampling_loc and out, so loads of the sampling parameters and the output stores are contiguous too.
// `value` of a level fits in L2/Infinity Cache (<= 20 MB at C5), so taps are cache hits after first touch.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

template <int VEC>
struct Ld;
template <>
struct Ld<4> {
  static __device__ __forceinline__ void acc(float (&a)[4], const float* p, float wgt) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    a[0] = fmaf(wgt, v.x, a[0]); a[1] = fmaf(wgt, v.y, a[1]); a[2] = fmaf(wgt, v.z, a[2]); a[3] = fmaf(wgt, v.w, a[3]);
  }
};
template <>
struct Ld<1> {
  static __device__ __forceinline__ void acc(float (&a)[1], const float* p, float wgt) { a[0] = fmaf(wgt, *p, a[0]); }
};

template <int VEC>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                                       const float* __restrict__ attw, float* __restrict__ out,
                                                       int S, int M, int D, int L, int Lq, int P, int64_t total) {
  const int dv = D / VEC;                         // lanes per (n,q,m)
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % dv) * VEC;
  const int64_t nqm = idx / dv;                   // (n*Lq + q)*M + m
  const int m = (int)(nqm % M);
  const int64_t n = nqm / ((int64_t)M * Lq);
  const float* vbase = value + (n * S * M + m) * (int64_t)D + c;   // + s*M*D per spatial position
  const float* lp = loc + nqm * L * P * 2;
  const float* wp = attw + nqm * L * P;
  const int64_t vstride = (int64_t)M * D;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float* vl = vbase + lsi[l] * vstride;
    for (int p = 0; p < P; ++p) {
      const float x = lp[(l * P + p) * 2], y = lp[(l * P + p) * 2 + 1];
      const float wgt = wp[l * P + p];
      const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h0 = (int)hf, w0 = (int)wf;
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        float s[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] = 0.f;
        const bool h0ok = h0 >= 0, h1ok = h0 + 1 <= H - 1, w0ok = w0 >= 0, w1ok = w0 + 1 <= W - 1;
        const float* p00 = vl + ((int64_t)h0 * W + w0) * vstride;
        if (h0ok && w0ok) Ld<VEC>::acc(s, p00, hh * hw);
        if (h0ok && w1ok) Ld<VEC>::acc(s, p00 + vstride, hh * lw);
        if (h1ok && w0ok) Ld<VEC>::acc(s, p00 + (int64_t)W * vstride, lh * hw);
        if (h1ok && w1ok) Ld<VEC>::acc(s, p00 + (int64_t)(W + 1) * vstride, lh * lw);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(wgt, s[i], acc[i]);
      }
    }
  }
  float* o = out + nqm * D + c;
  if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[VEC > 1 ? 1 : 0], acc[VEC > 2 ? 2 : 0], acc[VEC > 3 ? 3 : 0]);
  else o[0] = acc[0];
}

// Sampling parameters of MSDeformAttn.forward in one pass (ms_deform_attn.py:95-115): raw = the output of the sampling_offsets and
// attention_weights Linears evaluated as ONE Linear, [rows, M L P 2 | M L P]; loc = reference_points + offsets / (W_l, H_l);
// attw = softmax over the L P logits of a head.  One thread per (row, head).  Replaces two GEMM launches, the softmax and six small
// elementwise / copy kernels per encoder layer.
__global__ __launch_bounds__(256) void msda_prepare_kernel(const float* __restrict__ raw, const float* __restrict__ ref,
                                                           const int64_t* __restrict__ shapes, float* __restrict__ loc,
                                                           float* __restrict__ attw, int64_t rows, int M, int L, int P) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * M) return;
  const int64_t row = idx / M;
  const int m = (int)(idx - row * M);
  const int LP = L * P, stride = 3 * M * LP;
  const float* off = raw + row * stride + (int64_t)m * LP * 2;
  const float* lg = raw + row * stride + (int64_t)M * LP * 2 + (int64_t)m * LP;
  float mx = -INFINITY;
  for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lg[i]);
  float sum = 0.f;
  for (int i = 0; i < LP; ++i) sum += expf(lg[i] - mx);
  float* lo = loc + idx * LP * 2;
  float* wo = attw + idx * LP;
  for (int l = 0; l < L; ++l) {
    const float Wl = (float)shapes[2 * l + 1], Hl = (float)shapes[2 * l];
    const float rx = ref[(row * L + l) * 2], ry = ref[(row * L + l) * 2 + 1];
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      lo[2 * i] = rx + off[2 * i] / Wl;
      lo[2 * i + 1] = ry + off[2 * i + 1] / Hl;
      wo[i] = expf(lg[i] - mx) / sum;
    }
  }
}

}  // namespace

extern "C" int rba_msda_prepare_f32(const float* raw, const float* reference_points, const int64_t* spatial_shapes, float* loc, float* attw,
                                    int64_t rows, int M, int L, int P, void* stream) {
  RBA_CHECK_ARG(rows >= 0 && M >= 1 && L >= 1 && P >= 1);
  if (rows == 0) return 0;
  RBA_CHECK_ARG(raw && reference_points && spatial_shapes && loc && attw && rows * M < (int64_t)1 << 40);
  rba_begin();
  const int64_t total = rows * M;
  hipLaunchKernelGGL(msda_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw, reference_points,
                     spatial_shapes, loc, attw, rows, M, L, P);
  return rba_launch_status();
}

extern "C" int rba_ms_deform_attn_fwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                          const float* sampling_loc, const float* attn_weight, float* out,
                                          int N, int S, int M, int D, int L, int Lq, int P, void* stream) {
  RBA_CHECK_ARG(N >= 0 && S >= 1 && M >= 1 && D >= 1 && L >= 1 && Lq >= 0 && P >= 1);
  if (N == 0 || Lq == 0) return 0;
  RBA_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const bool vec4 = (D % 4 == 0) && ((((uintptr_t)value | (uintptr_t)out) & 15) == 0);
  const int vec = vec4 ? 4 : 1;
  const int64_t total = (int64_t)N * Lq * M * (D / vec);
  const int threads = 256;
  const int64_t blocks = (total + threads - 1) / threads;
  RBA_CHECK_ARG(blocks <= 0x7fffffffLL);
  if (vec4)
    hipLaunchKernelGGL((msda_fwd_kernel<4>), dim3((unsigned)blocks), dim3(threads), 0, st, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, out, S, M, D, L, Lq, P, total);
  else
    hipLaunchKernelGGL((msda_fwd_kernel<1>), dim3((unsigned)blocks), dim3(threads), 0, st, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, out, S, M, D, L, Lq, P, total);
  return rba_launch_status();
}
