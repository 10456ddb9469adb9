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

// tools / tests only (not part of the ABI contract): 1 = rba_ms_deform_attn_fwd_f32 always runs the generic kernel
RBA_KNOB(rba_k2_variant, 0);

namespace {

// T = float (VEC 4 or 1) or double (VEC 1): the reference's native op dispatches both (ops/src/cuda/ms_deform_attn_cuda.cu:69, AT_DISPATCH_FLOATING_TYPES)
template <typename T, int VEC>
struct Ld;
template <>
struct Ld<float, 4> {
  static __device__ __forceinline__ void acc(float (&a)[4], const float* p, float wgt) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    a[0] = fmaf(wgt, v.x, a[0]); a[1] = fmaf(wgt, v.y, a[1]); a[2] = fmaf(wgt, v.z, a[2]); a[3] = fmaf(wgt, v.w, a[3]);
  }
};
template <typename T>
struct Ld<T, 1> {
  static __device__ __forceinline__ void acc(T (&a)[1], const T* p, T wgt) { a[0] = fma(wgt, *p, a[0]); }
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lsi, const T* __restrict__ loc,
                                                       const T* __restrict__ attw, T* __restrict__ out,
                                                       int S, int M, int D, int L, int Lq, int P, int64_t total) {
  const int dv = D / VEC;                         // lanes per (n,q,m)
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % dv) * VEC;
  const int64_t nqm = idx / dv;                   // (n*Lq + q)*M + m
  const int m = (int)(nqm % M);
  const int64_t n = nqm / ((int64_t)M * Lq);
  const T* vbase = value + (n * S * M + m) * (int64_t)D + c;   // + s*M*D per spatial position
  const T* lp = loc + nqm * L * P * 2;
  const T* wp = attw + nqm * L * P;
  const int64_t vstride = (int64_t)M * D;
  T acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = (T)0;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const T* vl = vbase + lsi[l] * vstride;
    for (int p = 0; p < P; ++p) {
      const T x = lp[(l * P + p) * 2], y = lp[(l * P + p) * 2 + 1];
      const T wgt = wp[l * P + p];
      const T h_im = y * H - (T)0.5, w_im = x * W - (T)0.5;
      if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {
        const T hf = floor(h_im), wf = floor(w_im);
        const int h0 = (int)hf, w0 = (int)wf;
        const T lh = h_im - hf, lw = w_im - wf, hh = (T)1 - lh, hw = (T)1 - lw;
        T s[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] = (T)0;
        const bool h0ok = h0 >= 0, h1ok = h0 + 1 <= H - 1, w0ok = w0 >= 0, w1ok = w0 + 1 <= W - 1;
        const T* p00 = vl + ((int64_t)h0 * W + w0) * vstride;
        if (h0ok && w0ok) Ld<T, VEC>::acc(s, p00, hh * hw);
        if (h0ok && w1ok) Ld<T, VEC>::acc(s, p00 + vstride, hh * lw);
        if (h1ok && w0ok) Ld<T, VEC>::acc(s, p00 + (int64_t)W * vstride, lh * hw);
        if (h1ok && w1ok) Ld<T, VEC>::acc(s, p00 + (int64_t)(W + 1) * vstride, lh * lw);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fma(wgt, s[i], acc[i]);
      }
    }
  }
  T* o = out + nqm * D + c;
#pragma unroll
  for (int i = 0; i < VEC; ++i) o[i] = acc[i];                // (VEC = 4: four consecutive floats of a 16-byte-aligned address -> one 16-byte store)
}

// ---- round 3: the form the model runs (head_dim 32, P = 4, L = 1 or 3).  What the counters said about the kernel above at C5
// (19 320 queries x 3 levels, profiles/r03_k2_c5_pmc.txt): its 48 taps per (query, head) are 927 MB of 128-byte gathers per launch
// against 62 MB of algorithmic traffic, issued from a rolled loop one tap after the other.  Here
//   * a workgroup is 32 CONSECUTIVE queries of ONE head (8 lanes x 16 B = the head's 32 channels): neighbouring queries sample
//     overlapping neighbourhoods of the same channel slice, so most taps of a workgroup hit the lines its neighbours just pulled
//     into the vector L1 (the old mapping put 4 queries x 8 heads in a workgroup: 8 disjoint channel slices, no reuse);
//   * L and P are compile-time: the sampling parameters of a (query, head) are fetched with nine 16-byte loads, and all 16 taps of a
//     level are requested (branch-free: an out-of-image tap reads a clamped address with weight zero) before the first is consumed;
//   * FUSED: the sampling locations and the softmax over the L P logits (ms_deform_attn.py:95-115) are computed here from the raw
//     output of the sampling Linear -- rba_msda_prepare_f32 and its 23 MB round trip disappear (same expressions: bit-identical).
// Arithmetic per output exactly as msda_fwd_kernel (same fma order; a zero-weight tap adds an exact zero).
template <int L, int P, bool FUSED>
__global__ __launch_bounds__(256) void msda_fwd_lp_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                          const int64_t* __restrict__ lsi, const float* __restrict__ loc_or_raw,
                                                          const float* __restrict__ attw_or_ref, float* __restrict__ out, int S, int M,
                                                          int Lq) {
  constexpr int D = 32, LP = L * P, NSMP = (LP + 7) / 8;
  // per (query, head) group and sample: four tap offsets (bytes from the head's channel slice of position 0), four tap weights, the
  // attention weight.  Computed ONCE per sample by one lane of the group (sample j and j + 8 by lane j) and shared through LDS: the
  // counters of the first round-3 version showed 1 265 vector instructions per wave, most of them the SAME sample geometry evaluated
  // by all eight lanes of a group (profiles/r03_k2_c5_pmc.txt)
  __shared__ __attribute__((aligned(16))) uint32_t sh_off[32][LP][4];
  __shared__ __attribute__((aligned(16))) float sh_tw[32][LP][4];
  __shared__ float sh_wq[32][LP + 1];
  const int g = threadIdx.x >> 3, j = threadIdx.x & 7, c = j * 4;
  const int q = blockIdx.x * 32 + g, m = blockIdx.y, n = blockIdx.z;
  if (q >= Lq) return;
  const int64_t nq = (int64_t)n * Lq + q, nqm = nq * M + m;
  const uint32_t vstride_b = (uint32_t)M * D * 4;                                   // bytes between spatial positions

  // ---- phase A: this lane's samples
  float z[NSMP], sx[NSMP], sy[NSMP], wq[NSMP];
  if (FUSED) {
    const int stride = 3 * M * LP;
    const float* off = loc_or_raw + nq * stride + (int64_t)m * LP * 2;
    const float* lg = loc_or_raw + nq * stride + (int64_t)M * LP * 2 + (int64_t)m * LP;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NSMP; ++t) {
      const int i = j + 8 * t;
      const bool ok = i < LP;
      z[t] = ok ? lg[i] : -INFINITY;
      const float2 o = ok ? *reinterpret_cast<const float2*>(off + 2 * i) : make_float2(0.f, 0.f);
      sx[t] = o.x;
      sy[t] = o.y;
      mx = fmaxf(mx, z[t]);
    }
    // max over the group; the sum adds the L P terms in index order 0, 1, 2, ... exactly as msda_prepare_kernel's loop does
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 8));
    float e[NSMP];
#pragma unroll
    for (int t = 0; t < NSMP; ++t) e[t] = (j + 8 * t) < LP ? expf(z[t] - mx) : 0.f;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) sum += __shfl(e[i >> 3], i & 7, 8);
#pragma unroll
    for (int t = 0; t < NSMP; ++t) {
      const int i = min(j + 8 * t, LP - 1), l = i / P;
      const float Wl = (float)shapes[2 * l + 1], Hl = (float)shapes[2 * l];
      const float rx = attw_or_ref[(nq * L + l) * 2], ry = attw_or_ref[(nq * L + l) * 2 + 1];
      sx[t] = rx + sx[t] / Wl;
      sy[t] = ry + sy[t] / Hl;
      wq[t] = e[t] / sum;
    }
  } else {
    const float* lp = loc_or_raw + nqm * LP * 2;
    const float* wp = attw_or_ref + nqm * LP;
#pragma unroll
    for (int t = 0; t < NSMP; ++t) {
      const int i = min(j + 8 * t, LP - 1);
      const float2 o = *reinterpret_cast<const float2*>(lp + 2 * i);
      sx[t] = o.x;
      sy[t] = o.y;
      wq[t] = wp[i];
    }
  }
#pragma unroll
  for (int t = 0; t < NSMP; ++t) {
    const int i = j + 8 * t;
    if (i < LP) {
      const int l = i / P;
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const float h_im = sy[t] * H - 0.5f, w_im = sx[t] * W - 0.5f;
      const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = (int)hf, w0 = (int)wf;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool h0ok = inside && h0 >= 0, h1ok = inside && h0 + 1 <= H - 1, w0ok = w0 >= 0, w1ok = w0 + 1 <= W - 1;
      const int hc0 = min(max(h0, 0), H - 1), hc1 = min(max(h0 + 1, 0), H - 1), wc0 = min(max(w0, 0), W - 1), wc1 = min(max(w0 + 1, 0), W - 1);
      const uint32_t base = (uint32_t)lsi[l];
      // an out-of-image tap reads a clamped (valid) address with weight zero: it adds an exact zero, which is what skipping it adds
      const rba_u32x4 o4 = {(base + (uint32_t)(hc0 * W + wc0)) * vstride_b, (base + (uint32_t)(hc0 * W + wc1)) * vstride_b,
                            (base + (uint32_t)(hc1 * W + wc0)) * vstride_b, (base + (uint32_t)(hc1 * W + wc1)) * vstride_b};
      const f32x4 t4 = {(h0ok && w0ok) ? hh * hw : 0.f, (h0ok && w1ok) ? hh * lw : 0.f, (h1ok && w0ok) ? lh * hw : 0.f,
                        (h1ok && w1ok) ? lh * lw : 0.f};
      *reinterpret_cast<rba_u32x4*>(&sh_off[g][i][0]) = o4;
      *reinterpret_cast<f32x4*>(&sh_tw[g][i][0]) = t4;
      sh_wq[g][i] = wq[t];
    }
  }
  __builtin_amdgcn_wave_barrier();                                                  // a group lives inside one wave; LDS operations of a wave are in order

  // ---- phase B: all 4 P taps of a level in flight, then the fma chains in the generic kernel's order
  const char* vb = reinterpret_cast<const char*>(value + ((int64_t)n * S * M + m) * D + c);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < L; ++l) {
    f32x4 tap[P][4], tw[P];
    float wl[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const rba_u32x4 o4 = *reinterpret_cast<const rba_u32x4*>(&sh_off[g][l * P + p][0]);
      tw[p] = *reinterpret_cast<const f32x4*>(&sh_tw[g][l * P + p][0]);
      wl[p] = sh_wq[g][l * P + p];
#pragma unroll
      for (int t = 0; t < 4; ++t) tap[p][t] = *reinterpret_cast<const f32x4*>(vb + o4[t]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s4[0] = fmaf(tw[p][t], tap[p][t].x, s4[0]);
        s4[1] = fmaf(tw[p][t], tap[p][t].y, s4[1]);
        s4[2] = fmaf(tw[p][t], tap[p][t].z, s4[2]);
        s4[3] = fmaf(tw[p][t], tap[p][t].w, s4[3]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(wl[p], s4[i], acc[i]);
    }
  }
  *reinterpret_cast<float4*>(out + nqm * D + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

template <bool FUSED>
static int launch_msda_lp(const float* value, const int64_t* shapes, const int64_t* lsi, const float* a, const float* b, float* out, int N, int S,
                          int M, int L, int Lq, int P, hipStream_t st) {
  if (P != 4 || (L != 1 && L != 3) || M > 65535 || N > 65535) return -1;
  if ((int64_t)S * M * 128 >= ((int64_t)1 << 32)) return -1;     // the kernel's tap addresses are 32-bit byte offsets into one image's value [S, M, 32] fp32
  const dim3 grid((unsigned)((Lq + 31) / 32), (unsigned)M, (unsigned)N);
  if (L == 1) hipLaunchKernelGGL((msda_fwd_lp_kernel<1, 4, FUSED>), grid, dim3(256), 0, st, value, shapes, lsi, a, b, out, S, M, Lq);
  else hipLaunchKernelGGL((msda_fwd_lp_kernel<3, 4, FUSED>), grid, dim3(256), 0, st, value, shapes, lsi, a, b, out, S, M, Lq);
  return 0;
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
  if (rba_k2_variant == 0 && vec4 && D == 32 && ((((uintptr_t)sampling_loc | (uintptr_t)attn_weight) & 15) == 0) &&
      launch_msda_lp<false>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, N, S, M, L, Lq, P, st) == 0)
    return rba_launch_status();
  const int vec = vec4 ? 4 : 1;
  const int64_t total = (int64_t)N * Lq * M * (D / vec);
  const int threads = 256;
  const int64_t blocks = (total + threads - 1) / threads;
  RBA_CHECK_ARG(blocks <= 0x7fffffffLL);
  if (vec4)
    hipLaunchKernelGGL((msda_fwd_kernel<float, 4>), dim3((unsigned)blocks), dim3(threads), 0, st, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, out, S, M, D, L, Lq, P, total);
  else
    hipLaunchKernelGGL((msda_fwd_kernel<float, 1>), dim3((unsigned)blocks), dim3(threads), 0, st, value, spatial_shapes,
                       level_start_index, sampling_loc, attn_weight, out, S, M, D, L, Lq, P, total);
  return rba_launch_status();
}

// The double-precision entry of the same op: the reference FFI dispatches float AND double (ops/src/cuda/ms_deform_attn_cuda.cu:69) and the first check of
// its own test runs in double (ops/test.py:35-47).  Generic kernel, one thread per output element; not a tuned path (inference is fp32).
extern "C" int rba_ms_deform_attn_fwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                          const double* sampling_loc, const double* attn_weight, double* out,
                                          int N, int S, int M, int D, int L, int Lq, int P, void* stream) {
  RBA_CHECK_ARG(N >= 0 && S >= 1 && M >= 1 && D >= 1 && L >= 1 && Lq >= 0 && P >= 1);
  if (N == 0 || Lq == 0) return 0;
  RBA_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out);
  rba_begin();
  const int64_t total = (int64_t)N * Lq * M * D;
  const int64_t blocks = (total + 255) / 256;
  RBA_CHECK_ARG(blocks <= 0x7fffffffLL);
  hipLaunchKernelGGL((msda_fwd_kernel<double, 1>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, value, spatial_shapes, level_start_index,
                     sampling_loc, attn_weight, out, S, M, D, L, Lq, P, total);
  return rba_launch_status();
}

// MSDeformAttn.forward's core in ONE launch (pixel_decoder/ops/modules/ms_deform_attn.py:95-121): raw = the output of the sampling_offsets
// and attention_weights Linears evaluated as one Linear ([N * Lq, M L P 2 | M L P], as rba_msda_prepare_f32 takes it), reference_points
// [N * Lq, L, 2], value [N, S, M, 32] -> out [N, Lq, M * 32].  head_dim 32, P = 4, L in {1, 3} (hipErrorInvalidValue otherwise: the caller
// keeps rba_msda_prepare_f32 + rba_ms_deform_attn_fwd_f32).  Bit-identical to that pair.
extern "C" int rba_msda_fused_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const float* raw,
                                  const float* reference_points, float* out, int N, int S, int M, int D, int L, int Lq, int P, void* stream) {
  RBA_CHECK_ARG(N >= 0 && S >= 1 && M >= 1 && D == 32 && P == 4 && (L == 1 || L == 3) && Lq >= 0 && M <= 65535 && N <= 65535);
  RBA_CHECK_ARG((int64_t)S * M * 128 < ((int64_t)1 << 32));      // 32-bit tap offsets (launch_msda_lp); callers fall back to prepare + the generic kernel
  if (N == 0 || Lq == 0) return 0;
  RBA_CHECK_ARG(value && spatial_shapes && level_start_index && raw && reference_points && out);
  RBA_CHECK_ARG((((uintptr_t)value | (uintptr_t)out | (uintptr_t)raw) & 15) == 0);
  rba_begin();
  if (launch_msda_lp<true>(value, spatial_shapes, level_start_index, raw, reference_points, out, N, S, M, L, Lq, P, (hipStream_t)stream) != 0)
    return (int)hipErrorInvalidValue;
  return rba_launch_status();
}
