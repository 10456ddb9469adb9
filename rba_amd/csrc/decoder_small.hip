// Two small element-wise steps of the masked decoder that were the last ATen launches on its path ("<= 5 at::native launches").
//
// rba_quad_mean_f32: the attention-mask logits of an intermediate decoder layer.  The sparse prediction head evaluates the mask logits only at the
//   2 x 2 source pixels each attention cell interpolates (mask2former_transformer_decoder.py, forward_prediction_heads; reference :472-489 computes the
//   whole map and F.interpolate's it); the bilinear sample at the centre of a 2 x 2 cell is 0.5 (0.5 v0 + 0.5 v1) + 0.5 (0.5 v2 + 0.5 v3)
//   = ((v0 + v1) + (v2 + v3)) * 0.25 -- powers of two are exact, so this order IS the interpolation bit for bit.  v [R, 4, S] -> out [R, S].
// rba_softmax_drop_last_f32: F.softmax(mask_cls, -1)[..., :-1] (maskformer_model.py:381-383 of the reference: the class probabilities K1 contracts
//   with, without the "no object" column).  One wave per row (K + 1 <= 64 classes), max and sum by wave reductions, accurate expf.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

__global__ __launch_bounds__(256) void quad_mean_kernel(const float* __restrict__ v, float* __restrict__ out, int64_t R, int S) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;               // one output element
  if (i >= R * S) return;
  const int64_t r = i / S;
  const int s = (int)(i - r * S);
  const float* p = v + r * 4 * S + s;
  out[i] = ((p[0] + p[S]) + (p[2 * (int64_t)S] + p[3 * (int64_t)S])) * 0.25f;
}

__global__ __launch_bounds__(256) void softmax_drop_last_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t R, int K1) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float val = lane < K1 ? x[r * K1 + lane] : -INFINITY;
  float m = val;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  const float e = lane < K1 ? expf(val - m) : 0.f;
  float s = e;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if (lane < K1 - 1) out[r * (K1 - 1) + lane] = e / s;
}

}  // namespace

extern "C" int rba_quad_mean_f32(const float* v, float* out, int64_t rows, int S, void* stream) {
  RBA_CHECK_ARG(rows >= 0 && S >= 0);
  if (rows == 0 || S == 0) return 0;
  RBA_CHECK_ARG(v && out && (rows * S + 255) / 256 < ((int64_t)1 << 31));
  rba_begin();
  const int64_t n = rows * S;
  hipLaunchKernelGGL(quad_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, out, rows, S);
  return rba_launch_status();
}

extern "C" int rba_softmax_drop_last_f32(const float* logits, float* prob, int64_t rows, int K1, void* stream) {
  RBA_CHECK_ARG(rows >= 0 && K1 >= 2 && K1 <= 64);
  if (rows == 0) return 0;
  RBA_CHECK_ARG(logits && prob && rows < ((int64_t)1 << 31));
  rba_begin();
  hipLaunchKernelGGL(softmax_drop_last_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, prob, rows, K1);
  return rba_launch_status();
}
