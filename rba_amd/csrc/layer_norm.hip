// Fused (residual add + bias) + LayerNorm over the last dimension (reference: the `x = shortcut + drop_path(x)` /
// `norm` pairs of backbone/swin.py:284-293, msdeformattn.py:134-138, mask2former_transformer_decoder.py:48-58,106-118,
// 171-175, all post-/pre-norm patterns of the form  s = x + (t + b);  y = LN(s)).
//
//   s[r,c]   = x[r,c] (+ t[r,c]) (+ tb[c])        written to sum_out if requested (may alias x)
//   y[r,c]   = (s - mean_r) * rstd_r * gamma[c] + beta[c]
//
// HBM-bound row kernel: a row lives in the registers of a G-lane group (16 B per lane per step), mean and centred
// variance are two in-register passes + xor-shuffles, so every element is read once and written once (twice with
// sum_out).  Replaces a separate bias-add pass in the GEMM epilogue, an elementwise add kernel and the LN kernel.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

template <int G, int NV>
__global__ __launch_bounds__(256) void add_layer_norm_kernel(const float* x, const float* __restrict__ t,
                                                             const float* __restrict__ tb, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* sum_out /* may alias x */,
                                                             float* __restrict__ y, int64_t rows, int C, float eps) {
  constexpr int RPW = 64 / G;                              // rows per wave
  const int lane = threadIdx.x & 63, sub = lane % G, rsel = lane / G;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + rsel;
  const bool rvalid = row < rows;
  const int64_t base = (rvalid ? row : 0) * C;
  const int nv4 = C >> 2;
  constexpr bool EARLY = NV <= 2;                          // hoisting costs 8 NV registers: measured faster for NV <= 2, slower above
  f32x4 v[NV], g4[EARLY ? NV : 1], b4[EARLY ? NV : 1];
  float sum = 0.f;
  // gamma / beta are requested together with the row, not after the two reductions that would otherwise wait for them
#pragma unroll
  for (int j = 0; j < (EARLY ? NV : 0); ++j) {
    const int c4 = sub + j * G;
    g4[j] = b4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c4 < nv4) {
      g4[j] = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
      b4[j] = *reinterpret_cast<const f32x4*>(beta + 4 * c4);
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c4 = sub + j * G;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c4 < nv4) {
      a = *reinterpret_cast<const f32x4*>(x + base + 4 * c4);
      if (t) a += *reinterpret_cast<const f32x4*>(t + base + 4 * c4);
      if (tb) a += *reinterpret_cast<const f32x4*>(tb + 4 * c4);
      sum += (a.x + a.y) + (a.z + a.w);
    }
    v[j] = a;
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, RBA_WAVE);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c4 = sub + j * G;
    if (c4 < nv4) {
      const f32x4 d = v[j] - mean;
      sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, RBA_WAVE);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  if (!rvalid) return;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c4 = sub + j * G;
    if (c4 < nv4) {
      if (sum_out) *reinterpret_cast<f32x4*>(sum_out + base + 4 * c4) = v[j];
      const f32x4 gj = EARLY ? g4[EARLY ? j : 0] : *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
      const f32x4 bj = EARLY ? b4[EARLY ? j : 0] : *reinterpret_cast<const f32x4*>(beta + 4 * c4);
      *reinterpret_cast<f32x4*>(y + base + 4 * c4) = (v[j] - mean) * rstd * gj + bj;
    }
  }
}

template <int G, int NV>
int launch(const float* x, const float* t, const float* tb, const float* gamma, const float* beta, float* sum_out, float* y,
           int64_t rows, int C, float eps, hipStream_t st) {
  const int64_t rpb = 4 * (64 / G);
  const int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((add_layer_norm_kernel<G, NV>), dim3((unsigned)blocks), dim3(256), 0, st, x, t, tb, gamma, beta, sum_out, y,
                     rows, C, eps);
  return rba_launch_status();
}

}  // namespace

extern "C" int rba_add_layer_norm_f32(const float* x, const float* t, const float* t_bias, const float* gamma, const float* beta,
                                      float* sum_out, float* y, int64_t rows, int C, float eps, void* stream) {
  RBA_CHECK_ARG(rows >= 0 && C >= 4 && C % 4 == 0 && C <= 8192);
  if (rows == 0) return 0;
  RBA_CHECK_ARG(x && gamma && beta && y);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)t | (uintptr_t)t_bias | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)sum_out |
                  (uintptr_t)y) & 15) == 0);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const int nv4 = C / 4;
#define RBA_L(G, NV) return launch<G, NV>(x, t, t_bias, gamma, beta, sum_out, y, rows, C, eps, st)
  if (nv4 <= 16) RBA_L(16, 1);
  if (nv4 <= 32) RBA_L(32, 1);
  if (nv4 <= 64) RBA_L(64, 1);
  if (nv4 <= 128) RBA_L(64, 2);
  if (nv4 <= 256) RBA_L(64, 4);
  if (nv4 <= 512) RBA_L(64, 8);
  if (nv4 <= 1024) RBA_L(64, 16);
  RBA_L(64, 32);
#undef RBA_L
}
