// DenseHybrid anomaly head (reference: mask2former_transformer_decoder.py:216-230 BNReluConv, :365-366, :467-468
// `out['ood_pred'] = self.ood_pred(mask_features)`; maskformer_model.py:303-305 bilinear up-sampling with align_corners=True;
// evaluate_ood.py:161-173 get_densehybrid_score).
//   rba_bn_relu_conv1x1_f32:       out[b, o, p] = bias[o] + sum_c w[o, c] * relu(x[b, c, p] * scale[c] + shift[c])
//                                  (eval-mode BatchNorm2d folded into scale / shift by the caller, ReLU, 1 x 1 convolution).
//                                  HBM bound: reads the [C, P] mask features once (4 pixels = 16 B per lane per channel plane, the
//                                  per-channel constants are wave-uniform scalar loads), writes O planes.
//   rba_resample_bilinear_ac_f32:  F.interpolate(mode="bilinear", align_corners=True) of [C, h, w] -> [C, H, W] (ATen
//                                  upsample_bilinear2d: source coordinate dst * (in - 1) / (out - 1)).
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

template <int O>
__global__ __launch_bounds__(256) void bn_relu_conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out, int C, int64_t P) {
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= P) return;
  const float* xb = x + (int64_t)blockIdx.y * C * P + p0;
  float acc[O][4];
#pragma unroll
  for (int o = 0; o < O; ++o)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[o][i] = bias ? bias[o] : 0.f;
  const bool full = p0 + 4 <= P && ((P & 3) == 0);
  for (int c = 0; c < C; ++c) {
    float v[4];
    if (full) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(xb + (int64_t)c * P);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = p0 + i < P ? xb[(int64_t)c * P + i] : 0.f;
    }
    const float s = scale[c], t0 = shift[c];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaxf(fmaf(v[i], s, t0), 0.f);
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const float wc = w[o * C + c];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[o][i] = fmaf(wc, v[i], acc[o][i]);
    }
  }
  float* ob = out + (int64_t)blockIdx.y * O * P + p0;
#pragma unroll
  for (int o = 0; o < O; ++o) {
    if (full) {
      *reinterpret_cast<f32x4*>(ob + (int64_t)o * P) = (f32x4){acc[o][0], acc[o][1], acc[o][2], acc[o][3]};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (p0 + i < P) ob[(int64_t)o * P + i] = acc[o][i];
    }
  }
}

__global__ __launch_bounds__(256) void resample_ac_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int h, int w,
                                                          int H, int W, float sh, float sw) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const float fy = sh * (float)y, fx = sw * (float)x;
  int y0 = (int)fy, x0 = (int)fx;
  y0 = y0 < h - 1 ? y0 : h - 1;
  x0 = x0 < w - 1 ? x0 : w - 1;
  const int y1 = y0 < h - 1 ? y0 + 1 : y0, x1 = x0 < w - 1 ? x0 + 1 : x0;
  const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  for (int c = blockIdx.z; c < C; c += gridDim.z) {
    const float* r0 = in + ((int64_t)c * h + y0) * w;
    const float* r1 = in + ((int64_t)c * h + y1) * w;
    // ATen: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (w0lambda * v10 + w1lambda * v11)
    out[((int64_t)c * H + y) * W + x] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
  }
}

}  // namespace

extern "C" int rba_bn_relu_conv1x1_f32(const float* x, const float* scale, const float* shift, const float* weight, const float* bias,
                                       float* out, int B, int C, int O, int64_t P, void* stream) {
  RBA_CHECK_ARG(B >= 0 && C >= 1 && (O == 1 || O == 2 || O == 4) && P >= 0 && B <= 65535);
  if (B == 0 || P == 0) return 0;
  RBA_CHECK_ARG(x && scale && shift && weight && out);
  RBA_CHECK_ARG((P & 3) != 0 || ((((uintptr_t)x | (uintptr_t)out) & 15) == 0));
  rba_begin();
  const dim3 grid((unsigned)((P + 1023) / 1024), (unsigned)B);
#define RBA_L(N) hipLaunchKernelGGL(bn_relu_conv1x1_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, x, scale, shift, weight, bias, out, C, P)
  if (O == 1) RBA_L(1); else if (O == 2) RBA_L(2); else RBA_L(4);
#undef RBA_L
  return rba_launch_status();
}

extern "C" int rba_resample_bilinear_ac_f32(const float* x, float* out, int C, int h, int w, int H, int W, void* stream) {
  RBA_CHECK_ARG(C >= 0 && h >= 1 && w >= 1 && H >= 0 && W >= 0 && H <= 65535);
  if (C == 0 || H == 0 || W == 0) return 0;
  RBA_CHECK_ARG(x && out);
  rba_begin();
  const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const dim3 grid((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)(C < 64 ? C : 64));
  hipLaunchKernelGGL(resample_ac_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, h, w, H, W, sh, sw);
  return rba_launch_status();
}
