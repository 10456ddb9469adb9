// Front end of the Swin path: image normalisation + ImageList zero padding + the im2col of PatchEmbed's 4 x 4 / stride 4 convolution
// in one pass (reference: maskformer_model.py:255-257 `(x - pixel_mean) / pixel_std` + ImageList.from_tensors, backbone/swin.py:479-495
// PatchEmbed.proj).  out[token][k], token = ty * (Wp / 4) + tx, k = c * 16 + ky * 4 + kx = the flattened [3,4,4] weight index,
// k = 48 .. 63 zero (K padded to 64 for the f16x3 GEMM that follows: rba_split_linear_f16x3_f32 with the [E, 64] zero-padded weight).
// Pixels outside the h x w image (ImageList pads bottom / right with zeros AFTER normalisation) give 0.
// Replaces: uint8 -> float, subtract, divide, new_zeros, copy, MIOpen convolution, and the 67 MB NCHW -> token-major copy.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const T* __restrict__ img, float* __restrict__ out, int h, int w, int Wt,
                                                          int64_t tokens, float m0, float m1, float m2, float s0, float s1, float s2) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;                   // (token, 16-byte piece p of its 256-byte row)
  const int64_t token = idx >> 4;
  if (token >= tokens) return;
  const int p = (int)(idx & 15);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (p < 12) {
    const int c = p >> 2, ky = p & 3;
    const int ty = (int)(token / Wt), tx = (int)(token - (int64_t)ty * Wt);
    const int y = 4 * ty + ky, x0 = 4 * tx;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    if (y < h) {
      const T* row = img + ((int64_t)c * h + y) * w;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
        if (x0 + kx < w) v[kx] = ((float)row[x0 + kx] - mean) / sd;
    }
  }
  *reinterpret_cast<f32x4*>(out + token * 64 + 4 * p) = v;
}

template <typename T>
int launch_im2col(const T* img, float* out, int h, int w, int Hp, int Wp, const float* mean, const float* sd, hipStream_t st) {
  const int Wt = Wp / 4;
  const int64_t tokens = (int64_t)(Hp / 4) * Wt;
  const int64_t threads = tokens * 16;
  hipLaunchKernelGGL(patch_im2col_kernel<T>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, img, out, h, w, Wt, tokens, mean[0],
                     mean[1], mean[2], sd[0], sd[1], sd[2]);
  return rba_launch_status();
}

}  // namespace

// image [3, h, w] (uint8 or fp32, on the device) -> out [(Hp / 4) * (Wp / 4), 64] fp32; Hp >= h, Wp >= w multiples of 4; mean / std: 3 HOST floats each
extern "C" int rba_patch_im2col_u8(const uint8_t* image, float* out, int h, int w, int Hp, int Wp, const float* mean, const float* std,
                                   void* stream) {
  RBA_CHECK_ARG(image && out && mean && std && h >= 1 && w >= 1 && Hp >= h && Wp >= w && (Hp % 4) == 0 && (Wp % 4) == 0);
  RBA_CHECK_ARG(((uintptr_t)out & 15) == 0 && (int64_t)Hp * Wp < (int64_t)1 << 31);
  rba_begin();
  return launch_im2col<uint8_t>(image, out, h, w, Hp, Wp, mean, std, (hipStream_t)stream);
}

extern "C" int rba_patch_im2col_f32(const float* image, float* out, int h, int w, int Hp, int Wp, const float* mean, const float* std,
                                    void* stream) {
  RBA_CHECK_ARG(image && out && mean && std && h >= 1 && w >= 1 && Hp >= h && Wp >= w && (Hp % 4) == 0 && (Wp % 4) == 0);
  RBA_CHECK_ARG(((uintptr_t)out & 15) == 0 && (int64_t)Hp * Wp < (int64_t)1 << 31);
  rba_begin();
  return launch_im2col<float>(image, out, h, w, Hp, Wp, mean, std, (hipStream_t)stream);
}
