// Bilinear resample, align_corners=False (ATen upsample_bilinear2d semantics; reference call sites:
// maskformer_model.py:294-299 mask upsample, msdeformattn.py:358 FPN top-down, decoder.py:483 attn-mask
// downsample).  Pure bandwidth: one thread writes 4 consecutive output columns (16 B store) for a chunk
// of channels, source rows are re-read through L1/L2.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

constexpr int CCHUNK = 4;

template <bool ADD, bool VEC4>
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ in, const float* __restrict__ add,
                                                       float* __restrict__ out, int C, int h, int w, int H, int W,
                                                       float sh, float sw, int wq) {
  const int xg = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (xg >= wq) return;
  const int x0 = xg * 4;
  const BilinearTap ty = bilinear_tap(y, sh, h);
  BilinearTap tx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) tx[r] = bilinear_tap(x0 + r < W ? x0 + r : W - 1, sw, w);
  const int c0 = blockIdx.z * CCHUNK;
  const int c1 = c0 + CCHUNK < C ? c0 + CCHUNK : C;
  for (int c = c0; c < c1; ++c) {
    const float* r0 = in + ((int64_t)c * h + ty.i0) * w;
    const float* r1 = in + ((int64_t)c * h + ty.i1) * w;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float top = tx[r].l0 * r0[tx[r].i0] + tx[r].l1 * r0[tx[r].i1];
      const float bot = tx[r].l0 * r1[tx[r].i0] + tx[r].l1 * r1[tx[r].i1];
      v[r] = ty.l0 * top + ty.l1 * bot;
    }
    const int64_t o = ((int64_t)c * H + y) * W + x0;
    if (VEC4) {
      f32x4 t = {v[0], v[1], v[2], v[3]};
      if (ADD) t += *reinterpret_cast<const f32x4*>(add + o);
      *reinterpret_cast<f32x4*>(out + o) = t;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (x0 + r < W) out[o + r] = ADD ? v[r] + add[o + r] : v[r];
    }
  }
}

}  // namespace

extern "C" int rba_resample_bilinear_f32(const float* in, const float* add, float* out, int C, int h, int w, int H, int W,
                                         void* stream) {
  RBA_CHECK_ARG(C >= 0 && h >= 1 && w >= 1 && H >= 0 && W >= 0 && H <= 65535);
  if (C == 0 || H == 0 || W == 0) return 0;
  RBA_CHECK_ARG(in && out);
  const int cz = (C + CCHUNK - 1) / CCHUNK;
  RBA_CHECK_ARG(cz <= 65535);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const int wq = (W + 3) / 4;
  const int threads = wq >= 256 ? 256 : (wq >= 128 ? 128 : 64);
  dim3 grid((wq + threads - 1) / threads, H, cz);
  // ATen: scale = in/out computed in float (area_pixel_compute_scale, align_corners=False, no scale_factor)
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const bool vec4 = (W % 4 == 0) && ((((uintptr_t)out | (uintptr_t)add) & 15) == 0);
#define RBA_L(A, V) hipLaunchKernelGGL((resample_kernel<A, V>), grid, dim3(threads), 0, st, in, add, out, C, h, w, H, W, sh, sw, wq)
  if (add && vec4) RBA_L(true, true);
  else if (add) RBA_L(true, false);
  else if (vec4) RBA_L(false, true);
  else RBA_L(false, false);
#undef RBA_L
  return rba_launch_status();
}
