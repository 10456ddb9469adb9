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
  // one channel: 16 tap loads -> 4 outputs.  The body is unrolled over the CCHUNK channels of the block so that all their loads are in
  // flight together (a rolled loop paid the memory latency once per channel); the map is written once and read once by the next
  // kernel from a 839 MB tensor -> non-temporal stores.
  auto one = [&](int c) {
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
      if (ADD) {
        t += *reinterpret_cast<const f32x4*>(add + o);
        *reinterpret_cast<f32x4*>(out + o) = t;
      } else {
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(out + o));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (x0 + r < W) out[o + r] = ADD ? v[r] + add[o + r] : v[r];
    }
  };
  if (c0 + CCHUNK <= C) {
#pragma unroll
    for (int i = 0; i < CCHUNK; ++i) one(c0 + i);
  } else {
    for (int c = c0; c < C; ++c) one(c);
  }
}

// channels-last variant: in [h, w, C] -> out [H, W, C] (+ add).  A thread owns one float4 of channels of one output pixel:
// the four taps are four contiguous 16-byte loads, a wave covers whole pixel rows.  Same tap arithmetic as above.
template <bool ADD>
__global__ __launch_bounds__(256) void resample_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ add,
                                                            float* __restrict__ out, int C4, int h, int w, int H, int W, float sh,
                                                            float sw) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)H * W * C4;
  if (idx >= total) return;
  const int j = (int)(idx % C4);
  const int64_t pix = idx / C4;
  const int x = (int)(pix % W), y = (int)(pix / W);
  const BilinearTap ty = bilinear_tap(y, sh, h), tx = bilinear_tap(x, sw, w);
  const f32x4* ip = reinterpret_cast<const f32x4*>(in) + j;
  const f32x4 v00 = ip[((int64_t)ty.i0 * w + tx.i0) * C4], v01 = ip[((int64_t)ty.i0 * w + tx.i1) * C4];
  const f32x4 v10 = ip[((int64_t)ty.i1 * w + tx.i0) * C4], v11 = ip[((int64_t)ty.i1 * w + tx.i1) * C4];
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float top = tx.l0 * v00[k] + tx.l1 * v01[k];
    const float bot = tx.l0 * v10[k] + tx.l1 * v11[k];
    r[k] = ty.l0 * top + ty.l1 * bot;
  }
  if (ADD) r += reinterpret_cast<const f32x4*>(add)[idx];
  reinterpret_cast<f32x4*>(out)[idx] = r;
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

extern "C" int rba_resample_bilinear_nhwc_f32(const float* in, const float* add, float* out, int C, int h, int w, int H, int W,
                                              void* stream) {
  RBA_CHECK_ARG(C >= 0 && (C & 3) == 0 && h >= 1 && w >= 1 && H >= 0 && W >= 0);
  if (C == 0 || H == 0 || W == 0) return 0;
  RBA_CHECK_ARG(in && out && ((((uintptr_t)in | (uintptr_t)out | (uintptr_t)add) & 15) == 0));
  const int64_t total = (int64_t)H * W * (C >> 2);
  RBA_CHECK_ARG((total + 255) / 256 <= 0x7fffffff);
  rba_begin();
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (add)
    hipLaunchKernelGGL(resample_nhwc_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, add, out, C >> 2, h, w, H, W, sh, sw);
  else
    hipLaunchKernelGGL(resample_nhwc_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, add, out, C >> 2, h, w, H, W, sh, sw);
  return rba_launch_status();
}
