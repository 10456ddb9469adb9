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
// SOUT: the result is written as the split image of the 3 x 3 convolution that consumes it (split_linear_h3.h "PRE" / "CONVP"): row =
// row0 + pixel, the lane pair of an 8-channel piece exchanges halves with one DPP move (as the LayerNorm does).  C % 32 == 0.
// GroupNorm folded into the loads (round 3: the FPN's `GroupNorm(lateral)` and `ReLU(GroupNorm(conv))` feed nothing but this kernel, so their
// "apply" passes -- 268 MB each at the 256 x 512 level -- disappear): a tensor that arrives with (mean, rstd) per group is normalised on the
// fly, v * (gamma rstd) + (beta - mean gamma rstd), the arithmetic of gn_apply_nhwc_kernel, ReLU optional.
struct GnFold {
  const float* mr;      // [G][2] (mean, rstd) of this image, or nullptr: the tensor is used as it is
  const float* gamma;
  const float* beta;
  int cpg;              // channels per group
  int relu;
};
__device__ __forceinline__ void gn_fold_coeff(const GnFold& g, int j, f32x4& a, f32x4& b) {
  const int grp = (j << 2) / g.cpg;
  const float mean = g.mr[2 * grp], rstd = g.mr[2 * grp + 1];
  const f32x4 ga = reinterpret_cast<const f32x4*>(g.gamma)[j], be = reinterpret_cast<const f32x4*>(g.beta)[j];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a[k] = ga[k] * rstd;
    b[k] = be[k] - mean * a[k];
  }
}
__device__ __forceinline__ f32x4 gn_fold_apply(f32x4 v, const f32x4 a, const f32x4 b, int relu) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = fmaf(v[k], a[k], b[k]);
    v[k] = rba_clamp_below(v[k], rba_relu_floor(relu));
  }
  return v;
}

template <bool ADD, bool SOUT = false, bool GN = false>
__global__ __launch_bounds__(256) void resample_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ add,
                                                            float* __restrict__ out, int C4, int h, int w, int H, int W, float sh,
                                                            float sw, int64_t row0 = 0, GnFold gin = GnFold{nullptr, nullptr, nullptr, 4, 0},
                                                            GnFold gadd = GnFold{nullptr, nullptr, nullptr, 4, 0}) {
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int j;
  int64_t pix;
  if (SOUT) {
    // a wave = 8 consecutive pixels x one 32-channel block (8 lanes x 16 bytes = the pixel's whole 128-byte line of `add`): the 8 pixels'
    // pieces of one (k-half, g, h|l) column are 128 contiguous bytes of the image, so every store instruction writes whole lines
    const int64_t wv = idx >> 6;
    const int ln = (int)(idx & 63), nblk = C4 >> 3;
    j = (int)(wv % nblk) * 8 + (ln & 7);
    pix = (wv / nblk) * 8 + (ln >> 3);
    if (pix >= (int64_t)H * W) return;
    idx = pix * C4 + j;
  } else {
    const int64_t total = (int64_t)H * W * C4;
    if (idx >= total) return;
    j = (int)(idx % C4);
    pix = idx / C4;
  }
  const int x = (int)(pix % W), y = (int)(pix / W);
  const BilinearTap ty = bilinear_tap(y, sh, h), tx = bilinear_tap(x, sw, w);
  const f32x4* ip = reinterpret_cast<const f32x4*>(in) + j;
  f32x4 v00 = ip[((int64_t)ty.i0 * w + tx.i0) * C4], v01 = ip[((int64_t)ty.i0 * w + tx.i1) * C4];
  f32x4 v10 = ip[((int64_t)ty.i1 * w + tx.i0) * C4], v11 = ip[((int64_t)ty.i1 * w + tx.i1) * C4];
  if (GN && gin.mr) {
    f32x4 a, b;
    gn_fold_coeff(gin, j, a, b);
    v00 = gn_fold_apply(v00, a, b, gin.relu);
    v01 = gn_fold_apply(v01, a, b, gin.relu);
    v10 = gn_fold_apply(v10, a, b, gin.relu);
    v11 = gn_fold_apply(v11, a, b, gin.relu);
  }
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float top = tx.l0 * v00[k] + tx.l1 * v01[k];
    const float bot = tx.l0 * v10[k] + tx.l1 * v11[k];
    r[k] = ty.l0 * top + ty.l1 * bot;
  }
  if (ADD) {
    f32x4 av = reinterpret_cast<const f32x4*>(add)[idx];
    if (GN && gadd.mr) {
      f32x4 a, b;
      gn_fold_coeff(gadd, j, a, b);
      av = gn_fold_apply(av, a, b, gadd.relu);
    }
    r += av;
  }
  if (SOUT) {
    uint32_t hh[2], ll[2];
    rba_split_f16x2(r.x, r.y, hh[0], ll[0]);
    rba_split_f16x2(r.z, r.w, hh[1], ll[1]);
    const bool odd = j & 1;                                                        // C4 is even: the pair (j, j ^ 1) is a lane pair
    const uint32_t s0 = odd ? hh[0] : ll[0], s1 = odd ? hh[1] : ll[1];
    const uint32_t r0 = __builtin_amdgcn_mov_dpp(s0, 0xB1, 0xf, 0xf, true), r1 = __builtin_amdgcn_mov_dpp(s1, 0xB1, 0xf, 0xf, true);
    const rba_u32x4 piece = odd ? (rba_u32x4){r0, r1, ll[0], ll[1]} : (rba_u32x4){hh[0], hh[1], r0, r1};
    const int64_t row = row0 + pix;
    const int k8 = j >> 1;
    const int64_t off = ((row >> 5) * (int64_t)(C4 >> 3) + (k8 >> 2)) * 4096 + ((k8 & 1) * 2 + (odd ? 1 : 0)) * 1024 +
                        (((k8 >> 1) & 1) * 32 + (int)(row & 31)) * 16;
    *reinterpret_cast<rba_u32x4*>(reinterpret_cast<char*>(out) + off) = piece;
  } else {
    reinterpret_cast<f32x4*>(out)[idx] = r;
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

// rba_resample_bilinear_nhwc_f32 with the result written as rows row0 .. row0 + H W - 1 of a split image of C-channel rows (the A operand
// of rba_conv3x3_nhwc_f16x3_split_in_f32).  C % 32 == 0; total threads a multiple of 2 per pixel, so a lane pair never straddles pixels.
extern "C" int rba_resample_bilinear_nhwc_split_out_f32(const float* in, const float* add, void* out_frag, int C, int h, int w, int H, int W,
                                                        int64_t row0, void* stream) {
  RBA_CHECK_ARG(C >= 32 && (C % 32) == 0 && h >= 1 && w >= 1 && H >= 0 && W >= 0 && row0 >= 0);
  if (H == 0 || W == 0) return 0;
  RBA_CHECK_ARG(in && out_frag && (((uintptr_t)in | (uintptr_t)add | (uintptr_t)out_frag) & 15) == 0);
  const int64_t total = (((int64_t)H * W + 7) / 8) * 8 * (C >> 2);                   // whole 8-pixel groups: a multiple of 64 threads
  RBA_CHECK_ARG((total + 255) / 256 <= 0x7fffffffLL);                              // (row0 % 8 != 0 only splits the 128-byte runs)
  rba_begin();
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const dim3 grid((unsigned)((total + 255) / 256));
  float* o = reinterpret_cast<float*>(out_frag);
  if (add)
    hipLaunchKernelGGL((resample_nhwc_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, in, add, o, C >> 2, h, w, H, W, sh, sw, row0);
  else
    hipLaunchKernelGGL((resample_nhwc_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, in, add, o, C >> 2, h, w, H, W, sh, sw, row0);
  return rba_launch_status();
}

// The FPN's top-down step with both GroupNorms folded in (pixel_decoder/msdeformattn.py:352-358: `cur_fpn = lateral_conv(x)` [conv + GN],
// `y = cur_fpn + F.interpolate(prev, size=cur_fpn.shape[-2:], mode="bilinear")` where prev = the previous level's `output_conv` [conv + GN + ReLU]):
// in [h, w, C] = the previous level's RAW convolution output (in_mr == nullptr: used as it is), add [H, W, C] = the RAW lateral convolution output;
// *_mr = [G][2] (mean, rstd) from rba_group_norm_nhwc_stats_f32.  out: fp32 [H, W, C] (split_out == 0) or rows row0 .. of a split image.
extern "C" int rba_resample_bilinear_nhwc_gn_f32(const float* in, const float* in_mr, const float* in_gamma, const float* in_beta, int in_relu,
                                                 const float* add, const float* add_mr, const float* add_gamma, const float* add_beta, void* out,
                                                 int split_out, int C, int G, int h, int w, int H, int W, int64_t row0, void* stream) {
  RBA_CHECK_ARG(C >= 4 && (C % 4) == 0 && G >= 1 && (C % G) == 0 && ((C / G) % 4) == 0 && h >= 1 && w >= 1 && H >= 0 && W >= 0 && row0 >= 0);
  RBA_CHECK_ARG(!split_out || (C % 32) == 0);
  if (H == 0 || W == 0) return 0;
  RBA_CHECK_ARG(in && add && out && (!in_mr || (in_gamma && in_beta)) && (!add_mr || (add_gamma && add_beta)));
  RBA_CHECK_ARG((((uintptr_t)in | (uintptr_t)add | (uintptr_t)out | (uintptr_t)in_gamma | (uintptr_t)in_beta | (uintptr_t)add_gamma | (uintptr_t)add_beta) & 15) == 0);
  const int64_t total = split_out ? (((int64_t)H * W + 7) / 8) * 8 * (C >> 2) : (int64_t)H * W * (C >> 2);
  RBA_CHECK_ARG((total + 255) / 256 <= 0x7fffffffLL);
  rba_begin();
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  const dim3 grid((unsigned)((total + 255) / 256));
  const GnFold gi{in_mr, in_gamma, in_beta, C / G, in_relu}, ga{add_mr, add_gamma, add_beta, C / G, 0};
  float* o = reinterpret_cast<float*>(out);
  if (split_out)
    hipLaunchKernelGGL((resample_nhwc_kernel<true, true, true>), grid, dim3(256), 0, (hipStream_t)stream, in, add, o, C >> 2, h, w, H, W, sh, sw, row0, gi, ga);
  else
    hipLaunchKernelGGL((resample_nhwc_kernel<true, false, true>), grid, dim3(256), 0, (hipStream_t)stream, in, add, o, C >> 2, h, w, H, W, sh, sw, row0, gi, ga);
  return rba_launch_status();
}
