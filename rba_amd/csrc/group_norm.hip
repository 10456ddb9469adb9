// GroupNorm (+ optional ReLU) for the pixel decoder's conv -> GN(32) [-> ReLU] wrappers (reference:
// pixel_decoder/msdeformattn.py:222-235, 278-297 through Detectron2's Conv2d/get_norm("GN")).
//
// The library kernel reduces each of the 32 groups with ONE workgroup (32 workgroups on a 256-CU part:
// 2.3 ms per image in the first profile).  Here a group (C/G channels x HW, contiguous in NCHW) is cut into
// 64 KiB chunks: pass 1 writes per-chunk (n, mean, M2), pass 2 merges the chunk moments with Chan's formula in
// double precision and normalises its own chunk.  HBM-bound: x is read twice (second read mostly from
// Infinity Cache) and written once.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

constexpr int CHUNK = 16384;   // elements per workgroup (64 KiB), 256 threads x 16 float4

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RBA_WAVE);
  return v;
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ ws, int64_t gsize,
                                                       int splits) {
  const int s = blockIdx.x, g = blockIdx.y;
  const int64_t lo = (int64_t)s * CHUNK, hi = lo + CHUNK < gsize ? lo + CHUNK : gsize;
  const float* p = x + (int64_t)g * gsize;
  float sum = 0.f, sq = 0.f;
  if ((gsize & 3) == 0 && (((uintptr_t)p) & 15) == 0) {
    for (int64_t i = lo + threadIdx.x * 4; i < hi; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(p + i);
      sum += (v.x + v.y) + (v.z + v.w);
      sq = fmaf(v.x, v.x, sq); sq = fmaf(v.y, v.y, sq); sq = fmaf(v.z, v.z, sq); sq = fmaf(v.w, v.w, sq);
    }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const float v = p[i]; sum += v; sq = fmaf(v, v, sq); }
  }
  __shared__ double sh[8];
  double ds = wave_sum_d((double)sum), dq = wave_sum_d((double)sq);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[wave] = ds; sh[4 + wave] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ds = sh[0] + sh[1] + sh[2] + sh[3];
    dq = sh[4] + sh[5] + sh[6] + sh[7];
    const double n = (double)(hi - lo);
    const double mean = ds / n;
    double m2 = dq - ds * mean;
    float* o = ws + ((int64_t)g * splits + s) * 3;
    o[0] = (float)n; o[1] = (float)mean; o[2] = (float)(m2 > 0 ? m2 : 0);
  }
}

template <bool RELU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ ws,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, int64_t gsize, int splits, int HW,
                                                       int cpg, int G, float eps) {
  const int s = blockIdx.x, g = blockIdx.y;
  __shared__ float sh_mean, sh_rstd;
  if (threadIdx.x < 64) {   // one wave merges the chunk moments (Chan et al.) in double
    const float* w = ws + (int64_t)g * splits * 3;
    double n = 0, nm = 0;
    for (int i = threadIdx.x; i < splits; i += 64) { n += w[3 * i]; nm += (double)w[3 * i] * w[3 * i + 1]; }
    n = wave_sum_d(n); nm = wave_sum_d(nm);
    const double mean = nm / n;
    double m2 = 0;
    for (int i = threadIdx.x; i < splits; i += 64) {
      const double d = (double)w[3 * i + 1] - mean;
      m2 += (double)w[3 * i + 2] + (double)w[3 * i] * d * d;
    }
    m2 = wave_sum_d(m2);
    if (threadIdx.x == 0) { sh_mean = (float)mean; sh_rstd = (float)(1.0 / sqrt(m2 / n + (double)eps)); }
  }
  __syncthreads();
  const float mean = sh_mean, rstd = sh_rstd;
  const int64_t lo = (int64_t)s * CHUNK, hi = lo + CHUNK < gsize ? lo + CHUNK : gsize;
  const float* p = x + (int64_t)g * gsize;
  float* q = y + (int64_t)g * gsize;
  const int c0 = (g % G) * cpg;   // first channel of this group (batch folded into g)
  if ((HW & 3) == 0 && ((((uintptr_t)p) | ((uintptr_t)q)) & 15) == 0) {
    for (int64_t i = lo + threadIdx.x * 4; i < hi; i += 1024) {
      const int c = c0 + (int)(i / HW);   // HW % 4 == 0: the 4 elements share a channel
      const float a = gamma[c] * rstd, b = beta[c] - mean * a;
      float4 v = *reinterpret_cast<const float4*>(p + i);
      v.x = fmaf(v.x, a, b); v.y = fmaf(v.y, a, b); v.z = fmaf(v.z, a, b); v.w = fmaf(v.w, a, b);
      if (RELU) { v.x = rba_relu(v.x); v.y = rba_relu(v.y); v.z = rba_relu(v.z); v.w = rba_relu(v.w); }
      *reinterpret_cast<float4*>(q + i) = v;
    }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      const int c = c0 + (int)(i / HW);
      const float a = gamma[c] * rstd, b = beta[c] - mean * a;
      float v = fmaf(p[i], a, b);
      q[i] = RELU ? rba_relu(v) : v;
    }
  }
}

// ---- channels-last (NHWC) variant for the token-layout FPN path: x [B, P, C] (P pixels), statistics per (image, group) over
// P x (C / G) elements.  A thread owns one float4 of channels (one group when (C / G) % 4 == 0) and strides over pixels, so
// every access is a contiguous row; per-chunk moments use the same workspace layout and Chan merge as the NCHW kernels.
constexpr int PIX_CHUNK = 256;   // pixels per workgroup

__global__ __launch_bounds__(256) void gn_stats_nhwc_kernel(const float* __restrict__ x, float* __restrict__ ws, int P, int C,
                                                            int cpg, int splits) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int C4 = C >> 2, lanes = 256 / C4;                        // pixel lanes per workgroup
  const int j = threadIdx.x % C4, pl = threadIdx.x / C4;
  const int lo = s * PIX_CHUNK, hi = lo + PIX_CHUNK < P ? lo + PIX_CHUNK : P;
  const float4* xp = reinterpret_cast<const float4*>(x + (int64_t)b * P * C) + j;
  float sum = 0.f, sq = 0.f;
  if (pl < lanes)
#pragma unroll 8
    for (int p = lo + pl; p < hi; p += lanes) {
      const float4 v = xp[(int64_t)p * C4];
      sum += (v.x + v.y) + (v.z + v.w);
      sq = fmaf(v.x, v.x, sq); sq = fmaf(v.y, v.y, sq); sq = fmaf(v.z, v.z, sq); sq = fmaf(v.w, v.w, sq);
    }
  __shared__ float sh_s[256], sh_q[256];
  sh_s[threadIdx.x] = sum;
  sh_q[threadIdx.x] = sq;
  __syncthreads();
  const int G = C / cpg, q4 = cpg >> 2;                           // float4 chunks per group
  if (threadIdx.x < G) {
    const int g = threadIdx.x;
    double ds = 0, dq = 0;
    for (int l = 0; l < lanes; ++l)
      for (int k = 0; k < q4; ++k) { ds += sh_s[l * C4 + g * q4 + k]; dq += sh_q[l * C4 + g * q4 + k]; }
    const double n = (double)(hi - lo) * cpg;
    const double mean = ds / n;
    const double m2 = dq - ds * mean;
    float* o = ws + (((int64_t)b * G + g) * splits + s) * 3;
    o[0] = (float)n; o[1] = (float)mean; o[2] = (float)(m2 > 0 ? m2 : 0);
  }
}

// one wave per (image, group): Chan merge of the chunk moments in double -> (mean, rstd)
__global__ __launch_bounds__(64) void gn_merge_kernel(const float* __restrict__ ws, float* __restrict__ mr, int splits, float eps) {
  const float* w = ws + (int64_t)blockIdx.x * splits * 3;
  double n = 0, nm = 0;
  for (int i = threadIdx.x; i < splits; i += 64) { n += w[3 * i]; nm += (double)w[3 * i] * w[3 * i + 1]; }
  n = wave_sum_d(n); nm = wave_sum_d(nm);
  const double mean = nm / n;
  double m2 = 0;
  for (int i = threadIdx.x; i < splits; i += 64) {
    const double d = (double)w[3 * i + 1] - mean;
    m2 += (double)w[3 * i + 2] + (double)w[3 * i] * d * d;
  }
  m2 = wave_sum_d(m2);
  if (threadIdx.x == 0) { mr[2 * blockIdx.x] = (float)mean; mr[2 * blockIdx.x + 1] = (float)(1.0 / sqrt(m2 / n + (double)eps)); }
}

template <bool RELU>
__global__ __launch_bounds__(256) void gn_apply_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ mr,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, int P, int C, int cpg) {
  const int s = blockIdx.x, b = blockIdx.y;
  const int C4 = C >> 2, lanes = 256 / C4;
  const int j = threadIdx.x % C4, pl = threadIdx.x / C4;
  if (pl >= lanes) return;
  const int G = C / cpg, g = (j << 2) / cpg;
  const float mean = mr[2 * (b * G + g)], rstd = mr[2 * (b * G + g) + 1];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[j], be = reinterpret_cast<const float4*>(beta)[j];
  const float a0 = ga.x * rstd, a1 = ga.y * rstd, a2 = ga.z * rstd, a3 = ga.w * rstd;
  const float b0 = fmaf(-mean, a0, be.x), b1 = fmaf(-mean, a1, be.y), b2 = fmaf(-mean, a2, be.z), b3 = fmaf(-mean, a3, be.w);   // (what -ffp-contract=fast made of `be - mean * a`; explicit so that the folded forms match bit for bit)
  const int lo = s * PIX_CHUNK, hi = lo + PIX_CHUNK < P ? lo + PIX_CHUNK : P;
  const float4* xp = reinterpret_cast<const float4*>(x + (int64_t)b * P * C) + j;
  float4* yp = reinterpret_cast<float4*>(y + (int64_t)b * P * C) + j;
#pragma unroll 8
  for (int p = lo + pl; p < hi; p += lanes) {
    float4 v = xp[(int64_t)p * C4];
    v.x = fmaf(v.x, a0, b0); v.y = fmaf(v.y, a1, b1); v.z = fmaf(v.z, a2, b2); v.w = fmaf(v.w, a3, b3);
    if (RELU) { v.x = rba_relu(v.x); v.y = rba_relu(v.y); v.z = rba_relu(v.z); v.w = rba_relu(v.w); }
    yp[(int64_t)p * C4] = v;
  }
}

}  // namespace

extern "C" int64_t rba_group_norm_workspace_bytes(int B, int C, int HW, int G) {
  if (B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G) return 0;
  const int64_t gsize = (int64_t)(C / G) * HW;
  const int64_t splits = (gsize + CHUNK - 1) / CHUNK;
  return (int64_t)B * G * splits * 3 * (int64_t)sizeof(float);
}

extern "C" int rba_group_norm_f32(const float* x, const float* gamma, const float* beta, float* y, float* workspace,
                                  int B, int C, int HW, int G, float eps, int relu, void* stream) {
  RBA_CHECK_ARG(B >= 0 && C >= 1 && HW >= 0 && G >= 1 && C % G == 0);
  if (B == 0 || HW == 0) return 0;
  RBA_CHECK_ARG(x && gamma && beta && y && workspace);
  const int cpg = C / G;
  const int64_t gsize = (int64_t)cpg * HW;
  const int64_t splits = (gsize + CHUNK - 1) / CHUNK;
  RBA_CHECK_ARG(splits <= 0x7fffffff && (int64_t)B * G <= 65535);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)splits, B * G);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, st, x, workspace, gsize, (int)splits);
  if (relu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(256), 0, st, x, workspace, gamma, beta, y, gsize, (int)splits, HW, cpg, G, eps);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(256), 0, st, x, workspace, gamma, beta, y, gsize, (int)splits, HW, cpg, G, eps);
  return rba_launch_status();
}

extern "C" int64_t rba_group_norm_nhwc_workspace_bytes(int B, int P, int C, int G) {
  if (B <= 0 || C <= 0 || P <= 0 || G <= 0 || C % G) return 0;
  const int64_t splits = (P + PIX_CHUNK - 1) / PIX_CHUNK;
  return ((int64_t)B * G * splits * 3 + (int64_t)B * G * 2) * (int64_t)sizeof(float);
}

extern "C" int rba_group_norm_nhwc_f32(const float* x, const float* gamma, const float* beta, float* y, float* workspace, int B,
                                       int P, int C, int G, float eps, int relu, void* stream) {
  RBA_CHECK_ARG(B >= 0 && C >= 4 && P >= 0 && G >= 1 && C % G == 0 && (C / G) % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && G <= 256);
  if (B == 0 || P == 0) return 0;
  RBA_CHECK_ARG(x && gamma && beta && y && workspace && B <= 65535);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0);
  const int cpg = C / G;
  const int splits = (P + PIX_CHUNK - 1) / PIX_CHUNK;
  float* mr = workspace + (int64_t)B * G * splits * 3;
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(splits, B);
  hipLaunchKernelGGL(gn_stats_nhwc_kernel, grid, dim3(256), 0, st, x, workspace, P, C, cpg, splits);
  hipLaunchKernelGGL(gn_merge_kernel, dim3(B * G), dim3(64), 0, st, workspace, mr, splits, eps);
  if (relu)
    hipLaunchKernelGGL(gn_apply_nhwc_kernel<true>, grid, dim3(256), 0, st, x, mr, gamma, beta, y, P, C, cpg);
  else
    hipLaunchKernelGGL(gn_apply_nhwc_kernel<false>, grid, dim3(256), 0, st, x, mr, gamma, beta, y, P, C, cpg);
  return rba_launch_status();
}

// The statistics half of rba_group_norm_nhwc_f32 alone: mr [B][G][2] = (mean, rstd) per image and group, for consumers that fold the
// normalisation into their own loads (rba_resample_bilinear_nhwc_gn_f32).  workspace: rba_group_norm_nhwc_workspace_bytes; `mr` may point
// anywhere (B * G * 2 floats).
extern "C" int rba_group_norm_nhwc_stats_f32(const float* x, float* mr, float* workspace, int B, int P, int C, int G, float eps, void* stream) {
  RBA_CHECK_ARG(B >= 0 && C >= 4 && P >= 0 && G >= 1 && C % G == 0 && (C / G) % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && G <= 256);
  if (B == 0 || P == 0) return 0;
  RBA_CHECK_ARG(x && mr && workspace && B <= 65535 && (((uintptr_t)x) & 15) == 0);
  const int splits = (P + PIX_CHUNK - 1) / PIX_CHUNK;
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_nhwc_kernel, dim3(splits, B), dim3(256), 0, st, x, workspace, P, C, C / G, splits);
  hipLaunchKernelGGL(gn_merge_kernel, dim3(B * G), dim3(64), 0, st, workspace, mr, splits, eps);
  return rba_launch_status();
}

// The merge half alone: per-tile moments [B][G][splits][3] = (n, mean, M2) written by a producer's epilogue (rba_split_linear_f16x3_gn_moments_f32,
// rba_conv3x3_nhwc_f16x3_split_in_gn_moments_f32) -> mr [B][G][2] = (mean, rstd); the Chan merge in double of rba_group_norm_nhwc_stats_f32.
extern "C" int rba_group_norm_nhwc_merge_f32(const float* moments, float* mr, int B, int G, int splits, float eps, void* stream) {
  RBA_CHECK_ARG(B >= 0 && G >= 1 && splits >= 1 && (int64_t)B * G <= 0x7fffffff);
  if (B == 0) return 0;
  RBA_CHECK_ARG(moments && mr);
  rba_begin();
  hipLaunchKernelGGL(gn_merge_kernel, dim3(B * G), dim3(64), 0, (hipStream_t)stream, moments, mr, splits, eps);
  return rba_launch_status();
}
