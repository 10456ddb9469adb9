// K5 -- Swin (shifted-)window attention core (reference: backbone/swin.py:131-171 WindowAttention.forward with
// the pad / roll / window_partition / window_reverse / un-roll / crop of SwinTransformerBlock.forward :251-284
// and the SW-MSA mask of BasicLayer.forward :413-440 folded in as index arithmetic).
//
// The reference materialises five token-map copies per block (pad, roll, partition, reverse, un-roll) plus the
// [nW*nH, N, N] attention matrix (314 MB at stage 1 of Swin-B @1024x2048).  Here one workgroup owns one
// (window, head): K and V of its N = ws*ws tokens are gathered straight from the un-padded qkv tensor into LDS
// (padded tokens take the qkv bias = Linear(0)), each thread owns one query row and runs a chunked online
// softmax over the keys (16 keys per chunk: scores in registers, one rescale per chunk), and the result is
// scattered back to the token's original position.  The shift mask is recomputed from region ids.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

constexpr int CH = 16;   // keys per softmax chunk

template <int HD>
__global__ __launch_bounds__(256) void swin_window_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int H, int W, int Hp, int Wp, int nH, int ws, int shift, float scale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int N = ws * ws;
  float* Ks = lds;                 // [N][HD]
  float* Vs = lds + N * HD;        // [N][HD]
  int* rid = reinterpret_cast<int*>(lds + 2 * N * HD);   // [N] region id (shift mask)
  const int wx = blockIdx.x, wy = blockIdx.y;
  const int h = blockIdx.z % nH, b = blockIdx.z / nH;
  const int C = nH * HD;
  const int64_t tok_stride = 3 * (int64_t)C;
  const float* qkv_b = qkv + (int64_t)b * H * W * tok_stride;

  // ---- gather K, V (float4 granularity: N*8 float4 each) ----
  for (int i = threadIdx.x; i < N * (HD / 4); i += blockDim.x) {
    const int t = i / (HD / 4), d4 = i % (HD / 4);
    const int r = wy * ws + t / ws, c = wx * ws + t % ws;            // shifted-frame position
    int rr = r + shift, cc = c + shift;                               // padded-frame position
    rr = rr >= Hp ? rr - Hp : rr;
    cc = cc >= Wp ? cc - Wp : cc;
    float4 kk, vv;
    if (rr < H && cc < W) {
      const float* p = qkv_b + ((int64_t)rr * W + cc) * tok_stride + h * HD + d4 * 4;
      kk = *reinterpret_cast<const float4*>(p + C);
      vv = *reinterpret_cast<const float4*>(p + 2 * C);
    } else {
      kk = *reinterpret_cast<const float4*>(qkv_bias + C + h * HD + d4 * 4);
      vv = *reinterpret_cast<const float4*>(qkv_bias + 2 * C + h * HD + d4 * 4);
    }
    *reinterpret_cast<float4*>(Ks + t * HD + d4 * 4) = kk;
    *reinterpret_cast<float4*>(Vs + t * HD + d4 * 4) = vv;
    if (d4 == 0) {
      const int hid = r < Hp - ws ? 0 : (r < Hp - shift ? 1 : 2);
      const int wid = c < Wp - ws ? 0 : (c < Wp - shift ? 1 : 2);
      rid[t] = hid * 3 + wid;
    }
  }
  __syncthreads();

  const int t = threadIdx.x;
  if (t >= N) return;
  const int r = wy * ws + t / ws, c = wx * ws + t % ws;
  int rr = r + shift, cc = c + shift;
  rr = rr >= Hp ? rr - Hp : rr;
  cc = cc >= Wp ? cc - Wp : cc;
  const bool valid = rr < H && cc < W;
  // padded query rows are dropped by the crop, but keep the lanes busy-free: nothing to do for them
  if (!valid) return;

  float q[HD];
  {
    const float4* p = reinterpret_cast<const float4*>(qkv_b + ((int64_t)rr * W + cc) * tok_stride + h * HD);
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
      const float4 v4 = p[i];
      q[4 * i] = v4.x * scale; q[4 * i + 1] = v4.y * scale; q[4 * i + 2] = v4.z * scale; q[4 * i + 3] = v4.w * scale;
    }
  }
  const int myrid = rid[t];
  const float* brow = bias + ((int64_t)h * N + t) * N;
  float m = -INFINITY, l = 0.f, acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;

  for (int j0 = 0; j0 < N; j0 += CH) {
    float s[CH];
    float cmax = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < CH; ++jj) {
      const int j = j0 + jj;
      float sc = -INFINITY;
      if (j < N) {
        const float4* kr = reinterpret_cast<const float4*>(Ks + j * HD);   // wave-uniform address: LDS broadcast
        sc = 0.f;
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
          const float4 kk = kr[i];
          sc = fmaf(q[4 * i], kk.x, sc); sc = fmaf(q[4 * i + 1], kk.y, sc);
          sc = fmaf(q[4 * i + 2], kk.z, sc); sc = fmaf(q[4 * i + 3], kk.w, sc);
        }
        sc += brow[j];
        if (shift > 0 && rid[j] != myrid) sc += -100.0f;
      }
      s[jj] = sc;
      cmax = fmaxf(cmax, sc);
    }
    if (cmax > m) {
      const float a = __expf(m - cmax);
      l *= a;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] *= a;
      m = cmax;
    }
#pragma unroll
    for (int jj = 0; jj < CH; ++jj) {
      const int j = j0 + jj;
      if (j < N) {
        const float p = __expf(s[jj] - m);
        l += p;
        const float4* vr = reinterpret_cast<const float4*>(Vs + j * HD);
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
          const float4 vv = vr[i];
          acc[4 * i] = fmaf(p, vv.x, acc[4 * i]); acc[4 * i + 1] = fmaf(p, vv.y, acc[4 * i + 1]);
          acc[4 * i + 2] = fmaf(p, vv.z, acc[4 * i + 2]); acc[4 * i + 3] = fmaf(p, vv.w, acc[4 * i + 3]);
        }
      }
    }
  }
  const float inv = 1.0f / l;
  float4* o = reinterpret_cast<float4*>(out + ((int64_t)b * H * W + (int64_t)rr * W + cc) * C + h * HD);
#pragma unroll
  for (int i = 0; i < HD / 4; ++i)
    o[i] = make_float4(acc[4 * i] * inv, acc[4 * i + 1] * inv, acc[4 * i + 2] * inv, acc[4 * i + 3] * inv);
}

}  // namespace

extern "C" int rba_swin_window_attn_f32(const float* qkv, const float* qkv_bias, const float* bias, float* out,
                                        int B, int H, int W, int nH, int hd, int ws, int shift, void* stream) {
  RBA_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && nH >= 1 && (hd == 16 || hd == 32 || hd == 64));
  RBA_CHECK_ARG(ws >= 1 && ws * ws <= 256 && shift >= 0 && shift < ws);
  if (B == 0) return 0;
  RBA_CHECK_ARG(qkv && qkv_bias && bias && out);
  RBA_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)qkv_bias | (uintptr_t)out) & 15) == 0);
  const int Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws;
  const int N = ws * ws;
  RBA_CHECK_ARG((int64_t)B * nH <= 65535 && Hp / ws <= 65535);
  const int threads = (N + 63) / 64 * 64;
  rba_begin();
  const size_t shm = (size_t)(2 * N * hd) * sizeof(float) + (size_t)N * sizeof(int);
  const float scale = (float)(1.0 / sqrt((double)hd));   // head_dim ** -0.5 in double, then fp32 (swin.py:103,145)
  const dim3 grid(Wp / ws, Hp / ws, B * nH);
#define RBA_L(D) hipLaunchKernelGGL(swin_window_attn_kernel<D>, grid, dim3(threads), shm, (hipStream_t)stream, qkv, qkv_bias, bias, out, H, W, Hp, Wp, nH, ws, shift, scale)
  if (hd == 16) RBA_L(16);
  else if (hd == 32) RBA_L(32);
  else RBA_L(64);
#undef RBA_L
  return rba_launch_status();
}
