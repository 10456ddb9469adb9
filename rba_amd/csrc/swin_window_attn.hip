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
#include <stdlib.h>
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


// ------------------------------------------------------------------------------------------------------------
// MFMA version (head_dim 32).  v1 above is LDS-broadcast bound (16 ds_read_b128 per 64 FMAs).  Here each wave owns
// 16-query strips and runs both contractions on the matrix pipe with exact-fp32 v_mfma_f32_16x16x4_f32:
//   S^T = K . Q^T      A = K tile (keys x d, from LDS), B = Q^T (d x queries, registers)  -> D[key][query]
//   O   = P . V        A = P (queries x keys) -- which IS the D registers of S^T, element for element, because the
//                      MFMA k-index is only a summation label: step (c, r) of the P.V chain uses key c*16 + 4*kk + r,
//                      exactly the key lane (kk = lane>>4) holds in register r of tile c.  No transpose, no LDS trip.
// With D[key][query] the softmax axis (keys) is in-lane (NT*4 registers) plus two xor-shuffles (lanes 16/32 apart).
// The d-index of the QK^T chain is permuted (step s, slot kk -> d = 8*kk + s) so a lane's 8 K values are contiguous:
// two ds_read_b128 per key tile.  LDS row stride 36 floats keeps those and the per-key V reads off shared banks.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int NT, int WAVES, bool FRAG>
__global__ __launch_bounds__(64 * WAVES) void swin_window_attn_mfma_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias, float* __restrict__ out,
    int H, int W, int Hp, int Wp, int nH, int ws, int shift, float scale) {
  constexpr int HD = 32, RS = 36, NP = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ks = lds;                                          // [NP][RS]
  float* Vs = lds + NP * RS;                                // [NP][RS]
  int* tok = reinterpret_cast<int*>(lds + 2 * NP * RS);     // [NP] token index in the un-padded map, -1 = zero-padded, -2 = none
  int* rid = tok + NP;                                      // [NP] shift-mask region id
  const int N = ws * ws;
  const int wx = blockIdx.x, wy = blockIdx.y;
  const int h = blockIdx.z % nH, b = blockIdx.z / nH;
  const int C = nH * HD;
  const int64_t tok_stride = 3 * (int64_t)C;
  const float* qkv_b = qkv + (int64_t)b * H * W * tok_stride;
  const float* qb = qkv_bias + h * HD;

  for (int i = threadIdx.x; i < NP * (HD / 4); i += 64 * WAVES) {
    const int t = i >> 3, d4 = i & 7;
    float4 kk4 = make_float4(0.f, 0.f, 0.f, 0.f), vv4 = kk4;
    int tk = -2, rg = -1;
    if (t < N) {
      const int r = wy * ws + t / ws, c = wx * ws + t % ws;
      int rr = r + shift, cc = c + shift;
      rr = rr >= Hp ? rr - Hp : rr;
      cc = cc >= Wp ? cc - Wp : cc;
      if (rr < H && cc < W) {
        tk = rr * W + cc;
        const float* p = qkv_b + (int64_t)tk * tok_stride + h * HD + d4 * 4;
        kk4 = *reinterpret_cast<const float4*>(p + C);
        vv4 = *reinterpret_cast<const float4*>(p + 2 * C);
      } else {
        tk = -1;
        kk4 = *reinterpret_cast<const float4*>(qb + C + d4 * 4);
        vv4 = *reinterpret_cast<const float4*>(qb + 2 * C + d4 * 4);
      }
      const int hid = r < Hp - ws ? 0 : (r < Hp - shift ? 1 : 2);
      const int wid = c < Wp - ws ? 0 : (c < Wp - shift ? 1 : 2);
      rg = hid * 3 + wid;
    }
    *reinterpret_cast<float4*>(Ks + t * RS + d4 * 4) = kk4;
    *reinterpret_cast<float4*>(Vs + t * RS + d4 * 4) = vv4;
    if (d4 == 0) { tok[t] = tk; rid[t] = rg; }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kk = lane >> 4;
  const bool vec_bias = (N & 3) == 0;

  for (int strip = wave; strip < NT; strip += WAVES) {
    const int qt = strip * 16 + l15;                       // this lane's query (as B column / softmax owner)
    const int qtok = tok[qt];
    // ---- Q fragment: Q[qt][8*kk + s], s = 0..7, pre-scaled
    float q8[8];
    {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c4 = a;
      if (qtok >= 0) {
        const float4* p = reinterpret_cast<const float4*>(qkv_b + (int64_t)qtok * tok_stride + h * HD + kk * 8);
        a = p[0]; c4 = p[1];
      } else if (qtok == -1) {
        const float4* p = reinterpret_cast<const float4*>(qb + kk * 8);
        a = p[0]; c4 = p[1];
      }
      q8[0] = a.x * scale; q8[1] = a.y * scale; q8[2] = a.z * scale; q8[3] = a.w * scale;
      q8[4] = c4.x * scale; q8[5] = c4.y * scale; q8[6] = c4.z * scale; q8[7] = c4.w * scale;
    }
    // ---- S^T tiles: lane holds S[key = c*16 + 4*kk + r][query = qt]
    f32x4_t S[NT];
#pragma unroll
    for (int c = 0; c < NT; c += 2) {                      // two independent accumulator chains in flight
      const float4* kr = reinterpret_cast<const float4*>(Ks + (c * 16 + l15) * RS + kk * 8);
      const float4* kr2 = reinterpret_cast<const float4*>(Ks + ((c + 1 < NT ? c + 1 : c) * 16 + l15) * RS + kk * 8);
      const float4 k0 = kr[0], k1 = kr[1], j0 = kr2[0], j1 = kr2[1];
      const float ka[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
      const float kb[8] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w};
      f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[st], q8[st], a0, 0, 0, 0);
        if (c + 1 < NT) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[st], q8[st], a1, 0, 0, 0);
      }
      S[c] = a0;
      if (c + 1 < NT) S[c + 1] = a1;
      __builtin_amdgcn_sched_barrier(0);                   // keep the unrolled tiles from hoisting all LDS reads
    }
    // ---- + relative-position bias, shift mask, padding keys; row max
    const int myrid = rid[qt];
    const float* brow = bias + ((int64_t)h * N + (qt < N ? qt : 0)) * N;
    // FRAG: bias pre-permuted to [nH][strip][c][lane][4] (rba_swin_bias_fragments_f32): 1 KiB coalesced per load
    const float4* bfrag = reinterpret_cast<const float4*>(bias) + (((int64_t)h * NT + strip) * NT) * 64 + lane;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const int k0i = c * 16 + kk * 4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (FRAG) {
        const float4 t4 = bfrag[c * 64];
        bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
      } else if (qt < N) {
        if (vec_bias && k0i + 3 < N) {
          const float4 t4 = *reinterpret_cast<const float4*>(brow + k0i);
          bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (k0i + r < N) bv[r] = brow[k0i + r];
        }
      }
      const int4 kr4 = *reinterpret_cast<const int4*>(rid + k0i);
      const int krid[4] = {kr4.x, kr4.y, kr4.z, kr4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = S[c][r] + bv[r];
        if (shift > 0 && krid[r] != myrid) v += -100.0f;
        if (k0i + r >= N) v = -INFINITY;
        S[c][r] = v;
        m = fmaxf(m, v);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, RBA_WAVE));
    m = fmaxf(m, __shfl_xor(m, 32, RBA_WAVE));
    float lsum = 0.f;
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(S[c][r] - m);
        S[c][r] = p;
        lsum += p;
      }
    lsum += __shfl_xor(lsum, 16, RBA_WAVE);
    lsum += __shfl_xor(lsum, 32, RBA_WAVE);
    // ---- O = P . V  (two 16-wide d tiles)
    f32x4_t O0 = {0.f, 0.f, 0.f, 0.f}, O1 = O0;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const float* vr = Vs + (c * 16 + kk * 4) * RS + l15;
      float v0[4], v1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v0[r] = vr[r * RS]; v1[r] = vr[r * RS + 16]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        O0 = __builtin_amdgcn_mfma_f32_16x16x4f32(S[c][r], v0[r], O0, 0, 0, 0);
        O1 = __builtin_amdgcn_mfma_f32_16x16x4f32(S[c][r], v1[r], O1, 0, 0, 0);
      }
      if (c & 1) __builtin_amdgcn_sched_barrier(0);        // at most two tiles of V operands live
    }
    // ---- scatter: lane holds O[query = strip*16 + 4*kk + r][d = l15 (+16)]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = kk * 4 + r;
      const float inv = 1.0f / __shfl(lsum, qi, RBA_WAVE);
      const int t = tok[strip * 16 + qi];
      if (t >= 0) {
        float* o = out + ((int64_t)b * H * W + t) * C + h * HD + l15;
        o[0] = O0[r] * inv;
        o[16] = O1[r] * inv;
      }
    }
  }
}

template <int NT, int WAVES>
int launch_mfma(const float* qkv, const float* qkv_bias, const float* bias, const float* bias_frag, float* out, int B, int H,
                int W, int Hp, int Wp, int nH, int ws, int shift, float scale, hipStream_t st) {
  const size_t shm = (size_t)(2 * NT * 16 * 36) * sizeof(float) + (size_t)(2 * NT * 16) * sizeof(int);
  const dim3 grid(Wp / ws, Hp / ws, B * nH), block(64 * WAVES);
  if (bias_frag)
    hipLaunchKernelGGL((swin_window_attn_mfma_kernel<NT, WAVES, true>), grid, block, shm, st, qkv, qkv_bias, bias_frag, out,
                       H, W, Hp, Wp, nH, ws, shift, scale);
  else
    hipLaunchKernelGGL((swin_window_attn_mfma_kernel<NT, WAVES, false>), grid, block, shm, st, qkv, qkv_bias, bias, out,
                       H, W, Hp, Wp, nH, ws, shift, scale);
  return rba_launch_status();
}

// bias [nH,N,N] -> frag [nH,NT,NT,64,4]: frag[h][strip][c][lane][r] = bias[h][strip*16 + (lane&15)][c*16 + 4*(lane>>4) + r]
// (0 outside N): the order in which the MFMA kernel's lanes consume it.
__global__ void swin_bias_fragments_kernel(const float* __restrict__ bias, float* __restrict__ frag, int nH, int N, int NT) {
  const int64_t total = (int64_t)nH * NT * NT * 256;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i & 3), lane = (int)((i >> 2) & 63);
    int64_t j = i >> 8;
    const int c = (int)(j % NT); j /= NT;
    const int strip = (int)(j % NT);
    const int h = (int)(j / NT);
    const int q = strip * 16 + (lane & 15), k = c * 16 + 4 * (lane >> 4) + r;
    frag[i] = (q < N && k < N) ? bias[((int64_t)h * N + q) * N + k] : 0.f;
  }
}

}  // namespace

#include "swin_window_attn_h3.h"

// Register budget of the kernel (round 4, late): at 96 VGPRs a gfx950 CU holds ONE 9-wave workgroup although the occupancy API answers two (tools/micro/occ_probe.hip); at
// amdgpu_waves_per_eu(6, 8) the same source compiles to 79 VGPRs without scratch, two workgroups are resident, outputs bit-identical.  Launch time: equal to 15 % shorter
// depending on the box and the stage, 0-4 % with cold operands (profiles/r04_k5_wpe_ab.txt).  rba_k5_wpe (tools): 6 = default, 5 = the 96-VGPR build of rounds 2-4.
RBA_KNOB(rba_k5_wpe, 6);

namespace {
template <int NT, int WAVES>
int launch_h3(const float* qkv, const float* qkv_bias, const float* bias, const float* bias_frag, float* out, int B, int H, int W,
              int Hp, int Wp, int nH, int ws, int shift, float scale, hipStream_t st, bool split_out = false) {
  const size_t shm = (size_t)(4 * NT * 16 * 64) + (size_t)(2 * NT * 16) * sizeof(int);
  const dim3 grid(Wp / ws, Hp / ws, B * nH), block(64 * WAVES);
#define K5H_LAUNCH(FRAG, SOUT, WPE, BIAS) hipLaunchKernelGGL((swin_window_attn_h3_kernel<NT, WAVES, FRAG, SOUT, WPE>), grid, block, shm, st, qkv, qkv_bias, BIAS, out, H, W, Hp, Wp, nH, ws, shift, scale)
  const int w = rba_k5_wpe;
  if (split_out && !bias_frag) return (int)hipErrorInvalidValue;
  if (split_out) {
    if (w == 5) K5H_LAUNCH(true, true, 5, bias_frag); else if (w == 7) K5H_LAUNCH(true, true, 7, bias_frag); else K5H_LAUNCH(true, true, 6, bias_frag);
  } else if (bias_frag) {
    if (w == 5) K5H_LAUNCH(true, false, 5, bias_frag); else if (w == 7) K5H_LAUNCH(true, false, 7, bias_frag); else K5H_LAUNCH(true, false, 6, bias_frag);
  } else {
    if (w == 5) K5H_LAUNCH(false, false, 5, bias); else if (w == 7) K5H_LAUNCH(false, false, 7, bias); else K5H_LAUNCH(false, false, 6, bias);
  }
#undef K5H_LAUNCH
  return rba_launch_status();
}

}  // namespace

extern "C" int64_t rba_swin_bias_fragments_elems(int nH, int ws) {
  if (nH <= 0 || ws <= 0 || ws * ws > 256) return 0;
  const int64_t NT = (ws * ws + 15) / 16;
  return (int64_t)nH * NT * NT * 256;
}

extern "C" int rba_swin_bias_fragments_f32(const float* bias, float* frag, int nH, int ws, void* stream) {
  RBA_CHECK_ARG(bias && frag && nH >= 1 && ws >= 1 && ws * ws <= 256);
  rba_begin();
  const int N = ws * ws, NT = (N + 15) / 16;
  const int64_t total = (int64_t)nH * NT * NT * 256;
  hipLaunchKernelGGL(swin_bias_fragments_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bias,
                     frag, nH, N, NT);
  return rba_launch_status();
}

extern "C" int rba_swin_window_attn_f32(const float* qkv, const float* qkv_bias, const float* bias, const float* bias_frag,
                                        float* out, int B, int H, int W, int nH, int hd, int ws, int shift, void* stream) {
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
  if (hd == 32) {                                   // matrix-pipe path for the window sizes in use
    const int NT = (N + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    if (NT == 9) return launch_h3<9, 9>(qkv, qkv_bias, bias, bias_frag, out, B, H, W, Hp, Wp, nH, ws, shift, scale, st);
    if (NT == 4) return launch_mfma<4, 4>(qkv, qkv_bias, bias, bias_frag, out, B, H, W, Hp, Wp, nH, ws, shift, scale, st);
    if (NT == 3) return launch_mfma<3, 3>(qkv, qkv_bias, bias, bias_frag, out, B, H, W, Hp, Wp, nH, ws, shift, scale, st);
  }
  const dim3 grid(Wp / ws, Hp / ws, B * nH);
#define RBA_L(D) hipLaunchKernelGGL(swin_window_attn_kernel<D>, grid, dim3(threads), shm, (hipStream_t)stream, qkv, qkv_bias, bias, out, H, W, Hp, Wp, nH, ws, shift, scale)
  if (hd == 16) RBA_L(16);
  else if (hd == 32) RBA_L(32);
  else RBA_L(64);
#undef RBA_L
  return rba_launch_status();
}

// The same attention with the output written as the proj Linear's split A operand (rba_split_linear_f16x3_frag_f32): image of
// [B * H * W, nH * 32] rows (ceil(rows / 32) * 32 * C * 4 bytes).  Only the f16x3 matrix-pipe form has this epilogue: head_dim 32,
// 12 x 12 windows, bias_frag required (hipErrorInvalidValue otherwise: the caller keeps the fp32 entry point for other geometries).
extern "C" int rba_swin_window_attn_split_out_f32(const float* qkv, const float* qkv_bias, const float* bias_frag, void* out_frag, int B,
                                                  int H, int W, int nH, int hd, int ws, int shift, void* stream) {
  RBA_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && nH >= 1 && hd == 32 && ws == 12 && shift >= 0 && shift < ws);
  if (B == 0) return 0;
  RBA_CHECK_ARG(qkv && qkv_bias && bias_frag && out_frag);
  RBA_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)qkv_bias | (uintptr_t)out_frag) & 15) == 0);
  const int Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws;
  RBA_CHECK_ARG((int64_t)B * nH <= 65535 && Hp / ws <= 65535);
  rba_begin();
  const float scale = (float)(1.0 / sqrt((double)hd));
  return launch_h3<9, 9>(qkv, qkv_bias, nullptr, bias_frag, reinterpret_cast<float*>(out_frag), B, H, W, Hp, Wp, nH, ws, shift, scale,
                         (hipStream_t)stream, true);
}
