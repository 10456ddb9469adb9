// K3 -- masked multi-head cross attention core of the transformer decoder (reference:
// mask2former_transformer_decoder.py:106-118 nn.MultiheadAttention with bool attn_mask, the all-masked-row
// fix :433, and the threshold sigmoid(mask) < 0.5 of :483-487 which is fused here).
//
// 100 queries x 8 heads against up to ~15k keys, head_dim 32: tiny FLOPs, latency/launch bound.  One
// 256-thread workgroup per (batch, head, query); thread t walks keys t, t+256, ... with a private online
// softmax state (m, l, acc[32]); blocked keys cost one mask load and nothing else; the 256 states are merged
// by wave shuffles and a 4-entry LDS exchange.  K/V rows are 128 B contiguous per (key, head).
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

constexpr int HD = 32;

__device__ __forceinline__ void merge_state(float& m, float& l, float (&acc)[HD], float m2, float l2, const float (&acc2)[HD]) {
  const float mn = fmaxf(m, m2);
  const float a = (m == -INFINITY) ? 0.f : __expf(m - mn);
  const float b = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
  l = l * a + l2 * b;
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = acc[d] * a + acc2[d] * b;
  m = mn;
}

__global__ __launch_bounds__(256) void masked_xattn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ mlog,
                                                           float* __restrict__ out, int Q, int S, int nH) {
  const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float sh_any[4];
  __shared__ float sh_m[4], sh_l[4], sh_acc[4][HD];

  const float* mrow = mlog ? mlog + ((int64_t)b * Q + qi) * S : nullptr;
  // does the row have any un-blocked key?  (blocked iff sigmoid(x) < 0.5)
  bool use_mask = false;
  if (mrow) {
    float any = 0.f;
    for (int s = tid; s < S; s += 256) any = fmaxf(any, rba_sigmoid(mrow[s]) < 0.5f ? 0.f : 1.f);
    any = wave_reduce_max(any);
    if (lane == 0) sh_any[wave] = any;
    __syncthreads();
    use_mask = fmaxf(fmaxf(sh_any[0], sh_any[1]), fmaxf(sh_any[2], sh_any[3])) > 0.f;
  }

  const float scale = 0.17677669529663687f;  // 32^-0.5
  float qv[HD];
  const float* qp = q + (((int64_t)b * Q + qi) * nH + h) * HD;   // wave-uniform -> scalar loads
#pragma unroll
  for (int d = 0; d < HD; ++d) qv[d] = qp[d] * scale;

  float m = -INFINITY, l = 0.f, acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;

  const int64_t rs = (int64_t)nH * HD;
  const float* kb = k + (int64_t)b * S * rs + h * HD;
  const float* vb = v + (int64_t)b * S * rs + h * HD;
  for (int s = tid; s < S; s += 256) {
    if (use_mask && rba_sigmoid(mrow[s]) < 0.5f) continue;
    const float4* kr = reinterpret_cast<const float4*>(kb + s * rs);
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
      const float4 kk = kr[i];
      sc = fmaf(qv[4 * i], kk.x, sc); sc = fmaf(qv[4 * i + 1], kk.y, sc);
      sc = fmaf(qv[4 * i + 2], kk.z, sc); sc = fmaf(qv[4 * i + 3], kk.w, sc);
    }
    if (sc > m) {
      const float a = __expf(m - sc);   // m = -inf -> 0
      l *= a;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] *= a;
      m = sc;
    }
    const float p = __expf(sc - m);
    l += p;
    const float4* vr = reinterpret_cast<const float4*>(vb + s * rs);
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
      const float4 vv = vr[i];
      acc[4 * i] = fmaf(p, vv.x, acc[4 * i]); acc[4 * i + 1] = fmaf(p, vv.y, acc[4 * i + 1]);
      acc[4 * i + 2] = fmaf(p, vv.z, acc[4 * i + 2]); acc[4 * i + 3] = fmaf(p, vv.w, acc[4 * i + 3]);
    }
  }
  // merge across the wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, RBA_WAVE), l2 = __shfl_xor(l, o, RBA_WAVE);
    float acc2[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc2[d] = __shfl_xor(acc[d], o, RBA_WAVE);
    merge_state(m, l, acc, m2, l2, acc2);
  }
  if (lane == 0) {
    sh_m[wave] = m; sh_l[wave] = l;
#pragma unroll
    for (int d = 0; d < HD; ++d) sh_acc[wave][d] = acc[d];
  }
  __syncthreads();
  if (tid < HD) {
    float M = fmaxf(fmaxf(sh_m[0], sh_m[1]), fmaxf(sh_m[2], sh_m[3]));
    float L = 0.f, A = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = sh_m[w] == -INFINITY ? 0.f : __expf(sh_m[w] - M);
      L += sh_l[w] * f;
      A += sh_acc[w][tid] * f;
    }
    out[(((int64_t)b * Q + qi) * nH + h) * HD + tid] = A / L;
  }
}


// ------------------------------------------------------------------------------------------------------------
// v2: split-key flash attention on the matrix pipe.  v1 above lets every (query, head) workgroup re-read all K/V rows
// of its head: 100 x 8 x S x 256 B of L2 traffic per call (3 GB at S = 14720, 98 us/call in the 9-layer decoder).
// Here one workgroup owns (key range, head): a chunk of 144 keys (K and V, 2 x 144 x 128 B) is staged ONCE in LDS and
// consumed by all query strips (16 queries per wave, 7 waves for Q = 100), with the K5 dataflow: S^T = K.Q^T on
// v_mfma_f32_16x16x4_f32 so the softmax axis is in-lane and the probability registers are the A operand of P.V.
// Key ranges are reduced with the usual (m, l, O) merge.  Three launches: row flags (all-masked rule of decoder.py:433),
// partial attention, merge.
typedef float f32x4_x __attribute__((ext_vector_type(4)));
constexpr int XC = 144, XNT = 9, XRS = 36;      // keys per chunk, 16-key tiles per chunk, LDS row stride (floats)

__global__ __launch_bounds__(256) void xattn_rowflag_kernel(const float* __restrict__ mlog, int* __restrict__ flag, int S) {
  const float* row = mlog + (int64_t)blockIdx.x * S;          // blockIdx.x = b*Q + q
  float any = 0.f;
  for (int s = threadIdx.x; s < S; s += 256) any = fmaxf(any, rba_sigmoid(row[s]) < 0.5f ? 0.f : 1.f);
  any = wave_reduce_max(any);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = any;
  __syncthreads();
  if (threadIdx.x == 0) flag[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) > 0.f ? 1 : 0;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void xattn_partial_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                        const float* __restrict__ v, const float* __restrict__ mlog,
                                                                        const int* __restrict__ flag, float* __restrict__ ws,
                                                                        int Q, int S, int nH, int chunks_per_split, int splits) {
  __shared__ __attribute__((aligned(16))) float Ks[XC * XRS];
  __shared__ __attribute__((aligned(16))) float Vs[XC * XRS];
  const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kk = lane >> 4;
  const int64_t rs = (int64_t)nH * HD;
  const float* kb = k + (int64_t)b * S * rs + h * HD;
  const float* vb = v + (int64_t)b * S * rs + h * HD;
  const int qt = wave * 16 + l15;                             // this lane's query (softmax owner / B column)
  const bool qvalid = qt < Q;
  const float scale = 0.17677669529663687f;                   // 32^-0.5
  float q8[8];
  {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c4 = a;
    if (qvalid) {
      const float4* p = reinterpret_cast<const float4*>(q + (((int64_t)b * Q + qt) * nH + h) * HD + kk * 8);
      a = p[0]; c4 = p[1];
    }
    q8[0] = a.x * scale; q8[1] = a.y * scale; q8[2] = a.z * scale; q8[3] = a.w * scale;
    q8[4] = c4.x * scale; q8[5] = c4.y * scale; q8[6] = c4.z * scale; q8[7] = c4.w * scale;
  }
  const bool use_mask = mlog != nullptr && qvalid && flag[b * Q + qt] != 0;
  const float* mrow = mlog ? mlog + ((int64_t)b * Q + (qvalid ? qt : 0)) * S : nullptr;
  const bool vec_mask = (S & 3) == 0;

  float m_run = -INFINITY, l_run = 0.f;
  f32x4_x O0 = {0.f, 0.f, 0.f, 0.f}, O1 = O0;

  for (int ch = 0; ch < chunks_per_split; ++ch) {
    const int kbase = (split * chunks_per_split + ch) * XC;
    if (kbase >= S) break;                                    // uniform
    __syncthreads();                                          // previous chunk fully consumed
    for (int i = threadIdx.x; i < XC * (HD / 4); i += 64 * WAVES) {
      const int t = i >> 3, d4 = i & 7;
      float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f), v4 = k4;
      if (kbase + t < S) {
        k4 = *reinterpret_cast<const float4*>(kb + (int64_t)(kbase + t) * rs + d4 * 4);
        v4 = *reinterpret_cast<const float4*>(vb + (int64_t)(kbase + t) * rs + d4 * 4);
      }
      *reinterpret_cast<float4*>(Ks + t * XRS + d4 * 4) = k4;
      *reinterpret_cast<float4*>(Vs + t * XRS + d4 * 4) = v4;
    }
    __syncthreads();
    // ---- S^T tiles: lane holds S[key = kbase + c*16 + 4*kk + r][query = qt]
    f32x4_x Sx[XNT];
#pragma unroll
    for (int c = 0; c < XNT; c += 2) {
      const float4* kr = reinterpret_cast<const float4*>(Ks + (c * 16 + l15) * XRS + kk * 8);
      const float4* kr2 = reinterpret_cast<const float4*>(Ks + ((c + 1 < XNT ? c + 1 : c) * 16 + l15) * XRS + kk * 8);
      const float4 k0 = kr[0], k1 = kr[1], j0 = kr2[0], j1 = kr2[1];
      const float ka[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
      const float kc[8] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w};
      f32x4_x a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[st], q8[st], a0, 0, 0, 0);
        if (c + 1 < XNT) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[st], q8[st], a1, 0, 0, 0);
      }
      Sx[c] = a0;
      if (c + 1 < XNT) Sx[c + 1] = a1;
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- mask (blocked iff sigmoid(mask) < 0.5), keys beyond S, chunk max
    float cmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < XNT; ++c) {
      const int k0i = kbase + c * 16 + kk * 4;
      float mv[4] = {1.f, 1.f, 1.f, 1.f};                    // any value with sigmoid >= 0.5
      if (use_mask) {
        if (vec_mask && k0i + 3 < S) {
          const float4 t4 = *reinterpret_cast<const float4*>(mrow + k0i);
          mv[0] = t4.x; mv[1] = t4.y; mv[2] = t4.z; mv[3] = t4.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (k0i + r < S) mv[r] = mrow[k0i + r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sv = Sx[c][r];
        if (k0i + r >= S || (use_mask && rba_sigmoid(mv[r]) < 0.5f)) sv = -INFINITY;
        Sx[c][r] = sv;
        cmax = fmaxf(cmax, sv);
      }
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 16, RBA_WAVE));
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32, RBA_WAVE));
    const float m_new = fmaxf(m_run, cmax);
    const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);      // m_new == -inf only if m_run == -inf
    float psum = 0.f;
#pragma unroll
    for (int c = 0; c < XNT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (Sx[c][r] == -INFINITY) ? 0.f : __expf(Sx[c][r] - m_new);
        Sx[c][r] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 16, RBA_WAVE);
    psum += __shfl_xor(psum, 32, RBA_WAVE);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // ---- rescale O (rows = queries 4*kk + r of this strip) and accumulate P.V
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ar = __shfl(alpha, kk * 4 + r, RBA_WAVE);
      O0[r] *= ar;
      O1[r] *= ar;
    }
#pragma unroll
    for (int c = 0; c < XNT; ++c) {
      const float* vr = Vs + (c * 16 + kk * 4) * XRS + l15;
      float v0[4], v1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v0[r] = vr[r * XRS]; v1[r] = vr[r * XRS + 16]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        O0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Sx[c][r], v0[r], O0, 0, 0, 0);
        O1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Sx[c][r], v1[r], O1, 0, 0, 0);
      }
      if (c & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- partial result: ws[b][h][split][q][34] = (m, l, O[32]); lane holds O[query = 16*wave + 4*kk + r][d = l15 (+16)]
  float* wb = ws + ((((int64_t)b * nH + h) * splits + split) * Q) * 34;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = wave * 16 + kk * 4 + r;
    const float mr = __shfl(m_run, kk * 4 + r, RBA_WAVE), lr = __shfl(l_run, kk * 4 + r, RBA_WAVE);
    if (qi < Q) {
      float* o = wb + (int64_t)qi * 34;
      if (l15 == 0) { o[0] = mr; o[1] = lr; }
      o[2 + l15] = O0[r];
      o[18 + l15] = O1[r];
    }
  }
}

__global__ void xattn_merge_kernel(const float* __restrict__ ws, float* __restrict__ out, int Q, int nH, int splits) {
  const int qi = blockIdx.x, b = blockIdx.y;
  const int h = threadIdx.x >> 5, d = threadIdx.x & 31;
  if (h >= nH) return;
  const float* wb = ws + ((((int64_t)b * nH + h) * splits) * Q + qi) * 34;
  const int64_t sstride = (int64_t)Q * 34;
  float M = -INFINITY;
  for (int s = 0; s < splits; ++s) M = fmaxf(M, wb[s * sstride]);
  float L = 0.f, A = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float ms = wb[s * sstride];
    const float f = ms == -INFINITY ? 0.f : __expf(ms - M);
    L += wb[s * sstride + 1] * f;
    A += wb[s * sstride + 2 + d] * f;
  }
  out[(((int64_t)b * Q + qi) * nH + h) * HD + d] = A / L;
}

struct XPlan { int chunks_per_split, splits; };
static inline XPlan xattn_plan(int B, int S, int nH) {
  const int nchunks = (S + XC - 1) / XC;
  int target = 512 / (nH * B);                                  // aim for ~2 workgroups per CU
  if (target < 1) target = 1;
  int cps = (nchunks + target - 1) / target;
  if (cps < 1) cps = 1;
  return XPlan{cps, (nchunks + cps - 1) / cps};
}

}  // namespace

extern "C" int64_t rba_masked_xattn_workspace_bytes(int B, int Q, int S, int nH) {
  if (B <= 0 || Q <= 0 || S <= 0 || nH <= 0) return 0;
  const XPlan p = xattn_plan(B, S, nH);
  return ((int64_t)B * nH * p.splits * Q * 34 + (int64_t)B * Q) * 4;
}

extern "C" int rba_masked_xattn_f32(const float* q, const float* k, const float* v, const float* mask_logits, float* out,
                                    float* workspace, int B, int Q, int S, int nH, int hd, void* stream) {
  RBA_CHECK_ARG(B >= 0 && Q >= 0 && S >= 1 && nH >= 1 && hd == HD && nH <= 65535 && B <= 65535);
  if (B == 0 || Q == 0) return 0;
  RBA_CHECK_ARG(q && k && v && out && (((uintptr_t)k | (uintptr_t)v) & 15) == 0);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  // matrix-pipe path: needs the caller's workspace (rba_masked_xattn_workspace_bytes), Q <= 128 and 16 B aligned q
  if (workspace && Q <= 128 && nH <= 32 && (((uintptr_t)q | (uintptr_t)workspace) & 15) == 0) {
    const XPlan p = xattn_plan(B, S, nH);
    int* flag = reinterpret_cast<int*>(workspace + (int64_t)B * nH * p.splits * Q * 34);
    if (mask_logits) hipLaunchKernelGGL(xattn_rowflag_kernel, dim3(B * Q), dim3(256), 0, st, mask_logits, flag, S);
    const dim3 grid(p.splits, nH, B);
    const int strips = (Q + 15) / 16;
#define RBA_L(W) hipLaunchKernelGGL(xattn_partial_mfma_kernel<W>, grid, dim3(64 * W), 0, st, q, k, v, mask_logits, flag, workspace, Q, S, nH, p.chunks_per_split, p.splits)
    switch (strips) {
      case 1: RBA_L(1); break;
      case 2: RBA_L(2); break;
      case 3: RBA_L(3); break;
      case 4: RBA_L(4); break;
      case 5: RBA_L(5); break;
      case 6: RBA_L(6); break;
      case 7: RBA_L(7); break;
      default: RBA_L(8); break;
    }
#undef RBA_L
    hipLaunchKernelGGL(xattn_merge_kernel, dim3(Q, B), dim3(nH * 32), 0, st, workspace, out, Q, nH, p.splits);
    return rba_launch_status();
  }
  hipLaunchKernelGGL(masked_xattn_kernel, dim3(Q, nH, B), dim3(256), 0, st, q, k, v, mask_logits, out, Q, S, nH);
  return rba_launch_status();
}
