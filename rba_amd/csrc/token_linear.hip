// Row-complete token Linear for the SMALL Linears of the path (round 4): the MSDeformAttn encoder's value / sampling / output
// projections and linear2 (pixel_decoder/msdeformattn.py:101-140, ops/modules/ms_deform_attn.py:95-121), the decoder's key / value
// projections of the memory (transformer_decoder/mask2former_transformer_decoder.py:83-143), the 1x1 input projection.
// 2 048 ... 4 830 token rows, N = 96 ... 256 outputs, K = 256 ... 1 024: fewer than 64 tiles of 128 x 128, which K6 does not serve; until
// round 3 these were 27 hipBLASLt launches per image plus the element-wise kernels between them.
//
// One workgroup (4 waves) owns 16 token rows and ALL N columns, so everything that needs a complete row happens in its epilogue:
//   out = [LayerNorm(] residual + act(A W^T + bias) [)],   A = x (+ x_add: the `src + pos` of msdeformattn.py:133, added as the rows are loaded).
// Arithmetic = K6's f16x3 (split_linear_h3.h): every fp32 value is h + 2^-11 l with two f16 numbers, three v_mfma_f32_16x16x32_f16
// per product (h.h into a main accumulator; h.l and l.h into a low one, added with weight 2^-11), fp32-GEMM accuracy for |v| < 65504.
// The MFMAs run with the WEIGHT fragment as the A operand (D^T = W x^T): a lane then holds four consecutive output channels of one
// token row -- 16-byte stores, and a row's LayerNorm statistics are a sum over a lane's own registers, two xor-shuffles (16, 32) and
// one LDS exchange between the four waves.
// Weights: packed once per weight load (rba_token_linear_pack_f16x2) in fragment order, [N/16][K/32][h | l][lane][8 f16]: lane
// (n = lane % 16, kb = lane / 16) holds W[16 nt + n][32 b + 8 kb + 0..7] -- one coalesced 1 KiB wave load per fragment straight from
// L2 into registers, no LDS (every workgroup streams the whole weight: 256 KiB at K = N = 256, which is what bounds the kernel -- the
// launch is a few microseconds of L2 -> register traffic, not matrix work).
// Two problems that share their rows (value = Linear_v(src), raw = Linear_s(src + pos); k = Linear_k(memory + pos), v = Linear_v(memory))
// run as ONE launch: blockIdx.y selects the problem.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

typedef _Float16 tl_f16x8 __attribute__((ext_vector_type(8)));
typedef float tl_f32x4 __attribute__((ext_vector_type(4)));

struct TlProblem {
  const float* x_add;        // [M, K] or null
  const rba_u32x4* wp;       // packed weight
  const float* bias;         // [N] or null
  float* out;                // row m at out + m * ld
  int N;
  int ld;                    // row stride of out in floats (>= N): a problem may fill a column slice of a wider tensor
  int act;                   // 0 none, 2 ReLU
};
struct TlProblems {
  TlProblem p[3];
};

// one thread per 16-byte unit: unit = ((nt * KB + b) * 2 + plane) * 64 + lane
__global__ void token_linear_pack_kernel(const float* __restrict__ w, rba_u32x4* __restrict__ packed, int N, int K) {
  const int KB = K >> 5;
  const int64_t total = (int64_t)((N + 15) >> 4) * KB * 128;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), plane = (int)((i >> 6) & 1);
    const int64_t tb = i >> 7;
    const int b = (int)(tb % KB), nt = (int)(tb / KB);
    const int n = 16 * nt + (lane & 15), k0 = 32 * b + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = n < N ? w[(int64_t)n * K + k0 + j] : 0.f;
    rba_u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t h, l;
      rba_split_f16x2(v[2 * j], v[2 * j + 1], h, l);
      o[j] = plane ? l : h;
    }
    packed[i] = o;
  }
}

// CTW: 16-column tiles per wave (N <= 16 CTW TLW).  LN: residual + LayerNorm epilogue (single problem).  RT: 16-row token tiles per workgroup
// (1 or 2; with 2 every weight fragment a wave loads feeds two row tiles: half the workgroups, half the weight bytes pulled from L2).
constexpr int TLW = 4;
template <int CTW, bool LN, int RT = 1>
__global__ __launch_bounds__(64 * TLW) void token_linear_kernel(const float* __restrict__ x, TlProblems ps, const float* __restrict__ residual,
                                                           const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps, int M, int K) {
  const TlProblem p = blockIdx.y == 0 ? ps.p[0] : (blockIdx.y == 1 ? ps.p[1] : ps.p[2]);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = lane & 15, kb = lane >> 4;
  const int row0 = blockIdx.x * (16 * RT);
  int row[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) row[rt] = min(row0 + 16 * rt + t, M - 1);
  const int KB = K >> 5, NT = (p.N + 15) >> 4;
  const int nt0 = wave * CTW;                                                     // this wave's first column tile
  const bool has_add = p.x_add != nullptr;                                        // workgroup-uniform
  const rba_u32x4* wl = p.wp + lane;

  tl_f32x4 accm[RT][CTW], accl[RT][CTW];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < CTW; ++i) accm[rt][i] = accl[rt][i] = (tl_f32x4){0.f, 0.f, 0.f, 0.f};

  // NS-deep register ring: the loads of block b + NS - 1 are issued before block b is consumed.  Every workgroup streams the whole packed
  // weight (256 KiB at K = N = 256, 1 MiB at K = 1 024) and all workgroups walk the same lines at the same time.  Measured at 2 048 rows
  // (profiles/r04_token_linear.txt; two-problem launch / K = 256 + LayerNorm / K = 1 024 + LayerNorm): one block of look-ahead 12 / 18 (mean of
  // both LayerNorm forms) us; FOUR stages 10.6 / 9.6 / 20.4 us; eight waves x six stages (192 KB in flight per workgroup)
  // 11.3 / 9.6 / 25.6 us -- more bytes in flight do not help, so the launch is not latency-bound any more but bound by the L2 serving the
  // same weight lines to 128 workgroups at once; cutting the columns into groups (4x the workgroups, a quarter of the weight each) is
  // slower still (23 us at K = 1 024: three MFMAs per 2 KB of loads).  At K = 1 024 the library GEMM + LayerNorm pair it replaces took 16 us.
  constexpr int NS = RT == 2 ? 3 : 4;
  f32x4 a[NS][RT][2], ad[NS][RT][2];
  rba_u32x4 wf[NS][CTW][2];
  auto loads = [&](int b, int s) {
    if (b >= KB) return;                                                            // wave-uniform
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float* xa = x + (int64_t)row[rt] * K + 8 * kb + 32 * b;
      a[s][rt][0] = *reinterpret_cast<const f32x4*>(xa);
      a[s][rt][1] = *reinterpret_cast<const f32x4*>(xa + 4);
      if (has_add) {
        const float* xb = p.x_add + (int64_t)row[rt] * K + 8 * kb + 32 * b;
        ad[s][rt][0] = *reinterpret_cast<const f32x4*>(xb);
        ad[s][rt][1] = *reinterpret_cast<const f32x4*>(xb + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < CTW; ++i) {
      const int nt = min(nt0 + i, NT - 1);                                        // tiles beyond N: a valid address, result unused
      const rba_u32x4* src = wl + ((int64_t)nt * KB + b) * 128;
      wf[s][i][0] = src[0];
      wf[s][i][1] = src[64];
    }
  };
  auto compute = [&](int s) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      f32x4 u = a[s][rt][0], v = a[s][rt][1];
      if (has_add) {
        u = u + ad[s][rt][0];
        v = v + ad[s][rt][1];
      }
      rba_u32x4 hp, lp;
      uint32_t h_, l_;
      rba_split_f16x2(u.x, u.y, h_, l_); hp.x = h_; lp.x = l_;
      rba_split_f16x2(u.z, u.w, h_, l_); hp.y = h_; lp.y = l_;
      rba_split_f16x2(v.x, v.y, h_, l_); hp.z = h_; lp.z = l_;
      rba_split_f16x2(v.z, v.w, h_, l_); hp.w = h_; lp.w = l_;
      const tl_f16x8 xh = __builtin_bit_cast(tl_f16x8, hp), xl = __builtin_bit_cast(tl_f16x8, lp);
#pragma unroll
      for (int i = 0; i < CTW; ++i) {
        const tl_f16x8 wh = __builtin_bit_cast(tl_f16x8, wf[s][i][0]);
        accm[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, accm[rt][i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < CTW; ++i) {
        const tl_f16x8 wh = __builtin_bit_cast(tl_f16x8, wf[s][i][0]), wlo = __builtin_bit_cast(tl_f16x8, wf[s][i][1]);
        accl[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, xh, accl[rt][i], 0, 0, 0);
        accl[rt][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, accl[rt][i], 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) loads(s, s);
  for (int b = 0; b < KB; b += NS) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      loads(b + s + NS - 1, (s + NS - 1) % NS);
      if (b + s < KB) compute(s);                                                  // wave-uniform
    }
  }

  // lane: token rows row0 + 16 rt + t, channels n = 16 (nt0 + i) + 4 kb + r
  float val[RT][CTW][4];
#pragma unroll
  for (int i = 0; i < CTW; ++i) {
    const int n = 16 * (nt0 + i) + 4 * kb;
    const bool ok = nt0 + i < NT;                                                  // wave-uniform
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (ok && p.bias) {
      if (n + 3 < p.N) {
        bv = *reinterpret_cast<const f32x4*>(p.bias + n);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) bv[r] = p.bias[n + r];
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = fmaf(accl[rt][i][r], 0.00048828125f, accm[rt][i][r]) + bv[r];
        v = rba_clamp_below(v, rba_relu_floor(p.act == 2));
        val[rt][i][r] = ok ? v : 0.f;
      }
  }
  if (!LN) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool rowok = row0 + 16 * rt + t < M;
#pragma unroll
      for (int i = 0; i < CTW; ++i) {
        const int n = 16 * (nt0 + i) + 4 * kb;
        if (rowok && nt0 + i < NT) {
          if (n + 3 < p.N) {
            *reinterpret_cast<f32x4*>(p.out + (int64_t)row[rt] * p.ld + n) = (f32x4){val[rt][i][0], val[rt][i][1], val[rt][i][2], val[rt][i][3]};
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) p.out[(int64_t)row[rt] * p.ld + n + r] = val[rt][i][r];
          }
        }
      }
    }
    return;
  }
  // ---- residual + LayerNorm over the complete row (N % 16 == 0): s = residual + (x W^T + bias), y = (s - mean) rstd g + b, the
  // statistics as in rba_add_layer_norm_f32 (mean first, then the centred sum of squares)
  __shared__ float red[2][TLW][RT][16];
  float s1[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    s1[rt] = 0.f;
#pragma unroll
    for (int i = 0; i < CTW; ++i) {
      if (nt0 + i < NT) {
        const int n = 16 * (nt0 + i) + 4 * kb;
        const f32x4 rv = *reinterpret_cast<const f32x4*>(residual + (int64_t)row[rt] * p.N + n);          // LN form: ld == N
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          val[rt][i][r] = rv[r] + val[rt][i][r];
          s1[rt] += val[rt][i][r];
        }
      }
    }
    s1[rt] += __shfl_xor(s1[rt], 16, RBA_WAVE);
    s1[rt] += __shfl_xor(s1[rt], 32, RBA_WAVE);
    if (kb == 0) red[0][wave][rt][t] = s1[rt];
  }
  __syncthreads();
  float mean[RT], s2[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float tot = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < TLW; ++w_) tot += red[0][w_][rt][t];
    mean[rt] = tot / (float)p.N;
    s2[rt] = 0.f;
#pragma unroll
    for (int i = 0; i < CTW; ++i) {
      if (nt0 + i < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = val[rt][i][r] - mean[rt];
          s2[rt] = fmaf(d, d, s2[rt]);
        }
      }
    }
    s2[rt] += __shfl_xor(s2[rt], 16, RBA_WAVE);
    s2[rt] += __shfl_xor(s2[rt], 32, RBA_WAVE);
    if (kb == 0) red[1][wave][rt][t] = s2[rt];
  }
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float tot = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < TLW; ++w_) tot += red[1][w_][rt][t];
    const float rstd = rsqrtf(tot / (float)p.N + eps);
    const bool rowok = row0 + 16 * rt + t < M;
#pragma unroll
    for (int i = 0; i < CTW; ++i) {
      if (rowok && nt0 + i < NT) {
        const int n = 16 * (nt0 + i) + 4 * kb;
        const f32x4 g = *reinterpret_cast<const f32x4*>(ln_w + n), be = *reinterpret_cast<const f32x4*>(ln_b + n);
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (val[rt][i][r] - mean[rt]) * rstd * g[r] + be[r];
        *reinterpret_cast<f32x4*>(p.out + (int64_t)row[rt] * p.N + n) = y;
      }
    }
  }
}

// tools / tests: 0 = product rule (two row tiles per workgroup for K >= 512 and at least 4 096 rows), 1 = always one row tile, 2 = always two
RBA_KNOB(rba_token_rt, 0);

template <bool LN>
int launch_token_linear(const float* x, const TlProblems& ps, int nprob, const float* residual, const float* ln_w, const float* ln_b, float eps,
                        int M, int K, hipStream_t st) {
  int nmax = 0;
  for (int i = 0; i < nprob; ++i) nmax = ps.p[i].N > nmax ? ps.p[i].N : nmax;
  const int ctw = (((nmax + 15) >> 4) + TLW - 1) / TLW;
  // two row tiles where that still leaves at least 128 workgroups: at 2 048 rows it does not (64 workgroups: 28.6 us against 20.4 with one tile),
  // at 4 830 rows it does (40.0 -> 30.9 us; profiles/r04_token_linear.txt)
  const bool two = rba_token_rt == 2 || (rba_token_rt == 0 && K >= 512 && M >= 4096);
  const dim3 grid((unsigned)((M + (two ? 31 : 15)) / (two ? 32 : 16)), (unsigned)nprob);
#define RBA_TL(C)                                                                                                                     \
  if (two) hipLaunchKernelGGL((token_linear_kernel<C, LN, 2>), grid, dim3(64 * TLW), 0, st, x, ps, residual, ln_w, ln_b, eps, M, K);   \
  else hipLaunchKernelGGL((token_linear_kernel<C, LN, 1>), grid, dim3(64 * TLW), 0, st, x, ps, residual, ln_w, ln_b, eps, M, K)
  switch (ctw) {
    case 1: RBA_TL(1); break;
    case 2: RBA_TL(2); break;
    case 3: RBA_TL(3); break;
    case 4: RBA_TL(4); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef RBA_TL
  return rba_launch_status();
}

inline bool tl_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int rba_token_linear_pack_f16x2(const float* weight, void* packed, int N, int K, void* stream) {
  RBA_CHECK_ARG(weight && packed && N >= 1 && K >= 32 && K % 32 == 0 && tl_aligned(packed));
  rba_begin();
  const int64_t total = (int64_t)((N + 15) >> 4) * (K >> 5) * 128;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(token_linear_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, weight, reinterpret_cast<rba_u32x4*>(packed), N, K);
  return rba_launch_status();
}

int rba_token_linear_f32(const float* x, const float* x_add, const void* packed, const float* bias, const float* residual,
                         const float* ln_weight, const float* ln_bias, float ln_eps, float* out, int64_t M, int N, int K, int act,
                         void* stream) {
  RBA_CHECK_ARG(x && packed && out && M >= 0 && M < ((int64_t)1 << 31) - 16 && N >= 1 && N <= 256 && K >= 32 && K % 32 == 0);
  RBA_CHECK_ARG(act == 0 || act == 2);
  RBA_CHECK_ARG(tl_aligned(x) && tl_aligned(x_add) && tl_aligned(packed) && tl_aligned(out) && tl_aligned(bias) && N % 4 == 0);
  const bool ln = ln_weight != nullptr;
  RBA_CHECK_ARG(!ln || (residual && ln_bias && N % 16 == 0 && act == 0 && tl_aligned(residual) && tl_aligned(ln_weight) && tl_aligned(ln_bias)));
  RBA_CHECK_ARG(ln || !residual);
  if (M == 0) return 0;
  rba_begin();
  TlProblems ps{};
  ps.p[0] = TlProblem{x_add, reinterpret_cast<const rba_u32x4*>(packed), bias, out, N, N, act};
  if (ln) return launch_token_linear<true>(x, ps, 1, residual, ln_weight, ln_bias, ln_eps, (int)M, K, (hipStream_t)stream);
  return launch_token_linear<false>(x, ps, 1, nullptr, nullptr, nullptr, 0.f, (int)M, K, (hipStream_t)stream);
}

int rba_token_linear_multi_f32(const float* x, const rba_token_linear_problem* problems, int n_problems, int64_t M, int K, void* stream) {
  RBA_CHECK_ARG(x && problems && n_problems >= 1 && n_problems <= 3 && M >= 0 && M < ((int64_t)1 << 31) - 16 && K >= 32 && K % 32 == 0 && tl_aligned(x));
  TlProblems ps{};
  for (int i = 0; i < n_problems; ++i) {
    const rba_token_linear_problem& q = problems[i];
    RBA_CHECK_ARG(q.packed && q.out && q.N >= 1 && q.N <= 256 && q.N % 4 == 0 && q.ld_out >= q.N && q.ld_out % 4 == 0 && (q.act == 0 || q.act == 2));
    RBA_CHECK_ARG(tl_aligned(q.x_add) && tl_aligned(q.packed) && tl_aligned(q.bias) && tl_aligned(q.out));
    ps.p[i] = TlProblem{q.x_add, reinterpret_cast<const rba_u32x4*>(q.packed), q.bias, q.out, q.N, q.ld_out, q.act};
  }
  if (M == 0) return 0;
  rba_begin();
  return launch_token_linear<false>(x, ps, n_problems, nullptr, nullptr, nullptr, 0.f, (int)M, K, (hipStream_t)stream);
}

}  // extern "C"
