// Fused (residual add + bias) + LayerNorm over the last dimension (reference: the `x = shortcut + drop_path(x)` /
// `norm` pairs of backbone/swin.py:284-293, msdeformattn.py:134-138, mask2former_transformer_decoder.py:48-58,106-118,
// 171-175, all post-/pre-norm patterns of the form  s = x + (t + b);  y = LN(s)).
//
//   s[r,c]   = x[r,c] (+ t[r,c]) (+ tb[c])        written to sum_out if requested (may alias x)
//   y[r,c]   = (s - mean_r) * rstd_r * gamma[c] + beta[c]
//
// HBM-bound row kernel: a row lives in the registers of a G-lane group (16 B per lane per step), mean and centred
// variance are two in-register passes + xor-shuffles, so every element is read once and written once (twice with
// sum_out).  Replaces a separate bias-add pass in the GEMM epilogue, an elementwise add kernel and the LN kernel.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

// MERGE: the row is gathered, not contiguous -- PatchMerging's 2 x 2 neighbourhood (swin.py:311-337): output token (b, i, j) of the
// half-resolution grid = [x(2i, 2j) | x(2i+1, 2j) | x(2i, 2j+1) | x(2i+1, 2j+1)] (C = 4 Cin channels; positions outside an odd-sized map
// are zero, as the reference pads), read straight from the [B, H, W, Cin] token tensor: the concatenated tensor is never written.
struct MergeGeom {
  int H, W, Cin;           // source map; rows = B * ceil(H/2) * ceil(W/2), C = 4 Cin
};

// FRAG: y is not written row-major but as the split, fragment-ordered image the f16x3 GEMM reads as its A operand without any
// arithmetic (split_linear_h3.h, "PRE"): per (32-row group, 32-wide block of C) four 1 KiB pieces [h g0 | l g0 | h g1 | l g1], each
// [lh][row & 31][8 f16], k = 32 b + 16 lh + 8 g + i; h = f16(y), l = f16((y - h) 2^11).  A lane holds 4 consecutive channels; the even
// lane of a pair stores the 16-byte h piece, the odd lane the l piece (one DPP exchange): as many bytes and stores as the fp32 row.
// FRAG with WAVES = 8 (one row per wave, G = 64): the eight rows' pieces are transposed through LDS so that every global store
// instruction writes whole 128-byte lines of the image (8 consecutive rows x 16 bytes of one piece column) instead of eight scattered
// 16-byte fragments per row: [piece column][row & 7 XOR column & 7][16 B], conflict-free both ways.
template <int G, int NV, bool MERGE = false, bool FRAG = false, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES) void add_layer_norm_kernel(const float* x, const float* __restrict__ t,
                                                             const float* __restrict__ tb, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* sum_out /* may alias x */,
                                                             float* __restrict__ y, int64_t rows, int C, float eps,
                                                             MergeGeom mg = MergeGeom{0, 0, 0}) {
  constexpr int RPW = 64 / G;                              // rows per wave
  const int lane = threadIdx.x & 63, sub = lane % G, rsel = lane / G;
  const int64_t row = ((int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * RPW + rsel;
  constexpr bool TR = FRAG && WAVES == 8 && G == 64;                               // transposed stores
  __shared__ __attribute__((aligned(16))) rba_u32x4 tr_lds[TR ? NV * 64 * 8 : 1];
  const bool rvalid = row < rows;
  const int64_t base = (rvalid ? row : 0) * C;
  const int nv4 = C >> 2;
  int mi = 0, mj = 0;
  int64_t mb = 0;                                          // MERGE: this row's (batch offset in tokens, i, j)
  if (MERGE) {
    const int H2 = (mg.H + 1) >> 1, W2 = (mg.W + 1) >> 1;
    const int64_t r = rvalid ? row : 0;
    const int64_t bimg = r / ((int64_t)H2 * W2);
    const int rem = (int)(r - bimg * H2 * W2);
    mi = rem / W2;
    mj = rem - mi * W2;
    mb = bimg * mg.H * mg.W;
  }
  constexpr bool EARLY = NV <= 2;                          // hoisting costs 8 NV registers: measured faster for NV <= 2, slower above
  f32x4 v[NV], g4[EARLY ? NV : 1], b4[EARLY ? NV : 1];
  float sum = 0.f;
  // gamma / beta are requested together with the row, not after the two reductions that would otherwise wait for them
#pragma unroll
  for (int j = 0; j < (EARLY ? NV : 0); ++j) {
    const int c4 = sub + j * G;
    g4[j] = b4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c4 < nv4) {
      g4[j] = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
      b4[j] = *reinterpret_cast<const f32x4*>(beta + 4 * c4);
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c4 = sub + j * G;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c4 < nv4) {
      if (MERGE) {
        const int seg = (4 * c4) / mg.Cin, off = 4 * c4 - seg * mg.Cin;          // segment order (ee, oe, eo, oo): di = seg & 1, dj = seg >> 1
        const int si = 2 * mi + (seg & 1), sj = 2 * mj + (seg >> 1);
        if (si < mg.H && sj < mg.W) a = *reinterpret_cast<const f32x4*>(x + (mb + (int64_t)si * mg.W + sj) * mg.Cin + off);
      } else {
        a = *reinterpret_cast<const f32x4*>(x + base + 4 * c4);
      }
      if (t) a += *reinterpret_cast<const f32x4*>(t + base + 4 * c4);
      if (tb) a += *reinterpret_cast<const f32x4*>(tb + 4 * c4);
      sum += (a.x + a.y) + (a.z + a.w);
    }
    v[j] = a;
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, RBA_WAVE);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c4 = sub + j * G;
    if (c4 < nv4) {
      const f32x4 d = v[j] - mean;
      sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, RBA_WAVE);
  const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
  if (!TR && !rvalid) return;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c4 = sub + j * G;
    if (c4 < nv4) {
      if (sum_out && rvalid) *reinterpret_cast<f32x4*>(sum_out + base + 4 * c4) = v[j];
      const f32x4 gj = EARLY ? g4[EARLY ? j : 0] : *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
      const f32x4 bj = EARLY ? b4[EARLY ? j : 0] : *reinterpret_cast<const f32x4*>(beta + 4 * c4);
      const f32x4 o = (v[j] - mean) * rstd * gj + bj;
      if (FRAG) {
        uint32_t h[2], l[2];
        rba_split_f16x2(o.x, o.y, h[0], l[0]);
        rba_split_f16x2(o.z, o.w, h[1], l[1]);
        const bool odd = c4 & 1;
        const uint32_t s0 = odd ? h[0] : l[0], s1 = odd ? h[1] : l[1];             // what the partner lane stores
        const uint32_t r0 = __builtin_amdgcn_mov_dpp(s0, 0xB1, 0xf, 0xf, true), r1 = __builtin_amdgcn_mov_dpp(s1, 0xB1, 0xf, 0xf, true);
        const rba_u32x4 piece = odd ? (rba_u32x4){r0, r1, l[0], l[1]} : (rba_u32x4){h[0], h[1], r0, r1};
        const int k8 = c4 >> 1;
        if (TR) {
          // piece column = ((block * 4 + 2 g + (h|l)) * 2 + k-half): the order of the image within a row group, 512 bytes apart
          const int pcol = (((k8 >> 2) * 4 + (k8 & 1) * 2 + (odd ? 1 : 0)) << 1) + ((k8 >> 1) & 1);
          const int r8 = threadIdx.x >> 6;
          tr_lds[pcol * 8 + (r8 ^ (pcol & 7))] = piece;
        } else {
          const int64_t off = ((row >> 5) * (int64_t)(C >> 5) + (k8 >> 2)) * 4096 + ((k8 & 1) * 2 + (odd ? 1 : 0)) * 1024 +
                              (((k8 >> 1) & 1) * 32 + (int)(row & 31)) * 16;
          *reinterpret_cast<rba_u32x4*>(reinterpret_cast<char*>(y) + off) = piece;
        }
      } else {
        *reinterpret_cast<f32x4*>(y + base + 4 * c4) = o;
      }
    }
  }
  if (TR) {
    __syncthreads();
    const int64_t row0 = (int64_t)blockIdx.x * 8;                                   // the workgroup's eight rows: one aligned octet of a row group
    char* img = reinterpret_cast<char*>(y) + ((row0 >> 5) * (int64_t)(C >> 5)) * 4096 + (int)(row0 & 31) * 16;
    const int npc = nv4;                                                            // piece columns per row (C / 8 h pieces + C / 8 l pieces)
    for (int u = threadIdx.x; u < npc * 8; u += 64 * WAVES) {
      const int pcol = u >> 3, r8 = u & 7;
      const int x = pcol >> 1;
      if (row0 + r8 < rows)
        *reinterpret_cast<rba_u32x4*>(img + (int64_t)(x >> 2) * 4096 + (x & 3) * 1024 + (pcol & 1) * 512 + r8 * 16) = tr_lds[pcol * 8 + (r8 ^ (pcol & 7))];
    }
  }
}

template <int G, int NV, bool FRAG = false>
int launch(const float* x, const float* t, const float* tb, const float* gamma, const float* beta, float* sum_out, float* y,
           int64_t rows, int C, float eps, hipStream_t st) {
  const int64_t rpb = 4 * (64 / G);
  const int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((add_layer_norm_kernel<G, NV, false, FRAG>), dim3((unsigned)blocks), dim3(256), 0, st, x, t, tb, gamma, beta, sum_out, y,
                     rows, C, eps);
  return rba_launch_status();
}

template <int NV>
int launch_frag8(const float* x, const float* t, const float* tb, const float* gamma, const float* beta, float* sum_out, float* y,
                 int64_t rows, int C, float eps, hipStream_t st) {
  const int64_t blocks = (rows + 7) / 8;
  if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((add_layer_norm_kernel<64, NV, false, true, 8>), dim3((unsigned)blocks), dim3(512), 0, st, x, t, tb, gamma, beta, sum_out,
                     y, rows, C, eps);
  return rba_launch_status();
}

template <int G, int NV>
int launch_merge(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C, float eps, MergeGeom mg,
                 hipStream_t st) {
  const int64_t rpb = 4 * (64 / G);
  const int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((add_layer_norm_kernel<G, NV, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, nullptr, nullptr, gamma, beta, nullptr,
                     y, rows, C, eps, mg);
  return rba_launch_status();
}

}  // namespace

// PatchMerging's gather + LayerNorm in one pass: x [B, H, W, Cin] (token-major) -> y [B * ceil(H/2) * ceil(W/2), 4 Cin] =
// LN(cat(x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2])) with zero padding of odd maps (backbone/swin.py:311-337).  Cin % 4 == 0.
extern "C" int rba_merge_layer_norm_f32(const float* x, const float* gamma, const float* beta, float* y, int B, int H, int W, int Cin,
                                        float eps, void* stream) {
  RBA_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && Cin >= 4 && Cin % 4 == 0 && 4 * Cin <= 8192);
  if (B == 0) return 0;
  RBA_CHECK_ARG(x && gamma && beta && y);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y) & 15) == 0);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const int C = 4 * Cin, nv4 = C / 4;
  const int64_t rows = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  const MergeGeom mg{H, W, Cin};
#define RBA_L(G, NV) return launch_merge<G, NV>(x, gamma, beta, y, rows, C, eps, mg, st)
  if (nv4 <= 16) RBA_L(16, 1);
  if (nv4 <= 32) RBA_L(32, 1);
  if (nv4 <= 64) RBA_L(64, 1);
  if (nv4 <= 128) RBA_L(64, 2);
  if (nv4 <= 256) RBA_L(64, 4);
  if (nv4 <= 512) RBA_L(64, 8);
  if (nv4 <= 1024) RBA_L(64, 16);
  RBA_L(64, 32);
#undef RBA_L
}

extern "C" int rba_add_layer_norm_f32(const float* x, const float* t, const float* t_bias, const float* gamma, const float* beta,
                                      float* sum_out, float* y, int64_t rows, int C, float eps, void* stream) {
  RBA_CHECK_ARG(rows >= 0 && C >= 4 && C % 4 == 0 && C <= 8192);
  if (rows == 0) return 0;
  RBA_CHECK_ARG(x && gamma && beta && y);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)t | (uintptr_t)t_bias | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)sum_out |
                  (uintptr_t)y) & 15) == 0);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const int nv4 = C / 4;
#define RBA_L(G, NV) return launch<G, NV>(x, t, t_bias, gamma, beta, sum_out, y, rows, C, eps, st)
  if (nv4 <= 16) RBA_L(16, 1);
  if (nv4 <= 32) RBA_L(32, 1);
  if (nv4 <= 64) RBA_L(64, 1);
  if (nv4 <= 128) RBA_L(64, 2);
  if (nv4 <= 256) RBA_L(64, 4);
  if (nv4 <= 512) RBA_L(64, 8);
  if (nv4 <= 1024) RBA_L(64, 16);
  RBA_L(64, 32);
#undef RBA_L
}

// The same, with y written as the split fragment image of the f16x3 GEMM's A operand (see FRAG above; 128 B per row per 32 channels,
// rows padded to a multiple of 32: y_frag holds ceil(rows / 32) * 32 * C * 4 bytes, the padding rows are never written).  C % 32 == 0.
extern "C" int rba_add_layer_norm_frag_f32(const float* x, const float* t, const float* t_bias, const float* gamma, const float* beta,
                                           float* sum_out, void* y_frag, int64_t rows, int C, float eps, void* stream) {
  RBA_CHECK_ARG(rows >= 0 && C >= 32 && C % 32 == 0 && C <= 8192);
  if (rows == 0) return 0;
  RBA_CHECK_ARG(x && gamma && beta && y_frag);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)t | (uintptr_t)t_bias | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)sum_out |
                  (uintptr_t)y_frag) & 15) == 0);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const int nv4 = C / 4;
  float* y = reinterpret_cast<float*>(y_frag);
  // 64 lanes per row (the fp32 kernel's own partition of the row, so the same sums bit for bit): eight rows per workgroup, stores
  // transposed through LDS
  if (nv4 > 64 && nv4 <= 128) return launch_frag8<2>(x, t, t_bias, gamma, beta, sum_out, y, rows, C, eps, st);
  if (nv4 > 128 && nv4 <= 256) return launch_frag8<4>(x, t, t_bias, gamma, beta, sum_out, y, rows, C, eps, st);
  if (nv4 > 256 && nv4 <= 512) return launch_frag8<8>(x, t, t_bias, gamma, beta, sum_out, y, rows, C, eps, st);
#define RBA_L(G, NV) return launch<G, NV, true>(x, t, t_bias, gamma, beta, sum_out, y, rows, C, eps, st)
  if (nv4 <= 16) RBA_L(16, 1);
  if (nv4 <= 32) RBA_L(32, 1);
  if (nv4 <= 64) RBA_L(64, 1);
  if (nv4 <= 128) RBA_L(64, 2);
  if (nv4 <= 256) RBA_L(64, 4);
  if (nv4 <= 512) RBA_L(64, 8);
  if (nv4 <= 1024) RBA_L(64, 16);
  RBA_L(64, 32);
#undef RBA_L
}
