// K6 (header: kernel templates shared by split_linear.hip and tune/split_linear_tune.hip)
// v4 -- the bf16x6 Linear (see split_linear.hip for the arithmetic) with ALL operand staging done by LDS-DMA.
// (reference: the nn.Linear calls of backbone/swin.py:44-71 (Mlp), :131-171 (qkv / proj), :319-343 (PatchMerging).)
//
// What changed w.r.t. split_linear_pipe_kernel and why (profiles/r01_split_linear.txt: matrix pipe 63 % busy while resident;
// with the staging removed 65-70 us instead of 95-110):
//   * the activation tile goes to LDS as RAW fp32 (global_load_lds_dwordx4, 16 B per lane, no VGPRs, no ds_write): 4 B per
//     element instead of the 6 B of three bf16 planes, and the hi/mid/lo split moves to the fragment side: a lane's MFMA A
//     operand is 8 consecutive k of one row = 32 contiguous bytes = two ds_read_b128, split in registers (22 VALU per 8
//     floats) right before its MFMAs;
//   * a wave owns 32*RT rows x the tile's full width (waves stacked 4 x 1), so no other wave needs its A fragments: the split
//     is done exactly once per element, and A never needs a cross-wave hand-off beyond the DMA landing;
//   * the packed weight tile (rba_split_weight_bf16x3: already the LDS image) is DMA'd as before;
//   * one barrier per G 16-wide k sub-stages (G = 2: 48 MFMAs per wave between barriers instead of 24), DMA issued D
//     super-stages ahead into a ring of D + 1 super-buffers, counted s_waitcnt vmcnt -- the wave never waits for a load it
//     issued less than a full super-stage ago.
// LDS image of one 16-wide sub-stage: A [BM rows][4 chunks of 16 B] fp32 with chunk c of row r at slot c ^ g(r >> 2),
// g(x) = (x ^ (x >> 1)) & 3 (conflict-free ds_read_b128 for the MFMA row-per-lane pattern: brute-force checked), then the three
// weight planes [3][BN][2 half-slots] x 16 B exactly as packed in global memory.  A DMA writes lane i's 16 bytes to base + 16 i,
// so the swizzle is applied to the per-lane SOURCE address.
#include <stdlib.h>

#pragma once
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// One dynamic LDS block seen through two symbols (HIP places every `extern __shared__` array at the same base; the compiler
// treats them as distinct objects): the DMAs write through v4_dma, every ds_read goes through v4_lds.  Otherwise hipcc orders
// each LDS read behind the pending LDS-DMAs with s_waitcnt vmcnt(0) and the prefetch collapses.  The ordering that IS needed is
// the counted vmcnt + barrier at the head of each super-stage.
extern __shared__ __attribute__((aligned(16))) u32x4_t v4_lds[];
extern __shared__ __attribute__((aligned(16))) u32x4_t v4_dma[];

__device__ __forceinline__ uint32_t pack_bf16(float x0, float x1) {           // rne; lowers to v_cvt_pk_bf16_f32
  bf16x2_t v = {(__bf16)x0, (__bf16)x1};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo_as_f32(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_as_f32(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }

// 8 fp32 (two 16-byte LDS reads) -> the three bf16x8 MFMA operands hi / mid / lo (x = hi + mid + lo exactly)
__device__ __forceinline__ void split8(const f32x4 u, const f32x4 v, bf16x8_t& p0, bf16x8_t& p1, bf16x8_t& p2) {
  const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
  u32x4_t h, m, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t a = pack_bf16(x[2 * i], x[2 * i + 1]);
    const float r0 = x[2 * i] - lo_as_f32(a), r1 = x[2 * i + 1] - hi_as_f32(a);
    const uint32_t b = pack_bf16(r0, r1);
    const float s0 = r0 - lo_as_f32(b), s1 = r1 - hi_as_f32(b);
    h[i] = a;
    m[i] = b;
    l[i] = pack_bf16(s0, s1);
  }
  p0 = __builtin_bit_cast(bf16x8_t, h);
  p1 = __builtin_bit_cast(bf16x8_t, m);
  p2 = __builtin_bit_cast(bf16x8_t, l);
}

// Exact-form GELU (nn.GELU default, swin.py:51), erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): see split_linear.hip
__device__ __forceinline__ float gelu_erf(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = p * t * __expf(-x * x);
  const float one_plus_erf = v >= 0.f ? 2.0f - e : e;
  return 0.5f * v * one_plus_erf;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of the function: set it once per device (one process per GPU
// is the deployment model, but a process that touches a second device must not inherit the first one's "already set")
template <typename KernelT>
static inline int ensure_dynamic_lds(KernelT kernel, size_t bytes, unsigned char (&done)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
  if (!done[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    done[dev] = 1;
  }
  return 0;
}

__device__ __forceinline__ void glds16(const void* gsrc, u32x4_t* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ACT: 0 none, 1 exact GELU, 2 ReLU.  RT: 32-row MFMA tiles per wave (BM = 128 RT).  CT: 32-column tiles (BN = 32 CT).
// G: 16-wide k sub-stages per barrier.  D: DMA look-ahead in super-stages (ring of D + 1 super-buffers).
// PROBE (ablation builds, tools/gemm_v4_sweep.py only; results are wrong): bit 0 no DMA in the loop, bit 1 no hi/mid/lo split
// (raw bits as operands), bit 2 no epilogue stores, bit 3 operand fragments read once (MFMAs alone), bit 4 no barrier
template <int ACT, int RT, int CT, int G, int D, int PROBE = 0, int L = 0>
__global__ __launch_bounds__(256 + 64 * L) void split_linear_v4_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                                     const float* __restrict__ bias, float* __restrict__ C, int M,
                                                                     int N, int K, int MT, int NT) {
  constexpr int BM = 128 * RT, BN = 32 * CT;
  constexpr int A_UNITS = BM * 4, W_UNITS = 6 * BN, SUB = A_UNITS + W_UNITS;      // 16-byte units per sub-stage image
  // L = 0: the four MFMA waves issue the DMAs themselves.  L > 0: L extra LOADER waves (waves 4 .. 3 + L) do nothing else: an
  // LDS-DMA costs its issuing wave 60-185 cycles of issue (MI355X_MICROARCH.md), 10 of them per 48 MFMAs stalled the in-order
  // MFMA waves for a quarter of the kernel (ablation: 89 -> 74 us with the DMAs removed).
  constexpr int NI = L > 0 ? L : 4;                                                // issuing waves
  constexpr int PA = 8 * RT, P = PA + 3 * CT;                                      // 1-KiB pieces per sub-stage: A rows / W planes
  constexpr int NP = (P + NI - 1) / NI;                                            // pieces per issuing wave per sub-stage
  constexpr int NDMA = G * NP;                                                     // per issuing wave per super-stage
  constexpr int R = D + 1;                                                         // ring length in super-stages
  static_assert((D - 1) * NDMA <= 63, "vmcnt immediate");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order: consecutive workgroups go round-robin to the 8 XCDs; give each XCD a contiguous run of logical tiles
  // (n fastest) so the NT column tiles that re-read one A row-tile hit the same L2.
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int S16 = K >> 4;                                                          // 16-wide sub-stages
  const int NSS = S16 / G;                                                         // super-stages (host guarantees divisibility)
  const bool issuer = L > 0 ? wave >= 4 : true;
  const int iw = L > 0 ? wave - 4 : wave;

  // ---- DMA pieces of this issuing wave.  Piece e < PA: rows 16 e .. 16 e + 15 of the A tile, lane i the 16-byte chunk
  // c = (i & 3) ^ g(row >> 2) of row 16 e + (i >> 2).  Piece e >= PA: (plane p, 32-row group jt) of the packed weight tile =
  // 1 KiB contiguous in [N/128][K/16][3][128][2] x 16 B.  Surplus slots (NP NI > P) re-fetch the last piece.
  const char* src[NP];
  int dst[NP], step[NP];                                                           // LDS unit inside the sub-stage image; bytes per sub-stage
  const int Np = (N + 127) & ~127;
  if (issuer) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      int e = NP * iw + q;
      e = e < P ? e : P - 1;
      if (e < PA) {
        const int r = 16 * e + (lane >> 2);
        const int x = (r >> 2) & 7;
        const int c = (lane & 3) ^ ((x ^ (x >> 1)) & 3);
        int row = m0 + r;
        row = row < M ? row : M - 1;
        src[q] = reinterpret_cast<const char*>(A + (int64_t)row * K + 4 * c);
        dst[q] = e * 64;
        step[q] = 64;
      } else {
        const int ew = e - PA, p = ew / CT, jt = ew - p * CT;
        int row = n0 + 32 * jt;
        row = row <= Np - 32 ? row : Np - 32;                                      // columns beyond the padded weight: never stored
        src[q] = reinterpret_cast<const char*>(Wp + ((int64_t)(row >> 7) * S16) * 768 + p * 256 + (row & 127) * 2 + lane);
        dst[q] = A_UNITS + (p * BN + 32 * jt) * 2;
        step[q] = 768 * 16;
      }
    }
  }
  auto issue = [&](int ss, int slot) {                                             // super-stage ss -> ring slot
    const int sc = ss < NSS ? ss : NSS - 1;                                        // clamped surplus keeps the vmcnt counts uniform
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int s = sc * G + g;
      u32x4_t* base = v4_dma + (slot * G + g) * SUB;
#pragma unroll
      for (int q = 0; q < NP; ++q) glds16(src[q] + (int64_t)s * step[q], base + dst[q]);
    }
  };

  if (PROBE & 32) { if (wave < 4) __builtin_amdgcn_s_setprio(1); }                 // tune builds: static priority for the MFMA waves
  if (PROBE & 64) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }                //              ... or for the loader waves
  if (L > 0 && wave >= 4) {                                                        // ---------------- loader waves
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    int slot = 0;
    for (int ss = 0; ss < NSS; ++ss) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((D - 1) * NDMA) : "memory");
      int wr = slot + D;
      wr = wr >= R ? wr - R : wr;
      issue(ss + D, wr);
      slot = slot + 1 == R ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- fragment addresses (16-byte units inside a sub-stage image)
  const int l31 = lane & 31, lh = lane >> 5;
  int fa[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int r = 32 * (RT * wave + t) + l31;
    const int x = (r >> 2) & 7;
    fa[t] = r * 4 + ((2 * lh) ^ ((x ^ (x >> 1)) & 3));                             // second chunk: fa ^ 1
  }
  const int fb = A_UNITS + l31 * 2 + (lh ^ ((l31 >> 3) & 1));                      // + (p BN + 32 j) 2

  f32x16_t acc[RT][CT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

  if (L == 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
  }

  int slot = 0;
  for (int ss = 0; ss < NSS; ++ss) {
    // super-stage ss has landed once at most (D - 1) newer super-stages of this wave's DMAs are outstanding; the barrier makes
    // that true for every wave's pieces and also says that every wave has finished reading the slot refilled next.
    if (L > 0)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (PROBE & 16)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((D - 1) * NDMA) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((D - 1) * NDMA) : "memory");
    if (L == 0 && !(PROBE & 1)) {
      int wr = slot + D;
      wr = wr >= R ? wr - R : wr;
      issue(ss + D, wr);
    }
    const u32x4_t* img = v4_lds + slot * (G * SUB);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const u32x4_t* im = (PROBE & 8) ? v4_lds : img + g * SUB;
      bf16x8_t a[RT][3], b[CT][3];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const f32x4 u = __builtin_bit_cast(f32x4, im[fa[t]]);
        const f32x4 v = __builtin_bit_cast(f32x4, im[fa[t] ^ 1]);
        if (PROBE & 2) {
          a[t][0] = __builtin_bit_cast(bf16x8_t, u);
          a[t][1] = __builtin_bit_cast(bf16x8_t, v);
          a[t][2] = __builtin_bit_cast(bf16x8_t, u + v);
        } else {
          split8(u, v, a[t][0], a[t][1], a[t][2]);
        }
      }
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(bf16x8_t, im[fb + (p * BN + 32 * j) * 2]);
#define RBA_G(pa, pb)                                                                          \
  _Pragma("unroll") for (int t = 0; t < RT; ++t) _Pragma("unroll") for (int j = 0; j < CT; ++j) \
      acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][pa], b[j][pb], acc[t][j], 0, 0, 0);
      RBA_G(0, 0) RBA_G(0, 1) RBA_G(1, 0) RBA_G(1, 1) RBA_G(0, 2) RBA_G(2, 0)
#undef RBA_G
    }
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  if (L == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // drain the clamped surplus DMAs

  // ---- epilogue: lane holds D[row = 8 (r / 4) + 4 (lane / 32) + r % 4][col = lane % 32] of each 32 x 32 tile; the 16 values of
  // a tile are formed in 16 distinct registers and stored back to back (see store_tile in split_linear.hip)
  const bool interior = m0 + BM <= M && n0 + BN <= N;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = n0 + 32 * j + l31;
    const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      f32x16_t v = acc[t][j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] += bv;
        if (ACT == 1) v[r] = gelu_erf(v[r]);
        if (ACT == 2) v[r] = rba_relu(v[r]);
      }
      const int rbase = m0 + 32 * (RT * wave + t) + 4 * lh;
      float* dst = C + (int64_t)rbase * N + col;
      if (PROBE & 4) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += v[r];
        if (sum == 1234.5f) dst[0] = sum;
      } else if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(int64_t)(8 * (r >> 2) + (r & 3)) * N] = v[r];
      } else if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = 8 * (r >> 2) + (r & 3);
          if (rbase + ro < M) dst[(int64_t)ro * N] = v[r];
        }
      }
    }
  }
}

// 4 x 4 transpose across the four lanes of a quad: before, lane c holds X[i][c] in r_i; after, lane c holds X[c][k] in r_k.
__device__ __forceinline__ float dpp_quad(float v, bool xor2) {
  const int x = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, xor2 ? __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true)      // quad_perm [2,3,0,1]
                                        : __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
}
__device__ __forceinline__ void quad_transpose(float& r0, float& r1, float& r2, float& r3, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float t;
  t = dpp_quad(b0 ? r0 : r1, false); if (b0) r0 = t; else r1 = t;
  t = dpp_quad(b0 ? r2 : r3, false); if (b0) r2 = t; else r3 = t;
  t = dpp_quad(b1 ? r0 : r2, true);  if (b1) r0 = t; else r2 = t;
  t = dpp_quad(b1 ? r1 : r3, true);  if (b1) r1 = t; else r3 = t;
}

// ---- v5: PERSISTENT.  The v4 structure (loader waves + MFMA waves, ring of D + 1 super-buffers, one barrier per super-stage) with
// each workgroup walking a list of tiles: the loader waves run one continuous stream of super-stages ACROSS tile boundaries, so
// the first operands of tile t + 1 are in LDS while the MFMA waves still store tile t, and no workgroup is ever launched or
// retired mid-kernel (counters on v4, Swin stage-3 fc1: 6.3 of 8 wave slots occupied on average, the matrix pipe 88 % busy while
// occupied even with everything but the MFMAs removed -- a 32-stage tile is too short to amortise a workgroup's start and end).
// Tile order: the workgroups of one XCD (blockIdx % 8) own a contiguous run of logical tiles (n fastest) and work through it
// side by side, so that the column tiles re-reading one A row-tile, and the row tiles re-reading one W tile, meet in that XCD's L2.
template <int ACT, int RT, int CT, int G, int D, int L, bool TIMING = false, int MW = 4, bool WIDE = false>
__global__ __launch_bounds__(64 * (MW + L)) void split_linear_v5_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                                     const float* __restrict__ bias, float* __restrict__ C, int M,
                                                                     int N, int K, int MT, int NT, unsigned long long* dbg = nullptr) {
  constexpr int BM = 32 * RT * MW, BN = 32 * CT;
  constexpr int A_UNITS = BM * 4, W_UNITS = 6 * BN, SUB = A_UNITS + W_UNITS;
  constexpr int PA = BM / 16, P = PA + 3 * CT;
  constexpr int NP = (P + L - 1) / L;
  constexpr int NDMA = G * NP;
  constexpr int R = D + 1;
  static_assert(L >= 1 && (D - 1) * NDMA <= 63, "vmcnt immediate");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S16 = K >> 4, NSS = S16 / G;
  const int ntiles = MT * NT, nwg = gridDim.x;
  // i-th tile of this workgroup (-1: none)
  const bool xcd_runs = (ntiles & 7) == 0 && (nwg & 7) == 0;
  const int per = xcd_runs ? ntiles >> 3 : ntiles, gp = xcd_runs ? nwg >> 3 : nwg;
  const int first = xcd_runs ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, base_t = xcd_runs ? (int)(blockIdx.x & 7) * per : 0;
  const int my_tiles = first < per ? (per - first + gp - 1) / gp : 0;               // tiles first, first + gp, ... < per
  const int T = my_tiles * NSS;                                                    // super-stages this workgroup streams
  if (T == 0) return;
  const int Np = (N + 127) & ~127;

  if (wave >= MW) {                                                                // ---------------- loader waves
    const int iw = wave - MW;
    const char* src[NP];
    int dst[NP], step[NP];
    auto setup = [&](int ti) {                                                     // DMA source addresses of this wave's pieces
      const int tile = base_t + first + ti * gp;
      const int mt = tile / NT, nt = tile - mt * NT;
      const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        int e = NP * iw + q;
        e = e < P ? e : P - 1;
        if (e < PA) {
          const int r = 16 * e + (lane >> 2);
          const int x = (r >> 2) & 7;
          const int c = (lane & 3) ^ ((x ^ (x >> 1)) & 3);
          int row = m0 + r;
          row = row < M ? row : M - 1;
          src[q] = reinterpret_cast<const char*>(A + (int64_t)row * K + 4 * c);
          dst[q] = e * 64;
          step[q] = 64;
        } else {
          const int ew = e - PA, p = ew / CT, jt = ew - p * CT;
          int row = n0 + 32 * jt;
          row = row <= Np - 32 ? row : Np - 32;
          src[q] = reinterpret_cast<const char*>(Wp + ((int64_t)(row >> 7) * S16) * 768 + p * 256 + (row & 127) * 2 + lane);
          dst[q] = A_UNITS + (p * BN + 32 * jt) * 2;
          step[q] = 768 * 16;
        }
      }
    };
    int c_tile = 0, c_ss = 0;                                                      // cursor: next super-stage to issue
    setup(0);
    auto issue_next = [&](int slot) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int s = c_ss * G + g;
        u32x4_t* base = v4_dma + (slot * G + g) * SUB;
#pragma unroll
        for (int q = 0; q < NP; ++q) glds16(src[q] + (int64_t)s * step[q], base + dst[q]);
      }
      if (c_ss + 1 < NSS) {
        ++c_ss;
      } else if (c_tile + 1 < my_tiles) {
        ++c_tile;
        c_ss = 0;
        setup(c_tile);
      }                                                                            // else: stay (surplus issues re-fetch the last one)
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue_next(d);
    int slot = 0;
    unsigned long long t_vm = 0, t_bar = 0, t_iss = 0, t0 = 0, t_start = 0;
    if (TIMING) t_start = __builtin_readcyclecounter();
    for (int j = 0; j < T; ++j) {
      if (TIMING) {
        t0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NDMA) : "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_barrier" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        t_vm += t1 - t0;
        t_bar += t2 - t1;
        t0 = t2;
      } else {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((D - 1) * NDMA) : "memory");
      }
      int wr = slot + D;
      wr = wr >= R ? wr - R : wr;
      issue_next(wr);
      if (TIMING) t_iss += __builtin_readcyclecounter() - t0;
      slot = slot + 1 == R ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TIMING && dbg && iw == 0 && lane == 0) {
      dbg[blockIdx.x * 8 + 4] = t_vm;
      dbg[blockIdx.x * 8 + 5] = t_bar;
      dbg[blockIdx.x * 8 + 6] = t_iss;
      dbg[blockIdx.x * 8 + 7] = __builtin_readcyclecounter() - t_start;
    }
    return;
  }

  // ---------------- MFMA waves
  const int l31 = lane & 31, lh = lane >> 5;
  int fa[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int r = 32 * (RT * wave + t) + l31;
    const int x = (r >> 2) & 7;
    fa[t] = r * 4 + ((2 * lh) ^ ((x ^ (x >> 1)) & 3));
  }
  const int fb = A_UNITS + l31 * 2 + (lh ^ ((l31 >> 3) & 1));
  int slot = 0;
  unsigned long long m_bar = 0, m_cmp = 0, m_epi = 0, m_start = 0;
  if (TIMING) m_start = __builtin_readcyclecounter();
  for (int ti = 0; ti < my_tiles; ++ti) {
    f32x16_t acc[RT][CT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
    for (int ss = 0; ss < NSS; ++ss) {
      unsigned long long tb0 = 0, tb1 = 0;
      if (TIMING) tb0 = __builtin_readcyclecounter();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (TIMING) {
        tb1 = __builtin_readcyclecounter();
        m_bar += tb1 - tb0;
      }
      const u32x4_t* img = v4_lds + slot * (G * SUB);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const u32x4_t* im = img + g * SUB;
        bf16x8_t a[RT][3], b[CT][3];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const f32x4 u = __builtin_bit_cast(f32x4, im[fa[t]]);
          const f32x4 v = __builtin_bit_cast(f32x4, im[fa[t] ^ 1]);
          split8(u, v, a[t][0], a[t][1], a[t][2]);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(bf16x8_t, im[fb + (p * BN + 32 * j) * 2]);
#define RBA_G(pa, pb)                                                                          \
  _Pragma("unroll") for (int t = 0; t < RT; ++t) _Pragma("unroll") for (int j = 0; j < CT; ++j) \
      acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][pa], b[j][pb], acc[t][j], 0, 0, 0);
        RBA_G(0, 0) RBA_G(0, 1) RBA_G(1, 0) RBA_G(1, 1) RBA_G(0, 2) RBA_G(2, 0)
#undef RBA_G
      }
      slot = slot + 1 == R ? 0 : slot + 1;
      if (TIMING) {
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]));
        m_cmp += __builtin_readcyclecounter() - tb1;
      }
    }
    unsigned long long te0 = 0;
    if (TIMING) te0 = __builtin_readcyclecounter();
    // ---- epilogue of this tile (the loader is already filling the ring with the next tile's first super-stages)
    const int tile = base_t + first + ti * gp;
    const int mt = tile / NT, nt = tile - mt * NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const bool interior = m0 + BM <= M && n0 + BN <= N;
    if (WIDE && (N & 3) == 0) {
      // 16-byte stores: a 4 x 4 transpose inside each lane quad (two DPP butterfly steps) turns "lane = one column, registers =
      // 4 consecutive rows" into "lane = one row, registers = 4 consecutive columns": 4 store instructions per 32 x 32 tile
      // instead of 16 (the epilogue was store-ISSUE bound: ~107 cycles per store instruction, 10 % of the kernel)
      const int qc = l31 & 3, col4 = 4 * (l31 >> 2);
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int col = n0 + 32 * j + col4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias && col < N) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int rbase = m0 + 32 * (RT * wave + t) + 4 * lh + qc;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            float r0 = acc[t][j][4 * g4], r1 = acc[t][j][4 * g4 + 1], r2 = acc[t][j][4 * g4 + 2], r3 = acc[t][j][4 * g4 + 3];
            quad_transpose(r0, r1, r2, r3, lane);
            f32x4 v = {r0 + bv.x, r1 + bv.y, r2 + bv.z, r3 + bv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (ACT == 1) v[k] = gelu_erf(v[k]);
              if (ACT == 2) v[k] = rba_relu(v[k]);
            }
            const int row = rbase + 8 * g4;
            if (interior || (row < M && col < N)) *reinterpret_cast<f32x4*>(C + (int64_t)row * N + col) = v;
          }
        }
      }
    } else {
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int col = n0 + 32 * j + l31;
      const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        f32x16_t v = acc[t][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] += bv;
          if (ACT == 1) v[r] = gelu_erf(v[r]);
          if (ACT == 2) v[r] = rba_relu(v[r]);
        }
        const int rbase = m0 + 32 * (RT * wave + t) + 4 * lh;
        float* dst = C + (int64_t)rbase * N + col;
        if (interior) {
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[(int64_t)(8 * (r >> 2) + (r & 3)) * N] = v[r];
        } else if (col < N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = 8 * (r >> 2) + (r & 3);
            if (rbase + ro < M) dst[(int64_t)ro * N] = v[r];
          }
        }
      }
    }
    }
    if (TIMING) m_epi += __builtin_readcyclecounter() - te0;
  }
  if (TIMING && dbg && wave == 0 && lane == 0) {
    dbg[blockIdx.x * 8 + 0] = m_bar;
    dbg[blockIdx.x * 8 + 1] = m_cmp;
    dbg[blockIdx.x * 8 + 2] = m_epi;
    dbg[blockIdx.x * 8 + 3] = __builtin_readcyclecounter() - m_start;
  }
}

template <int ACT, int RT, int CT, int G, int D, int L, int MW = 4, bool WIDE = false>
int launch_v5(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, int wgs_per_cu,
              hipStream_t stream) {
  constexpr int BM = 32 * RT * MW, BN = 32 * CT;
  constexpr size_t dyn = (size_t)(D + 1) * G * (BM * 4 + 6 * BN) * 16;
  static_assert(dyn <= 160 * 1024, "LDS");
  if ((K >> 4) % G || (K >> 4) / G < D) return (int)hipErrorInvalidValue;
  const int64_t MT = (M + BM - 1) / BM;
  const int NT = (N + BN - 1) / BN;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  static unsigned char attr_done[64];
  if (const int rc = ensure_dynamic_lds(split_linear_v5_kernel<ACT, RT, CT, G, D, L, false, MW, WIDE>, dyn, attr_done)) return rc;
  const int64_t tiles = MT * NT;
  int64_t grid = (int64_t)256 * wgs_per_cu;
  grid = tiles < grid ? tiles : grid;
  if ((tiles & 7) == 0 && grid >= 8) grid &= ~(int64_t)7;
  hipLaunchKernelGGL((split_linear_v5_kernel<ACT, RT, CT, G, D, L, false, MW, WIDE>), dim3((unsigned)grid), dim3(64 * (MW + L)), dyn,
                     stream, x, wp, bias, out, (int)M, N, K, (int)MT, NT, (unsigned long long*)nullptr);
  return 0;
}

template <int RT, int CT, int G, int D, int L, int MW = 4, bool WIDE = false>
int launch_v5_act(int act, const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, int wpc,
                  hipStream_t st) {
  if (act == 1) return launch_v5<1, RT, CT, G, D, L, MW, WIDE>(x, wp, bias, out, M, N, K, wpc, st);
  if (act == 2) return launch_v5<2, RT, CT, G, D, L, MW, WIDE>(x, wp, bias, out, M, N, K, wpc, st);
  return launch_v5<0, RT, CT, G, D, L, MW, WIDE>(x, wp, bias, out, M, N, K, wpc, st);
}

template <int ACT, int RT, int CT, int G, int D, int PROBE = 0, int L = 0>
int launch_v4(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t stream) {
  constexpr int BM = 128 * RT, BN = 32 * CT;
  constexpr size_t dyn = (size_t)(D + 1) * G * (BM * 4 + 6 * BN) * 16;
  static_assert(dyn <= 160 * 1024, "LDS");
  if ((K >> 4) % G) return (int)hipErrorInvalidValue;
  const int64_t MT = (M + BM - 1) / BM;
  const int NT = (N + BN - 1) / BN;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  static unsigned char attr_done[64];
  if (const int rc = ensure_dynamic_lds(split_linear_v4_kernel<ACT, RT, CT, G, D, PROBE, L>, dyn, attr_done)) return rc;
  hipLaunchKernelGGL((split_linear_v4_kernel<ACT, RT, CT, G, D, PROBE, L>), dim3((unsigned)(MT * NT)), dim3(256 + 64 * L), dyn, stream, x, wp, bias, out,
                     (int)M, N, K, (int)MT, NT);
  return 0;
}

template <int RT, int CT, int G, int D, int L = 0>
int launch_v4_act(int act, const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t st) {
  if (act == 1) return launch_v4<1, RT, CT, G, D, 0, L>(x, wp, bias, out, M, N, K, st);
  if (act == 2) return launch_v4<2, RT, CT, G, D, 0, L>(x, wp, bias, out, M, N, K, st);
  return launch_v4<0, RT, CT, G, D, 0, L>(x, wp, bias, out, M, N, K, st);
}

}  // namespace
