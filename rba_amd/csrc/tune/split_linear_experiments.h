// K6 experiments that did NOT beat the product configuration (tools/gemm_v4_sweep.py; profiles/r02_k6_*.txt), kept for the record:
//   v7 -- the activation operand loaded straight into MFMA-fragment registers (only the weights through LDS): equal within 1-3 %;
//   v8 -- v7 with LDS full/empty counters instead of s_barrier: 5-20 % slower (polling).
// Included only by split_linear_tune.hip (librba_tune.so).
#pragma once
#include "../split_linear_dma.h"

namespace {

// ---- v7: the activation operand never touches LDS.  An MFMA wave owns 32 rows; its A operand for a 32-wide k super-stage is,
// per lane, 16 consecutive floats of ONE row (64 contiguous bytes: lanes 0-31 the first half of the row's 128-byte line, lanes
// 32-63 the second half), loaded straight from global memory one super-stage ahead and split in registers.  The MFMA k index is
// only a summation label, so the lane's floats 8 g .. 8 g + 7 serve MFMA step g, and the matching B operand of lane half lh in
// step g is the packed weight block of sub-stage lh, half-slot g (the weights are pre-packed, any k relabelling is free).  Only
// the weight tile goes through LDS (24 KB instead of 40 KB per super-stage and workgroup): the LDS-DMA rate of the loader waves --
// the co-bottleneck of v4, each 1-KiB DMA costs its wave ~250 cycles (profiles/r02_k6_v5_timing.txt) -- drops by 40 %.
template <int ACT, int CT, int D, int L>
__global__ __launch_bounds__(256 + 64 * L, CT <= 4 ? (L >= 4 ? 4 : 3) : 2) void split_linear_v7_kernel(const float* __restrict__ A, const u32x4_t* __restrict__ Wp,
                                                                     const float* __restrict__ bias, float* __restrict__ C, int M,
                                                                     int N, int K, int MT, int NT) {
  constexpr int BM = 128, BN = 32 * CT;
  constexpr int W_UNITS = 6 * BN, SUP = 2 * W_UNITS;                               // 16-byte units per super-stage (two sub-stage blocks)
  constexpr int P = 6 * CT;                                                        // 1-KiB weight pieces per super-stage
  constexpr int NP = (P + L - 1) / L;
  constexpr int R = D + 1;
  static_assert(L >= 1 && (D - 1) * NP <= 63, "vmcnt immediate");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int S16 = K >> 4, NSS = K >> 5;
  const int Np = (N + 127) & ~127;

  if (wave >= 4) {                                                                 // ---------------- loader waves: weights only
    const int iw = wave - 4;
    const u32x4_t* src[NP];
    int dst[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      int e = NP * iw + q;
      e = e < P ? e : P - 1;
      const int g = e / (3 * CT), ew = e - g * 3 * CT, p = ew / CT, jt = ew - p * CT;
      int row = n0 + 32 * jt;
      row = row <= Np - 32 ? row : Np - 32;
      src[q] = Wp + ((int64_t)(row >> 7) * S16 + g) * 768 + p * 256 + (row & 127) * 2 + lane;
      dst[q] = g * W_UNITS + (p * BN + 32 * jt) * 2;
    }
    auto issue = [&](int ss, int slot) {
      const int sc = ss < NSS ? ss : NSS - 1;
#pragma unroll
      for (int q = 0; q < NP; ++q) glds16(src[q] + (int64_t)sc * 1536, v4_dma + slot * SUP + dst[q]);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    int slot = 0;
    for (int ss = 0; ss < NSS; ++ss) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((D - 1) * NP) : "memory");
      int wr = slot + D;
      wr = wr >= R ? wr - R : wr;
      issue(ss + D, wr);
      slot = slot + 1 == R ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---------------- MFMA waves
  const int l31 = lane & 31, lh = lane >> 5;
  int arow = m0 + 32 * wave + l31;
  arow = arow < M ? arow : M - 1;
  const f32x4* ap = reinterpret_cast<const f32x4*>(A + (int64_t)arow * K + 16 * lh);  // + 8 float4 per super-stage
  // B operand of MFMA step g: sub-stage block lh, half-slot g ^ swizzle(row)
  const int sw = (l31 >> 3) & 1;
  const int fb0 = lh * W_UNITS + l31 * 2 + (0 ^ sw), fb1 = lh * W_UNITS + l31 * 2 + (1 ^ sw);
  f32x16_t acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  auto step = [&](const f32x4 (&ac)[4], f32x4 (&an)[4], int ss, int slot) {
    const int sn = ss + 1 < NSS ? ss + 1 : NSS - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) an[i] = ap[sn * 8 + i];                            // next super-stage's activations
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const u32x4_t* img = v4_lds + slot * SUP;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8_t a[3], b[CT][3];
      split8(ac[2 * g], ac[2 * g + 1], a[0], a[1], a[2]);
      const int fb = g ? fb1 : fb0;
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(bf16x8_t, img[fb + (p * BN + 32 * j) * 2]);
#define RBA_G(pa, pb) \
  _Pragma("unroll") for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[j][pb], acc[j], 0, 0, 0);
      RBA_G(0, 0) RBA_G(0, 1) RBA_G(1, 0) RBA_G(1, 1) RBA_G(0, 2) RBA_G(2, 0)
#undef RBA_G
    }
  };
  f32x4 a0[4], a1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = ap[i];
  int slot = 0, ss = 0;
  for (; ss + 1 < NSS; ss += 2) {
    step(a0, a1, ss, slot);
    slot = slot + 1 == R ? 0 : slot + 1;
    step(a1, a0, ss + 1, slot);
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  if (ss < NSS) step(a0, a1, ss, slot);

  const bool interior = m0 + BM <= M && n0 + BN <= N;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = n0 + 32 * j + l31;
    const float bv = (bias && col < N) ? bias[col] : 0.f;
    f32x16_t v = acc[j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] += bv;
      if (ACT == 1) v[r] = gelu_erf(v[r]);
      if (ACT == 2) v[r] = fmaxf(v[r], 0.f);
    }
    const int rbase = m0 + 32 * wave + 4 * lh;
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

template <int ACT, int CT, int D, int L>
int launch_v7(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t stream) {
  constexpr int BN = 32 * CT;
  constexpr size_t dyn = (size_t)(D + 1) * 2 * 6 * BN * 16;
  if ((K >> 5) < D) return (int)hipErrorInvalidValue;
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + BN - 1) / BN;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  static const hipError_t attr = hipFuncSetAttribute((const void*)split_linear_v7_kernel<ACT, CT, D, L>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL((split_linear_v7_kernel<ACT, CT, D, L>), dim3((unsigned)(MT * NT)), dim3(256 + 64 * L), dyn, stream, x, wp, bias,
                     out, (int)M, N, K, (int)MT, NT);
  return 0;
}

template <int CT, int D, int L>
int launch_v7_act(int act, const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t st) {
  if (act == 1) return launch_v7<1, CT, D, L>(x, wp, bias, out, M, N, K, st);
  if (act == 2) return launch_v7<2, CT, D, L>(x, wp, bias, out, M, N, K, st);
  return launch_v7<0, CT, D, L>(x, wp, bias, out, M, N, K, st);
}

// ---- v8: v7 without a single s_barrier in the main loop.  A rendezvous per super-stage ties every wave to the slowest one: the
// loaders (whose DMA issue rate fluctuates with back-pressure) and the four MFMA waves waited 16-25 % / 20 % of their time at it
// (profiles/r02_k6_v5_timing.txt) although the data had landed long before.  Here the ring slots carry two LDS counters each:
// full[slot] (+1 per loader wave whose pieces of that generation have landed) and empty[slot] (+1 per MFMA wave that has finished
// reading it).  A loader may run up to R super-stages ahead; an MFMA wave only ever waits for ITS data, never for another MFMA wave.
__device__ __forceinline__ void lds_wait_ge(const int* cnt, int target) {
  const uint32_t addr = (uint32_t)(uintptr_t)cnt;
  for (;;) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    if (__builtin_amdgcn_readfirstlane(v) >= target) break;
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void lds_signal(int* cnt, int lane) {
  const uint32_t addr = (uint32_t)(uintptr_t)cnt;
  const int one = 1;
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(one) : "memory");
}

template <int ACT, int CT, int R, int L>
__global__ __launch_bounds__(256 + 64 * L, CT <= 4 ? (L >= 4 ? 4 : 3) : 2) void split_linear_v8_kernel(
    const float* __restrict__ A, const u32x4_t* __restrict__ Wp, const float* __restrict__ bias, float* __restrict__ C, int M, int N,
    int K, int MT, int NT) {
  constexpr int BM = 128, BN = 32 * CT;
  constexpr int W_UNITS = 6 * BN, SUP = 2 * W_UNITS;
  constexpr int P = 6 * CT;
  constexpr int NP = (P + L - 1) / L;
  static_assert(L >= 1 && NP <= 63, "vmcnt immediate");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int nb = MT * NT;
  if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
  const int mt = bid / NT, nt = bid - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int S16 = K >> 4, NSS = K >> 5;
  const int Np = (N + 127) & ~127;
  int* full = reinterpret_cast<int*>(v4_lds + R * SUP);                            // [R] then empty[R]
  int* empty = full + R;
  if (tid < 2 * R) full[tid] = 0;
  __syncthreads();

  if (wave >= 4) {                                                                 // ---------------- loader waves: weights only
    const int iw = wave - 4;
    const u32x4_t* src[NP];
    int dst[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      int e = NP * iw + q;
      e = e < P ? e : P - 1;
      const int g = e / (3 * CT), ew = e - g * 3 * CT, p = ew / CT, jt = ew - p * CT;
      int row = n0 + 32 * jt;
      row = row <= Np - 32 ? row : Np - 32;
      src[q] = Wp + ((int64_t)(row >> 7) * S16 + g) * 768 + p * 256 + (row & 127) * 2 + lane;
      dst[q] = g * W_UNITS + (p * BN + 32 * jt) * 2;
    }
    int slot = 0, gen = 0, pslot = 0;                                              // slot / generation of super-stage jj; slot of jj - 1
    for (int jj = 0; jj < NSS; ++jj) {
      if (gen > 0) lds_wait_ge(empty + slot, 4 * gen);                             // every MFMA wave has read the previous generation
#pragma unroll
      for (int q = 0; q < NP; ++q) glds16(src[q] + (int64_t)jj * 1536, v4_dma + slot * SUP + dst[q]);
      if (jj > 0) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");                  // super-stage jj - 1 has landed
        lds_signal(full + pslot, lane);
      }
      pslot = slot;
      if (++slot == R) { slot = 0; ++gen; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_signal(full + pslot, lane);
    return;
  }

  // ---------------- MFMA waves
  const int l31 = lane & 31, lh = lane >> 5;
  int arow = m0 + 32 * wave + l31;
  arow = arow < M ? arow : M - 1;
  const f32x4* ap = reinterpret_cast<const f32x4*>(A + (int64_t)arow * K + 16 * lh);
  const int sw = (l31 >> 3) & 1;
  const int fb0 = lh * W_UNITS + l31 * 2 + (0 ^ sw), fb1 = lh * W_UNITS + l31 * 2 + (1 ^ sw);
  f32x16_t acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  int slot = 0, gen = 0;
  auto step = [&](const f32x4 (&ac)[4], f32x4 (&an)[4], int ss) {
    const int sn = ss + 1 < NSS ? ss + 1 : NSS - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) an[i] = ap[sn * 8 + i];
    lds_wait_ge(full + slot, L * (gen + 1));
    const u32x4_t* img = v4_lds + slot * SUP;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8_t a[3], b[CT][3];
      split8(ac[2 * g], ac[2 * g + 1], a[0], a[1], a[2]);
      const int fb = g ? fb1 : fb0;
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(bf16x8_t, img[fb + (p * BN + 32 * j) * 2]);
#define RBA_G(pa, pb) \
  _Pragma("unroll") for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[j][pb], acc[j], 0, 0, 0);
      RBA_G(0, 0) RBA_G(0, 1) RBA_G(1, 0) RBA_G(1, 1) RBA_G(0, 2) RBA_G(2, 0)
#undef RBA_G
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             // this wave's reads of the slot are complete
    lds_signal(empty + slot, lane);
    if (++slot == R) { slot = 0; ++gen; }
  };
  f32x4 a0[4], a1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = ap[i];
  int ss = 0;
  for (; ss + 1 < NSS; ss += 2) {
    step(a0, a1, ss);
    step(a1, a0, ss + 1);
  }
  if (ss < NSS) step(a0, a1, ss);

  const bool interior = m0 + BM <= M && n0 + BN <= N;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = n0 + 32 * j + l31;
    const float bv = (bias && col < N) ? bias[col] : 0.f;
    f32x16_t v = acc[j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[r] += bv;
      if (ACT == 1) v[r] = gelu_erf(v[r]);
      if (ACT == 2) v[r] = fmaxf(v[r], 0.f);
    }
    const int rbase = m0 + 32 * wave + 4 * lh;
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

template <int ACT, int CT, int R, int L>
int launch_v8(const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t stream) {
  constexpr int BN = 32 * CT;
  constexpr size_t dyn = (size_t)R * 2 * 6 * BN * 16 + 64;
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + BN - 1) / BN;
  if (MT * NT >= (int64_t)1 << 31) return (int)hipErrorInvalidValue;
  static const hipError_t attr = hipFuncSetAttribute((const void*)split_linear_v8_kernel<ACT, CT, R, L>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL((split_linear_v8_kernel<ACT, CT, R, L>), dim3((unsigned)(MT * NT)), dim3(256 + 64 * L), dyn, stream, x, wp, bias,
                     out, (int)M, N, K, (int)MT, NT);
  return 0;
}

template <int CT, int R, int L>
int launch_v8_act(int act, const float* x, const u32x4_t* wp, const float* bias, float* out, int64_t M, int N, int K, hipStream_t st) {
  if (act == 1) return launch_v8<1, CT, R, L>(x, wp, bias, out, M, N, K, st);
  if (act == 2) return launch_v8<2, CT, R, L>(x, wp, bias, out, M, N, K, st);
  return launch_v8<0, CT, R, L>(x, wp, bias, out, M, N, K, st);
}

}  // namespace

// ======================================================================================================================
// f16x3 experiments (round 2) that lose to the product kernels of split_linear_h3.h (profiles/r02_k6_f16x3.txt):
//   * the software-pipelined kernel (split_linear_h3p_kernel, now in the product) WITHOUT explicit scheduling groups: 83 us vs
//     74 us on Swin stage-3 fc1 -- the compiler re-serialises reads and MFMAs; with __builtin_amdgcn_sched_group_barrier patterns
//     69 us; moving the weight staging under the third column tile as well: 72 us;
//   * starting the second workgroup of every CU 4-32 us late (so that one workgroup's epilogue meets the other's k loop):
//     74 -> 74 / 75 / 77 / 90 us: a workgroup's k loop is bound by its own dependency chain, not by its neighbour;
//   * eight waves per 128 x 128 tile (two column halves): no gain on the 256-tile shapes it was meant for (72 vs 72 us).
#include "../split_linear_h3.h"
namespace {
}  // namespace
