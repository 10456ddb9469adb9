// Shared K1 templates (rba_reduce.hip = the shipped entry points, rba_reduce_tune.hip = probes and experimental variants).
#pragma once
#include <stdlib.h>
#include "common.h"
#include "../../include/rba_hip.h"

namespace rba_k1 {


template <int VEC>
struct VecT;
template <>
struct VecT<4> { using type = f32x4; };
template <>
struct VecT<2> { using type = f32x2; };
template <>
struct VecT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  using T = typename VecT<VEC>::type;
  const T t = __builtin_nontemporal_load(reinterpret_cast<const T*>(p));   // streamed once: keep it out of L2's way
  if constexpr (VEC == 1) v[0] = t;
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = t[i];
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  using T = typename VecT<VEC>::type;
  T t;
  if constexpr (VEC == 1) t = v[0];
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) t[i] = v[i];
  }
  *reinterpret_cast<T*>(p) = t;
}

// epilogue shared by both kernels: tanh-sum, optional sem_seg / argmax stores for VEC pixels at p0
// score modes (the reference's interchangeable anomaly_score_func's on the same sem_seg):
//   0  RbA               -sum_k tanh(sem_k)        evaluate_ood.py:143-150
//   1  energy            -logsumexp_k(sem_k)       evaluate_ood.py:152-159
//   2  neg. logit sum    -sum_k sem_k              support.py:115-132
template <int KMAX, int VEC>
__device__ __forceinline__ void rba_score(const float (&acc)[KMAX][VEC], int K, int mode, float (&r)[VEC]) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) r[i] = 0.f;
  if (mode == 0) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K)
#pragma unroll
        for (int i = 0; i < VEC; ++i) r[i] -= rba_tanh(acc[k][i]);
  } else if (mode == 2) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K)
#pragma unroll
        for (int i = 0; i < VEC; ++i) r[i] -= acc[k][i];
  } else {
    float mx[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) mx[i] = acc[0][i];
#pragma unroll
    for (int k = 1; k < KMAX; ++k)
      if (k < K)
#pragma unroll
        for (int i = 0; i < VEC; ++i) mx[i] = fmaxf(mx[i], acc[k][i]);
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K)
#pragma unroll
        for (int i = 0; i < VEC; ++i) r[i] += expf(acc[k][i] - mx[i]);
#pragma unroll
    for (int i = 0; i < VEC; ++i) r[i] = -(mx[i] + logf(r[i]));
  }
}

// epilogue shared by both kernels: score, optional sem_seg / argmax stores for VEC pixels at p0
template <int KMAX, int VEC, bool SEM, bool ARG>
__device__ __forceinline__ void rba_epilogue(float (&acc)[KMAX][VEC], int K, int mode, float* rba, float* sem, int32_t* argmax,
                                             int64_t p0, int64_t plane) {
  float r[VEC];
  rba_score<KMAX, VEC>(acc, K, mode, r);
  int best[VEC];
  float bestv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { best[i] = 0; bestv[i] = acc[0][i]; }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) {
      if (ARG) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (acc[k][i] > bestv[i]) { bestv[i] = acc[k][i]; best[i] = k; }
      }
      if (SEM) store_vec<VEC>(sem + (int64_t)k * plane + p0, acc[k]);
    }
  }
  store_vec<VEC>(rba + p0, r);
  if (ARG) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) argmax[p0 + i] = best[i];
  }
}

// One thread owns VEC consecutive pixels.  KMAX = compile-time bound on K (== K on the fast path).
template <int KMAX, int VEC, bool SEM, bool ARG>
__global__ __launch_bounds__(256) void rba_reduce_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                         float* __restrict__ rba, float* __restrict__ sem,
                                                         int32_t* __restrict__ argmax, int Q, int K, int64_t HW, int mode) {
  const int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (p0 >= HW) return;
  float acc[KMAX][VEC];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[k][i] = 0.f;

  const float* mp = mask + p0;
#pragma unroll 4
  for (int q = 0; q < Q; ++q) {
    float m[VEC], s[VEC];
    load_vec<VEC>(mp + (int64_t)q * HW, m);
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = rba_sigmoid(m[i]);
    const float* pq = prob + q * K;   // wave-uniform -> s_load
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        const float pk = pq[k];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[k][i] = fmaf(pk, s[i], acc[k][i]);
      }
    }
  }
  rba_epilogue<KMAX, VEC, SEM, ARG>(acc, K, mode, rba, sem, argmax, p0, HW);
}

// x4 upsample fused in front: thread owns 4 consecutive output pixels of one output row, i.e. output
// columns 4*j .. 4*j+3 which interpolate low-res columns j-1, j, j+1 and rows (i0, i1) of the low-res map.
template <int KMAX, bool SEM, bool ARG>
__global__ __launch_bounds__(256) void rba_reduce_up4_kernel(const float* __restrict__ low, const float* __restrict__ prob,
                                                             float* __restrict__ rba, float* __restrict__ sem,
                                                             int32_t* __restrict__ argmax, int Q, int K, int h, int w,
                                                             int crop_h, int crop_w, int wq /* ceil(crop_w/4) */, int mode) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;   // low-res column == group of 4 output columns
  const int y = blockIdx.y;                              // output row
  if (j >= wq) return;
  const BilinearTap ty = bilinear_tap(y, 0.25f, h);
  // output x = 4j+r, r=0..3: src = j + (r+0.5)/4 - 0.5 -> taps (j-1,j) for r<2, (j,j+1) for r>=2, clamped
  BilinearTap tx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) tx[r] = bilinear_tap(4 * j + r, 0.25f, w);
  const int jm = j > 0 ? j - 1 : 0, jp = j < w - 1 ? j + 1 : w - 1;

  float acc[KMAX][4];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;

  const float* r0 = low + (int64_t)ty.i0 * w;
  const float* r1 = low + (int64_t)ty.i1 * w;
  const int64_t plane = (int64_t)h * w;
#pragma unroll 2
  for (int q = 0; q < Q; ++q) {
    const float a0 = r0[jm], a1 = r0[j], a2 = r0[jp];
    const float b0 = r1[jm], b1 = r1[j], b2 = r1[jp];
    r0 += plane; r1 += plane;
    float s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // ATen order: l0h*(l0w*v00 + l1w*v01) + l1h*(l0w*v10 + l1w*v11).  Columns: r<2 -> (j-1, j), r>=2 -> (j, j+1);
      // at a clamped border ATen's (i0,i1,l1) is (0,1,0) resp. (w-1,w-1,l1): jm/jp clamping gives the same value.
      const float v00 = r < 2 ? a0 : a1, v01 = r < 2 ? a1 : a2;
      const float v10 = r < 2 ? b0 : b1, v11 = r < 2 ? b1 : b2;
      const float top = tx[r].l0 * v00 + tx[r].l1 * v01;
      const float bot = tx[r].l0 * v10 + tx[r].l1 * v11;
      s[r] = rba_sigmoid(ty.l0 * top + ty.l1 * bot);
    }
    const float* pq = prob + q * K;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        const float pk = pq[k];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[k][i] = fmaf(pk, s[i], acc[k][i]);
      }
    }
  }
  const int64_t oplane = (int64_t)crop_h * crop_w;
  const int64_t p0 = (int64_t)y * crop_w + 4 * j;
  if (4 * j + 3 < crop_w && (crop_w & 3) == 0) {
    rba_epilogue<KMAX, 4, SEM, ARG>(acc, K, mode, rba, sem, argmax, p0, oplane);
  } else {  // ragged right edge / unaligned rows: scalar stores
    float r[4];
    rba_score<KMAX, 4>(acc, K, mode, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * j + i >= crop_w) break;
      float bv = acc[0][i];
      int b = 0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          if (acc[k][i] > bv) { bv = acc[k][i]; b = k; }
          if (SEM) sem[(int64_t)k * oplane + p0 + i] = acc[k][i];
        }
      }
      rba[p0 + i] = r[i];
      if (ARG) argmax[p0 + i] = b;
    }
  }
}

template <int K, int VEC, bool SEM, bool ARG, int U, int WPS>
__global__ __launch_bounds__(256, WPS) void rba_reduce_fast_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                                 float* __restrict__ rba, float* __restrict__ sem,
                                                                 int32_t* __restrict__ argmax, int Q, int64_t HW, int tiles, int mode) {
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t p0 = ((int64_t)tile * 256 + threadIdx.x) * VEC;
    if (p0 >= HW) continue;
    float acc[K][VEC];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[k][i] = 0.f;
    const float* mp = mask + p0;
    float buf[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) load_vec<VEC>(mp + (int64_t)(u < Q ? u : Q - 1) * HW, buf[u]);
    const int Qmain = Q / U * U;
    for (int q0 = 0; q0 < Qmain; q0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = q0 + u;
        float s[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] = rba_sigmoid(buf[u][i]);
        const int qn = q + U < Q ? q + U : Q - 1;          // clamped prefetch (re-reads the last plane from L2)
        load_vec<VEC>(mp + (int64_t)qn * HW, buf[u]);
        const float* pq = prob + q * K;                    // wave-uniform -> scalar loads, SGPR operands
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float pk = pq[k];
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[k][i] = fmaf(pk, s[i], acc[k][i]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {                          // tail: Q % U planes, already in the ring
      const int q = Qmain + u;
      if (q < Q) {
        const float* pq = prob + q * K;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float si = rba_sigmoid(buf[u][i]);
#pragma unroll
          for (int k = 0; k < K; ++k) acc[k][i] = fmaf(pq[k], si, acc[k][i]);
        }
      }
    }
    rba_epilogue<K, VEC, SEM, ARG>(acc, K, mode, rba, sem, argmax, p0, HW);
  }
}

// Explicitly packed formulation of the fast kernel (4 pixels per thread): accumulators and sigmoid values live in
// float2 register pairs from the start, so every FMA is a v_pk_fma_f32 whose operands need no pairing moves
// (the scalar formulation is SLP-vectorised by the compiler, which inserts 21 v_mov per two planes to build the pairs).
// Same operations in the same order per element as rba_reduce_fast_kernel: bit-identical results.
template <int K, bool SEM, bool ARG, int U, int WPS, bool DYN>
__global__ __launch_bounds__(256, WPS) void rba_reduce_pk_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                               float* __restrict__ rba, float* __restrict__ sem,
                                                               int32_t* __restrict__ argmax, int Q, int64_t HW, int tiles, int mode,
                                                               unsigned int* __restrict__ counters) {
  // DYN: tiles are handed out by an atomic counter in a caller-provided workspace (two zeroed uint32), so workgroups on slower
  // CUs take fewer tiles -- with a static split identical workgroups finish between 103 and 159 us (residency probe in
  // profiles/r01_k1_bandwidth_probes.txt).  The last workgroup to leave zeroes the workspace again.
  __shared__ unsigned int sh_tile;
  auto next_tile = [&](int prev) -> int {
    if (!DYN) return prev < 0 ? (int)blockIdx.x : prev + (int)gridDim.x;
    __syncthreads();
    if (threadIdx.x == 0) sh_tile = atomicAdd(counters, 1u);
    __syncthreads();
    return (int)sh_tile;
  };
  for (int tile = next_tile(-1); tile < tiles; tile = next_tile(tile)) {
    const int64_t p0 = ((int64_t)tile * 256 + threadIdx.x) * 4;
    if (p0 < HW) {
    f32x2 a01[K], a23[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a01[k] = a23[k] = (f32x2){0.f, 0.f};
    const float* mp = mask + p0;
    f32x4 buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)(u < Q ? u : Q - 1) * HW));
    const int Qmain = Q / U * U;
    for (int q0 = 0; q0 < Qmain; q0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = q0 + u;
        const f32x2 s01 = rba_sigmoid2((f32x2){buf[u].x, buf[u].y});
        const f32x2 s23 = rba_sigmoid2((f32x2){buf[u].z, buf[u].w});
        const int qn = q + U < Q ? q + U : Q - 1;
        buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)qn * HW));
        const float* pq = prob + q * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const f32x2 pk = {pq[k], pq[k]};
          a01[k] = __builtin_elementwise_fma(pk, s01, a01[k]);
          a23[k] = __builtin_elementwise_fma(pk, s23, a23[k]);
        }
      }
    }
    float acc[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) { acc[k][0] = a01[k].x; acc[k][1] = a01[k].y; acc[k][2] = a23[k].x; acc[k][3] = a23[k].y; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = Qmain + u;
      if (q < Q) {
        const float* pq = prob + q * K;
        const float b[4] = {buf[u].x, buf[u].y, buf[u].z, buf[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float si = rba_sigmoid(b[i]);
#pragma unroll
          for (int k = 0; k < K; ++k) acc[k][i] = fmaf(pq[k], si, acc[k][i]);
        }
      }
    }
    rba_epilogue<K, 4, SEM, ARG>(acc, K, mode, rba, sem, argmax, p0, HW);
    }
  }
  if (DYN && threadIdx.x == 0) {
    const unsigned int done = atomicAdd(counters + 1, 1u);
    if (done == gridDim.x - 1) {                                   // every other workgroup has fetched its last (out-of-range) tile
      atomicExch(counters, 0u);
      atomicExch(counters + 1, 0u);
    }
  }
}

// counters == nullptr: static tile split (re-entrant, nothing but the arguments); otherwise dynamic assignment through the workspace
template <int K, int U, int WPS>
int launch_reduce_pk(const float* mask, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int64_t HW, int mode,
                     unsigned int* counters, hipStream_t st) {
  const int64_t tiles = (HW + 1023) / 1024;
  if (tiles > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  const int64_t cap = 256 * WPS;
  int64_t grid = tiles;
  // no more tiles than workgroup slots (e.g. 736 x 1280: 920 tiles on 1024 slots): every workgroup takes exactly one tile, and fetching it from
  // the atomic counter (two barriers, an atomic round trip, the exit count) only delays its first loads -- measured 0.54 -> 0.67 of 8 TB/s
  // (tools/k1_sweep.py 60 --hw 736 1280, profiles/r03_k1_c5.txt); the dynamic hand-out pays when workgroups take several tiles each
  if (tiles <= cap) counters = nullptr;
  if (tiles > cap) {
    const int64_t rounds = (tiles + cap - 1) / cap;
    grid = counters ? cap : (tiles + rounds - 1) / rounds;
  }
#define RBA_L(S, A, D) \
  hipLaunchKernelGGL((rba_reduce_pk_kernel<K, S, A, U, WPS, D>), dim3((unsigned)grid), dim3(256), 0, st, mask, prob, rba, sem, argmax, Q, HW, (int)tiles, mode, counters)
  if (counters) {
    if (sem && argmax) RBA_L(true, true, true); else if (sem) RBA_L(true, false, true); else if (argmax) RBA_L(false, true, true); else RBA_L(false, false, true);
  } else {
    if (sem && argmax) RBA_L(true, true, false); else if (sem) RBA_L(true, false, false); else if (argmax) RBA_L(false, true, false); else RBA_L(false, false, false);
  }
#undef RBA_L
  return rba_launch_status();
}

// LDS-DMA ring formulation (4 pixels per thread): each wave streams its query planes through a private ring of R 1-KiB LDS
// slots with global_load_lds_dwordx4 (no destination VGPRs, so R planes per wave are in flight instead of 2), reads its own
// 16 bytes back with ds_read_b128 once the counted vmcnt says the plane has landed, and refills the slot.  No barrier: a lane
// only ever reads what it loaded itself.  The compiler does not order a ds_read behind a pending LDS-DMA, so the wait is an
// explicit s_waitcnt vmcnt(R - 1) and the read is inline asm.  Arithmetic identical to rba_reduce_pk_kernel (bit-identical).
template <int K, bool SEM, bool ARG, int R, int WPS>
__global__ __launch_bounds__(256, WPS) void rba_reduce_dma_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                                float* __restrict__ rba, float* __restrict__ sem,
                                                                int32_t* __restrict__ argmax, int Q, int64_t HW, int tiles, int mode) {
  static_assert(R >= 2 && R <= 15, "vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* wring = ring_raw + wave * (R * 1024);                       // this wave's ring
  const uint32_t lds_lane = (uint32_t)(uintptr_t)(wring) + lane * 16;        // LDS byte address of this lane's 16 B in slot 0
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t p0 = ((int64_t)tile * 256 + threadIdx.x) * 4;              // HW % 1024 == 0 is required by the launcher
    f32x2 a01[K], a23[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a01[k] = a23[k] = (f32x2){0.f, 0.f};
    const float* mp = mask + p0;
#pragma unroll
    for (int u = 0; u < R; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mp + (int64_t)(u < Q ? u : Q - 1) * HW),
                                       (__attribute__((address_space(3))) void*)(wring + u * 1024), 16, 0, 0);
    for (int q0 = 0; q0 < Q; q0 += R) {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int q = q0 + u;
        if (q < Q) {                                                         // wave-uniform
          f32x4 v;
          // plane q is the oldest of the R DMAs in flight: wait until at most R - 1 are outstanding, then read it back
          asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b128 %0, %1 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                       : "=v"(v) : "v"(lds_lane), "n"(R - 1), "n"(u * 1024) : "memory");
          const int qn = q + R < Q ? q + R : Q - 1;                          // clamped refill keeps the count exact
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mp + (int64_t)qn * HW),
                                           (__attribute__((address_space(3))) void*)(wring + u * 1024), 16, 0, 0);
          const f32x2 s01 = {rba_sigmoid(v.x), rba_sigmoid(v.y)};
          const f32x2 s23 = {rba_sigmoid(v.z), rba_sigmoid(v.w)};
          const float* pq = prob + q * K;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const f32x2 pk = {pq[k], pq[k]};
            a01[k] = __builtin_elementwise_fma(pk, s01, a01[k]);
            a23[k] = __builtin_elementwise_fma(pk, s23, a23[k]);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // drain the clamped refills before the ring is reused
    float acc[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) { acc[k][0] = a01[k].x; acc[k][1] = a01[k].y; acc[k][2] = a23[k].x; acc[k][3] = a23[k].y; }
    rba_epilogue<K, 4, SEM, ARG>(acc, K, mode, rba, sem, argmax, p0, HW);
  }
}

template <int K, bool SEM, bool ARG, int R, int WPS>
int launch_reduce_dma(const float* mask, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int64_t HW, int mode,
                      hipStream_t st) {
  if (HW % 1024) return (int)hipErrorInvalidValue;
  const int64_t tiles = HW / 1024;
  if (tiles > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  const int64_t cap = 256 * WPS;
  int64_t grid = tiles;
  if (tiles > cap) {
    const int64_t rounds = (tiles + cap - 1) / cap;
    grid = (tiles + rounds - 1) / rounds;
  }
  hipLaunchKernelGGL((rba_reduce_dma_kernel<K, SEM, ARG, R, WPS>), dim3((unsigned)grid), dim3(256), 4 * R * 1024, st, mask, prob, rba, sem,
                     argmax, Q, HW, (int)tiles, mode);
  return rba_launch_status();
}

template <int K, int VEC, int U, int WPS>
int launch_reduce_fast(const float* mask, const float* prob, float* rba, float* sem, int32_t* argmax, int Q,
                       int64_t HW, hipStream_t st, int mode = 0) {
  const int64_t per_block = 256 * (int64_t)VEC;
  const int64_t tiles = (HW + per_block - 1) / per_block;
  if (tiles > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  // 256 CUs x WPS resident 256-thread blocks (WPS waves per SIMD); round the grid so that every block
  // gets the same number of tiles when possible
  const int64_t cap = 256 * WPS;
  int64_t grid = tiles;
  if (tiles > cap) {
    const int64_t rounds = (tiles + cap - 1) / cap;
    grid = (tiles + rounds - 1) / rounds;
  }
#define RBA_L(S, A) \
  hipLaunchKernelGGL((rba_reduce_fast_kernel<K, VEC, S, A, U, WPS>), dim3((unsigned)grid), dim3(256), 0, st, mask, prob, rba, sem, argmax, Q, HW, (int)tiles, mode)
  if (sem && argmax) RBA_L(true, true);
  else if (sem) RBA_L(true, false);
  else if (argmax) RBA_L(false, true);
  else RBA_L(false, false);
#undef RBA_L
  return rba_launch_status();
}

typedef float f32x4_m __attribute__((ext_vector_type(4)));

template <int KMAX, int VEC>
int launch_reduce(const float* mask, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int K,
                  int64_t HW, hipStream_t st, int mode = 0) {
  const int threads = 256;
  const int64_t per_block = (int64_t)threads * VEC;
  const unsigned blocks = (unsigned)((HW + per_block - 1) / per_block);
#define RBA_L(S, A) \
  hipLaunchKernelGGL((rba_reduce_kernel<KMAX, VEC, S, A>), dim3(blocks), dim3(threads), 0, st, mask, prob, rba, sem, argmax, Q, K, HW, mode)
  if (sem && argmax) RBA_L(true, true);
  else if (sem) RBA_L(true, false);
  else if (argmax) RBA_L(false, true);
  else RBA_L(false, false);
#undef RBA_L
  return rba_launch_status();
}

// The same fused kernel for a compile-time K with the arithmetic of rba_reduce_pk_kernel (round 3: this is the product's default K1 path --
// MaskFormer.rba_scores, i.e. what the evaluator runs -- and it is VALU-bound: 52 MB in, 8 MB out).  Per query and 4 output pixels the generic
// kernel above issues ~120 scalar vector instructions; here the two column pairs (r = 0, 1 share low-res columns j-1, j; r = 2, 3 share j, j+1)
// are interpolated with packed fp32 (ATen's operation order per element), the sigmoids are rba_sigmoid2 and the 4 K class FMAs are 2 K
// v_pk_fma_f32 with the class probability as a scalar operand: ~54 packed + 8 transcendental.  Two queries per trip, the twelve taps of the
// next two requested before the current two are consumed.
template <int K, bool SEM, bool ARG>
__global__ __launch_bounds__(256) void rba_reduce_up4_pk_kernel(const float* __restrict__ low, const float* __restrict__ prob,
                                                                float* __restrict__ rba, float* __restrict__ sem,
                                                                int32_t* __restrict__ argmax, int Q, int h, int w, int crop_h, int crop_w,
                                                                int wq, int mode) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (j >= wq) return;
  const BilinearTap ty = bilinear_tap(y, 0.25f, h);
  BilinearTap tx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) tx[r] = bilinear_tap(4 * j + r, 0.25f, w);
  const int jm = j > 0 ? j - 1 : 0, jc = j < w ? j : w - 1, jp = j < w - 1 ? j + 1 : w - 1;
  const f32x2 l0a = {tx[0].l0, tx[1].l0}, l1a = {tx[0].l1, tx[1].l1}, l0b = {tx[2].l0, tx[3].l0}, l1b = {tx[2].l1, tx[3].l1};
  f32x2 a01[K], a23[K];
#pragma unroll
  for (int k = 0; k < K; ++k) a01[k] = a23[k] = (f32x2){0.f, 0.f};
  const int64_t plane = (int64_t)h * w;
  const float* r0 = low + (int64_t)ty.i0 * w;
  const float* r1 = low + (int64_t)ty.i1 * w;
  float t[2][6];
  auto taps = [&](int q, float (&d)[6]) {
    const int qc = q < Q ? q : Q - 1;
    const float* p0 = r0 + qc * plane;
    const float* p1 = r1 + qc * plane;
    d[0] = p0[jm]; d[1] = p0[jc]; d[2] = p0[jp];
    d[3] = p1[jm]; d[4] = p1[jc]; d[5] = p1[jp];
  };
  auto one = [&](int q, const float (&d)[6]) {
    // ATen: l0h * (l0w * v00 + l1w * v01) + l1h * (l0w * v10 + l1w * v11); columns r < 2 -> (j-1, j), r >= 2 -> (j, j+1)
    const f32x2 topa = l0a * d[0] + l1a * d[1], bota = l0a * d[3] + l1a * d[4];
    const f32x2 topb = l0b * d[1] + l1b * d[2], botb = l0b * d[4] + l1b * d[5];
    const f32x2 s01 = rba_sigmoid2(topa * ty.l0 + bota * ty.l1);
    const f32x2 s23 = rba_sigmoid2(topb * ty.l0 + botb * ty.l1);
    const float* pq = prob + q * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const f32x2 pk = {pq[k], pq[k]};
      a01[k] = __builtin_elementwise_fma(pk, s01, a01[k]);
      a23[k] = __builtin_elementwise_fma(pk, s23, a23[k]);
    }
  };
  taps(0, t[0]);
  taps(1, t[1]);
  int q = 0;
  for (; q + 1 < Q; q += 2) {
    float n0[6], n1[6];
    taps(q + 2, n0);
    taps(q + 3, n1);
    one(q, t[0]);
    one(q + 1, t[1]);
#pragma unroll
    for (int i = 0; i < 6; ++i) { t[0][i] = n0[i]; t[1][i] = n1[i]; }
  }
  if (q < Q) one(q, t[0]);
  float acc[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) { acc[k][0] = a01[k].x; acc[k][1] = a01[k].y; acc[k][2] = a23[k].x; acc[k][3] = a23[k].y; }
  const int64_t oplane = (int64_t)crop_h * crop_w;
  const int64_t p0 = (int64_t)y * crop_w + 4 * j;
  if (4 * j + 3 < crop_w && (crop_w & 3) == 0) {
    rba_epilogue<K, 4, SEM, ARG>(acc, K, mode, rba, sem, argmax, p0, oplane);
  } else {
    float r[4];
    rba_score<K, 4>(acc, K, mode, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * j + i >= crop_w) break;
      float bv = acc[0][i];
      int b = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (acc[k][i] > bv) { bv = acc[k][i]; b = k; }
        if (SEM) sem[(int64_t)k * oplane + p0 + i] = acc[k][i];
      }
      rba[p0 + i] = r[i];
      if (ARG) argmax[p0 + i] = b;
    }
  }
}

template <int K>
int launch_up4_pk(const float* low, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int h, int w, int crop_h, int crop_w,
                  hipStream_t st, int mode) {
  const int wq = (crop_w + 3) / 4;
  const int threads = wq >= 256 ? 256 : (wq >= 128 ? 128 : 64);
  dim3 grid((wq + threads - 1) / threads, crop_h);
#define RBA_L(S, A) \
  hipLaunchKernelGGL((rba_reduce_up4_pk_kernel<K, S, A>), grid, dim3(threads), 0, st, low, prob, rba, sem, argmax, Q, h, w, crop_h, crop_w, wq, mode)
  if (sem && argmax) RBA_L(true, true);
  else if (sem) RBA_L(true, false);
  else if (argmax) RBA_L(false, true);
  else RBA_L(false, false);
#undef RBA_L
  return rba_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// The fused kernel with the class contraction on the matrix pipe ("up4 mx", end of round 3; score only).  Unlike the HBM form of K1 this
// kernel reads 52 MB that sit in L2 / Infinity Cache -- no load pattern to protect -- so a lane is free to take the queries the MFMA B operand
// wants: v_mfma_f32_32x32x16_f16 with M = classes (K <= 32), N = 32 pixels, k = 16 queries; lane (n = lane % 32, kb = lane / 32) holds
// queries 16 s + 8 kb .. + 7 of ITS OWN pixel.  A lane still interpolates four neighbouring output pixels per query from 3 x 2 taps (the
// packed-fp32 arithmetic of rba_reduce_up4_pk_kernel, ATen's order), which makes four column tiles (tile r = pixels 4 n + r) and 64
// accumulator registers; the two lane halves work on different queries of the same 128 pixels.  sigma and P are split h + l with UNSCALED f16
// l (both in [0, 1]; see the note at rba_reduce_m4_kernel in tune/rba_reduce_experiments.h: absolute error 2^-25), three products into one
// fp32 accumulator.  Per pixel and query the vector work drops from 13.5 packed + 2 transcendental to 5.5 + 2; the 19 x 4 FMAs per query
// become 12 MFMAs per 16 queries.  Class probabilities: split once per workgroup into LDS in A-fragment order ([16-query step][h | l][lane] x 16 B).
typedef _Float16 up4_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 up4_f16x2 __attribute__((ext_vector_type(2)));
typedef float up4_f32x16 __attribute__((ext_vector_type(16)));

// Two pairs (a0, a1), (b0, b1) -> packed f16 h and packed f16 residual l = f16(x - h): v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16 reading h's
// halves in place.  The four mix instructions are ONE asm statement in the order lo, lo, hi, hi: hipcc enforces three gfx950 wait-state rules
// for instructions it emits itself and cannot enforce inside asm (it does not decode asm text), so the statement satisfies them by construction:
//  * a VALU that reads a register written by a 16-bit-destination VALU (v_fma_mixlo_f16 writes half a register; v_fma_mixhi_f16 of the SAME
//    register reads it to keep that half) gets one wait state from hipcc -- round 3's form issued mixlo, mixhi of one register back to back
//    (40 of 40 sites in the emitted ISA, tools/isa_hazards.py); interleaving two registers gives every mixhi its wait state for free;
//  * a VALU that reads a fresh transcendental result (the inputs are v_rcp_f32 results of the sigmoid) needs one wait state: leading s_nop;
//  * the consumer of the last half-written register (an MFMA operand here) needs one as well: trailing s_nop.
// Measured (profiles/r04_k1_mx_soak.txt, profiles/r04_mix_hazard_probe.txt): the back-to-back form is NOT observably wrong on MI355X --
// 2.3e10 pairs in tools/micro/mix_hazard.hip and 6 000 three-stream launches of round 3's build show no differing bit, same as this form --
// so this is conformance with the compiler's own rule, not a demonstrated fix; the single run-to-run difference round 3 saw once was never
// reproduced.  tests/test_host_cpu.py::test_emitted_isa_has_no_unfenced_16bit_destination_hazards keeps every code object of the library clean.
__device__ __forceinline__ void up4_split_quad(float a0, float a1, float b0, float b1, uint32_t& ha, uint32_t& la, uint32_t& hb, uint32_t& lb) {
  const up4_f16x2 va = {(_Float16)a0, (_Float16)a1}, vb = {(_Float16)b0, (_Float16)b1};
  const uint32_t pa = __builtin_bit_cast(uint32_t, va), pb = __builtin_bit_cast(uint32_t, vb);
  uint32_t ra, rb;
  asm("s_nop 0\n\t"
      "v_fma_mixlo_f16 %0, %2, %4, %5 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %3, %4, %7 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %2, %4, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %3, %4, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 0"
      : "=&v"(ra), "=&v"(rb) : "v"(pa), "v"(pb), "v"(-1.0f), "v"(a0), "v"(a1), "v"(b0), "v"(b1));
  ha = pa; la = ra; hb = pb; lb = rb;
}

// SEM / ARG (round 4): the accumulators hold sem[class][pixel] -- the evaluator's `return_preds=True` path (support.py:385-388) and the stock
// get_RbA / get_logits on out[0]["sem_seg"] (evaluate_ood.py:143-150) get the same kernel: a lane stores its classes' four pixels as one
// 16-byte piece per class (32 lanes = 512 contiguous bytes of a class row), the argmax is the first maximum in class order (torch.argmax /
// the packed VALU kernel's rule), combined across the two lane halves with ties to the smaller class index.
template <int K, bool SEM = false, bool ARG = false>
__global__ __launch_bounds__(256, 3) void rba_reduce_up4_mx_kernel(const float* __restrict__ low, const float* __restrict__ prob,
                                                                   float* __restrict__ rba, float* __restrict__ sem, int32_t* __restrict__ argmax,
                                                                   int Q, int h, int w, int crop_h, int crop_w,
                                                                   int xtiles, int tiles, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char up4lds[];       // [ceil(Q / 16)][2][64] x 16 B
  const int QS = (Q + 15) >> 4;
  for (int e = threadIdx.x; e < QS * 64; e += 256) {
    const int ln = e & 63, ks = e >> 6, m = ln & 31, q0 = 16 * ks + 8 * (ln >> 5);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (m < K && q0 + i < Q) ? prob[(q0 + i) * K + m] : 0.f;
    rba_u32x4 th, tl;
    uint32_t a, b;
    uint32_t c, d;
    up4_split_quad(v[0], v[1], v[2], v[3], a, b, c, d); th.x = a; tl.x = b; th.y = c; tl.y = d;
    up4_split_quad(v[4], v[5], v[6], v[7], a, b, c, d); th.z = a; tl.z = b; th.w = c; tl.w = d;
    *reinterpret_cast<rba_u32x4*>(up4lds + (size_t)ks * 2048 + ln * 16) = th;
    *reinterpret_cast<rba_u32x4*>(up4lds + (size_t)ks * 2048 + 1024 + ln * 16) = tl;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, kb = lane >> 5;
  const int tile = blockIdx.x * 4 + wave;                                       // one wave = 128 output pixels of one row
  if (tile >= tiles) return;
  const int y = tile / xtiles, xt = tile - y * xtiles;
  const int j = 32 * xt + n;                                                    // this lane's low-res column = its four output columns / 4
  const BilinearTap ty = bilinear_tap(y, 0.25f, h);
  BilinearTap tx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) tx[r] = bilinear_tap(4 * j + r, 0.25f, w);
  const int jl = j < w ? j : w - 1;
  const int jm = jl > 0 ? jl - 1 : 0, jc = jl, jp = jl < w - 1 ? jl + 1 : w - 1;
  const f32x2 l0a = {tx[0].l0, tx[1].l0}, l1a = {tx[0].l1, tx[1].l1}, l0b = {tx[2].l0, tx[3].l0}, l1b = {tx[2].l1, tx[3].l1};
  const int plane = h * w;                                                      // Q * h * w < 2^29 (launcher): byte offsets fit 32 bits
  // Addressing: query u = 16 s + i of lane half 0 and u + 8 of half 1 are read in ONE instruction = wave-uniform base (low + u * plane, scalar
  // arithmetic) + a per-lane byte offset (row, column, and the half's 8 planes clamped to Q - 1: three vector instructions per query).
  // (A variant that skipped the clamp behind a wave-uniform "not the tail" branch was 0 % faster and produced intermittently wrong tiles --
  // loads into the same registers on both sides of the branch; not pursued.)
  uint32_t off[6];
  {
    const int r0 = ty.i0 * w, r1 = ty.i1 * w;
    const int c[6] = {r0 + jm, r0 + jc, r0 + jp, r1 + jm, r1 + jc, r1 + jp};
#pragma unroll
    for (int k = 0; k < 6; ++k) off[k] = (uint32_t)c[k] * 4u;
  }
  auto taps = [&](int qq, float (&d)[6]) {
    const int u = 16 * (qq >> 3) + (qq & 7);                                    // wave-uniform
    const int uc = u < Q ? u : Q - 1;
    const char* base = reinterpret_cast<const char*>(low + (int64_t)uc * plane);
    int qc = u + 8 * kb;
    qc = qc < Q ? qc : Q - 1;
    const uint32_t extra = (uint32_t)(qc - uc) * (uint32_t)plane * 4u;           // padded queries meet zero probabilities
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = *reinterpret_cast<const float*>(base + (off[k] + extra));
  };
  up4_f32x16 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
  float t[4][6];                                                                 // ring: query i of a step lives in t[i % 4]; four queries in flight
#pragma unroll
  for (int i = 0; i < 4; ++i) taps(i, t[i]);
  for (int ks = 0; ks < QS; ++ks) {
    f32x2 s01[8], s23[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float d0 = t[i & 3][0], d1 = t[i & 3][1], d2 = t[i & 3][2], d3 = t[i & 3][3], d4 = t[i & 3][4], d5 = t[i & 3][5];
      taps(8 * ks + i + 4, t[i & 3]);                                           // beyond the last step: clamped to Q - 1, never used
      // ATen: l0h * (l0w * v00 + l1w * v01) + l1h * (l0w * v10 + l1w * v11); columns r < 2 -> (j-1, j), r >= 2 -> (j, j+1)
      const f32x2 topa = l0a * d0 + l1a * d1, bota = l0a * d3 + l1a * d4;
      const f32x2 topb = l0b * d1 + l1b * d2, botb = l0b * d4 + l1b * d5;
      s01[i] = rba_sigmoid2(topa * ty.l0 + bota * ty.l1);
      s23[i] = rba_sigmoid2(topb * ty.l0 + botb * ty.l1);
    }
    const up4_f16x8 ah = __builtin_bit_cast(up4_f16x8, *reinterpret_cast<const rba_u32x4*>(up4lds + (size_t)ks * 2048 + lane * 16));
    const up4_f16x8 al = __builtin_bit_cast(up4_f16x8, *reinterpret_cast<const rba_u32x4*>(up4lds + (size_t)ks * 2048 + 1024 + lane * 16));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rba_u32x4 bh, bl;
      uint32_t a, b, c, d;
#define RBA_SG(i) (r == 0 ? s01[i].x : (r == 1 ? s01[i].y : (r == 2 ? s23[i].x : s23[i].y)))
      up4_split_quad(RBA_SG(0), RBA_SG(1), RBA_SG(2), RBA_SG(3), a, b, c, d); bh.x = a; bl.x = b; bh.y = c; bl.y = d;
      up4_split_quad(RBA_SG(4), RBA_SG(5), RBA_SG(6), RBA_SG(7), a, b, c, d); bh.z = a; bl.z = b; bh.w = c; bl.w = d;
#undef RBA_SG
      const up4_f16x8 vbh = __builtin_bit_cast(up4_f16x8, bh), vbl = __builtin_bit_cast(up4_f16x8, bl);
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vbh, acc[r], 0, 0, 0);
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, vbl, acc[r], 0, 0, 0);
      acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, vbh, acc[r], 0, 0, 0);
    }
  }
  // acc[r][i] = sem[class 8 (i / 4) + 4 kb + i % 4][pixel 4 j + r]: score over this half's classes, then across the two lane halves
  float out[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float part = 0.f, mx = -INFINITY;
    if (mode == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int cls = 8 * (i >> 2) + 4 * kb + (i & 3);
        if (8 * (i >> 2) + (i & 3) < K && cls < K) mx = fmaxf(mx, acc[r][i]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, RBA_WAVE));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int cls = 8 * (i >> 2) + 4 * kb + (i & 3);
      if (8 * (i >> 2) + (i & 3) < K) {                                          // compile-time prune of rows no half can use
        const float v = acc[r][i];
        const float term = mode == 0 ? rba_tanh(v) : (mode == 2 ? v : expf(v - mx));
        part += cls < K ? term : 0.f;
      }
    }
    part += __shfl_xor(part, 32, RBA_WAVE);
    out[r] = mode == 1 ? -(mx + logf(part)) : -part;
  }
  const int64_t p0 = (int64_t)y * crop_w + 4 * j;
  if (kb == 0 && 4 * j < crop_w)
    *reinterpret_cast<f32x4*>(rba + p0) = (f32x4){out[0], out[1], out[2], out[3]};                           // crop_w % 4 == 0 (launcher)
  if (SEM && 4 * j < crop_w) {
    const int64_t oplane = (int64_t)crop_h * crop_w;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int cls = 8 * (i >> 2) + 4 * kb + (i & 3);
      if (8 * (i >> 2) + (i & 3) < K && cls < K)
        __builtin_nontemporal_store((f32x4){acc[0][i], acc[1][i], acc[2][i], acc[3][i]}, reinterpret_cast<f32x4*>(sem + (int64_t)cls * oplane + p0));
    }
  }
  if (ARG) {
    int best[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float bv = -INFINITY;
      int b = 0x7fffffff;                                                          // a half without a valid class (K <= 4 kb) never wins
#pragma unroll
      for (int i = 0; i < 16; ++i) {                                               // increasing class order within this half
        const int cls = 8 * (i >> 2) + 4 * kb + (i & 3);
        if (8 * (i >> 2) + (i & 3) < K) {
          const float v = acc[r][i];
          const bool take = cls < K && (v > bv || b == 0x7fffffff);                // first valid class seeds the scan (also when it is -inf / NaN-free data)
          bv = take ? v : bv;
          b = take ? cls : b;
        }
      }
      const float ov = __shfl_xor(bv, 32, RBA_WAVE);
      const int ob = __shfl_xor(b, 32, RBA_WAVE);
      best[r] = (ov > bv || (ov == bv && ob < b)) ? ob : b;
    }
    if (kb == 0 && 4 * j < crop_w) *reinterpret_cast<rba_u32x4*>(argmax + p0) = (rba_u32x4){(uint32_t)best[0], (uint32_t)best[1], (uint32_t)best[2], (uint32_t)best[3]};
  }
}

template <int K>
int launch_up4_mx(const float* low, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int h, int w, int crop_h, int crop_w,
                  hipStream_t st, int mode) {
  const int wq = crop_w / 4;
  const int xtiles = (wq + 31) / 32;
  const int64_t tiles = (int64_t)xtiles * crop_h;
  if (tiles > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  const size_t shm = (size_t)((Q + 15) / 16) * 2048;
  if (shm > 64 * 1024 || (int64_t)(Q + 8) * h * w >= (1LL << 29)) return (int)hipErrorInvalidValue;
#define RBA_L(S, A)                                                                                                                  \
  hipLaunchKernelGGL((rba_reduce_up4_mx_kernel<K, S, A>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), shm, st, low, prob, rba, sem, argmax, Q, \
                     h, w, crop_h, crop_w, xtiles, (int)tiles, mode)
  if (sem && argmax) RBA_L(true, true);
  else if (sem) RBA_L(true, false);
  else if (argmax) RBA_L(false, true);
  else RBA_L(false, false);
#undef RBA_L
  return rba_launch_status();
}

template <int KMAX>
int launch_up4(const float* low, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int K, int h, int w,
               int crop_h, int crop_w, hipStream_t st, int mode) {
  const int wq = (crop_w + 3) / 4;
  const int threads = wq >= 256 ? 256 : (wq >= 128 ? 128 : 64);
  dim3 grid((wq + threads - 1) / threads, crop_h);
#define RBA_L(S, A) \
  hipLaunchKernelGGL((rba_reduce_up4_kernel<KMAX, S, A>), grid, dim3(threads), 0, st, low, prob, rba, sem, argmax, Q, K, h, w, crop_h, crop_w, wq, mode)
  if (sem && argmax) RBA_L(true, true);
  else if (sem) RBA_L(true, false);
  else if (argmax) RBA_L(false, true);
  else RBA_L(false, false);
#undef RBA_L
  return rba_launch_status();
}


// ------------------------------------------------------------------------------------------------------------
// K1 on the matrix pipe with wave-private LDS transposition ("wl").  Diagnosis (profiles/r01_k1_bandwidth_probes.txt):
// the VALU kernel is VALU-bound -- 76 fp32 FMAs + 4 sigmoids per lane per plane cost ~165 us whether or not they depend
// on the loaded data, while its load pattern alone streams in 128 us.  So the contraction moves to the matrix pipe, but
// the loads keep the good pattern (one wave-load = 1 KiB of ONE plane, ring of 2): a wave takes planes q..q+3 one at a
// time, writes sigmoid(mask) for its 256 pixels into a 4 KiB wave-private LDS tile [4 planes][256 px] (no workgroup
// barrier: LDS ops of one wave complete in order), then reads the MFMA B operand back transposed -- lane (k = lane/16,
// j = lane%16) reads plane k, pixel 16 g + j -- and issues 16 v_mfma_f32_16x16x4_f32 (16 pixel groups x 4 queries, classes
// 0..15 on the rows).  Classes 16..18 stay on VALU with scalar-register probabilities (the plane index is wave-uniform).
template <int KX, int U>
__global__ __launch_bounds__(256, 4) void rba_reduce_mfma_wl_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                                    float* __restrict__ rba, int Q, int K, int64_t HW, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int QP = (Q + 3) & ~3;
  float* Pm = lds;                                                    // [QP][16] classes 0..15 (zero rows beyond Q)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kk = lane >> 4;
  float* Sw = lds + QP * 16 + wave * 1024;                            // this wave's [4][256] tile
  for (int i = threadIdx.x; i < QP * 16; i += 256) {
    const int q = i >> 4, c = i & 15;
    Pm[i] = (q < Q && c < K) ? prob[q * K + c] : 0.f;
  }
  __syncthreads();
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t p0 = ((int64_t)tile * 4 + wave) * 256 + 4 * lane;   // this lane's 4 pixels (load side)
    const bool active = p0 < HW;
    const float* mp = mask + (active ? p0 : 0);
    f32x4_m acc[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = (f32x4_m){0.f, 0.f, 0.f, 0.f};
    float ex[KX > 0 ? KX : 1][4];
#pragma unroll
    for (int e = 0; e < (KX > 0 ? KX : 1); ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) ex[e][i] = 0.f;
    f32x4 buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)(u < Q ? u : Q - 1) * HW));
    for (int q0 = 0; q0 < QP; q0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + u;
        const f32x4 m4 = buf[u % U];
        const int qn = q + U < Q ? q + U : Q - 1;
        buf[u % U] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)qn * HW));
        f32x4 sg;
#pragma unroll
        for (int i = 0; i < 4; ++i) sg[i] = rba_sigmoid(m4[i]);
        *reinterpret_cast<f32x4*>(Sw + u * 256 + 4 * lane) = sg;
        if (KX > 0 && q < Q) {                                        // wave-uniform
          const float* pq = prob + q * K + 16;
#pragma unroll
          for (int e = 0; e < KX; ++e) {
            const float pe = pq[e];
#pragma unroll
            for (int i = 0; i < 4; ++i) ex[e][i] = fmaf(pe, sg[i], ex[e][i]);
          }
        }
      }
      const float a = Pm[(q0 + kk) * 16 + l15];
      const float* sb = Sw + kk * 256 + l15;
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sb[16 * g], acc[g], 0, 0, 0);
    }
    // acc[g][r] = sem[class 4 kk + r][pixel 16 g + l15]: tanh-sum over the lane's 4 classes, then over the 4 lane groups
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float tsum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) tsum += rba_tanh(acc[g][r]);
      tsum += __shfl_xor(tsum, 16, RBA_WAVE);
      tsum += __shfl_xor(tsum, 32, RBA_WAVE);
      if (kk == 0) Sw[16 * g + l15] = tsum;                           // re-use the tile: totals by pixel
    }
    const f32x4 t4 = *reinterpret_cast<const f32x4*>(Sw + 4 * lane);   // same wave wrote it: in-order LDS, no barrier needed
    float r4[4] = {t4[0], t4[1], t4[2], t4[3]};
    if (KX > 0) {
#pragma unroll
      for (int e = 0; e < KX; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) r4[i] += rba_tanh(ex[e][i]);
    }
    if (active) *reinterpret_cast<f32x4*>(rba + p0) = (f32x4){-r4[0], -r4[1], -r4[2], -r4[3]};
  }
}

template <int KX, int U>
int launch_reduce_mfma_wl(const float* mask, const float* prob, float* rba, int Q, int K, int64_t HW, int bpc, hipStream_t st) {
  const int64_t tiles = (HW + 1023) / 1024;
  if (tiles > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  const int64_t cap = 256LL * bpc;
  int64_t grid = tiles;
  if (tiles > cap) { const int64_t rounds = (tiles + cap - 1) / cap; grid = (tiles + rounds - 1) / rounds; }
  const size_t shm = ((size_t)((Q + 3) & ~3) * 16 + 4 * 1024) * sizeof(float);
  if (shm > 64 * 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((rba_reduce_mfma_wl_kernel<KX, U>), dim3((unsigned)grid), dim3(256), shm, st, mask, prob, rba, Q, K, HW, (int)tiles);
  return rba_launch_status();
}



}  // namespace rba_k1
