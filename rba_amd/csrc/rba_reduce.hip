// K1 -- the RbA "rejected by all" reduction (reference: mask2former/maskformer_model.py:381-386 +
// evaluate_ood.py:143-150 + support.py:385-388), and its variant fused with the x4 mask upsample.
//
//   sem[k,p] = sum_q P[q,k] * sigmoid(mask[q,p]);  rba[p] = -sum_k tanh(sem[k,p]);  argmax[p] = argmax_k sem
//
// HBM-bound scan: every mask plane is read exactly once with 16 B/lane coalesced loads, the Q x K class
// probabilities are wave-uniform (scalar loads -> SGPR operands of the FMAs), K accumulators per pixel
// live in VGPRs, nothing but rba (and optionally sem_seg / argmax) is written.  Algorithmic bytes per
// launch: 4*Q*HW + 4*Q*K + 4*HW (+ 4*K*HW with sem_seg, + 4*HW with argmax).
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

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
template <int KMAX, int VEC, bool SEM, bool ARG>
__device__ __forceinline__ void rba_epilogue(float (&acc)[KMAX][VEC], int K, float* rba, float* sem, int32_t* argmax,
                                             int64_t p0, int64_t plane) {
  float r[VEC];
  int best[VEC];
  float bestv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { r[i] = 0.f; best[i] = 0; bestv[i] = acc[0][i]; }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        r[i] -= rba_tanh(acc[k][i]);
        if (ARG && acc[k][i] > bestv[i]) { bestv[i] = acc[k][i]; best[i] = k; }
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
                                                         int32_t* __restrict__ argmax, int Q, int K, int64_t HW) {
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
  rba_epilogue<KMAX, VEC, SEM, ARG>(acc, K, rba, sem, argmax, p0, HW);
}

// x4 upsample fused in front: thread owns 4 consecutive output pixels of one output row, i.e. output
// columns 4*j .. 4*j+3 which interpolate low-res columns j-1, j, j+1 and rows (i0, i1) of the low-res map.
template <int KMAX, bool SEM, bool ARG>
__global__ __launch_bounds__(256) void rba_reduce_up4_kernel(const float* __restrict__ low, const float* __restrict__ prob,
                                                             float* __restrict__ rba, float* __restrict__ sem,
                                                             int32_t* __restrict__ argmax, int Q, int K, int h, int w,
                                                             int crop_h, int crop_w, int wq /* ceil(crop_w/4) */) {
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
    rba_epilogue<KMAX, 4, SEM, ARG>(acc, K, rba, sem, argmax, p0, oplane);
  } else {  // ragged right edge / unaligned rows: scalar stores
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * j + i >= crop_w) break;
      float r = 0.f, bv = acc[0][i];
      int b = 0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          r -= rba_tanh(acc[k][i]);
          if (acc[k][i] > bv) { bv = acc[k][i]; b = k; }
          if (SEM) sem[(int64_t)k * oplane + p0 + i] = acc[k][i];
        }
      }
      rba[p0 + i] = r;
      if (ARG) argmax[p0 + i] = b;
    }
  }
}

// Fast path: K is a compile-time constant (no per-class guards -> straight-line v_pk_fma with SGPR operands),
// a ring of U prefetched mask planes keeps U x 1 KiB loads in flight per wave (the compiler otherwise waits
// vmcnt(0) after every load), and the grid is persistent: `tiles` tiles of 256*VEC pixels are strided over a
// grid sized to whole multiples of the resident-wave capacity, so no partially filled last round.
template <int K, int VEC, bool SEM, bool ARG, int U, int WPS>
__global__ __launch_bounds__(256, WPS) void rba_reduce_fast_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                                 float* __restrict__ rba, float* __restrict__ sem,
                                                                 int32_t* __restrict__ argmax, int Q, int64_t HW, int tiles) {
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
    rba_epilogue<K, VEC, SEM, ARG>(acc, K, rba, sem, argmax, p0, HW);
  }
}

template <int K, int VEC, int U, int WPS>
int launch_reduce_fast(const float* mask, const float* prob, float* rba, float* sem, int32_t* argmax, int Q,
                       int64_t HW, hipStream_t st) {
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
  hipLaunchKernelGGL((rba_reduce_fast_kernel<K, VEC, S, A, U, WPS>), dim3((unsigned)grid), dim3(256), 0, st, mask, prob, rba, sem, argmax, Q, HW, (int)tiles)
  if (sem && argmax) RBA_L(true, true);
  else if (sem) RBA_L(true, false);
  else if (argmax) RBA_L(false, true);
  else RBA_L(false, false);
#undef RBA_L
  return rba_launch_status();
}

template <int KMAX, int VEC>
int launch_reduce(const float* mask, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int K,
                  int64_t HW, hipStream_t st) {
  const int threads = 256;
  const int64_t per_block = (int64_t)threads * VEC;
  const unsigned blocks = (unsigned)((HW + per_block - 1) / per_block);
#define RBA_L(S, A) \
  hipLaunchKernelGGL((rba_reduce_kernel<KMAX, VEC, S, A>), dim3(blocks), dim3(threads), 0, st, mask, prob, rba, sem, argmax, Q, K, HW)
  if (sem && argmax) RBA_L(true, true);
  else if (sem) RBA_L(true, false);
  else if (argmax) RBA_L(false, true);
  else RBA_L(false, false);
#undef RBA_L
  return rba_launch_status();
}

template <int KMAX>
int launch_up4(const float* low, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int K, int h, int w,
               int crop_h, int crop_w, hipStream_t st) {
  const int wq = (crop_w + 3) / 4;
  const int threads = wq >= 256 ? 256 : (wq >= 128 ? 128 : 64);
  dim3 grid((wq + threads - 1) / threads, crop_h);
#define RBA_L(S, A) \
  hipLaunchKernelGGL((rba_reduce_up4_kernel<KMAX, S, A>), grid, dim3(threads), 0, st, low, prob, rba, sem, argmax, Q, K, h, w, crop_h, crop_w, wq)
  if (sem && argmax) RBA_L(true, true);
  else if (sem) RBA_L(true, false);
  else if (argmax) RBA_L(false, true);
  else RBA_L(false, false);
#undef RBA_L
  return rba_launch_status();
}

}  // namespace

extern "C" int rba_hip_version(void) { return 100; }

extern "C" int rba_reduce_f32(const float* mask, const float* cls_prob, float* rba, float* sem_seg, int32_t* argmax,
                              int Q, int K, int64_t HW, void* stream) {
  RBA_CHECK_ARG(Q >= 1 && K >= 1 && K <= 160 && HW >= 0);
  if (HW == 0) return 0;
  RBA_CHECK_ARG(mask && cls_prob && rba);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const bool vec4 = (HW % 4 == 0) && ((((uintptr_t)mask | (uintptr_t)rba | (uintptr_t)sem_seg) & 15) == 0);
  if (K == 19 && vec4) return launch_reduce_fast<19, 4, 2, 4>(mask, cls_prob, rba, sem_seg, argmax, Q, HW, st);
  if (K == 20 && vec4) return launch_reduce_fast<20, 4, 2, 4>(mask, cls_prob, rba, sem_seg, argmax, Q, HW, st);
  if (K <= 32 && vec4) return launch_reduce<32, 4>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st);
  if (K <= 32) return launch_reduce<32, 1>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st);
  if (K <= 80) return launch_reduce<80, 1>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st);
  return launch_reduce<160, 1>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st);
}

extern "C" int rba_reduce_up4_f32(const float* mask_lowres, const float* cls_prob, float* rba, float* sem_seg,
                                  int32_t* argmax, int Q, int K, int h, int w, int crop_h, int crop_w, void* stream) {
  RBA_CHECK_ARG(Q >= 1 && K >= 1 && K <= 32 && h >= 1 && w >= 1);
  RBA_CHECK_ARG(crop_h >= 0 && crop_w >= 0 && crop_h <= 4 * h && crop_w <= 4 * w && crop_h <= 65535);
  if (crop_h == 0 || crop_w == 0) return 0;
  RBA_CHECK_ARG(mask_lowres && cls_prob && rba);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  // vector stores need 16 B aligned rows; the kernel falls back to scalar stores when crop_w % 4 != 0
  RBA_CHECK_ARG((((uintptr_t)rba | (uintptr_t)sem_seg) & 15) == 0);
  if (K == 19) return launch_up4<19>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, K, h, w, crop_h, crop_w, st);
  return launch_up4<32>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, K, h, w, crop_h, crop_w, st);
}

// Tuning hook (not part of the public ABI in include/rba_hip.h): K = 19 score-only variants of the fast kernel.
extern "C" int rba_reduce_f32_tune(const float* mask, const float* cls_prob, float* rba, int Q, int64_t HW, int variant,
                                   void* stream) {
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0: return launch_reduce_fast<19, 4, 4, 4>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 1: return launch_reduce_fast<19, 4, 2, 4>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 2: return launch_reduce_fast<19, 4, 2, 4>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 3: return launch_reduce_fast<19, 4, 3, 4>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 4: return launch_reduce_fast<19, 2, 4, 8>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 5: return launch_reduce_fast<19, 2, 8, 6>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 6: return launch_reduce_fast<19, 4, 6, 3>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 7: return launch_reduce_fast<19, 4, 8, 2>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 8: return launch_reduce<19, 4>(mask, cls_prob, rba, nullptr, nullptr, Q, 19, HW, st);
    case 9: return launch_reduce_fast<19, 2, 4, 6>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 10: return launch_reduce_fast<19, 4, 2, 3>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 11: return launch_reduce_fast<19, 1, 8, 8>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    default: return (int)hipErrorInvalidValue;
  }
}
