// K1 -- the RbA "rejected by all" reduction (reference: mask2former/maskformer_model.py:381-386 +
// evaluate_ood.py:143-150 + support.py:385-388), and its variant fused with the x4 mask upsample.
//
//   sem[k,p] = sum_q P[q,k] * sigmoid(mask[q,p]);  rba[p] = -sum_k tanh(sem[k,p]);  argmax[p] = argmax_k sem
//
// HBM-bound scan: every mask plane is read exactly once with 16 B/lane coalesced loads, the Q x K class
// probabilities are wave-uniform (scalar loads -> SGPR operands of the FMAs), K accumulators per pixel
// live in VGPRs, nothing but rba (and optionally sem_seg / argmax) is written.  Algorithmic bytes per
// launch: 4*Q*HW + 4*Q*K + 4*HW (+ 4*K*HW with sem_seg, + 4*HW with argmax).
#include <stdlib.h>
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

// Score-only fast path on the matrix pipe (16 <= K <= 20).  The contraction sem[k,p] = sum_q P[q,k] s[q,p] is a
// [K x Q] . [Q x pixels] product: classes 0..15 are the 16 rows of v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma
// chain = ascending q, 100 % of the tile used), the K-16 remaining classes stay on VALU.  VALU is left with the
// sigmoids only, so it no longer competes with the HBM stream for issue time, and ~40 VGPRs allow 8 waves/SIMD of
// loads in flight.  One wave = 64 consecutive pixels x all queries; per step it loads ONE float4 per lane =
// 4 planes (q0 + lane/16) x 64 pixels (four 256-B segments), runs 4 MFMAs (one per float4 component: B column j is
// pixel 4j+i) and 4*KX VALU FMAs.  A operand P[q][class] and the extra-class probabilities come from an 8 KB LDS table.
typedef float f32x4_m __attribute__((ext_vector_type(4)));

template <int KX, int U>
__global__ __launch_bounds__(256) void rba_reduce_mfma_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                              float* __restrict__ rba, int Q, int K, int64_t HW,
                                                              int64_t ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int QP = (Q + 3) & ~3;
  float* Pm = lds;                 // [QP][16]  classes 0..15
  float* Px = lds + QP * 16;       // [QP][4]   classes 16..19 (zero padded)
  for (int i = threadIdx.x; i < QP * 16; i += 256) {
    const int q = i >> 4, c = i & 15;
    Pm[i] = (q < Q && c < K) ? prob[q * K + c] : 0.f;
  }
  for (int i = threadIdx.x; i < QP * 4; i += 256) {
    const int q = i >> 2, c = 16 + (i & 3);
    Px[i] = (q < Q && c < K) ? prob[q * K + c] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, l15 = lane & 15, kk = lane >> 4;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const int steps = QP / 4;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t p = tile * 64 + 4 * l15;
    const bool active = p < HW;                          // HW % 4 == 0: a lane's 4 pixels are in or out together
    const float* mp = mask + (active ? p : 0);
    f32x4_m acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4_m){0.f, 0.f, 0.f, 0.f};
    float ex[KX > 0 ? KX : 1][4];
#pragma unroll
    for (int e = 0; e < (KX > 0 ? KX : 1); ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) ex[e][i] = 0.f;
    f32x4 buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int q = 4 * u + kk;
      q = q < Q ? q : Q - 1;
      buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)q * HW));
    }
    for (int t0 = 0; t0 < steps; t0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u;
        if (t < steps) {
          const f32x4 m4 = buf[u];
          int qn = 4 * (t + U) + kk;                     // clamped prefetch; padded queries have P = 0
          qn = qn < Q ? qn : Q - 1;
          buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)qn * HW));
          const float a = Pm[(4 * t + kk) * 16 + l15];
          float sg[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) sg[i] = rba_sigmoid(m4[i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sg[i], acc[i], 0, 0, 0);
          if (KX > 0) {
            const float4 px = *reinterpret_cast<const float4*>(Px + (4 * t + kk) * 4);
            const float pe[4] = {px.x, px.y, px.z, px.w};
#pragma unroll
            for (int e = 0; e < KX; ++e)
#pragma unroll
              for (int i = 0; i < 4; ++i) ex[e][i] = fmaf(pe[e], sg[i], ex[e][i]);
          }
        }
      }
    }
    // lane holds sem[class = 4*kk + r][pixel = p + i] in acc[i][r]; tanh-sum over its 4 classes, then over kk
    float r4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float tsum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) tsum += rba_tanh(acc[i][r]);
      tsum += __shfl_xor(tsum, 16, RBA_WAVE);
      tsum += __shfl_xor(tsum, 32, RBA_WAVE);
      r4[i] = tsum;
    }
    if (KX > 0) {
#pragma unroll
      for (int e = 0; e < KX; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = ex[e][i];                              // partial over queries q = kk (mod 4)
          v += __shfl_xor(v, 16, RBA_WAVE);
          v += __shfl_xor(v, 32, RBA_WAVE);
          r4[i] += rba_tanh(v);
        }
    }
    if (active && kk == 0) *reinterpret_cast<f32x4*>(rba + p) = (f32x4){-r4[0], -r4[1], -r4[2], -r4[3]};
  }
}

template <int KX, int U>
int launch_reduce_mfma(const float* mask, const float* prob, float* rba, int Q, int K, int64_t HW, int wps, hipStream_t st) {
  const int64_t ntiles = (HW + 63) / 64;
  const int64_t nblk_needed = (ntiles + 3) / 4;
  int64_t grid = 256LL * wps;                           // persistent: wps blocks (4 waves) per CU
  if (grid > nblk_needed) grid = nblk_needed;
  const size_t shm = (size_t)((Q + 3) & ~3) * 20 * sizeof(float);
  if (shm > 64 * 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((rba_reduce_mfma_kernel<KX, U>), dim3((unsigned)grid), dim3(256), shm, st, mask, prob, rba, Q, K, HW, ntiles);
  return rba_launch_status();
}

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

}  // namespace

// defined further down (matrix-pipe K1 with wave-private LDS transposition)
template <int KX, int U>
int launch_reduce_mfma_wl(const float* mask, const float* prob, float* rba, int Q, int K, int64_t HW, int bpc, hipStream_t st);

extern "C" int rba_hip_version(void) { return 120; }

extern "C" int rba_reduce_f32(const float* mask, const float* cls_prob, float* rba, float* sem_seg, int32_t* argmax,
                              int Q, int K, int64_t HW, int score_mode, void* stream) {
  RBA_CHECK_ARG(Q >= 1 && K >= 1 && K <= 160 && HW >= 0 && score_mode >= 0 && score_mode <= 2);
  const int mode = score_mode;
  if (HW == 0) return 0;
  RBA_CHECK_ARG(mask && cls_prob && rba);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const bool vec4 = (HW % 4 == 0) && ((((uintptr_t)mask | (uintptr_t)rba | (uintptr_t)sem_seg) & 15) == 0);
  // opt-in (RBA_K1_VARIANT=mfma, A/B hook): score-only RbA with 16 <= K <= 20 on the matrix pipe (wave-private LDS
  // transposition).  Inside the pipeline it measured 195 us against 172 us for the VALU kernel below, so it is not the default.
  static const bool k1_mfma = getenv("RBA_K1_VARIANT") && getenv("RBA_K1_VARIANT")[0] == 'm';
  if (k1_mfma && vec4 && mode == 0 && !sem_seg && !argmax && K >= 16 && K <= 20 && Q <= 800) {
    switch (K - 16) {
      case 0: return launch_reduce_mfma_wl<0, 2>(mask, cls_prob, rba, Q, K, HW, 8, st);
      case 1: return launch_reduce_mfma_wl<1, 2>(mask, cls_prob, rba, Q, K, HW, 8, st);
      case 2: return launch_reduce_mfma_wl<2, 2>(mask, cls_prob, rba, Q, K, HW, 8, st);
      case 3: return launch_reduce_mfma_wl<3, 2>(mask, cls_prob, rba, Q, K, HW, 8, st);
      default: return launch_reduce_mfma_wl<4, 2>(mask, cls_prob, rba, Q, K, HW, 8, st);
    }
  }
  if (K == 19 && vec4) return launch_reduce_fast<19, 4, 2, 4>(mask, cls_prob, rba, sem_seg, argmax, Q, HW, st, mode);
  if (K == 20 && vec4) return launch_reduce_fast<20, 4, 2, 4>(mask, cls_prob, rba, sem_seg, argmax, Q, HW, st, mode);
  if (K <= 32 && vec4) return launch_reduce<32, 4>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st, mode);
  if (K <= 32) return launch_reduce<32, 1>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st, mode);
  if (K <= 80) return launch_reduce<80, 1>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st, mode);
  return launch_reduce<160, 1>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st, mode);
}

extern "C" int rba_reduce_up4_f32(const float* mask_lowres, const float* cls_prob, float* rba, float* sem_seg,
                                  int32_t* argmax, int Q, int K, int h, int w, int crop_h, int crop_w, int score_mode,
                                  void* stream) {
  RBA_CHECK_ARG(Q >= 1 && K >= 1 && K <= 32 && h >= 1 && w >= 1 && score_mode >= 0 && score_mode <= 2);
  RBA_CHECK_ARG(crop_h >= 0 && crop_w >= 0 && crop_h <= 4 * h && crop_w <= 4 * w && crop_h <= 65535);
  if (crop_h == 0 || crop_w == 0) return 0;
  RBA_CHECK_ARG(mask_lowres && cls_prob && rba);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  // vector stores need 16 B aligned rows; the kernel falls back to scalar stores when crop_w % 4 != 0
  RBA_CHECK_ARG((((uintptr_t)rba | (uintptr_t)sem_seg) & 15) == 0);
  if (K == 19) return launch_up4<19>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, K, h, w, crop_h, crop_w, st, score_mode);
  return launch_up4<32>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, K, h, w, crop_h, crop_w, st, score_mode);
}

// LDS-staged matrix-pipe variant.  Bandwidth probes (profiles/r01_k1_bandwidth_probes.txt) show this buffer streams at
// 7.0 TB/s when every wave has ONE 1 KiB load of ONE plane in flight at 8 waves/SIMD, and loses 10-25 % for every
// additional plane concurrently in flight.  So: wave w of a 4-wave block loads 1 KiB (256 px) of plane 4t+w, applies
// the sigmoid to its own 4 values and parks them in LDS [4 planes][256 px]; after one barrier each wave reads the MFMA
// B fragments of its 64-pixel quarter (lane (k,j): 4 pixels of plane 4t+k, one ds_read_b128) and runs 4 MFMAs.
// ~45 VGPRs -> 8 waves/SIMD.  LDS is double buffered, one barrier per 4 queries.
template <int KX>
__global__ __launch_bounds__(256, 8) void rba_reduce_mfma_lds_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                                     float* __restrict__ rba, int Q, int K, int64_t HW,
                                                                     int64_t ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int QP = (Q + 3) & ~3;
  float* Pm = lds;                     // [QP][16]
  float* Px = lds + QP * 16;           // [QP][4]
  float* Sb = lds + QP * 20;           // [2][4][256]
  for (int i = threadIdx.x; i < QP * 16; i += 256) {
    const int q = i >> 4, c = i & 15;
    Pm[i] = (q < Q && c < K) ? prob[q * K + c] : 0.f;
  }
  for (int i = threadIdx.x; i < QP * 4; i += 256) {
    const int q = i >> 2, c = 16 + (i & 3);
    Px[i] = (q < Q && c < K) ? prob[q * K + c] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kk = lane >> 4;
  const int steps = QP / 4;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // load side: this lane streams pixels pl..pl+3 of planes wave, wave+4, ...
    const int64_t pl = tile * 256 + 4 * lane;
    const float* mp = mask + (pl < HW ? pl : 0);
    // compute side: this lane owns output pixels pc..pc+3 (within the wave's 64-pixel quarter)
    const int64_t pc = tile * 256 + 64 * wave + 4 * l15;
    f32x4_m acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4_m){0.f, 0.f, 0.f, 0.f};
    float ex[KX > 0 ? KX : 1][4];
#pragma unroll
    for (int e = 0; e < (KX > 0 ? KX : 1); ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) ex[e][i] = 0.f;
    f32x4 cur = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)(wave < Q ? wave : Q - 1) * HW));
    for (int t = 0; t < steps; ++t) {
      f32x4 sg;
#pragma unroll
      for (int i = 0; i < 4; ++i) sg[i] = rba_sigmoid(cur[i]);
      int qn = 4 * (t + 1) + wave;
      qn = qn < Q ? qn : Q - 1;
      cur = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)qn * HW));   // next plane, in flight over the barrier
      float* sb = Sb + (t & 1) * 1024;
      *reinterpret_cast<f32x4*>(sb + wave * 256 + 4 * lane) = sg;
      __syncthreads();
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb + kk * 256 + 64 * wave + 4 * l15);
      const float a = Pm[(4 * t + kk) * 16 + l15];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b4[i], acc[i], 0, 0, 0);
      if (KX > 0) {
        const float4 px = *reinterpret_cast<const float4*>(Px + (4 * t + kk) * 4);
        const float pe[4] = {px.x, px.y, px.z, px.w};
#pragma unroll
        for (int e = 0; e < KX; ++e)
#pragma unroll
          for (int i = 0; i < 4; ++i) ex[e][i] = fmaf(pe[e], b4[i], ex[e][i]);
      }
    }
    float r4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float tsum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) tsum += rba_tanh(acc[i][r]);
      tsum += __shfl_xor(tsum, 16, RBA_WAVE);
      tsum += __shfl_xor(tsum, 32, RBA_WAVE);
      r4[i] = tsum;
    }
    if (KX > 0) {
#pragma unroll
      for (int e = 0; e < KX; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = ex[e][i];
          v += __shfl_xor(v, 16, RBA_WAVE);
          v += __shfl_xor(v, 32, RBA_WAVE);
          r4[i] += rba_tanh(v);
        }
    }
    if (pc < HW && kk == 0) *reinterpret_cast<f32x4*>(rba + pc) = (f32x4){-r4[0], -r4[1], -r4[2], -r4[3]};
    __syncthreads();     // the next tile's first LDS write must not overtake this tile's last reads
  }
}

template <int KX>
int launch_reduce_mfma_lds(const float* mask, const float* prob, float* rba, int Q, int K, int64_t HW, int bpc, hipStream_t st) {
  const int64_t ntiles = (HW + 255) / 256;
  int64_t grid = 256LL * bpc;
  if (grid > ntiles) grid = ntiles;
  const size_t shm = ((size_t)((Q + 3) & ~3) * 20 + 2 * 1024) * sizeof(float);
  if (shm > 64 * 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((rba_reduce_mfma_lds_kernel<KX>), dim3((unsigned)grid), dim3(256), shm, st, mask, prob, rba, Q, K, HW, ntiles);
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

// "wl2": as wl, but software-pipelined inside the wave: the 16 MFMAs of plane group n-1 (B operands preloaded from LDS tile
// (n-1)&1) are issued four at a time between the sigmoid/VALU work of the four planes of group n (written to tile n&1), so the
// matrix pipe runs under the VALU work instead of after it.
template <int KX, int U>
__global__ __launch_bounds__(256, 4) void rba_reduce_mfma_wl2_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                                     float* __restrict__ rba, int Q, int K, int64_t HW, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int QP = (Q + 3) & ~3;
  float* Pm = lds;                                                    // [QP + 4][16], zero rows beyond Q
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kk = lane >> 4;
  float* Sw = lds + (QP + 4) * 16 + wave * 2048;                      // this wave's two [4][256] tiles
  for (int i = threadIdx.x; i < (QP + 4) * 16; i += 256) {
    const int q = i >> 4, c = i & 15;
    Pm[i] = (q < Q && c < K) ? prob[q * K + c] : 0.f;
  }
  __syncthreads();
  const int G = QP / 4;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t p0 = ((int64_t)tile * 4 + wave) * 256 + 4 * lane;
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
    // n = 0 .. G: iteration n does the VALU work of group n (if n < G) and the MFMAs of group n-1 (if n > 0)
    for (int n = 0; n <= G; ++n) {
      float a = 0.f, b[16];
      const bool do_mma = n > 0, do_valu = n < G;
      if (do_mma) {
        a = Pm[((n - 1) * 4 + kk) * 16 + l15];
        const float* sb = Sw + ((n - 1) & 1) * 1024 + kk * 256 + l15;
#pragma unroll
        for (int g = 0; g < 16; ++g) b[g] = sb[16 * g];
      }
      float* sw = Sw + (n & 1) * 1024;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (do_valu) {
          const int q = n * 4 + u;
          const f32x4 m4 = buf[u % U];
          const int qn = q + U < Q ? q + U : Q - 1;
          buf[u % U] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)qn * HW));
          f32x4 sg;
#pragma unroll
          for (int i = 0; i < 4; ++i) sg[i] = rba_sigmoid(m4[i]);
          *reinterpret_cast<f32x4*>(sw + u * 256 + 4 * lane) = sg;
          if (KX > 0 && q < Q) {
            const float* pq = prob + q * K + 16;
#pragma unroll
            for (int e = 0; e < KX; ++e) {
              const float pe = pq[e];
#pragma unroll
              for (int i = 0; i < 4; ++i) ex[e][i] = fmaf(pe, sg[i], ex[e][i]);
            }
          }
        }
        if (do_mma) {
#pragma unroll
          for (int g = 4 * u; g < 4 * u + 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[g], acc[g], 0, 0, 0);
        }
      }
    }
    float* st = Sw;                                                    // totals by pixel, re-using tile 0
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float tsum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) tsum += rba_tanh(acc[g][r]);
      tsum += __shfl_xor(tsum, 16, RBA_WAVE);
      tsum += __shfl_xor(tsum, 32, RBA_WAVE);
      if (kk == 0) st[16 * g + l15] = tsum;
    }
    const f32x4 t4 = *reinterpret_cast<const f32x4*>(st + 4 * lane);
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
int launch_reduce_mfma_wl2(const float* mask, const float* prob, float* rba, int Q, int K, int64_t HW, int bpc, hipStream_t st) {
  const int64_t tiles = (HW + 1023) / 1024;
  if (tiles > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  const int64_t cap = 256LL * bpc;
  int64_t grid = tiles;
  if (tiles > cap) { const int64_t rounds = (tiles + cap - 1) / cap; grid = (tiles + rounds - 1) / rounds; }
  const size_t shm = ((size_t)(((Q + 3) & ~3) + 4) * 16 + 4 * 2048) * sizeof(float);
  if (shm > 64 * 1024) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((rba_reduce_mfma_wl2_kernel<KX, U>), dim3((unsigned)grid), dim3(256), shm, st, mask, prob, rba, Q, K, HW, (int)tiles);
  return rba_launch_status();
}

// Bandwidth probes (tuning only): same plane-by-plane access pattern as the fast kernel, trivial math.
template <int VEC, int U, int WPS, int TPB>
__global__ __launch_bounds__(TPB, (WPS * 256 + TPB - 1) / TPB) void rba_bw_probe_kernel(const float* __restrict__ mask, float* __restrict__ rba,
                                                                          int Q, int64_t HW, int tiles) {
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t p0 = ((int64_t)tile * TPB + threadIdx.x) * VEC;
    if (p0 >= HW) continue;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    const float* mp = mask + p0;
    float buf[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) load_vec<VEC>(mp + (int64_t)(u < Q ? u : Q - 1) * HW, buf[u]);
    for (int q0 = 0; q0 + U <= Q; q0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += buf[u][i];
        const int qn = q0 + u + U < Q ? q0 + u + U : Q - 1;
        load_vec<VEC>(mp + (int64_t)qn * HW, buf[u]);
      }
    }
    store_vec<VEC>(rba + p0, acc);
  }
}
template <int VEC, int U, int WPS, int TPB>
int launch_bw_probe(const float* mask, float* rba, int Q, int64_t HW, hipStream_t st) {
  const int64_t per_block = TPB * (int64_t)VEC;
  const int64_t tiles = (HW + per_block - 1) / per_block;
  const int64_t cap = 256LL * WPS * 256 / TPB;
  int64_t grid = tiles;
  if (tiles > cap) { const int64_t rounds = (tiles + cap - 1) / cap; grid = (tiles + rounds - 1) / rounds; }
  hipLaunchKernelGGL((rba_bw_probe_kernel<VEC, U, WPS, TPB>), dim3((unsigned)grid), dim3(TPB), 0, st, mask, rba, Q, HW, (int)tiles);
  return rba_launch_status();
}
// plane-major probe: the whole grid sweeps plane 0, then plane 1, ... (what a [Q,HW] -> [HW] reduction would look like
// if accumulators lived in memory); only to see the raw streaming rate of this buffer
__global__ __launch_bounds__(256) void rba_bw_linear_kernel(const float* __restrict__ mask, float* __restrict__ rba, int64_t n4) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4* m4 = reinterpret_cast<const f32x4*>(mask);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    acc += __builtin_nontemporal_load(m4 + i);
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) rba[0] = acc.x;
}

// probe of the MFMA kernel's load pattern: lane group g = lane/16 reads plane q0+g, 64 px per wave per load
template <int U>
__global__ __launch_bounds__(256) void rba_bw_probe4_kernel(const float* __restrict__ mask, float* __restrict__ rba, int Q, int64_t HW,
                                                           int64_t ntiles) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, kk = lane >> 4;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t p = tile * 64 + 4 * l15;
    const float* mp = mask + p;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)(4 * u + kk) * HW));
    for (int t0 = 0; t0 < Q / 4; t0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc += buf[u];
        int qn = 4 * (t0 + u + U) + kk;
        qn = qn < Q ? qn : Q - 1;
        buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mp + (int64_t)qn * HW));
      }
    }
    if (kk == 0) *reinterpret_cast<f32x4*>(rba + p) = acc;
  }
}
// probe: K1's loads + K1's VALU work, but the VALU work does not depend on the loaded data (DEP = false) or does (DEP = true)
template <int U, int WPS, bool DEP, int WORK = 0>
__global__ __launch_bounds__(256, WPS) void rba_valu_probe_kernel(const float* __restrict__ mask, const float* __restrict__ prob,
                                                                  float* __restrict__ rba, int Q, int64_t HW, int tiles) {
  constexpr int K = 19, VEC = 4;
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
    float fake[VEC] = {0.1f * threadIdx.x, 0.2f, 0.3f, 0.4f};
    float sink = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) load_vec<VEC>(mp + (int64_t)(u < Q ? u : Q - 1) * HW, buf[u]);
    for (int q0 = 0; q0 + U <= Q; q0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = q0 + u;
        float s[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          s[i] = WORK == 1 ? (DEP ? buf[u][i] : fake[i]) : rba_sigmoid(DEP ? buf[u][i] : fake[i]);
          if (!DEP) { sink += buf[u][i]; fake[i] += 1e-3f; }
        }
        const int qn = q + U < Q ? q + U : Q - 1;
        load_vec<VEC>(mp + (int64_t)qn * HW, buf[u]);
        const float* pq = prob + q * K;
        if (WORK == 2) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[0][i] += s[i];
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const float pk = pq[k];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[k][i] = fmaf(pk, s[i], acc[k][i]);
          }
        }
      }
    }
    float r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { r[i] = sink; }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int i = 0; i < VEC; ++i) r[i] += acc[k][i];
    store_vec<VEC>(rba + p0, r);
  }
}
template <int U, int WPS, bool DEP, int WORK = 0>
int launch_valu_probe(const float* mask, const float* prob, float* rba, int Q, int64_t HW, hipStream_t st) {
  const int64_t tiles = (HW + 1023) / 1024;
  const int64_t cap = 256LL * WPS;
  int64_t grid = tiles;
  if (tiles > cap) { const int64_t rounds = (tiles + cap - 1) / cap; grid = (tiles + rounds - 1) / rounds; }
  hipLaunchKernelGGL((rba_valu_probe_kernel<U, WPS, DEP, WORK>), dim3((unsigned)grid), dim3(256), 0, st, mask, prob, rba, Q, HW, (int)tiles);
  return rba_launch_status();
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
    case 12: return launch_reduce_fast<19, 4, 1, 4>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 13: return launch_reduce_fast<19, 4, 1, 5>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 14: return launch_reduce_fast<19, 4, 2, 5>(mask, cls_prob, rba, nullptr, nullptr, Q, HW, st);
    case 82: return launch_valu_probe<2, 4, true, 1>(mask, cls_prob, rba, Q, HW, st);
    case 83: return launch_valu_probe<2, 4, true, 2>(mask, cls_prob, rba, Q, HW, st);
    case 80: return launch_valu_probe<2, 4, false>(mask, cls_prob, rba, Q, HW, st);
    case 81: return launch_valu_probe<2, 4, true>(mask, cls_prob, rba, Q, HW, st);
    case 90: return launch_reduce_mfma_wl<3, 2>(mask, cls_prob, rba, Q, 19, HW, 4, st);
    case 91: return launch_reduce_mfma_wl<3, 1>(mask, cls_prob, rba, Q, 19, HW, 4, st);
    case 92: return launch_reduce_mfma_wl<3, 4>(mask, cls_prob, rba, Q, 19, HW, 4, st);
    case 93: return launch_reduce_mfma_wl<3, 2>(mask, cls_prob, rba, Q, 19, HW, 3, st);
    case 94: return launch_reduce_mfma_wl<3, 2>(mask, cls_prob, rba, Q, 19, HW, 8, st);
    case 95: return launch_reduce_mfma_wl2<3, 2>(mask, cls_prob, rba, Q, 19, HW, 4, st);
    case 96: return launch_reduce_mfma_wl2<3, 2>(mask, cls_prob, rba, Q, 19, HW, 8, st);
    case 97: return launch_reduce_mfma_wl2<3, 4>(mask, cls_prob, rba, Q, 19, HW, 4, st);
    case 50: return launch_reduce_mfma_lds<3>(mask, cls_prob, rba, Q, 19, HW, 8, st);
    case 51: return launch_reduce_mfma_lds<3>(mask, cls_prob, rba, Q, 19, HW, 4, st);
    case 52: return launch_reduce_mfma_lds<3>(mask, cls_prob, rba, Q, 19, HW, 6, st);
    case 53: return launch_reduce_mfma_lds<3>(mask, cls_prob, rba, Q, 19, HW, 16, st);
    case 54: return launch_reduce_mfma_lds<3>(mask, cls_prob, rba, Q, 19, HW, 32, st);
    case 40: { hipLaunchKernelGGL(rba_bw_probe4_kernel<2>, dim3(256 * 4), dim3(256), 0, st, mask, rba, Q, HW, (HW + 63) / 64); return rba_launch_status(); }
    case 41: { hipLaunchKernelGGL(rba_bw_probe4_kernel<2>, dim3(256 * 8), dim3(256), 0, st, mask, rba, Q, HW, (HW + 63) / 64); return rba_launch_status(); }
    case 42: { hipLaunchKernelGGL(rba_bw_probe4_kernel<1>, dim3(256 * 8), dim3(256), 0, st, mask, rba, Q, HW, (HW + 63) / 64); return rba_launch_status(); }
    case 43: { hipLaunchKernelGGL(rba_bw_probe4_kernel<4>, dim3(256 * 2), dim3(256), 0, st, mask, rba, Q, HW, (HW + 63) / 64); return rba_launch_status(); }
    case 44: return launch_bw_probe<4, 2, 2, 256>(mask, rba, Q, HW, st);
    case 45: return launch_bw_probe<4, 2, 6, 256>(mask, rba, Q, HW, st);
    case 46: return launch_bw_probe<4, 2, 8, 256>(mask, rba, Q, HW, st);
    case 47: return launch_bw_probe<4, 1, 8, 256>(mask, rba, Q, HW, st);
    case 48: return launch_bw_probe<4, 4, 4, 256>(mask, rba, Q, HW, st);
    case 49: return launch_bw_probe<4, 1, 4, 256>(mask, rba, Q, HW, st);
    case 30: return launch_bw_probe<4, 2, 4, 256>(mask, rba, Q, HW, st);
    case 31: return launch_bw_probe<4, 4, 8, 256>(mask, rba, Q, HW, st);
    case 32: return launch_bw_probe<4, 8, 8, 256>(mask, rba, Q, HW, st);
    case 33: return launch_bw_probe<4, 4, 8, 512>(mask, rba, Q, HW, st);
    case 34: return launch_bw_probe<4, 4, 8, 1024>(mask, rba, Q, HW, st);
    case 35: return launch_bw_probe<2, 8, 8, 256>(mask, rba, Q, HW, st);
    case 36: { hipLaunchKernelGGL(rba_bw_linear_kernel, dim3(2048), dim3(256), 0, st, mask, rba, (int64_t)Q * HW / 4); return rba_launch_status(); }
    case 37: { hipLaunchKernelGGL(rba_bw_linear_kernel, dim3(8192), dim3(256), 0, st, mask, rba, (int64_t)Q * HW / 4); return rba_launch_status(); }
    case 20: return launch_reduce_mfma<3, 2>(mask, cls_prob, rba, Q, 19, HW, 8, st);
    case 21: return launch_reduce_mfma<3, 4>(mask, cls_prob, rba, Q, 19, HW, 8, st);
    case 22: return launch_reduce_mfma<3, 2>(mask, cls_prob, rba, Q, 19, HW, 6, st);
    case 23: return launch_reduce_mfma<3, 4>(mask, cls_prob, rba, Q, 19, HW, 4, st);
    case 24: return launch_reduce_mfma<3, 1>(mask, cls_prob, rba, Q, 19, HW, 8, st);
    case 25: return launch_reduce_mfma<3, 3>(mask, cls_prob, rba, Q, 19, HW, 8, st);
    case 26: return launch_reduce_mfma<3, 4>(mask, cls_prob, rba, Q, 19, HW, 6, st);
    case 27: return launch_reduce_mfma<3, 2>(mask, cls_prob, rba, Q, 19, HW, 16, st);
    default: return (int)hipErrorInvalidValue;
  }
}
