// Open-set panoptic epilogue of the RbA map (reference: MaskFormer.panoptic_inference, maskformer_model.py:454-481):
//   binary = rba > threshold;  3x3 morphological opening, then closing (cv2.morphologyEx with a 3x3 box, default border = the
//   operation's neutral value, i.e. only in-image neighbours count);  4-connected components labelled in raster order
//   (cv2.connectedComponents(connectivity=4)).
// Kernels: threshold, 3x3 erode / dilate on a uint8 map, and a union-find labelling whose roots are the smallest linear index of
// each component (atomicMin unions of the left / upper neighbour, then path flattening) -- deterministic whatever the execution
// order.  Consecutive numbering of the roots in raster order is a prefix sum done by the caller.
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

__global__ void threshold_kernel(const float* __restrict__ score, uint8_t* __restrict__ out, int64_t n, float thr) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = score[i] > thr ? 1 : 0;
}

template <bool DILATE>
__global__ void morph3x3_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  bool v = !DILATE;                                               // erode: all in-image neighbours set; dilate: any
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= W) continue;
      const bool p = in[(int64_t)yy * W + xx] != 0;
      v = DILATE ? (v || p) : (v && p);
    }
  }
  out[(int64_t)y * W + x] = v ? 1 : 0;
}

__device__ __forceinline__ int uf_find(const int* L, int i) {
  // device-scope loads: parents are lowered concurrently by other CUs' atomicMin; a stale value would still be an ancestor
  // (parents only ever decrease within a component), the atomicMin in uf_union re-checks against the live value
  int p = __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != i) { i = p; p = __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  return i;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  bool done;
  do {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a < b) { const int old = atomicMin(L + b, a); done = old == b; b = old; }
    else if (b < a) { const int old = atomicMin(L + a, b); done = old == a; a = old; }
    else done = true;
  } while (!done);
}

__global__ void ccl_init_kernel(const uint8_t* __restrict__ m, int* __restrict__ L, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) L[i] = m[i] ? (int)i : -1;
}
__global__ void ccl_merge_kernel(const uint8_t* __restrict__ m, int* __restrict__ L, int H, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int i = y * W + x;
  if (!m[i]) return;
  if (x > 0 && m[i - 1]) uf_union(L, i, i - 1);
  if (y > 0 && m[i - W]) uf_union(L, i, i - W);
}
__global__ void ccl_flatten_kernel(int* __restrict__ L, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (L[i] >= 0) L[i] = uf_find(L, (int)i);
}

}  // namespace

extern "C" int rba_threshold_u8(const float* score, uint8_t* out, int64_t n, float threshold, void* stream) {
  RBA_CHECK_ARG(n >= 0);
  if (n == 0) return 0;
  RBA_CHECK_ARG(score && out);
  rba_begin();
  const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(threshold_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, score, out, n, threshold);
  return rba_launch_status();
}

extern "C" int rba_morph3x3_u8(const uint8_t* in, uint8_t* out, int H, int W, int dilate, void* stream) {
  RBA_CHECK_ARG(H >= 0 && W >= 0 && H <= 65535);
  if (H == 0 || W == 0) return 0;
  RBA_CHECK_ARG(in && out && in != out);
  rba_begin();
  const dim3 grid((W + 255) / 256, H);
  if (dilate) hipLaunchKernelGGL(morph3x3_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, out, H, W);
  else hipLaunchKernelGGL(morph3x3_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, out, H, W);
  return rba_launch_status();
}

extern "C" int rba_ccl4_roots_i32(const uint8_t* mask, int32_t* roots, int H, int W, void* stream) {
  RBA_CHECK_ARG(H >= 0 && W >= 0 && H <= 65535 && (int64_t)H * W < ((int64_t)1 << 31));
  if (H == 0 || W == 0) return 0;
  RBA_CHECK_ARG(mask && roots);
  rba_begin();
  const int64_t n = (int64_t)H * W;
  const unsigned g1 = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ccl_init_kernel, dim3(g1), dim3(256), 0, st, mask, roots, n);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3((W + 255) / 256, H), dim3(256), 0, st, mask, roots, H, W);
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3(g1), dim3(256), 0, st, roots, n);
  return rba_launch_status();
}
