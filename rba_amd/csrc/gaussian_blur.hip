// Gaussian smoothing of the anomaly-score map: the reference evaluator's optional transforms.GaussianBlur(7, sigma=1)
// (support.py:366-383; torchvision semantics: 1-D kernel exp(-0.5 (x/sigma)^2) on linspace(-(k-1)/2, (k-1)/2, k) normalised
// to 1, 2-D kernel = outer product, reflect padding).  One thread per output pixel, the tile and its halo staged in LDS,
// k*k fused multiply-adds in the order of the 2-D correlation.  Pure bandwidth (one read + one write of the map).
#include "common.h"
#include "../../include/rba_hip.h"

namespace {

constexpr int TX = 64, TY = 4, KMAX = 15;

__device__ __forceinline__ int reflect(int i, int n) {            // torch "reflect": -1 -> 1, n -> n - 2
  i = i < 0 ? -i : i;
  i = i >= n ? 2 * n - 2 - i : i;
  return i < 0 ? 0 : i;                                           // beyond n - 1 + pad: only read by threads outside the image
}

__global__ __launch_bounds__(TX * TY) void gaussian_blur_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                                int k, float sigma) {
  __shared__ float tile[TY + KMAX - 1][TX + KMAX - 1];
  __shared__ float w1[KMAX];
  const int r = k >> 1, tw = TX + k - 1, th = TY + k - 1;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
  const int tid = threadIdx.y * TX + threadIdx.x;
  if (tid < k) {
    const float half = (k - 1) * 0.5f;
    float s = 0.f;
    for (int i = 0; i < k; ++i) { const float t = (-half + i) / sigma; s += expf(-0.5f * t * t); }
    const float t = (-half + tid) / sigma;
    w1[tid] = expf(-0.5f * t * t) / s;
  }
  for (int i = tid; i < tw * th; i += TX * TY) {
    const int ty = i / tw, tx = i - ty * tw;
    tile[ty][tx] = in[(int64_t)reflect(y0 + ty - r, H) * W + reflect(x0 + tx - r, W)];
  }
  __syncthreads();
  const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
  if (x >= W || y >= H) return;
  float acc = 0.f;
  for (int dy = 0; dy < k; ++dy) {
    const float wy = w1[dy];
    for (int dx = 0; dx < k; ++dx) acc = fmaf(tile[threadIdx.y + dy][threadIdx.x + dx], wy * w1[dx], acc);
  }
  out[(int64_t)y * W + x] = acc;
}

}  // namespace

extern "C" int rba_gaussian_blur_f32(const float* in, float* out, int H, int W, int kernel_size, float sigma, void* stream) {
  RBA_CHECK_ARG(H >= 0 && W >= 0 && kernel_size >= 1 && kernel_size <= KMAX && (kernel_size & 1) && sigma > 0.f);
  if (H == 0 || W == 0) return 0;
  RBA_CHECK_ARG(in && out && in != out && H > kernel_size / 2 && W > kernel_size / 2);      // reflect padding needs pad < size
  RBA_CHECK_ARG((H + TY - 1) / TY <= 65535);
  rba_begin();
  hipLaunchKernelGGL(gaussian_blur_kernel, dim3((W + TX - 1) / TX, (H + TY - 1) / TY), dim3(TX, TY), 0, (hipStream_t)stream, in, out,
                     H, W, kernel_size, sigma);
  return rba_launch_status();
}
