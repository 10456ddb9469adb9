// Probe build of the GroupNorm-folded mask-feature projection (split_linear_h3l_kernel<NCHW, GNF>), tools/gnfold_probe2.py: RBA_GNF_FORM = 0
// (0 = the product's max + poison, 1 = max(y, f) + (y - y), the form that goes wrong on MI355X), debug dump of the staged values on.  Tools only (librba_tune.so).
#define RBA_GNF_FORM 0
#define RBA_GNF_DEBUG 1
#include "../split_linear_h3.h"

extern "C" int rba_gnf_probe_form0_dbg1(const float* x, const float* mr, const float* gamma, const float* beta, int G, int relu, const void* weight_packed,
                                       const float* bias, float* out, int64_t M, int N, int K, int rows_per_image, unsigned long long* dbg, void* stream) {
  const GnFold gn{mr, gamma, beta, G, K / G, relu ? 1 : 0};
  const int64_t MT = (M + 127) / 128;
  const int NT = (N + 127) / 128;
  hipLaunchKernelGGL((split_linear_h3l_kernel<0, 4, 0, false, false, false, true, true>), dim3((unsigned)(MT * NT)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<const u32x4_t*>(weight_packed), bias, out, (int)M, N, K, (int)MT, NT, dbg, ConvShape{0, 0, 0}, nullptr, rows_per_image, gn);
  return (int)hipGetLastError();
}
