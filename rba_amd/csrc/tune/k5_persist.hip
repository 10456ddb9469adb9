// The persistent / prefetching form of K5 (swin_window_attn_h3p.h): measured against the product's one-shot kernel by tools/k5_persist_ab.py.  Tools only.
#include <stdlib.h>
#include "../common.h"

namespace {
typedef float f32x4_t __attribute__((ext_vector_type(4)));
}
#include "../swin_window_attn_h3.h"
#include "swin_window_attn_h3p.h"

extern "C" int rba_k5_persist(const float* qkv, const float* qkv_bias, const float* bias_frag, void* out, int B, int H, int W, int nH, int shift,
                              int split_out, void* stream) {
  const int Hp = (H + 11) / 12 * 12, Wp = (W + 11) / 12 * 12;
  const float scale = (float)(1.0 / sqrt(32.0));
  float* o = reinterpret_cast<float*>(out);
  return split_out ? launch_h3_persist<true>(qkv, qkv_bias, bias_frag, o, B, H, W, Hp, Wp, nH, shift, scale, (hipStream_t)stream)
                   : launch_h3_persist<false>(qkv, qkv_bias, bias_frag, o, B, H, W, Hp, Wp, nH, shift, scale, (hipStream_t)stream);
}
