// K5's f16x3 kernel at different register budgets (tools/k5_wpe_ab.py): amdgpu_waves_per_eu(WPE, 8), WPE = 5 (96 VGPRs), 6 (80), 7 (72), 8 (64).
// Compiled as the ablation build (K5H_ABLATE: a runtime `ablate` argument, 0 = the product arithmetic).
// A gfx950 CU holds only ONE 9-wave workgroup at 96 VGPRs although 18 waves x 96 registers fit its four SIMDs on paper (tools/micro/occ_probe.hip:
// 576 threads: 96 VGPRs -> 1 workgroup per CU, 80 -> 2, 64 -> 3): the same arithmetic, different residency.  Tools only (librba_tune.so).
#include <stdlib.h>
#define K5H_ABLATE 1
#include "../common.h"

namespace {
typedef float f32x4_t __attribute__((ext_vector_type(4)));
}
#include "../swin_window_attn_h3.h"

extern "C" int rba_k5_wpe(const float* qkv, const float* qkv_bias, const float* bias_frag, void* out, int B, int H, int W, int nH, int shift,
                          int split_out, int wpe, int ablate, void* stream) {
  const int ws = 12;
  const int Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws;
  const size_t shm = (size_t)(4 * 9 * 16 * 64) + (size_t)(2 * 9 * 16) * sizeof(int);
  const dim3 grid(Wp / ws, Hp / ws, B * nH), block(64 * 9);
  const float scale = (float)(1.0 / sqrt(32.0));
  float* o = reinterpret_cast<float*>(out);
  hipStream_t st = (hipStream_t)stream;
#define K5_L(S, Wv) hipLaunchKernelGGL((swin_window_attn_h3_kernel<9, 9, true, S, Wv>), grid, block, shm, st, qkv, qkv_bias, bias_frag, o, H, W, Hp, Wp, nH, ws, shift, scale, ablate)
  if (split_out) {
    if (wpe == 5) K5_L(true, 5); else if (wpe == 6) K5_L(true, 6); else if (wpe == 7) K5_L(true, 7); else if (wpe == 8) K5_L(true, 8); else return -1;
  } else {
    if (wpe == 5) K5_L(false, 5); else if (wpe == 6) K5_L(false, 6); else if (wpe == 7) K5_L(false, 7); else if (wpe == 8) K5_L(false, 8); else return -1;
  }
#undef K5_L
  return (int)hipGetLastError();
}
