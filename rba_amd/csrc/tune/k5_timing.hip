// Timing build of K5's f16x3 kernel (tools/k5_timeline.py): the product kernel of swin_window_attn_h3.h compiled with K5H_TIMING, so
// that lane 0 of every wave stamps the 100 MHz wall clock at its phase boundaries:
//   [0] start  [1] gather loads issued  [2] gather data arrived  [3] split + LDS writes issued  [4] barrier passed
//   [5] Q split, bias fragments arrived  [6] Q K^T done  [7] softmax done  [8] P V done  [9] stores issued  [10] HW_ID | XCC_ID << 32
// Tools only (librba_tune.so); the forced waits in front of stamps 2 and 5 are the only change to the instruction stream.
#include <stdlib.h>
#define K5H_TIMING 1
#include "../common.h"

namespace {
typedef float f32x4_t __attribute__((ext_vector_type(4)));
}
#include "../swin_window_attn_h3.h"

// dbg: [workgroups = (Wp/12) * (Hp/12) * B * nH][9 waves][12] uint64
extern "C" int rba_k5_timing(const float* qkv, const float* qkv_bias, const float* bias_frag, void* out, int B, int H, int W, int nH,
                             int shift, int split_out, int wpe, unsigned long long* dbg, void* stream) {
  const int ws = 12;
  const int Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws;
  const size_t shm = (size_t)(4 * 9 * 16 * 64) + (size_t)(2 * 9 * 16) * sizeof(int);
  const dim3 grid(Wp / ws, Hp / ws, B * nH), block(64 * 9);
  const float scale = (float)(1.0 / sqrt(32.0));
  if (wpe == 6)
    hipLaunchKernelGGL((swin_window_attn_h3_kernel<9, 9, true, true, 6>), grid, block, shm, (hipStream_t)stream, qkv, qkv_bias, bias_frag,
                       reinterpret_cast<float*>(out), H, W, Hp, Wp, nH, ws, shift, scale, dbg);
  else if (split_out)
    hipLaunchKernelGGL((swin_window_attn_h3_kernel<9, 9, true, true>), grid, block, shm, (hipStream_t)stream, qkv, qkv_bias, bias_frag,
                       reinterpret_cast<float*>(out), H, W, Hp, Wp, nH, ws, shift, scale, dbg);
  else
    hipLaunchKernelGGL((swin_window_attn_h3_kernel<9, 9, true, false>), grid, block, shm, (hipStream_t)stream, qkv, qkv_bias, bias_frag,
                       reinterpret_cast<float*>(out), H, W, Hp, Wp, nH, ws, shift, scale, dbg);
  return (int)hipGetLastError();
}
