// Timing build of K7 (tools/k7_timeline.py): the product kernel of swin_attn_block.h compiled with K7_TIMING, so that lane 0 of every wave stamps the
// shader clock (s_memtime) at its phase boundaries:
//   [0] start  [1] norm1 + split done  per head h: [2+5h] barrier B passed  [3+5h] q, k, v done  [4+5h] barrier A passed  [5+5h] attention done
//   [6+5h] proj done   [31] epilogue stores issued
// Tools only (librba_tune.so).
#define K7_TIMING 1
#include "../swin_attn_block.h"

// dbg: [workgroups = (Wp/12) * (Hp/12) * B][9 waves][32] uint64
extern "C" int rba_k7_timing(float* x, float* y2, const float* g1, const float* b1, float eps1, const void* img, const float* qkv_bias,
                             const float* bias_frag, const float* proj_bias, const float* g2, const float* b2, float eps2, int B, int H, int W, int C,
                             int shift, unsigned long long* dbg, void* stream) {
  if (C != 128) return (int)hipErrorInvalidValue;
  const int Hp = (H + K7_WS - 1) / K7_WS * K7_WS, Wp = (W + K7_WS - 1) / K7_WS * K7_WS;
  const float scale = (float)(1.0 / sqrt(32.0));
  const dim3 grid(Wp / K7_WS, Hp / K7_WS, B), block(64 * K7_WAVES);
  constexpr size_t shm = k7_lds_bytes(128);
  hipError_t e = hipFuncSetAttribute((const void*)swin_attn_block_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((swin_attn_block_kernel<128, true>), grid, block, shm, (hipStream_t)stream, x, y2, g1, b1, eps1,
                     reinterpret_cast<const unsigned char*>(img), qkv_bias, bias_frag, proj_bias, g2, b2, eps2, H, W, Hp, Wp, shift, scale, nullptr, dbg);
  return (int)hipGetLastError();
}
