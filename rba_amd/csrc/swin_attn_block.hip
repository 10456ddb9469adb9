// K7 -- entry points of the fused Swin attention half (kernel and design notes: swin_attn_block.h)
#include "swin_attn_block.h"
#include "../../include/rba_hip.h"

namespace {

template <int C>
int k7_launch(float* x, float* y2, const float* g1, const float* b1, float eps1, const void* img, const float* qkv_bias, const float* bias_frag,
              const float* proj_bias, const float* g2, const float* b2, float eps2, int B, int H, int W, int shift, hipStream_t st) {
  const int Hp = (H + K7_WS - 1) / K7_WS * K7_WS, Wp = (W + K7_WS - 1) / K7_WS * K7_WS;
  if (Hp / K7_WS > 65535 || B > 65535) return (int)hipErrorInvalidValue;
  const float scale = (float)(1.0 / sqrt(32.0));                                 // head_dim ** -0.5 (swin.py:103, 145)
  const dim3 grid(Wp / K7_WS, Hp / K7_WS, B), block(64 * K7_WAVES);
  constexpr size_t shm = k7_lds_bytes(C);
  static unsigned char done[2][64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
  const unsigned char* im = reinterpret_cast<const unsigned char*>(img);
  if (y2) {
    if (!done[1][dev]) {
      const hipError_t e = hipFuncSetAttribute((const void*)swin_attn_block_kernel<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (e != hipSuccess) return (int)e;
      done[1][dev] = 1;
    }
    hipLaunchKernelGGL((swin_attn_block_kernel<C, true>), grid, block, shm, st, x, y2, g1, b1, eps1, im, qkv_bias, bias_frag, proj_bias, g2, b2, eps2, H, W,
                       Hp, Wp, shift, scale, nullptr);
  } else {
    if (!done[0][dev]) {
      const hipError_t e = hipFuncSetAttribute((const void*)swin_attn_block_kernel<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      if (e != hipSuccess) return (int)e;
      done[0][dev] = 1;
    }
    hipLaunchKernelGGL((swin_attn_block_kernel<C, false>), grid, block, shm, st, x, nullptr, g1, b1, eps1, im, qkv_bias, bias_frag, proj_bias, nullptr,
                       nullptr, 0.f, H, W, Hp, Wp, shift, scale, nullptr);
  }
  return rba_launch_status();
}

// attention only (PROJ = false): norm1 -> qkv -> window attention, output = the proj Linear's split image
template <int C>
int k7_launch_qkv(const float* x, void* out_frag, const float* g1, const float* b1, float eps1, const void* img, const float* qkv_bias, const float* bias_frag,
                  int B, int H, int W, int shift, hipStream_t st) {
  const int Hp = (H + K7_WS - 1) / K7_WS * K7_WS, Wp = (W + K7_WS - 1) / K7_WS * K7_WS;
  if (Hp / K7_WS > 65535 || B > 65535) return (int)hipErrorInvalidValue;
  const float scale = (float)(1.0 / sqrt(32.0));
  const dim3 grid(Wp / K7_WS, Hp / K7_WS, B), block(64 * K7_WAVES);
  constexpr size_t shm = k7_lds_bytes(C, false);
  static unsigned char done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
  if (!done[dev]) {
    const hipError_t e = hipFuncSetAttribute((const void*)swin_attn_block_kernel<C, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return (int)e;
    done[dev] = 1;
  }
  hipLaunchKernelGGL((swin_attn_block_kernel<C, false, false>), grid, block, shm, st, const_cast<float*>(x), nullptr, g1, b1, eps1,
                     reinterpret_cast<const unsigned char*>(img), qkv_bias, bias_frag, nullptr, nullptr, nullptr, 0.f, H, W, Hp, Wp, shift, scale, out_frag);
  return rba_launch_status();
}

}  // namespace

extern "C" int rba_swin_attn_block_supported(int C, int ws) { return (C == 128 && ws == K7_WS) ? 1 : 0; }
extern "C" int rba_swin_attn_qkv_supported(int C, int ws) { return ((C == 128 || C == 192 || C == 256) && ws == K7_WS) ? 1 : 0; }

extern "C" int rba_swin_attn_qkv_split_out_f32(const float* x, void* out_frag, const float* norm1_weight, const float* norm1_bias, float eps1,
                                               const void* weight_image, const float* qkv_bias, const float* bias_frag, int B, int H, int W, int C, int ws,
                                               int shift, void* stream) {
  RBA_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && rba_swin_attn_qkv_supported(C, ws) && shift >= 0 && shift < ws);
  if (B == 0) return 0;
  RBA_CHECK_ARG(x && out_frag && norm1_weight && norm1_bias && weight_image && qkv_bias && bias_frag);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)out_frag | (uintptr_t)norm1_weight | (uintptr_t)norm1_bias | (uintptr_t)weight_image | (uintptr_t)qkv_bias |
                  (uintptr_t)bias_frag) & 15) == 0);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  if (C == 192) return k7_launch_qkv<192>(x, out_frag, norm1_weight, norm1_bias, eps1, weight_image, qkv_bias, bias_frag, B, H, W, shift, st);   // Swin-L stage 1
  if (C == 256) return k7_launch_qkv<256>(x, out_frag, norm1_weight, norm1_bias, eps1, weight_image, qkv_bias, bias_frag, B, H, W, shift, st);
  return k7_launch_qkv<128>(x, out_frag, norm1_weight, norm1_bias, eps1, weight_image, qkv_bias, bias_frag, B, H, W, shift, st);
}

extern "C" int64_t rba_swin_attn_block_weight_bytes(int C) {
  if (C <= 0 || C % 32) return 0;
  return (int64_t)(C / 32) * k7_head_bytes(C);
}

extern "C" int rba_swin_attn_block_pack_f32(const float* qkv_weight, const float* proj_weight, void* image, int C, void* stream) {
  RBA_CHECK_ARG(qkv_weight && proj_weight && image && C >= 32 && C % 32 == 0 && C <= 1024);
  RBA_CHECK_ARG((((uintptr_t)qkv_weight | (uintptr_t)proj_weight | (uintptr_t)image) & 15) == 0);
  rba_begin();
  const int64_t total = (int64_t)(C / 32) * (3 * (C / 32) * 2 + C / 16) * 64;
  hipLaunchKernelGGL(swin_attn_block_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qkv_weight, proj_weight,
                     reinterpret_cast<unsigned char*>(image), C);
  return rba_launch_status();
}

extern "C" int rba_swin_attn_block_f32(float* x, float* y2, const float* norm1_weight, const float* norm1_bias, float eps1, const void* weight_image,
                                       const float* qkv_bias, const float* bias_frag, const float* proj_bias, const float* norm2_weight,
                                       const float* norm2_bias, float eps2, int B, int H, int W, int C, int ws, int shift, void* stream) {
  RBA_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && rba_swin_attn_block_supported(C, ws) && shift >= 0 && shift < ws);
  if (B == 0) return 0;
  RBA_CHECK_ARG(x && norm1_weight && norm1_bias && weight_image && qkv_bias && bias_frag && proj_bias);
  RBA_CHECK_ARG(!y2 || (norm2_weight && norm2_bias));
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)y2 | (uintptr_t)norm1_weight | (uintptr_t)norm1_bias | (uintptr_t)weight_image | (uintptr_t)qkv_bias |
                  (uintptr_t)bias_frag | (uintptr_t)proj_bias | (uintptr_t)norm2_weight | (uintptr_t)norm2_bias) & 15) == 0);
  rba_begin();
  return k7_launch<128>(x, y2, norm1_weight, norm1_bias, eps1, weight_image, qkv_bias, bias_frag, proj_bias, norm2_weight, norm2_bias, eps2, B, H, W, shift,
                        (hipStream_t)stream);
}
