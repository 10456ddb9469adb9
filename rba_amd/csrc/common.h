// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of librba_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "knobs.h"

#define RBA_WAVE 64

// native clang vectors (the nontemporal builtins reject HIP's float4 class)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t rba_u32x4 __attribute__((ext_vector_type(4)));

#define RBA_CHECK_ARG(cond)                      \
  do {                                           \
    if (!(cond)) return (int)hipErrorInvalidValue; \
  } while (0)

// hipGetLastError() is per-thread and sticky: an unrelated earlier failure of the host program (e.g. a device
// probe) would otherwise be reported as ours.  Every entry point calls rba_begin() before launching.
static inline void rba_begin() { (void)hipGetLastError(); }
static inline int rba_launch_status() { return (int)hipGetLastError(); }

// 1 / (1 + e^-x): v_exp_f32 + v_rcp_f32, abs error <~ 1e-7 (ample for the 1e-4 score tolerance).
__device__ __forceinline__ float rba_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// The same sigmoid on two values with the multiply and the add packed (v_pk_mul_f32 / v_pk_add_f32): identical operations per
// element (x * -log2(e), v_exp_f32, + 1, v_rcp_f32), two VALU issues fewer per pair.
__device__ __forceinline__ f32x2 rba_sigmoid2(f32x2 x) {
  const f32x2 t = x * -1.44269504088896340736f;
  const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
  return (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

// ReLU that PRESERVES NaN (fmaxf(NaN, 0) is 0): the f16x3 kernels answer an out-of-range operand with NaN, and that NaN must reach the score map, where
// the evaluator's finiteness check sends the image to the full-range bf16x6 kernels -- a ReLU that swallowed it would turn a loud failure into a finite,
// wrong score.  Round 6: compare + select, `x != x ? x : max(x, floor)` -- exactly torch.relu for every input: relu(+inf) = +inf, relu(-inf) = 0,
// and with the activation off (floor = -inf) the value passes untouched, -inf and -0.0 included.  Rounds 4-5 used the arithmetic form max(x, 0) + (x - x), which
// also turned +-inf into NaN and -0.0 into +0.0 on the no-activation path.  Three VALU instructions either way.  (Round 5 once blamed compare + select forms for
// wrong rows in the GroupNorm fold; the cause was one packed multiply, `v_pk_mul_f32 ... op_sel:[0,1]` -- tools/gnf_asm_probe.py, csrc/split_linear_gnf.hip --
// and the build now refuses a library that contains that form: isa_hazards.gate.)
__device__ __forceinline__ float rba_relu(float x) { return x != x ? x : fmaxf(x, 0.f); }
// The same under a run-time (workgroup-uniform) switch: floor = rba_relu_floor(on) once, then rba_clamp_below(x, floor) per value.
__device__ __forceinline__ float rba_relu_floor(bool on) { return on ? 0.f : -INFINITY; }
__device__ __forceinline__ float rba_clamp_below(float x, float floor) { return x != x ? x : fmaxf(x, floor); }

// tanh(x) = sign(x) * (1 - 2 / (e^{2|x|} + 1)); e^{2|x|} -> inf gives exactly 1.
__device__ __forceinline__ float rba_tanh(float x) {
  float ax = fabsf(x);
  float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * ax) + 1.0f);
  return copysignf(t, x);
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, RBA_WAVE));
  return v;
}
__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RBA_WAVE);
  return v;
}

// Source coordinate of ATen's upsample_bilinear2d (align_corners=False): scale*(dst+0.5)-0.5 clamped at 0.
struct BilinearTap {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ BilinearTap bilinear_tap(int dst, float scale, int in_size) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  int i0 = (int)src;  // src >= 0: trunc == floor
  i0 = i0 < in_size - 1 ? i0 : in_size - 1;
  BilinearTap t;
  t.i0 = i0;
  t.i1 = i0 < in_size - 1 ? i0 + 1 : i0;
  t.l1 = src - (float)i0;
  t.l0 = 1.0f - t.l1;
  return t;
}

// The f16x3 split of two fp32 values (split_linear_h3.h): h = f16(x) (rne), l = f16((x - h) 2^11) = f16(fma(h, -2^11, 2^11 x)), the
// fma exact.  Packed pairs: h = {f16(a), f16(b)}, l likewise.  v_cvt_pk_f16_f32, v_pk_mul_f32, v_fma_mixlo/mixhi_f16 reading the f16
// halves of h in place: four VALU.  |x| >= 65504 gives h = inf and a NaN l (documented domain of the f16x3 mode).
__device__ __forceinline__ void rba_split_f16x2(float a, float b, uint32_t& h, uint32_t& l) {
  typedef _Float16 rba_f16x2 __attribute__((ext_vector_type(2)));
  const rba_f16x2 hh = {(_Float16)a, (_Float16)b};
  const uint32_t ap = __builtin_bit_cast(uint32_t, hh);
  const f32x2 t = (f32x2){a, b} * 2048.0f;
  uint32_t r;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(ap), "v"(-2048.0f), "v"(t.x));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(ap), "v"(-2048.0f), "v"(t.y));
  h = ap;
  l = r;
}
