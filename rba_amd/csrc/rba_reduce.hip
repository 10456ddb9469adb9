// K1 -- the RbA "rejected by all" reduction (reference: mask2former/maskformer_model.py:381-386 +
// evaluate_ood.py:143-150 + support.py:385-388), and its variant fused with the x4 mask upsample.
//
//   sem[k,p] = sum_q P[q,k] * sigmoid(mask[q,p]);  rba[p] = -sum_k tanh(sem[k,p]);  argmax[p] = argmax_k sem
//
// Scan/reduce priced against the HBM roofline: every mask plane is read exactly once with 16 B/lane coalesced
// loads, the Q x K class probabilities are wave-uniform (scalar loads -> SGPR operands of the FMAs), K accumulators
// per pixel live in VGPRs, nothing but rba (and optionally sem_seg / argmax) is written.  Algorithmic bytes per
// launch: 4*Q*HW + 4*Q*K + 4*HW (+ 4*K*HW with sem_seg, + 4*HW with argmax).  Measured: ~170 us = 62 % of 8 TB/s at
// Q=100, K=19, 1024x2048, HBM traffic 1.005x algorithmic; the remaining gap to the 128 us of the load pattern is VALU
// time (see DESIGN.md and profiles/r01_k1_bandwidth_probes.txt).  Kernels live in rba_reduce_kernels.h; probes and
// experimental variants in tune/rba_reduce_tune.hip (built into librba_tune.so for tools/, not into this library).
#include "rba_reduce_kernels.h"

using namespace rba_k1;

extern "C" int rba_hip_version(void) { return 190; }

// tools / tests only: 1 = rba_reduce_up4_f32 runs the generic (round 1-2) kernel for K = 19 / 20 too, 2 = always the packed VALU kernel (no MFMA form)
RBA_KNOB(rba_k1_up4_variant, 0);

static int reduce_impl(const float* mask, const float* cls_prob, float* rba, float* sem_seg, int32_t* argmax, int Q, int K, int64_t HW,
                       int score_mode, unsigned int* counters, void* stream) {
  RBA_CHECK_ARG(Q >= 1 && K >= 1 && K <= 160 && HW >= 0 && score_mode >= 0 && score_mode <= 2);
  const int mode = score_mode;
  if (HW == 0) return 0;
  RBA_CHECK_ARG(mask && cls_prob && rba);
  rba_begin();
  hipStream_t st = (hipStream_t)stream;
  const bool vec4 = (HW % 4 == 0) && ((((uintptr_t)mask | (uintptr_t)rba | (uintptr_t)sem_seg) & 15) == 0);
  // K = 19 / 20 (Cityscapes without / with void): compile-time K, packed FMAs, 2-deep prefetch ring, 4 workgroups per CU
  if (K == 19 && vec4) return launch_reduce_pk<19, 2, 4>(mask, cls_prob, rba, sem_seg, argmax, Q, HW, mode, counters, st);
  if (K == 20 && vec4) return launch_reduce_pk<20, 2, 4>(mask, cls_prob, rba, sem_seg, argmax, Q, HW, mode, counters, st);
  if (K <= 32 && vec4) return launch_reduce<32, 4>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st, mode);
  if (K <= 32) return launch_reduce<32, 1>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st, mode);
  if (K <= 80) return launch_reduce<80, 1>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st, mode);
  return launch_reduce<160, 1>(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, st, mode);
}

extern "C" int rba_reduce_f32(const float* mask, const float* cls_prob, float* rba, float* sem_seg, int32_t* argmax,
                              int Q, int K, int64_t HW, int score_mode, void* stream) {
  return reduce_impl(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, score_mode, nullptr, stream);
}

extern "C" int rba_reduce_ws_f32(const float* mask, const float* cls_prob, float* rba, float* sem_seg, int32_t* argmax,
                                 int Q, int K, int64_t HW, int score_mode, void* workspace, void* stream) {
  RBA_CHECK_ARG(workspace && (((uintptr_t)workspace) & 3) == 0);
  return reduce_impl(mask, cls_prob, rba, sem_seg, argmax, Q, K, HW, score_mode, reinterpret_cast<unsigned int*>(workspace), stream);
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
  // rows a multiple of four wide, at most 512 queries: the class contraction on the matrix pipe (since round 4 also with sem_seg / argmax outputs;
  // rba_k1_up4_variant = 3 (tools / tests): the matrix-pipe kernel for score-only calls only, the round-3 dispatch)
  const bool mx = (rba_k1_up4_variant == 0 || (rba_k1_up4_variant == 3 && !sem_seg && !argmax)) && (crop_w & 3) == 0 && Q <= 512 &&
                  (int64_t)(Q + 8) * h * w < (1LL << 29) && (((uintptr_t)argmax) & 15) == 0;
  if (K == 19 && mx) return launch_up4_mx<19>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, h, w, crop_h, crop_w, st, score_mode);
  if (K == 20 && mx) return launch_up4_mx<20>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, h, w, crop_h, crop_w, st, score_mode);
  if (K == 19 && rba_k1_up4_variant != 1) return launch_up4_pk<19>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, h, w, crop_h, crop_w, st, score_mode);
  if (K == 20 && rba_k1_up4_variant != 1) return launch_up4_pk<20>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, h, w, crop_h, crop_w, st, score_mode);
  if (K == 19) return launch_up4<19>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, K, h, w, crop_h, crop_w, st, score_mode);
  return launch_up4<32>(mask_lowres, cls_prob, rba, sem_seg, argmax, Q, K, h, w, crop_h, crop_w, st, score_mode);
}

