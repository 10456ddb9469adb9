// K6 -- the mask-feature projection with the last GroupNorm (+ ReLU) of the pixel decoder folded into its A path (split_linear_h3l_kernel<NCHW, GNF>,
// split_linear_h3.h; reference pixel_decoder/msdeformattn.py:357-362).  Its own translation unit because it is compiled WITHOUT packed fp32 instructions
// (build.py: UNPACKED_ALWAYS).  Round 5 (profiles/r05_gnfold_select.txt, tools/gnf_asm_probe.py): in builds where the compiler wrote the coefficient
// a = gamma * rstd of the fold as `v_pk_mul_f32 vD[0:1], v_gamma[0:1], v_(mean, rstd)[0:1] op_sel:[0,1]` (the LOW product takes the HIGH register of
// source 1), that LOW product came out as exactly 0 in lanes 48-63 of a wave a few hundred times per launch -- whole rows of the staged tile normalised with
// a = 0, b = beta.  Hand-edited assembly of such a build pins it on that one instruction form: the same products written as two v_mul_f32, as a packed
// multiply on a broadcast pair, or with the operands swapped (`op_sel:[1,0]`: the cross select on source 0) are exact in every run; wait states, s_waitcnt
// vmcnt(0) / lgkmcnt(0), other destination or source registers and a drained matrix pipe change nothing.  Which form the compiler picks depends on register
// allocation (the ReLU variants of the round-5 search only moved it).  Without packed fp32 there is no such instruction in this kernel.
// tools/micro/pk_opsel_after_load.hip reproduces it standalone (profiles/r05_pk_opsel_erratum.txt): beside a wave that issues MFMAs with plain VALU between them, the LOW result
// of v_pk_mul_f32 / v_pk_add_f32 with the LOW select on source 1 is 0 in lanes 48-63 for up to 10 % of those lanes' results; every other select form is exact.
#include "split_linear_h3.h"

// The same projection reading a RAW convolution output: GroupNorm(G groups, statistics mr [B][G][2] from rba_group_norm_nhwc_stats_f32) (+ ReLU) is applied
// while the rows are staged -- `mask_features(output_conv(y))` (pixel_decoder/msdeformattn.py:357-362) without ever writing the normalised 1/4-resolution map
// (268 MB of traffic at 1024 x 2048).  Same arithmetic as rba_group_norm_nhwc_f32 followed by the entry above: bit-identical.  rows_per_image % 128 == 0.
extern "C" int rba_split_linear_nchw_out_gn_f16x3_f32(const float* x, const float* mr, const float* gamma, const float* beta, int G, int relu,
                                                      const void* weight_packed, const float* bias, float* out, int64_t M, int N, int K,
                                                      int rows_per_image, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 1 && K >= 32 && (K % 32) == 0 && rows_per_image >= 1 && G >= 1 && (K % G) == 0 && ((K / G) % 4) == 0);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && mr && gamma && beta && weight_packed && out && (M % rows_per_image) == 0 && (rows_per_image % 128) == 0 && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_packed | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0);
  rba_begin();
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_packed);
  const GnFold gn{mr, gamma, beta, G, K / G, relu ? 1 : 0};
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  const int rc = (tiles128 >= 160 || N <= 64) ? launch_h3l_nchw_gn<4>(x, gn, wp, bias, out, M, N, K, rows_per_image, (hipStream_t)stream)
                                              : launch_h3l_nchw_gn<2>(x, gn, wp, bias, out, M, N, K, rows_per_image, (hipStream_t)stream);
  if (rc) return rc;
  return rba_launch_status();
}
