// K6 -- fp32-accurate Linear on the bf16 matrix pipe ("bf16x6"): product dispatch of rba_split_linear_f32 onto the all-LDS-DMA
// kernels of split_linear_dma.h.  (reference: the nn.Linear calls of backbone/swin.py:44-71, :131-171, :319-343.)
// Configurations (tools/gemm_v4_sweep.py on every Swin-B / Swin-L / C5 token shape, profiles/r02_split_linear.txt):
//   * default: 128 x 128 tile, 4 MFMA waves (32 rows x 128 columns each) + 4 loader waves, one 16-wide k sub-stage per barrier, DMA two
//     stages ahead into a ring of three 20 KB stage buffers (60 KB LDS: two workgroups per CU and room for another stream's kernels --
//     with the equally fast 80 KB ring the two-stream bench lost 2 %);
//   * fewer than 256 such tiles (Swin stage 4 at one image): 128 x 64 tiles, persistent workgroups -- twice the workgroups, so every
//     CU still gets two.
#include "split_linear_h3.h"
#include "split_linear_h3q.h"
#include "mlp_fused_h3.h"

// A/B switch for tools and tests (not part of the ABI contract): 0 = product dispatch, 1 = always the round-2 pipelined 128 x 128 kernel
// (split_linear_h3p_kernel), 2 = the sub-tile kernel with the deferred epilogue (split_linear_h3q.h) wherever it applies, 100 + p = its ablation builds
RBA_KNOB(rba_k6_variant, 0);
RBA_KNOB_DEFINE(rba_k6_stagger, 0);
RBA_KNOB_DEFINE(rba_k6_occ, 2);
RBA_KNOB_DEFINE(rba_k6_rs_min_k, 512);  // the 256 x 128 form only from this K on (0: any K; see h3p_use_rs2)
RBA_KNOB_DEFINE(rba_k6_ks, 0);          // K-split 8-wave form of the single-resident launches: 0 = by rule (h3p_use_ks2), 1 = never, 2 = wherever legal
RBA_KNOB_DEFINE(rba_k6_rs, 0);          // 256 x 128 / 8-wave form: 0 = by tile count and stream hint (split_linear_h3.h), 1 = never, 2 = always, 3 = from 64 tiles
extern "C" __attribute__((visibility("hidden"))) int rba_concurrent_streams_hint = 1;

// The one piece of caller-set state of the library (include/rba_hip.h): how many streams of this process launch forwards CONCURRENTLY.  With two
// or more, the half-chip K6 launches (128 tiles of 256 x 128: Swin-B stage-3 proj / fc2, and the 1.5-round qkv) run the 8-wave form too: its
// workgroups own whole CUs, so such a launch takes 128 CUs and leaves the other 128 to the other streams' kernels instead of half of every CU
// (3 streams: 136.3 -> 139.7 images/s; alone it is slower, 116.3 -> 112.0: profiles/r04_bench_*.json).  Results are bit-identical either way.
extern "C" int rba_set_concurrent_streams(int n) {
  const int prev = rba_concurrent_streams_hint;
  rba_concurrent_streams_hint = n >= 1 ? n : 1;
  return prev;
}

extern "C" int rba_split_linear_f32(const float* x, const void* weight_planes, const float* bias, float* out, int64_t M, int N,
                                    int K, int act, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 1 && K >= 32 && (K % 32) == 0 && act >= 0 && act <= 2);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_planes && out && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_planes | (uintptr_t)out) & 15) == 0);
  rba_begin();
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_planes);
  hipStream_t st = (hipStream_t)stream;
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  int rc;
  if (tiles128 < 256 && N > 64 && (K >> 5) >= 2)
    rc = launch_v5_act<1, 2, 2, 2, 4>(act, x, wp, bias, out, M, N, K, 2, st);
  else
    rc = launch_v4_act<1, 4, 1, 2, 4>(act, x, wp, bias, out, M, N, K, st);
  if (rc) return rc;
  return rba_launch_status();
}

// ---- the f16x3 form (split_linear_h3.h): three f16 MFMAs per fp32 product, |x|, |w| < 65504
extern "C" int rba_split_weight_f16x2(const float* weight, void* packed, int N, int K, void* stream) {
  RBA_CHECK_ARG(N >= 0 && K >= 0 && (K % 32) == 0);
  if (N == 0 || K == 0) return 0;
  RBA_CHECK_ARG(weight && packed && (((uintptr_t)weight | (uintptr_t)packed) & 15) == 0);
  rba_begin();
  const int64_t total = (int64_t)((N + 127) >> 7) * (K >> 4) * 256;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(split_weight_f16x2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, weight, reinterpret_cast<u32x4_t*>(packed),
                     N, K);
  return rba_launch_status();
}

extern "C" int rba_split_linear_f16x3_f32(const float* x, const void* weight_packed, const float* bias, float* out, int64_t M, int N,
                                          int K, int act, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 1 && K >= 32 && (K % 32) == 0 && act >= 0 && act <= 2);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_packed && out && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_packed | (uintptr_t)out) & 15) == 0);
  rba_begin();
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_packed);
  hipStream_t st = (hipStream_t)stream;
  // K <= 256 (Swin stages 1-2: many short tiles) runs the LDS-staged form, longer K the straight-to-register forms (software-
  // pipelined with 128-column tiles, plain with 64-column tiles); 128 x 64 tiles
  // when there are fewer than 160 tiles of 128 x 128.  Chosen in the whole model (bench.py, one stream: 89.1 images/s; with the
  // LDS-staged form up to K = 1024, which the isolated sweep of profiles/r02_k6_f16x3.txt prefers by 2-4 %, 85.8).
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  const bool wide = tiles128 >= 160 || N <= 64;      // 128-column tiles from 160 tiles up (sweeps: 116-128 tiles prefer 64, 168-192 prefer 128)
  int rc;
  if (K <= 256) rc = wide ? launch_h3l_act<4>(act, x, wp, bias, out, M, N, K, st) : launch_h3l_act<2>(act, x, wp, bias, out, M, N, K, st);
  else rc = wide ? launch_h3p_act(act, x, wp, bias, out, M, N, K, st) : launch_h3_act<2>(act, x, wp, bias, out, M, N, K, st);
  if (rc) return rc;
  return rba_launch_status();
}

// 3 x 3 / stride 1 / pad 1 convolution over NHWC activations on the f16x3 kernel: weight_packed = rba_split_weight_f16x2 of the
// [N, 9 C] matrix w[n][(3 ky + kx) C + c].  (msdeformattn.py:278-297 `layer_{j}` output convolutions.)
extern "C" int rba_conv3x3_nhwc_f16x3_f32(const float* x, const void* weight_packed, const float* bias, float* out, int B, int H, int W,
                                          int C, int N, void* stream) {
  RBA_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && C >= 32 && (C % 32) == 0 && N >= 1);
  const int64_t M = (int64_t)B * H * W;
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_packed && out && M * C < (int64_t)1 << 30 && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_packed | (uintptr_t)out) & 15) == 0);
  rba_begin();
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_packed);
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  const int rc = (tiles128 >= 256 || N <= 64) ? launch_h3l_conv<4>(x, wp, bias, out, M, N, H, W, C, (hipStream_t)stream)
                                              : launch_h3l_conv<2>(x, wp, bias, out, M, N, H, W, C, (hipStream_t)stream);
  if (rc) return rc;
  return rba_launch_status();
}

// The FPN's lateral 1 x 1 convolution / output 3 x 3 convolution leaving the GroupNorm moments of their own output (GNM, split_linear_h3.h): `moments`
// [B][G][rows_per_image / 128][3] = (n, mean, M2) per 128-row tile and group of N / G consecutive output channels -- rba_group_norm_nhwc_merge_f32 turns them
// into the (mean, rstd) that rba_group_norm_nhwc_stats_f32 would compute with a pass over the output (pixel_decoder/msdeformattn.py:222-235, 278-297: Conv2d(norm=GN)).
// fp32 rows in; K <= 256; N % 128 == 0; N / G in {4, 8, 16, 32}; rows_per_image % 128 == 0.
extern "C" int rba_split_linear_f16x3_gn_moments_f32(const float* x, const void* weight_packed, const float* bias, float* out, int64_t M, int N, int K,
                                                     int rows_per_image, int G, float* moments, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 128 && (N % 128) == 0 && K >= 32 && (K % 32) == 0 && K <= 256 && G >= 1 && (N % G) == 0 && rows_per_image >= 128);
  const int cpg = N / G;
  RBA_CHECK_ARG((cpg == 4 || cpg == 8 || cpg == 16 || cpg == 32) && (rows_per_image % 128) == 0);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_packed && out && moments && (M % rows_per_image) == 0 && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_packed | (uintptr_t)out) & 15) == 0);
  rba_begin();
  const int rc = launch_h3l_gnm(x, reinterpret_cast<const u32x4_t*>(weight_packed), bias, out, M, N, K, GnMoments{moments, G, cpg, rows_per_image}, (hipStream_t)stream);
  if (rc) return rc;
  return rba_launch_status();
}

// split image in (rba_conv3x3_nhwc_f16x3_split_in_f32's operand), (H W) % 128 == 0, N % 128 == 0, N / G in {4, 8, 16, 32}
extern "C" int rba_conv3x3_nhwc_f16x3_split_in_gn_moments_f32(const void* x_frag, const void* weight_packed, const float* bias, float* out, int B, int H,
                                                              int W, int C, int N, int G, float* moments, void* stream) {
  RBA_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && C >= 32 && (C % 32) == 0 && C <= 2048 && N >= 128 && (N % 128) == 0 && G >= 1 && (N % G) == 0);
  const int cpg = N / G;
  const int64_t P = (int64_t)H * W, M = (int64_t)B * P;
  RBA_CHECK_ARG((cpg == 4 || cpg == 8 || cpg == 16 || cpg == 32) && (P % 128) == 0 && P < (int64_t)1 << 31);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x_frag && weight_packed && out && moments && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x_frag | (uintptr_t)weight_packed | (uintptr_t)out | (uintptr_t)bias) & 15) == 0);
  rba_begin();
  const int rc = launch_h3p_conv_pre_gnm(x_frag, reinterpret_cast<const u32x4_t*>(weight_packed), bias, out, M, N, H, W, C, GnMoments{moments, G, cpg, (int)P},
                                         (hipStream_t)stream);
  if (rc) return rc;
  return rba_launch_status();
}

// out = (residual + x W^T) + bias: the residual add of a transformer block (`x = x + proj(attn)`, `x = x + fc2(h)`: backbone/swin.py:284-293)
// folded into the GEMM epilogue; `out` may alias `residual`.
extern "C" int rba_split_linear_f16x3_res_f32(const float* x, const void* weight_packed, const float* bias, const float* residual, float* out,
                                              int64_t M, int N, int K, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 1 && K >= 32 && (K % 32) == 0);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_packed && residual && out && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_packed | (uintptr_t)out | (uintptr_t)residual) & 15) == 0);
  rba_begin();
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_packed);
  hipStream_t st = (hipStream_t)stream;
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  const bool wide = tiles128 >= 160 || N <= 64;
  int rc;
  if (K <= 256) rc = wide ? launch_h3l_res<4>(x, wp, bias, residual, out, M, N, K, st) : launch_h3l_res<2>(x, wp, bias, residual, out, M, N, K, st);
  else rc = wide ? launch_h3p_res(x, wp, bias, residual, out, M, N, K, st) : launch_h3_res<2>(x, wp, bias, residual, out, M, N, K, st);
  if (rc) return rc;
  return rba_launch_status();
}

// The same Linear with the A operand supplied as the producer's split fragment image (rba_add_layer_norm_frag_f32, ...): no
// activation arithmetic and only contiguous 1 KiB wave loads in the GEMM.  x_frag: ceil(M / 32) * 32 * K * 4 bytes, layout in
// split_linear_h3.h ("PRE").  act 0 / 1 (GELU) / 2 (ReLU); residual (nullable, act must be 0): out = residual + x W^T + bias.
extern "C" int rba_split_linear_f16x3_frag_f32(const void* x_frag, const void* weight_packed, const float* bias, const float* residual,
                                               float* out, int64_t M, int N, int K, int act, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 1 && K >= 32 && (K % 32) == 0 && act >= 0 && act <= 2 && !(residual && act));
  if (M == 0) return 0;
  RBA_CHECK_ARG(x_frag && weight_packed && out && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x_frag | (uintptr_t)weight_packed | (uintptr_t)out | (uintptr_t)residual) & 15) == 0);
  rba_begin();
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_packed);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  // Where the sub-tile kernel wins (tools/k6_h3q_ab.py on the Swin-B / Swin-L / C5 shapes, profiles/r03_k6_h3q.txt): launches of fewer than 256
  // tiles of 128 x 128, where the 128 x 128 kernel leaves every SIMD a single wave (Swin stage 4: proj 30 -> 22 us, fc2 91 -> 66 us; C5 stage 3-4
  // fc2 1.5x); with 256 tiles or more its doubled A-operand traffic (a 64-column sub-tile re-reads the row panel twice as often) costs more
  // than the second wave and the deferred epilogue bring (stage-3 qkv 45 -> 61 us), so those stay on the 128 x 128 kernel.
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  const bool sub_tiles = rba_k6_variant == 2 || (rba_k6_variant == 0 && tiles128 < (residual ? 256 : 200));
  if (sub_tiles && h3q_supported(M, N, K)) {
    if (residual) rc = launch_h3q<H3Q_RES, 0>(x_frag, wp, bias, residual, out, M, N, K, st);
    else if (act == 1) rc = launch_h3q<H3Q_F32, 1>(x_frag, wp, bias, nullptr, out, M, N, K, st);
    else if (act == 2) rc = launch_h3q<H3Q_F32, 2>(x_frag, wp, bias, nullptr, out, M, N, K, st);
    else rc = launch_h3q<H3Q_F32, 0>(x_frag, wp, bias, nullptr, out, M, N, K, st);
  } else {
    rc = h3p_single_resident(M, N) ? launch_h3p_pre_act<1>(act, x_frag, wp, bias, residual, out, M, N, K, st)
                                   : launch_h3p_pre_act<2>(act, x_frag, wp, bias, residual, out, M, N, K, st);
  }
  if (rc) return rc;
  return rba_launch_status();
}

// Linear + GELU whose OUTPUT is the next Linear's split fragment image (Mlp.fc1 -> fc2 of backbone/swin.py:35-41): the operand-swapped
// pipelined kernel.  x: fp32 rows (x_is_split 0) or a split image (1).  out_frag: ceil(M / 32) * 32 * N * 4 bytes.  N % 32 == 0.
extern "C" int rba_split_linear_f16x3_gelu_split_out(const void* x, int x_is_split, const void* weight_packed, const float* bias,
                                                     void* out_frag, int64_t M, int N, int K, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 32 && (N % 32) == 0 && K >= 32 && (K % 32) == 0);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_packed && out_frag && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_packed | (uintptr_t)out_frag | (uintptr_t)bias) & 15) == 0);
  rba_begin();
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_packed);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (x_is_split && rba_k6_variant >= 100 && h3q_supported(M, N, K)) {            // tools: ablation builds of the sub-tile kernel
    switch (rba_k6_variant - 100) {
#define H3Q_PROBE(P) case P: rc = launch_h3q<H3Q_SPLIT, 1, P>(x, wp, bias, nullptr, out_frag, M, N, K, st); break;
      H3Q_PROBE(1) H3Q_PROBE(2) H3Q_PROBE(4) H3Q_PROBE(8) H3Q_PROBE(16) H3Q_PROBE(3) H3Q_PROBE(7) H3Q_PROBE(15) H3Q_PROBE(31) H3Q_PROBE(5)
#undef H3Q_PROBE
      default: rc = launch_h3q<H3Q_SPLIT, 1>(x, wp, bias, nullptr, out_frag, M, N, K, st);
    }
  } else if (x_is_split && h3q_supported(M, N, K) &&
             (rba_k6_variant == 2 || (rba_k6_variant == 0 && K >= 768 && ((M + 127) / 128) * ((N + 127) / 128) > 512 && !h3p_use_rs2(M, N, K))))
    // (round 4: where the 256 x 128 form of the pipelined kernel applies it beats both -- Swin-L stage 3: 130 (128 x 128) / 136-146 (sub-tiles) / 117 us)
    // fc1 + GELU with the epilogue deferred into the next sub-tile's k loop: pays where the 128 x 128 kernel needs more than one round of
    // workgroups AND the k loop is long (Swin-L: stage 3 165 -> 143 us, stage 4 147 -> 142 us); a one-round launch (Swin-B stage 4: 512
    // tiles, 61 vs 72 us) or a 16-block loop (Swin-B stage 3: 65 vs 71 us) is better off on the 128 x 128 kernel
    rc = launch_h3q<H3Q_SPLIT, 1>(x, wp, bias, nullptr, out_frag, M, N, K, st);
  else if (h3p_single_resident(M, N))
    rc = x_is_split ? launch_h3p_fout<1, true, 1>(x, wp, bias, out_frag, M, N, K, st) : launch_h3p_fout<1, false, 1>(x, wp, bias, out_frag, M, N, K, st);
  else
    rc = x_is_split ? launch_h3p_fout<1, true, 2>(x, wp, bias, out_frag, M, N, K, st) : launch_h3p_fout<1, false, 2>(x, wp, bias, out_frag, M, N, K, st);
  if (rc) return rc;
  return rba_launch_status();
}

// The same convolution reading the split image of x (rba_resample_bilinear_nhwc_split_out_f32 writes it: the FPN's `lateral + upsample`
// sum feeds only this convolution, pixel_decoder/msdeformattn.py:352-361): the pipelined kernel, no activation arithmetic.  x_frag =
// image of [B H W, C] rows.  For launches of at least 256 tiles of 128 x 128 (the caller keeps the fp32 entry point otherwise).
extern "C" int rba_conv3x3_nhwc_f16x3_split_in_f32(const void* x_frag, const void* weight_packed, const float* bias, float* out, int B, int H,
                                                   int W, int C, int N, void* stream) {
  RBA_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && C >= 32 && (C % 32) == 0 && C <= 2048 && N >= 1);
  const int64_t M = (int64_t)B * H * W;
  if (M == 0) return 0;
  RBA_CHECK_ARG(x_frag && weight_packed && out && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x_frag | (uintptr_t)weight_packed | (uintptr_t)out | (uintptr_t)bias) & 15) == 0);
  rba_begin();
  const int rc = launch_h3p_conv_pre(x_frag, reinterpret_cast<const u32x4_t*>(weight_packed), bias, out, M, N, H, W, C, (hipStream_t)stream);
  if (rc) return rc;
  return rba_launch_status();
}

// out = residual + fc2(GELU(fc1(x))) for C = 128 in one kernel (mlp_fused_h3.h): x [M, 128] fp32 rows, w1_packed = rba_split_weight_f16x2 of
// fc1.weight [HID, 128], w2_packed of fc2.weight [128, HID], HID % 32 == 0; `out` may be `residual`.  Bit-identical to
// rba_split_linear_f16x3_gelu_split_out + rba_split_linear_f16x3_frag_f32(residual).  (Mlp + residual of backbone/swin.py:35-41, 293.)
extern "C" int rba_swin_mlp_fused_f16x3_f32(const float* x, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                                            const float* residual, float* out, int64_t M, int C, int HID, void* stream) {
  RBA_CHECK_ARG(M >= 0 && C == 128 && HID >= 64 && (HID % 32) == 0);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && w1_packed && b1 && w2_packed && residual && out && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1_packed | (uintptr_t)w2_packed | (uintptr_t)residual | (uintptr_t)out | (uintptr_t)b1) & 15) == 0);
  rba_begin();
  const int rc = launch_mlp_fused(x, reinterpret_cast<const u32x4_t*>(w1_packed), b1, reinterpret_cast<const u32x4_t*>(w2_packed), b2, residual, out, M,
                                  HID, (hipStream_t)stream);
  if (rc) return rc;
  return rba_launch_status();
}

// The same kernel from the block's residual stream: x <- x + fc2(GELU(fc1(norm2(x)))) in place, norm2 computed by the kernel itself from the rows it loads
// anyway (backbone/swin.py:293 whole): neither the LayerNorm launch nor its output tensor exist.
extern "C" int rba_swin_mlp_fused_ln_f16x3_f32(float* x, const float* norm_weight, const float* norm_bias, float eps, const void* w1_packed, const float* b1,
                                               const void* w2_packed, const float* b2, int64_t M, int C, int HID, void* stream) {
  RBA_CHECK_ARG(M >= 0 && C == 128 && HID >= 64 && (HID % 32) == 0);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && norm_weight && norm_bias && w1_packed && b1 && w2_packed && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1_packed | (uintptr_t)w2_packed | (uintptr_t)b1 | (uintptr_t)norm_weight | (uintptr_t)norm_bias) & 15) == 0);
  rba_begin();
  const int rc = launch_mlp_fused(x, reinterpret_cast<const u32x4_t*>(w1_packed), b1, reinterpret_cast<const u32x4_t*>(w2_packed), b2, x, x, M, HID,
                                  (hipStream_t)stream, norm_weight, norm_bias, eps);
  if (rc) return rc;
  return rba_launch_status();
}

// x [B * P, K] (NHWC rows) -> out [B, N, P] (NCHW) on the f16x3 kernel: the mask-feature projection (pixel_decoder/msdeformattn.py:362,
// `self.mask_features(y)`), whose consumer K4 reads [C][pixels].  weight_packed = rba_split_weight_f16x2.  (The bf16x6 form of the same
// operator is rba_split_linear_nchw_out_f32.)
extern "C" int rba_split_linear_nchw_out_f16x3_f32(const float* x, const void* weight_packed, const float* bias, float* out, int64_t M, int N,
                                                   int K, int rows_per_image, void* stream) {
  RBA_CHECK_ARG(M >= 0 && N >= 1 && K >= 32 && (K % 32) == 0 && rows_per_image >= 1);
  if (M == 0) return 0;
  RBA_CHECK_ARG(x && weight_packed && out && (M % rows_per_image) == 0 && M < (int64_t)1 << 31);
  RBA_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight_packed | (uintptr_t)out) & 15) == 0);
  rba_begin();
  const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(weight_packed);
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  const int rc = (tiles128 >= 160 || N <= 64) ? launch_h3l_nchw<4>(x, wp, bias, out, M, N, K, rows_per_image, (hipStream_t)stream)
                                              : launch_h3l_nchw<2>(x, wp, bias, out, M, N, K, rows_per_image, (hipStream_t)stream);
  if (rc) return rc;
  return rba_launch_status();
}

// (rba_split_linear_nchw_out_gn_f16x3_f32, the GroupNorm-folded form, lives in split_linear_gnf.hip: a translation unit compiled WITHOUT packed fp32 instructions)
