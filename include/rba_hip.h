/*
 * rba_hip.h -- C ABI of librba_hip.so, the MI355X (gfx950) kernels on RbA's inference hot path.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated; no environment variables are read and
 *     there is no mutable global state except the launch-geometry hint of rba_set_concurrent_streams -- `nm -D librba_hip.so` shows no writable rba_ symbol:
 *     the kernel-variant knobs of csrc/knobs.h are compile-time constants here and variables only in the tools' knobs build librba_hip_knobs.so (the
 *     only other process-wide side effects: the first launch of a kernel that needs more than 64 KiB of LDS sets that kernel's
 *     hipFuncAttributeMaxDynamicSharedMemorySize once, and the persistent kernels cache the device's CU count); re-entrant; tensors
 *     are dense row-major ("contiguous") in the index order written next to them;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); kernels are only enqueued;
 *   - return value is a hipError_t as int (0 = hipSuccess); argument errors return hipErrorInvalidValue (1);
 *   - fp32 everywhere (the reference op refuses half: pixel_decoder/msdeformattn.py:323,329).
 *
 * Reference interfaces replaced (paths relative to the reference repo NazirNayal8/RbA):
 *   rba_ms_deform_attn_fwd_f32  <- MultiScaleDeformableAttention.ms_deform_attn_forward
 *                                  (pixel_decoder/ops/src/vision.cpp:19, ops/src/ms_deform_attn.h:25-44,
 *                                   ops/src/cuda/ms_deform_attn_cuda.cu:25-85, ms_deform_im2col_cuda.cuh:242-304,928-959)
 *   rba_reduce_f32              <- MaskFormer.semantic_inference (mask2former/maskformer_model.py:381-386)
 *                                  + get_RbA (evaluate_ood.py:143-150) + argmax (support.py:385-388)
 *   rba_reduce_up4_f32          <- the same preceded by the x4 mask upsample (maskformer_model.py:294-299)
 *                                  and followed by the sem_seg_postprocess crop (maskformer_model.py:330-332)
 *   rba_resample_bilinear_f32   <- F.interpolate(mode="bilinear", align_corners=False) call sites
 *                                  (maskformer_model.py:294-299, msdeformattn.py:358, mask2former_transformer_decoder.py:483)
 *   rba_masked_xattn_f32        <- nn.MultiheadAttention core with bool attn_mask
 *                                  (mask2former_transformer_decoder.py:106-118, 433, 483-487)
 *   rba_mask_logits_f32         <- torch.einsum("bqc,bchw->bqhw") (mask2former_transformer_decoder.py:479)
 *   rba_mask_logits_f16x3_f32   <- the same call site, f16x3 arithmetic
 *   rba_swin_window_attn_f32    <- WindowAttention core + window_partition/reverse + roll + pad
 *                                  (backbone/swin.py:44-71, 131-171, 251-284)
 *   rba_skinny_linear_f32       <- nn.Linear / in_proj / MLP on the decoder's [100, B, 256] query tensors
 *                                  (mask2former_transformer_decoder.py:25-212)
 *   rba_bn_relu_conv1x1_f32     <- BNReluConv(hidden_dim, 2, k=1) `ood_pred` head (mask2former_transformer_decoder.py:216-230, 467-468)
 *   rba_split_linear_f32        <- nn.Linear on the backbone's token tensors: qkv / proj / Mlp.fc1(+GELU) / Mlp.fc2 /
 *                                  PatchMerging.reduction (backbone/swin.py:44-71, 131-171, 319-343)
 *   rba_token_linear_f32        <- nn.Linear on the encoder / decoder-memory token tensors (pixel_decoder/msdeformattn.py:101-140,
 *                                  ops/modules/ms_deform_attn.py:95-121, mask2former_transformer_decoder.py:83-143) [+ residual + LayerNorm]
 *   rba_gaussian_blur_f32       <- transforms.GaussianBlur(7, sigma=1) on the anomaly map (support.py:366-383)
 *   rba_threshold_u8 / rba_morph3x3_u8 / rba_ccl4_roots_i32 <- the open-set branch of MaskFormer.panoptic_inference
 *                                  (maskformer_model.py:454-481: threshold, cv2.morphologyEx open/close, cv2.connectedComponents)
 *   rba_add_layer_norm_f32      <- `x = x + proj(...)` followed by nn.LayerNorm (swin.py:284-293 and the post-norm layers
 *                                  of msdeformattn.py:134-138, mask2former_transformer_decoder.py:48-58,106-118,171-175)
 *   rba_group_norm_f32          <- GroupNorm(32) [+ ReLU] behind Detectron2's Conv2d(norm=get_norm("GN"), activation)
 *                                  (pixel_decoder/msdeformattn.py:222-235, 278-297)
 */
#ifndef RBA_HIP_H
#define RBA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Library/ABI version (major*100 + minor). */
int rba_hip_version(void);

/* Launch-geometry hint, the library's only caller-set state: the number of streams of this process that launch forwards concurrently
 * (default 1).  With n >= 2 the K6 launches that fill only half the chip take whole CUs (8-wave 256 x 128 workgroups) and leave the rest to the
 * other streams' kernels.  Affects speed only: every kernel form gives bit-identical results.  Returns the previous setting (n < 1 counts as 1).
 * Not thread-safe against concurrent launches; set it before the streams start (bench.py, evaluate_ood.run_evaluations do). */
int rba_set_concurrent_streams(int n);

/* K1.  mask [Q,HW] full-resolution mask logits; cls_prob [Q,K] = softmax(class logits)[:, :-1].
 *   sem[k,p]  = sum_q cls_prob[q,k] * sigmoid(mask[q,p])      (ascending-q fp32 FMA order)
 *   rba[p]    = - sum_k tanh(sem[k,p])                     (score_mode 0, RbA: evaluate_ood.py:143-150)
 *             = - logsumexp_k sem[k,p]                     (score_mode 1, energy: evaluate_ood.py:152-159)
 *             = - sum_k sem[k,p]                           (score_mode 2, negative logit sum: support.py:115-132)
 *   argmax[p] = first k maximising sem[k,p]
 * rba [HW] required; sem_seg [K,HW] and argmax [HW] (int32) optional (NULL = not written).  1 <= K <= 160. */
int rba_reduce_f32(const float* mask, const float* cls_prob, float* rba, float* sem_seg, int32_t* argmax,
                   int Q, int K, int64_t HW, int score_mode, void* stream);

/* The same with dynamic tile assignment: `workspace` = 8 bytes of device memory owned by the caller, zero before the first use
 * (the kernel leaves it zero); one workspace per stream that may run K1 concurrently.  Workgroups fetch their tiles from an atomic
 * counter in it, which evens out the per-CU rate spread of the static split (same results bit for bit). */
int rba_reduce_ws_f32(const float* mask, const float* cls_prob, float* rba, float* sem_seg, int32_t* argmax, int Q, int K,
                      int64_t HW, int score_mode, void* workspace, void* stream);

/* K1 fused with the x4 bilinear upsample (align_corners=False) in front and the crop behind it.
 * mask_lowres [Q,h,w]; the virtual full-resolution map is [Q,4h,4w]; outputs cover rows < crop_h and
 * columns < crop_w of it: rba [crop_h,crop_w], sem_seg [K,crop_h,crop_w] or NULL, argmax or NULL. */
int rba_reduce_up4_f32(const float* mask_lowres, const float* cls_prob, float* rba, float* sem_seg,
                       int32_t* argmax, int Q, int K, int h, int w, int crop_h, int crop_w, int score_mode, void* stream);

/* Bilinear resample, align_corners=False, no antialias (ATen upsample_bilinear2d semantics):
 * in [C,h,w] -> out [C,H,W];  if `add` != NULL (same shape as out): out = resample(in) + add. */
int rba_resample_bilinear_f32(const float* in, const float* add, float* out, int C, int h, int w, int H, int W,
                              void* stream);

/* K2.  Multi-scale deformable attention forward.
 * value [N,S,M,D]; spatial_shapes [L,2] int64 (H_l,W_l); level_start_index [L] int64;
 * sampling_loc [N,Lq,M,L,P,2] (x,y) normalised to [0,1]; attn_weight [N,Lq,M,L,P]; out [N,Lq,M*D].
 * Sample position h = y*H_l - 0.5, w = x*W_l - 0.5; bilinear, taps outside the map contribute 0. */
int rba_ms_deform_attn_fwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* sampling_loc, const float* attn_weight, float* out,
                               int N, int S, int M, int D, int L, int Lq, int P, void* stream);
/* The same op in double precision -- the reference FFI dispatches float and double (ops/src/cuda/ms_deform_attn_cuda.cu:69, AT_DISPATCH_FLOATING_TYPES;
 * ops/test.py:35-47 checks the double path first).  Same shapes and checks; generic kernel. */
int rba_ms_deform_attn_fwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const double* sampling_loc, const double* attn_weight, double* out,
                               int N, int S, int M, int D, int L, int Lq, int P, void* stream);

/* The sampling parameters of MSDeformAttn.forward in one pass (pixel_decoder/ops/modules/ms_deform_attn.py:95-115):
 * raw [rows, M*L*P*2 | M*L*P] = the sampling_offsets and attention_weights Linears evaluated as one; reference_points [rows, L, 2];
 * spatial_shapes [L,2] int64 (H, W) -> loc [rows, M, L, P, 2] = reference + offset / (W_l, H_l), attw [rows, M, L, P] = softmax over L*P. */
int rba_msda_prepare_f32(const float* raw, const float* reference_points, const int64_t* spatial_shapes, float* loc, float* attw,
                         int64_t rows, int M, int L, int P, void* stream);

/* MSDeformAttn.forward's core in one launch: sampling locations + softmax (ms_deform_attn.py:95-115) computed inside the gather kernel
 * from `raw` (layout as rba_msda_prepare_f32), reference_points [N*Lq, L, 2], value [N, S, M, 32] -> out [N, Lq, M*32].
 * head_dim 32, P = 4, L in {1, 3}; bit-identical to rba_msda_prepare_f32 + rba_ms_deform_attn_fwd_f32. */
int rba_msda_fused_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const float* raw,
                       const float* reference_points, float* out, int N, int S, int M, int D, int L, int Lq, int P, void* stream);

/* K3.  Masked multi-head cross attention core (projections are done by the caller):
 * q [B,Q,nH,hd] (already includes the in_proj bias, NOT yet scaled), k,v [B,S,nH,hd];
 * mask_logits [B,Q,S] or NULL: key s is blocked for query q (all heads) iff sigmoid(mask_logits) < 0.5,
 * except that a row with every key blocked attends to all keys.  out [B,Q,nH*hd].
 * scale = hd^-0.5 applied to q.  hd must be 32.
 * workspace: rba_masked_xattn_workspace_bytes(B,Q,S,nH) bytes of scratch enables the split-key matrix-pipe path
 * (partial (m,l,O) per key range + merge); NULL selects the one-workgroup-per-(query,head) kernel. */
int64_t rba_masked_xattn_workspace_bytes(int B, int Q, int S, int nH);
int rba_masked_xattn_f32(const float* q, const float* k, const float* v, const float* mask_logits, float* out,
                         float* workspace, int B, int Q, int S, int nH, int hd, void* stream);

/* K4.  Mask logits: out[b,q,n] = sum_c embed[b,q,c] * feat[b,c,n]   (embed [B,Q,C], feat [B,C,N], out [B,Q,N]). */
int rba_mask_logits_f32(const float* embed, const float* feat, float* out, int B, int Q, int C, int64_t N,
                        void* stream);
/* The same contraction in f16x3 arithmetic (three f16 matrix-pipe products per fp32 product, main + low accumulators; |x| < 65504, NaN
 * beyond): the form the model uses in its default arithmetic mode.  Shapes outside Q <= 112, C % 32 == 0, C <= 256 run rba_mask_logits_f32. */
int rba_mask_logits_f16x3_f32(const float* embed, const float* feat, float* out, int B, int Q, int C, int64_t N,
                              void* stream);

/* K5.  Swin (shifted-)window attention core over a token map, fusing zero-pad to a multiple of the window,
 * cyclic shift, window partition, q*scale @ k^T + relative-position bias (+ shift mask), softmax, @ v,
 * window reverse, un-shift and crop.
 * qkv [B,H*W,3,nH,hd] = Linear(norm1(x)) of the UNPADDED tokens; padded tokens take qkv_bias [3*nH*hd]
 * (= Linear(0)).  bias [nH,ws*ws,ws*ws] gathered relative-position bias.  shift = 0 or ws/2.
 * bias_frag: optional (NULL = unused) copy of `bias` permuted by rba_swin_bias_fragments_f32 into the order the
 * MFMA path's lanes consume it (coalesced loads); computed once per weight load.
 * out [B,H*W,nH*hd] (attention output before the proj Linear).  hd in {16,32,64}; ws*ws <= 256. */
int rba_swin_window_attn_f32(const float* qkv, const float* qkv_bias, const float* bias, const float* bias_frag,
                             float* out, int B, int H, int W, int nH, int hd, int ws, int shift, void* stream);
/* bias [nH,ws*ws,ws*ws] -> frag with rba_swin_bias_fragments_elems(nH, ws) floats. */
int64_t rba_swin_bias_fragments_elems(int nH, int ws);
int rba_swin_bias_fragments_f32(const float* bias, float* frag, int nH, int ws, void* stream);

/* GroupNorm over x [B,C,HW] with G groups (statistics over (C/G) x HW), y = (x-mean)*rstd*gamma[c] + beta[c],
 * optional ReLU.  `workspace` must hold rba_group_norm_workspace_bytes(B,C,HW,G) bytes (per-chunk moments).
 * x and y may alias. */
int64_t rba_group_norm_workspace_bytes(int B, int C, int HW, int G);
int rba_group_norm_f32(const float* x, const float* gamma, const float* beta, float* y, float* workspace,
                       int B, int C, int HW, int G, float eps, int relu, void* stream);

/* Channels-last variants of the two operators above for the token-layout (NHWC) FPN path:
 * group norm over x [B,P,C] (statistics per image and group over P x C/G elements; (C/G) % 4 == 0, 256 % (C/4) == 0, C <= 1024),
 * bilinear resample in [h,w,C] -> out [H,W,C] (+ add [H,W,C]); C % 4 == 0. */
int64_t rba_group_norm_nhwc_workspace_bytes(int B, int P, int C, int G);
int rba_group_norm_nhwc_f32(const float* x, const float* gamma, const float* beta, float* y, float* workspace, int B, int P, int C,
                            int G, float eps, int relu, void* stream);
int rba_resample_bilinear_nhwc_f32(const float* in, const float* add, float* out, int C, int h, int w, int H, int W, void* stream);

/* Fused residual add + LayerNorm over rows of length C (C % 4 == 0, C <= 8192):
 *   s = x (+ t) (+ t_bias[c]);  sum_out = s (optional, may alias x);  y = (s - mean) * rstd * gamma + beta.
 * t, t_bias, sum_out may be NULL.  (swin.py:284-293, msdeformattn.py:134-138, mask2former_transformer_decoder.py:48-58) */
int rba_add_layer_norm_f32(const float* x, const float* t, const float* t_bias, const float* gamma, const float* beta,
                           float* sum_out, float* y, int64_t rows, int C, float eps, void* stream);

/* PatchMerging's 2x2 gather + LayerNorm in one pass (backbone/swin.py:311-337): x [B, H, W, Cin] token-major ->
 * y [B * ceil(H/2) * ceil(W/2), 4*Cin] = LN(cat(x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2])), odd maps zero-padded. */
int rba_merge_layer_norm_f32(const float* x, const float* gamma, const float* beta, float* y, int B, int H, int W, int Cin, float eps,
                             void* stream);

/* Skinny linear layer: out[m,n] = act(sum_k x[m,k] * weight[n,k] + bias[n]), x [M,K] with M <= 128, weight [N,K]
 * (nn.Linear layout), bias [N] or NULL, relu != 0 applies torch.relu exactly (NaN stays NaN -- the f16x3 kernels' loud answer to an out-of-range
 * operand must reach the score map --, +inf stays +inf, -inf becomes 0; every ReLU of this library: csrc/common.h rba_relu).  K % 32 == 0.  For the decoder's 100-query GEMMs
 * (mask2former_transformer_decoder.py:25-212). */
int rba_skinny_linear_f32(const float* x, const float* weight, const float* bias, float* out, int M, int N, int K,
                          int relu, void* stream);
/* The same with the query position embedding folded in: output columns n < add_cols (a multiple of 16) see x + x_add, the others x -- the
 * q / k / v projections of a self-attention layer (q = k = W (tgt + query_pos), v = W tgt; mask2former_transformer_decoder.py:48-58) as ONE
 * launch over the stacked in_proj weight [3E, E] with add_cols = 2E; x_add may be NULL.  seg_n > 0 (N % seg_n == 0) writes the result as
 * N / seg_n contiguous [M, seg_n] blocks (q, k, v separately contiguous) instead of [M, N]. */
int rba_skinny_linear_add_f32(const float* x, const float* x_add, int add_cols, const float* weight, const float* bias, float* out,
                              int M, int N, int K, int relu, int seg_n, void* stream);

/* fp32-accurate Linear on the bf16 matrix pipe: every fp32 value is the exact sum of three bf16 values, and six
 * bf16 x bf16 MFMAs (exact products, fp32 accumulation) reproduce the fp32 product to < 2^-24 relative.
 * rba_split_weight_bf16x3: weight [N,K] fp32 -> `packed`, 6*Np*K bytes (Np = N rounded up to 128, padding rows zero): the
 *                          three bf16 planes tiled as the kernel's LDS image, [Np/128][K/16][3][128][2][8] bf16 with the
 *                          8-element half h of row r in slot h ^ ((r >> 3) & 1).  Once per weight load.  K % 32 == 0.
 * rba_split_linear_f32:    out[m,n] = act(sum_k x[m,k] * weight[n,k] + bias[n]); x [M,K] fp32, `weight_packed` from
 *                          rba_split_weight_bf16x3 for the same (N, K), bias [N] or NULL; act 0 = none, 1 = exact (erf) GELU, 2 = ReLU. */
int rba_split_weight_bf16x3(const float* weight, void* packed, int N, int K, void* stream);
int rba_split_linear_f32(const float* x, const void* weight_packed, const float* bias, float* out, int64_t M, int N, int K,
                         int act, void* stream);

/* The same Linear with THREE f16 MFMAs per fp32 product (csrc/split_linear_h3.h): x = h + 2^-11 l with h = f16(x),
 * l = f16((x - h) 2^11); x w = h_x h_w + 2^-11 (h_x l_w + l_x h_w) to 2^-22 relative per term (as accurate as an fp32 GEMM,
 * whose accumulation rounding dominates), for |x|, |w| < 65504 -- beyond f16's range the output row is NaN.
 * rba_split_weight_f16x2:     weight [N,K] fp32 -> `packed`, 4*Np*K bytes: [Np/128][K/16][2][128][2][8] f16, sub-stage 2b+g of the
 *                             32-wide k block b holds for row r, in slot h ^ ((r >> 3) & 1), k = 32b + 16h + 8g + (0..7).  K % 32 == 0.
 * rba_split_linear_f16x3_f32: as rba_split_linear_f32 with `weight_packed` from rba_split_weight_f16x2. */
int rba_split_weight_f16x2(const float* weight, void* packed, int N, int K, void* stream);
int rba_split_linear_f16x3_f32(const float* x, const void* weight_packed, const float* bias, float* out, int64_t M, int N, int K,
                               int act, void* stream);
/* out = (residual + x W^T) + bias, no activation: the `x = x + proj(...)` / `x = x + fc2(...)` of a transformer block folded into the
 * GEMM epilogue (backbone/swin.py:284-293), in the order of operations of rba_add_layer_norm_f32's own add (bit-identical to it).
 * residual [M,N]; `out` may be `residual` itself. */
int rba_split_linear_f16x3_res_f32(const float* x, const void* weight_packed, const float* bias, const float* residual, float* out,
                                   int64_t M, int N, int K, void* stream);

/* Split activations: a [M, K] fp32 activation matrix held ONLY as the f16x3 GEMM's A operand, written by its producer (one split per
 * element instead of one per column tile of every consumer; contiguous 1 KiB wave loads and no arithmetic in the GEMM).  Image =
 * ceil(M / 32) * 32 * K * 4 bytes: per (32-row group rg, 32-wide block b of K) four 1 KiB pieces [h g0 | l g0 | h g1 | l g1], each
 * [half][row & 31][8 f16] with k = 32 b + 16 half + 8 g + i; h = f16(x) (rne), l = f16((x - h) * 2^11).  Rows M .. are padding and never
 * written.  K % 32 == 0.
 *   rba_add_layer_norm_frag_f32     = rba_add_layer_norm_f32 with y written as that image (norm1 -> qkv, norm2 -> fc1 of
 *                                     backbone/swin.py:235-295)
 *   rba_split_linear_f16x3_frag_f32 = rba_split_linear_f16x3_f32 / _res_f32 (residual non-NULL, act 0) reading it; bit-identical to the
 *                                     fp32-input entry points on the same values. */
int rba_add_layer_norm_frag_f32(const float* x, const float* t, const float* t_bias, const float* gamma, const float* beta,
                                float* sum_out, void* y_frag, int64_t rows, int C, float eps, void* stream);
int rba_split_linear_f16x3_frag_f32(const void* x_frag, const void* weight_packed, const float* bias, const float* residual, float* out,
                                    int64_t M, int N, int K, int act, void* stream);
/*   rba_swin_window_attn_split_out_f32 = rba_swin_window_attn_f32 with the attention output written as the proj Linear's split image
 *                                     (backbone/swin.py:165-168 -> :169); head_dim 32, 12 x 12 windows, bias_frag required. */
int rba_swin_window_attn_split_out_f32(const float* qkv, const float* qkv_bias, const float* bias_frag, void* out_frag, int B, int H, int W,
                                       int nH, int hd, int ws, int shift, void* stream);
/* K7 -- the attention half of a Swin block in one launch (backbone/swin.py:235-284: norm1 -> pad / roll / window_partition ->
 * WindowAttention (:131-171: qkv, relative-position bias, SW-MSA mask of :413-440, softmax, proj) -> window_reverse / un-roll / crop ->
 * `x = shortcut + x`), and optionally norm2 of :284-293:
 *     x [B, H*W, C] <- x + proj(window_attention(qkv(norm1(x))));    y2 [B, H*W, C] = norm2(x) when y2 != NULL
 * One workgroup per 12 x 12 window; f16x3 arithmetic (|values| < 65504, NaN beyond).  weight_image = rba_swin_attn_block_pack_f32 of
 * qkv.weight [3C, C] and proj.weight [C, C] (rba_swin_attn_block_weight_bytes(C) bytes); bias_frag = rba_swin_bias_fragments_f32 of the
 * gathered relative-position bias [C/32, 144, 144].  rba_swin_attn_block_supported(C, ws) says which geometries have a kernel
 * (hipErrorInvalidValue otherwise: the caller keeps the unfused sequence rba_add_layer_norm -> rba_split_linear -> rba_swin_window_attn
 * -> rba_split_linear). */
int rba_swin_attn_block_supported(int C, int ws);
int64_t rba_swin_attn_block_weight_bytes(int C);
int rba_swin_attn_block_pack_f32(const float* qkv_weight, const float* proj_weight, void* image, int C, void* stream);
int rba_swin_attn_block_f32(float* x, float* y2, const float* norm1_weight, const float* norm1_bias, float eps1, const void* weight_image,
                            const float* qkv_bias, const float* bias_frag, const float* proj_bias, const float* norm2_weight,
                            const float* norm2_bias, float eps2, int B, int H, int W, int C, int ws, int shift, void* stream);
/* K7, attention only: the same kernel without proj / residual -- norm1 -> qkv -> (shifted-)window attention, its output written as the proj Linear's split
 * image (rba_split_linear_f16x3_frag_f32 reads it; layout of rba_swin_window_attn_split_out_f32): the qkv tensor and norm1's output never exist.  For the
 * widths whose proj accumulators do not fit beside the resident rows (C = 256: Swin-B stage 2; C = 192: Swin-L stage 1); x is only read.  out_frag: ceil(B H W / 32) * 32 * C * 4 bytes. */
int rba_swin_attn_qkv_supported(int C, int ws);
int rba_swin_attn_qkv_split_out_f32(const float* x, void* out_frag, const float* norm1_weight, const float* norm1_bias, float eps1,
                                    const void* weight_image, const float* qkv_bias, const float* bias_frag, int B, int H, int W, int C, int ws,
                                    int shift, void* stream);
/*   rba_resample_bilinear_nhwc_split_out_f32 -> rba_conv3x3_nhwc_f16x3_split_in_f32 = the FPN's `lateral + F.interpolate(prev)` sum handed to
 *                                     its 3 x 3 output convolution as a split image of [B H W, C] rows (pixel_decoder/msdeformattn.py:352-361);
 *                                     the convolution (>= 256 tiles of 128 x 128) gathers the pieces of the neighbour pixels' rows. */
int rba_resample_bilinear_nhwc_split_out_f32(const float* in, const float* add, void* out_frag, int C, int h, int w, int H, int W,
                                             int64_t row0, void* stream);

/* The FPN's top-down step with its GroupNorms folded into the loads (pixel_decoder/msdeformattn.py:352-358): in = the previous level's raw
 * output-convolution result (normalised + ReLU'd on the fly when in_mr is given), add = the raw lateral-convolution result (normalised on
 * the fly when add_mr is given); *_mr = [G][2] (mean, rstd) of the image from rba_group_norm_nhwc_stats_f32.  out = fp32 [H, W, C] or, with
 * split_out, rows row0 .. of the split image the 3 x 3 convolution reads.  The arithmetic of the unfused GroupNorm + resample launches (equal up to fma contraction: one ulp). */
int rba_resample_bilinear_nhwc_gn_f32(const float* in, const float* in_mr, const float* in_gamma, const float* in_beta, int in_relu,
                                      const float* add, const float* add_mr, const float* add_gamma, const float* add_beta, void* out,
                                      int split_out, int C, int G, int h, int w, int H, int W, int64_t row0, void* stream);
/* (mean, rstd) per image and group of channels-last x [B, P, C]: the statistics half of rba_group_norm_nhwc_f32. */
int rba_group_norm_nhwc_stats_f32(const float* x, float* mr, float* workspace, int B, int P, int C, int G, float eps, void* stream);
int rba_conv3x3_nhwc_f16x3_split_in_f32(const void* x_frag, const void* weight_packed, const float* bias, float* out, int B, int H, int W,
                                        int C, int N, void* stream);
/*   rba_swin_mlp_fused_f16x3_f32 = the whole Mlp + residual of a Swin block with C = 128 in one kernel (csrc/mlp_fused_h3.h): the [M, HID]
 *                                     hidden tensor never leaves the registers (fc1's transposed accumulators become fc2's A fragments after
 *                                     one v_permlane32_swap); bit-identical to the two-kernel hand-over below. */
int rba_swin_mlp_fused_f16x3_f32(const float* x, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2,
                                 const float* residual, float* out, int64_t M, int C, int HID, void* stream);
/*   rba_swin_mlp_fused_ln_f16x3_f32 = the same kernel with norm2 in its prologue: x <- x + fc2(GELU(fc1(LayerNorm(x)))) in place (backbone/swin.py:293). */
int rba_swin_mlp_fused_ln_f16x3_f32(float* x, const float* norm_weight, const float* norm_bias, float eps, const void* w1_packed, const float* b1,
                                    const void* w2_packed, const float* b2, int64_t M, int C, int HID, void* stream);
/*   rba_split_linear_f16x3_gelu_split_out = GELU(x W^T + bias) written as the split image of the NEXT Linear (Mlp.fc1 -> fc2,
 *                                     backbone/swin.py:35-41): the GEMM runs with its MFMA operands swapped (D^T = W x^T), so a lane ends
 *                                     up with consecutive output channels of one row and stores 16-byte pieces.  x: fp32 rows
 *                                     (x_is_split 0) or a split image (1).  N % 32 == 0; out_frag ceil(M / 32) * 32 * N * 4 bytes. */
int rba_split_linear_f16x3_gelu_split_out(const void* x, int x_is_split, const void* weight_packed, const float* bias, void* out_frag,
                                          int64_t M, int N, int K, void* stream);

/* The same GEMM with NHWC rows in and NCHW out: out[(b*N + n)*P + p] = sum_k x[b*P + p, k] * weight[n, k] + bias[n],
 * P = rows_per_image, M % P == 0 (the mask-feature 1x1 convolution of pixel_decoder/msdeformattn.py:298-306). */
int rba_split_linear_nchw_out_f32(const float* x, const void* weight_packed, const float* bias, float* out, int64_t M, int N,
                                  int K, int rows_per_image, void* stream);

/* The same operator (NHWC rows in, NCHW out) on the f16x3 kernel; weight_packed = rba_split_weight_f16x2.  The mask-feature projection
 * `self.mask_features(y)` (pixel_decoder/msdeformattn.py:362). */
int rba_split_linear_nchw_out_f16x3_f32(const float* x, const void* weight_packed, const float* bias, float* out, int64_t M, int N, int K,
                                        int rows_per_image, void* stream);

/* The same projection on a RAW convolution output x whose GroupNorm(G) (+ ReLU when relu != 0) is applied while the rows are loaded:
 * `self.mask_features(self.layer_1(y))` (pixel_decoder/msdeformattn.py:357-362; Conv2d(norm=GN, activation=relu) :278-297) without the pass
 * that writes the normalised map.  mr [B][G][2] = (mean, rstd) from rba_group_norm_nhwc_stats_f32; gamma / beta [K]; K % G == 0,
 * (K / G) % 4 == 0, rows_per_image % 128 == 0.  Bit-identical to rba_group_norm_nhwc_f32 followed by the entry above. */
int rba_split_linear_nchw_out_gn_f16x3_f32(const float* x, const float* mr, const float* gamma, const float* beta, int G, int relu,
                                           const void* weight_packed, const float* bias, float* out, int64_t M, int N, int K,
                                           int rows_per_image, void* stream);

/* 3x3 / stride 1 / pad 1 convolution over NHWC activations as an implicit GEMM on the same kernel:
 * x [B,H,W,C] -> out [B,H,W,N]; weight_packed = rba_split_weight_bf16x3 of the [N, 9*C] matrix w[n][(3*ky + kx)*C + c]
 * (conv weight [N,C,3,3] permuted to [N,3,3,C]); bias [N] or NULL.  C % 32 == 0.
 * (the FPN output convolutions `layer_{j}` of pixel_decoder/msdeformattn.py:278-297, 357-360) */
int rba_conv3x3_nhwc_f32(const float* x, const void* weight_packed, const float* bias, float* out, int B, int H, int W, int C,
                         int N, void* stream);

/* The same convolution on the f16x3 kernel: weight_packed = rba_split_weight_f16x2 of the [N, 9*C] matrix above.  C % 32 == 0. */
int rba_conv3x3_nhwc_f16x3_f32(const float* x, const void* weight_packed, const float* bias, float* out, int B, int H, int W, int C,
                               int N, void* stream);

/* GroupNorm statistics without a pass over the tensor (round 4): the FPN's lateral 1x1 convolution (fp32 rows in, K <= 256) and its 3x3 output
 * convolution (split image in) also leave, per 128-row tile and group of N/G consecutive output channels, the moments (n, mean, M2) of their OUTPUT:
 * moments [B][G][rows_per_image/128][3] floats.  rba_group_norm_nhwc_merge_f32 merges them (Chan, in double) into mr [B][G][2] = (mean, rstd) --
 * what rba_group_norm_nhwc_stats_f32 computes by reading the output again (pixel_decoder/msdeformattn.py:222-235, 278-297: Conv2d(norm=GroupNorm(32, C))).
 * N % 128 == 0, N/G in {4, 8, 16, 32}, rows_per_image % 128 == 0 (H*W for the convolution).  The outputs are those of the entries without moments, bit for bit. */
int rba_split_linear_f16x3_gn_moments_f32(const float* x, const void* weight_packed, const float* bias, float* out, int64_t M, int N, int K,
                                          int rows_per_image, int G, float* moments, void* stream);
int rba_conv3x3_nhwc_f16x3_split_in_gn_moments_f32(const void* x_frag, const void* weight_packed, const float* bias, float* out, int B, int H,
                                                   int W, int C, int N, int G, float* moments, void* stream);
int rba_group_norm_nhwc_merge_f32(const float* moments, float* mr, int B, int G, int splits, float eps, void* stream);

/* Front end of the Swin path in one pass: (image - mean) / std, ImageList zero padding to Hp x Wp, and the im2col of PatchEmbed's
 * 4x4 / stride-4 convolution (maskformer_model.py:255-257, backbone/swin.py:479-495): image [3,h,w] (device; uint8 or fp32) ->
 * out [(Hp/4)*(Wp/4), 64] fp32 with out[token][c*16 + ky*4 + kx] and columns 48..63 zero; mean / std: 3 host floats each.
 * Followed by rba_split_linear_f16x3_f32 with the [E, 64] zero-padded projection weight and by rba_add_layer_norm_f32. */
int rba_patch_im2col_u8(const uint8_t* image, float* out, int h, int w, int Hp, int Wp, const float* mean, const float* std, void* stream);
int rba_patch_im2col_f32(const float* image, float* out, int h, int w, int Hp, int Wp, const float* mean, const float* std, void* stream);

/* Gaussian smoothing of the score map (the evaluator's optional transforms.GaussianBlur(7, sigma=1), support.py:366-383):
 * out[H,W] = correlation of in[H,W] (reflect-padded by kernel_size/2) with the normalised outer-product kernel of
 * exp(-0.5 (x/sigma)^2), x = -(k-1)/2 .. (k-1)/2.  kernel_size odd, <= 15; in != out. */
int rba_gaussian_blur_f32(const float* in, float* out, int H, int W, int kernel_size, float sigma, void* stream);

/* Two small steps of the masked decoder (decoder_small.hip):
 * rba_quad_mean_f32:         out[r][s] = ((v[r][0][s] + v[r][1][s]) + (v[r][2][s] + v[r][3][s])) * 0.25 for v [rows, 4, S]: the bilinear sample at the
 *                            centre of a 2 x 2 cell = the attention-mask logits of an intermediate decoder layer from mask logits evaluated at the
 *                            four source pixels of every attention cell (what F.interpolate of the whole map gives, bit for bit;
 *                            mask2former_transformer_decoder.py:472-489).
 * rba_softmax_drop_last_f32: prob[r][k] = softmax(logits[r][:])[k] for k < K1 - 1 (F.softmax(mask_cls, -1)[..., :-1], maskformer_model.py:381-383:
 *                            K1's class probabilities without the "no object" column).  2 <= K1 <= 64. */
int rba_quad_mean_f32(const float* v, float* out, int64_t rows, int S, void* stream);
int rba_softmax_drop_last_f32(const float* logits, float* prob, int64_t rows, int K1, void* stream);

/* Open-set panoptic epilogue of the RbA map (MaskFormer.panoptic_inference, maskformer_model.py:454-481):
 * rba_threshold_u8:   out[i] = score[i] > threshold
 * rba_morph3x3_u8:    3x3 box erosion (dilate = 0) or dilation (dilate = 1) of a 0/1 map, out-of-image neighbours ignored
 *                     (cv2.morphologyEx's default border); opening = erode, dilate; closing = dilate, erode.  in != out.
 * rba_ccl4_roots_i32: 4-connected components: roots[p] = smallest linear index of p's component, -1 on background
 *                     (numbering the distinct roots in increasing order gives cv2.connectedComponents' raster-order labels). */
int rba_threshold_u8(const float* score, uint8_t* out, int64_t n, float threshold, void* stream);
int rba_morph3x3_u8(const uint8_t* in, uint8_t* out, int H, int W, int dilate, void* stream);
int rba_ccl4_roots_i32(const uint8_t* mask, int32_t* roots, int H, int W, void* stream);

/* DenseHybrid anomaly head (mask2former_transformer_decoder.py:216-230, 365-366, 467-468; maskformer_model.py:303-305).
 * rba_bn_relu_conv1x1_f32:      out[b,o,p] = bias[o] + sum_c weight[o,c] * relu(x[b,c,p] * scale[c] + shift[c]); x [B,C,P], out [B,O,P],
 *                               O in {1,2,4}; scale/shift = eval-mode BatchNorm2d folded by the caller; bias may be NULL.
 * rba_resample_bilinear_ac_f32: F.interpolate(mode="bilinear", align_corners=True): x [C,h,w] -> out [C,H,W]. */
int rba_bn_relu_conv1x1_f32(const float* x, const float* scale, const float* shift, const float* weight, const float* bias, float* out,
                            int B, int C, int O, int64_t P, void* stream);
int rba_resample_bilinear_ac_f32(const float* x, float* out, int C, int h, int w, int H, int W, void* stream);

/* Row-complete token Linear for the path's SMALL Linears (fewer than 64 tiles of 128 x 128: not served by rba_split_linear_*): the
 * MSDeformAttn encoder's value / sampling / output projections and linear2 (pixel_decoder/msdeformattn.py:101-140,
 * ops/modules/ms_deform_attn.py:95-121), the decoder's key / value projections of the memory
 * (transformer_decoder/mask2former_transformer_decoder.py:83-143), the 1x1 input projection (msdeformattn.py:328-333).  f16x3 arithmetic
 * (fp32-GEMM accuracy for |v| < 65504, NaN beyond).  One workgroup owns 16 rows and all N <= 256 columns:
 * rba_token_linear_pack_f16x2: weight [N,K] fp32 -> `packed`, ceil(N/16) * (K/32) * 2048 bytes, [N/16][K/32][h | l][64 lanes][8 f16], lane
 *                             (n = lane % 16, kb = lane / 16) holding W[16 nt + n][32 b + 8 kb + 0..7]; rows beyond N zero.  K % 32 == 0.
 * rba_token_linear_f32:       out[m,:] = act((x[m,:] + x_add[m,:]) W^T + bias), act 0 none / 2 ReLU; with ln_weight != NULL instead
 *                             out[m,:] = LayerNorm(residual[m,:] + (x W^T + bias)) * ln_weight + ln_bias (N % 16 == 0, act 0): the
 *                             post-norm `src = norm(src + dropout(src2))` of msdeformattn.py:134-138 in the GEMM's epilogue.
 *                             x_add, bias may be NULL; residual only with ln_weight.  out must not alias x.
 * rba_token_linear_multi_f32: up to three Linears over the SAME rows x [M,K] in ONE launch, problem i:
 *                             out_i[m * ld_out + n] = act((x[m,:] + x_add_i[m,:]) W_i^T + bias_i)[n], n < N_i (x_add, bias may be NULL; N % 4 == 0,
 *                             ld_out >= N lets a problem fill a column slice of a wider tensor): value = value_proj(src) beside the
 *                             sampling_offsets / attention_weights Linears of `src + pos` (ms_deform_attn.py:102-110), or k = k_proj(memory + pos)
 *                             beside v = v_proj(memory) (mask2former_transformer_decoder.py:106-118). */
typedef struct {
  const float* x_add;  /* [M,K] or NULL */
  const void* packed;  /* rba_token_linear_pack_f16x2 of the weight [N,K] */
  const float* bias;   /* [N] or NULL */
  float* out;
  int N;
  int ld_out;          /* row stride of out, in floats */
  int act;             /* 0 none, 2 ReLU */
} rba_token_linear_problem;
int rba_token_linear_pack_f16x2(const float* weight, void* packed, int N, int K, void* stream);
int rba_token_linear_f32(const float* x, const float* x_add, const void* packed, const float* bias, const float* residual,
                         const float* ln_weight, const float* ln_bias, float ln_eps, float* out, int64_t M, int N, int K, int act,
                         void* stream);
int rba_token_linear_multi_f32(const float* x, const rba_token_linear_problem* problems, int n_problems, int64_t M, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RBA_HIP_H */
