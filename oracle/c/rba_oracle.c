/* ORACLE -- test infrastructure, not product code.
 *
 * Plain-C scalar restatement of the two kernels the reference itself implements natively or names as its score:
 *   oracle_rba_reduce_f32      mask2former/maskformer_model.py:381-386 (softmax'd class probs x sigmoid(mask),
 *                              summed over queries) + evaluate_ood.py:150 (-sum_k tanh) + support.py:385-388 (argmax)
 *   oracle_ms_deform_attn_f32  pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:38-89,242-304 (the arithmetic of the
 *                              reference's CUDA kernel, = ops/functions/ms_deform_attn_func.py:52-72 on CPU)
 * Pinned by tests/test_oracle_golden.py against tests/golden/g1_*.npz and g2_*.npz (produced by the reference's code).
 * Built by oracle/Makefile into oracle/_build/liboracle.so; loaded only by tests/ and bench.py's cpu_baseline leg.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* mask [Q,HW], prob [Q,K] -> sem [K,HW] (may be NULL), rba [HW], argmax [HW] (may be NULL).  Ascending-q order. */
int oracle_rba_reduce_f32(const float* mask, const float* prob, float* rba, float* sem, int32_t* argmax, int Q, int K,
                          int64_t HW) {
  float* acc = (float*)malloc(sizeof(float) * (size_t)K);
  if (!acc) return 1;
  for (int64_t p = 0; p < HW; ++p) {
    for (int k = 0; k < K; ++k) acc[k] = 0.0f;
    for (int q = 0; q < Q; ++q) {
      const float s = 1.0f / (1.0f + expf(-mask[(int64_t)q * HW + p]));
      for (int k = 0; k < K; ++k) acc[k] = fmaf(prob[q * K + k], s, acc[k]);
    }
    float r = 0.0f, best = acc[0];
    int bi = 0;
    for (int k = 0; k < K; ++k) {
      r -= tanhf(acc[k]);
      if (acc[k] > best) { best = acc[k]; bi = k; }
      if (sem) sem[(int64_t)k * HW + p] = acc[k];
    }
    rba[p] = r;
    if (argmax) argmax[p] = bi;
  }
  free(acc);
  return 0;
}

static float bilinear_tap(const float* v, int64_t stride, int H, int W, float h, float w, int c) {
  const int h0 = (int)floorf(h), w0 = (int)floorf(w);
  const float lh = h - h0, lw = w - w0, hh = 1.0f - lh, hw = 1.0f - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h0 >= 0 && w0 >= 0) v1 = v[((int64_t)h0 * W + w0) * stride + c];
  if (h0 >= 0 && w0 + 1 <= W - 1) v2 = v[((int64_t)h0 * W + w0 + 1) * stride + c];
  if (h0 + 1 <= H - 1 && w0 >= 0) v3 = v[((int64_t)(h0 + 1) * W + w0) * stride + c];
  if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) v4 = v[((int64_t)(h0 + 1) * W + w0 + 1) * stride + c];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

/* value [N,S,M,D], shapes [L,2], lsi [L], loc [N,Lq,M,L,P,2], w [N,Lq,M,L,P] -> out [N,Lq,M*D] */
int oracle_ms_deform_attn_f32(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                              const float* w, float* out, int N, int S, int M, int D, int L, int Lq, int P) {
  for (int n = 0; n < N; ++n)
    for (int q = 0; q < Lq; ++q)
      for (int m = 0; m < M; ++m)
        for (int c = 0; c < D; ++c) {
          float col = 0.0f;
          for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const float* v = value + ((int64_t)n * S + lsi[l]) * M * D + (int64_t)m * D;
            for (int p = 0; p < P; ++p) {
              const int64_t i = ((((int64_t)n * Lq + q) * M + m) * L + l) * P + p;
              const float h_im = loc[2 * i + 1] * H - 0.5f, w_im = loc[2 * i] * W - 0.5f;
              if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                col += bilinear_tap(v, (int64_t)M * D, H, W, h_im, w_im, c) * w[i];
            }
          }
          out[(((int64_t)n * Lq + q) * M + m) * D + c] = col;
        }
  return 0;
}
