/*
 * aslp_oracle_temporal.c -- TEST INFRASTRUCTURE ONLY (see aslp_oracle.h).
 *
 * Plain-C restatement of the two depthwise temporal components, op by op in the reference's order:
 *   RowConvolution  src/aslp-nnet/nnet-row-convolution.cc:105-176
 *   CompactFsmn     src/aslp-nnet/nnet-cfsmn-component.h:170-262
 * Pinning: both are checked against their op sequences issued on the reference's own CuMatrix library
 * (oracle/gen_cumatrix_blas_golden.cpp -> tests/golden/cumatrix_blas_ops.bin, tests/test_oracle_ref_blas_cpu.py: output, input diff,
 * tap gradients, ragged lengths for RowConvolution); tests/test_oracle_temporal_cpu.py checks the backward passes against central
 * differences besides and documents where the reference deliberately is not the exact gradient.
 */
#include <stdlib.h>
#include <string.h>

#include "aslp_oracle.h"

/* ---- RowConvolution ---------------------------------------------------------------------------------
 * in/out [T*S x D], row = t*S + s; w [D x (K+1)] dense; in_buf [S*(T+K) x D], row = s*(T+K) + t. */
void orc_rowconv_propagate(const float *w, int D, int K, const float *in, int ldi, int T, int S, const int32_t *seq_len,
                           float *in_buf, float *out, int ldo) {
  const int Ts = T + K;
  memset(in_buf, 0, sizeof(float) * (size_t)Ts * S * D);
  for (int s = 0; s < S; s++) {
    const int L = seq_len[s];
    for (int t = 0; t < L + K; t++) { /* frames past the end repeat the last one (:118-126) */
      const int src = t < L ? t : L - 1;
      memcpy(in_buf + (size_t)(s * Ts + t) * D, in + (size_t)(src * S + s) * ldi, sizeof(float) * D);
    }
    for (int t = 0; t < L; t++) { /* out.Row = diag(w * yh), yh = in_buf rows t..t+K (:128-133) */
      const float *yh = in_buf + (size_t)(s * Ts + t) * D;
      float *o = out + (size_t)(t * S + s) * ldo;
      for (int d = 0; d < D; d++) {
        float acc = 0.0f;
        for (int k = 0; k <= K; k++) acc += w[(size_t)d * (K + 1) + k] * yh[(size_t)k * D + d];
        o[d] = 1.0f * acc + 0.0f;
      }
    }
  }
}

/* in_diff rows of frames t >= seq_len[s] are left untouched (the caller zeroes in_diff, nnet-component.h:335).
 * w_diff is overwritten. */
void orc_rowconv_backpropagate(const float *w, int D, int K, const float *out_diff, int ldod, int T, int S, const int32_t *seq_len,
                               const float *in_buf, float *in_diff_buf, float *w_diff, float *in_diff, int ldid) {
  const int Ts = T + K;
  memset(in_diff_buf, 0, sizeof(float) * (size_t)Ts * S * D);
  memset(w_diff, 0, sizeof(float) * (size_t)D * (K + 1));
  for (int s = 0; s < S; s++)
    for (int t = 0; t < seq_len[s]; t++) {
      const float *od = out_diff + (size_t)(t * S + s) * ldod;
      const float *yh = in_buf + (size_t)(s * Ts + t) * D;
      float *yd = in_diff_buf + (size_t)(s * Ts + t) * D;
      for (int d = 0; d < D; d++)
        for (int k = 0; k <= K; k++) {
          float c = 0.0f + 1.0f * w[(size_t)d * (K + 1) + k]; /* conv_diff_buf = w; MulRowsVec(od) (:156-158) */
          c *= od[d];
          yd[(size_t)k * D + d] += 1.0f * c;
          float g = 0.0f + 1.0f * yh[(size_t)k * D + d]; /* conv_diff_buf = yh^T; MulRowsVec(od) (:161-164) */
          g *= od[d];
          w_diff[(size_t)d * (K + 1) + k] += 1.0f * g;
        }
    }
  for (int s = 0; s < S; s++)
    for (int t = 0; t < seq_len[s]; t++) /* what fell on the replicated tail frames is dropped (:168-174) */
      memcpy(in_diff + (size_t)(t * S + s) * ldid, in_diff_buf + (size_t)(s * Ts + t) * D, sizeof(float) * D);
}

void orc_rowconv_update(float *w, float *w_corr, const float *w_diff, int D, int K, float lr, float mmt) { /* :178-186 */
  for (size_t i = 0; i < (size_t)D * (K + 1); i++) {
    w_corr[i] *= mmt;
    w_corr[i] += 1.0f * w_diff[i];
    w[i] += -lr * w_corr[i];
  }
}

/* ---- CompactFsmn: one sequence of T rows, coef [(P+F+1) x D] ---------------------------------------- */
void orc_fsmn_propagate(const float *coef, int D, int P, int F, const float *in, int ldi, int T, float *out, int ldo) {
  const int C = P + F + 1;
  float *pad = (float *)calloc((size_t)(T + C - 1) * D, sizeof(float));
  float *tmp = (float *)calloc((size_t)T * C * D, sizeof(float));
  for (int t = 0; t < T; t++) memcpy(pad + (size_t)(P + t) * D, in + (size_t)t * ldi, sizeof(float) * D);
  orc_add_conv_mat_mat_elements(tmp, D, D, pad, D, T + C - 1, coef, D, C, 1.0f, 0.0f);
  for (int t = 0; t < T; t++) memcpy(out + (size_t)t * ldo, in + (size_t)t * ldi, sizeof(float) * D);
  orc_add_row_sum_mat(out, ldo, T, D, tmp, D, T * C, 1.0f, 1.0f);
  free(pad);
  free(tmp);
}

/* coef_corr is overwritten (AddRowSumMat beta 0: no momentum, :219), then clipped */
void orc_fsmn_backpropagate(const float *coef, int D, int P, int F, const float *in, int ldi, const float *out_diff, int ldod, int T,
                            float clip, float *coef_corr, float *in_diff, int ldid) {
  const int C = P + F + 1;
  float *pad = (float *)calloc((size_t)(T + C - 1) * D, sizeof(float));
  float *tmp = (float *)calloc((size_t)T * C * D, sizeof(float));
  float *rev = (float *)malloc(sizeof(float) * (size_t)C * D);
  for (int t = 0; t < T; t++) memcpy(pad + (size_t)(P + t) * D, in + (size_t)t * ldi, sizeof(float) * D);
  for (int i = 0; i < C; i++) /* tmp rows i*T..: pad[i..i+T) .* out_diff (:213-217) */
    for (int t = 0; t < T; t++)
      for (int d = 0; d < D; d++)
        tmp[(size_t)(i * T + t) * D + d] = 1.0f * pad[(size_t)(i + t) * D + d] * out_diff[(size_t)t * ldod + d] + 0.0f;
  orc_add_row_sum_mat(coef_corr, D, C, D, tmp, D, T * C, 1.0f, 0.0f);
  memset(pad, 0, sizeof(float) * (size_t)(T + C - 1) * D);
  for (int t = 0; t < T; t++) memcpy(pad + (size_t)(F + t) * D, out_diff + (size_t)t * ldod, sizeof(float) * D);
  for (int i = 0; i < C; i++) memcpy(rev + (size_t)(C - 1 - i) * D, coef + (size_t)i * D, sizeof(float) * D);
  orc_add_conv_mat_mat_elements(tmp, D, D, pad, D, T + C - 1, rev, D, C, 1.0f, 0.0f);
  for (int t = 0; t < T; t++) memcpy(in_diff + (size_t)t * ldid, out_diff + (size_t)t * ldod, sizeof(float) * D);
  orc_add_row_sum_mat(in_diff, ldid, T, D, tmp, D, T * C, 1.0f, 1.0f);
  if (clip > 0.0f)
    for (size_t i = 0; i < (size_t)C * D; i++) {
      if (coef_corr[i] < -clip) coef_corr[i] = -clip;
      if (coef_corr[i] > clip) coef_corr[i] = clip;
    }
  free(pad);
  free(tmp);
  free(rev);
}

void orc_fsmn_update(float *coef, const float *coef_corr, int D, int P, int F, float lr) { /* :264-268, lr = learn_rate * coef */
  for (size_t i = 0; i < (size_t)(P + F + 1) * D; i++) coef[i] += -lr * coef_corr[i];
}
