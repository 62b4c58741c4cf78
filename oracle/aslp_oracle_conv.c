/*
 * aslp_oracle_conv.c -- TEST INFRASTRUCTURE ONLY (see aslp_oracle.h).
 *
 * CPU restatement of the front-end components of the CNN / cFSMN recipes, in the reference's own structure (per-patch column
 * blocks, one product per patch, AddCols passes over the reversed column map, mask + AddMat per (pool, member)):
 *   LinearTransform          aslp-nnet/nnet-linear-transform.h:127-160
 *   ConvolutionalComponent   aslp-nnet/nnet-convolutional-component.h:268-470
 *   MaxPoolingComponent      aslp-nnet/nnet-max-pooling-component.h:101-162
 *   LengthNormComponent      aslp-nnet/nnet-various.h:338-358
 *   PnormComponent / Maxout  aslp-nnet/nnet-activation.h:341-373 -> matrix/kaldi-matrix.cc:1071-1138, 2530-2558, kaldi-vector.cc:520-557
 * Pinned by tests/golden/component_ops.bin (the same op sequences issued on the reference's CuMatrix library, generator
 * oracle/gen_component_golden.cpp) and by the reference's own known answers (aslp-nnet/nnet-component-test.cc:53-206, extracted as
 * data into tests/golden/component_known_answers.json by oracle/gen_component_known_answers.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "aslp_oracle.h"

/* ---- LinearTransform ------------------------------------------------------------------------------------------------ */
void orc_linear_propagate(float *out, int ldo, const float *in, int ldi, int rows, const float *W, int ldw, int in_dim, int out_dim) {
  orc_add_mat_mat(out, rows, out_dim, ldo, 1.0f, in, ldi, 0, W, ldw, 1, in_dim, 0.0f);   /* :129 */
}
void orc_linear_backpropagate(float *in_diff, int ldid, const float *out_diff, int ldod, int rows, const float *W, int ldw, int in_dim, int out_dim) {
  orc_add_mat_mat(in_diff, rows, in_dim, ldid, 1.0f, out_diff, ldod, 0, W, ldw, 0, out_dim, 0.0f);   /* :135 */
}
/* :139-160; o->bias_learn_rate_coef and o->max_norm are not used by this component */
void orc_linear_update(float *W, int ldw, float *W_corr, int ldc, const float *input, int ldi, const float *diff, int ldd, int rows, int in_dim,
                       int out_dim, const orc_affine_opts *o) {
  const float lr = o->learn_rate, mmt = o->momentum, l2 = o->l2_penalty, l1 = o->l1_penalty;
  orc_add_mat_mat(W_corr, out_dim, in_dim, ldc, 1.0f, diff, ldd, 1, input, ldi, 0, rows, mmt);   /* :149 */
  if (l2 != 0.0f) {   /* :151-153  W += (-lr l2 N) W */
    const float a = -lr * l2 * rows;
    for (int r = 0; r < out_dim; r++)
      for (int c = 0; c < in_dim; c++) W[(size_t)r * ldw + c] += a * W[(size_t)r * ldw + c];
  }
  if (l1 != 0.0f) orc_regularize_l1(W, ldw, W_corr, ldc, out_dim, in_dim, lr * l1 * rows, lr);   /* :155-157 */
  {
    const float a = -lr * o->learn_rate_coef;   /* :159 */
    for (int r = 0; r < out_dim; r++)
      for (int c = 0; c < in_dim; c++) W[(size_t)r * ldw + c] += a * W_corr[(size_t)r * ldc + c];
  }
}

/* ---- ConvolutionalComponent --------------------------------------------------------------------------------------- */
static int32_t *conv_column_map(int num_patches, int num_splice, int patch_dim, int patch_step, int patch_stride) {   /* :318-325 */
  const int filter_dim = num_splice * patch_dim;
  int32_t *map = (int32_t *)malloc(sizeof(int32_t) * (size_t)filter_dim * num_patches);
  int index = 0;
  for (int p = 0; p < num_patches; p++)
    for (int s = 0; s < num_splice; s++)
      for (int d = 0; d < patch_dim; d++, index++) map[index] = p * patch_step + s * patch_stride + d;
  return map;
}
/* patches: [rows x filter_dim * num_patches] (vectorized_feature_patches_, ld = that width); out [rows x num_filters * num_patches] */
void orc_conv_propagate(float *out, int ldo, float *patches, const float *in, int ldi, int rows, int in_dim, const float *filters, int ldf,
                        const float *bias, int num_filters, int patch_dim, int patch_step, int patch_stride) {
  const int num_splice = in_dim / patch_stride, num_patches = 1 + (patch_stride - patch_dim) / patch_step, filter_dim = num_splice * patch_dim;
  const int ldp = filter_dim * num_patches;
  int32_t *map = conv_column_map(num_patches, num_splice, patch_dim, patch_step, patch_stride);
  orc_copy_cols_idx(patches, ldp, in, ldi, rows, map, ldp);   /* :329 */
  free(map);
  for (int p = 0; p < num_patches; p++) {   /* :332-339 */
    float *tgt = out + (size_t)p * num_filters;
    for (int r = 0; r < rows; r++)
      for (int f = 0; f < num_filters; f++) tgt[(size_t)r * ldo + f] = bias[f];   /* AddVecToRows(1.0, bias, 0.0) */
    orc_add_mat_mat(tgt, rows, num_filters, ldo, 1.0f, patches + (size_t)p * filter_dim, ldp, 0, filters, ldf, 1, filter_dim, 1.0f);
  }
}
/* patch_diffs: [rows x filter_dim * num_patches] (feature_patch_diffs_); in_diff [rows x in_dim] is zeroed first, as Component::Backpropagate does */
void orc_conv_backpropagate(float *in_diff, int ldid, float *patch_diffs, const float *out_diff, int ldod, int rows, int in_dim, const float *filters,
                            int ldf, int num_filters, int patch_dim, int patch_step, int patch_stride) {
  const int num_splice = in_dim / patch_stride, num_patches = 1 + (patch_stride - patch_dim) / patch_step, filter_dim = num_splice * patch_dim;
  const int ldp = filter_dim * num_patches;
  for (int p = 0; p < num_patches; p++)   /* :399-407 */
    orc_add_mat_mat(patch_diffs + (size_t)p * filter_dim, rows, filter_dim, ldp, 1.0f, out_diff + (size_t)p * num_filters, ldod, 0, filters, ldf, 0,
                    num_filters, 0.0f);
  for (int r = 0; r < rows; r++) memset(in_diff + (size_t)r * ldid, 0, sizeof(float) * in_dim);
  /* ReverseIndexes + RearrangeIndexes (:346-388): pass k adds, for every input column i, the k-th (ascending) j with map[j] == i */
  int32_t *map = conv_column_map(num_patches, num_splice, patch_dim, patch_step, patch_stride);
  int *count = (int *)calloc(in_dim, sizeof(int));
  int L = 0;
  for (int j = 0; j < ldp; j++) { count[map[j]]++; if (count[map[j]] > L) L = count[map[j]]; }
  int32_t *pass = (int32_t *)malloc(sizeof(int32_t) * in_dim);
  for (int k = 0; k < L; k++) {
    for (int i = 0; i < in_dim; i++) pass[i] = -1;
    memset(count, 0, sizeof(int) * in_dim);
    for (int j = 0; j < ldp; j++) {
      if (count[map[j]] == k) pass[map[j]] = j;
      count[map[j]]++;
    }
    orc_add_cols_idx(in_diff, ldid, patch_diffs, ldp, rows, pass, in_dim);   /* :419-420 */
  }
  free(pass); free(count); free(map);
}
/* :425-470; filters_grad [num_filters x filter_dim] (ld = filter_dim), bias_grad [num_filters] are reset and returned */
void orc_conv_update(float *filters, int ldf, float *bias, float *filters_grad, float *bias_grad, const float *patches, const float *diff, int ldd,
                     int rows, int in_dim, int num_filters, int patch_dim, int patch_step, int patch_stride, float learn_rate, float learn_rate_coef,
                     float bias_learn_rate_coef, float max_norm) {
  const int num_splice = in_dim / patch_stride, num_patches = 1 + (patch_stride - patch_dim) / patch_step, filter_dim = num_splice * patch_dim;
  const int ldp = filter_dim * num_patches;
  memset(filters_grad, 0, sizeof(float) * (size_t)num_filters * filter_dim);
  memset(bias_grad, 0, sizeof(float) * num_filters);
  for (int p = 0; p < num_patches; p++) {   /* :441-450 */
    const float *diff_patch = diff + (size_t)p * num_filters;
    orc_add_mat_mat(filters_grad, num_filters, filter_dim, filter_dim, 1.0f, diff_patch, ldd, 1, patches + (size_t)p * filter_dim, ldp, 0, rows, 1.0f);
    /* bias_grad_.AddRowSumMat(1.0, diff_patch, 1.0): column sums, kaldi-vector.cc AddRowSumMat (double accumulator per column for <= 64 rows,
     * sgemv with ones above; summed here in double) */
    for (int f = 0; f < num_filters; f++) {
      double sum = 0.0;
      for (int r = 0; r < rows; r++) sum += diff_patch[(size_t)r * ldd + f];
      bias_grad[f] = (float)(1.0 * sum + 1.0 * bias_grad[f]);
    }
  }
  for (int r = 0; r < num_filters; r++)   /* :456 */
    for (int c = 0; c < filter_dim; c++) filters[(size_t)r * ldf + c] += -learn_rate * learn_rate_coef * filters_grad[(size_t)r * filter_dim + c];
  for (int f = 0; f < num_filters; f++) bias[f] += -learn_rate * bias_learn_rate_coef * bias_grad[f];   /* :457 */
  if (max_norm > 0.0f) {   /* :460-470 */
    for (int r = 0; r < num_filters; r++) {
      float l2 = 0.0f;   /* lin_sqr = W .* W (float), l2.AddColSumMat(1.0, lin_sqr, 0.0), ApplyPow(0.5) */
      double acc = 0.0;
      for (int c = 0; c < filter_dim; c++) { const float w = filters[(size_t)r * ldf + c]; acc += (double)(w * w); }
      l2 = sqrtf((float)acc);
      float scl = l2 * (1.0f / max_norm);   /* scl.Scale(1.0 / max_norm) */
      if (scl < 1.0f) scl = 1.0f;           /* ApplyFloor(1.0) */
      scl = 1.0f / scl;                     /* InvertElements */
      for (int c = 0; c < filter_dim; c++) filters[(size_t)r * ldf + c] *= scl;   /* MulRowsVec */
    }
  }
}

/* ---- MaxPoolingComponent -------------------------------------------------------------------------------------------- */
void orc_max_pool_propagate(float *out, int ldo, const float *in, int ldi, int rows, int in_dim, int pool_size, int pool_step, int pool_stride) {
  const int num_patches = in_dim / pool_stride, num_pools = 1 + (num_patches - pool_size) / pool_step;
  for (int q = 0; q < num_pools; q++) {   /* :107-115 */
    float *pool = out + (size_t)q * pool_stride;
    for (int r = 0; r < rows; r++)
      for (int k = 0; k < pool_stride; k++) pool[(size_t)r * ldo + k] = -1e20f;
    for (int m = 0; m < pool_size; m++) {
      const float *src = in + (size_t)(m + q * pool_step) * pool_stride;
      for (int r = 0; r < rows; r++)
        for (int k = 0; k < pool_stride; k++) {   /* MatrixBase::Max: element-wise maximum */
          const float a = src[(size_t)r * ldi + k];
          if (a > pool[(size_t)r * ldo + k]) pool[(size_t)r * ldo + k] = a;
        }
    }
  }
}
void orc_max_pool_backpropagate(float *in_diff, int ldid, const float *in, int ldi, const float *out, int ldo, const float *out_diff, int ldod, int rows,
                                int in_dim, int pool_size, int pool_step, int pool_stride) {
  const int num_patches = in_dim / pool_stride, num_pools = 1 + (num_patches - pool_size) / pool_step;
  int *summands = (int *)calloc(num_patches, sizeof(int));
  for (int r = 0; r < rows; r++) memset(in_diff + (size_t)r * ldid, 0, sizeof(float) * in_dim);   /* :131 */
  for (int q = 0; q < num_pools; q++)   /* :133-153 */
    for (int m = 0; m < pool_size; m++) {
      const int p = m + q * pool_step;
      for (int r = 0; r < rows; r++)
        for (int k = 0; k < pool_stride; k++) {
          const float mask = in[(size_t)r * ldi + (size_t)p * pool_stride + k] == out[(size_t)r * ldo + (size_t)q * pool_stride + k] ? 1.0f : 0.0f;
          const float src = out_diff[(size_t)r * ldod + (size_t)q * pool_stride + k] * mask;   /* src.MulElements(mask) */
          in_diff[(size_t)r * ldid + (size_t)p * pool_stride + k] += 1.0f * src;               /* tgt.AddMat(1.0, src) */
        }
      summands[p] += 1;
    }
  for (int p = 0; p < num_patches; p++) {   /* :156-160 */
    const float scale = (float)(1.0 / summands[p]);
    for (int r = 0; r < rows; r++)
      for (int k = 0; k < pool_stride; k++) in_diff[(size_t)r * ldid + (size_t)p * pool_stride + k] *= scale;
  }
  free(summands);
}

/* ---- LengthNormComponent -------------------------------------------------------------------------------------------- */
void orc_length_norm_propagate(float *out, int ldo, float *row_scales, const float *in, int ldi, int rows, int cols) {
  for (int r = 0; r < rows; r++) {   /* :344-351 */
    double sum = 0.0;   /* AddColSumMat: double accumulator for <= 64 columns (kaldi-vector.cc:741-748), sgemv above; double here */
    for (int c = 0; c < cols; c++) { const float v = in[(size_t)r * ldi + c]; sum += (double)(v * v); }
    const float norm = sqrtf((float)sum);   /* ApplyPow(0.5) */
    row_scales[r] = (float)(1 / norm);      /* InvertElements: static_cast<Real>(1 / x) */
    for (int c = 0; c < cols; c++) out[(size_t)r * ldo + c] = in[(size_t)r * ldi + c] * row_scales[r];
  }
}
void orc_length_norm_backpropagate(float *in_diff, int ldid, const float *out_diff, int ldod, const float *row_scales, int rows, int cols) {
  for (int r = 0; r < rows; r++)   /* :356-357 */
    for (int c = 0; c < cols; c++) in_diff[(size_t)r * ldid + c] = out_diff[(size_t)r * ldod + c] * row_scales[r];
}

/* ---- group p-norm / group max ------------------------------------------------------------------------------------------ */
static float vec_norm(const float *x, int n, float p) {   /* kaldi-vector.cc:520-557 */
  float sum = 0.0f;
  if (p == 0.0f) {
    for (int i = 0; i < n; i++) if (x[i] != 0.0f) sum += 1.0f;
    return sum;
  } else if (p == 1.0f) {
    for (int i = 0; i < n; i++) sum += fabsf(x[i]);
    return sum;
  } else if (p == 2.0f) {
    for (int i = 0; i < n; i++) sum += x[i] * x[i];
    return sqrtf(sum);
  }
  int ok = 1;
  for (int i = 0; i < n; i++) {
    const float t = powf(fabsf(x[i]), p);
    if (t == HUGE_VALF) ok = 0;
    sum += t;
  }
  if (ok) return powf(sum, 1.0f / p);
  float max_abs = 0.0f;
  for (int i = 0; i < n; i++) if (fabsf(x[i]) > max_abs) max_abs = fabsf(x[i]);
  sum = 0.0f;
  for (int i = 0; i < n; i++) sum += powf(fabsf(x[i] * (1.0f / max_abs)), p);
  return powf(sum, 1.0f / p) * max_abs;
}
void orc_group_pnorm(float *y, int ldy, const float *x, int ldx, int rows, int out_cols, int group, float power) {   /* kaldi-matrix.cc:2530-2538 */
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < out_cols; j++) y[(size_t)i * ldy + j] = vec_norm(x + (size_t)i * ldx + (size_t)j * group, group, power);
}
void orc_group_pnorm_deriv(float *d, int ldd, const float *in, int ldi, const float *out, int ldo, int rows, int in_cols, int group, float power) {
  for (int i = 0; i < rows; i++)   /* kaldi-matrix.cc:1088-1118 */
    for (int j = 0; j < in_cols; j++) {
      const float iv = in[(size_t)i * ldi + j], ov = out[(size_t)i * ldo + j / group];
      float v;
      if (power == 1.0f) v = iv == 0 ? 0 : (iv > 0 ? 1 : -1);
      else if (ov == 0) v = 0;
      else v = powf(fabsf(iv), power - 1) * powf(ov, 1 - power) * (iv >= 0 ? 1 : -1);
      d[(size_t)i * ldd + j] = v;
    }
}
void orc_group_max(float *y, int ldy, const float *x, int ldx, int rows, int out_cols, int group) {   /* kaldi-matrix.cc:2541-2558 */
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < out_cols; j++) {
      float m = -1e20f;
      for (int k = 0; k < group; k++) if (x[(size_t)i * ldx + (size_t)j * group + k] > m) m = x[(size_t)i * ldx + (size_t)j * group + k];
      y[(size_t)i * ldy + j] = m;
    }
}
void orc_group_max_deriv(float *d, int ldd, const float *in, int ldi, const float *out, int ldo, int rows, int in_cols, int group) {
  for (int i = 0; i < rows; i++)   /* kaldi-matrix.cc:1121-1138 */
    for (int j = 0; j < in_cols; j++) d[(size_t)i * ldd + j] = in[(size_t)i * ldi + j] == out[(size_t)i * ldo + j / group] ? 1.0f : 0.0f;
}
void orc_mul_rows_group_mat(float *y, int ldy, const float *src, int lds, int rows, int cols, int group) {   /* kaldi-matrix.cc:1071-1085 */
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) y[(size_t)i * ldy + j] *= src[(size_t)i * lds + j / group];
}
