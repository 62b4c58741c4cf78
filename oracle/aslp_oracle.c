/*
 * aslp_oracle.c -- TEST INFRASTRUCTURE ONLY (see aslp_oracle.h).
 *
 * Plain-C restatement of the reference CPU path (HAVE_CUDA undefined: every
 * CuMatrix method falls through to src/matrix + CBLAS).  Expression forms (literal
 * types, evaluation order) are kept as in the reference so that float/double
 * promotion rounds the same way.  Paths cited are relative to /root/reference/src.
 */
#include "aslp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
void orc_set_num_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_num_threads(void) { return g_threads; }

/* ------------------------------------------------------------------------------- */
/* AddMatMat: cu-matrix.cc:1027-1061 -> MatrixBase::AddMatMat -> cblas_sgemm.
 * The reference hands this to an optimised BLAS; so that the CPU baseline timed from this file is
 * a fair one, the product is computed by a small cache-blocked AVX2/FMA kernel of the usual
 * BLAS shape (packed panels, 6 x 16 register tile, OpenMP over C tiles) instead of a naive loop.
 * Numerics: fp32 accumulation with k ascending inside KC-blocks of 256 -- the same class of
 * rounding as any BLAS sgemm (the parity tolerance of 1e-4 relative covers the order). */
#include <immintrin.h>
#define GEMM_MR 6
#define GEMM_NR 16
#define GEMM_KC 256
#define GEMM_MC 48  /* multiple of MR */
#define GEMM_NC 256 /* multiple of NR */

/* acc[6][16] += Ap[kc][6] (x) Bp[kc][16] */
static inline void gemm_micro(int kc, const float *Ap, const float *Bp, float *acc, int ldacc) {
  __m256 c00 = _mm256_loadu_ps(acc + 0 * ldacc), c01 = _mm256_loadu_ps(acc + 0 * ldacc + 8);
  __m256 c10 = _mm256_loadu_ps(acc + 1 * ldacc), c11 = _mm256_loadu_ps(acc + 1 * ldacc + 8);
  __m256 c20 = _mm256_loadu_ps(acc + 2 * ldacc), c21 = _mm256_loadu_ps(acc + 2 * ldacc + 8);
  __m256 c30 = _mm256_loadu_ps(acc + 3 * ldacc), c31 = _mm256_loadu_ps(acc + 3 * ldacc + 8);
  __m256 c40 = _mm256_loadu_ps(acc + 4 * ldacc), c41 = _mm256_loadu_ps(acc + 4 * ldacc + 8);
  __m256 c50 = _mm256_loadu_ps(acc + 5 * ldacc), c51 = _mm256_loadu_ps(acc + 5 * ldacc + 8);
  for (int k = 0; k < kc; k++) {
    const __m256 b0 = _mm256_loadu_ps(Bp + (size_t)k * GEMM_NR), b1 = _mm256_loadu_ps(Bp + (size_t)k * GEMM_NR + 8);
    const float *a = Ap + (size_t)k * GEMM_MR;
    __m256 av;
    av = _mm256_broadcast_ss(a + 0); c00 = _mm256_fmadd_ps(av, b0, c00); c01 = _mm256_fmadd_ps(av, b1, c01);
    av = _mm256_broadcast_ss(a + 1); c10 = _mm256_fmadd_ps(av, b0, c10); c11 = _mm256_fmadd_ps(av, b1, c11);
    av = _mm256_broadcast_ss(a + 2); c20 = _mm256_fmadd_ps(av, b0, c20); c21 = _mm256_fmadd_ps(av, b1, c21);
    av = _mm256_broadcast_ss(a + 3); c30 = _mm256_fmadd_ps(av, b0, c30); c31 = _mm256_fmadd_ps(av, b1, c31);
    av = _mm256_broadcast_ss(a + 4); c40 = _mm256_fmadd_ps(av, b0, c40); c41 = _mm256_fmadd_ps(av, b1, c41);
    av = _mm256_broadcast_ss(a + 5); c50 = _mm256_fmadd_ps(av, b0, c50); c51 = _mm256_fmadd_ps(av, b1, c51);
  }
  _mm256_storeu_ps(acc + 0 * ldacc, c00); _mm256_storeu_ps(acc + 0 * ldacc + 8, c01);
  _mm256_storeu_ps(acc + 1 * ldacc, c10); _mm256_storeu_ps(acc + 1 * ldacc + 8, c11);
  _mm256_storeu_ps(acc + 2 * ldacc, c20); _mm256_storeu_ps(acc + 2 * ldacc + 8, c21);
  _mm256_storeu_ps(acc + 3 * ldacc, c30); _mm256_storeu_ps(acc + 3 * ldacc + 8, c31);
  _mm256_storeu_ps(acc + 4 * ldacc, c40); _mm256_storeu_ps(acc + 4 * ldacc + 8, c41);
  _mm256_storeu_ps(acc + 5 * ldacc, c50); _mm256_storeu_ps(acc + 5 * ldacc + 8, c51);
}

void orc_add_mat_mat(float *C, int M, int N, int ldc, float alpha, const float *A, int lda,
                     int transA, const float *B, int ldb, int transB, int K, float beta) {
  if (M <= 0 || N <= 0) return;
  const int npan = (N + GEMM_NR - 1) / GEMM_NR;
  const int Kp = K > 0 ? K : 1;
  /* B packed once as column panels: Bp[panel][k][16], zero padded */
  float *Bp = (float *)aligned_alloc(64, sizeof(float) * (size_t)npan * Kp * GEMM_NR);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int p = 0; p < npan; p++) {
    float *dst = Bp + (size_t)p * Kp * GEMM_NR;
    for (int k = 0; k < K; k++)
      for (int j = 0; j < GEMM_NR; j++) {
        const int n = p * GEMM_NR + j;
        float v = 0.0f;
        if (n < N) v = transB ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n];
        dst[(size_t)k * GEMM_NR + j] = v;
      }
  }
  const int mblocks = (M + GEMM_MC - 1) / GEMM_MC, nblocks = (N + GEMM_NC - 1) / GEMM_NC;
#pragma omp parallel num_threads(g_threads)
  {
    float *Ap = (float *)aligned_alloc(64, sizeof(float) * GEMM_MC * GEMM_KC);
    float *acc = (float *)aligned_alloc(64, sizeof(float) * GEMM_MC * GEMM_NC);
#pragma omp for schedule(dynamic) collapse(2)
    for (int ib = 0; ib < mblocks; ib++)
      for (int jb = 0; jb < nblocks; jb++) {
        const int i0 = ib * GEMM_MC, mc = M - i0 < GEMM_MC ? M - i0 : GEMM_MC;
        const int j0 = jb * GEMM_NC, nc = N - j0 < GEMM_NC ? N - j0 : GEMM_NC;
        const int mtiles = (mc + GEMM_MR - 1) / GEMM_MR, ntiles = (nc + GEMM_NR - 1) / GEMM_NR;
        memset(acc, 0, sizeof(float) * GEMM_MC * GEMM_NC);
        for (int k0 = 0; k0 < K; k0 += GEMM_KC) {
          const int kc = K - k0 < GEMM_KC ? K - k0 : GEMM_KC;
          /* pack the A block as row tiles: Ap[tile][k][6] */
          for (int t = 0; t < mtiles; t++)
            for (int k = 0; k < kc; k++)
              for (int r = 0; r < GEMM_MR; r++) {
                const int i = i0 + t * GEMM_MR + r;
                float v = 0.0f;
                if (i < M) v = transA ? A[(size_t)(k0 + k) * lda + i] : A[(size_t)i * lda + k0 + k];
                Ap[((size_t)t * kc + k) * GEMM_MR + r] = v;
              }
          for (int u = 0; u < ntiles; u++) {
            const float *bp = Bp + ((size_t)(j0 / GEMM_NR + u) * Kp + k0) * GEMM_NR;
            for (int t = 0; t < mtiles; t++)
              gemm_micro(kc, Ap + (size_t)t * kc * GEMM_MR, bp, acc + (size_t)t * GEMM_MR * GEMM_NC + u * GEMM_NR, GEMM_NC);
          }
        }
        for (int r = 0; r < mc; r++) {
          float *c = C + (size_t)(i0 + r) * ldc + j0;
          const float *a = acc + (size_t)r * GEMM_NC;
          if (beta == 0.0f) {
            for (int j = 0; j < nc; j++) c[j] = alpha * a[j];
          } else {
            for (int j = 0; j < nc; j++) c[j] = alpha * a[j] + beta * c[j];
          }
        }
      }
    free(Ap);
    free(acc);
  }
  free(Bp);
}

/* ------------------------------------------------------------------------------- */
void orc_sigmoid(float *y, int ldy, const float *x, int ldx, int rows, int cols) {
  /* matrix/kaldi-vector.cc:923-936 (overflow-safe two-branch form) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      float v = x[(size_t)r * ldx + c];
      if (v > 0.0) {
        v = 1.0 / (1.0 + expf(-v));
      } else {
        float ex = expf(v);
        v = ex / (ex + 1.0);
      }
      y[(size_t)r * ldy + c] = v;
    }
}

void orc_tanh(float *y, int ldy, const float *x, int ldx, int rows, int cols) {
  /* matrix/kaldi-vector.cc:885-898 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      float v = x[(size_t)r * ldx + c];
      if (v > 0.0) {
        float inv_expx = expf(-v);
        v = -1.0 + 2.0 / (1.0 + inv_expx * inv_expx);
      } else {
        float inv_expx = expf(v);
        v = 1.0 - 2.0 / (1.0 + inv_expx * inv_expx);
      }
      y[(size_t)r * ldy + c] = v;
    }
}

void orc_diff_sigmoid(float *eout, int ldo, const float *y, int ldy, const float *e, int lde,
                      int rows, int cols) {
  /* matrix/kaldi-matrix.cc:2713-2727 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      float v = y[(size_t)r * ldy + c], d = e[(size_t)r * lde + c];
      eout[(size_t)r * ldo + c] = d * v * (1.0 - v);
    }
}

void orc_diff_tanh(float *eout, int ldo, const float *y, int ldy, const float *e, int lde,
                   int rows, int cols) {
  /* matrix/kaldi-matrix.cc:2730-2744 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      float v = y[(size_t)r * ldy + c], d = e[(size_t)r * lde + c];
      eout[(size_t)r * ldo + c] = d * (1.0 - (v * v));
    }
}

void orc_softmax_rows(float *y, int ldy, const float *x, int ldx, int rows, int cols) {
  /* cu-matrix.cc:1351-1371 CPU branch: copy then per-row VectorBase::ApplySoftMax
   * (matrix/kaldi-vector.cc:852-859): max, exp(x-max) summed in float, scale 1/sum. */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++) {
    const float *xr = x + (size_t)r * ldx;
    float *yr = y + (size_t)r * ldy;
    float max = xr[0], sum = 0.0;
    for (int c = 1; c < cols; c++)
      if (xr[c] > max) max = xr[c];
    for (int c = 0; c < cols; c++) sum += (yr[c] = expf(xr[c] - max));
    float s = 1.0 / sum;
    for (int c = 0; c < cols; c++) yr[c] *= s;
  }
}

void orc_find_row_max_id(const float *m, int ld, int rows, int cols, int32_t *id) {
  /* cu-matrix.cc:1493-1510: first strict maximum, start value -1e21, id -1 if none */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++) {
    float max = -1e21;
    int32_t max_id = -1;
    const float *row = m + (size_t)r * ld;
    for (int c = 0; c < cols; c++)
      if (max < row[c]) {
        max = row[c];
        max_id = c;
      }
    id[r] = max_id;
  }
}

void orc_splice(float *y, int ldy, const float *x, int ldx, int rows, int in_cols,
                const int32_t *offsets, int n_off) {
  /* cu-math.cc:153-166 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int off = 0; off < n_off; off++) {
      int r_off = r + offsets[off];
      if (r_off < 0) r_off = 0;
      if (r_off >= rows) r_off = rows - 1;
      memcpy(y + (size_t)r * ldy + (size_t)off * in_cols, x + (size_t)r_off * ldx,
             sizeof(float) * in_cols);
    }
}

void orc_copy_cols(float *y, int ldy, const float *x, int ldx, int rows, const int32_t *copy_from,
                   int out_cols) {
  /* cu-math.cc:195-208 (cu::Copy) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < out_cols; c++) y[(size_t)r * ldy + c] = x[(size_t)r * ldx + copy_from[c]];
}

void orc_randomize(float *y, int ldy, const float *x, int ldx, int cols, const int32_t *copy_from,
                   int n_idx) {
  /* cu-math.cc:118-125 */
  for (int i = 0; i < n_idx; i++)
    memcpy(y + (size_t)i * ldy, x + (size_t)copy_from[i] * ldx, sizeof(float) * cols);
}

void orc_add_row_sum_mat(float *dst, int ldd, int dst_rows, int cols, const float *src, int lds,
                         int src_rows, float alpha, float beta) {
  /* cu-matrix.cc:3024-3033: dst.Row(k).AddRowSumMat(alpha, src.RowRange(k*P, P), beta) with
   * VectorBase::AddRowSumMat (matrix/kaldi-vector.cc:716-733): for P <= 64 it is
   * scal(beta) followed by P axpy's (fp32, row order); above that a gemv with ones. */
  int P = src_rows / dst_rows;
  for (int k = 0; k < dst_rows; k++)
    for (int c = 0; c < cols; c++) {
      float *d = dst + (size_t)k * ldd + c;
      if (P <= 64) {
        float v = beta * *d;
        for (int p = 0; p < P; p++) v += alpha * src[(size_t)(k * P + p) * lds + c];
        *d = v;
      } else {
        float sum = 0.0f;
        for (int p = 0; p < P; p++) sum += src[(size_t)(k * P + p) * lds + c];
        *d = alpha * sum + beta * *d;
      }
    }
}

void orc_add_conv_mat_mat_elements(float *dst, int ldd, int cols, const float *A, int lda,
                                   int a_rows, const float *B, int ldb, int b_rows, float alpha,
                                   float beta) {
  /* cu-matrix.cc:3062-3071 + kaldi-matrix.cc:483-500 */
  for (int k = 0; k < a_rows - b_rows + 1; k++)
    for (int c = 0; c < b_rows; c++)
      for (int j = 0; j < cols; j++) {
        float *d = dst + (size_t)(k * b_rows + c) * ldd + j;
        *d = beta * *d + alpha * A[(size_t)(k + c) * lda + j] * B[(size_t)c * ldb + j];
      }
}

void orc_regularize_l1(float *w, int ldw, float *g, int ldg, int rows, int cols, float l1,
                       float lr) {
  /* cu-math.cc:54-73 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      float *wp = w + (size_t)r * ldw + c, *gp = g + (size_t)r * ldg + c;
      if (*wp == 0.0) continue;
      float l1_signed = l1;
      if (*wp < 0.0) l1_signed = -l1;
      float before = *wp;
      float after = *wp - lr * *gp - l1_signed;
      if ((after > 0.0) ^ (before > 0.0)) {
        *wp = 0.0;
        *gp = 0.0;
      } else {
        *wp -= l1_signed;
      }
    }
}

/* ------------------------------------------------------------------------------- */
/* element-wise / broadcast arithmetic (reference CPU branches, plain loops) */
void orc_add_mat_mat_elements(float *dst, int ldd, const float *A, int lda, const float *B, int ldb, int rows, int cols,
                              float alpha, float beta) {
  /* kaldi-matrix.cc:483-501: data[j] = beta*data[j] + alpha*dataA[j]*dataB[j] */
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++)
      dst[(size_t)r * ldd + c] = beta * dst[(size_t)r * ldd + c] + alpha * A[(size_t)r * lda + c] * B[(size_t)r * ldb + c];
}
void orc_add_mat_diag_vec(float *dst, int ldd, const float *M, int m_row_stride, int m_col_stride, const float *v, int rows,
                          int cols, float alpha) {
  /* kaldi-matrix.cc:448-480 with beta == 1: data[i][j] += alpha * v[j] * M[i][j] (strides swapped for kTrans) */
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) dst[(size_t)i * ldd + j] += alpha * v[j] * M[(size_t)i * m_row_stride + (size_t)j * m_col_stride];
}
void orc_add_vec_to_rows(float *dst, int ldd, const float *row, int rows, int cols, float alpha) {
  /* kaldi-matrix.cc:2754-2759 (the <= 64 column branch; wider matrices take the same sums through BLAS ger) */
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) dst[(size_t)i * ldd + j] += alpha * row[j];
}
void orc_add_vec_to_cols(float *dst, int ldd, const float *col, int rows, int cols, float alpha) {
  /* kaldi-matrix.cc:2786-2792: to_add = alpha * v[i] first, then added */
  for (int i = 0; i < rows; i++) {
    const float to_add = alpha * col[i];
    for (int j = 0; j < cols; j++) dst[(size_t)i * ldd + j] += to_add;
  }
}
void orc_mul_cols_vec(float *dst, int ldd, const float *scale, int rows, int cols) {
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) dst[(size_t)i * ldd + j] *= scale[j];
}
void orc_mul_rows_vec(float *dst, int ldd, const float *scale, int rows, int cols) {
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) dst[(size_t)i * ldd + j] *= scale[i];
}
void orc_copy_cols_idx(float *dst, int ldd, const float *src, int lds, int rows, const int32_t *idx, int n_idx) {
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < n_idx; c++) dst[(size_t)r * ldd + c] = idx[c] < 0 ? 0.0f : src[(size_t)r * lds + idx[c]];
}
void orc_add_cols_idx(float *dst, int ldd, const float *src, int lds, int rows, const int32_t *idx, int n_idx) {
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < n_idx; c++)
      if (idx[c] >= 0) dst[(size_t)r * ldd + c] += src[(size_t)r * lds + idx[c]];
}

/* ------------------------------------------------------------------------------- */
/* AffineTransform */
void orc_affine_propagate(float *out, int ldo, const float *in, int ldi, int rows, const float *W,
                          int ldw, const float *bias, int in_dim, int out_dim) {
  /* nnet-affine-transform.h:186-191: out = 1*bias (beta 0); out += in * W^T */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < out_dim; c++) out[(size_t)r * ldo + c] = bias[c];
  orc_add_mat_mat(out, rows, out_dim, ldo, 1.0f, in, ldi, 0, W, ldw, 1, in_dim, 1.0f);
}

void orc_affine_backpropagate(float *in_diff, int ldid, const float *out_diff, int ldod, int rows,
                              const float *W, int ldw, int in_dim, int out_dim) {
  /* nnet-affine-transform.h:193-197: in_diff = out_diff * W */
  orc_add_mat_mat(in_diff, rows, in_dim, ldid, 1.0f, out_diff, ldod, 0, W, ldw, 0, out_dim, 0.0f);
}

void orc_affine_update(float *W, int ldw, float *bias, float *W_corr, int ldc, float *bias_corr,
                       const float *input, int ldi, const float *diff, int ldd, int rows,
                       int in_dim, int out_dim, const orc_affine_opts *o) {
  /* nnet-affine-transform.h:200-245 */
  const float lr = o->learn_rate * o->learn_rate_coef;
  const float lr_bias = o->learn_rate * o->bias_learn_rate_coef;
  const float mmt = o->momentum, l2 = o->l2_penalty, l1 = o->l1_penalty;
  const int num_frames = rows;
  /* gradient incl. momentum: sums over frames, not means */
  orc_add_mat_mat(W_corr, out_dim, in_dim, ldc, 1.0f, diff, ldd, 1, input, ldi, 0, rows, mmt);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int c = 0; c < out_dim; c++) { /* bias_corr_.AddRowSumMat(1.0, diff, mmt) */
    float sum = 0.0f;
    for (int r = 0; r < rows; r++) sum += diff[(size_t)r * ldd + c];
    bias_corr[c] = 1.0f * sum + mmt * bias_corr[c];
  }
  if (l2 != 0.0) {
    float a = -lr * l2 * num_frames;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int r = 0; r < out_dim; r++)
      for (int c = 0; c < in_dim; c++) W[(size_t)r * ldw + c] += a * W[(size_t)r * ldw + c];
  }
  if (l1 != 0.0) orc_regularize_l1(W, ldw, W_corr, ldc, out_dim, in_dim, lr * l1 * num_frames, lr);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < out_dim; r++)
    for (int c = 0; c < in_dim; c++) W[(size_t)r * ldw + c] += -lr * W_corr[(size_t)r * ldc + c];
  for (int c = 0; c < out_dim; c++) bias[c] += -lr_bias * bias_corr[c];
  if (o->max_norm > 0.0) { /* :231-243 shrink rows to the max-norm sphere */
    for (int r = 0; r < out_dim; r++) {
      float s = 0.0f;
      for (int c = 0; c < in_dim; c++) s += W[(size_t)r * ldw + c] * W[(size_t)r * ldw + c];
      float nrm = powf(s, 0.5f);
      float scl = nrm * (1.0 / o->max_norm);
      if (scl < 1.0) scl = 1.0;
      scl = 1.0 / scl;
      for (int c = 0; c < in_dim; c++) W[(size_t)r * ldw + c] *= scl;
    }
  }
}

/* ------------------------------------------------------------------------------- */
void orc_relu(float *y, int ldy, const float *x, int ldx, int rows, int cols) {
  /* nnet-activation.h:286-290: copy, ApplyFloor(0) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      float v = x[(size_t)r * ldx + c];
      y[(size_t)r * ldy + c] = v < 0.0f ? 0.0f : v;
    }
}
void orc_diff_relu(float *in_diff, int ldo, const float *in, int ldi, const float *out_diff,
                   int lde, int rows, int cols) {
  /* nnet-activation.h:292-297: heaviside(in) * out_diff; heaviside(x) = x > 0 ? 1 : 0 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++)
      in_diff[(size_t)r * ldo + c] =
          (in[(size_t)r * ldi + c] > 0.0f ? 1.0f : 0.0f) * out_diff[(size_t)r * lde + c];
}

void orc_splice_backpropagate(float *in_diff, int ldid, const float *out_diff, int ldod, int rows,
                              int in_cols, const int32_t *offsets, int n_off) {
  /* nnet-various.h:143-175: in_diff[t] = sum_c out_diff[clamp(t + off_c)][c-th block]
   * (a gather with +offset -- not the adjoint of the forward -- kept as is). */
  for (int t = 0; t < rows; t++)
    for (int j = 0; j < in_cols; j++) in_diff[(size_t)t * ldid + j] = 0.0f;
  for (int c = 0; c < n_off; c++)
    for (int t = 0; t < rows; t++) {
      int o = t + offsets[c];
      if (o < 0) o = 0;
      if (o >= rows) o = rows - 1;
      for (int j = 0; j < in_cols; j++) {
        float v = out_diff[(size_t)o * ldod + (size_t)c * in_cols + j];
        if (c == 0)
          in_diff[(size_t)t * ldid + j] = v;
        else
          in_diff[(size_t)t * ldid + j] += v;
      }
    }
}

/* ------------------------------------------------------------------------------- */
/* BatchNormalization */
static void colsum_scaled(float *v, const float *m, int ld, int rows, int cols, float alpha,
                          float beta) {
  /* CuVector::AddRowSumMat(alpha, M, beta): v = alpha * sum_rows(M) + beta * v.
   * CPU: VectorBase::AddRowSumMat (kaldi-vector.cc) -> float accumulation */
  /* columns in blocks of 16 per thread: unit-stride reads, each column still summed r = 0..rows-1 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int c0 = 0; c0 < cols; c0 += 16) {
    float s[16] = {0};
    const int nc = cols - c0 < 16 ? cols - c0 : 16;
    for (int r = 0; r < rows; r++)
      for (int j = 0; j < nc; j++) s[j] += m[(size_t)r * ld + c0 + j];
    for (int j = 0; j < nc; j++)
      v[c0 + j] = beta == 0.0f ? alpha * s[j] : alpha * s[j] + beta * v[c0 + j]; /* gemv: beta 0 does not read v */
  }
}

void orc_bn_propagate(orc_bn_state *s, float *out, int ldo, const float *in, int ldi, int rows,
                      float *xs) {
  /* nnet-batch-normalization.h:177-220 */
  const int D = s->dim;
  const int B = rows;
  if (!s->acc_cleaned) { /* :178-181 */
    s->acc_cleaned = 1;
    for (int c = 0; c < D; c++) s->acc_means[c] = s->acc_vars[c] = 0.0;
    s->num_acc_frames = 0;
  }
  colsum_scaled(s->mean_vec, in, ldi, B, D, 1.0 / (B), 0.0f); /* mu */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < B; r++)
    for (int c = 0; c < D; c++) xs[(size_t)r * D + c] = in[(size_t)r * ldi + c] + -1.0f * s->mean_vec[c];
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < B; r++) /* out = xs .* xs */
    for (int c = 0; c < D; c++) {
      float x = xs[(size_t)r * D + c];
      out[(size_t)r * ldo + c] = 1.0f * x * x; /* beta = 0 on a zeroed buffer (nnet-component.h:311) */
    }
  colsum_scaled(s->var_vec, out, ldo, B, D, 1.0 / (B), 0.0f);
  for (int c = 0; c < D; c++) { /* :202-204  +1e-7, pow 0.5, invert */
    float v = s->var_vec[c] + 0.0000001f;
    v = powf(v, 0.5f);
    s->var_vec[c] = 1.0f / v;
  }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < B; r++)
    for (int c = 0; c < D; c++) {
      float x = xs[(size_t)r * D + c] * s->var_vec[c];
      xs[(size_t)r * D + c] = x;
      out[(size_t)r * ldo + c] = x * s->scale[c] + 1.0f * s->shift[c];
    }
  /* :216-220 running statistics in double; x*x is formed in float first */
  s->num_acc_frames += B;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int c = 0; c < D; c++) {
    double sm = 0.0, sv = 0.0;
    for (int r = 0; r < B; r++) {
      float x = in[(size_t)r * ldi + c];
      float xx = 0.0f + 1.0f * x * x;
      sm += (double)x;
      sv += (double)xx;
    }
    s->acc_means[c] += sm;
    s->acc_vars[c] += sv;
  }
}

void orc_bn_backpropagate(orc_bn_state *s, float *in_diff, int ldid, const float *in, int ldi,
                          const float *out_diff, int ldod, int rows, float momentum, float *xs) {
  /* nnet-batch-normalization.h:222-277 */
  const int D = s->dim, B = rows;
  float *bufE = (float *)malloc(sizeof(float) * (size_t)B * D);
  float *dvar = (float *)malloc(sizeof(float) * D);
  float *dmean = (float *)malloc(sizeof(float) * D);
  /* dGamma, dBeta (sums, with momentum) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < B; r++)
    for (int c = 0; c < D; c++) bufE[(size_t)r * D + c] = xs[(size_t)r * D + c] * out_diff[(size_t)r * ldod + c];
  colsum_scaled(s->dscale, bufE, D, B, D, 1.0f, momentum);
  colsum_scaled(s->dshift, out_diff, ldod, B, D, 1.0f, momentum);
  /* 1. XsharpO_ <- dy * gamma */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < B; r++)
    for (int c = 0; c < D; c++) xs[(size_t)r * D + c] = out_diff[(size_t)r * ldod + c] * s->scale[c];
  /* 2. delta-var */
  for (int c = 0; c < D; c++) {
    float v = powf(s->var_vec[c], 3.0f);
    dvar[c] = v * -0.5f;
  }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < B; r++)
    for (int c = 0; c < D; c++) {
      float e = in[(size_t)r * ldi + c] + -1.0f * s->mean_vec[c];
      e = e * xs[(size_t)r * D + c];
      bufE[(size_t)r * D + c] = e * dvar[c];
    }
  colsum_scaled(dvar, bufE, D, B, D, 1.0f, 0.0f);
  /* 3. delta-mean */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < B; r++)
    for (int c = 0; c < D; c++) bufE[(size_t)r * D + c] = xs[(size_t)r * D + c] * s->var_vec[c] * -1.0f;
  colsum_scaled(dmean, bufE, D, B, D, 1.0f, 0.0f);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < B; r++)
    for (int c = 0; c < D; c++) {
      float e = in[(size_t)r * ldi + c] + -1.0f * s->mean_vec[c];
      e = e * (float)(2.0 / B);
      bufE[(size_t)r * D + c] = e * dvar[c];
    }
  colsum_scaled(dmean, bufE, D, B, D, -1.0f, 1.0f);
  /* 4. in_diff */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < B; r++)
    for (int c = 0; c < D; c++) {
      float v = xs[(size_t)r * D + c] * s->var_vec[c];
      v += 1.0f * bufE[(size_t)r * D + c];
      in_diff[(size_t)r * ldid + c] = (float)(1.0 / B) * dmean[c] + 1.0f * v;
    }
  free(bufE);
  free(dvar);
  free(dmean);
}

void orc_bn_update(orc_bn_state *s, float lr) {
  for (int c = 0; c < s->dim; c++) {
    s->scale[c] += -lr * s->dscale[c];
    s->shift[c] += -lr * s->dshift[c];
  }
}

void orc_bn_global_stats_from_acc(orc_bn_state *s) {
  /* ReadData, :56-94 */
  for (int c = 0; c < s->dim; c++) {
    s->mean_vec[c] = 0.0f;
    s->var_vec[c] = 1.0f;
  }
  if (s->num_acc_frames <= 0.0) return;
  float var_floor = 1e-10;
  for (int d = 0; d < s->dim; d++) {
    float mean = s->acc_means[d] / s->num_acc_frames;
    float var = s->acc_vars[d] / s->num_acc_frames - mean * mean;
    if (var <= var_floor) var = var_floor;
    s->mean_vec[d] = mean;
    s->var_vec[d] = 1.0 / sqrt(var + 0.0000001f);
  }
}

void orc_bn_feedforward(orc_bn_state *s, float *out, int ldo, const float *in, int ldi, int rows) {
  /* :139-175 */
  const int D = s->dim, B = rows;
  if (s->num_acc_frames <= 0) {
    float *xs = (float *)malloc(sizeof(float) * (size_t)B * D);
    colsum_scaled(s->mean_vec, in, ldi, B, D, 1.0 / (B), 0.0f);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int r = 0; r < B; r++)
      for (int c = 0; c < D; c++) {
        float x = in[(size_t)r * ldi + c] + -1.0f * s->mean_vec[c];
        xs[(size_t)r * D + c] = x;
        out[(size_t)r * ldo + c] = x * x;
      }
    colsum_scaled(s->var_vec, out, ldo, B, D, 1.0 / (B), 0.0f);
    for (int c = 0; c < D; c++) s->var_vec[c] = 1.0f / powf(s->var_vec[c] + 0.0000001f, 0.5f);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int r = 0; r < B; r++)
      for (int c = 0; c < D; c++)
        out[(size_t)r * ldo + c] = xs[(size_t)r * D + c] * s->var_vec[c] * s->scale[c] + s->shift[c];
    free(xs);
  } else {
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int r = 0; r < B; r++)
      for (int c = 0; c < D; c++) {
        float x = in[(size_t)r * ldi + c] + -1.0f * s->mean_vec[c];
        x = x * s->var_vec[c];
        out[(size_t)r * ldo + c] = x * s->scale[c] + s->shift[c];
      }
  }
}

/* ------------------------------------------------------------------------------- */
void orc_xent_eval(const float *fw, const float *net_out, int ldn, const float *tgt, int ldt,
                   int rows, int cols, float *diff, int ldd, orc_xent_stats *st) {
  /* nnet-loss.cc:63-122.  CuMatrix::Sum() on the CPU is MatrixBase::Sum() (double
   * accumulation, kaldi-matrix.cc:1016). */
  float *w = (float *)malloc(sizeof(float) * rows);
  double num_frames = 0.0;
  for (int r = 0; r < rows; r++) {
    float ts = 0.0f; /* target_sum_.AddColSumMat(1.0, targets, 0.0) */
    for (int c = 0; c < cols; c++) ts += tgt[(size_t)r * ldt + c];
    w[r] = fw[r] * ts;
    num_frames += w[r];
  }
  int32_t *id_out = (int32_t *)malloc(sizeof(int32_t) * rows);
  int32_t *id_tgt = (int32_t *)malloc(sizeof(int32_t) * rows);
  orc_find_row_max_id(net_out, ldn, rows, cols, id_out);
  orc_find_row_max_id(tgt, ldt, rows, cols, id_tgt);
  double correct = 0.0, xent = 0.0, ent = 0.0, lik = 0.0;
  double *part = (double *)malloc(sizeof(double) * 3 * rows); /* per-row sums (rows in parallel), added in row order below */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int r = 0; r < rows; r++) {
    double rx = 0.0, re = 0.0, rl = 0.0;
    for (int c = 0; c < cols; c++) {
      float y = net_out[(size_t)r * ldn + c], t = tgt[(size_t)r * ldt + c];
      diff[(size_t)r * ldd + c] = (y + -1.0f * t) * w[r];
      float ly = logf(y + 1e-20f);
      rx += (double)(ly * t * w[r]);
      float lt = logf(t + 1e-20f);
      re += (double)(lt * t * w[r]);
      rl += (double)(y * t * w[r]);
    }
    part[3 * r] = rx; part[3 * r + 1] = re; part[3 * r + 2] = rl;
  }
  for (int r = 0; r < rows; r++) {
    correct += w[r] * (id_out[r] == id_tgt[r] ? 1.0 : 0.0);
    xent += part[3 * r]; ent += part[3 * r + 1]; lik += part[3 * r + 2];
  }
  free(part);
  st->frames = num_frames;
  st->correct = correct;
  st->loss = -xent;
  st->entropy = -ent;
  st->likelyhood = lik;
  free(w);
  free(id_out);
  free(id_tgt);
}

void orc_mse_eval(const float *fw, const float *net_out, int ldn, const float *tgt, int ldt,
                  int rows, int cols, float *diff, int ldd, double *loss, double *frames) {
  /* nnet-loss.cc:205-236 */
  double fsum = 0.0;
  for (int r = 0; r < rows; r++) fsum += fw[r];
  int num_frames = (int)fsum; /* int32 num_frames = frame_weights.Sum() */
  double sq = 0.0;
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      float d = (net_out[(size_t)r * ldn + c] + -1.0f * tgt[(size_t)r * ldt + c]) * fw[r];
      diff[(size_t)r * ldd + c] = d;
      sq += (double)(d * d * fw[r]);
    }
  *loss = 0.5 * sq;
  *frames = num_frames;
}

void orc_multitask_eval(int n_tasks, const int *kinds, const int *dims, const float *weights, const float *fw, const float *net_out,
                        int ldn, const float *tgt, int ldt, int rows, float *diff, int ldd, orc_xent_stats *xent_st, double *mse_loss,
                        double *mse_frames) {
  /* nnet-loss.cc:341-368: every task evaluates its own column block of the network output against the same block of the dense
   * target matrix (PosteriorToMatrix over ALL columns, :350), its diff is scaled by the task weight (:362) and copied into the
   * block (:364).  The column offsets are the running sums of the dims (:330-333). */
  int off = 0;
  for (int i = 0; i < n_tasks; i++) {
    float *d = diff + off;
    if (kinds[i] == 0) {
      orc_xent_eval(fw, net_out + off, ldn, tgt + off, ldt, rows, dims[i], d, ldd, &xent_st[i]);
    } else {
      orc_mse_eval(fw, net_out + off, ldn, tgt + off, ldt, rows, dims[i], d, ldd, &mse_loss[i], &mse_frames[i]);
    }
    for (int r = 0; r < rows; r++)
      for (int c = 0; c < dims[i]; c++) d[(size_t)r * ldd + c] *= weights[i]; /* diff_aux.Scale(loss_weights_[i]) */
    off += dims[i];
  }
}

/* ------------------------------------------------------------------------------- */
/* Whole DNN train step (cpu_baseline "port"): the chain Nnet::Propagate ->
 * Xent::Eval -> Nnet::Backpropagate of nnet-nnet.cc:70-154 for a "simple" net
 * (InputLayer/OutputLayer copies and the zero+AddMat links are exact copies and
 * are elided: they do not change values). */
struct orc_dnn {
  int in_dim, hid, nh, out_dim, bn, mb, L;
  float **W, **b, **Wc, **bc;
  orc_bn_state *bns;
  float **xs;
  float **aff_out, **bn_out, **act_out; /* per layer forward buffers */
  float *softmax_out, *tgt, *diff, *fw;
  float **d_act, **d_bn, **d_aff;
};

static float urand(unsigned *s) {
  *s = *s * 1664525u + 1013904223u;
  return (float)((*s >> 8) & 0xFFFFFF) / 16777216.0f;
}
static float grand(unsigned *s) {
  float u1 = urand(s) + 1e-7f, u2 = urand(s);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
}

orc_dnn *orc_dnn_create(int in_dim, int hid, int nh, int out_dim, int with_bn, int mb,
                        unsigned seed) {
  orc_dnn *d = (orc_dnn *)calloc(1, sizeof(orc_dnn));
  d->in_dim = in_dim; d->hid = hid; d->nh = nh; d->out_dim = out_dim; d->bn = with_bn; d->mb = mb;
  d->L = nh + 1;
  int L = d->L;
  d->W = calloc(L, sizeof(float *)); d->b = calloc(L, sizeof(float *));
  d->Wc = calloc(L, sizeof(float *)); d->bc = calloc(L, sizeof(float *));
  d->aff_out = calloc(L, sizeof(float *)); d->bn_out = calloc(L, sizeof(float *));
  d->act_out = calloc(L, sizeof(float *)); d->xs = calloc(L, sizeof(float *));
  d->d_act = calloc(L, sizeof(float *)); d->d_bn = calloc(L, sizeof(float *));
  d->d_aff = calloc(L, sizeof(float *));
  d->bns = calloc(L, sizeof(orc_bn_state));
  unsigned s = seed ? seed : 777u;
  for (int l = 0; l < L; l++) {
    int di = l == 0 ? in_dim : hid, dout = l == L - 1 ? out_dim : hid;
    d->W[l] = malloc(sizeof(float) * (size_t)di * dout);
    d->Wc[l] = calloc((size_t)di * dout, sizeof(float));
    d->b[l] = malloc(sizeof(float) * dout);
    d->bc[l] = calloc(dout, sizeof(float));
    /* nnet-affine-transform.h:99-113 Gaussian init, ParamStddev 0.04; bias mean/range
     * as in SURVEY Appendix A (hidden -2/4, output 0/0) */
    for (size_t i = 0; i < (size_t)di * dout; i++) d->W[l][i] = 0.04f * grand(&s);
    for (int i = 0; i < dout; i++)
      d->b[l][i] = l == L - 1 ? 0.0f : -2.0f + (urand(&s) - 0.5f) * 4.0f;
    d->aff_out[l] = malloc(sizeof(float) * (size_t)mb * dout);
    d->d_aff[l] = malloc(sizeof(float) * (size_t)mb * dout);
    d->d_act[l] = malloc(sizeof(float) * (size_t)mb * di); /* diff wrt this layer's input */
    if (l < L - 1) {
      d->act_out[l] = malloc(sizeof(float) * (size_t)mb * dout);
      if (with_bn) {
        d->bn_out[l] = malloc(sizeof(float) * (size_t)mb * dout);
        d->d_bn[l] = malloc(sizeof(float) * (size_t)mb * dout);
        d->xs[l] = malloc(sizeof(float) * (size_t)mb * dout);
        orc_bn_state *b = &d->bns[l];
        b->dim = dout;
        b->scale = malloc(sizeof(float) * dout); b->shift = calloc(dout, sizeof(float));
        for (int i = 0; i < dout; i++) b->scale[i] = 1.0f;
        b->dscale = calloc(dout, sizeof(float)); b->dshift = calloc(dout, sizeof(float));
        b->mean_vec = calloc(dout, sizeof(float)); b->var_vec = calloc(dout, sizeof(float));
        b->acc_means = calloc(dout, sizeof(double)); b->acc_vars = calloc(dout, sizeof(double));
      }
    }
  }
  d->softmax_out = malloc(sizeof(float) * (size_t)mb * out_dim);
  d->tgt = calloc((size_t)mb * out_dim, sizeof(float));
  d->diff = malloc(sizeof(float) * (size_t)mb * out_dim);
  d->fw = malloc(sizeof(float) * mb);
  for (int i = 0; i < mb; i++) d->fw[i] = 1.0f;
  return d;
}

void orc_dnn_destroy(orc_dnn *d) {
  if (!d) return;
  for (int l = 0; l < d->L; l++) {
    free(d->W[l]); free(d->b[l]); free(d->Wc[l]); free(d->bc[l]);
    free(d->aff_out[l]); free(d->bn_out[l]); free(d->act_out[l]); free(d->xs[l]);
    free(d->d_act[l]); free(d->d_bn[l]); free(d->d_aff[l]);
    orc_bn_state *b = &d->bns[l];
    free(b->scale); free(b->shift); free(b->dscale); free(b->dshift); free(b->mean_vec);
    free(b->var_vec); free(b->acc_means); free(b->acc_vars);
  }
  free(d->W); free(d->b); free(d->Wc); free(d->bc); free(d->aff_out); free(d->bn_out);
  free(d->act_out); free(d->xs); free(d->d_act); free(d->d_bn); free(d->d_aff); free(d->bns);
  free(d->softmax_out); free(d->tgt); free(d->diff); free(d->fw);
  free(d);
}

double orc_dnn_train_step(orc_dnn *d, const float *in, const int32_t *labels, float lr,
                          float mmt) {
  const int L = d->L, mb = d->mb;
  const float *x = in;
  int xdim = d->in_dim;
  /* forward */
  for (int l = 0; l < L; l++) {
    int dout = l == L - 1 ? d->out_dim : d->hid;
    orc_affine_propagate(d->aff_out[l], dout, x, xdim, mb, d->W[l], xdim, d->b[l], xdim, dout);
    if (l < L - 1) {
      const float *pre = d->aff_out[l];
      if (d->bn) {
        orc_bn_propagate(&d->bns[l], d->bn_out[l], dout, d->aff_out[l], dout, mb, d->xs[l]);
        pre = d->bn_out[l];
      }
      orc_sigmoid(d->act_out[l], dout, pre, dout, mb, dout);
      x = d->act_out[l];
      xdim = dout;
    }
  }
  orc_softmax_rows(d->softmax_out, d->out_dim, d->aff_out[L - 1], d->out_dim, mb, d->out_dim);
  /* loss: one-hot posterior -> dense targets (PosteriorToMatrix) */
  memset(d->tgt, 0, sizeof(float) * (size_t)mb * d->out_dim);
  for (int r = 0; r < mb; r++) d->tgt[(size_t)r * d->out_dim + labels[r]] = 1.0f;
  orc_xent_stats st;
  orc_xent_eval(d->fw, d->softmax_out, d->out_dim, d->tgt, d->out_dim, mb, d->out_dim, d->diff,
                d->out_dim, &st);
  /* backward: Softmax backward is a copy (nnet-activation.h:51-59) */
  const float *dy = d->diff;
  orc_affine_opts o = {lr, mmt, 0.0f, 0.0f, 1.0f, 1.0f, 0.0f};
  for (int l = L - 1; l >= 0; l--) {
    int di = l == 0 ? d->in_dim : d->hid, dout = l == L - 1 ? d->out_dim : d->hid;
    const float *lin = l == 0 ? in : d->act_out[l - 1];
    const float *daff = dy;
    if (l < L - 1) {
      const float *dpre = dy;
      /* Sigmoid backward */
      orc_diff_sigmoid(d->d_aff[l], dout, d->act_out[l], dout, dy, dout, mb, dout);
      dpre = d->d_aff[l];
      if (d->bn) {
        orc_bn_backpropagate(&d->bns[l], d->d_bn[l], dout, d->aff_out[l], dout, dpre, dout, mb,
                             mmt, d->xs[l]);
        orc_bn_update(&d->bns[l], lr);
        daff = d->d_bn[l];
      } else {
        daff = dpre;
      }
    }
    /* the reference back-propagates through every component, also the first
     * (nnet-nnet.cc:124-125), then updates it immediately */
    orc_affine_backpropagate(d->d_act[l], di, daff, dout, mb, d->W[l], di, di, dout);
    orc_affine_update(d->W[l], di, d->b[l], d->Wc[l], di, d->bc[l], lin, di, daff, dout, mb, di,
                      dout, &o);
    dy = d->d_act[l];
  }
  return st.loss;
}

int orc_dnn_num_layers(const orc_dnn *d) { return d->L; }
float *orc_dnn_weight(orc_dnn *d, int l, int *rows, int *cols) {
  if (rows) *rows = l == d->L - 1 ? d->out_dim : d->hid;
  if (cols) *cols = l == 0 ? d->in_dim : d->hid;
  return d->W[l];
}
float *orc_dnn_bias(orc_dnn *d, int l) { return d->b[l]; }
float *orc_dnn_bn_scale(orc_dnn *d, int l) { return d->bns[l].scale; }
float *orc_dnn_bn_shift(orc_dnn *d, int l) { return d->bns[l].shift; }
const float *orc_dnn_output(const orc_dnn *d) { return d->softmax_out; }

/* oracle/gen_cumatrix_blas_golden.cpp: Uniform() and Fill() */
void orc_golden_uniform_fill(unsigned long long *state, float *out, long n, float lo, float hi) {
  unsigned long long g = *state;
  for (long i = 0; i < n; i++) {
    g ^= g << 13; g ^= g >> 7; g ^= g << 17;
    out[i] = lo + (hi - lo) * (float)((double)(g >> 40) * (1.0 / 16777216.0));
  }
  *state = g;
}

