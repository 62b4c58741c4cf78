/*
 * aslp_kernels.h -- kernel-level C ABI of libaslp_hip.so (boundary B1 + B2 of SURVEY.md §8b).
 *
 * Every entry point keeps the name and argument order of the reference wrapper it
 * replaces in src/aslp-cudamatrix/cu-kernels-ansi.h (line cited per function), so a
 * maintainer can relink the reference's CuMatrix against this library for the hot-path
 * subset.  Differences, all deliberate:
 *   - `Gr`/`Bl` (CUDA launch geometry) are accepted and ignored: the library chooses its
 *     own CDNA4 geometry (64-wide wavefronts, 16-byte lanes).  Exception:
 *     cudaF_add_conv_mat_mat_elements reads Bl.y because the reference encodes the
 *     filter length there (cu-matrix.cc:3052-3056).
 *   - kernels are enqueued on the stream set with aslp_set_stream() (default: the null
 *     stream, like the reference); they are asynchronous.  Errors are sticky and read
 *     with aslp_get_last_error() (the reference wraps cudaGetLastError in CU_SAFE_CALL).
 *   - all pointers are DEVICE pointers unless the parameter is named *_host.
 * Matrices are row-major fp32, `MatrixDim{rows, cols, stride}`; "rows = frames".
 */
#ifndef ASLP_KERNELS_H_
#define ASLP_KERNELS_H_
#include "aslp_matrixdim.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime plumbing (replaces CuDevice, cu-device.h:43-151) --------------------- */
void aslp_set_stream(void *hip_stream);          /* per host thread */
void *aslp_get_stream(void);
int aslp_get_last_error(char *buf, int buflen);  /* 0 = no error; clears it */
int aslp_device_sync(void);
const char *aslp_version(void);

/* ---- elementwise (cu-kernels-ansi.h line in brackets) ------------------------------ */
void cudaF_set_const(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, float value, MatrixDim d);              /* [76]  */
void cudaF_add(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, float value, MatrixDim d);                    /* [78]  */
void cudaF_scale(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, float value, MatrixDim d);                  /* [81]  */
void cudaF_apply_log(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, MatrixDim d);                           /* [82]  */
void cudaF_apply_exp(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, MatrixDim d);                           /* [59]  */
void cudaF_apply_pow(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, float power, MatrixDim d);              /* [60]  */
void cudaF_apply_heaviside(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, MatrixDim d);                     /* [62]  */
void cudaF_apply_floor(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, float floor_val, MatrixDim d);        /* [63]  */
void cudaF_apply_ceiling(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, float ceiling_val, MatrixDim d);    /* [72]  */
void cudaF_invert_elements(aslp_dim3 Gr, aslp_dim3 Bl, float *data, MatrixDim d);                    /* [130] */
void cudaF_mul_elements(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, const float *A, MatrixDim dst_d, int src_stride);  /* [83] */
void cudaF_mul_cols_vec(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, const float *scale, MatrixDim d);    /* [86] mat[r][c] *= scale[c] */
void cudaF_mul_rows_vec(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, const float *scale, MatrixDim d);    /* [87] mat[r][c] *= scale[r] */
void cudaF_add_mat(aslp_dim3 Gr, aslp_dim3 Bl, float alpha, const float *src, float *dst, MatrixDim d, int src_stride, int A_trans); /* [92] dst += alpha*src */
void cudaF_add_vec_to_cols(aslp_dim3 Gr, aslp_dim3 Bl, float alpha, const float *col, float beta, float *dst, MatrixDim d);          /* [95] */
void cudaF_add_vec_to_rows(aslp_dim3 Gr, aslp_dim3 Bl, float alpha, const float *row, float beta, float *dst, MatrixDim d);          /* [96] */
void cudaF_add_mat_diag_vec(aslp_dim3 Gr, aslp_dim3 Bl, float alpha, float *mat, MatrixDim mat_dim, const float *mat2,
                            int mat2_row_stride, int mat2_col_stride, const float *vec, float beta);                                 /* [97] */
void cudaF_add_mat_mat_elements(aslp_dim3 Gr, aslp_dim3 Bl, float *data, const float *srcA_data, const float *srcB_data,
                                MatrixDim dim, int srcA_stride, int srcB_stride, float alpha, float beta);                           /* [98] */
void cudaF_add_row_sum_mat(aslp_dim3 Gr, aslp_dim3 Bl, float *data, const float *src_data, MatrixDim dim, int src_stride,
                           int patch_nrows, float alpha, float beta);                                                                /* [99]  ASLP */
void cudaF_add_conv_mat_mat_elements(aslp_dim3 Gr, aslp_dim3 Bl, float *data, const float *srcA_data, const float *srcB_data,
                                     MatrixDim dim, int srcA_stride, int srcB_stride, float alpha, float beta);                      /* [100] ASLP */
void cudaF_sigmoid(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x, MatrixDim d, int src_stride);                               /* [148] */
void cudaF_diff_sigmoid(aslp_dim3 Gr, aslp_dim3 Bl, float *eout, const float *e, const float *y, MatrixDim d, int e_stride, int y_stride); /* [149] */
void cudaF_tanh(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x, MatrixDim d, int src_stride);                                  /* [150] */
void cudaF_diff_tanh(aslp_dim3 Gr, aslp_dim3 Bl, float *eout, const float *e, const float *y, MatrixDim d, int e_stride, int y_stride);    /* [151] */
void cudaF_regularize_l1(aslp_dim3 Gr, aslp_dim3 Bl, float *wei, float *grad, float l1, float lr, MatrixDim d, int stride_grad);     /* [153] */

/* ---- index / gather ops (bit-exact) ------------------------------------------------ */
void cudaF_copy_cols(aslp_dim3 Gr, aslp_dim3 Bl, float *dst, const float *src, const MatrixIndexT_cuda *reorder, MatrixDim dst_dim, int src_stride); /* [64] */
void cudaF_add_cols(aslp_dim3 Gr, aslp_dim3 Bl, float *dst, const float *src, const MatrixIndexT_cuda *reorder, MatrixDim dst_dim, int src_stride);  /* [65] */
void cudaF_copy_rows(aslp_dim3 Gr, aslp_dim3 Bl, float *dst, const float *src, const MatrixIndexT_cuda *reorder, MatrixDim dst_dim, int src_stride); /* [66] */
void cudaF_add_rows(aslp_dim3 Gr, aslp_dim3 Bl, float alpha, float *dst, const float *src, const MatrixIndexT_cuda *reorder, MatrixDim dst_dim, int src_stride); /* [69] */
void cudaF_randomize(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x, const int32_cuda *copy_from, MatrixDim d_out, MatrixDim d_in); /* [158] */
void cudaF_splice(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x, const int32_cuda *off, MatrixDim d_out, MatrixDim d_in);          /* [159] */
void cudaF_copy(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x, const int32_cuda *copy_from, MatrixDim d_out, MatrixDim d_in);      /* [161] */
void cudaI32_set_const(aslp_dim3 Gr, aslp_dim3 Bl, int32_cuda *mat, int32_cuda value, MatrixDim d);                                       /* [37]  */

/* ---- reductions ---------------------------------------------------------------------- */
void cudaF_softmax_reduce(size_t Gr, size_t Bl, float *y, const float *x, MatrixDim d, int src_stride);        /* [143] */
void cudaF_log_softmax_reduce(size_t Gr, size_t Bl, float *y, const float *x, MatrixDim d, int src_stride);    /* [144] */
/* [154] processes the Bl.x columns starting at `mat` (column offset `voff`), keeping the running
 * row maximum in vec_val / vec_id exactly like the reference's per-256-column calls. */
void cudaF_find_row_max_id(aslp_dim3 Gr, aslp_dim3 Bl, const float *mat, float *vec_val, int32_cuda *vec_id, int32_cuda voff, MatrixDim d);
void cudaF_diff_xent(aslp_dim3 Gr, aslp_dim3 Bl, const int32_cuda *vec_tgt, float *mat_net_out, float *vec_log_post, MatrixDim d);       /* [155] */
/* [115] v = alpha * diag(op(M) * op(N)) + beta * v  (peephole gradients) */
void cudaF_add_diag_mat_mat(int Gr, int Bl, float alpha, float *v, int v_dim, const float *M, int M_cols, int M_row_stride,
                            int M_col_stride, const float *N, int N_row_stride, int N_col_stride, int threads_per_element, float beta);
void cudaF_add_vec_vec(int Gr, int Bl, float alpha, float *v, const float *x, const float *y, float beta, int dim);                     /* [118] ASLP */
void cudaF_vec_sum(int Gr, int Bl, float *v, float *value, int dim, int inc);                                                           /* [122] */

/* ---- element-wise / group ops of MaxPooling, Pnorm, Maxout (csrc/conv_pool.hip) ---- */
void cudaF_max(aslp_dim3 Gr, aslp_dim3 Bl, float *mat, const float *A, MatrixDim dst_d, int src_stride);                                  /* [85] mat = max(mat, A) */
void cudaF_mul_rows_group_mat(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x, MatrixDim d, int src_stride, int group_size);        /* [88] */
void cudaF_calc_pnorm_deriv(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x1, const float *x2, MatrixDim d, int src_stride, int group_size, float power); /* [89] */
void cudaF_calc_group_max_deriv(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x1, const float *x2, MatrixDim d, int src_stride, int group_size);          /* [90] */
void cudaF_group_pnorm(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x, MatrixDim d, int src_stride, int group_size, float power);  /* [146] */
void cudaF_group_max(aslp_dim3 Gr, aslp_dim3 Bl, float *y, const float *x, MatrixDim d, int src_stride, int group_size);                 /* [147] */
void cudaF_equal_element_mask(aslp_dim3 Gr, aslp_dim3 Bl, const float *mat1, const float *mat2, float *mask, MatrixDim mat1_dim, int mat2_stride,
                              int mask_stride);                                                                                          /* [182] */

/* ---- ConvolutionalComponent / MaxPoolingComponent / LengthNormComponent / Pnorm as whole ops (csrc/conv_pool.hip) ----
 * ConvolutionalComponent (nnet-convolutional-component.h:301-481): the reference gathers the P patches of a frame into P column blocks
 * (CopyCols) and runs P products; here the patches of frame n are the P consecutive ROWS n * P .. n * P + P - 1 of `patches`
 * ([rows * P x num_splice * patch_dim], leading dimension ldp), so that forward, in-diff and filter gradient are ONE product each. */
void aslp_conv_gather_patches(float *patches, int ldp, const float *in, MatrixDim d_in, int num_patches, int num_splice, int patch_dim, int patch_step,
                              int patch_stride);
/* in_diff[n][s * patch_stride + q] = sum over patches p containing q (ascending p) of patch_diff[n * P + p][s * patch_dim + q - p * step]
 * (the AddCols passes over the reversed column map, :413-421, in their order) */
void aslp_conv_in_diff(float *in_diff, MatrixDim d_id, const float *patch_diff, int ldp, int num_patches, int num_splice, int patch_dim, int patch_step,
                       int patch_stride);
/* nnet-max-pooling-component.h:101-162: out [rows x num_pools * pool_stride]; the backward pass routes out_diff to the inputs that equal their
 * pool's maximum, summed over overlapping pools (ascending) and divided by the number of pools a patch belongs to */
void aslp_max_pool_forward(float *out, int ld_out, const float *in, MatrixDim d_in, int pool_size, int pool_step, int pool_stride);
void aslp_max_pool_backward(float *in_diff, int ld_id, const float *in, MatrixDim d_in, const float *out, int ld_out, const float *out_diff, int ld_od,
                            int pool_size, int pool_step, int pool_stride);
/* nnet-various.h:338-352: out = in * row_scales, row_scales[r] = 1 / sqrt(sum_c in[r][c]^2) (kept for the backward MulRowsVec) */
void aslp_length_norm_forward(float *out, int ld_out, const float *in, MatrixDim d_in, float *row_scales);
/* GroupPnormDeriv / GroupMaxDeriv followed by MulRowsGroupMat(out_diff) in one pass (nnet-activation.h:346-349, 370-373) */
void aslp_group_pnorm_backward(float *in_diff, int ld_id, const float *in, MatrixDim d_in, const float *out, int ld_out, const float *out_diff, int ld_od,
                               int group_size, float power);
void aslp_group_max_backward(float *in_diff, int ld_id, const float *in, MatrixDim d_in, const float *out, int ld_out, const float *out_diff, int ld_od,
                             int group_size);

/* Library-native whole-op forms used by the host engine (no reference twin: the reference
 * builds these from cuBLAS gemv with a ones-vector, cu-vector.cc:1145-1166). */
/* v[c] = alpha * sum_r M[r][c] + beta * v[c]      (CuVectorBase::AddRowSumMat) */
void aslp_add_row_sum_mat_vec(float alpha, const float *M, MatrixDim d, float beta, float *v);
/* the same with the SGD step fused into the finalize pass: v = alpha*colsum + beta*v; w += w_alpha * v
 * (bias_corr_ / bias_ of AffineTransform::Update, nnet-affine-transform.h:211,229) */
void aslp_add_row_sum_mat_vec_sgd(float alpha, const float *M, MatrixDim d, float beta, float *v, float *w, float w_alpha);
/* v[r] = alpha * sum_c M[r][c] + beta * v[r]      (CuVectorBase::AddColSumMat) */
void aslp_add_col_sum_mat_vec(float alpha, const float *M, MatrixDim d, float beta, float *v);
/* whole-row argmax, first strict maximum (CuMatrixBase::FindRowMaxId, cu-matrix.cc:1466) */
void aslp_find_row_max_id(const float *M, MatrixDim d, int32_cuda *id);
/* sum of all elements -> *out_dev (double, one value)  (CuMatrixBase::Sum) */
void aslp_matrix_sum(const float *M, MatrixDim d, double *out_dev);
void aslp_copy_mat(float *dst, MatrixDim d, const float *src, int src_stride);
/* dst [d] = src^T, src [d.cols x d.rows] (CuMatrixBase::CopyFromMat(M, kTrans), cu-matrix.cc:297-316) */
void aslp_copy_mat_trans(float *dst, MatrixDim d, const float *src, int src_stride);
void aslp_vec_axpy(float alpha, const float *x, float *y, int dim);           /* y += alpha x */
void aslp_vec_axpy2(float alpha, const float *x1, float *y1, const float *x2, float *y2, int dim); /* two at once */
void aslp_vec_diff(float *out, const float *a, const float *b, int n);        /* out = a - b */
/* ---- SOD model sync: one pass per tensor (csrc/optim_kernels.hip; aslp-parallel/optimizer.h:40-171, sod-worker.cc:46-58) ----
 * Advances the solver state (state1 / state2: what the solver keeps per element, zero at the start; unused ones may be NULL),
 * applies the step to param and copies the new param to prev:
 *   sgd       p -= lr g
 *   momentum  m = lr g + momentum m;  p -= m                                          (state1 = m)
 *   adagrad   G = g g + G;             p -= lr g / sqrt(max(G, 1e-8))                  (state1 = G)
 *   rmsprop   G = 0.1 g g + 0.9 G;     p -= lr g / sqrt(max(G, 1e-8))                  (state1 = G)
 *   adadelta  G = (1-gamma) g g + gamma G;  d = g sqrt(max(D, 1e-8)) / sqrt(max(G, 1e-8));  p -= d;  D = (1-gamma) d d + gamma D
 *   adam      m = (1-b1) g + b1 m;  v = (1-b2) g g + b2 v;  p -= lr corr1 m / sqrt(max(v corr2, 1e-8))
 *             with corr1 = 1 / (1 - b1^t), corr2 = 1 / (1 - b2^t) computed by the caller for its step counter t = 1, 2, ... */
enum { ASLP_SOD_SGD = 0, ASLP_SOD_MOMENTUM = 1, ASLP_SOD_ADAGRAD = 2, ASLP_SOD_RMSPROP = 3, ASLP_SOD_ADADELTA = 4, ASLP_SOD_ADAM = 5 };
typedef struct aslp_sod_solver_ {
  int solver;
  float lr, momentum, gamma, beta1, beta2, corr1, corr2;
} aslp_sod_solver;
void aslp_sod_solve(const aslp_sod_solver *a, const float *grad, float *param, float *prev, float *state1, float *state2, int n);
void aslp_f2d(double *dst, const float *src, int n);
void aslp_d2f(float *dst, const double *src, int n);

/* ---- B2: BLAS seam (cublas-wrappers.h:28-47, call site cu-matrix.cc:1027-1061) ------ */
/* Row-major C[M x N] = alpha * op(A) * op(B) + beta * C with fp32 MFMA; transX: 0 = kNoTrans,
 * 1 = kTrans (MatrixTransposeType).  A is [M x K] (or [K x M] if transA), B is [K x N] (or
 * [N x K] if transB).  Returns 0 or a negative argument-error code. */
int aslp_sgemm(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda,
               const float *B, int ldb, float beta, float *C, int ldc);
typedef struct aslp_planes_ aslp_planes;   /* prepared operand planes: below, behind aslp_gemm_split16 */
/* What a kernel that WRITES a matrix may leave for the products that will read it (csrc/split16.h): the matrix' planes, scaled by a bound
 * of |value| that is known before the launch (`slot`: its bits, in device memory), and / or one maximum of |value| per workgroup
 * (`parts`, at most 256) from which the conversion pass takes its scale without reading the matrix for it.  All members optional. */
typedef struct aslp_planes_out_ {
  void *hi, *lo;          /* fp16 planes [rows rounded to 64][ld], or NULL */
  int ld;                 /* halves per plane row */
  const unsigned *slot;   /* device word: bits of the bound (float) the planes are scaled by */
  float *parts;           /* device, 256 floats: per-workgroup maxima, or NULL */
  int nparts;             /* out: how many the launch wrote (0: none, the kernel that ran does not leave them) */
  int planes_written;     /* out: 1 = the launch found the matrix' maximum itself, stored its bits to *slot and wrote hi / lo scaled by it
                           * (a kernel whose output has no bound known before the launch: aslp_bn_backward_step_p); parts are not left then */
} aslp_planes_out;
/* Fused epilogue form.  Applied in this order on the fp32 accumulator `acc`:
 *   v = alpha*acc + beta*C;  if (bias) v += bias[col];  if (clip > 0) v = clamp(v, -clip, clip);
 *   C = v;  if (W) W[row][col] += w_alpha * v  (SGD: w_alpha = -lr);
 *   if (act_out) act_out[row][col] = act(v)  with act: 1 sigmoid, 2 tanh, 3 relu. */
typedef struct aslp_gemm_epilogue_ {
  const float *bias;   /* [N] or NULL */
  float clip;          /* <= 0: off */
  float *W;            /* same shape as C, or NULL */
  int ldw;
  float w_alpha;
  float *act_out;      /* or NULL */
  int ld_act;
  int act;
  /* transA products only (A is [K x M]): colsum[m] = sum_k A[k][m] + colsum_beta * colsum[m]; if colsum_w,
   * colsum_w[m] += colsum_w_alpha * colsum[m].  This is the bias gradient + bias SGD step of AffineTransform::Update
   * (nnet-affine-transform.h:214-216, 227) riding on the weight-gradient GEMM, whose A operand is the same `diff`. */
  float *colsum;       /* [M] or NULL */
  float colsum_beta;
  float *colsum_w;     /* [M] or NULL */
  float colsum_w_alpha;
  /* Column statistics of the OUTPUT, for a BatchNormalization that consumes it (the forward products: beta == 0, no W): for every
   * group of 32 consecutive rows g = row / 32 and every column n, over the values v just stored (after alpha, bias, clip),
   *   colstats[(a * groups + g) * colstats_ld + n],  groups = ceil(M / 32),  a = 0: sum v   1: sum (float)(v * v)   2: sum (double)v * v
   * in double.  The statistics pass of aslp_bn_forward_stats then adds `groups` partials per column instead of reading the matrix.
   * Always filled on return (by the GEMM kernel's epilogue where it can, by one extra pass over C otherwise). */
  double *colstats;    /* [3 * groups * colstats_ld] or NULL */
  int colstats_ld;     /* >= N */
  /* the beta term read from another matrix: C = alpha op(A) op(B) + beta * c_src (+ ...) instead of + beta * C.  E.g. the LSTM's
   * d_r = out_diff + dGATES(next) W_r (lc.h:791) without first copying out_diff into d_r.  NULL: the classic form. */
  const float *c_src;  /* [M x N], leading dimension ld_c_src, or NULL */
  int ld_c_src;
  /* What the epilogue leaves for the split-fp16 products that will read its output (honoured by the split-fp16 kernel only:
   * aslp_gemm_last_parts() tells whether the launch did).  planes_of = 1: the planes of W after the fused step (`planes.slot` must hold
   * a bound of |W + w_alpha C| before the launch: aslp_weight_bound); 2: the planes of act_out (bound: 1 for sigmoid / tanh).
   * wmax_parts / cmax_parts (each NULL or aslp_gemm_last_parts() floats): one maximum per workgroup of |W| after the step / of |C|. */
  aslp_planes_out planes;
  int planes_of;
  float *wmax_parts, *cmax_parts;
  /* planes_of = 1 without a separate aslp_weight_bound launch: the kernel forms the bound itself before its first tile (every workgroup
   * the same value, workgroup 0 stores it to planes.slot) from the bound_n per-workgroup maxima of |W| and |C_old| (bound_c_parts may be
   * NULL when beta == 0) the previous step left, the bounds of its two operands' planes and K. */
  const float *bound_w_parts, *bound_c_parts;
  int bound_n;
} aslp_gemm_epilogue;
/* number of per-workgroup maxima (and proof that the planes were written) of the calling thread's latest aslp_sgemm* call; 0 = the kernel that
 * ran does not leave them */
int aslp_gemm_last_parts(void);
/* Bound of |W + w_alpha C| for the next fused weight step, C = clip(alpha A^T B + beta C_old): from the per-workgroup maxima of |W| (n_w) and
 * |C_old| (n_c, may be 0) the previous step left, the bounds of the two operands' planes and K; written to *slot_out (device).  One
 * small launch on the current stream. */
void aslp_weight_bound(const float *w_parts, int n_w, const float *c_parts, int n_c, const aslp_planes *a, const aslp_planes *b, int K,
                       float alpha, float beta, float w_alpha, float clip, aslp_planes *w_planes);
/* Parameters were written through the raw pointers of GetGpuParams (model averaging, a test perturbing weights): planes of weights the
 * components keep from step to step are stale from here on.  The native sync workers call it before and after every exchange.  It also makes
 * the calling thread's stream wait for weight updates its latest backward pass left running on the side stream (Nnet::Backpropagate puts
 * that wait off into the next forward pass): call it on the training thread BEFORE reading or writing parameters through raw pointers. */
void aslp_params_changed(void);
/* A/B switch (ASLP_KEEP_WEIGHT_PLANES): 0 = the components convert their weights in every step instead of keeping the planes the
 * weight-gradient epilogue wrote; -1 = back to the environment's choice (default on) */
void aslp_keep_weight_planes(int on);
/* per-workgroup maxima of |src| into parts (256 floats): the start of the chain above */
void aslp_absmax_parts(const float *src, MatrixDim d, float *parts);
/* Large products of aslp_sgemm_ex on the fp16 matrix instruction with each fp32 operand carried as two fp16 pieces behind a power-of-two
 * scale of its matrix (csrc/gemm_split16.hip: 22 significant bits, fp32 accumulation; results agree with the fp32 instruction's to fp32
 * rounding).  on = 1 / 0 switches it for this process, -1 hands the choice back to ASLP_GEMM_SPLIT_F16 (default on).  No reference
 * counterpart (cuBLAS sgemm on fp32 CUDA cores). */
void aslp_gemm_split16(int on);
/* Tile configuration of those products, by number (csrc/gemm_split16.hip: 304 / 308 / 311 one-role kernels, 351 = 128 x 128 with producer and
 * consumer waves, 312 = 311 whatever ASLP_GEMM_S16_PC says, 328 = 128 x 128 with both operands reduction-major); 0 = the heuristic, -1 hands
 * the choice back to ASLP_GEMM_SPLIT_F16_TILE.  A number the product's layout or epilogue does not admit falls back to the heuristic's
 * choice.  Tests and devtools only: every configuration forms the same bits. */
void aslp_gemm_split16_tile(int cfg);
/* Prepared operands of such products: the two fp16 planes of an fp32 matrix, in the matrix' own layout (csrc/split16.h), made once and
 * read by every product the matrix takes part in (as op(A) or op(B), transposed or not).  aslp_planes_convert: one maximum pass and one
 * conversion pass over src [d.rows x d.cols] (cols and stride multiples of 4, 16-byte aligned).  The engine's components keep such
 * planes per tensor and step and let the kernels that write a tensor write its planes (nnet/nnet-basic.h). */
aslp_planes *aslp_planes_new(void);
void aslp_planes_free(aslp_planes *p);
int aslp_planes_convert(aslp_planes *p, const float *src, MatrixDim d);
/* aslp_sgemm_ex with the planes of A and / or B (NULL: that operand is converted inside the call); the planes must be those of the
 * matrix the fp32 pointer names.  Runs on the fp32 instruction like aslp_sgemm_ex when aslp_gemm_split16 is off or the shape is not
 * served. */
/* the members a producer needs, from an aslp_planes whose bound has been set / whose parts are to be filled */
void aslp_planes_reserve(aslp_planes *p, int rows, int cols);
void aslp_planes_set_bound(aslp_planes *p, float bound);
void aslp_planes_as_output(const aslp_planes *p, aslp_planes_out *out);
int aslp_sgemm_planes_ex(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, const aslp_planes *pa,
                         const float *B, int ldb, const aslp_planes *pb, float beta, float *C, int ldc, const aslp_gemm_epilogue *ep);
int aslp_sgemm_ex(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda,
                  const float *B, int ldb, float beta, float *C, int ldc, const aslp_gemm_epilogue *ep);
/* Two products of the same shape, leading dimensions, alpha and beta in ONE launch: C0 = alpha op(A0) op(B0) + beta C0 (+ ep0),
 * C1 = alpha op(A1) op(B1) + beta C1 (+ ep1).  Each output is computed exactly as aslp_sgemm_ex computes it (two such calls are what
 * runs wherever the paired kernel does not apply); only the K split chosen for a grid that cannot fill the chip may differ from
 * the single product's, i.e. the order in which the K chunks' partial sums are added (still fixed: run-to-run reproducible).
 * The outputs (and every array the epilogues write) must not overlap.
 * The two directions of a bidirectional recurrent layer issue each of their batched products as such a pair. */
int aslp_sgemm_pair_ex(int transA, int transB, int M, int N, int K, float alpha, const float *A0, const float *A1, int lda,
                       const float *B0, const float *B1, int ldb, float beta, float *C0, float *C1, int ldc,
                       const aslp_gemm_epilogue *ep0, const aslp_gemm_epilogue *ep1);
/* GEMM launch statistics for bench.py's roofline: per variant launches / flops / (if
 * profiling was enabled with aslp_gemm_profile(1)) event-timed milliseconds. */
void aslp_gemm_profile(int enable);
void aslp_gemm_profile_reset(void);
/* tuning aid: force tile config 1..5 (0 = heuristic).  1: 32x128x16  2: 64x64x16  3: 128x64x32  4: 128x128x32  5: 128x128x16 */
void aslp_gemm_force_tile(int cfg);
/* tile configuration of the calling thread's latest aslp_sgemm / aslp_sgemm_ex (the numbers aslp_gemm_profile_tile names) */
int aslp_gemm_last_tile(void);
/* variant: 0 = NT, 1 = NN, 2 = TN, 3 = TT.  Returns number of launches. */
long aslp_gemm_profile_get(int variant, double *flops, double *ms);
/* the tile configuration that carried most of that variant's flops since the last reset: returns its number, writes a
 * description ("gemm_f32_glds 64x128x32, 8 waves, LDS-DMA, 3 stages") into buf */
int aslp_gemm_profile_tile(int variant, char *buf, int buflen);
void aslp_gemm_profile_dump(void);   /* devtools: per-shape table of the event-timed launches, to stderr */
/* Named region timers (HIP events on the launch stream, off by default): the engine brackets "lstm_recurrence_fwd",
 * "lstm_recurrence_bwd" (the timestep loops of the LSTM family), "gru_recurrence_fwd" / "_bwd" and "ctc_loss" (Warp-CTC / Eesen
 * forward-backward).  aslp_region_get returns the number of regions recorded under `name` and their summed milliseconds. */
void aslp_region_profile(int enable);
void aslp_region_reset(void);
long aslp_region_get(const char *name, double *ms);

/* ---- fused hot-path ops (one pass each; see DESIGN.md for bytes/unit) ----------------- */
/* BatchNormalization training forward (nnet-batch-normalization.h:177-220): writes out,
 * xhat (XsharpO_), mean[D], inv_std[D] ("var_vec_"), and adds the batch sums to the double
 * running statistics acc_means/acc_vars (may be NULL). */
void aslp_bn_forward(const float *in, MatrixDim d, float *out, int out_stride, float *xhat, int xhat_stride,
                     const float *scale, const float *shift, float *mean, float *inv_std,
                     double *acc_means, double *acc_vars, float var_floor);
/* BatchNormalization backward (same file :222-277): dscale = sum(xhat*dy) + mmt*dscale,
 * dshift = sum(dy) + mmt*dshift, in_diff as the reference's 4 steps; xhat is overwritten
 * with dy*scale like the reference (XsharpO_). in_diff may be NULL. */
void aslp_bn_backward(const float *in, MatrixDim d, const float *out_diff, int od_stride, float *xhat, int xhat_stride,
                      const float *scale, const float *mean, const float *inv_std, float *dscale, float *dshift,
                      float momentum, float *in_diff, int id_stride);
/* inference with given mean / inv_std (same file :166-173) */
/* The same with a Sigmoid component fused behind the normalisation (nnet-activation.h:153-175): forward also writes
 * act_out = sigmoid(bn_out) (out may then be NULL); backward takes the diff w.r.t. the SIGMOID output plus that output
 * act_y and forms dy = out_diff .* y .* (1 - y) on the fly.  act_* == NULL gives the plain functions above. */
void aslp_bn_forward_act(const float *in, MatrixDim d, float *out, int out_stride, float *xhat, int xhat_stride, const float *scale,
                         const float *shift, float *mean, float *inv_std, double *acc_means, double *acc_vars, float var_floor, float *act_out,
                         int act_stride);
void aslp_bn_backward_act(const float *in, MatrixDim d, const float *out_diff, int od_stride, float *xhat, int xhat_stride, const float *scale,
                          const float *mean, const float *inv_std, float *dscale, float *dshift, float momentum, float *in_diff, int id_stride,
                          const float *act_y, int act_stride);
/* aslp_bn_forward_act with the column statistics already formed by the producer of `in` (aslp_gemm_epilogue.colstats: `groups` partial
 * sums per column and statistic, leading dimension stats_ld): no statistics pass and no exchange between workgroups, the launch only
 * streams `in` once.  Same results up to the order in which the partial sums are added (all in double).  Returns 1 if it ran, 0 if
 * this shape is not served (the caller then uses aslp_bn_forward_act; nothing was written). */
int aslp_bn_forward_stats(const float *in, MatrixDim d, float *out, int out_stride, const float *scale, const float *shift, float *mean,
                          float *inv_std, double *acc_means, double *acc_vars, float var_floor, float *act_out, int act_stride,
                          const double *colstats, int groups, int stats_ld);
/* ... which also leaves the planes of act_out (sigmoid outputs: bound 1; `act_planes->slot` must hold a bound >= 1) */
int aslp_bn_forward_stats_p(const float *in, MatrixDim d, float *out, int out_stride, const float *scale, const float *shift, float *mean,
                            float *inv_std, double *acc_means, double *acc_vars, float var_floor, float *act_out, int act_stride,
                            const double *colstats, int groups, int stats_ld, const aslp_planes_out *act_planes);

/* aslp_bn_backward_act + BatchNormalization::Update (nnet-batch-normalization.h:280-284) taken in the statistics
 * finalize: scale -= learn_rate * dscale, shift -= learn_rate * dshift; in_diff is formed with the scale the
 * forward pass used. */
void aslp_bn_backward_step(MatrixDim d, const float *out_diff, int od_stride, float *xhat, int xhat_stride, float *scale, float *shift,
                           const float *inv_std, float *dscale, float *dshift, float momentum, float learn_rate, float *in_diff, int id_stride,
                           const float *act_y, int act_stride, const float *in, const float *mean);
/* ... which also leaves the per-workgroup maxima of |in_diff| (diff_out->parts / nparts; nparts stays 0 where the kernel that served the
 * shape does not form them) */
void aslp_bn_backward_step_p(MatrixDim d, const float *out_diff, int od_stride, float *xhat, int xhat_stride, float *scale, float *shift,
                             const float *inv_std, float *dscale, float *dshift, float momentum, float learn_rate, float *in_diff, int id_stride,
                             const float *act_y, int act_stride, const float *in, const float *mean, aslp_planes_out *diff_out);
/* xhat == NULL in the forward / backward entry points above: no normalised copy of the input is kept; the backward pass forms
 * x_hat again from `in` (d.stride) and `mean` with the forward pass' own operations.  Only where this returns 1 (the
 * single-launch panel kernels: cols % 16 == 0, rows <= 1024) and all operands are 16-byte aligned. */
int aslp_bn_panel_supported(int rows, int cols);
void aslp_bn_apply(const float *in, MatrixDim d, float *out, int out_stride, const float *mean, const float *inv_std,
                   const float *scale, const float *shift);
/* Xent::Eval (nnet-loss.cc:63-122) in one pass over [rows x cols]:
 * diff = (y - t) * w_row with w_row = frame_weight * sum_c t;  stats_dev[0..4] (double) +=
 * {frames, correct, -sum w t log(y+1e-20), -sum w t log(t+1e-20), sum w t y}.
 * Dense targets (tgt != NULL) or one label per row (labels != NULL: one-hot posterior). */
void aslp_xent_eval(const float *net_out, MatrixDim d, const float *tgt, int tgt_stride, const int32_cuda *labels,
                    const float *frame_weights, float *diff, int diff_stride, double *stats_dev);
/* Softmax::PropagateFnc (nnet-activation.h:46-49) + Xent::Eval over the activations IN FRONT of the softmax, one
 * pass; post_out (nullable) receives the posteriors, bit-identical to cudaF_softmax_reduce. Only for
 * aslp_softmax_xent_supported(cols) (513..8192 classes). */
int aslp_softmax_xent_supported(int cols);
void aslp_softmax_xent_eval(const float *acts, MatrixDim d, const float *tgt, int tgt_stride, const int32_cuda *labels,
                            const float *frame_weights, float *diff, int diff_stride, double *stats_dev, float *post_out, int post_stride);
/* aslp_xent_eval / aslp_softmax_xent_eval (softmax != 0) which also leave the planes of `diff` (diff_planes->slot must hold a bound of
 * |diff| = |y - t| w: max w for posteriors against a distribution).  Returns 1 if the planes were written (label targets on the
 * register-cached row kernel), 0 if only diff was. */
int aslp_xent_eval_p(const float *net_out, MatrixDim d, const int32_cuda *labels, const float *frame_weights, float *diff, int diff_stride,
                     double *stats_dev, int softmax, const aslp_planes_out *diff_planes);
/* dst [d] <- src (dst may be NULL) and the planes of src in one launch: the launch's workgroups find the matrix maximum among themselves,
 * store its bits to *out->slot and write out->hi / lo scaled by it -- what aslp_copy_mat + aslp_planes_convert leave, in one launch instead
 * of three.  Returns 1 (out->planes_written set) or 0 = not served (more rows than the chip holds at once, unaligned operands): nothing
 * was written, the caller copies and converts. */
int aslp_copy_mat_planes(float *dst, MatrixDim d, const float *src, int src_stride, aslp_planes_out *out);
/* The same single launch also serves aslp_planes_convert and the components' own conversions where the matrices fit one resident grid;
 * aslp_coop_convert(0) sends them through the maximum pass + conversion pass again (same planes, bit for bit: tests compare the two). */
void aslp_coop_convert(int on);
/* aslp_xent_eval_p in two halves, for a caller that evaluates batch after batch and reads its accumulators rarely (Xent: at Report()):
 * aslp_xent_eval_rows leaves the batch's per-row statistics in rowstats_out [d.rows x 5 doubles] instead of adding them up -- one launch
 * less in front of the backward pass of every step; aslp_xent_sum_rowstats adds `batches` such blocks of `rows` rows each (consecutive in
 * memory, in evaluation order) to stats_dev, with the bits of one aslp_xent_eval_p per batch. */
int aslp_xent_eval_rows(const float *net_out, MatrixDim d, const int32_cuda *labels, const float *frame_weights, float *diff, int diff_stride,
                        double *rowstats_out, int softmax, const aslp_planes_out *diff_planes);
void aslp_xent_sum_rowstats(const double *rowstats, int rows, int batches, double *stats_dev);
/* cudaF_diff_sigmoid which also leaves the planes of its result: |e y (1 - y)| <= max |e| / 4 with max |e| from the n_parts per-workgroup
 * maxima the producer of e left (aslp_gemm_epilogue.cmax_parts).  The kernel stores the bound to out_planes->slot itself.  Returns 1 if the
 * planes were written, 0 if only eout was (operands not 16-byte aligned, no maxima). */
int aslp_diff_sigmoid_p(float *eout, const float *e, const float *y, MatrixDim d, int e_stride, int y_stride, const float *e_max_parts, int n_parts,
                        const aslp_planes_out *out_planes);
/* PosteriorToMatrix scatter: mat[row[i]][col[i]] += val[i]  (hmm/posterior.cc, used nnet-loss.cc:168) */
void aslp_scatter_add(float *mat, MatrixDim d, const int32_cuda *rows, const int32_cuda *cols, const float *vals, int n);
/* Splice backward (nnet-various.h:143-175): in_diff[t] = sum_k out_diff[clamp(t+off[k])][k-th block] */
void aslp_splice_backward(float *in_diff, MatrixDim d_in, const float *out_diff, int od_stride, const int32_cuda *off, int n_off);
/* Dropout (nnet-activation.h:240-258) in one launch per pass: mask = [u < retention] with u from a counter-based
 * generator keyed by (seed, element index); out = in * mask / retention; in_diff = out_diff * mask / retention. */
void aslp_dropout_forward(float *out, int out_stride, const float *in, MatrixDim d, float *mask, int mask_stride, float retention,
                          unsigned long long seed);
void aslp_dropout_backward(float *in_diff, int id_stride, const float *out_diff, MatrixDim d, const float *mask, int mask_stride, float retention);
/* ApplyFloor(lo) then ApplyCeiling(hi) in one pass (the gradient clipping of the recurrent components) */
void aslp_apply_clamp(float *mat, MatrixDim d, float lo, float hi);
/* ReLU backward: in_diff = heaviside(in) * out_diff (nnet-activation.h:292-297) */
void aslp_diff_relu(float *in_diff, const float *in, const float *out_diff, MatrixDim d, int in_stride, int od_stride);
/* ---- recurrent gate blocks, one launch per timestep (csrc/rnn_cells.hip) -------------------------
 * Pointers are to the row block of the (T+2)*S x width activation / diff buffers (row = t*S + s),
 * column 0; column layout [g|i|f|o|c|h|m|r] (cifg: [g|f|o|c|h|m|r]).
 * forward (nnet-blstm-projected-streams-lc.h:571-609): y_cur's gate columns hold the pre-activations
 *   (x-part + bias + recurrent part); writes g,i,f,o,c,h,m.  seq_lengths != NULL: rows with
 *   t > seq_lengths[s] are zeroed (nnet-blstm-projected-streams.h:654-657).
 * backward (:783-835): d_cur's m column holds dL/dm; writes d_g,d_i,d_f,d_o,d_c,d_h.  d_next / y_next:
 *   the step processed just before in BPTT order; y_prev: the step the forward recursion read. */
void aslp_lstm_cell_forward(float *y_cur, const float *y_prev, int ld, int S, int C, int cifg, const float *peep_i, const float *peep_f,
                            const float *peep_o, const int32_cuda *seq_lengths, int t);
void aslp_lstm_cell_backward(float *d_cur, const float *d_next, const float *y_cur, const float *y_next, const float *y_prev, int ld, int S,
                             int C, int cifg, const float *peep_i, const float *peep_f, const float *peep_o);
/* Vector gradients of a recurrent component, up to 8 (both directions of a bidirectional layer) in ONE launch (nnet-blstm-projected-streams-lc.h:1005-1058: the bias
 * gradient AddRowSumMat, the peephole gradients AddDiagMatMat, each followed by ApplyFloor / ApplyCeiling; :1092-1104 the step):
 *   corr[c] = clamp( sum_r d[r*ldd + c] * (x ? x[r*ldx + c] : 1) + mmt * corr[c], -clip, clip )     (clip <= 0: no clamp)
 *   param[c] += neg_lr * corr[c]                                                                   (neg_lr == 0: no step) */
typedef struct aslp_rnn_vec_grad_ {
  const float *d;          /* first row / first column of the diff block */
  const float *x; int ldx; /* element-wise weight block (NULL: plain column sum) */
  int n;                   /* columns */
  float *corr, *param;
} aslp_rnn_vec_grad;
void aslp_rnn_vec_grads(const aslp_rnn_vec_grad *jobs, int njobs, int ldd, int rows, float mmt, float clip, float neg_lr);
/* Fused recurrence step (csrc/rnn_fused.hip): ONE launch per timestep covers every direction.
 * forward : gates(t) += m(t-1) * w^T with w = W_eff = W_r W_rm [G*C x C] (W_r itself without projection), then the
 *           whole gate block of aslp_lstm_cell_forward; the projection r = m W_rm^T is NOT on the sequential path.
 * backward: d_m(t) += dGATES(next) * W_eff (w = W_eff^T [C x G*C]), then the gate block of aslp_lstm_cell_backward.
 * Requirements: C, ld, ldw multiples of 4 (16-byte operand loads). */
typedef struct aslp_lstm_step_dir_ {
  float *y_cur; const float *y_prev;                 /* activation row blocks: step t / recursion-previous */
  float *d_cur; const float *d_next; const float *y_next; /* backward: diff blocks t / recursion-next, activations of recursion-next */
  const float *w;                                    /* forward: W_eff; backward: W_eff^T */
  const float *peep_i, *peep_f, *peep_o;
  const int32_cuda *seq_lengths; int t;              /* forward length masking (may be NULL) */
  int has_next;                                      /* backward: 0 on the first BPTT step (next block is all zero) */
  int no_product;                                    /* forward: skip the m(t-1) W_eff^T term (the caller already added it) */
} aslp_lstm_step_dir;
typedef struct aslp_lstm_step_ {
  aslp_lstm_step_dir dir[2];
  int ndir, ld, ldw, S, C, cifg;
} aslp_lstm_step;
void aslp_lstm_step_forward(const aslp_lstm_step *a);
void aslp_lstm_step_backward(const aslp_lstm_step *a);
/* The same recurrence for a WHOLE sequence in one launch per pass (csrc/rnn_persistent.hip): the workgroups stay resident
 * for all T timesteps, the weights they multiply with live in registers, and m(t) / dGATES(t) travel between workgroups
 * through the buffers themselves (write-through stores, agent-scope loads, "not yet written" = the bit pattern 0xFFFFFFFF).
 *   y : activation buffer [(T+2)*S x ld] (row = t*S + s), forward: gate columns of rows 1..T hold x-part + bias
 *   d : diff buffer of the same shape (backward only), m columns of rows 1..T hold dL/dm from the layer above
 * Contract: the caller prepared y with aslp_lstm_seq_fill BEFORE it stored anything into it (forward; the backward pass
 * exchanges through an internal ring and takes an ordinarily zero-initialised d), and calls the kernel only when aslp_lstm_seq_supported says 1 (directions x ceil(S / 8) <= 8 chains, C <= 512,
 * the grid co-resident on the device); otherwise it keeps the per-timestep entry points above. */
typedef struct aslp_lstm_seq_dir_ {
  float *y;
  float *d;
  const float *w;                         /* W_eff [G*C x C], both passes */
  const float *peep_i, *peep_f, *peep_o;
  const int32_cuda *seq_lengths;          /* forward length masking (may be NULL) */
  int reverse;                            /* 0: the recursion runs t = 1..T, 1: t = T..1 (BPTT runs against it) */
  int skip_first_product;                 /* forward: the caller already added the recurrent term of the first step */
  /* forward, optional: the first step's recurrent term is r(0) W_first^T instead of m(0) W_eff^T -- the carried history of a layer with
   * a projection holds r(0) formed with the weights of the previous batch (lc.h:575) --, r(0) = columns [col_first, col_first + k_first)
   * of the history row block, w_first = W_r [G*C x k_first] (leading dimension ldw_first).  Served inside the launch for k_first <= 256
   * (aslp_lstm_seq_first_product_supported); otherwise the caller adds the term itself and sets skip_first_product. */
  const float *w_first;
  int ldw_first, k_first, col_first;
} aslp_lstm_seq_dir;
typedef struct aslp_lstm_seq_ {
  aslp_lstm_seq_dir dir[2];
  int ndir, ld, ldw, T, S, C, cifg;
  /* backward, optional: the kernel also leaves what the bias and peephole gradients (lc.h:1005-1058) are sums of -- per chain (direction
   * x group of 8 streams; chain index = group * ndir + direction, at most 8) the sums over its streams and all T timesteps of
   *   0..3: d_g, d_i, d_f, d_o      4: d_i * c(t-1)      5: d_f * c(t-1)      6: d_o * c(t)        (no d_i rows with cifg)
   * at grad_partial[(chain * 7 + k) * grad_ld + cell].  aslp_lstm_seq_vec_grads finishes them; NULL: nothing is formed. */
  float *grad_partial;
  int grad_ld;
  /* stream window: this launch serves streams [s_begin, s_begin + s_count) of the S streams the buffers hold (s_count == 0: all of them).
   * More streams than one launch has chains for (8 chains of 8 streams: 32 bidirectional, 64 unidirectional) go in several launches. */
  int s_begin, s_count;
  /* backward, optional: per direction 256 floats -- the kernel leaves the largest finite |d_g, d_i, d_f, d_o| each workgroup wrote (0 for the
   * other direction's workgroups), i.e. per-workgroup maxima of the direction's dGATES block for aslp_planes conversions without a maximum
   * pass.  Only with one launch per pass (s_count == 0); aslp_lstm_seq_last_dmax() says whether the kernel that ran wrote them. */
  float *dmax_parts[2];
} aslp_lstm_seq;
int aslp_lstm_seq_last_dmax(void);   /* 256 if the calling thread's latest aslp_lstm_seq_backward left dmax_parts, else 0 */
int aslp_lstm_seq_supported(const aslp_lstm_seq *a, int backward);
/* Several processes on ONE GPU (ShmComm ranks, or two jobs given the same device): on != 0 makes every persistent LSTM / GRU launch hold a
 * per-device cross-process lock until its kernel has completed, so that two grids which each need the whole device are never half resident
 * beside each other (they would wait for each other's unscheduled workgroups until the spin limit).  Same as ASLP_DEVICE_SHARED=1.  Default off.
 * No reference counterpart: the reference's per-timestep kernels need no co-residency. */
void aslp_device_shared(int on);
/* The persistent LSTM recurrences multiply on v_mfma_f32_16x16x32_f16 with every fp32 operand carried as two fp16 pieces behind power-of-two
 * scales (the default: closer to a float64 product than an fp32 fma chain, csrc/rnn_persistent.hip lstm_seq_fwd_h / lstm_seq_bwd_h).  on = 0
 * puts them on the fp32 instruction v_mfma_f32_4x4x1 (lstm_seq_fwd / lstm_seq_bwd), 1 back, -1 hands the choice to ASLP_LSTM_SPLIT_F16. */
void aslp_lstm_split16(int on);
int aslp_lstm_seq_first_product_supported(int k_first);
int aslp_lstm_seq_first_product_supported_for(int k_first, int C);   /* ... in a layer of C cells (the staging row is 128 floats for C <= 128) */
/* Streams per chain the launch for these arguments uses: 8 (one 512-thread workgroup per CU).  grad_partial then has
 * ndir * ceil(streams / that) chains of 7 rows. */
int aslp_lstm_seq_chain_streams(const aslp_lstm_seq *a, int backward);
/* row blocks 0 and T+1 := 0 (all ld columns); columns [col0, col0 + ncols) of row blocks 1..T := 0xFFFFFFFF words.
 * For the forward kernel: col0 = the m column block (G + 2) * C, ncols = C. */
void aslp_lstm_seq_fill(float *buf, int ld, int T, int S, int col0, int ncols);
/* the same for the two directions' buffers of one layer (same ld, T, S) in one launch; init0 (S rows of ld_init floats, init_cols of them
 * meaningful; may be NULL): copied into row block 0 of buf0 -- the carried history of the forward direction -- instead of zeros */
void aslp_lstm_seq_fill_pair(float *buf0, float *buf1, int ld, int T, int S, int col0, int ncols, const float *init0, int ld_init, int init_cols);
void aslp_lstm_seq_forward(const aslp_lstm_seq *a);
/* diagnostics: hand-off re-polls (per wave) since the last reset, summed over all persistent launches; synchronises */
/* devtools: after aslp_lstm_seq_timing(3, NULL) (forward) or (4, NULL) (backward) every workgroup of such a launch records the
 * 100 MHz clock at entry and exit (a ring of the 8 latest persistent launches of either kind); out receives the (entry, exit) pairs of
 * workgroups 0 .. n-1 of the launch `launches_back` (0 = latest .. 7) persistent launches ago.  Synchronises. */
void aslp_lstm_seq_residency(unsigned long long *out, int n, int launches_back);
unsigned aslp_lstm_seq_polls(int reset);
/* diagnostics (devtools/bench_lc.py): phase timing of the forward kernel, see csrc/rnn_persistent.hip */
void aslp_lstm_seq_timing(int enable, unsigned long long *out);
void aslp_lstm_seq_backward(const aslp_lstm_seq *a);
/* After aslp_lstm_seq_backward with grad_partial set: bias / peephole gradients of direction `dir` from the per-chain sums (added in chain
 * order), then what aslp_rnn_vec_grads does with them: corr = grad + mmt * corr, clipped element-wise to [-clip, clip] if clip > 0, and
 * param += neg_lr * corr if neg_lr != 0.  bias_corr / bias: [G*C] in the buffer's gate order; peephole vectors [C] (peep_i_* NULL with cifg). */
void aslp_lstm_seq_vec_grads(const aslp_lstm_seq *a, int dir, float *bias_corr, float *bias, float *peep_i_corr, float *peep_i, float *peep_f_corr,
                             float *peep_f, float *peep_o_corr, float *peep_o, float mmt, float clip, float neg_lr);
/* both directions of a bidirectional layer in one launch: vec8_dirN = the eight pointers above in that order, for direction N */
void aslp_lstm_seq_vec_grads2(const aslp_lstm_seq *a, float *const *vec8_dir0, float *const *vec8_dir1, float mmt, float clip, float neg_lr);
/* GruStreams, the whole recurrence of T timesteps as ONE launch per pass (csrc/rnn_persistent.hip; the scheme of aslp_lstm_seq_*):
 * y / d: [(T + 2) * S x ld] activations / diffs, row block 0 = the carried history h(0) (forward), row blocks 0 and T + 1 of d zero,
 * columns [z|r|m|g|h], H each.  Before the launch the caller has stored
 *   forward:  the x-parts (+ bias) of z, r, m in columns [0, 3H) of row blocks 1..T, and aslp_lstm_seq_fill(y, ld, T, S, 3H, 2H)
 *             BEFORE that (g and h start as "not yet published");
 *   backward: aslp_lstm_seq_fill(d, ld, T, S, 0, 3H), then the loss's share of d_h in column block 4 of row blocks 1..T.
 * w_zr / w_m: forward W_zr_h [2H x H] and W_m_g [H x H]; backward their TRANSPOSES [H x 2H] and [H x H] (rows K-contiguous).
 * aslp_gru_seq_supported: S <= 64, H <= 512, H % 4 == 0, the grid co-resident; otherwise use aslp_gru_step_*. */
typedef struct aslp_gru_seq_ {
  float *y;
  float *d;            /* backward only */
  const float *w_zr;
  const float *w_m;
  int ldw_zr, ldw_m, ld, T, S, H;
  int s_begin, s_count;   /* stream window as in aslp_lstm_seq (s_count == 0: all S streams) */
} aslp_gru_seq;
int aslp_gru_seq_supported(const aslp_gru_seq *a, int backward);
void aslp_gru_seq_forward(const aslp_gru_seq *a);
void aslp_gru_seq_backward(const aslp_gru_seq *a);
/* GruStreams (nnet-gru-streams.h:275-303, 344-383), columns [z|r|m|g|h] */
/* GRU recurrence, one timestep, both dependent products fused with their gate arithmetic (csrc/gru_fused.hip):
 * forward = aslp_gru_forward1/2 with the two skinny GEMMs folded in; backward likewise, reading TRANSPOSED copies of the
 * recurrent matrices (w_zr_h_t [H x 2H], w_m_g_t [H x H]).  Needs H % 4 == 0 and 16-byte aligned, 4-float-pitched rows. */
int aslp_gru_step_supported(int H);
void aslp_gru_step_forward(float *y_cur, const float *y_prev, const float *w_zr_h, int ld_zr, const float *w_m_g, int ld_mg, int ld, int S, int H);
void aslp_gru_step_backward(float *d_cur, const float *d_next, const float *y_cur, const float *y_next, const float *y_prev, const float *w_zr_h_t,
                            int ld_zr_t, const float *w_m_g_t, int ld_mg_t, int ld, int S, int H, int has_next);
void aslp_gru_forward1(float *y_cur, const float *y_prev, int ld, int S, int H);
void aslp_gru_forward2(float *y_cur, const float *y_prev, int ld, int S, int H);
void aslp_gru_backward1(float *d_cur, const float *d_next, const float *y_cur, const float *y_next, int ld, int S, int H);
void aslp_gru_backward2(float *d_cur, const float *y_cur, const float *y_prev, int ld, int S, int H);
/* ---- depthwise temporal filters (csrc/temporal.hip) ------------------------------------------------
 * CompactFsmn (nnet-cfsmn-component.h:170-262), one sequence of T rows, coef [(past+future+1) x D]:
 *   reverse = 0 (Propagate):       out[t] = src[t] + sum_j coef[j]     .* src[t + j - past]
 *   reverse = 1 (in-diff, :232-249) out[t] = src[t] + sum_j coef[C-1-j] .* src[t + j - future]
 *   coef_grad (:213-219, 251-254): coef_corr[i] = clip(sum_t in[t + i - past] .* out_diff[t])  (overwrites) */
void aslp_fsmn_filter(float *out, int ldo, const float *src, int lds, const float *coef, int ldc, int D, int past, int future, int T,
                      int reverse);
void aslp_fsmn_coef_grad(float *coef_corr, int ldc, const float *in, int ldi, const float *out_diff, int ldod, int D, int past, int future,
                         int T, float clip);
/* The whole backward pass (BackpropagateFnc :204-256 + Update :258-262) in two launches: in_diff as reverse = 1 above and the tap
 * gradients of every chunk of frames from ONE staging of in / out_diff; then coef_corr as coef_grad above and -- with lr != 0 --
 * coef += -lr * coef_corr.  in_diff must not alias out_diff. */
void aslp_fsmn_backward(float *in_diff, int ldid, float *coef_corr, int ldcc, float *coef, int ldc, const float *in, int ldi, const float *out_diff,
                        int ldod, int D, int past, int future, int T, float clip, float lr);
/* RowConvolution (nnet-row-convolution.cc:105-176): rows t*S + s, w dense [D x (K+1)], seq_len [S] on the device.
 *   forward : out[t] = sum_k w[:,k] .* in[min(t + k, L_s - 1)]   for t < L_s, 0 beyond
 *   backward: in_diff[t] = sum_{k <= t} w[:,k] .* out_diff[t - k] for t < L_s, 0 beyond
 *   wgrad   : w_diff[d][k] = sum_{s, t < L_s} in[min(t + k, L_s - 1)][d] * out_diff[t][d]  (overwrites) */
void aslp_rowconv_forward(float *out, int ldo, const float *in, int ldi, const float *w, int D, int K, int T, int S,
                          const int32_cuda *seq_len);
void aslp_rowconv_backward(float *in_diff, int ldid, const float *out_diff, int ldod, const float *w, int D, int K, int T, int S,
                           const int32_cuda *seq_len);
void aslp_rowconv_wgrad(float *w_diff, const float *in, int ldi, const float *out_diff, int ldod, int D, int K, int T, int S,
                        const int32_cuda *seq_len);
/* backward + wgrad in ONE pass over in / out_diff (each read once, in_diff written once) and a small launch that adds the tap partials in a
 * fixed order; with update != 0 that launch also takes the step of Update (:178-186): w_corr = momentum w_corr + w_diff, w -= learn_rate w_corr */
void aslp_rowconv_backward_fused(float *in_diff, int ldid, float *w_diff, const float *in, int ldi, const float *out_diff, int ldod, float *w, int D,
                                 int K, int T, int S, const int32_cuda *seq_len, float *w_corr, float momentum, float learn_rate, int update);
/* ---- Eesen CTC rows (cu-kernels-ansi.h:366-394; device code cu-kernels.cu:3276-3534) -----------------------
 * One lattice row per launch, log domain with log_zero = -1e30 (ctc-utils.h:28-95); `prob` holds LOG
 * probabilities for alpha/beta and probabilities for error; labels are blank-augmented, -1 padded. */
void cudaF_compute_ctc_alpha(aslp_dim3 Gr, aslp_dim3 Bl, float *alpha, int row_idx, MatrixDim dim_alpha, const float *prob, MatrixDim dim_prob,
                             const int *labels);
void cudaF_compute_ctc_beta(aslp_dim3 Gr, aslp_dim3 Bl, float *beta, int row_idx, MatrixDim dim_beta, const float *prob, MatrixDim dim_prob,
                            const int *labels);
void cudaF_compute_ctc_error(aslp_dim3 Gr, aslp_dim3 Bl, float *error, MatrixDim dim_error, const float *alpha, const float *beta,
                             MatrixDim dim_alpha, const float *prob, const int *labels, float pzx);
void cudaF_compute_ctc_alpha_multiple_sequence(aslp_dim3 Gr, aslp_dim3 Bl, float *alpha, int seq_num, int row_idx, MatrixDim dim_alpha,
                                               const float *prob, MatrixDim dim_prob, const int *labels, int dim_label_stride,
                                               const int *seq_lengths);
void cudaF_compute_ctc_beta_multiple_sequence(aslp_dim3 Gr, aslp_dim3 Bl, float *beta, int seq_num, int row_idx, MatrixDim dim_beta,
                                              const float *prob, MatrixDim dim_prob, const int *labels, int dim_label_stride,
                                              const int *seq_lengths, const int *label_lengths);
void cudaF_compute_ctc_error_multiple_sequence(aslp_dim3 Gr, aslp_dim3 Bl, float *error, int seq_num, MatrixDim dim_error, const float *alpha,
                                               const float *beta, MatrixDim dim_alpha, const float *prob, const int *labels,
                                               int dim_label_stride, const int *seq_lengths, const float *pzx);
/* max-norm row shrink (nnet-affine-transform.h:231-243) */
void aslp_max_norm_rows(float *W, MatrixDim d, float max_norm);

#ifdef __cplusplus
}
#endif
#endif
