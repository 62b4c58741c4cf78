/*
 * aslp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C CPU restatement of the reference's (robin1001/kaldi-aslp) CPU
 * ("matrix/ + BLAS") path for the aslp-nnet training step.  It exists so the
 * HIP product path can be checked against it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link / load this library; the product
 * (kaldi-aslp_amd/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - CTC (orc_ctc_*): pinned against the reference's own Warp-CTC CPU code built
 *     from /root/reference into oracle/_ref/ and against the known-answer vectors of
 *     src/warp-ctc/tests/test_cpu.cpp.
 *   - substrate ops: pinned against outputs of the reference's own CuMatrix CPU branch, built where it lies:
 *     tests/golden/cumatrix_ops.bin (BLAS-free operations) and tests/golden/cumatrix_blas_ops.bin (the BLAS-backed ones, linked
 *     against the OpenBLAS inside the image's scipy wheel: product, softmax, column / row sums, wide broadcasts).
 *   - Component level: AffineTransform, BatchNormalization, every LSTM gate block (LstmProjectedStreams, both directions of
 *     BLstmProjectedStreamsLC, LstmCifgProjectedStreams, Lstm / BLstm), GruStreams, RowConvolution and CompactFsmn are pinned
 *     against their op sequences issued on the reference's library (oracle/gen_cumatrix_blas_golden.cpp,
 *     tests/test_oracle_ref_blas_cpu.py).  The component headers themselves need OpenFst and cannot be compiled here, so
 *     control flow that lives only there (sequence-length masking, chunk / stream-reset bookkeeping, Xent's bookkeeping) stays
 *     restated from source, cited per fn: PARITY UNPINNED for those parts.
 *
 * All matrices are row-major float with an explicit leading dimension (stride) in
 * elements; "rows = frames".  Each function cites the reference file:line it follows
 * (paths relative to /root/reference/src).
 */
#ifndef ASLP_ORACLE_H_
#define ASLP_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- substrate (aslp-cudamatrix CPU branches -> matrix/) ------------------------- */

/* C = alpha*op(A)*op(B) + beta*C.  cu-matrix.cc:1027-1061 -> cblas_sgemm.
 * M,N = dims of C; K = inner dim; transX != 0 means op(X) = X^T. */
void orc_add_mat_mat(float *C, int M, int N, int ldc, float alpha,
                     const float *A, int lda, int transA,
                     const float *B, int ldb, int transB, int K, float beta);
void orc_set_num_threads(int n);   /* OpenMP threads used by orc_add_mat_mat */
int  orc_get_num_threads(void);

void orc_sigmoid(float *y, int ldy, const float *x, int ldx, int rows, int cols);      /* matrix/kaldi-vector.cc:923-936 */
void orc_tanh(float *y, int ldy, const float *x, int ldx, int rows, int cols);         /* matrix/kaldi-vector.cc:885-898 */
void orc_diff_sigmoid(float *eout, int ldo, const float *y, int ldy, const float *e, int lde, int rows, int cols); /* cu-matrix.cc:1399 -> eout = y(1-y)e */
void orc_diff_tanh(float *eout, int ldo, const float *y, int ldy, const float *e, int lde, int rows, int cols);    /* cu-matrix.cc:1445 -> eout = (1-y^2)e */
void orc_softmax_rows(float *y, int ldy, const float *x, int ldx, int rows, int cols); /* cu-matrix.cc:1351-1371, kaldi-vector.cc:852-859 */
void orc_find_row_max_id(const float *m, int ld, int rows, int cols, int32_t *id);     /* cu-matrix.cc:1466-1510 */
void orc_splice(float *y, int ldy, const float *x, int ldx, int rows, int in_cols,
                const int32_t *offsets, int n_off);                                    /* cu-math.cc:132-168 */
void orc_copy_cols(float *y, int ldy, const float *x, int ldx, int rows,
                   const int32_t *copy_from, int out_cols);                            /* cu-math.cc:173-210 (cu::Copy) */
void orc_randomize(float *y, int ldy, const float *x, int ldx, int cols,
                   const int32_t *copy_from, int n_idx);                               /* cu-math.cc:80-127 */
void orc_add_row_sum_mat(float *dst, int ldd, int dst_rows, int cols, const float *src,
                         int lds, int src_rows, float alpha, float beta);              /* cu-matrix.cc:3010-3034 (ASLP) */
void orc_add_conv_mat_mat_elements(float *dst, int ldd, int cols, const float *A, int lda,
                                   int a_rows, const float *B, int ldb, int b_rows,
                                   float alpha, float beta);                           /* cu-matrix.cc:3037-3073 (ASLP) */
void orc_regularize_l1(float *w, int ldw, float *g, int ldg, int rows, int cols, float l1, float lr); /* cu-math.cc:37-75 */
/* element-wise / broadcast arithmetic the components are composed from (all pinned by tests/golden/cumatrix_ops.bin) */
void orc_add_mat_mat_elements(float *dst, int ldd, const float *A, int lda, const float *B, int ldb, int rows, int cols,
                              float alpha, float beta);                                /* kaldi-matrix.cc:483-501 */
void orc_add_mat_diag_vec(float *dst, int ldd, const float *M, int m_row_stride, int m_col_stride, const float *v,
                          int rows, int cols, float alpha);                            /* kaldi-matrix.cc:448-480 (beta = 1) */
void orc_add_vec_to_rows(float *dst, int ldd, const float *row, int rows, int cols, float alpha);  /* kaldi-matrix.cc:2749-2766 */
void orc_add_vec_to_cols(float *dst, int ldd, const float *col, int rows, int cols, float alpha);  /* kaldi-matrix.cc:2780-2799 */
void orc_mul_cols_vec(float *dst, int ldd, const float *scale, int rows, int cols);    /* kaldi-matrix.cc:1141-1150 */
void orc_mul_rows_vec(float *dst, int ldd, const float *scale, int rows, int cols);    /* kaldi-matrix.cc:1057-1068 */
void orc_copy_cols_idx(float *dst, int ldd, const float *src, int lds, int rows, const int32_t *idx, int n_idx);  /* kaldi-matrix.cc:2561-2584 */
void orc_add_cols_idx(float *dst, int ldd, const float *src, int lds, int rows, const int32_t *idx, int n_idx);   /* kaldi-matrix.cc:2587-2610 */

/* ---- components (aslp-nnet) ------------------------------------------------------- */

/* AffineTransform, nnet-affine-transform.h:186-245 */
void orc_affine_propagate(float *out, int ldo, const float *in, int ldi, int rows,
                          const float *W, int ldw, const float *bias, int in_dim, int out_dim);
void orc_affine_backpropagate(float *in_diff, int ldid, const float *out_diff, int ldod, int rows,
                              const float *W, int ldw, int in_dim, int out_dim);
typedef struct {
  float learn_rate, momentum, l2_penalty, l1_penalty;       /* nnet-trnopts.h:29-47 */
  float learn_rate_coef, bias_learn_rate_coef, max_norm;    /* nnet-affine-transform.h:37-40 */
} orc_affine_opts;
void orc_affine_update(float *W, int ldw, float *bias, float *W_corr, int ldc, float *bias_corr,
                       const float *input, int ldi, const float *diff, int ldd, int rows,
                       int in_dim, int out_dim, const orc_affine_opts *o);

/* ReLU, nnet-activation.h:281-298 */
void orc_relu(float *y, int ldy, const float *x, int ldx, int rows, int cols);
void orc_diff_relu(float *in_diff, int ldo, const float *in, int ldi, const float *out_diff, int lde, int rows, int cols);

/* Splice backward quirk (gather with +offset, not the adjoint), nnet-various.h:143-175 */
void orc_splice_backpropagate(float *in_diff, int ldid, const float *out_diff, int ldod, int rows,
                              int in_cols, const int32_t *offsets, int n_off);

/* BatchNormalization, nnet-batch-normalization.h:139-284.
 * State that the component keeps between calls is passed explicitly. */
typedef struct {
  int dim;
  float *scale, *shift;          /* gamma, beta                                   */
  float *dscale, *dshift;        /* momentum-carrying gradient buffers             */
  float *mean_vec, *var_vec;     /* batch mean; 1/sqrt(var+1e-7) ("var_vec_")      */
  double *acc_means, *acc_vars;  /* running sums (double), :218-220                */
  double num_acc_frames;
  int acc_cleaned;               /* :178-181 first Propagate of the process clears */
} orc_bn_state;
void orc_bn_propagate(orc_bn_state *s, float *out, int ldo, const float *in, int ldi, int rows,
                      float *xsharp /* rows x dim, ld = dim: XsharpO_ */);
void orc_bn_backpropagate(orc_bn_state *s, float *in_diff, int ldid, const float *in, int ldi,
                          const float *out_diff, int ldod, int rows, float momentum,
                          float *xsharp /* in: xhat from propagate; out: dy*gamma (:241-242) */);
void orc_bn_update(orc_bn_state *s, float learn_rate);                      /* :280-284 */
/* inference path with global stats, :139-175 + ReadData :56-94 */
void orc_bn_feedforward(orc_bn_state *s, float *out, int ldo, const float *in, int ldi, int rows);
void orc_bn_global_stats_from_acc(orc_bn_state *s);                         /* ReadData :73-93 */

/* Xent::Eval, nnet-loss.cc:63-156.  targets dense [rows x cols]. */
typedef struct {
  double frames, correct, loss, entropy, likelyhood;   /* per-call increments */
} orc_xent_stats;
void orc_xent_eval(const float *frame_weights, const float *net_out, int ldn, const float *targets,
                   int ldt, int rows, int cols, float *diff, int ldd, orc_xent_stats *st);
/* Mse::Eval, nnet-loss.cc:205-258 */
void orc_mse_eval(const float *frame_weights, const float *net_out, int ldn, const float *targets,
                  int ldt, int rows, int cols, float *diff, int ldd, double *loss, double *frames);
/* MultiTaskLoss::Eval, nnet-loss.cc:341-368 ('multitask,<type>,<dim>,<weight>,...', :296-339).  kinds[i]: 0 = xent, 1 = mse; task i owns
 * columns [sum dims[<i], + dims[i]) of net_out / targets / diff; per-call increments of task i go to xent_st[i] or mse_loss[i] /
 * mse_frames[i] (the other kind's slot is left alone); diff blocks are scaled by weights[i]. */
void orc_multitask_eval(int n_tasks, const int *kinds, const int *dims, const float *weights, const float *frame_weights,
                        const float *net_out, int ldn, const float *targets, int ldt, int rows, float *diff, int ldd,
                        orc_xent_stats *xent_st, double *mse_loss, double *mse_frames);

/* ---- front-end components of the CNN / cFSMN recipes (aslp_oracle_conv.c) ------------------------------------------- */
/* LinearTransform, nnet-linear-transform.h:127-160 (orc_affine_opts: learn_rate, momentum, l2, l1, learn_rate_coef are used) */
void orc_linear_propagate(float *out, int ldo, const float *in, int ldi, int rows, const float *W, int ldw, int in_dim, int out_dim);
void orc_linear_backpropagate(float *in_diff, int ldid, const float *out_diff, int ldod, int rows, const float *W, int ldw, int in_dim, int out_dim);
void orc_linear_update(float *W, int ldw, float *W_corr, int ldc, const float *input, int ldi, const float *diff, int ldd, int rows, int in_dim,
                       int out_dim, const orc_affine_opts *o);
/* ConvolutionalComponent, nnet-convolutional-component.h:268-470.  patches / patch_diffs: [rows x filter_dim * num_patches], ld = that width
 * (vectorized_feature_patches_, feature_patch_diffs_); filters [num_filters x filter_dim]; filters_grad dense (ld = filter_dim) */
void orc_conv_propagate(float *out, int ldo, float *patches, const float *in, int ldi, int rows, int in_dim, const float *filters, int ldf,
                        const float *bias, int num_filters, int patch_dim, int patch_step, int patch_stride);
void orc_conv_backpropagate(float *in_diff, int ldid, float *patch_diffs, const float *out_diff, int ldod, int rows, int in_dim, const float *filters,
                            int ldf, int num_filters, int patch_dim, int patch_step, int patch_stride);
void orc_conv_update(float *filters, int ldf, float *bias, float *filters_grad, float *bias_grad, const float *patches, const float *diff, int ldd,
                     int rows, int in_dim, int num_filters, int patch_dim, int patch_step, int patch_stride, float learn_rate, float learn_rate_coef,
                     float bias_learn_rate_coef, float max_norm);
/* MaxPoolingComponent, nnet-max-pooling-component.h:101-162 */
void orc_max_pool_propagate(float *out, int ldo, const float *in, int ldi, int rows, int in_dim, int pool_size, int pool_step, int pool_stride);
void orc_max_pool_backpropagate(float *in_diff, int ldid, const float *in, int ldi, const float *out, int ldo, const float *out_diff, int ldod, int rows,
                                int in_dim, int pool_size, int pool_step, int pool_stride);
/* LengthNormComponent, nnet-various.h:338-358 */
void orc_length_norm_propagate(float *out, int ldo, float *row_scales, const float *in, int ldi, int rows, int cols);
void orc_length_norm_backpropagate(float *in_diff, int ldid, const float *out_diff, int ldod, const float *row_scales, int rows, int cols);
/* GroupPnorm / GroupPnormDeriv / GroupMax / GroupMaxDeriv / MulRowsGroupMat, kaldi-matrix.cc:1071-1138, 2530-2558 */
void orc_group_pnorm(float *y, int ldy, const float *x, int ldx, int rows, int out_cols, int group, float power);
void orc_group_pnorm_deriv(float *d, int ldd, const float *in, int ldi, const float *out, int ldo, int rows, int in_cols, int group, float power);
void orc_group_max(float *y, int ldy, const float *x, int ldx, int rows, int out_cols, int group);
void orc_group_max_deriv(float *d, int ldd, const float *in, int ldi, const float *out, int ldo, int rows, int in_cols, int group);
void orc_mul_rows_group_mat(float *y, int ldy, const float *src, int lds, int rows, int cols, int group);

/* ---- whole DNN train step for bench.py's cpu_baseline ("port") -------------------- */
/* Chain: [Affine (+BN) + Sigmoid] x n_hidden, Affine, Softmax, Xent, backward with
 * immediate Update in reference order (nnet-nnet.cc:70-154).  Buffers are owned by
 * the caller through an opaque handle. */
typedef struct orc_dnn orc_dnn;
orc_dnn *orc_dnn_create(int in_dim, int hidden_dim, int n_hidden, int out_dim, int with_bn,
                        int minibatch, unsigned seed);
void orc_dnn_destroy(orc_dnn *d);
/* one step on a resident minibatch; returns cross-entropy of the batch */
double orc_dnn_train_step(orc_dnn *d, const float *in, const int32_t *labels, float learn_rate,
                          float momentum);
/* accessors so tests can compare the C chain with the HIP engine on shared weights */
int  orc_dnn_num_layers(const orc_dnn *d);                 /* affine layers */
float *orc_dnn_weight(orc_dnn *d, int layer, int *rows, int *cols);
float *orc_dnn_bias(orc_dnn *d, int layer);
float *orc_dnn_bn_scale(orc_dnn *d, int layer);
float *orc_dnn_bn_shift(orc_dnn *d, int layer);
const float *orc_dnn_output(const orc_dnn *d);             /* softmax output [mb x out] */

/* The golden generators' uniform generator (oracle/gen_cumatrix_blas_golden.cpp Uniform(): xorshift64 13 / 7 / 17, the top 24 bits): out[i] =
 * lo + (hi - lo) * u, advancing *state.  A fixture that records the generator state in front of a tensor instead of the tensor itself
 * (tests/golden/lstm_fullwidth.bin) is read back with this. */
void orc_golden_uniform_fill(unsigned long long *state, float *out, long n, float lo, float hi);

/* ---- recurrent components (aslp_oracle_rnn.c) ---------------------------------------------- */
typedef struct {   /* one direction of an LSTM-family component; all matrices dense row-major */
  int D, C, R;     /* input dim, cells, projection dim (0 = no projection: Lstm / BLstm) */
  int cifg;        /* LstmCifgProjectedStreams: gates g,f,o only */
  float *w_x;      /* [G*C x D]   w_gifo_x_ */
  float *w_r;      /* [G*C x rec] w_gifo_r_, rec = R or C */
  float *bias;     /* [G*C] */
  float *peep_i, *peep_f, *peep_o; /* [C] each (peep_i unused for cifg) */
  float *w_rm;     /* [R x C] w_r_m_ (NULL without projection) */
} orc_lstm_dir;
int orc_lstm_width(const orc_lstm_dir *p);
void orc_lstm_forward(const orc_lstm_dir *p, const float *in, int ldi, int T, int S, int reverse, const float *init_state,
                      const int32_t *seq_len, float *buf);
void orc_lstm_backward(const orc_lstm_dir *p, const float *out_diff, int ldo, int T, int S, int reverse, const float *buf,
                       float *dbuf, float *in_diff, int ldid, float in_diff_beta);
void orc_lstm_grads(const orc_lstm_dir *p, orc_lstm_dir *g, const float *in, int ldi, int T, int S, int reverse, const float *buf,
                    const float *dbuf, float mmt, float clip);
void orc_lstm_update(orc_lstm_dir *p, const orc_lstm_dir *g, float lr);
typedef struct {
  int D, H;
  float *w_zrm_x; /* [3H x D] */
  float *w_zr_h;  /* [2H x H] */
  float *w_m_g;   /* [H x H]  */
  float *bias;    /* [3H]     */
} orc_gru;
void orc_gru_forward(const orc_gru *p, const float *in, int ldi, int T, int S, const float *init_state, float *buf);
void orc_gru_backward(const orc_gru *p, const float *out_diff, int ldo, int T, int S, const float *buf, float *dbuf, float *in_diff, int ldid);
void orc_gru_grads(const orc_gru *p, orc_gru *g, const float *in, int ldi, int T, int S, const float *buf, const float *dbuf, float mmt, float clip);
void orc_gru_update(orc_gru *p, const orc_gru *g, float lr);

/* ---- depthwise temporal components (aslp_oracle_temporal.c) ------------------------------------- */
void orc_rowconv_propagate(const float *w, int D, int K, const float *in, int ldi, int T, int S, const int32_t *seq_len,
                           float *in_buf, float *out, int ldo);
void orc_rowconv_backpropagate(const float *w, int D, int K, const float *out_diff, int ldod, int T, int S, const int32_t *seq_len,
                               const float *in_buf, float *in_diff_buf, float *w_diff, float *in_diff, int ldid);
void orc_rowconv_update(float *w, float *w_corr, const float *w_diff, int D, int K, float lr, float mmt);
void orc_fsmn_propagate(const float *coef, int D, int P, int F, const float *in, int ldi, int T, float *out, int ldo);
void orc_fsmn_backpropagate(const float *coef, int D, int P, int F, const float *in, int ldi, const float *out_diff, int ldod, int T,
                            float clip, float *coef_corr, float *in_diff, int ldid);
void orc_fsmn_update(float *coef, const float *coef_corr, int D, int P, int F, float lr);

/* ---- Warp-CTC (src/warp-ctc/include/detail/cpu_ctc.h) + the WarpCtc wrapper (aslp-nnet/warp-ctc.cc) */
int orc_ctc_cost_and_grad(const float *acts, float *grads, const int *flat_labels, const int *label_lengths,
                          const int *input_lengths, int alphabet_size, int minibatch, float *costs);
typedef struct {   /* members of WarpCtc, warp-ctc.h:96-113 */
  double loss_sum, loss_square_sum, loss_sum_bak, loss_square_sum_bak, obj;
  int normal_num, stat_period /* 500 */, frames, sequences;
} orc_ctc_filter_state;
void orc_ctc_loss_filter(const float *costs, const int *frame_num, int mb, orc_ctc_filter_state *st, int *keep);
void orc_eesen_ctc_mseq(const float *net_out, int ld, int T, int S, int A, const int *flat_labels, const int *label_lens,
                        const int *frame_num, float *diff, int ldd, float *pzx);
void orc_eesen_ctc_loss_filter(const float *costs, const int *frame_num, int mb, orc_ctc_filter_state *st, int *keep);
int orc_ctc_token_errors(const float *net_out, int ld, int T, int A, const int *ref, int ref_len, int *hyp_len);

#ifdef __cplusplus
}
#endif
#endif
