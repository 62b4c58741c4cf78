// ctc_eesen.hip -- the per-row Eesen CTC entry points of the kernel ABI (B1):
// cudaF_compute_ctc_{alpha,beta,error}[_multiple_sequence]  (aslp-cudamatrix/cu-kernels-ansi.h:366-394,
// device code cu-kernels.cu:3276-3534, log-domain conventions ctc-utils.h:52-95: log_zero = -1e30).
// They exist so a relinked CuMatrix::ComputeCtc* keeps working call for call; the host engine's
// Ctc loss does NOT use them -- it runs the whole lattice in two launches (ctc.hip,
// aslp_eesen_ctc_mseq) instead of 2T of these plus an O(T*A*(2L+1)) error kernel.
#include "aslp_kernels.h"
#include "common.h"

namespace aslp {
namespace {

constexpr float kLogZero = -1e30f, kLogInf = 1e30f, kExpLimit = 88.722839f, kMax = 3.4028235e+038f;
__device__ __forceinline__ float add_ab(float a, float b) { return (a == kLogZero || b == kLogZero) ? kLogZero : a + b; }
__device__ __forceinline__ float sub_ab(float a, float b) { return a == kLogZero ? kLogZero : (b == kLogZero ? kLogInf : a - b); }
__device__ __forceinline__ float exp_a(float a) { return a <= kLogZero ? 0.0f : (a >= kExpLimit ? kMax : expf(a)); }
__device__ __forceinline__ float log_a_plus_b(float a, float b) {
  const float hi = b < a ? a : b, lo = b < a ? b : a;
  return add_ab(hi, logf(1 + exp_a(sub_ab(lo, hi))));
}

// one lattice row: thread per (sequence, state).  lab: this sequence's blank-augmented labels (-1 padding).
// DIR = +1: alpha (reads row-1, states j, j-1, j-2); DIR = -1: beta (reads row+1, states j, j+1, j+2).
template <int DIR>
__global__ void __launch_bounds__(kBlock) ctc_row_kernel(float *lat, int seq_num, int row, MatrixDim dl, const float *prob, MatrixDim dp,
                                                         const int *labels, int lab_stride, const int *seq_lengths, const int *label_lengths,
                                                         int single_rows) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= seq_num * dl.cols) return;
  const int i = idx / dl.cols, j = idx - i * dl.cols;
  float *cell = lat + (long)(row * seq_num + i) * dl.stride + j;
  const int cls = labels[i * lab_stride + j];
  const int rows_i = seq_lengths ? seq_lengths[i] : single_rows;
  if (cls == -1 || row >= rows_i) { *cell = kLogZero; return; }
  const int width = label_lengths ? label_lengths[i] : dl.cols;  // states of this sequence (beta boundary)
  const float p = prob[(long)(row * seq_num + i) * dp.stride + cls];
  const float *other = lat + (long)((row - DIR) * seq_num + i) * dl.stride + j;
  const bool first = DIR > 0 ? row == 0 : row == rows_i - 1;
  if (first) {
    const bool start_state = DIR > 0 ? j < 2 : j > width - 3;
    *cell = start_state ? p : kLogZero;
    return;
  }
  const int dist = DIR > 0 ? j : width - 1 - j;  // how many neighbour states exist on the incoming side
  if (dist >= 2) {
    const int cls2 = labels[i * lab_stride + j - 2 * DIR];
    const float two = log_a_plus_b(other[-DIR], other[0]);
    if (j % 2 == 0 || cls2 == cls) *cell = add_ab(p, two);
    else *cell = add_ab(p, log_a_plus_b(other[-2 * DIR], two));
  } else if (dist == 1) {
    *cell = add_ab(p, log_a_plus_b(other[-DIR], other[0]));
  } else {
    *cell = add_ab(p, other[0]);
  }
}

// error[r][k] = -exp( logsum_{j: lab[j]==k} (alpha+beta)[r][j] - (pzx + 2 log y[r][k]) ); thread per (row, class)
__global__ void __launch_bounds__(kBlock) ctc_error_kernel(float *err, int seq_num, MatrixDim de, const float *alpha, const float *beta, MatrixDim da,
                                                           const float *prob, const int *labels, int lab_stride, const int *seq_lengths,
                                                           const float *pzx_dev, float pzx_scalar) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)de.rows * de.cols) return;
  const int r = idx / de.cols, k = idx - (long)r * de.cols;
  const int seq = r % seq_num, t = r / seq_num;
  if (seq_lengths && t >= seq_lengths[seq]) return;
  float e = kLogZero;
  for (int s = 0; s < da.cols; s++) {
    const int l = labels[seq * lab_stride + s];
    if (l == k) e = log_a_plus_b(e, add_ab(alpha[(long)r * da.stride + s], beta[(long)r * da.stride + s]));
  }
  const float y = prob[(long)r * de.stride + k];
  const float z = pzx_dev ? pzx_dev[seq] : pzx_scalar;
  err[(long)r * de.stride + k] = -1.0f * exp_a(sub_ab(e, add_ab(z, y == 0 ? kLogZero : 2 * logf(y))));
}

template <int DIR>
void launch_row(float *lat, int seq_num, int row, MatrixDim dl, const float *prob, MatrixDim dp, const int *labels, int lab_stride,
                const int *seq_lengths, const int *label_lengths, const char *name) {
  const int n = seq_num * dl.cols;
  if (n <= 0) return;
  hipLaunchKernelGGL(ctc_row_kernel<DIR>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, cur_stream(), lat, seq_num, row, dl, prob, dp, labels,
                     lab_stride, seq_lengths, label_lengths, dl.rows);
  check_launch(name);
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

void cudaF_compute_ctc_alpha(aslp_dim3, aslp_dim3, float *alpha, int row_idx, MatrixDim dim_alpha, const float *prob, MatrixDim dim_prob,
                             const int *labels) {
  launch_row<1>(alpha, 1, row_idx, dim_alpha, prob, dim_prob, labels, dim_alpha.cols, nullptr, nullptr, "cudaF_compute_ctc_alpha");
}
void cudaF_compute_ctc_beta(aslp_dim3, aslp_dim3, float *beta, int row_idx, MatrixDim dim_beta, const float *prob, MatrixDim dim_prob,
                            const int *labels) {
  launch_row<-1>(beta, 1, row_idx, dim_beta, prob, dim_prob, labels, dim_beta.cols, nullptr, nullptr, "cudaF_compute_ctc_beta");
}
void cudaF_compute_ctc_error(aslp_dim3, aslp_dim3, float *error, MatrixDim dim_error, const float *alpha, const float *beta, MatrixDim dim_alpha,
                             const float *prob, const int *labels, float pzx) {
  const long n = (long)dim_error.rows * dim_error.cols;
  if (n <= 0) return;
  hipLaunchKernelGGL(ctc_error_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, cur_stream(), error, 1, dim_error, alpha, beta, dim_alpha,
                     prob, labels, dim_alpha.cols, nullptr, nullptr, pzx);
  check_launch("cudaF_compute_ctc_error");
}
void cudaF_compute_ctc_alpha_multiple_sequence(aslp_dim3, aslp_dim3, float *alpha, int seq_num, int row_idx, MatrixDim dim_alpha, const float *prob,
                                               MatrixDim dim_prob, const int *labels, int dim_label_stride, const int *seq_lengths) {
  launch_row<1>(alpha, seq_num, row_idx, dim_alpha, prob, dim_prob, labels, dim_label_stride, seq_lengths, nullptr,
                "cudaF_compute_ctc_alpha_multiple_sequence");
}
void cudaF_compute_ctc_beta_multiple_sequence(aslp_dim3, aslp_dim3, float *beta, int seq_num, int row_idx, MatrixDim dim_beta, const float *prob,
                                              MatrixDim dim_prob, const int *labels, int dim_label_stride, const int *seq_lengths,
                                              const int *label_lengths) {
  launch_row<-1>(beta, seq_num, row_idx, dim_beta, prob, dim_prob, labels, dim_label_stride, seq_lengths, label_lengths,
                 "cudaF_compute_ctc_beta_multiple_sequence");
}
void cudaF_compute_ctc_error_multiple_sequence(aslp_dim3, aslp_dim3, float *error, int seq_num, MatrixDim dim_error, const float *alpha,
                                               const float *beta, MatrixDim dim_alpha, const float *prob, const int *labels, int dim_label_stride,
                                               const int *seq_lengths, const float *pzx) {
  const long n = (long)dim_error.rows * dim_error.cols;
  if (n <= 0) return;
  hipLaunchKernelGGL(ctc_error_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, cur_stream(), error, seq_num, dim_error, alpha, beta,
                     dim_alpha, prob, labels, dim_label_stride, seq_lengths, pzx, 0.0f);
  check_launch("cudaF_compute_ctc_error_multiple_sequence");
}

}  // extern "C"
