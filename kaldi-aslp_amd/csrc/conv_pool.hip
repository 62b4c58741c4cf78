// conv_pool.hip -- the front-end components of the CNN / cFSMN recipes on gfx950: ConvolutionalComponent's patch gather and
// in-diff gather-sum, MaxPoolingComponent, LengthNormComponent, the group p-norm / group max pair (Pnorm, Maxout), and the
// element-wise entry points of the kernel ABI they replace (cu-kernels-ansi.h:85-90,146-147,182).
//
// All of these are HBM-bound index / element-wise passes: one thread per output element (or float4 of elements), rows contiguous
// across the lanes of a wave so every access is coalesced; nothing is staged (each input element is read once or, for overlapping
// patches / pools, re-read from L2 by a neighbouring lane of the same wave).  The products of the convolution are the library's
// GEMM (csrc/gemm_glds.hip): see nnet/nnet-conv.h for how the reference's per-patch products become ONE product per pass.
#include "aslp_kernels.h"
#include "common.h"

namespace aslp {
namespace {

// ---- ConvolutionalComponent (nnet-convolutional-component.h:301-331) -------------------------------------------------------
// patches[(n * P + p)][s * patch_dim + d] = in[n][p * step + s * stride + d]        (the reference's column map, :318-325, with the
// vectorised patches of a frame laid out as P consecutive ROWS of filter_dim instead of P column blocks: the P per-patch products
// [N x filter_dim] x filters^T then are one product [(N P) x filter_dim] x filters^T whose output rows ARE the reference's
// out[n][p * num_filters + f] when the output rows are unpadded)
__global__ void __launch_bounds__(kBlock) conv_gather_kernel(float *patches, int ldp, const float *in, int ld_in, int N, int P, int num_splice,
                                                             int patch_dim, int patch_step, int patch_stride) {
  const int filter_dim = num_splice * patch_dim;
  const long total = (long)N * P * filter_dim;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
    const int c = (int)(i % filter_dim);
    const long row = i / filter_dim;
    const int p = (int)(row % P);
    const long n = row / P;
    const int s = c / patch_dim, d = c - s * patch_dim;
    patches[row * ldp + c] = in[n * ld_in + p * patch_step + s * patch_stride + d];
  }
}
// in_diff[n][s * stride + q] = sum over the patches p that contain position q (ascending p, fp32, starting from the first term) of
// patch_diff[(n * P + p)][s * patch_dim + q - p * step]: what the reference's AddCols passes over the reversed column map add up
// (:413-421; AddCols k adds the k-th smallest j with column_map[j] == c, and j grows with p), in the same order.
__global__ void __launch_bounds__(kBlock) conv_indiff_kernel(float *in_diff, int ld_id, const float *patch_diff, int ldp, int N, int P, int num_splice,
                                                             int patch_dim, int patch_step, int patch_stride) {
  const int in_dim = num_splice * patch_stride;
  const long total = (long)N * in_dim;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
    const int c = (int)(i % in_dim);
    const long n = i / in_dim;
    const int s = c / patch_stride, q = c - s * patch_stride;
    // p * step <= q < p * step + patch_dim
    int p_lo = q - patch_dim + 1;
    p_lo = p_lo <= 0 ? 0 : (p_lo + patch_step - 1) / patch_step;
    int p_hi = q / patch_step;
    if (p_hi > P - 1) p_hi = P - 1;
    float acc = 0.f;
    const float *pd = patch_diff + (n * P) * (long)ldp + s * patch_dim;
    for (int p = p_lo; p <= p_hi; p++) acc += pd[(long)p * ldp + (q - p * patch_step)];
    in_diff[n * ld_id + c] = acc;
  }
}

// ---- MaxPoolingComponent (nnet-max-pooling-component.h:101-162) -------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) maxpool_fwd_kernel(float *out, int ld_out, const float *in, int ld_in, int N, int num_pools, int pool_size,
                                                             int pool_step, int pool_stride) {
  const int out_dim = num_pools * pool_stride;
  const long total = (long)N * out_dim;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
    const int c = (int)(i % out_dim);
    const long n = i / out_dim;
    const int q = c / pool_stride, k = c - q * pool_stride;
    const float *src = in + n * ld_in + (long)(q * pool_step) * pool_stride + k;
    float m = -1e20f;   // :110 "reset (large negative value)"
    for (int r = 0; r < pool_size; r++) m = fmaxf(m, src[(long)r * pool_stride]);
    out[n * ld_out + c] = m;
  }
}
// in_diff[n][p * stride + k] = (sum over the pools q that contain patch p, ascending q, of out_diff[n][q * stride + k] where the input
// equals that pool's maximum) * (1 / #pools containing p)       (:130-160: mask by EqualElementMask, AddMat per (q, r), then Scale)
__global__ void __launch_bounds__(kBlock) maxpool_bwd_kernel(float *in_diff, int ld_id, const float *in, int ld_in, const float *out, int ld_out,
                                                             const float *out_diff, int ld_od, int N, int num_patches, int num_pools, int pool_size,
                                                             int pool_step, int pool_stride) {
  const int in_dim = num_patches * pool_stride;
  const long total = (long)N * in_dim;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
    const int c = (int)(i % in_dim);
    const long n = i / in_dim;
    const int p = c / pool_stride, k = c - p * pool_stride;
    int q_lo = p - pool_size + 1;
    q_lo = q_lo <= 0 ? 0 : (q_lo + pool_step - 1) / pool_step;
    int q_hi = p / pool_step;
    if (q_hi > num_pools - 1) q_hi = num_pools - 1;
    const float x = in[n * ld_in + c];
    float acc = 0.f;
    int summands = 0;
    for (int q = q_lo; q <= q_hi; q++) {
      const float mask = x == out[n * ld_out + (long)q * pool_stride + k] ? 1.0f : 0.0f;
      acc += out_diff[n * ld_od + (long)q * pool_stride + k] * mask;
      summands++;
    }
    // the reference asserts summands > 0 (:157); a patch no pool covers cannot come out of its own sanity checks
    const float scale = summands > 0 ? (float)(1.0 / (double)summands) : 0.f;
    in_diff[n * ld_id + c] = acc * scale;
  }
}

// ---- LengthNormComponent (nnet-various.h:338-358) ---------------------------------------------------------------------------
// one wave per row: scale = 1 / sqrt(sum x^2) (sum in double: the reference sums <= 64 columns in double, kaldi-vector.cc:741-748,
// wider rows through sgemv), out = in * scale; the scales are kept for the backward pass
__global__ void __launch_bounds__(kBlock) length_norm_fwd_kernel(float *out, int ld_out, const float *in, int ld_in, float *row_scales, int N, int D) {
  const int lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < N; row += (long)gridDim.x * (kBlock / 64)) {
    const float *x = in + row * ld_in;
    double ss = 0.0;
    for (int j = lane; j < D; j += 64) { const float v = x[j]; ss += (double)(v * v); }   // l2_aux_ = x .* x is a float matrix in the reference
    ss = wave_sum_d(ss);
    const float norm = sqrtf((float)ss);
    const float sc = 1.0f / norm;
    if (lane == 0) row_scales[row] = sc;
    float *y = out + row * ld_out;
    for (int j = lane; j < D; j += 64) y[j] = x[j] * sc;
  }
}

// ---- group p-norm / group max (kaldi-matrix.cc:1071-1138, 2530-2558; kaldi-vector.cc:520-557) ---------------------------------
__device__ __forceinline__ float group_pnorm_of(const float *x, int g, float power) {
  float sum = 0.f;
  if (power == 0.0f) {
    for (int k = 0; k < g; k++) if (x[k] != 0.0f) sum += 1.0f;
    return sum;
  } else if (power == 1.0f) {
    for (int k = 0; k < g; k++) sum += fabsf(x[k]);
    return sum;
  } else if (power == 2.0f) {
    for (int k = 0; k < g; k++) sum += x[k] * x[k];
    return sqrtf(sum);
  }
  bool ok = true;
  for (int k = 0; k < g; k++) {
    const float t = powf(fabsf(x[k]), power);
    if (t == HUGE_VALF) ok = false;
    sum += t;
  }
  if (ok) return powf(sum, 1.0f / power);
  float max_abs = 0.f;   // overflow: rescale by the largest magnitude (kaldi-vector.cc:549-554)
  for (int k = 0; k < g; k++) max_abs = fmaxf(max_abs, fabsf(x[k]));
  const float inv = 1.0f / max_abs;
  sum = 0.f;
  for (int k = 0; k < g; k++) sum += powf(fabsf(x[k] * inv), power);
  return powf(sum, 1.0f / power) * max_abs;
}
template <bool MAX>
__global__ void __launch_bounds__(kBlock) group_reduce_kernel(float *y, int ld_y, const float *x, int ld_x, int N, int out_cols, int group, float power) {
  const long total = (long)N * out_cols;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
    const int j = (int)(i % out_cols);
    const long n = i / out_cols;
    const float *src = x + n * ld_x + (long)j * group;
    float v;
    if (MAX) {
      v = -1e20f;
      for (int k = 0; k < group; k++) v = src[k] > v ? src[k] : v;
    } else {
      v = group_pnorm_of(src, group, power);
    }
    y[n * ld_y + j] = v;
  }
}
// d[n][j] = derivative of the group function wrt input j (GroupPnormDeriv / GroupMaxDeriv), optionally times scale[n][j / group]
// (MulRowsGroupMat): the two passes of the components' BackpropagateFnc in one
template <bool MAX>
__global__ void __launch_bounds__(kBlock) group_deriv_kernel(float *d, int ld_d, const float *in, int ld_in, const float *out, int ld_out, const float *scale,
                                                             int ld_s, int N, int in_cols, int group, float power) {
  const long total = (long)N * in_cols;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
    const int j = (int)(i % in_cols);
    const long n = i / in_cols;
    const float xv = in[n * ld_in + j], yv = out[n * ld_out + j / group];
    float v;
    if (MAX) v = xv == yv ? 1.0f : 0.0f;
    else if (power == 1.0f) v = xv == 0.0f ? 0.0f : (xv > 0.0f ? 1.0f : -1.0f);
    else if (yv == 0.0f) v = 0.0f;
    else v = powf(fabsf(xv), power - 1.0f) * powf(yv, 1.0f - power) * (xv >= 0.0f ? 1.0f : -1.0f);
    if (scale) v *= scale[n * ld_s + j / group];
    d[n * ld_d + j] = v;
  }
}
__global__ void __launch_bounds__(kBlock) mul_rows_group_kernel(float *y, int ld_y, const float *x, int ld_x, int N, int cols, int group) {
  const long total = (long)N * cols;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
    const int j = (int)(i % cols);
    const long n = i / cols;
    y[n * ld_y + j] *= x[n * ld_x + j / group];
  }
}
__global__ void __launch_bounds__(kBlock) max_mat_kernel(float *mat, int ld, const float *A, int ld_a, int N, int cols) {
  const long total = (long)N * cols;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
    const int j = (int)(i % cols);
    const long n = i / cols;
    const float a = A[n * ld_a + j], m = mat[n * ld + j];
    mat[n * ld + j] = a > m ? a : m;
  }
}
__global__ void __launch_bounds__(kBlock) equal_mask_kernel(const float *m1, int ld1, const float *m2, int ld2, float *mask, int ldm, int N, int cols) {
  const long total = (long)N * cols;
  for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
    const int j = (int)(i % cols);
    const long n = i / cols;
    mask[n * ldm + j] = m1[n * ld1 + j] == m2[n * ld2 + j] ? 1.0f : 0.0f;
  }
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

void aslp_conv_gather_patches(float *patches, int ldp, const float *in, MatrixDim d_in, int num_patches, int num_splice, int patch_dim, int patch_step,
                              int patch_stride) {
  if (d_in.rows <= 0 || num_patches <= 0) return;
  if (!patches || !in || num_splice <= 0 || patch_dim <= 0 || patch_step <= 0 || patch_stride <= 0 || num_splice * patch_stride > d_in.cols ||
      (num_patches - 1) * patch_step + patch_dim > patch_stride || ldp < num_splice * patch_dim) {
    set_error("aslp_conv_gather_patches: bad arguments");
    return;
  }
  const long total = (long)d_in.rows * num_patches * num_splice * patch_dim;
  hipLaunchKernelGGL(conv_gather_kernel, dim3(grid_for(total)), dim3(kBlock), 0, cur_stream(), patches, ldp, in, d_in.stride, d_in.rows, num_patches, num_splice,
                     patch_dim, patch_step, patch_stride);
  check_launch("aslp_conv_gather_patches");
}

void aslp_conv_in_diff(float *in_diff, MatrixDim d_id, const float *patch_diff, int ldp, int num_patches, int num_splice, int patch_dim, int patch_step,
                       int patch_stride) {
  if (d_id.rows <= 0 || d_id.cols <= 0) return;
  if (!in_diff || !patch_diff || num_splice <= 0 || patch_dim <= 0 || patch_step <= 0 || patch_stride <= 0 || num_splice * patch_stride != d_id.cols ||
      (num_patches - 1) * patch_step + patch_dim > patch_stride || ldp < num_splice * patch_dim) {
    set_error("aslp_conv_in_diff: bad arguments");
    return;
  }
  hipLaunchKernelGGL(conv_indiff_kernel, dim3(grid_for((long)d_id.rows * d_id.cols)), dim3(kBlock), 0, cur_stream(), in_diff, d_id.stride, patch_diff, ldp,
                     d_id.rows, num_patches, num_splice, patch_dim, patch_step, patch_stride);
  check_launch("aslp_conv_in_diff");
}

void aslp_max_pool_forward(float *out, int ld_out, const float *in, MatrixDim d_in, int pool_size, int pool_step, int pool_stride) {
  if (d_in.rows <= 0) return;
  if (!out || !in || pool_size <= 0 || pool_step <= 0 || pool_stride <= 0 || d_in.cols % pool_stride != 0 || d_in.cols / pool_stride < pool_size) {
    set_error("aslp_max_pool_forward: bad arguments");
    return;
  }
  const int num_patches = d_in.cols / pool_stride, num_pools = 1 + (num_patches - pool_size) / pool_step;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long)d_in.rows * num_pools * pool_stride)), dim3(kBlock), 0, cur_stream(), out, ld_out, in, d_in.stride,
                     d_in.rows, num_pools, pool_size, pool_step, pool_stride);
  check_launch("aslp_max_pool_forward");
}

void aslp_max_pool_backward(float *in_diff, int ld_id, const float *in, MatrixDim d_in, const float *out, int ld_out, const float *out_diff, int ld_od,
                            int pool_size, int pool_step, int pool_stride) {
  if (d_in.rows <= 0) return;
  if (!in_diff || !in || !out || !out_diff || pool_size <= 0 || pool_step <= 0 || pool_stride <= 0 || d_in.cols % pool_stride != 0 ||
      d_in.cols / pool_stride < pool_size) {
    set_error("aslp_max_pool_backward: bad arguments");
    return;
  }
  const int num_patches = d_in.cols / pool_stride, num_pools = 1 + (num_patches - pool_size) / pool_step;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)d_in.rows * d_in.cols)), dim3(kBlock), 0, cur_stream(), in_diff, ld_id, in, d_in.stride, out, ld_out,
                     out_diff, ld_od, d_in.rows, num_patches, num_pools, pool_size, pool_step, pool_stride);
  check_launch("aslp_max_pool_backward");
}

void aslp_length_norm_forward(float *out, int ld_out, const float *in, MatrixDim d_in, float *row_scales) {
  if (d_in.rows <= 0 || d_in.cols <= 0) return;
  if (!out || !in || !row_scales) { set_error("aslp_length_norm_forward: bad arguments"); return; }
  hipLaunchKernelGGL(length_norm_fwd_kernel, dim3(grid_for(d_in.rows, kBlock / 64)), dim3(kBlock), 0, cur_stream(), out, ld_out, in, d_in.stride, row_scales,
                     d_in.rows, d_in.cols);
  check_launch("aslp_length_norm_forward");
}

/* y [rows x cols] = group function of x [rows x cols * group]; in_diff = d(group function) .* out_diff (per group) */
void aslp_group_pnorm_backward(float *in_diff, int ld_id, const float *in, MatrixDim d_in, const float *out, int ld_out, const float *out_diff, int ld_od,
                               int group_size, float power) {
  if (d_in.rows <= 0 || d_in.cols <= 0) return;
  if (!in_diff || !in || !out || group_size <= 0 || d_in.cols % group_size != 0) { set_error("aslp_group_pnorm_backward: bad arguments"); return; }
  hipLaunchKernelGGL((group_deriv_kernel<false>), dim3(grid_for((long)d_in.rows * d_in.cols)), dim3(kBlock), 0, cur_stream(), in_diff, ld_id, in, d_in.stride, out,
                     ld_out, out_diff, ld_od, d_in.rows, d_in.cols, group_size, power);
  check_launch("aslp_group_pnorm_backward");
}
void aslp_group_max_backward(float *in_diff, int ld_id, const float *in, MatrixDim d_in, const float *out, int ld_out, const float *out_diff, int ld_od,
                             int group_size) {
  if (d_in.rows <= 0 || d_in.cols <= 0) return;
  if (!in_diff || !in || !out || group_size <= 0 || d_in.cols % group_size != 0) { set_error("aslp_group_max_backward: bad arguments"); return; }
  hipLaunchKernelGGL((group_deriv_kernel<true>), dim3(grid_for((long)d_in.rows * d_in.cols)), dim3(kBlock), 0, cur_stream(), in_diff, ld_id, in, d_in.stride, out,
                     ld_out, out_diff, ld_od, d_in.rows, d_in.cols, group_size, 0.0f);
  check_launch("aslp_group_max_backward");
}

/* ---- kernel ABI (cu-kernels-ansi.h; Gr / Bl accepted and ignored) ---- */
void cudaF_max(aslp_dim3, aslp_dim3, float *mat, const float *A, MatrixDim dst_d, int src_stride) {   /* [85] */
  if (dst_d.rows <= 0 || dst_d.cols <= 0) return;
  hipLaunchKernelGGL(max_mat_kernel, dim3(grid_for((long)dst_d.rows * dst_d.cols)), dim3(kBlock), 0, cur_stream(), mat, dst_d.stride, A, src_stride, dst_d.rows,
                     dst_d.cols);
  check_launch("max");
}
void cudaF_mul_rows_group_mat(aslp_dim3, aslp_dim3, float *y, const float *x, MatrixDim d, int src_stride, int group_size) {   /* [88] */
  if (d.rows <= 0 || d.cols <= 0 || group_size <= 0) return;
  hipLaunchKernelGGL(mul_rows_group_kernel, dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), y, d.stride, x, src_stride, d.rows, d.cols,
                     group_size);
  check_launch("mul_rows_group_mat");
}
void cudaF_calc_pnorm_deriv(aslp_dim3, aslp_dim3, float *y, const float *x1, const float *x2, MatrixDim d, int src_stride, int group_size, float power) {   /* [89] */
  if (d.rows <= 0 || d.cols <= 0 || group_size <= 0) return;
  // y, x1 share d.stride; x2 (the group norms) has src_stride (cu-matrix.cc GroupPnormDeriv)
  hipLaunchKernelGGL((group_deriv_kernel<false>), dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), y, d.stride, x1, d.stride, x2, src_stride,
                     (const float *)nullptr, 0, d.rows, d.cols, group_size, power);
  check_launch("calc_pnorm_deriv");
}
void cudaF_calc_group_max_deriv(aslp_dim3, aslp_dim3, float *y, const float *x1, const float *x2, MatrixDim d, int src_stride, int group_size) {   /* [90] */
  if (d.rows <= 0 || d.cols <= 0 || group_size <= 0) return;
  hipLaunchKernelGGL((group_deriv_kernel<true>), dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), y, d.stride, x1, d.stride, x2, src_stride,
                     (const float *)nullptr, 0, d.rows, d.cols, group_size, 0.0f);
  check_launch("calc_group_max_deriv");
}
void cudaF_group_pnorm(aslp_dim3, aslp_dim3, float *y, const float *x, MatrixDim d, int src_stride, int group_size, float power) {   /* [146] */
  if (d.rows <= 0 || d.cols <= 0 || group_size <= 0) return;
  hipLaunchKernelGGL((group_reduce_kernel<false>), dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), y, d.stride, x, src_stride, d.rows, d.cols,
                     group_size, power);
  check_launch("group_pnorm");
}
void cudaF_group_max(aslp_dim3, aslp_dim3, float *y, const float *x, MatrixDim d, int src_stride, int group_size) {   /* [147] */
  if (d.rows <= 0 || d.cols <= 0 || group_size <= 0) return;
  hipLaunchKernelGGL((group_reduce_kernel<true>), dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), y, d.stride, x, src_stride, d.rows, d.cols,
                     group_size, 0.0f);
  check_launch("group_max");
}
void cudaF_equal_element_mask(aslp_dim3, aslp_dim3, const float *mat1, const float *mat2, float *mask, MatrixDim mat1_dim, int mat2_stride, int mask_stride) {   /* [182] */
  if (mat1_dim.rows <= 0 || mat1_dim.cols <= 0) return;
  hipLaunchKernelGGL(equal_mask_kernel, dim3(grid_for((long)mat1_dim.rows * mat1_dim.cols)), dim3(kBlock), 0, cur_stream(), mat1, mat1_dim.stride, mat2,
                     mat2_stride, mask, mask_stride, mat1_dim.rows, mat1_dim.cols);
  check_launch("equal_element_mask");
}

}  // extern "C"
