// reduce_kernels.hip -- row / column reductions of the substrate for gfx950.
//
// Row reductions (softmax, argmax): one wave (small rows) or one 4-wave block per row; the
// row is read from HBM ONCE into registers, reduced with 64-lane shuffles (+ a 4-entry LDS
// exchange for the block form) and written once -- the reference's _softmax_reduce
// (cu-kernels.cu:1858-1922) re-reads the row three times.  Column reductions: colreduce.h.
#include "aslp_kernels.h"
#include "colreduce.h"
#include "common.h"

namespace aslp {
namespace {

constexpr int kMaxPerThread = 32;  // row elements cached per thread (cols <= 32 * threads)

template <int T>
__device__ __forceinline__ float block_reduce_max(float v, float *sh) {
  v = wave_max(v);
  if (T == 64) return v;
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int i = 1; i < T / 64; i++) r = fmaxf(r, sh[i]);
  __syncthreads();
  return r;
}
template <int T>
__device__ __forceinline__ float block_reduce_sum(float v, float *sh) {
  v = wave_sum(v);
  if (T == 64) return v;
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int i = 1; i < T / 64; i++) r += sh[i];
  __syncthreads();
  return r;
}

// T threads cooperate on one row; blockDim = (T, 256/T) so small rows still fill 4 waves.
// VEC (T = 256, cols % 4 == 0, 16-byte aligned rows): slot k of a thread is column 4 (tid + 256 (k / 4)) + k % 4 -- four consecutive columns per
// 16-byte access instead of one column per 4-byte access; the thread's sum runs over its slots in slot order either way.  xent_rows_kernel
// (nn_fused.hip) uses the same two maps under the same rule, so folding the Softmax into the loss changes no bit.
template <int T, bool LOG, int PER = kMaxPerThread, bool VEC = false>
__global__ void __launch_bounds__(256) softmax_rows_kernel(float *y, const float *x, int rows, int cols, int ldy, int ldx) {
  static_assert(!VEC || (T == 256 && PER % 4 == 0), "the 16-byte map is for whole-workgroup rows");
  __shared__ float sh_all[4][4];
  float *sh = sh_all[threadIdx.y];
  auto col_of = [](int k) { return VEC ? 4 * ((int)threadIdx.x + (k >> 2) * T) + (k & 3) : (int)threadIdx.x + k * T; };
  for (int r = blockIdx.x * blockDim.y + threadIdx.y; r < rows; r += gridDim.x * blockDim.y) {
    const float *xr = x + (long)r * ldx;
    float *yr = y + (long)r * ldy;
    float v[PER];
    float m = -INFINITY;
    if (VEC) {
#pragma unroll
      for (int k4 = 0; k4 < PER / 4; k4++) {
        const int c = 4 * ((int)threadIdx.x + k4 * T);
        if (c < cols) {
          const float4 t4 = *reinterpret_cast<const float4 *>(xr + c);
          v[4 * k4] = t4.x; v[4 * k4 + 1] = t4.y; v[4 * k4 + 2] = t4.z; v[4 * k4 + 3] = t4.w;
          m = fmaxf(fmaxf(m, fmaxf(t4.x, t4.y)), fmaxf(t4.z, t4.w));
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < PER; k++) {
        int c = threadIdx.x + k * T;
        if (c < cols) {
          v[k] = xr[c];
          m = fmaxf(m, v[k]);
        }
      }
    }
    m = block_reduce_max<T>(m, sh);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      int c = col_of(k);
      if (c < cols) {
        if (LOG) {
          v[k] -= m;
          s += expf(v[k]);
        } else {
          v[k] = expf(v[k] - m);
          s += v[k];
        }
      }
    }
    s = block_reduce_sum<T>(s, sh);
    float k2 = LOG ? logf(s) : 1.0f / s;
    if (VEC) {
#pragma unroll
      for (int k4 = 0; k4 < PER / 4; k4++) {
        const int c = 4 * ((int)threadIdx.x + k4 * T);
        if (c < cols) {
          float o[4];
#pragma unroll
          for (int q = 0; q < 4; q++) o[q] = LOG ? v[4 * k4 + q] - k2 : v[4 * k4 + q] * k2;
          *reinterpret_cast<float4 *>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < PER; k++) {
        int c = threadIdx.x + k * T;
        if (c < cols) yr[c] = LOG ? v[k] - k2 : v[k] * k2;
      }
    }
  }
}

// very wide rows: three streaming passes (max, sum, write)
template <bool LOG>
__global__ void __launch_bounds__(256) softmax_rows_wide(float *y, const float *x, int rows, int cols, int ldy, int ldx) {
  __shared__ float sh[4];
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float *xr = x + (long)r * ldx;
    float *yr = y + (long)r * ldy;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, xr[c]);
    m = block_reduce_max<256>(m, sh);
    float s = 0.0f;
    for (int c = threadIdx.x; c < cols; c += 256) s += expf(xr[c] - m);
    s = block_reduce_sum<256>(s, sh);
    float k2 = LOG ? logf(s) : 1.0f / s;
    for (int c = threadIdx.x; c < cols; c += 256) yr[c] = LOG ? (xr[c] - m) - k2 : expf(xr[c] - m) * k2;
  }
}

template <bool LOG>
void launch_softmax(float *y, const float *x, MatrixDim d, int src_stride) {
  if (d.rows <= 0 || d.cols <= 0) return;
  if (d.cols <= 64 * kMaxPerThread / 4) {  // <= 512 columns: one wave per row, 4 rows per block
    int g = (d.rows + 3) / 4;
    if (g > kMaxGrid * 2) g = kMaxGrid * 2;
    hipLaunchKernelGGL((softmax_rows_kernel<64, LOG>), dim3(g), dim3(64, 4), 0, cur_stream(), y, x, d.rows, d.cols, d.stride, src_stride);
  } else if (d.cols <= 256 * kMaxPerThread) {
    int g = d.rows > kMaxGrid * 4 ? kMaxGrid * 4 : d.rows;
    // per-thread slots: the smallest of 4 / 8 / 16 / 32 that covers the row (fully unrolled loops; same lane -> column map and
    // summation order for every choice, so the result does not depend on it)
    const int per = (d.cols + 255) / 256;
    // 16-byte accesses where every row starts on a 16-byte boundary (softmax_rows_vec_ok: the rule xent_rows_kernel follows too)
    const bool vec = softmax_rows_vec_ok(d.cols) && (d.stride & 3) == 0 && (src_stride & 3) == 0 && aligned16(y) && aligned16(x);
#define ASLP_SOFTMAX_ROWS(P)                                                                                                                        \
    do {                                                                                                                                            \
      if (vec) hipLaunchKernelGGL((softmax_rows_kernel<256, LOG, P, true>), dim3(g), dim3(256, 1), 0, cur_stream(), y, x, d.rows, d.cols, d.stride, src_stride); \
      else hipLaunchKernelGGL((softmax_rows_kernel<256, LOG, P>), dim3(g), dim3(256, 1), 0, cur_stream(), y, x, d.rows, d.cols, d.stride, src_stride);           \
    } while (0)
    // (the slots per thread are chosen from the scalar map's count for both maps: cols / 1024 <= cols / 256)
    if (per <= 4) ASLP_SOFTMAX_ROWS(4);
    else if (per <= 8) ASLP_SOFTMAX_ROWS(8);
    else if (per <= 16) ASLP_SOFTMAX_ROWS(16);
    else ASLP_SOFTMAX_ROWS(32);
#undef ASLP_SOFTMAX_ROWS
  } else {
    int g = d.rows > kMaxGrid ? kMaxGrid : d.rows;
    hipLaunchKernelGGL((softmax_rows_wide<LOG>), dim3(g), dim3(256), 0, cur_stream(), y, x, d.rows, d.cols, d.stride, src_stride);
  }
  check_launch(LOG ? "log_softmax" : "softmax");
}

// (value, index) argmax with the reference's tie rule: first strict maximum wins, i.e. the
// smallest index among equal maxima; NaNs never win (cu-matrix.cc:1493-1510).
__device__ __forceinline__ void argmax_combine(float &v, int &i, float ov, int oi) {
  if (ov > v || (ov == v && oi >= 0 && (i < 0 || oi < i))) {
    v = ov;
    i = oi;
  }
}

// one wave per row; init from (vec_val, vec_id) when `carry` (the reference's running form)
__global__ void __launch_bounds__(256) row_argmax_kernel(const float *mat, float *vec_val, int32_t *vec_id, int voff, int rows,
                                                         int cols, int ld, bool carry) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
    const float *row = mat + (long)r * ld;
    float best = -1e21f;
    int bi = -1;
    for (int c = lane; c < cols; c += 64) {
      float x = row[c];
      if (best < x) {  // strict: keeps the earliest index within this lane
        best = x;
        bi = c + voff;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o, 64);
      int oi = __shfl_xor(bi, o, 64);
      argmax_combine(best, bi, ov, oi);
    }
    if (lane == 0) {
      if (carry) {
        float pv = vec_val[r];
        if (pv < best) {  // strict, earlier blocks win ties (cu-kernels.cu:2141-2170)
          vec_val[r] = best;
          vec_id[r] = bi;
        }
      } else {
        if (vec_val) vec_val[r] = best;
        vec_id[r] = bi;
      }
    }
  }
}

// cu-kernels.cu:2172 _diff_xent: log_post[r] = log(y[r][tgt]); y[r][tgt] -= 1
__global__ void diff_xent_kernel(const int32_t *tgt, float *mat, float *log_post, MatrixDim d) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < d.rows; r += gridDim.x * blockDim.x) {
    long idx = (long)r * d.stride + tgt[r];
    float v = mat[idx];
    log_post[r] = logf(v);
    mat[idx] = v - 1.0f;
  }
}

// v[r] = alpha * sum_c M[r][c] + beta * v[r]; one wave per row
__global__ void __launch_bounds__(256) rowsum_kernel(float alpha, const float *M, MatrixDim d, float beta, float *v) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + w; r < d.rows; r += gridDim.x * 4) {
    const float *row = M + (long)r * d.stride;
    float s = 0.0f;
    for (int c = lane; c < d.cols; c += 64) s += row[c];
    s = wave_sum(s);
    if (lane == 0) v[r] = beta == 0.0f ? alpha * s : alpha * s + beta * v[r];
  }
}

// total sum in double: per-row partials then one block
__global__ void __launch_bounds__(256) rowsum_d_kernel(const float *M, MatrixDim d, double *rowpart) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + w; r < d.rows; r += gridDim.x * 4) {
    const float *row = M + (long)r * d.stride;
    double s = 0.0;
    for (int c = lane; c < d.cols; c += 64) s += (double)row[c];
    s = wave_sum_d(s);
    if (lane == 0) rowpart[r] = s;
  }
}
__global__ void __launch_bounds__(256) final_sum_d_kernel(const double *part, int n, double *out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void __launch_bounds__(256) vec_sum_kernel(const float *v, float *value, int dim, int inc) {
  __shared__ float sh[4];
  float s = 0.0f;
  for (int i = threadIdx.x; i < dim; i += 256) s += v[(long)i * inc];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *value = sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- column reductions -----------------------------------------------------------------
struct ColSumF {
  static constexpr bool kVec = true;
  const float *M; int ld;
  template <int VW>
  __device__ void operator()(int r, int c, float (&acc)[1][VW]) const {
    float v[VW];
    loadv<VW>(M + (long)r * ld + c, v);
#pragma unroll
    for (int i = 0; i < VW; i++) acc[0][i] += v[i];
  }
};
struct ColSumG {
  float alpha, beta; float *v;
  float *w; float w_alpha;   // optional fused SGD step on a parameter vector: w += w_alpha * v_new
  __device__ void operator()(int c, const float (&s)[1]) const {
    float nv = beta == 0.0f ? alpha * s[0] : alpha * s[0] + beta * v[c];
    v[c] = nv;
    if (w) w[c] += w_alpha * nv;
  }
};
// v[c] = alpha * sum_j M'[c][j] * N'[j][c] + beta v[c] with generic strides
struct DiagMMF {
  static constexpr bool kVec = false;
  const float *M; long m_rs, m_cs; const float *N; long n_rs, n_cs;
  // here "r" runs over the summed dimension j, "c" over v's index
  template <int VW>
  __device__ void operator()(int j, int c, float (&acc)[1][VW]) const { acc[0][0] += M[c * m_rs + j * m_cs] * N[j * n_rs + c * n_cs]; }
};

// ASLP _add_row_sum_mat (cu-kernels.cu:754-770): dst[j][c] = alpha * sum_{k<P} src[j*P+k][c] + beta*dst[j][c]
template <bool VEC>
__global__ void __launch_bounds__(kBlock) group_row_sum_kernel(float *dst, const float *src, MatrixDim d, int src_stride, int P,
                                                               float alpha, float beta) {
  constexpr int W = VEC ? 4 : 1;
  int cw = d.cols / W;
  long n = (long)d.rows * cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int j = (int)(i / cw), c = (int)(i - (long)j * cw) * W;
    const float *s = src + (long)j * P * src_stride + c;
    if (VEC) {
      float4 acc = make_float4(0, 0, 0, 0);
      for (int k = 0; k < P; k++, s += src_stride) {
        float4 v = *reinterpret_cast<const float4 *>(s);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      float4 *dp = reinterpret_cast<float4 *>(dst + (long)j * d.stride + c);
      float4 o = beta == 0.0f ? make_float4(0, 0, 0, 0) : *dp;
      o.x = alpha * acc.x + beta * o.x; o.y = alpha * acc.y + beta * o.y;
      o.z = alpha * acc.z + beta * o.z; o.w = alpha * acc.w + beta * o.w;
      *dp = o;
    } else {
      float acc = 0.0f;
      for (int k = 0; k < P; k++, s += src_stride) acc += *s;
      float *dp = dst + (long)j * d.stride + c;
      *dp = beta == 0.0f ? alpha * acc : alpha * acc + beta * *dp;
    }
  }
}

// ASLP _add_conv_mat_mat_elements (cu-kernels.cu:772-782): dst[k*C + c][:] = alpha * A[k+c][:] .* B[c][:] + beta*dst
__global__ void __launch_bounds__(kBlock) conv_mat_mat_elements_kernel(float *dst, const float *A, const float *B, MatrixDim d,
                                                                       int sa, int sb, int C, float alpha, float beta) {
  long n = (long)d.rows * d.cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int row = (int)(i / d.cols), col = (int)(i - (long)row * d.cols);
    int k = row / C, c = row - k * C;
    float *dp = dst + (long)row * d.stride + col;
    float v = alpha * A[(long)(k + c) * sa + col] * B[(long)c * sb + col];
    *dp = beta == 0.0f ? v : v + beta * *dp;
  }
}

// max-norm (nnet-affine-transform.h:231-243): one wave per row
__global__ void __launch_bounds__(256) max_norm_kernel(float *W, MatrixDim d, float max_norm) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + w; r < d.rows; r += gridDim.x * 4) {
    float *row = W + (long)r * d.stride;
    float s = 0.0f;
    for (int c = lane; c < d.cols; c += 64) s += row[c] * row[c];
    s = wave_sum(s);
    float scl = sqrtf(s) * (1.0f / max_norm);
    if (scl < 1.0f) scl = 1.0f;
    scl = 1.0f / scl;
    for (int c = lane; c < d.cols; c += 64) row[c] *= scl;
  }
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

void cudaF_softmax_reduce(size_t, size_t, float *y, const float *x, MatrixDim d, int src_stride) { launch_softmax<false>(y, x, d, src_stride); }
void cudaF_log_softmax_reduce(size_t, size_t, float *y, const float *x, MatrixDim d, int src_stride) { launch_softmax<true>(y, x, d, src_stride); }

void cudaF_find_row_max_id(aslp_dim3, aslp_dim3 Bl, const float *mat, float *vec_val, int32_cuda *vec_id, int32_cuda voff, MatrixDim d) {
  if (d.rows <= 0) return;
  int cols = (int)Bl.x;  // the reference launches one call per (<=256)-column block, Bl.x wide
  int g = (d.rows + 3) / 4;
  if (g > kMaxGrid) g = kMaxGrid;
  hipLaunchKernelGGL(row_argmax_kernel, dim3(g), dim3(256), 0, cur_stream(), mat, vec_val, vec_id, voff, d.rows, cols, d.stride, true);
  check_launch("find_row_max_id");
}
void aslp_find_row_max_id(const float *M, MatrixDim d, int32_cuda *id) {
  if (d.rows <= 0) return;
  int g = (d.rows + 3) / 4;
  if (g > kMaxGrid) g = kMaxGrid;
  hipLaunchKernelGGL(row_argmax_kernel, dim3(g), dim3(256), 0, cur_stream(), M, (float *)nullptr, id, 0, d.rows, d.cols, d.stride, false);
  check_launch("aslp_find_row_max_id");
}
void cudaF_diff_xent(aslp_dim3, aslp_dim3, const int32_cuda *vec_tgt, float *mat_net_out, float *vec_log_post, MatrixDim d) {
  if (d.rows <= 0) return;
  hipLaunchKernelGGL(diff_xent_kernel, dim3(grid_for(d.rows)), dim3(kBlock), 0, cur_stream(), vec_tgt, mat_net_out, vec_log_post, d);
  check_launch("diff_xent");
}

void aslp_add_row_sum_mat_vec(float alpha, const float *M, MatrixDim d, float beta, float *v) {
  colreduce<1, float>("add_row_sum_mat_vec", d.rows, d.cols, ColSumF{M, d.stride}, ColSumG{alpha, beta, v, nullptr, 0.0f},
                      aligned16(M) && d.stride % 4 == 0);
}
void aslp_add_row_sum_mat_vec_sgd(float alpha, const float *M, MatrixDim d, float beta, float *v, float *w, float w_alpha) {
  colreduce<1, float>("add_row_sum_mat_vec_sgd", d.rows, d.cols, ColSumF{M, d.stride}, ColSumG{alpha, beta, v, w, w_alpha},
                      aligned16(M) && d.stride % 4 == 0);
}
void aslp_add_col_sum_mat_vec(float alpha, const float *M, MatrixDim d, float beta, float *v) {
  if (d.rows <= 0) return;
  int g = (d.rows + 3) / 4;
  if (g > kMaxGrid) g = kMaxGrid;
  hipLaunchKernelGGL(rowsum_kernel, dim3(g), dim3(256), 0, cur_stream(), alpha, M, d, beta, v);
  check_launch("add_col_sum_mat_vec");
}
void aslp_matrix_sum(const float *M, MatrixDim d, double *out_dev) {
  int rows = d.rows > 0 ? d.rows : 0;
  double *part = static_cast<double *>(scratch(kScratchReduce, sizeof(double) * (size_t)(rows + 1)));
  if (!part) return;
  if (rows > 0) {
    int g = (rows + 3) / 4;
    if (g > kMaxGrid) g = kMaxGrid;
    hipLaunchKernelGGL(rowsum_d_kernel, dim3(g), dim3(256), 0, cur_stream(), M, d, part);
  }
  hipLaunchKernelGGL(final_sum_d_kernel, dim3(1), dim3(256), 0, cur_stream(), part, rows, out_dev);
  check_launch("matrix_sum");
}
void cudaF_vec_sum(int, int, float *v, float *value, int dim, int inc) {
  hipLaunchKernelGGL(vec_sum_kernel, dim3(1), dim3(256), 0, cur_stream(), v, value, dim, inc);
  check_launch("vec_sum");
}
void cudaF_add_diag_mat_mat(int, int, float alpha, float *v, int v_dim, const float *M, int M_cols, int M_row_stride, int M_col_stride,
                            const float *N, int N_row_stride, int N_col_stride, int, float beta) {
  // v[i] = alpha * sum_j M[i*rs + j*cs] * N[j*rs' + i*cs'] + beta*v[i]; the summed index j plays "rows"
  colreduce<1, float>("add_diag_mat_mat", M_cols, v_dim, DiagMMF{M, M_row_stride, M_col_stride, N, N_row_stride, N_col_stride},
                      ColSumG{alpha, beta, v, nullptr, 0.0f});
}
void cudaF_add_row_sum_mat(aslp_dim3, aslp_dim3, float *data, const float *src, MatrixDim dim, int src_stride, int patch_nrows, float alpha, float beta) {
  if (dim.rows <= 0 || dim.cols <= 0) return;
  bool vec = dim.cols % 4 == 0 && dim.stride % 4 == 0 && src_stride % 4 == 0 && aligned16(data) && aligned16(src);
  long n = (long)dim.rows * (vec ? dim.cols / 4 : dim.cols);
  if (vec) hipLaunchKernelGGL((group_row_sum_kernel<true>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), data, src, dim, src_stride, patch_nrows, alpha, beta);
  else hipLaunchKernelGGL((group_row_sum_kernel<false>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), data, src, dim, src_stride, patch_nrows, alpha, beta);
  check_launch("add_row_sum_mat");
}
void cudaF_add_conv_mat_mat_elements(aslp_dim3, aslp_dim3 Bl, float *data, const float *A, const float *B, MatrixDim dim, int sa, int sb, float alpha, float beta) {
  if (dim.rows <= 0 || dim.cols <= 0) return;
  int C = (int)Bl.y;  // filter length: the reference encodes it in the launch geometry (cu-matrix.cc:3052-3056)
  if (C <= 0) { set_error("add_conv_mat_mat_elements: Bl.y (rows of B) must be > 0"); return; }
  hipLaunchKernelGGL(conv_mat_mat_elements_kernel, dim3(grid_for((long)dim.rows * dim.cols)), dim3(kBlock), 0, cur_stream(), data, A, B, dim, sa, sb, C, alpha, beta);
  check_launch("add_conv_mat_mat_elements");
}
void aslp_max_norm_rows(float *W, MatrixDim d, float max_norm) {
  if (d.rows <= 0 || max_norm <= 0.0f) return;
  int g = (d.rows + 3) / 4;
  if (g > kMaxGrid) g = kMaxGrid;
  hipLaunchKernelGGL(max_norm_kernel, dim3(g), dim3(256), 0, cur_stream(), W, d, max_norm);
  check_launch("max_norm_rows");
}

}  // extern "C"
