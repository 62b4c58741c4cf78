// blas_seam.hip -- seam B2 (include/aslp_blas.h): cuBLAS-signature, column-major adapters over the row-major kernels.
#include "aslp_blas.h"

#include <hip/hip_runtime.h>

#include "aslp_kernels.h"
#include "common.h"
#include "scratch.h"

struct aslp_blas_handle_s {
  hipStream_t stream = nullptr;
  bool own_stream = false;  // false: launches follow the calling thread's current stream (aslp_set_stream)
};

namespace aslp {
namespace {

// runs f with the handle's stream as the calling thread's current stream
template <class F>
int on_stream(aslp_blas_handle_t h, F f) {
  if (!h) { set_error("aslp_blas: NULL handle"); return 1; }
  hipStream_t saved = cur_stream();
  if (h->own_stream) set_cur_stream(h->stream);
  f();
  if (h->own_stream) set_cur_stream(saved);
  return has_error() ? 1 : 0;
}

__global__ void __launch_bounds__(kBlock) axpy_strided(int n, float alpha, const float *x, long incx, float *y, long incy) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i * incy] += alpha * x[i * incx];
}
__global__ void __launch_bounds__(kBlock) scal_strided(int n, float alpha, float *x, long incx) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i * incx] *= alpha;
}
__global__ void __launch_bounds__(kBlock) copy_strided(int n, const float *x, long incx, float *y, long incy) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i * incy] = x[i * incx];
}
// one workgroup: fixed order, reproducible
__global__ void __launch_bounds__(1024) dot_strided(int n, const float *x, long incx, const float *y, long incy, float *out) {
  __shared__ float part[16];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) s += x[i * incx] * y[i * incy];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; w++) t += part[w];
    *out = t;
  }
}
// A (column-major, lda) += alpha x y^T: thread per element, i (rows, contiguous) fastest
__global__ void __launch_bounds__(kBlock) ger_colmajor(int m, int n, float alpha, const float *x, long incx, const float *y, long incy, float *A, long lda) {
  const long total = (long)m * n;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long j = e / m, i = e - j * m;
    A[i + j * lda] += alpha * x[i * incx] * y[j * incy];
  }
}
// y[i] = alpha * sum_j A[i + j lda] x[j] + beta y[i]: threads over i (contiguous in a column)
__global__ void __launch_bounds__(kBlock) gemv_n_colmajor(int m, int n, float alpha, const float *A, long lda, const float *x, long incx, float beta,
                                                         float *y, long incy) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < m; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < n; j++) s += A[i + j * lda] * x[j * incx];
    y[i * incy] = alpha * s + (beta == 0.0f ? 0.0f : beta * y[i * incy]);
  }
}
// y[j] = alpha * sum_i A[i + j lda] x[i] + beta y[j]: one wave per column
__global__ void __launch_bounds__(kBlock) gemv_t_colmajor(int m, int n, float alpha, const float *A, long lda, const float *x, long incx, float beta,
                                                         float *y, long incy) {
  const int lane = threadIdx.x & 63, wpb = kBlock / 64;
  for (long j = blockIdx.x * (long)wpb + (threadIdx.x >> 6); j < n; j += (long)gridDim.x * wpb) {
    float s = 0.f;
    for (int i = lane; i < m; i += 64) s += A[i + j * lda] * x[i * incx];
    s = wave_sum(s);
    if (lane == 0) y[j * incy] = alpha * s + (beta == 0.0f ? 0.0f : beta * y[j * incy]);
  }
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

int aslp_blas_create(aslp_blas_handle_t *handle) {
  if (!handle) return 1;
  *handle = new aslp_blas_handle_s();
  return 0;
}
int aslp_blas_destroy(aslp_blas_handle_t handle) { delete handle; return 0; }
int aslp_blas_set_stream(aslp_blas_handle_t handle, void *stream) {
  if (!handle) return 1;
  handle->stream = static_cast<hipStream_t>(stream);
  handle->own_stream = stream != nullptr;
  return 0;
}

int aslp_blas_sgemm(aslp_blas_handle_t handle, aslp_blas_operation_t transa, aslp_blas_operation_t transb, int m, int n, int k, float alpha,
                    const float *A, int lda, const float *B, int ldb, float beta, float *C, int ldc) {
  // column-major C[m x n] is the row-major matrix C^T [n x m] with row stride ldc, and C^T = op(B)^T op(A)^T: a column-major
  // operand stored with leading dimension ld IS its transpose in row-major with row stride ld
  int rc = 0;
  const int st = on_stream(handle, [&] {
    rc = aslp_sgemm(transb == ASLP_BLAS_OP_T ? 1 : 0, transa == ASLP_BLAS_OP_T ? 1 : 0, n, m, k, alpha, B, ldb, A, lda, beta, C, ldc);
    if (rc) set_error("aslp_blas_sgemm: bad arguments (code " + std::to_string(rc) + ")");
  });
  return st || rc;
}
int aslp_blas_sger(aslp_blas_handle_t handle, int m, int n, float alpha, const float *x, int incx, const float *y, int incy, float *A, int lda) {
  if (m <= 0 || n <= 0) return 0;
  return on_stream(handle, [&] {
    hipLaunchKernelGGL(ger_colmajor, dim3(grid_for((long)m * n)), dim3(kBlock), 0, cur_stream(), m, n, alpha, x, (long)incx, y, (long)incy, A, (long)lda);
    check_launch("aslp_blas_sger");
  });
}
int aslp_blas_sgemv(aslp_blas_handle_t handle, aslp_blas_operation_t trans, int m, int n, float alpha, const float *A, int lda, const float *x,
                    int incx, float beta, float *y, int incy) {
  if (m <= 0 || n <= 0) return 0;
  return on_stream(handle, [&] {
    if (trans == ASLP_BLAS_OP_N) hipLaunchKernelGGL(gemv_n_colmajor, dim3(grid_for(m)), dim3(kBlock), 0, cur_stream(), m, n, alpha, A, (long)lda, x, (long)incx, beta, y, (long)incy);
    else hipLaunchKernelGGL(gemv_t_colmajor, dim3(grid_for((long)n * 64)), dim3(kBlock), 0, cur_stream(), m, n, alpha, A, (long)lda, x, (long)incx, beta, y, (long)incy);
    check_launch("aslp_blas_sgemv");
  });
}
int aslp_blas_sdot(aslp_blas_handle_t handle, int n, const float *x, int incx, const float *y, int incy, float *result_host) {
  if (!result_host) return 1;
  if (n <= 0) { *result_host = 0.0f; return 0; }
  return on_stream(handle, [&] {
    float *d = static_cast<float *>(scratch(kScratchReduce, 64));
    if (!d) return;
    hipLaunchKernelGGL(dot_strided, dim3(1), dim3(1024), 0, cur_stream(), n, x, (long)incx, y, (long)incy, d);
    check_launch("aslp_blas_sdot");
    ASLP_CHECK_HIP(hipMemcpyAsync(result_host, d, sizeof(float), hipMemcpyDeviceToHost, cur_stream()));
    ASLP_CHECK_HIP(hipStreamSynchronize(cur_stream()));
  });
}
int aslp_blas_saxpy(aslp_blas_handle_t handle, int n, float alpha, const float *x, int incx, float *y, int incy) {
  if (n <= 0) return 0;
  return on_stream(handle, [&] {
    hipLaunchKernelGGL(axpy_strided, dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), n, alpha, x, (long)incx, y, (long)incy);
    check_launch("aslp_blas_saxpy");
  });
}
int aslp_blas_sscal(aslp_blas_handle_t handle, int n, float alpha, float *x, int incx) {
  if (n <= 0) return 0;
  return on_stream(handle, [&] {
    hipLaunchKernelGGL(scal_strided, dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), n, alpha, x, (long)incx);
    check_launch("aslp_blas_sscal");
  });
}
int aslp_blas_scopy(aslp_blas_handle_t handle, int n, const float *x, int incx, float *y, int incy) {
  if (n <= 0) return 0;
  return on_stream(handle, [&] {
    hipLaunchKernelGGL(copy_strided, dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), n, x, (long)incx, y, (long)incy);
    check_launch("aslp_blas_scopy");
  });
}

}  // extern "C"
