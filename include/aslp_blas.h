/*
 * aslp_blas.h -- seam B2 of SURVEY.md §8b: the BLAS calls the reference's CuMatrix / CuVector issue through
 * src/aslp-cudamatrix/cublas-wrappers.h:28-133, with cuBLAS' own argument order and COLUMN-MAJOR convention, so that the
 * reference's cu-matrix.cc / cu-vector.cc can be relinked against this library instead of edited
 * (cu-matrix.cc:1049-1053 calls cublas_gemm(handle, transb, transa, m, n, k, alpha, B, ldb, A, lda, beta, C, ldc)).
 *
 *   cublas_gemm  (cublas-wrappers.h:28-33)   -> aslp_blas_sgemm      cublas_axpy (:122-125) -> aslp_blas_saxpy
 *   cublas_ger   (:40-43)                    -> aslp_blas_sger       cublas_scal (:113-116) -> aslp_blas_sscal
 *   cublas_dot   (:78-81)                    -> aslp_blas_sdot       cublas_copy (:104-107) -> aslp_blas_scopy
 *   cublas_gemv  (:131-135)                  -> aslp_blas_sgemv      cublasCreate / cublasSetStream / cublasDestroy
 *                                                                   (cu-device.cc:128-131, CuDevice::GetHandle cu-device.h:50)
 * All matrix / vector pointers are device memory; alpha / beta are passed by value like the wrappers do; results of sdot
 * land in host memory like cuBLAS' default pointer mode (the call synchronises its stream).  Return value: 0 =
 * CUBLAS_STATUS_SUCCESS, non-zero = failure (message in aslp_get_last_error of aslp_kernels.h).  fp32 only: the reference
 * instantiates the double versions but the training path never calls them (BaseFloat = float).
 * These are thin adapters: the column-major product is handed to the row-major MFMA kernels with the operands swapped.
 */
#ifndef ASLP_BLAS_H_
#define ASLP_BLAS_H_
#ifdef __cplusplus
extern "C" {
#endif

typedef struct aslp_blas_handle_s *aslp_blas_handle_t;                          /* cublasHandle_t */
typedef enum { ASLP_BLAS_OP_N = 0, ASLP_BLAS_OP_T = 1 } aslp_blas_operation_t;  /* CUBLAS_OP_N = 0, CUBLAS_OP_T = 1 */

int aslp_blas_create(aslp_blas_handle_t *handle);                    /* cublasCreate  */
int aslp_blas_destroy(aslp_blas_handle_t handle);                    /* cublasDestroy */
int aslp_blas_set_stream(aslp_blas_handle_t handle, void *stream);   /* cublasSetStream (hipStream_t; NULL = the calling thread's current stream) */

/* C[m x n] = alpha * op(A)[m x k] * op(B)[k x n] + beta * C, column-major, leading dimensions lda / ldb / ldc */
int aslp_blas_sgemm(aslp_blas_handle_t handle, aslp_blas_operation_t transa, aslp_blas_operation_t transb, int m, int n, int k, float alpha,
                    const float *A, int lda, const float *B, int ldb, float beta, float *C, int ldc);
/* A[m x n] (column-major, lda) += alpha * x y^T */
int aslp_blas_sger(aslp_blas_handle_t handle, int m, int n, float alpha, const float *x, int incx, const float *y, int incy, float *A, int lda);
/* y = alpha * op(A) x + beta * y, A column-major [m x n] */
int aslp_blas_sgemv(aslp_blas_handle_t handle, aslp_blas_operation_t trans, int m, int n, float alpha, const float *A, int lda, const float *x,
                    int incx, float beta, float *y, int incy);
int aslp_blas_sdot(aslp_blas_handle_t handle, int n, const float *x, int incx, const float *y, int incy, float *result_host);
int aslp_blas_saxpy(aslp_blas_handle_t handle, int n, float alpha, const float *x, int incx, float *y, int incy);
int aslp_blas_sscal(aslp_blas_handle_t handle, int n, float alpha, float *x, int incx);
int aslp_blas_scopy(aslp_blas_handle_t handle, int n, const float *x, int incx, float *y, int incy);

#ifdef __cplusplus
}
#endif
#endif
