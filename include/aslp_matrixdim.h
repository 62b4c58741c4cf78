/*
 * aslp_matrixdim.h -- the two POD argument types of the kernel C ABI.
 *
 * Replaces: src/aslp-cudamatrix/cu-matrixdim.h:52-56 (MatrixDim, passed by value to
 * every cudaF_* wrapper) and CUDA's dim3 (launch geometry the reference's CuMatrix
 * passes in; this library accepts and ignores it -- it picks its own CDNA4 geometry).
 */
#ifndef ASLP_MATRIXDIM_H_
#define ASLP_MATRIXDIM_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct MatrixDim_ {
  int32_t rows;
  int32_t cols;
  int32_t stride; /* in elements, >= cols */
} MatrixDim;

/* layout-compatible with CUDA/HIP dim3 (three 32-bit unsigned, passed by value) */
typedef struct aslp_dim3_ {
  uint32_t x, y, z;
} aslp_dim3;

typedef int32_t int32_cuda;
typedef int32_t MatrixIndexT_cuda;

#ifdef __cplusplus
}
#endif
#endif
