// ew_kernels.hip -- elementwise + gather kernels of the CuMatrix substrate for gfx950.
//
// All of these are HBM-bound: one read + one write per
// element.  Layout rule: a lane moves 16 bytes (float4) whenever pointers and strides are
// 16-byte aligned, a wave therefore covers 1 KiB of one row per instruction; otherwise a
// scalar path with the same indexing.  Grids are capped at 256 CUs x 8 blocks and
// grid-strided.  Reference twins are cited by cu-kernels.cu line.
#include "aslp_kernels.h"
#include "common.h"
#include "split16.h"

namespace aslp {
namespace {

// dst[r][c] = f(dst[r][c], a[r][c], b[r][c], r, c)
template <class F>
__global__ void __launch_bounds__(kBlock) map_scalar(float *dst, int ldd, const float *a, int lda, const float *b,
                                                     int ldb, int rows, int cols, F f) {
  long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    float d = dst[(long)r * ldd + c];
    float av = a ? a[(long)r * lda + c] : 0.0f;
    float bv = b ? b[(long)r * ldb + c] : 0.0f;
    dst[(long)r * ldd + c] = f(d, av, bv, r, c);
  }
}

template <class F, bool READ_DST>
__global__ void __launch_bounds__(kBlock) map_vec4(float *dst, int ldd, const float *a, int lda, const float *b,
                                                   int ldb, int rows, int cols4, F f) {
  long n = (long)rows * cols4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / cols4), c = (int)(i - (long)r * cols4) * 4;
    float4 d = READ_DST ? *reinterpret_cast<const float4 *>(dst + (long)r * ldd + c) : make_float4(0, 0, 0, 0);
    float4 av = a ? *reinterpret_cast<const float4 *>(a + (long)r * lda + c) : make_float4(0, 0, 0, 0);
    float4 bv = b ? *reinterpret_cast<const float4 *>(b + (long)r * ldb + c) : make_float4(0, 0, 0, 0);
    float4 o;
    o.x = f(d.x, av.x, bv.x, r, c);
    o.y = f(d.y, av.y, bv.y, r, c + 1);
    o.z = f(d.z, av.z, bv.z, r, c + 2);
    o.w = f(d.w, av.w, bv.w, r, c + 3);
    *reinterpret_cast<float4 *>(dst + (long)r * ldd + c) = o;
  }
}

template <bool READ_DST, class F>
void launch_map(const char *name, float *dst, MatrixDim d, const float *a, int lda, const float *b, int ldb, F f) {
  if (d.rows <= 0 || d.cols <= 0) return;
  bool vec = (d.cols % 4 == 0) && (d.stride % 4 == 0) && aligned16(dst) && (!a || (lda % 4 == 0 && aligned16(a))) &&
             (!b || (ldb % 4 == 0 && aligned16(b)));
  if (vec) {
    long n = (long)d.rows * (d.cols / 4);
    hipLaunchKernelGGL((map_vec4<F, READ_DST>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), dst, d.stride, a,
                       lda, b, ldb, d.rows, d.cols / 4, f);
  } else {
    long n = (long)d.rows * d.cols;
    hipLaunchKernelGGL((map_scalar<F>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), dst, d.stride, a, lda, b,
                       ldb, d.rows, d.cols, f);
  }
  check_launch(name);
}

// ---- functors ------------------------------------------------------------------------
struct SetConst { float v; __device__ float operator()(float, float, float, int, int) const { return v; } };
struct AddConst { float v; __device__ float operator()(float d, float, float, int, int) const { return d + v; } };
struct Scale { float v; __device__ float operator()(float d, float, float, int, int) const { return d * v; } };
struct Log { __device__ float operator()(float d, float, float, int, int) const { return logf(d); } };
struct Exp { __device__ float operator()(float d, float, float, int, int) const { return expf(d); } };
struct Pow {  // cu-kernels.cu:1287 _apply_pow: 2 -> square, 0.5 -> sqrt (neg checked by caller), else pow
  float p;
  __device__ float operator()(float d, float, float, int, int) const {
    if (p == 1.0f) return d;
    if (p == 2.0f) return d * d;
    if (p == 0.5f) return sqrtf(d);
    return powf(d, p);
  }
};
struct Heaviside { __device__ float operator()(float d, float, float, int, int) const { return d > 0.0f ? 1.0f : 0.0f; } };
struct Floor { float v; __device__ float operator()(float d, float, float, int, int) const { return d < v ? v : d; } };
struct Ceil { float v; __device__ float operator()(float d, float, float, int, int) const { return d > v ? v : d; } };
struct Clamp { float lo, hi; __device__ float operator()(float d, float, float, int, int) const { return d < lo ? lo : (d > hi ? hi : d); } };
struct Invert { __device__ float operator()(float d, float, float, int, int) const { return 1.0f / d; } };
struct MulElem { __device__ float operator()(float d, float a, float, int, int) const { return d * a; } };
struct MulColsVec { const float *s; __device__ float operator()(float d, float, float, int, int c) const { return d * s[c]; } };
struct MulRowsVec { const float *s; __device__ float operator()(float d, float, float, int r, int) const { return d * s[r]; } };
struct AddMat { float alpha; __device__ float operator()(float d, float a, float, int, int) const { return alpha * a + d; } };
struct AddVecToCols { float alpha, beta; const float *v; __device__ float operator()(float d, float, float, int r, int) const { return alpha * v[r] + beta * d; } };
struct AddVecToRows { float alpha, beta; const float *v; __device__ float operator()(float d, float, float, int, int c) const { return alpha * v[c] + beta * d; } };
struct AddVecToRows0 { float alpha; const float *v; __device__ float operator()(float, float, float, int, int c) const { return alpha * v[c]; } };
struct AddMatDiagVec { float alpha, beta; const float *v; __device__ float operator()(float d, float a, float, int, int c) const { return alpha * a * v[c] + beta * d; } };
struct AddMatMatElem { float alpha, beta; __device__ float operator()(float d, float a, float b, int, int) const { return alpha * a * b + beta * d; } };
struct AddMatMatElem0 { float alpha; __device__ float operator()(float, float a, float b, int, int) const { return alpha * a * b; } };
struct Sigmoid { __device__ float operator()(float, float a, float, int, int) const { return sigmoid_ref(a); } };
struct Tanh { __device__ float operator()(float, float a, float, int, int) const { return tanh_ref(a); } };
// matrix/kaldi-matrix.cc:2713-2744
struct DiffSigmoid { __device__ float operator()(float, float e, float y, int, int) const { return e * y * (1.0f - y); } };
struct DiffTanh { __device__ float operator()(float, float e, float y, int, int) const { return e * (1.0f - y * y); } };
struct DiffRelu { __device__ float operator()(float, float in, float od, int, int) const { return in > 0.0f ? od : 0.0f; } };
struct CopyMat { __device__ float operator()(float, float a, float, int, int) const { return a; } };

// Sigmoid backward which also leaves the fp16 planes of its result (csrc/split16.h): |e y (1 - y)| <= max |e| / 4, and the maxima of |e| come
// with e from the kernel that wrote it (n_parts per-workgroup values), so the scale is known before the first element is written.  The
// arithmetic of an element is DiffSigmoid's.  Workgroup 0 stores the bound for the planes' readers.
__global__ void __launch_bounds__(kBlock) diff_sigmoid_planes_kernel(float *__restrict__ eout, int ldo, const float *__restrict__ e, int lde,
                                                                     const float *__restrict__ y, int ldy, int rows, int cols4,
                                                                     const float *__restrict__ e_max_parts, int n_parts, S16Out po) {
  __shared__ float wm[kBlock / 64];
  float m = 0.f;
  for (int i = threadIdx.x; i < n_parts; i += kBlock) m = fmaxf(m, e_max_parts[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  m = wm[0];
  for (int w = 1; w < kBlock / 64; w++) m = fmaxf(m, wm[w]);
  const float bound = 0.25f * m;
  if (blockIdx.x == 0 && threadIdx.x == 0) *const_cast<unsigned *>(po.slot) = __float_as_uint(bound);
  const float s = ldexpf(1.f, s16_exponent(__float_as_uint(bound)));
  const DiffSigmoid f{};
  const long n = (long)rows * cols4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols4), c = (int)(i - (long)r * cols4) * 4;
    const float4 ev = *reinterpret_cast<const float4 *>(e + (long)r * lde + c), yv = *reinterpret_cast<const float4 *>(y + (long)r * ldy + c);
    const float4 o = make_float4(f(0.f, ev.x, yv.x, r, c), f(0.f, ev.y, yv.y, r, c + 1), f(0.f, ev.z, yv.z, r, c + 2), f(0.f, ev.w, yv.w, r, c + 3));
    *reinterpret_cast<float4 *>(eout + (long)r * ldo + c) = o;
    half4 hi, lo;
    s16_split4(o, s, &hi, &lo);
    *reinterpret_cast<half4 *>(po.hi + (long)r * po.ld + c) = hi;
    *reinterpret_cast<half4 *>(po.lo + (long)r * po.ld + c) = lo;
  }
}

// transposed AddMat (rare; cu-kernels.cu:584 A_trans branch)
__global__ void add_mat_trans_kernel(float alpha, const float *src, float *dst, MatrixDim d, int src_stride) {
  long n = (long)d.rows * d.cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / d.cols), c = (int)(i % d.cols);
    dst[(long)r * d.stride + c] += alpha * src[(long)c * src_stride + r];
  }
}

// dst = src^T (no zeroing launch in front of an add: the recurrent layers refresh K-contiguous copies of their weights once per backward pass).
// 32 x 32 tiles through LDS (pitch 33: the column reads fall on 32 different banks): both the reads of src and the writes of dst are
// 128-byte rows -- a thread-per-element version read src with a stride of a whole row per lane.  dst is [d.rows x d.cols], src [d.cols x d.rows].
__global__ void __launch_bounds__(256) copy_mat_trans_kernel(const float *src, float *dst, MatrixDim d, int src_stride) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;     // tile of dst: rows r0 .., columns c0 ..
#pragma unroll
  for (int j = 0; j < 4; j++) {   // src rows = dst columns c0 + ty + 8 j, src columns = dst rows r0 + tx
    const int sc = r0 + tx, sr = c0 + ty + 8 * j;
    if (sr < d.cols && sc < d.rows) tile[ty + 8 * j][tx] = src[(long)sr * src_stride + sc];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    if (r < d.rows && c < d.cols) dst[(long)r * d.stride + c] = tile[tx][ty + 8 * j];
  }
}

// _add_mat_diag_vec with a transposed mat2 (generic strides)
__global__ void add_mat_diag_vec_strided(float alpha, float *mat, MatrixDim d, const float *mat2, int rs, int cs,
                                         const float *vec, float beta) {
  long n = (long)d.rows * d.cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / d.cols), c = (int)(i % d.cols);
    mat[(long)r * d.stride + c] = alpha * mat2[(long)r * rs + (long)c * cs] * vec[c] + beta * mat[(long)r * d.stride + c];
  }
}

// cu-math.cc:37-75 / cu-kernels.cu:2113 _regularize_l1
__global__ void regularize_l1_kernel(float *wei, float *grad, float l1, float lr, MatrixDim d, int stride_grad) {
  long n = (long)d.rows * d.cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / d.cols), c = (int)(i % d.cols);
    long wi = (long)r * d.stride + c, gi = (long)r * stride_grad + c;
    float w = wei[wi];
    if (w == 0.0f) continue;
    float l1_signed = w < 0.0f ? -l1 : l1;
    float after = w - lr * grad[gi] - l1_signed;
    if ((after > 0.0f) ^ (w > 0.0f)) {
      wei[wi] = 0.0f;
      grad[gi] = 0.0f;
    } else {
      wei[wi] = w - l1_signed;
    }
  }
}

// ---- gathers (bit-exact copies) -------------------------------------------------------
// Row gather: dst[r][:] = src[idx[r]][:] (idx < 0 -> zeros, cu-kernels.cu:1413 _copy_rows), optionally
// accumulate (cu-kernels.cu:1462 _add_rows).  One float4 per lane when aligned.
template <bool ADD, bool VEC>
__global__ void __launch_bounds__(kBlock) row_gather(float alpha, float *dst, const float *src, const int32_t *idx, int rows,
                                                     int cols, int ldd, int lds) {
  constexpr int W = VEC ? 4 : 1;
  int cw = cols / W;
  long n = (long)rows * cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / cw), c = (int)(i - (long)r * cw) * W;
    int sr = idx[r];
    if (VEC) {
      float4 v = sr < 0 ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4 *>(src + (long)sr * lds + c);
      float4 *dp = reinterpret_cast<float4 *>(dst + (long)r * ldd + c);
      if (ADD) {
        float4 o = *dp;
        o.x += alpha * v.x; o.y += alpha * v.y; o.z += alpha * v.z; o.w += alpha * v.w;
        *dp = o;
      } else {
        *dp = v;
      }
    } else {
      float v = sr < 0 ? 0.0f : src[(long)sr * lds + c];
      if (ADD) dst[(long)r * ldd + c] += alpha * v; else dst[(long)r * ldd + c] = v;
    }
  }
}

template <bool ADD>
void launch_row_gather(const char *name, float alpha, float *dst, const float *src, const int32_t *idx, int rows, int cols,
                       int ldd, int lds) {
  if (rows <= 0 || cols <= 0) return;
  bool vec = cols % 4 == 0 && ldd % 4 == 0 && lds % 4 == 0 && aligned16(dst) && aligned16(src);
  long n = (long)rows * (vec ? cols / 4 : cols);
  if (vec)
    hipLaunchKernelGGL((row_gather<ADD, true>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), alpha, dst, src, idx,
                       rows, cols, ldd, lds);
  else
    hipLaunchKernelGGL((row_gather<ADD, false>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), alpha, dst, src, idx,
                       rows, cols, ldd, lds);
  check_launch(name);
}

// Column gather: dst[r][c] = src[r][reorder[c]] (reorder < 0 -> 0; cu-kernels.cu:1373/1394, 2075 _copy)
template <bool ADD>
__global__ void __launch_bounds__(kBlock) col_gather(float *dst, const float *src, const int32_t *reorder, int rows, int cols,
                                                     int ldd, int lds) {
  long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    int sc = reorder[c];
    float v = sc < 0 ? 0.0f : src[(long)r * lds + sc];
    if (ADD) dst[(long)r * ldd + c] += v; else dst[(long)r * ldd + c] = v;
  }
}

// Splice (cu-kernels.cu:1998 _splice): y[r][k*D + c] = x[clamp(r + off[k], 0, R-1)][c].
// One work item = one float4 (or one float) of the OUTPUT row, so writes are fully coalesced
// and each input row segment is re-read from L2 by the n_off consumers.
template <bool VEC>
__global__ void __launch_bounds__(kBlock) splice_kernel(float *y, const float *x, const int32_t *off, int rows, int out_cols,
                                                        int in_cols, int ldy, int ldx) {
  constexpr int W = VEC ? 4 : 1;
  int cw = out_cols / W;
  long n = (long)rows * cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / cw), oc = (int)(i - (long)r * cw) * W;
    int k = oc / in_cols, c = oc - k * in_cols;
    int sr = r + off[k];
    sr = sr < 0 ? 0 : (sr >= rows ? rows - 1 : sr);
    if (VEC)
      *reinterpret_cast<float4 *>(y + (long)r * ldy + oc) = *reinterpret_cast<const float4 *>(x + (long)sr * ldx + c);
    else
      y[(long)r * ldy + oc] = x[(long)sr * ldx + c];
  }
}

// Splice backward as the reference computes it (nnet-various.h:143-175).
template <bool VEC>
__global__ void __launch_bounds__(kBlock) splice_bwd_kernel(float *in_diff, const float *od, const int32_t *off, int n_off,
                                                            int rows, int in_cols, int ldi, int ldo) {
  constexpr int W = VEC ? 4 : 1;
  int cw = in_cols / W;
  long n = (long)rows * cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int t = (int)(i / cw), c = (int)(i - (long)t * cw) * W;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int k = 0; k < n_off; k++) {
      int sr = t + off[k];
      sr = sr < 0 ? 0 : (sr >= rows ? rows - 1 : sr);
      const float *p = od + (long)sr * ldo + (long)k * in_cols + c;
      if (VEC) {
        float4 v = *reinterpret_cast<const float4 *>(p);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      } else {
        acc.x += *p;
      }
    }
    if (VEC) *reinterpret_cast<float4 *>(in_diff + (long)t * ldi + c) = acc;
    else in_diff[(long)t * ldi + c] = acc.x;
  }
}

__global__ void set_const_i32(int32_t *mat, int32_t value, MatrixDim d) {
  long n = (long)d.rows * d.cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    mat[(i / d.cols) * d.stride + i % d.cols] = value;
}

__global__ void scatter_add_kernel(float *mat, MatrixDim d, const int32_t *rows, const int32_t *cols, const float *vals, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int r = rows[i], c = cols[i];
    if (r >= 0 && r < d.rows && c >= 0 && c < d.cols) atomicAdd(mat + (long)r * d.stride + c, vals[i]);
  }
}

__global__ void f2d_kernel(double *dst, const float *src, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = (double)src[i];
}
__global__ void d2f_kernel(float *dst, const double *src, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
__global__ void add_vec_vec_kernel(float alpha, float *v, const float *x, const float *y, float beta, int dim) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x)
    v[i] = alpha * x[i] * y[i] + beta * v[i];
}
__global__ void axpy2_kernel(float alpha, const float *x1, float *y1, const float *x2, float *y2, int dim) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x) {
    y1[i] += alpha * x1[i];
    y2[i] += alpha * x2[i];
  }
}
__global__ void axpy_kernel(float alpha, const float *x, float *y, int dim) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

// ---- Dropout (nnet-activation.h:240-258) ---------------------------------------------------------------------
// mask(r, c) = [u(r, c) < retention], u from a counter-based generator (one 64-bit mix of (seed, element index): no state
// to carry, any launch geometry gives the same mask); out = in * mask / retention.  The reference draws its mask from
// CuRand (five launches); the masks are different random streams, the distribution is the same.
__device__ __forceinline__ float dropout_uniform(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);  // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);  // 24 bits -> [0, 1)
}
__global__ void dropout_fwd_kernel(float *out, int ldo, const float *in, int ldi, float *mask, int ldm, int rows, int cols, float retention,
                                   unsigned long long seed) {
  const long n = (long)rows * cols;
  const float inv = 1.0f / retention;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    const float m = dropout_uniform(seed, (unsigned long long)i) < retention ? 1.0f : 0.0f;
    mask[(long)r * ldm + c] = m;
    out[(long)r * ldo + c] = in[(long)r * ldi + c] * m * inv;
  }
}
__global__ void dropout_bwd_kernel(float *in_diff, int ldid, const float *od, int ldod, const float *mask, int ldm, int rows, int cols,
                                   float retention) {
  const long n = (long)rows * cols;
  const float inv = 1.0f / retention;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    in_diff[(long)r * ldid + c] = od[(long)r * ldod + c] * mask[(long)r * ldm + c] * inv;
  }
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

void cudaF_set_const(aslp_dim3, aslp_dim3, float *mat, float value, MatrixDim d) { launch_map<false>("set_const", mat, d, nullptr, 0, nullptr, 0, SetConst{value}); }
void cudaF_add(aslp_dim3, aslp_dim3, float *mat, float value, MatrixDim d) { launch_map<true>("add", mat, d, nullptr, 0, nullptr, 0, AddConst{value}); }
void cudaF_scale(aslp_dim3, aslp_dim3, float *mat, float value, MatrixDim d) { launch_map<true>("scale", mat, d, nullptr, 0, nullptr, 0, Scale{value}); }
void cudaF_apply_log(aslp_dim3, aslp_dim3, float *mat, MatrixDim d) { launch_map<true>("apply_log", mat, d, nullptr, 0, nullptr, 0, Log{}); }
void cudaF_apply_exp(aslp_dim3, aslp_dim3, float *mat, MatrixDim d) { launch_map<true>("apply_exp", mat, d, nullptr, 0, nullptr, 0, Exp{}); }
void cudaF_apply_pow(aslp_dim3, aslp_dim3, float *mat, float power, MatrixDim d) { launch_map<true>("apply_pow", mat, d, nullptr, 0, nullptr, 0, Pow{power}); }
void cudaF_apply_heaviside(aslp_dim3, aslp_dim3, float *mat, MatrixDim d) { launch_map<true>("apply_heaviside", mat, d, nullptr, 0, nullptr, 0, Heaviside{}); }
void aslp_apply_clamp(float *mat, MatrixDim d, float lo, float hi) { launch_map<true>("apply_clamp", mat, d, nullptr, 0, nullptr, 0, Clamp{lo, hi}); }
void cudaF_apply_floor(aslp_dim3, aslp_dim3, float *mat, float v, MatrixDim d) { launch_map<true>("apply_floor", mat, d, nullptr, 0, nullptr, 0, Floor{v}); }
void cudaF_apply_ceiling(aslp_dim3, aslp_dim3, float *mat, float v, MatrixDim d) { launch_map<true>("apply_ceiling", mat, d, nullptr, 0, nullptr, 0, Ceil{v}); }
void cudaF_invert_elements(aslp_dim3, aslp_dim3, float *data, MatrixDim d) { launch_map<true>("invert_elements", data, d, nullptr, 0, nullptr, 0, Invert{}); }
void cudaF_mul_elements(aslp_dim3, aslp_dim3, float *mat, const float *A, MatrixDim d, int src_stride) { launch_map<true>("mul_elements", mat, d, A, src_stride, nullptr, 0, MulElem{}); }
void cudaF_mul_cols_vec(aslp_dim3, aslp_dim3, float *mat, const float *scale, MatrixDim d) { launch_map<true>("mul_cols_vec", mat, d, nullptr, 0, nullptr, 0, MulColsVec{scale}); }
void cudaF_mul_rows_vec(aslp_dim3, aslp_dim3, float *mat, const float *scale, MatrixDim d) { launch_map<true>("mul_rows_vec", mat, d, nullptr, 0, nullptr, 0, MulRowsVec{scale}); }

void cudaF_add_mat(aslp_dim3, aslp_dim3, float alpha, const float *src, float *dst, MatrixDim d, int src_stride, int A_trans) {
  if (!A_trans) {
    launch_map<true>("add_mat", dst, d, src, src_stride, nullptr, 0, AddMat{alpha});
  } else {
    if (d.rows <= 0 || d.cols <= 0) return;
    hipLaunchKernelGGL(add_mat_trans_kernel, dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), alpha, src, dst, d, src_stride);
    check_launch("add_mat_trans");
  }
}
void cudaF_add_vec_to_cols(aslp_dim3, aslp_dim3, float alpha, const float *col, float beta, float *dst, MatrixDim d) {
  launch_map<true>("add_vec_to_cols", dst, d, nullptr, 0, nullptr, 0, AddVecToCols{alpha, beta, col});
}
void cudaF_add_vec_to_rows(aslp_dim3, aslp_dim3, float alpha, const float *row, float beta, float *dst, MatrixDim d) {
  // beta == 0 must not read dst (the reference pre-zeroes; NaN garbage would otherwise leak through 0*NaN)
  if (beta == 0.0f) launch_map<false>("add_vec_to_rows", dst, d, nullptr, 0, nullptr, 0, AddVecToRows0{alpha, row});
  else launch_map<true>("add_vec_to_rows", dst, d, nullptr, 0, nullptr, 0, AddVecToRows{alpha, beta, row});
}
void cudaF_add_mat_diag_vec(aslp_dim3, aslp_dim3, float alpha, float *mat, MatrixDim mat_dim, const float *mat2, int mat2_row_stride,
                            int mat2_col_stride, const float *vec, float beta) {
  if (mat2_col_stride == 1) {
    launch_map<true>("add_mat_diag_vec", mat, mat_dim, mat2, mat2_row_stride, nullptr, 0, AddMatDiagVec{alpha, beta, vec});
  } else {
    if (mat_dim.rows <= 0 || mat_dim.cols <= 0) return;
    hipLaunchKernelGGL(add_mat_diag_vec_strided, dim3(grid_for((long)mat_dim.rows * mat_dim.cols)), dim3(kBlock), 0, cur_stream(),
                       alpha, mat, mat_dim, mat2, mat2_row_stride, mat2_col_stride, vec, beta);
    check_launch("add_mat_diag_vec_strided");
  }
}
void cudaF_add_mat_mat_elements(aslp_dim3, aslp_dim3, float *data, const float *A, const float *B, MatrixDim dim, int sa, int sb, float alpha, float beta) {
  if (beta == 0.0f) launch_map<false>("add_mat_mat_elements", data, dim, A, sa, B, sb, AddMatMatElem0{alpha});
  else launch_map<true>("add_mat_mat_elements", data, dim, A, sa, B, sb, AddMatMatElem{alpha, beta});
}
void cudaF_sigmoid(aslp_dim3, aslp_dim3, float *y, const float *x, MatrixDim d, int src_stride) { launch_map<false>("sigmoid", y, d, x, src_stride, nullptr, 0, Sigmoid{}); }
void cudaF_tanh(aslp_dim3, aslp_dim3, float *y, const float *x, MatrixDim d, int src_stride) { launch_map<false>("tanh", y, d, x, src_stride, nullptr, 0, Tanh{}); }
void cudaF_diff_sigmoid(aslp_dim3, aslp_dim3, float *eout, const float *e, const float *y, MatrixDim d, int e_stride, int y_stride) {
  launch_map<false>("diff_sigmoid", eout, d, e, e_stride, y, y_stride, DiffSigmoid{});
}
int aslp_diff_sigmoid_p(float *eout, const float *e, const float *y, MatrixDim d, int e_stride, int y_stride, const float *e_max_parts, int n_parts,
                        const aslp_planes_out *out_planes) {
  const bool vec = d.rows > 0 && d.cols > 0 && d.cols % 4 == 0 && d.stride % 4 == 0 && e_stride % 4 == 0 && y_stride % 4 == 0 && aligned16(eout) &&
                   aligned16(e) && aligned16(y);
  if (!vec || !e_max_parts || n_parts <= 0 || !out_planes || !out_planes->hi || !out_planes->slot || out_planes->ld < d.cols) {
    cudaF_diff_sigmoid(aslp_dim3(), aslp_dim3(), eout, e, y, d, e_stride, y_stride);
    return 0;
  }
  S16Out po = {static_cast<h16 *>(out_planes->hi), static_cast<h16 *>(out_planes->lo), out_planes->ld, out_planes->slot, nullptr};
  const long n = (long)d.rows * (d.cols / 4);
  hipLaunchKernelGGL(diff_sigmoid_planes_kernel, dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), eout, d.stride, e, e_stride, y, y_stride, d.rows,
                     d.cols / 4, e_max_parts, n_parts, po);
  check_launch("diff_sigmoid_planes");
  return 1;
}
void cudaF_diff_tanh(aslp_dim3, aslp_dim3, float *eout, const float *e, const float *y, MatrixDim d, int e_stride, int y_stride) {
  launch_map<false>("diff_tanh", eout, d, e, e_stride, y, y_stride, DiffTanh{});
}
void aslp_diff_relu(float *in_diff, const float *in, const float *out_diff, MatrixDim d, int in_stride, int od_stride) {
  launch_map<false>("diff_relu", in_diff, d, in, in_stride, out_diff, od_stride, DiffRelu{});
}
void aslp_copy_mat(float *dst, MatrixDim d, const float *src, int src_stride) { launch_map<false>("copy_mat", dst, d, src, src_stride, nullptr, 0, CopyMat{}); }
void aslp_copy_mat_trans(float *dst, MatrixDim d, const float *src, int src_stride) {
  if (d.rows <= 0 || d.cols <= 0) return;
  hipLaunchKernelGGL(copy_mat_trans_kernel, dim3((d.cols + 31) / 32, (d.rows + 31) / 32), dim3(256), 0, cur_stream(), src, dst, d, src_stride);
  check_launch("copy_mat_trans");
}

void cudaF_regularize_l1(aslp_dim3, aslp_dim3, float *wei, float *grad, float l1, float lr, MatrixDim d, int stride_grad) {
  if (d.rows <= 0 || d.cols <= 0) return;
  hipLaunchKernelGGL(regularize_l1_kernel, dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), wei, grad, l1, lr, d, stride_grad);
  check_launch("regularize_l1");
}

void cudaF_copy_rows(aslp_dim3, aslp_dim3, float *dst, const float *src, const MatrixIndexT_cuda *reorder, MatrixDim dd, int src_stride) {
  launch_row_gather<false>("copy_rows", 0.0f, dst, src, reorder, dd.rows, dd.cols, dd.stride, src_stride);
}
void cudaF_add_rows(aslp_dim3, aslp_dim3, float alpha, float *dst, const float *src, const MatrixIndexT_cuda *reorder, MatrixDim dd, int src_stride) {
  launch_row_gather<true>("add_rows", alpha, dst, src, reorder, dd.rows, dd.cols, dd.stride, src_stride);
}
void cudaF_randomize(aslp_dim3, aslp_dim3, float *y, const float *x, const int32_cuda *copy_from, MatrixDim d_out, MatrixDim d_in) {
  // cu-math.cc:105-110: d_out.rows = number of indices
  launch_row_gather<false>("randomize", 0.0f, y, x, copy_from, d_out.rows, d_out.cols, d_out.stride, d_in.stride);
}
void cudaF_copy_cols(aslp_dim3, aslp_dim3, float *dst, const float *src, const MatrixIndexT_cuda *reorder, MatrixDim dd, int src_stride) {
  if (dd.rows <= 0 || dd.cols <= 0) return;
  hipLaunchKernelGGL((col_gather<false>), dim3(grid_for((long)dd.rows * dd.cols)), dim3(kBlock), 0, cur_stream(), dst, src, reorder, dd.rows, dd.cols, dd.stride, src_stride);
  check_launch("copy_cols");
}
void cudaF_add_cols(aslp_dim3, aslp_dim3, float *dst, const float *src, const MatrixIndexT_cuda *reorder, MatrixDim dd, int src_stride) {
  if (dd.rows <= 0 || dd.cols <= 0) return;
  hipLaunchKernelGGL((col_gather<true>), dim3(grid_for((long)dd.rows * dd.cols)), dim3(kBlock), 0, cur_stream(), dst, src, reorder, dd.rows, dd.cols, dd.stride, src_stride);
  check_launch("add_cols");
}
void cudaF_copy(aslp_dim3, aslp_dim3, float *y, const float *x, const int32_cuda *copy_from, MatrixDim d_out, MatrixDim d_in) {
  if (d_out.rows <= 0 || d_out.cols <= 0) return;
  hipLaunchKernelGGL((col_gather<false>), dim3(grid_for((long)d_out.rows * d_out.cols)), dim3(kBlock), 0, cur_stream(), y, x, copy_from, d_out.rows, d_out.cols, d_out.stride, d_in.stride);
  check_launch("copy");
}
void cudaF_splice(aslp_dim3, aslp_dim3, float *y, const float *x, const int32_cuda *off, MatrixDim d_out, MatrixDim d_in) {
  if (d_out.rows <= 0 || d_out.cols <= 0) return;
  bool vec = d_in.cols % 4 == 0 && d_in.stride % 4 == 0 && d_out.stride % 4 == 0 && aligned16(y) && aligned16(x);
  long n = (long)d_out.rows * (vec ? d_out.cols / 4 : d_out.cols);
  if (vec) hipLaunchKernelGGL((splice_kernel<true>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), y, x, off, d_out.rows, d_out.cols, d_in.cols, d_out.stride, d_in.stride);
  else hipLaunchKernelGGL((splice_kernel<false>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), y, x, off, d_out.rows, d_out.cols, d_in.cols, d_out.stride, d_in.stride);
  check_launch("splice");
}
void aslp_splice_backward(float *in_diff, MatrixDim d_in, const float *out_diff, int od_stride, const int32_cuda *off, int n_off) {
  if (d_in.rows <= 0 || d_in.cols <= 0) return;
  bool vec = d_in.cols % 4 == 0 && d_in.stride % 4 == 0 && od_stride % 4 == 0 && aligned16(in_diff) && aligned16(out_diff);
  long n = (long)d_in.rows * (vec ? d_in.cols / 4 : d_in.cols);
  if (vec) hipLaunchKernelGGL((splice_bwd_kernel<true>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), in_diff, out_diff, off, n_off, d_in.rows, d_in.cols, d_in.stride, od_stride);
  else hipLaunchKernelGGL((splice_bwd_kernel<false>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), in_diff, out_diff, off, n_off, d_in.rows, d_in.cols, d_in.stride, od_stride);
  check_launch("splice_backward");
}
void cudaI32_set_const(aslp_dim3, aslp_dim3, int32_cuda *mat, int32_cuda value, MatrixDim d) {
  if (d.rows <= 0 || d.cols <= 0) return;
  hipLaunchKernelGGL(set_const_i32, dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), mat, value, d);
  check_launch("set_const_i32");
}
void aslp_scatter_add(float *mat, MatrixDim d, const int32_cuda *rows, const int32_cuda *cols, const float *vals, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_add_kernel, dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), mat, d, rows, cols, vals, n);
  check_launch("scatter_add");
}
void aslp_f2d(double *dst, const float *src, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(f2d_kernel, dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), dst, src, n);
  check_launch("f2d");
}
void aslp_d2f(float *dst, const double *src, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(d2f_kernel, dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), dst, src, n);
  check_launch("d2f");
}
void cudaF_add_vec_vec(int, int, float alpha, float *v, const float *x, const float *y, float beta, int dim) {
  if (dim <= 0) return;
  hipLaunchKernelGGL(add_vec_vec_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, cur_stream(), alpha, v, x, y, beta, dim);
  check_launch("add_vec_vec");
}
void aslp_vec_axpy2(float alpha, const float *x1, float *y1, const float *x2, float *y2, int dim) {
  if (dim <= 0) return;
  hipLaunchKernelGGL(axpy2_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, cur_stream(), alpha, x1, y1, x2, y2, dim);
  check_launch("axpy2");
}
void aslp_dropout_forward(float *out, int out_stride, const float *in, MatrixDim d, float *mask, int mask_stride, float retention,
                          unsigned long long seed) {
  if (d.rows <= 0 || d.cols <= 0) return;
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), out, out_stride, in, d.stride, mask,
                     mask_stride, d.rows, d.cols, retention, seed);
  check_launch("dropout_forward");
}
void aslp_dropout_backward(float *in_diff, int id_stride, const float *out_diff, MatrixDim d, const float *mask, int mask_stride, float retention) {
  if (d.rows <= 0 || d.cols <= 0) return;
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for((long)d.rows * d.cols)), dim3(kBlock), 0, cur_stream(), in_diff, id_stride, out_diff, d.stride,
                     mask, mask_stride, d.rows, d.cols, retention);
  check_launch("dropout_backward");
}
void aslp_vec_axpy(float alpha, const float *x, float *y, int dim) {
  if (dim <= 0) return;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(dim)), dim3(kBlock), 0, cur_stream(), alpha, x, y, dim);
  check_launch("axpy");
}

}  // extern "C"
