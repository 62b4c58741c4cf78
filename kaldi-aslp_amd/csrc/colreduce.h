// colreduce.h -- deterministic two-stage column reduction used by bias gradients, BatchNorm
// statistics and peephole gradients.
//
// The reference does these with cuBLAS gemv against a freshly allocated ones-vector
// (cu-vector.cc:1145-1166) or a one-thread-per-column loop.  Here stage 1 streams the matrix
// once: a wave covers 1 KiB of one row per instruction (64 lanes x float4) -- or 256 B on the
// scalar path -- 4 waves take 4 rows at a time, and the row range is split over enough blocks
// (>= ~512) to keep every CU's memory pipeline full; each block writes one partial per column.
// Stage 2 adds the partials in a fixed order and applies the op's epilogue, so results are
// bit-reproducible run to run (no float atomics).
#pragma once
#include "common.h"
#include "scratch.h"

namespace aslp {

constexpr int kColLanes = 64;   // lanes across columns (one wave)
constexpr int kRowLanes = 4;    // waves per block, each strides rows

// F: struct with  template <int VW> __device__ void operator()(int r, int c, T (&acc)[NOUT][VW]) const
//    (adds columns c..c+VW-1 of row r into acc); VW = 4 only when F::kVec and the launch is aligned.
template <int NOUT, int VW, class T, class F>
__global__ void __launch_bounds__(kColLanes *kRowLanes) colreduce_stage1(F f, int rows, int cols, int rows_per_chunk, T *partial) {
  __shared__ T red[kRowLanes][NOUT][VW][kColLanes];
  const int x = threadIdx.x, y = threadIdx.y;
  const int c = (blockIdx.x * kColLanes + x) * VW;
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  T acc[NOUT][VW];
#pragma unroll
  for (int k = 0; k < NOUT; k++)
#pragma unroll
    for (int v = 0; v < VW; v++) acc[k][v] = T(0);
  if (c < cols) {
    int r = r0 + y;
    // 4 independent rows in flight per lane
    for (; r + 3 * kRowLanes < r1; r += 4 * kRowLanes) {
      f.template operator()<VW>(r, c, acc);
      f.template operator()<VW>(r + kRowLanes, c, acc);
      f.template operator()<VW>(r + 2 * kRowLanes, c, acc);
      f.template operator()<VW>(r + 3 * kRowLanes, c, acc);
    }
    for (; r < r1; r += kRowLanes) f.template operator()<VW>(r, c, acc);
  }
#pragma unroll
  for (int k = 0; k < NOUT; k++)
#pragma unroll
    for (int v = 0; v < VW; v++) red[y][k][v][x] = acc[k][v];
  __syncthreads();
  if (y == 0 && c < cols) {
#pragma unroll
    for (int k = 0; k < NOUT; k++)
#pragma unroll
      for (int v = 0; v < VW; v++) {
        T s = red[0][k][v][x];
#pragma unroll
        for (int j = 1; j < kRowLanes; j++) s += red[j][k][v][x];
        partial[((long)blockIdx.y * NOUT + k) * cols + c + v] = s;
      }
  }
}

// G: struct with  __device__ void operator()(int c, const T (&sum)[NOUT]) const
// Finalize: 16 column lanes x 16 chunk lanes per workgroup (cols/16 workgroups: the pass is pure latency, so it
// wants many short dependent chains rather than few long ones); the per-lane sums are combined through LDS in
// chunk-lane order, so the result does not depend on scheduling.
constexpr int kFinCols = 16, kFinLanes = 16;
template <int NOUT, class T, class G>
__global__ void __launch_bounds__(kFinCols *kFinLanes) colreduce_stage2(G g, int chunks, int cols, const T *partial) {
  __shared__ T red[kFinLanes][NOUT][kFinCols];
  const int x = threadIdx.x, y = threadIdx.y;
  const int c = blockIdx.x * kFinCols + x;
  T sum[NOUT];
#pragma unroll
  for (int k = 0; k < NOUT; k++) sum[k] = T(0);
  if (c < cols) {
    int j = y;
    for (; j + 3 * kFinLanes < chunks; j += 4 * kFinLanes) {  // 4 independent loads in flight per output
      T v[4][NOUT];
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int k = 0; k < NOUT; k++) v[u][k] = partial[((long)(j + u * kFinLanes) * NOUT + k) * cols + c];
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int k = 0; k < NOUT; k++) sum[k] += v[u][k];
    }
    for (; j < chunks; j += kFinLanes)
#pragma unroll
      for (int k = 0; k < NOUT; k++) sum[k] += partial[((long)j * NOUT + k) * cols + c];
  }
#pragma unroll
  for (int k = 0; k < NOUT; k++) red[y][k][x] = sum[k];
  __syncthreads();
  if (y == 0 && c < cols) {
#pragma unroll
    for (int k = 0; k < NOUT; k++) {
      T s = red[0][k][x];
#pragma unroll
      for (int j = 1; j < kFinLanes; j++) s += red[j][k][x];
      sum[k] = s;
    }
    g(c, sum);
  }
}

// `vec_ok`: every matrix the functor reads is 16-byte aligned with a stride multiple of 4
template <int NOUT, class T, class F, class G>
void colreduce(const char *name, int rows, int cols, F f, G g, bool vec_ok = false, int slot = kScratchReduce) {
  if (cols <= 0) return;
  const bool vec = vec_ok && F::kVec && cols % 4 == 0;
  const int vw = vec ? 4 : 1;
  const int ctiles = (cols / vw + kColLanes - 1) / kColLanes;
  // aim for >= 512 blocks, at least 4*kRowLanes rows per block
  int want_chunks = (512 + ctiles - 1) / ctiles;
  int rpc = rows <= 0 ? 1 : (rows + want_chunks - 1) / want_chunks;
  if (rpc < 4 * kRowLanes) rpc = 4 * kRowLanes;
  rpc = (rpc + kRowLanes - 1) / kRowLanes * kRowLanes;
  const int chunks = rows <= 0 ? 1 : (rows + rpc - 1) / rpc;
  T *partial = static_cast<T *>(scratch(slot, sizeof(T) * (size_t)chunks * NOUT * cols));
  if (!partial) return;
  dim3 grid(ctiles, chunks), block(kColLanes, kRowLanes);
  if (vec) hipLaunchKernelGGL((colreduce_stage1<NOUT, 4, T, F>), grid, block, 0, cur_stream(), f, rows, cols, rpc, partial);
  else hipLaunchKernelGGL((colreduce_stage1<NOUT, 1, T, F>), grid, block, 0, cur_stream(), f, rows, cols, rpc, partial);
  hipLaunchKernelGGL((colreduce_stage2<NOUT, T, G>), dim3((cols + kFinCols - 1) / kFinCols), dim3(kFinCols, kFinLanes), 0, cur_stream(), g,
                     chunks, cols, partial);
  check_launch(name);
}

// load VW consecutive floats
template <int VW>
__device__ __forceinline__ void loadv(const float *p, float (&v)[VW]) {
  if (VW == 4) {
    float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1 % VW] = t.y; v[2 % VW] = t.z; v[3 % VW] = t.w;
  } else {
    v[0] = *p;
  }
}

}  // namespace aslp
