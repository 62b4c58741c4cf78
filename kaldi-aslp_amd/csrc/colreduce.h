// colreduce.h -- deterministic two-stage column reduction used by bias gradients, BatchNorm
// statistics and peephole gradients.
//
// The reference does these with cuBLAS gemv against a freshly allocated ones-vector
// (cu-vector.cc:1145-1166) or a one-thread-per-column loop; here stage 1 streams the matrix
// once, fully coalesced (a wave reads 256 contiguous bytes of a row, 4 waves = 4 rows), and
// writes per-chunk partials; stage 2 adds the chunks in a fixed order, so results are
// bit-reproducible run to run (no float atomics).
#pragma once
#include "common.h"
#include "scratch.h"

namespace aslp {

constexpr int kColTile = 64;        // columns per block (one per lane)
constexpr int kRowLanes = 4;        // waves per block, each strides rows
constexpr int kRowsPerChunk = 128;  // rows reduced by one block

// F: struct with  __device__ void operator()(int r, int c, T (&acc)[NOUT]) const  (adds into acc)
template <int NOUT, class T, class F>
__global__ void __launch_bounds__(kColTile *kRowLanes) colreduce_stage1(F f, int rows, int cols, T *partial) {
  __shared__ T red[kRowLanes][NOUT][kColTile];
  const int x = threadIdx.x, y = threadIdx.y;
  const int c = blockIdx.x * kColTile + x;
  const int r0 = blockIdx.y * kRowsPerChunk;
  int r1 = r0 + kRowsPerChunk;
  if (r1 > rows) r1 = rows;
  T acc[NOUT];
#pragma unroll
  for (int k = 0; k < NOUT; k++) acc[k] = T(0);
  if (c < cols)
    for (int r = r0 + y; r < r1; r += kRowLanes) f(r, c, acc);
#pragma unroll
  for (int k = 0; k < NOUT; k++) red[y][k][x] = acc[k];
  __syncthreads();
  if (y == 0 && c < cols) {
#pragma unroll
    for (int k = 0; k < NOUT; k++) {
      T s = red[0][k][x];
#pragma unroll
      for (int j = 1; j < kRowLanes; j++) s += red[j][k][x];
      partial[((long)blockIdx.y * NOUT + k) * cols + c] = s;
    }
  }
}

// G: struct with  __device__ void operator()(int c, const T (&sum)[NOUT]) const
template <int NOUT, class T, class G>
__global__ void __launch_bounds__(kBlock) colreduce_stage2(G g, int chunks, int cols, const T *partial) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  T sum[NOUT];
#pragma unroll
  for (int k = 0; k < NOUT; k++) sum[k] = T(0);
  for (int j = 0; j < chunks; j++)
#pragma unroll
    for (int k = 0; k < NOUT; k++) sum[k] += partial[((long)j * NOUT + k) * cols + c];
  g(c, sum);
}

template <int NOUT, class T, class F, class G>
void colreduce(const char *name, int rows, int cols, F f, G g, int slot = kScratchReduce) {
  if (cols <= 0) return;
  int chunks = rows <= 0 ? 1 : (rows + kRowsPerChunk - 1) / kRowsPerChunk;
  T *partial = static_cast<T *>(scratch(slot, sizeof(T) * (size_t)chunks * NOUT * cols));
  if (!partial) return;
  dim3 grid((cols + kColTile - 1) / kColTile, chunks), block(kColTile, kRowLanes);
  hipLaunchKernelGGL((colreduce_stage1<NOUT, T, F>), grid, block, 0, cur_stream(), f, rows, cols, partial);
  hipLaunchKernelGGL((colreduce_stage2<NOUT, T, G>), dim3((cols + kBlock - 1) / kBlock), dim3(kBlock), 0, cur_stream(), g,
                     chunks, cols, partial);
  check_launch(name);
}

}  // namespace aslp
