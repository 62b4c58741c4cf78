// rnn_mfma.h -- the skinny-product building blocks shared by the fused recurrent step kernels (rnn_fused.hip,
// gru_fused.hip): a 32 x 32 output tile (32 streams x 32 gate columns), K split over the 4 waves of a workgroup, both
// operands K-contiguous in global memory, partial tiles combined through LDS in wave order.
#pragma once
#include "common.h"

namespace aslp {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kPad = 33;  // LDS row pitch of a 32 x 32 partial tile

__device__ __forceinline__ float dsigm(float y, float d) { return d * y * (1.0f - y); }
__device__ __forceinline__ float dtanh(float y, float d) { return d * (1.0f - y * y); }

// acc += A[32 x K-slice] * B[32 x K-slice]^T over the 8-wide K-chunks [qbegin, qend): a CONTIGUOUS slice per wave,
// so every 128-B line of an operand row is pulled by exactly one wave.
// arow / brow: this lane's operand rows (lane & 31), K-contiguous, 16-B aligned; K % 4 == 0.
template <int U = 8>
__device__ __forceinline__ void mfma_k_slices(f32x16 &acc, const float *__restrict__ arow, const float *__restrict__ brow, int K, int qbegin,
                                              int qend, int h) {
  // U: 8-wide K-chunks in flight
  for (int q0 = qbegin; q0 < qend; q0 += U) {
    float4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int k = 8 * (q0 + u) + 4 * h;
      const bool ok = (q0 + u) < qend && k < K;
      const int kk = ok ? k : 0;
      a[u] = *reinterpret_cast<const float4 *>(arow + kk);
      b[u] = *reinterpret_cast<const float4 *>(brow + kk);
      if (!ok) { a[u] = make_float4(0.f, 0.f, 0.f, 0.f); b[u] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
  }
}

// C/D layout of v_mfma_f32_32x32x2f32: element e of lane l is row (e&3) + 8*(e>>2) + 4*(l>>5), column l&31
__device__ __forceinline__ void store_tile(float *tile, const f32x16 &acc, int lane) {
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int e = 0; e < 16; e++) tile[((e & 3) + 8 * (e >> 2) + 4 * h) * kPad + n] = acc[e];
}


}  // namespace
}  // namespace aslp
