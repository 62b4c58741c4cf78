// gemm_common.h -- pieces shared by the fp32 MFMA GEMM kernels (gemm.hip, gemm_glds.hip).
#pragma once
#include <type_traits>

#include "aslp_kernels.h"
#include "common.h"

namespace aslp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) -- indices stay constant expressions,
// so register arrays indexed with them never decay to scratch memory
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>());
    static_for<B + 1, E>(f);
  }
}

struct GemmArgs {
  const float *A, *B;
  float *C;
  int M, N, K, lda, ldb, ldc;
  float alpha, beta;
  aslp_gemm_epilogue ep;
  int a_vec, b_vec;  // 16-byte vector loads allowed
  int tiles_m, tiles_n;
  // split-K (gemm_glds.hip): blockIdx.y = chunk of k_chunk (a multiple of the K tile) reduction steps, whose plain product
  // goes to C + chunk * split_stride (a scratch stack of packed M x N partials; the epilogue runs in the reduce kernel)
  int split_k, k_chunk;
  long split_stride;
};

// workgroup id -> (tile row, tile column): each XCD (id % 8, private 4 MB L2) gets a compact 2-D sub-grid of
// tiles when the grid divides evenly, else a contiguous run (bijective in both cases)
template <int BM, int BN>
__device__ __forceinline__ void xcd_tile(const GemmArgs &g, int &tm, int &tn) {
  const int nt = g.tiles_m * g.tiles_n;
  const int bid0 = blockIdx.x, xcd = bid0 % 8, j = bid0 / 8;
  int px = 0;
  long best = -1;
  for (int cand = 1; cand <= 8; cand *= 2) {
    const int py = 8 / cand;
    if (g.tiles_m % cand || g.tiles_n % py) continue;
    const long cost = (long)(g.tiles_m / cand) * BM + (long)(g.tiles_n / py) * BN;
    if (best < 0 || cost < best) { best = cost; px = cand; }
  }
  if (px > 0) {
    const int py = 8 / px, sn = g.tiles_n / py;
    tm = (xcd / py) * (g.tiles_m / px) + j / sn;
    tn = (xcd % py) * sn + j % sn;
  } else {
    const int q = nt / 8, r = nt % 8;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    tm = bid / g.tiles_n;
    tn = bid % g.tiles_n;
  }
}

// Fused epilogue on one wave's accumulators.  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (e&3) + 8*(e>>2) + 4*(lane>>5).  (row0, col0): global position of the wave patch.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, const f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int lh) {
  const aslp_gemm_epilogue &ep = g.ep;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++) {
      const int col = col0 + n * 32 + l31;
      if (col >= g.N) continue;
      const float bias = ep.bias ? ep.bias[col] : 0.0f;
      if (ep.W != nullptr || g.beta != 0.0f) {  // (uniform) outputs that READ memory: old C for beta, W for the fused SGD step
        // Everything the 16 outputs of this lane read is requested first: the compiler may not move a load of W above a
        // store to C on its own (the two could alias), and 16 load -> store round trips in a row are what the fused step
        // used to cost (TN weight gradient 79.4 -> 76.8 us).
        float c_old[16], w_old[16];
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          const bool ok = row < g.M;
          c_old[e] = (ok && g.beta != 0.0f) ? g.C[(long)row * g.ldc + col] : 0.0f;
          w_old[e] = (ok && ep.W) ? ep.W[(long)row * ep.ldw + col] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (row >= g.M) continue;
          float *cp = g.C + (long)row * g.ldc + col;
          float v = g.alpha * acc[i][n][e] + g.beta * c_old[e] + bias;
          if (ep.clip > 0.0f) v = fminf(fmaxf(v, -ep.clip), ep.clip);
          if (ep.W && g.beta == 0.0f) __builtin_nontemporal_store(v, cp);  // gradient written once, not read again this step
          else *cp = v;
          if (ep.W) ep.W[(long)row * ep.ldw + col] = w_old[e] + ep.w_alpha * v;
          if (ep.act_out) {
            float a = ep.act == 1 ? sigmoid_ref(v) : ep.act == 2 ? tanh_ref(v) : ep.act == 3 ? fmaxf(v, 0.0f) : v;
            ep.act_out[(long)row * ep.ld_act + col] = a;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (row >= g.M) continue;
          float v = g.alpha * acc[i][n][e] + bias;
          if (ep.clip > 0.0f) v = fminf(fmaxf(v, -ep.clip), ep.clip);
          g.C[(long)row * g.ldc + col] = v;
          if (ep.act_out) {
            float a = ep.act == 1 ? sigmoid_ref(v) : ep.act == 2 ? tanh_ref(v) : ep.act == 3 ? fmaxf(v, 0.0f) : v;
            ep.act_out[(long)row * ep.ld_act + col] = a;
          }
        }
      }
    }
}

// gemm_glds.hip: direct-to-LDS kernels.  Returns false if the problem is not eligible (caller falls back).
bool gemm_glds_launch(GemmArgs &g, bool a_kc, bool b_kc, int cfg);

}  // namespace aslp
