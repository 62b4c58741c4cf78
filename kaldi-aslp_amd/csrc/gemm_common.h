// gemm_common.h -- pieces shared by the fp32 MFMA GEMM kernels (gemm.hip, gemm_glds.hip).
#pragma once
#include <type_traits>

#include "aslp_kernels.h"
#include "common.h"
#include "split16.h"

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
  int wide_epilogue;  // 16-byte epilogue accesses through the LDS transpose (A/B switch: ASLP_GEMM_WIDE_EPI=0)
  int tiles_m, tiles_n;
  // split-K (gemm_glds.hip): blockIdx.y = chunk of k_chunk (a multiple of the K tile) reduction steps, whose plain product
  // goes to C + chunk * split_stride (a scratch stack of packed M x N partials; the epilogue runs in the reduce kernel)
  int split_k, k_chunk;
  long split_stride;
  // pair (gemm_glds.hip, aslp_sgemm_pair_ex): a second product of the same shape, leading dimensions, alpha and beta in the same
  // launch -- blockIdx.z == 1 works on (A1, B1, C1, ep1).  The two directions of a bidirectional recurrent layer issue every
  // one of their batched products twice with different operands; most of those grids cannot fill the chip alone.
  int pair;
  const float *A1, *B1;
  float *C1;
  aslp_gemm_epilogue ep1;
};

// workgroup id -> (tile row, tile column): each XCD (id % 8, private 4 MB L2) gets a compact 2-D sub-grid of
// tiles when the grid divides evenly, else a contiguous run (bijective in both cases)
template <int BM, int BN>
__device__ __forceinline__ void xcd_tile(const GemmArgs &g, int &tm, int &tn, int bid0 = blockIdx.x) {
  const int nt = g.tiles_m * g.tiles_n;
  const int xcd = bid0 % 8, j = bid0 / 8;
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

// What an epilogue leaves for the split-fp16 products that will read its output (aslp_gemm_epilogue.planes / *_parts), per wave:
// the scale of the planes it writes (0: none) and the running maxima of |W after the step| and |C|.
struct EpiExtra {
  float pscale = 0.f;
  float wmax = 0.f, cmax = 0.f;
};
__device__ __forceinline__ float epi_finite_abs(float v) { const float a = fabsf(v); return a < 3.0e38f ? a : 0.f; }
__device__ __forceinline__ void epi_plane_store(const aslp_gemm_epilogue &ep, float scale, float v, long row, int col) {
  h16 h, l;
  s16_split(v, scale, &h, &l);
  static_cast<h16 *>(ep.planes.hi)[row * ep.planes.ld + col] = h;
  static_cast<h16 *>(ep.planes.lo)[row * ep.planes.ld + col] = l;
}
__device__ __forceinline__ void epi_plane_store4(const aslp_gemm_epilogue &ep, float scale, const float4 v, long row, int col) {
  half4 h, l;
  s16_split4(v, scale, &h, &l);
  *reinterpret_cast<half4 *>(static_cast<h16 *>(ep.planes.hi) + row * ep.planes.ld + col) = h;
  *reinterpret_cast<half4 *>(static_cast<h16 *>(ep.planes.lo) + row * ep.planes.ld + col) = l;
}

// Fused epilogue on one wave's accumulators.  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (e&3) + 8*(e>>2) + 4*(lane>>5).  (row0, col0): global position of the wave patch.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, const f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int lh, EpiExtra &x, bool track) {
  const aslp_gemm_epilogue &ep = g.ep;
  const bool w_planes = track && x.pscale != 0.f && ep.planes_of == 1 && ep.W, a_planes = track && x.pscale != 0.f && ep.planes_of == 2 && ep.act_out;
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
          c_old[e] = (ok && g.beta != 0.0f) ? (ep.c_src ? ep.c_src[(long)row * ep.ld_c_src + col] : g.C[(long)row * g.ldc + col]) : 0.0f;
          w_old[e] = (ok && ep.W) ? ep.W[(long)row * ep.ldw + col] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (row >= g.M) continue;
          float *cp = g.C + (long)row * g.ldc + col;
          float v = fmaf(g.alpha, acc[i][n][e], fmaf(g.beta, c_old[e], bias));
          if (ep.clip > 0.0f) v = fminf(fmaxf(v, -ep.clip), ep.clip);
          if (ep.W && g.beta == 0.0f) __builtin_nontemporal_store(v, cp);  // gradient written once, not read again this step
          else *cp = v;
          if (ep.W) {
            const float wn = fmaf(ep.w_alpha, v, w_old[e]);
            ep.W[(long)row * ep.ldw + col] = wn;
            if (w_planes) epi_plane_store(ep, x.pscale, wn, row, col);
            if (track) x.wmax = fmaxf(x.wmax, epi_finite_abs(wn));
          }
          if (track) x.cmax = fmaxf(x.cmax, epi_finite_abs(v));
          if (ep.act_out) {
            float a = ep.act == 1 ? sigmoid_ref(v) : ep.act == 2 ? tanh_ref(v) : ep.act == 3 ? fmaxf(v, 0.0f) : v;
            ep.act_out[(long)row * ep.ld_act + col] = a;
            if (a_planes) epi_plane_store(ep, x.pscale, a, row, col);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (row >= g.M) continue;
          float v = fmaf(g.alpha, acc[i][n][e], bias);
          if (ep.clip > 0.0f) v = fminf(fmaxf(v, -ep.clip), ep.clip);
          g.C[(long)row * g.ldc + col] = v;
          if (track) x.cmax = fmaxf(x.cmax, epi_finite_abs(v));
          if (ep.act_out) {
            float a = ep.act == 1 ? sigmoid_ref(v) : ep.act == 2 ? tanh_ref(v) : ep.act == 3 ? fmaxf(v, 0.0f) : v;
            ep.act_out[(long)row * ep.ld_act + col] = a;
            if (a_planes) epi_plane_store(ep, x.pscale, a, row, col);
          }
        }
      }
    }
}

// Column statistics of a wave's output patch for aslp_gemm_epilogue.colstats (forward products only: beta == 0, no W -- the host
// guarantees it): the values are the ones the epilogue stores, v = fma(alpha, acc, bias) clipped.  A lane holds 16 rows of one
// column; the two lane halves hold the other 16 rows of the same 32-row group.
template <int TM, int TN>
__device__ __forceinline__ void gemm_colstats(const GemmArgs &g, const f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int lh) {
  const aslp_gemm_epilogue &ep = g.ep;
  const int groups = (g.M + 31) >> 5;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++) {
      const int col = col0 + n * 32 + l31;
      const float bias = (ep.bias && col < g.N) ? ep.bias[col] : 0.0f;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int row = row0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float v = fmaf(g.alpha, acc[i][n][e], bias);
        if (ep.clip > 0.0f) v = fminf(fmaxf(v, -ep.clip), ep.clip);
        if (row < g.M) {
          s0 += (double)v;
          s1 += (double)(v * v);
          s2 += (double)v * (double)v;
        }
      }
      s0 += __shfl_xor(s0, 32, 64);
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      const int grp = (row0 + i * 32) >> 5;
      if (lh == 0 && col < g.N && grp < groups) {
        double *p = ep.colstats + (long)grp * ep.colstats_ld + col;
        const long plane = (long)groups * ep.colstats_ld;
        p[0] = s0; p[plane] = s1; p[2 * plane] = s2;
      }
    }
}

// The same epilogue with 16-byte global accesses.  A lane of the 32x32 MFMA holds 16 ROWS of one column, so the epilogue above
// issues one 4-byte access per element: 16 stores (48 memory instructions with the fused SGD step) per wave and patch, and the
// tail of the kernel is store-ISSUE bound (one workgroup per CU: nothing overlaps it).  Here the wave first passes its patch
// through its own 32 x 36-float slice of the (now idle) operand LDS: written in the accumulator layout, read back as rows, so
// that every lane holds 4 x 4 consecutive columns -- 4 instead of 16 accesses per array, each wave instruction 8 full 128-byte lines.
// The arithmetic per element is the scalar epilogue's.  Eligibility (uniform, checked by the caller): N, ldc (ldw, ld_act) multiples
// of 4, all pointers 16-byte aligned.  `tile`: this wave's LDS slice; the caller has passed a workgroup barrier since the last operand read.
constexpr int kEpiPitch = 36;
__host__ __device__ __forceinline__ bool gemm_epilogue_wide_ok(const GemmArgs &g) {
  const aslp_gemm_epilogue &ep = g.ep;
  auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  return (g.N & 3) == 0 && (g.ldc & 3) == 0 && al(g.C) && (!ep.bias || al(ep.bias)) && (!ep.W || ((ep.ldw & 3) == 0 && al(ep.W))) &&
         (!ep.c_src || ((ep.ld_c_src & 3) == 0 && al(ep.c_src))) &&
         (!ep.act_out || ((ep.ld_act & 3) == 0 && al(ep.act_out)));
}
// one lane's 4 consecutive outputs of a row (columns col .. col+3): the arithmetic of the epilogue ...
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gemm_epilogue_value4(const GemmArgs &g, float4 acc4, float4 c_old, float4 w_old, float4 bias, f32x4v *out, float4 *wn) {
  const aslp_gemm_epilogue &ep = g.ep;
  float o[4] = {acc4.x, acc4.y, acc4.z, acc4.w};
  const float co[4] = {c_old.x, c_old.y, c_old.z, c_old.w}, bs[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    // explicit fused multiply-adds: the same rounding wherever this arithmetic is compiled (the narrow epilogue, the split-K
    // reduction and the register-staged kernel spell it the same way), whatever contraction the optimiser would pick
    float t = (ep.W != nullptr || g.beta != 0.0f) ? fmaf(g.alpha, o[q], fmaf(g.beta, co[q], bs[q])) : fmaf(g.alpha, o[q], bs[q]);
    if (ep.clip > 0.0f) t = fminf(fmaxf(t, -ep.clip), ep.clip);
    o[q] = t;
  }
  *out = f32x4v{o[0], o[1], o[2], o[3]};
  wn->x = fmaf(ep.w_alpha, o[0], w_old.x); wn->y = fmaf(ep.w_alpha, o[1], w_old.y);   // (only stored when ep.W is set)
  wn->z = fmaf(ep.w_alpha, o[2], w_old.z); wn->w = fmaf(ep.w_alpha, o[3], w_old.w);
}
// ... and its stores
__device__ __forceinline__ void gemm_epilogue_store4(const GemmArgs &g, const f32x4v &out, const float4 &wn, int row, int col, float pscale = 0.f) {
  const aslp_gemm_epilogue &ep = g.ep;
  if (pscale != 0.f && ep.planes_of == 1 && ep.W) epi_plane_store4(ep, pscale, wn, row, col);
  f32x4v *cp = reinterpret_cast<f32x4v *>(g.C + (long)row * g.ldc + col);
  if (ep.W && g.beta == 0.0f) __builtin_nontemporal_store(out, cp);  // gradient written once, not read again this step
  else *cp = out;
  if (ep.W) *reinterpret_cast<float4 *>(ep.W + (long)row * ep.ldw + col) = wn;
  if (ep.act_out) {
    float a[4];
#pragma unroll
    for (int q = 0; q < 4; q++) a[q] = ep.act == 1 ? sigmoid_ref(out[q]) : ep.act == 2 ? tanh_ref(out[q]) : ep.act == 3 ? fmaxf(out[q], 0.0f) : out[q];
    *reinterpret_cast<float4 *>(ep.act_out + (long)row * ep.ld_act + col) = make_float4(a[0], a[1], a[2], a[3]);
    if (pscale != 0.f && ep.planes_of == 2) epi_plane_store4(ep, pscale, make_float4(a[0], a[1], a[2], a[3]), row, col);
  }
}

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_wide(const GemmArgs &g, const f32x16 (&acc)[TM][TN], int row0, int col0, int lane, float *tile,
                                                   EpiExtra &x, bool track) {
  const aslp_gemm_epilogue &ep = g.ep;
  const int l31 = lane & 31, lh = lane >> 5, c4 = lane & 7, rr = lane >> 3;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++) {
#pragma unroll
      for (int e = 0; e < 16; e++) tile[((e & 3) + 8 * (e >> 2) + 4 * lh) * kEpiPitch + l31] = acc[i][n][e];
      const int col = col0 + n * 32 + 4 * c4;
      const bool colok = col < g.N;
      float4 v[4], c_old[4], w_old[4];
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      // Every read below is unconditional per lane: lanes past the matrix edge read a clamped, valid position and never store.  Only the
      // wave-uniform "is this piece asked for at all" conditions branch.  (A per-lane `ok ? *p : zero` became a pointer select between
      // the operand and a zero in scratch memory: flat loads split into three, behind exec-mask branches -- in the one place where the
      // weight-gradient kernel is bound by how fast it gets its reads out.)
      const int colc = colok ? col : 0;
      float4 bias = zero;
      if (ep.bias) bias = *reinterpret_cast<const float4 *>(ep.bias + colc);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = rr + 8 * j, row = row0 + i * 32 + r, rowc = row < g.M ? row : g.M - 1;
        v[j] = *reinterpret_cast<const float4 *>(tile + r * kEpiPitch + 4 * c4);
        c_old[j] = zero;
        if (g.beta != 0.0f) {
          if (ep.c_src) c_old[j] = *reinterpret_cast<const float4 *>(ep.c_src + (long)rowc * ep.ld_c_src + colc);
          else c_old[j] = *reinterpret_cast<const float4 *>(g.C + (long)rowc * g.ldc + colc);
        }
        w_old[j] = zero;
        if (ep.W) w_old[j] = *reinterpret_cast<const float4 *>(ep.W + (long)rowc * ep.ldw + colc);
      }
      f32x4v out[4];
      float4 wn[4];
#pragma unroll
      for (int j = 0; j < 4; j++) gemm_epilogue_value4(g, v[j], c_old[j], w_old[j], bias, &out[j], &wn[j]);
      // A patch that lies inside the matrix (wave-uniform; every patch of the layer products) stores in a straight line: with the
      // per-lane edge test around each row's stores the compiler put an s_waitcnt vmcnt(0) behind every row group -- the wave sat
      // out the acknowledgement of its stores four times per patch, at the tail of the kernel where nothing else is left to run.
      const bool interior = row0 + i * 32 + 32 <= g.M && col0 + n * 32 + 32 <= g.N;
      if (interior && !ep.act_out) {
        const long r0 = row0 + i * 32 + rr;
        if (ep.W && g.beta == 0.0f) {
#pragma unroll
          for (int j = 0; j < 4; j++) __builtin_nontemporal_store(out[j], reinterpret_cast<f32x4v *>(g.C + (r0 + 8 * j) * g.ldc + col));
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) *reinterpret_cast<f32x4v *>(g.C + (r0 + 8 * j) * g.ldc + col) = out[j];
        }
        if (ep.W) {
#pragma unroll
          for (int j = 0; j < 4; j++) *reinterpret_cast<float4 *>(ep.W + (r0 + 8 * j) * ep.ldw + col) = wn[j];
          if (track && x.pscale != 0.f && ep.planes_of == 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) epi_plane_store4(ep, x.pscale, wn[j], r0 + 8 * j, col);
          }
        }
        if (track) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            x.cmax = fmaxf(x.cmax, fmaxf(fmaxf(epi_finite_abs(out[j][0]), epi_finite_abs(out[j][1])), fmaxf(epi_finite_abs(out[j][2]), epi_finite_abs(out[j][3]))));
            if (ep.W) x.wmax = s16_absmax4(x.wmax, wn[j]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int row = row0 + i * 32 + rr + 8 * j;
          if (!colok || row >= g.M) continue;
          gemm_epilogue_store4(g, out[j], wn[j], row, col, track ? x.pscale : 0.f);
          if (track) {
            x.cmax = fmaxf(x.cmax, fmaxf(fmaxf(epi_finite_abs(out[j][0]), epi_finite_abs(out[j][1])), fmaxf(epi_finite_abs(out[j][2]), epi_finite_abs(out[j][3]))));
            if (ep.W) x.wmax = s16_absmax4(x.wmax, wn[j]);
          }
        }
      }
    }
}

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, const f32x16 (&acc)[TM][TN], int row0, int col0, int l31, int lh) {
  EpiExtra none;
  gemm_epilogue<TM, TN>(g, acc, row0, col0, l31, lh, none, false);
}
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_wide(const GemmArgs &g, const f32x16 (&acc)[TM][TN], int row0, int col0, int lane, float *tile) {
  EpiExtra none;
  gemm_epilogue_wide<TM, TN>(g, acc, row0, col0, lane, tile, none, false);
}

// gemm_glds.hip: direct-to-LDS kernels.  Returns false if the problem is not eligible (caller falls back).
bool gemm_glds_launch(GemmArgs &g, bool a_kc, bool b_kc, int cfg, int *cfg_used);  // *cfg_used: the tile configuration that ran
// second half of a split-K product (gemm_glds.hip): C = epilogue(alpha * sum_s part[s] + beta * C), chunks added in order; r: the original
// product's arguments (pair: the second product's partials follow the first's)
// returns the number of workgroups (= per-workgroup maxima left in r.ep.cmax_parts when that is set)
int gemm_splitk_reduce(const float *part, int split, long stride, const GemmArgs &r);
// gemm_split16.hip: the same product on the fp16 matrix instruction with two-piece fp32-equivalent operands (A/B: ASLP_GEMM_SPLIT_F16=1).
// cfg 0 = default tile.  false: not eligible.  Forms ep.colstats itself; ep.colsum stays the caller's.
bool gemm_split16_enabled();
struct S16View;
// pa / pb: prepared planes of the operands (split16.h) or NULL = convert g.A / g.B in scratch
bool gemm_split16_launch(GemmArgs &g, bool a_kc, bool b_kc, int cfg, const S16View *pa = nullptr, const S16View *pb = nullptr);
// ... both from prepared planes; a1 / b1: the second product's operands when g.pair
bool gemm_split16_planes_launch(GemmArgs &g, bool a_kc, bool b_kc, const S16View &a, const S16View &b, const S16View *a1, const S16View *b1, int cfg);

}  // namespace aslp
