// gemm.hip -- fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces the cuBLAS sgemm seam (src/aslp-cudamatrix/cublas-wrappers.h:28-47, call site
// CuMatrixBase::AddMatMat cu-matrix.cc:1027-1061): row-major
//     C[M x N] = alpha * op(A) * op(B) + beta * C   (+ fused epilogue, see aslp_kernels.h)
// gfx950 has no xf32/TF32 path: the f32-input MFMA is exact fp32 (bitwise an fmaf chain) at
// 157.3 TFLOP/s peak, which is what keeps Propagate/Backpropagate inside the 1e-4 parity bar.
//
// Tiling (256 threads = 4 waves, one per SIMD; each wave owns a (BM/WGM) x (BN/WGN) patch
// made of 32x32 MFMA tiles; accumulators stay in registers for the whole K loop):
//   * operands whose K index is contiguous in memory ("KC": A of NT/NN, B of NT) are staged
//     as [rows][BK+4] and read back with ONE ds_read_b128 per 4 MFMA k-steps: lane l takes
//     k = 8*h + 4*(l>>5) + {0..3} of row (l&31).  The +4 pad makes the 16-lane b128 groups
//     conflict-free (stride 20 dwords = 5 slots, odd).
//   * operands whose row index is contiguous ("RC": A of TN, B of NN/TN) are staged as
//     [BK][rows+4] and read with ds_read_b32 using the SAME k permutation, so both operands
//     of one MFMA agree on which two k's a step consumes.
//   * global->LDS goes through registers (float4 per lane, coalesced along the contiguous
//     dimension), double-buffered in LDS: the loads of tile t+1 are issued before the MFMAs
//     of tile t and written to the other buffer after them -- one barrier per K tile.
// The K order inside a tile is a permutation of 0..BK-1; fp32 sums are therefore not
// bitwise those of a k-ascending loop (parity is tolerance-based for GEMM rows).
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "gemm_common.h"
#include "split16.h"

namespace aslp {
namespace {

template <int BK>
struct TileK {
  static constexpr int KC_LD = BK + 4;  // row stride of a KC tile: odd number of 16-byte slots
};

template <int BR, int BK, int NT>
struct Stage {
  static constexpr int TOTAL = BR * BK / 4;          // float4 per tile
  static constexpr int NV = (TOTAL + NT - 1) / NT;   // float4 per thread per tile
  static constexpr bool FULL = TOTAL % NT == 0;      // otherwise the tail threads idle
};

// ---- global -> registers ------------------------------------------------------------------
// VEC variants are branch-free: addresses are clamped into the matrix and out-of-range k is
// zeroed with a select, so the loads of one tile issue back to back with no s_waitcnt between
// them (hipcc serialises "branch around each load" forms, cdna guide §5 trap (c)).  They need
// 16-byte aligned pointers/strides and the contiguous extent to be a multiple of 4.
// Rows beyond the matrix are clamped (they only feed accumulator rows that are never stored).
//
// KC operand: src is [R x K] row-major (ld), tile rows r0.., k0..
template <int BR, int BK, int NT, bool VEC>
__device__ __forceinline__ void gload_kc(const float *__restrict__ src, int ld, int R, int K, int r0, int k0,
                                         float4 (&v)[Stage<BR, BK, NT>::NV]) {
  constexpr int C4 = BK / 4;
  typedef Stage<BR, BK, NT> S;
#pragma unroll
  for (int i = 0; i < S::NV; i++) {
    int idx = threadIdx.x + i * NT;
    int row = idx / C4, c4 = idx % C4;
    int gr = r0 + row, gk = k0 + c4 * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
      if (S::FULL || idx < S::TOTAL) {
        const bool kv = gk < K;  // K % 4 == 0: the whole float4 is in or out
        gr = gr < R ? gr : R - 1;
        x = *reinterpret_cast<const float4 *>(src + (long)gr * ld + (kv ? gk : 0));  // masked at store time
      }
    } else {
      if ((S::FULL || idx < S::TOTAL) && gr < R) {
        const float *p = src + (long)gr * ld + gk;
        if (gk < K) x.x = p[0];
        if (gk + 1 < K) x.y = p[1];
        if (gk + 2 < K) x.z = p[2];
        if (gk + 3 < K) x.w = p[3];
      }
    }
    v[i] = x;
  }
}
// The k-tail zeroing of the VEC loads happens here, at LDS-store time: touching the loaded
// registers right after the load would make the compiler wait for them (vmcnt) immediately and
// kill the prefetch distance.
template <int BR, int BK, int NT, bool VEC>
__device__ __forceinline__ void sstore_kc(float *lds, const float4 (&v)[Stage<BR, BK, NT>::NV], int k0, int K) {
  constexpr int C4 = BK / 4;
  typedef Stage<BR, BK, NT> S;
#pragma unroll
  for (int i = 0; i < S::NV; i++) {
    int idx = threadIdx.x + i * NT;
    int row = idx / C4, c4 = idx % C4;
    float4 x = v[i];
    if (VEC) {
      const bool kv = k0 + c4 * 4 < K;
      x.x = kv ? x.x : 0.f; x.y = kv ? x.y : 0.f; x.z = kv ? x.z : 0.f; x.w = kv ? x.w : 0.f;
    }
    if (S::FULL || idx < S::TOTAL) *reinterpret_cast<float4 *>(lds + row * TileK<BK>::KC_LD + c4 * 4) = x;
  }
}
// RC operand: src is [K x R] row-major (ld), tile k0.., rows r0..
template <int BR, int BK, int NT, bool VEC>
__device__ __forceinline__ void gload_rc(const float *__restrict__ src, int ld, int R, int K, int r0, int k0,
                                         float4 (&v)[Stage<BR, BK, NT>::NV]) {
  constexpr int C4 = BR / 4;
  typedef Stage<BR, BK, NT> S;
#pragma unroll
  for (int i = 0; i < S::NV; i++) {
    int idx = threadIdx.x + i * NT;
    int k = idx / C4, c4 = idx % C4;
    int gk = k0 + k, gr = r0 + c4 * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
      if (S::FULL || idx < S::TOTAL) {
        const bool kv = gk < K;
        gr = gr + 3 < R ? gr : R - 4;  // R % 4 == 0
        x = *reinterpret_cast<const float4 *>(src + (long)(kv ? gk : 0) * ld + gr);  // masked at store time
      }
    } else {
      if ((S::FULL || idx < S::TOTAL) && gk < K) {
        const float *p = src + (long)gk * ld + gr;
        if (gr < R) x.x = p[0];
        if (gr + 1 < R) x.y = p[1];
        if (gr + 2 < R) x.z = p[2];
        if (gr + 3 < R) x.w = p[3];
      }
    }
    v[i] = x;
  }
}
template <int BR, int BK, int NT, bool VEC>
__device__ __forceinline__ void sstore_rc(float *lds, const float4 (&v)[Stage<BR, BK, NT>::NV], int k0, int K) {
  constexpr int C4 = BR / 4;
  typedef Stage<BR, BK, NT> S;
#pragma unroll
  for (int i = 0; i < S::NV; i++) {
    int idx = threadIdx.x + i * NT;
    int k = idx / C4, c4 = idx % C4;
    float4 x = v[i];
    if (VEC) {
      const bool kv = k0 + k < K;
      x.x = kv ? x.x : 0.f; x.y = kv ? x.y : 0.f; x.z = kv ? x.z : 0.f; x.w = kv ? x.w : 0.f;
    }
    if (S::FULL || idx < S::TOTAL) *reinterpret_cast<float4 *>(lds + k * (BR + 4) + c4 * 4) = x;
  }
}

// ---- per-element forms (vector path only) used by the interleaved main loop ----------------------------
template <int BR, int BK, int NT, bool KC>
__device__ __forceinline__ float4 gload_one(const float *__restrict__ src, int ld, int R, int K, int r0, int k0, int i) {
  const int idx = threadIdx.x + i * NT;
  if (KC) {
    constexpr int C4 = BK / 4;
    const int row = idx / C4, c4 = idx % C4;
    int gr = r0 + row;
    const int gk = k0 + c4 * 4;
    gr = gr < R ? gr : R - 1;
    return *reinterpret_cast<const float4 *>(src + (long)gr * ld + (gk < K ? gk : 0));
  } else {
    constexpr int C4 = BR / 4;
    const int k = idx / C4, c4 = idx % C4;
    const int gk = k0 + k;
    int gr = r0 + c4 * 4;
    gr = gr + 3 < R ? gr : R - 4;
    return *reinterpret_cast<const float4 *>(src + (long)(gk < K ? gk : 0) * ld + gr);
  }
}
template <int BR, int BK, int NT, bool KC>
__device__ __forceinline__ void sstore_one(float *lds, float4 x, int k0, int K, int i) {
  const int idx = threadIdx.x + i * NT;
  if (KC) {
    constexpr int C4 = BK / 4;
    const int row = idx / C4, c4 = idx % C4;
    const bool kv = k0 + c4 * 4 < K;
    x.x = kv ? x.x : 0.f; x.y = kv ? x.y : 0.f; x.z = kv ? x.z : 0.f; x.w = kv ? x.w : 0.f;
    *reinterpret_cast<float4 *>(lds + row * TileK<BK>::KC_LD + c4 * 4) = x;
  } else {
    constexpr int C4 = BR / 4;
    const int k = idx / C4, c4 = idx % C4;
    const bool kv = k0 + k < K;
    x.x = kv ? x.x : 0.f; x.y = kv ? x.y : 0.f; x.z = kv ? x.z : 0.f; x.w = kv ? x.w : 0.f;
    *reinterpret_cast<float4 *>(lds + k * (BR + 4) + c4 * 4) = x;
  }
}

template <int BR, int BK, bool KC>
struct OperandTile {
  static constexpr int kFloats = KC ? BR * TileK<BK>::KC_LD : BK * (BR + 4);
};

template <int BM, int BN, int BK, bool A_KC, bool B_KC>
constexpr int gemm_lds_bytes() {
  return 2 * (OperandTile<BM, BK, A_KC>::kFloats + OperandTile<BN, BK, B_KC>::kFloats) * (int)sizeof(float);
}

// A_KC: op(A) rows are contiguous in k (transA == 0).  B_KC: op(B) columns are contiguous in k (transB == 1).
template <int BM, int BN, int BK, int WGM, int WGN, bool A_KC, bool B_KC, bool VEC, int LOOP>
__global__ void __launch_bounds__(64 * WGM * WGN) gemm_f32_mfma(GemmArgs g) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;  // wave patch
  constexpr int TM = WM / 32, TN = WN / 32;    // 32x32 MFMA tiles per wave
  static_assert((WGM * WGN == 4 || WGM * WGN == 8) && TM >= 1 && TN >= 1, "bad wave grid");
  constexpr int KC_LD = TileK<BK>::KC_LD;
  constexpr int A_FLOATS = OperandTile<BM, BK, A_KC>::kFloats, B_FLOATS = OperandTile<BN, BK, B_KC>::kFloats;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // layout [A0 | B0 | A1 | B1]
  auto a_buf = [&](int b) { return lds + b * (A_FLOATS + B_FLOATS); };
  auto b_buf = [&](int b) { return lds + b * (A_FLOATS + B_FLOATS) + A_FLOATS; };

  // XCD-aware tile order.  Consecutive workgroup ids land on different XCDs (id % 8), each with a private
  // 4 MB L2; every operand panel an XCD touches is pulled through the fabric once per XCD, so give each
  // XCD a compact 2-D sub-grid of tiles (px x py = 8) instead of a strip: for 16 x 16 tiles a 2 x 4 split
  // needs 8 + 4 panels per XCD where a strip of 32 consecutive tiles needs 2 + 16.
  const int nt = g.tiles_m * g.tiles_n;
  int tm, tn;
  {
    const int bid0 = blockIdx.x, xcd = bid0 % 8, j = bid0 / 8;
    int px = 0;
    // prefer the split whose sub-grid is closest to square in elements (BM*sm vs BN*sn)
    long best = -1;
    for (int cand = 1; cand <= 8; cand *= 2) {
      const int py = 8 / cand;
      if (g.tiles_m % cand || g.tiles_n % py) continue;
      const long cost = (long)(g.tiles_m / cand) * BM + (long)(g.tiles_n / py) * BN;
      if (best < 0 || cost < best) { best = cost; px = cand; }
    }
    if (px > 0) {
      const int py = 8 / px, sn = g.tiles_n / py, sm = g.tiles_m / px;
      (void)sm;
      tm = (xcd / py) * (g.tiles_m / px) + j / sn;
      tn = (xcd % py) * sn + j % sn;
    } else {  // ragged grid: contiguous run of tiles per XCD
      const int q = nt / 8, r = nt % 8;
      const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
      tm = bid / g.tiles_n;
      tn = bid % g.tiles_n;
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

  // two register stages: while tile t is multiplied out of LDS, tile t+1 sits in one stage
  // (already landed) and tile t+2 is in flight into the other -- >= one full MFMA phase of
  // latency slack for every global load.
  float4 ra0[Stage<BM, BK, NT>::NV], rb0[Stage<BN, BK, NT>::NV], ra1[Stage<BM, BK, NT>::NV], rb1[Stage<BN, BK, NT>::NV];
  const int ktiles = (g.K + BK - 1) / BK;

#define ASLP_GLOAD(kt, RA, RB)                                                                   \
  do {                                                                                           \
    if (A_KC) gload_kc<BM, BK, NT, VEC>(g.A, g.lda, g.M, g.K, m0, (kt)*BK, RA);                      \
    else gload_rc<BM, BK, NT, VEC>(g.A, g.lda, g.M, g.K, m0, (kt)*BK, RA);                           \
    if (B_KC) gload_kc<BN, BK, NT, VEC>(g.B, g.ldb, g.N, g.K, n0, (kt)*BK, RB);                      \
    else gload_rc<BN, BK, NT, VEC>(g.B, g.ldb, g.N, g.K, n0, (kt)*BK, RB);                           \
  } while (0)
#define ASLP_SSTORE(buf, kt, RA, RB)                                                             \
  do {                                                                                           \
    if (A_KC) sstore_kc<BM, BK, NT, VEC>(a_buf(buf), RA, (kt)*BK, g.K);                              \
    else sstore_rc<BM, BK, NT, VEC>(a_buf(buf), RA, (kt)*BK, g.K);                                   \
    if (B_KC) sstore_kc<BN, BK, NT, VEC>(b_buf(buf), RB, (kt)*BK, g.K);                              \
    else sstore_rc<BN, BK, NT, VEC>(b_buf(buf), RB, (kt)*BK, g.K);                                   \
  } while (0)

  if constexpr (LOOP == 0) {
  // ---- loop style 0: LDS reads interleaved with the MFMAs of the same K tile ----
  auto compute = [&](int cur) {
    const float *a_s = a_buf(cur), *b_s = b_buf(cur);
#pragma unroll
    for (int h = 0; h < BK / 8; h++) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int row = wm * WM + i * 32 + l31;
        if (A_KC) {
          float4 t = *reinterpret_cast<const float4 *>(a_s + row * KC_LD + h * 8 + lh * 4);
          af[i][0] = t.x; af[i][1] = t.y; af[i][2] = t.z; af[i][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) af[i][j] = a_s[(h * 8 + lh * 4 + j) * (BM + 4) + row];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; i++) {
        const int col = wn * WN + i * 32 + l31;
        if (B_KC) {
          float4 t = *reinterpret_cast<const float4 *>(b_s + col * KC_LD + h * 8 + lh * 4);
          bf[i][0] = t.x; bf[i][1] = t.y; bf[i][2] = t.z; bf[i][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) bf[i][j] = b_s[(h * 8 + lh * 4 + j) * (BN + 4) + col];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int n = 0; n < TN; n++)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j], bf[n][j], acc[i][n], 0, 0, 0);
    }
  };

  // Branch-free steady state: the k-tile count is rounded up to even (a tile beyond K loads from
  // clamped addresses and is zeroed at store time, so it adds 0), and loads/stores for tiles past
  // the end are issued unconditionally into buffers nobody reads.  With no conditional code
  // between the loads and their use the compiler keeps exact vmcnt counts: a store of stage X
  // waits only for stage X, the other stage's loads stay in flight across the barrier.
  const int kt_end = (ktiles + 1) & ~1;
  if (ktiles > 0) {
    ASLP_GLOAD(0, ra0, rb0);
    ASLP_SSTORE(0, 0, ra0, rb0);
    ASLP_GLOAD(1, ra1, rb1);
  }
  __syncthreads();

  for (int kt = 0; kt < kt_end; kt += 2) {
    // even step: tile kt in LDS[0], tile kt+1 in stage 1 (landed), load kt+2 into stage 0
    ASLP_GLOAD(kt + 2, ra0, rb0);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch issue ABOVE the MFMA phase
    compute(0);
    __builtin_amdgcn_sched_barrier(0);
    ASLP_SSTORE(1, kt + 1, ra1, rb1);
    __syncthreads();
    // odd step: tile kt+1 in LDS[1], tile kt+2 in stage 0, load kt+3 into stage 1
    ASLP_GLOAD(kt + 3, ra1, rb1);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    ASLP_SSTORE(0, kt + 2, ra0, rb0);
    __syncthreads();
  }
  } else if constexpr (LOOP == 5) {
  // ---- loop style 5: fragment double buffering with everything else issued in the shadow of the MFMAs ----
  // A wave issues in program order, so whatever precedes the first MFMA of a K tile (address VALU, the LDS
  // stores of the next tile, the global loads of the one after, the barrier, the fragment reads) leaves the
  // matrix pipe idle -- for BOTH waves of a SIMD, which the barrier keeps in lockstep.  Here each of those
  // units is slotted between two MFMAs of the current tile (sched_barrier pins the order): stores and
  // global loads in the first half, the barrier in the middle, the next tile's fragment reads in the
  // second half.  Only the barrier skew itself remains exposed.
  static_assert(VEC, "interleaved loop needs the vector load path");
  constexpr int KH = BK / 8;
  constexpr int NVA = Stage<BM, BK, NT>::NV, NVB = Stage<BN, BK, NT>::NV;
  static_assert(Stage<BM, BK, NT>::FULL && Stage<BN, BK, NT>::FULL, "tile loads must divide evenly over the threads");
  constexpr int NM = KH * 4 * TM * TN;    // MFMAs per K tile per wave
  constexpr int NPRE = 2 * (NVA + NVB);   // store units then load units
  constexpr int NRD = KH * (TM + TN);     // fragment read units
  constexpr int SB = NM / 2 - 1;          // the barrier follows this MFMA slot
  static_assert(NM >= 4, "tile too small to interleave");
  struct Frag {
    float a[KH][TM][4], b[KH][TN][4];
  };
  auto read_unit = [&](int cur, Frag &f, auto R_) {
    constexpr int r = decltype(R_)::value;
    const float *a_s = a_buf(cur), *b_s = b_buf(cur);
    constexpr int h = r / (TM + TN), t = r % (TM + TN);
    if constexpr (t < TM) {
      const int row = wm * WM + t * 32 + l31;
      if (A_KC) {
        float4 v = *reinterpret_cast<const float4 *>(a_s + row * KC_LD + h * 8 + lh * 4);
        f.a[h][t][0] = v.x; f.a[h][t][1] = v.y; f.a[h][t][2] = v.z; f.a[h][t][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) f.a[h][t][j] = a_s[(h * 8 + lh * 4 + j) * (BM + 4) + row];
      }
    } else {
      constexpr int tb = t - TM;
      const int col = wn * WN + tb * 32 + l31;
      if (B_KC) {
        float4 v = *reinterpret_cast<const float4 *>(b_s + col * KC_LD + h * 8 + lh * 4);
        f.b[h][tb][0] = v.x; f.b[h][tb][1] = v.y; f.b[h][tb][2] = v.z; f.b[h][tb][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) f.b[h][tb][j] = b_s[(h * 8 + lh * 4 + j) * (BN + 4) + col];
      }
    }
  };
  auto mma_unit = [&](const Frag &f, auto M_) {
    constexpr int m = decltype(M_)::value;
    constexpr int n = m % TN, i = (m / TN) % TM, j = (m / (TN * TM)) % 4, h = m / (TN * TM * 4);
    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[h][i][j], f.b[h][n][j], acc[i][n], 0, 0, 0);
  };
  // unit u of the pre-barrier work: [store A | store B | load A | load B]
  auto pre_unit = [&](auto U_, int buf, int kt_store, int kt_load, float4 (&RA)[NVA], float4 (&RB)[NVB]) {
    constexpr int u = decltype(U_)::value;
    if constexpr (u < NVA) sstore_one<BM, BK, NT, A_KC>(a_buf(buf), RA[u], kt_store * BK, g.K, u);
    else if constexpr (u < NVA + NVB) sstore_one<BN, BK, NT, B_KC>(b_buf(buf), RB[u - NVA], kt_store * BK, g.K, u - NVA);
    else if constexpr (u < 2 * NVA + NVB) RA[u - NVA - NVB] = gload_one<BM, BK, NT, A_KC>(g.A, g.lda, g.M, g.K, m0, kt_load * BK, u - NVA - NVB);
    else RB[u - 2 * NVA - NVB] = gload_one<BN, BK, NT, B_KC>(g.B, g.ldb, g.N, g.K, n0, kt_load * BK, u - 2 * NVA - NVB);
  };
  // one K tile: MFMAs from fcur; tile kt_store goes registers -> LDS[buf]; tile kt_load global -> the same registers;
  // after the barrier the fragments of LDS[buf] are read into fnxt
  auto half_step = [&](const Frag &fcur, Frag &fnxt, int buf, int kt_store, int kt_load, float4 (&RA)[NVA], float4 (&RB)[NVB]) {
    static_for<0, NM>([&](auto S_) {
      constexpr int sidx = decltype(S_)::value;
      mma_unit(fcur, S_);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (sidx <= SB) {
        static_for<sidx * NPRE / (SB + 1), (sidx + 1) * NPRE / (SB + 1)>([&](auto U_) { pre_unit(U_, buf, kt_store, kt_load, RA, RB); });
        if constexpr (sidx == SB) __syncthreads();
      } else {
        constexpr int NS = NM - SB - 1;
        static_for<(sidx - SB - 1) * NRD / NS, (sidx - SB) * NRD / NS>([&](auto R_) { read_unit(buf, fnxt, R_); });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  const int kt_end = (ktiles + 1) & ~1;
  Frag f0, f1;
  if (ktiles > 0) {
    ASLP_GLOAD(0, ra0, rb0);
    ASLP_SSTORE(0, 0, ra0, rb0);
    ASLP_GLOAD(1, ra1, rb1);
    ASLP_GLOAD(2, ra0, rb0);
  }
  __syncthreads();
  if (ktiles > 0) static_for<0, NRD>([&](auto R_) { read_unit(0, f0, R_); });
  for (int kt = 0; kt < kt_end; kt += 2) {
    half_step(f0, f1, 1, kt + 1, kt + 3, ra1, rb1);
    half_step(f1, f0, 0, kt + 2, kt + 4, ra0, rb0);
  }
  } else {
  // ---- loop style 1: fragment double buffering ----
  // Operand fragments of one whole K tile live in registers (BK/8 b128 reads per 32-row sub-tile), double
  // buffered: the ds_reads of tile t+1 are issued right after the barrier that publishes it and complete
  // under the MFMAs of tile t, so LDS latency is off the critical path and each K tile costs ONE barrier.
  constexpr int KH = BK / 8;
  struct Frag {
    float a[KH][TM][4], b[KH][TN][4];
  };
  auto read_frags = [&](int cur, Frag &f) {
    const float *a_s = a_buf(cur), *b_s = b_buf(cur);
#pragma unroll
    for (int h = 0; h < KH; h++) {
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int row = wm * WM + i * 32 + l31;
        if (A_KC) {
          float4 t = *reinterpret_cast<const float4 *>(a_s + row * KC_LD + h * 8 + lh * 4);
          f.a[h][i][0] = t.x; f.a[h][i][1] = t.y; f.a[h][i][2] = t.z; f.a[h][i][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) f.a[h][i][j] = a_s[(h * 8 + lh * 4 + j) * (BM + 4) + row];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; i++) {
        const int col = wn * WN + i * 32 + l31;
        if (B_KC) {
          float4 t = *reinterpret_cast<const float4 *>(b_s + col * KC_LD + h * 8 + lh * 4);
          f.b[h][i][0] = t.x; f.b[h][i][1] = t.y; f.b[h][i][2] = t.z; f.b[h][i][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) f.b[h][i][j] = b_s[(h * 8 + lh * 4 + j) * (BN + 4) + col];
        }
      }
    }
  };
  auto mma = [&](const Frag &f) {
#pragma unroll
    for (int h = 0; h < KH; h++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int n = 0; n < TN; n++)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[h][i][j], f.b[h][n][j], acc[i][n], 0, 0, 0);
  };

  // Branch-free steady state: the k-tile count is rounded up to even (a tile beyond K loads from
  // clamped addresses and is zeroed at store time, so it adds 0), and loads/stores for tiles past
  // the end are issued unconditionally into buffers nobody reads.  With no conditional code
  // between the loads and their use the compiler keeps exact vmcnt counts.
  // Invariant at the top of an even step kt: LDS[0] holds tile kt and its fragments are already in f0;
  // register stage 1 holds tile kt+1 (landed), stage 0 holds tile kt+2 (in flight since the previous step):
  // every global load has two MFMA phases to land before its registers are stored to LDS.
  const int kt_end = (ktiles + 1) & ~1;
  Frag f0, f1;
  if (ktiles > 0) {
    ASLP_GLOAD(0, ra0, rb0);
    ASLP_SSTORE(0, 0, ra0, rb0);
    ASLP_GLOAD(1, ra1, rb1);
    ASLP_GLOAD(2, ra0, rb0);
  }
  __syncthreads();
  if (ktiles > 0) read_frags(0, f0);

  // ablation switches (styles 2,3,4,6,7 give wrong results; they exist to time the pieces, devtools/bench_gemm.py)
  constexpr bool DO_GLOAD = LOOP == 1, DO_SSTORE = LOOP == 1 || LOOP == 2 || LOOP == 6, DO_BARRIER = LOOP == 1 || LOOP == 2 || LOOP == 7,
                 DO_READS = LOOP != 4;
  for (int kt = 0; kt < kt_end; kt += 2) {
    if (DO_SSTORE) ASLP_SSTORE(1, kt + 1, ra1, rb1);      // tile kt+1 -> LDS[1] (its last readers passed the previous barrier)
    if (DO_GLOAD) ASLP_GLOAD(kt + 3, ra1, rb1);
    if (DO_BARRIER) __syncthreads();
    if (DO_READS) read_frags(1, f1);                     // issue; lands under the MFMAs below
    __builtin_amdgcn_sched_barrier(0);
    mma(f0);
    __builtin_amdgcn_sched_barrier(0);
    if (DO_SSTORE) ASLP_SSTORE(0, kt + 2, ra0, rb0);
    if (DO_GLOAD) ASLP_GLOAD(kt + 4, ra0, rb0);
    if (DO_BARRIER) __syncthreads();
    if (DO_READS) read_frags(0, f0);
    __builtin_amdgcn_sched_barrier(0);
    mma(f1);
    __builtin_amdgcn_sched_barrier(0);
  }
  }
#undef ASLP_GLOAD
#undef ASLP_SSTORE

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
  const aslp_gemm_epilogue &ep = g.ep;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int n = 0; n < TN; n++) {
      const int col = n0 + wn * WN + n * 32 + l31;
      if (col >= g.N) continue;
      const float bias = ep.bias ? ep.bias[col] : 0.0f;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int row = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row >= g.M) continue;
        float *cp = g.C + (long)row * g.ldc + col;
        // the arithmetic of gemm_epilogue_store4 (gemm_common.h), spelled the same way
        float v = (ep.W != nullptr || g.beta != 0.0f) ? fmaf(g.alpha, acc[i][n][e], fmaf(g.beta, g.beta != 0.0f ? (ep.c_src ? ep.c_src[(long)row * ep.ld_c_src + col] : *cp) : 0.0f, bias)) : fmaf(g.alpha, acc[i][n][e], bias);
        if (ep.clip > 0.0f) v = fminf(fmaxf(v, -ep.clip), ep.clip);
        *cp = v;
        if (ep.W) ep.W[(long)row * ep.ldw + col] = fmaf(ep.w_alpha, v, ep.W[(long)row * ep.ldw + col]);
        if (ep.act_out) {
          float a = ep.act == 1 ? sigmoid_ref(v) : ep.act == 2 ? tanh_ref(v) : ep.act == 3 ? fmaxf(v, 0.0f) : v;
          ep.act_out[(long)row * ep.ld_act + col] = a;
        }
      }
    }
}

// aslp_gemm_epilogue.colstats by one pass over the finished C, for the kernels that do not form them in their epilogue
// (register-staged tiles): thread = column, blockIdx.y = 32-row group; the same three sums in the same per-group layout.
__global__ void __launch_bounds__(64) colstats_kernel(const float *__restrict__ C, int ldc, int M, int N, double *__restrict__ stats, int ld, int groups) {
  const int col = blockIdx.x * 64 + threadIdx.x, grp = blockIdx.y;
  if (col >= N) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const int r1 = min(M, 32 * grp + 32);
  for (int r = 32 * grp; r < r1; r++) {
    const float v = C[(long)r * ldc + col];
    s0 += (double)v; s1 += (double)(v * v); s2 += (double)v * (double)v;
  }
  double *p = stats + (long)grp * ld + col;
  const long plane = (long)groups * ld;
  p[0] = s0; p[plane] = s1; p[2 * plane] = s2;
}

// ---- profile counters (bench.py roofline) -------------------------------------------------
struct GemmProf {
  long launches = 0;
  double flops = 0.0, ms = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<long> pending_shape;  // parallel to pending: (M << 40) | (N << 20) | K of the launch
  std::map<long, std::pair<long, double>> shape_ms;  // shape -> launches, milliseconds (aslp_gemm_profile_dump)
  std::map<int, double> cfg_flops;  // tile configuration actually launched -> flops it carried
};
GemmProf g_prof[4];
thread_local int t_last_cfg = 0;    // what launch_variant chose for the calling thread's latest product
bool g_prof_on = false;
std::mutex g_prof_mu;
std::vector<hipEvent_t> g_event_pool;  // recycled: no event creation / destruction inside a timed region

hipEvent_t take_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void drain(GemmProf &p) {
  for (size_t i = 0; i < p.pending.size(); i++) {
    auto &ev = p.pending[i];
    float ms = 0.f;
    if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      p.ms += ms;
      auto &sh = p.shape_ms[p.pending_shape[i]];
      sh.first++;
      sh.second += ms;
    }
    g_event_pool.push_back(ev.first);
    g_event_pool.push_back(ev.second);
  }
  p.pending.clear();
  p.pending_shape.clear();
}

int g_force_tile = 0;  // devtools: 0 = heuristic, else 1..5 picks a config

template <int BM, int BN, int BK, int WGM, int WGN, bool A_KC, bool B_KC, bool VEC, int LOOP = 0>
void launch_cfg(GemmArgs &g) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  constexpr int lds_bytes = gemm_lds_bytes<BM, BN, BK, A_KC, B_KC>();
  auto kern = gemm_f32_mfma<BM, BN, BK, WGM, WGN, A_KC, B_KC, VEC, LOOP>;
  static bool attr_set = false;  // one per template instantiation
  if (!attr_set) {
    if (lds_bytes > 48 * 1024)
      ASLP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(64 * WGM * WGN), lds_bytes, cur_stream(), g);
}

template <bool A_KC, bool B_KC, bool VEC>
void launch_variant(GemmArgs &g) {
  // Tile choice: the largest tile that still gives >= ~256 blocks (one per CU); skinny M
  // (the S-row recurrent GEMMs of the LSTM family) gets the 32-row tile.
  auto blocks = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * (g.pair ? 2 : 1); };
  int cfg;
  // Measured on MI355X (devtools/bench_gemm.py, profiles/gemm_tiles_r01.txt): the 8-wave 64x128x32
  // tile is best whenever it tiles the problem into >= ~200 full blocks; otherwise the 4-wave
  // 64x64x32 tile (2+ blocks co-resident per CU) copes best with ragged edges / small grids.
  // Measured on MI355X (devtools/bench_gemm.py, profiles/): the LDS-DMA kernels (cfg >= 200, gemm_glds.hip) win
  // wherever they are eligible; 64x128 / 8 waves when it tiles the problem into >= ~200 full blocks, else 64x64 /
  // 4 waves (2+ workgroups per CU copes best with ragged edges and small grids).  Skinny M keeps the 32-row tile.
  // experiment hook: ASLP_GEMM_TILE_{NT,NN,TN,TT}=<cfg> overrides the choice for one operand layout
  static const int env_tile[4] = {getenv("ASLP_GEMM_TILE_NT") ? atoi(getenv("ASLP_GEMM_TILE_NT")) : 0, getenv("ASLP_GEMM_TILE_NN") ? atoi(getenv("ASLP_GEMM_TILE_NN")) : 0,
                                  getenv("ASLP_GEMM_TILE_TN") ? atoi(getenv("ASLP_GEMM_TILE_TN")) : 0, getenv("ASLP_GEMM_TILE_TT") ? atoi(getenv("ASLP_GEMM_TILE_TT")) : 0};
  const int env_cfg = env_tile[A_KC ? (B_KC ? 0 : 1) : (B_KC ? 3 : 2)];
  const bool big_grid = g.N % 128 == 0 && g.M % 64 == 0 && blocks(64, 128) >= 200;
  if (g_force_tile) cfg = g_force_tile;
  else if (env_cfg && big_grid) cfg = env_cfg;
  // both operands K-contiguous (the forward products): the 4-wave 64x64 tile, two workgroups per CU, measured 3 % ahead of
  // the 8-wave tile inside the training step (devtools/sweep_tiles.sh: 70.0 vs 72.5 us average over the NT launches)
  // neither operand K-contiguous (the weight-gradient products): the same 64 x 128 tile on 4 waves (32 x 64 per wave, half the
  // fragment reads per MFMA); after the epilogue rewrite measured 1 % ahead on cfg2 inside the training step (devtools/sweep_env.sh:
  // 773.3 vs 781.7 k frames/s over three alternations, TN 74.5 vs 73.7 us) and neutral on the LC-BLSTM step
  else if (big_grid && !A_KC && !B_KC) cfg = 208;
  else if (big_grid && !(A_KC && B_KC)) cfg = 212;
  else cfg = 207;
  // (32 x 64 / 32 x 128 tiles for products whose 64 x 64 grid leaves a third of the chip idle were measured on the LC-BLSTM step:
  //  1920 x 256 x 512 16.4 vs 17.1 us, but the long-K members lose their K split: 4.21 vs 3.75 ms per step.  Not taken.)
  t_last_cfg = cfg;
  if (cfg >= 200) {
    int used = cfg;
    if (gemm_glds_launch(g, A_KC, B_KC, cfg, &used)) {  // column sums / column statistics done in-kernel
      t_last_cfg = used;
      if (!A_KC) { g.ep.colsum = nullptr; g.ep1.colsum = nullptr; }
      g.ep.colstats = nullptr;
      return;
    }
    if (g.pair) { t_last_cfg = -1; return; }  // only the LDS-DMA kernels take pairs: the caller issues two single products
    if (cfg != 205 && cfg != 206 && cfg != 207 && cfg != 208 && cfg != 211 && cfg != 212 && cfg != 213) {
      set_error("aslp_sgemm: unknown tile configuration " + std::to_string(cfg) + " (ASLP_GEMM_TILE_* / aslp_gemm_force_tile)");
      return;
    }
    cfg = g.M <= 32 ? 1 : (cfg == 207 ? 7 : 12);  // not eligible: register-staged kernel of the same tile
    t_last_cfg = cfg;
  }
  if (g.pair) { t_last_cfg = -1; return; }
  switch (cfg) {
    case 1: launch_cfg<32, 128, 16, 1, 4, A_KC, B_KC, VEC>(g); break;
    case 2: launch_cfg<64, 64, 16, 2, 2, A_KC, B_KC, VEC>(g); break;
    case 3: launch_cfg<128, 64, 32, 2, 2, A_KC, B_KC, VEC>(g); break;
    case 4: launch_cfg<128, 128, 32, 2, 2, A_KC, B_KC, VEC>(g); break;
    case 6: launch_cfg<128, 64, 64, 2, 2, A_KC, B_KC, VEC>(g); break;
    case 7: launch_cfg<64, 64, 32, 2, 2, A_KC, B_KC, VEC>(g); break;
    case 8: launch_cfg<64, 128, 32, 2, 2, A_KC, B_KC, VEC>(g); break;
    case 9: launch_cfg<64, 64, 64, 2, 2, A_KC, B_KC, VEC>(g); break;
    case 10: launch_cfg<128, 64, 32, 4, 2, A_KC, B_KC, VEC>(g); break;   // 8 waves
    case 11: launch_cfg<128, 128, 32, 2, 4, A_KC, B_KC, VEC>(g); break;  // 8 waves
    case 12: launch_cfg<64, 128, 32, 2, 4, A_KC, B_KC, VEC>(g); break;   // 8 waves
#ifdef ASLP_GEMM_EXPERIMENTS  // devtools builds only (make EXPERIMENTS=1): loop-style variants and the ablation kernels, which
                              // knowingly return wrong results -- never compiled into the product library
    case 107: launch_cfg<64, 64, 32, 2, 2, A_KC, B_KC, VEC, 1>(g); break;   // loop style 1 variants
    case 112: launch_cfg<64, 128, 32, 2, 4, A_KC, B_KC, VEC, 1>(g); break;
    case 101: launch_cfg<32, 128, 16, 1, 4, A_KC, B_KC, VEC, 1>(g); break;
    case 108: launch_cfg<64, 128, 32, 2, 2, A_KC, B_KC, VEC, 1>(g); break;
    case 104: launch_cfg<128, 128, 32, 2, 2, A_KC, B_KC, VEC, 1>(g); break;
    case 111: launch_cfg<128, 128, 32, 2, 4, A_KC, B_KC, VEC, 1>(g); break;
    case 116: launch_cfg<64, 128, 16, 2, 2, A_KC, B_KC, VEC, 1>(g); break;
    case 120: launch_cfg<64, 128, 64, 2, 4, A_KC, B_KC, VEC, 1>(g); break;
    case 152: launch_cfg<64, 128, 32, 2, 4, A_KC, B_KC, VEC, VEC ? 5 : 0>(g); break;   // interleaved loop
    case 157: launch_cfg<64, 64, 32, 2, 2, A_KC, B_KC, VEC, VEC ? 5 : 0>(g); break;
    case 158: launch_cfg<64, 128, 32, 2, 2, A_KC, B_KC, VEC, VEC ? 5 : 0>(g); break;
    case 154: launch_cfg<128, 128, 32, 2, 2, A_KC, B_KC, VEC, VEC ? 5 : 0>(g); break;
    case 151: launch_cfg<128, 128, 32, 2, 4, A_KC, B_KC, VEC, VEC ? 5 : 0>(g); break;
    case 132: launch_cfg<64, 128, 32, 2, 4, A_KC, B_KC, VEC, 2>(g); break;   // ablations (wrong results): no global loads
    case 133: launch_cfg<64, 128, 32, 2, 4, A_KC, B_KC, VEC, 3>(g); break;   //   + no LDS stores / barriers
    case 134: launch_cfg<64, 128, 32, 2, 4, A_KC, B_KC, VEC, 4>(g); break;   //   + no LDS reads (MFMA only)
    case 136: launch_cfg<64, 128, 32, 2, 4, A_KC, B_KC, VEC, 6>(g); break;   //   reads + stores, no barrier, no global loads
    case 137: launch_cfg<64, 128, 32, 2, 4, A_KC, B_KC, VEC, 7>(g); break;   //   reads + barrier, no stores, no global loads
    case 121: launch_cfg<64, 128, 64, 2, 2, A_KC, B_KC, VEC, 1>(g); break;
    case 122: launch_cfg<64, 128, 64, 2, 4, A_KC, B_KC, VEC, 0>(g); break;
    case 117: launch_cfg<128, 64, 32, 2, 2, A_KC, B_KC, VEC, 1>(g); break;
#endif
    case 13: launch_cfg<128, 128, 16, 2, 2, A_KC, B_KC, VEC>(g); break;
    default: set_error("aslp_sgemm: unknown tile configuration " + std::to_string(cfg) + " (ASLP_GEMM_TILE_* / aslp_gemm_force_tile)"); break;
  }
}

template <bool A_KC, bool B_KC>
void launch_aligned(GemmArgs &g) {
  // vector path: 16-byte aligned operands whose contiguous extent is a multiple of 4 floats
  const int a_ext = A_KC ? g.K : g.M, b_ext = B_KC ? g.K : g.N;
  const bool vec = g.a_vec && g.b_vec && a_ext % 4 == 0 && b_ext % 4 == 0 && a_ext >= 4 && b_ext >= 4;
  if (vec) launch_variant<A_KC, B_KC, true>(g);
  else launch_variant<A_KC, B_KC, false>(g);
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

// aslp_sgemm_ex with optional prepared planes of either operand (split16.h); an operand without planes is converted in scratch
static int sgemm_impl(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb, float beta,
                      float *C, int ldc, const aslp_gemm_epilogue *ep, const S16View *pa, const S16View *pb) {
  gemm_split16_reset_last_parts();
  if (M < 0 || N < 0 || K < 0) return -1;
  if (M == 0 || N == 0) return 0;
  if (!C || ldc < N) return -2;
  if (K > 0 && (!A || !B)) return -3;
  if (K > 0 && (lda < (transA ? M : K) || ldb < (transB ? K : N))) return -4;
  GemmArgs g;
  g.split_k = 0; g.k_chunk = 0; g.split_stride = 0;
  g.pair = 0; g.A1 = g.B1 = nullptr; g.C1 = nullptr; g.ep1 = aslp_gemm_epilogue();
  static const int wide = [] { const char *e = getenv("ASLP_GEMM_WIDE_EPI"); return e ? atoi(e) : 1; }();
  g.wide_epilogue = wide;
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta;
  if (ep) g.ep = *ep; else g.ep = aslp_gemm_epilogue();  // zero-initialised: every optional piece off
  if (g.ep.colsum && !transA) return -5;  // column sums are defined for transposed A only
  if (g.ep.colstats && (beta != 0.0f || g.ep.W || g.ep.colstats_ld < N)) return -6;  // statistics of a plain forward product only
  if (g.ep.c_src && g.ep.ld_c_src < N) return -7;
  g.a_vec = aligned16(A) && lda % 4 == 0;
  g.b_vec = aligned16(B) && ldb % 4 == 0;
  // report order: 0 = NT, 1 = NN, 2 = TN, 3 = TT
  int slot = (!transA && transB) ? 0 : (!transA && !transB) ? 1 : (transA && !transB) ? 2 : 3;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool prof = g_prof_on;
  if (prof) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    e0 = take_event();
    e1 = take_event();
  }
  if (prof) (void)hipEventRecord(e0, cur_stream());
  static const int s16_cfg = [] { const char *e = getenv("ASLP_GEMM_SPLIT_F16_TILE"); return e ? atoi(e) : 0; }();
  if (gemm_split16_enabled() && !g_force_tile && gemm_split16_launch(g, !transA, transB != 0, s16_cfg, pa, pb)) {   // (a forced tile asks for an fp32 kernel by name)
    g.ep.colstats = nullptr;   // formed in its epilogue, like the column sums of a transposed A
    g.ep.colsum = nullptr;
    t_last_cfg = gemm_split16_last_tile();
  } else if (!transA && transB) launch_aligned<true, true>(g);
  else if (!transA && !transB) launch_aligned<true, false>(g);
  else if (transA && !transB) launch_aligned<false, false>(g);
  else launch_aligned<false, true>(g);
  if (g.ep.colstats) {  // the chosen kernel does not form them in its epilogue: one pass over the finished C
    const int groups = (M + 31) / 32;
    hipLaunchKernelGGL(colstats_kernel, dim3((N + 63) / 64, groups), dim3(64), 0, cur_stream(), C, ldc, M, N, g.ep.colstats, g.ep.colstats_ld, groups);
  }
  if (g.ep.colsum) {  // the chosen kernel could not fold the column sums in: one extra pass over A (same values)
    MatrixDim da = {K, M, lda};
    if (g.ep.colsum_w) aslp_add_row_sum_mat_vec_sgd(1.0f, A, da, g.ep.colsum_beta, g.ep.colsum, g.ep.colsum_w, g.ep.colsum_w_alpha);
    else aslp_add_row_sum_mat_vec(1.0f, A, da, g.ep.colsum_beta, g.ep.colsum);
  }
  check_launch("aslp_sgemm");
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[slot].launches++;
    g_prof[slot].flops += 2.0 * (double)M * (double)N * (double)K;
    g_prof[slot].cfg_flops[t_last_cfg] += 2.0 * (double)M * (double)N * (double)K;
    if (prof) {
      (void)hipEventRecord(e1, cur_stream());
      g_prof[slot].pending.emplace_back(e0, e1);
      g_prof[slot].pending_shape.push_back(((long)M << 40) | ((long)N << 20) | (long)K);
    }
  }
  return 0;
}

int aslp_sgemm_ex(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb, float beta,
                  float *C, int ldc, const aslp_gemm_epilogue *ep) {
  return sgemm_impl(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, ep, nullptr, nullptr);
}

// ---- prepared operand planes (include/aslp_kernels.h) ---------------------------------------------------------------------------
aslp_planes *aslp_planes_new(void) { return reinterpret_cast<aslp_planes *>(new PlaneSet()); }
void aslp_planes_free(aslp_planes *p) { delete reinterpret_cast<PlaneSet *>(p); }
int aslp_planes_convert(aslp_planes *p, const float *src, MatrixDim d) {
  if (!p || !src) return -1;
  const bool ok = reinterpret_cast<PlaneSet *>(p)->ConvertFrom(src, d.rows, d.cols, d.stride);
  check_launch("aslp_planes_convert");
  return ok ? 0 : -2;
}
void aslp_planes_reserve(aslp_planes *p, int rows, int cols) { if (p) reinterpret_cast<PlaneSet *>(p)->Reserve(rows, cols); }
void aslp_planes_set_bound(aslp_planes *p, float bound) { if (p) reinterpret_cast<PlaneSet *>(p)->SetBound(bound); }
void aslp_planes_as_output(const aslp_planes *p, aslp_planes_out *out) {
  if (!out) return;
  *out = aslp_planes_out();
  if (!p) return;
  const PlaneSet *ps = reinterpret_cast<const PlaneSet *>(p);
  const S16View v = ps->View();
  out->hi = v.hi; out->lo = v.lo; out->ld = v.ld; out->slot = v.slot; out->parts = ps->Parts(); out->nparts = 0; out->planes_written = 0;
}
int aslp_sgemm_planes_ex(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, const aslp_planes *pa, const float *B,
                         int ldb, const aslp_planes *pb, float beta, float *C, int ldc, const aslp_gemm_epilogue *ep) {
  S16View va, vb;
  if (pa) va = reinterpret_cast<const PlaneSet *>(pa)->View();
  if (pb) vb = reinterpret_cast<const PlaneSet *>(pb)->View();
  return sgemm_impl(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, ep, pa ? &va : nullptr, pb ? &vb : nullptr);
}

// Two products of one shape in one launch (include/aslp_kernels.h).  Falls back to two launches wherever the paired kernel does
// not apply (unaligned operands, K % 4, column sums on a kernel that cannot fold them, the register-staged tiles).
int aslp_sgemm_pair_ex(int transA, int transB, int M, int N, int K, float alpha, const float *A0, const float *A1, int lda, const float *B0,
                       const float *B1, int ldb, float beta, float *C0, float *C1, int ldc, const aslp_gemm_epilogue *ep0,
                       const aslp_gemm_epilogue *ep1) {
  return aslp::sgemm_pair_views(transA, transB, M, N, K, alpha, A0, A1, lda, B0, B1, ldb, beta, C0, C1, ldc, ep0, ep1, nullptr, nullptr, nullptr, nullptr);
}
}  // extern "C"

// the pair with prepared planes (windows) of all four operands, or of none (csrc/split16.h)
int aslp::sgemm_pair_views(int transA, int transB, int M, int N, int K, float alpha, const float *A0, const float *A1, int lda, const float *B0,
                           const float *B1, int ldb, float beta, float *C0, float *C1, int ldc, const aslp_gemm_epilogue *ep0,
                           const aslp_gemm_epilogue *ep1, const S16View *va0, const S16View *va1, const S16View *vb0, const S16View *vb1) {
  static const int enabled = [] { const char *e = getenv("ASLP_GEMM_PAIR"); return e ? atoi(e) : 1; }();
  const bool views = va0 && va1 && vb0 && vb1;
  auto two = [&]() {
    const int rc = sgemm_impl(transA, transB, M, N, K, alpha, A0, lda, B0, ldb, beta, C0, ldc, ep0, views ? va0 : nullptr, views ? vb0 : nullptr);
    return rc ? rc : sgemm_impl(transA, transB, M, N, K, alpha, A1, lda, B1, ldb, beta, C1, ldc, ep1, views ? va1 : nullptr, views ? vb1 : nullptr);
  };
  if (M <= 0 || N <= 0 || K <= 0 || !A0 || !A1 || !B0 || !B1 || !C0 || !C1 || ldc < N) return two();  // argument errors are reported there
  if (lda < (transA ? M : K) || ldb < (transB ? K : N)) return two();
  const bool colsum = (ep0 && (ep0->colsum || ep0->colstats)) || (ep1 && (ep1->colsum || ep1->colstats));
  if (!enabled || g_prof_on || colsum) return two();   // per-launch event timing and the column-sum fallback work on single products
  GemmArgs g;
  g.split_k = 0; g.k_chunk = 0; g.split_stride = 0;
  static const int wide = [] { const char *e = getenv("ASLP_GEMM_WIDE_EPI"); return e ? atoi(e) : 1; }();
  g.wide_epilogue = wide;
  g.A = A0; g.B = B0; g.C = C0; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta;
  g.ep = ep0 ? *ep0 : aslp_gemm_epilogue();
  g.pair = 1; g.A1 = A1; g.B1 = B1; g.C1 = C1;
  g.ep1 = ep1 ? *ep1 : aslp_gemm_epilogue();
  g.a_vec = aligned16(A0) && aligned16(A1) && lda % 4 == 0;
  g.b_vec = aligned16(B0) && aligned16(B1) && ldb % 4 == 0;
  if (!g.a_vec || !g.b_vec) return two();
  gemm_split16_reset_last_parts();
  if (views && gemm_split16_serves(M, N, K) && !g_force_tile && K % 64 == 0 &&
      gemm_split16_planes_launch(g, !transA, transB != 0, *va0, *vb0, va1, vb1, 0)) {
    t_last_cfg = gemm_split16_last_tile();
    check_launch("aslp_sgemm_pair (split-fp16)");
    const int slot = (!transA && transB) ? 0 : (!transA && !transB) ? 1 : (transA && !transB) ? 2 : 3;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[slot].launches += 2;
    g_prof[slot].flops += 4.0 * (double)M * (double)N * (double)K;
    g_prof[slot].cfg_flops[t_last_cfg] += 4.0 * (double)M * (double)N * (double)K;
    return 0;
  }
  if (!transA && transB) launch_aligned<true, true>(g);
  else if (!transA && !transB) launch_aligned<true, false>(g);
  else if (transA && !transB) launch_aligned<false, false>(g);
  else launch_aligned<false, true>(g);
  if (t_last_cfg < 0) return two();
  check_launch("aslp_sgemm_pair");
  {
    const int slot = (!transA && transB) ? 0 : (!transA && !transB) ? 1 : (transA && !transB) ? 2 : 3;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[slot].launches += 2;
    g_prof[slot].flops += 4.0 * (double)M * (double)N * (double)K;
    g_prof[slot].cfg_flops[t_last_cfg] += 4.0 * (double)M * (double)N * (double)K;
  }
  return 0;
}

extern "C" {
int aslp_sgemm(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb, float beta,
               float *C, int ldc) {
  return aslp_sgemm_ex(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, nullptr);
}

void aslp_gemm_force_tile(int cfg) { g_force_tile = cfg; }
int aslp_gemm_last_tile(void) { return t_last_cfg; }
void aslp_gemm_profile(int enable) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = enable != 0;
  if (g_prof_on)  // events for ~100 training steps up front, so the timed region only records
    while (g_event_pool.size() < 4096) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) break;
      g_event_pool.push_back(e);
    }
}
void aslp_gemm_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto &p : g_prof) {
    drain(p);
    p.launches = 0;
    p.flops = 0.0;
    p.ms = 0.0;
    p.cfg_flops.clear();
    p.shape_ms.clear();
  }
}
long aslp_gemm_profile_get(int variant, double *flops, double *ms) {
  if (variant < 0 || variant > 3) return -1;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  drain(g_prof[variant]);
  if (flops) *flops = g_prof[variant].flops;
  if (ms) *ms = g_prof[variant].ms;
  return g_prof[variant].launches;
}
void aslp_gemm_profile_dump(void) {  // devtools: per-shape table of the event-timed launches since the last reset, to stderr
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const char *names[4] = {"NT", "NN", "TN", "TT"};
  for (int v = 0; v < 4; v++) {
    drain(g_prof[v]);
    for (auto &kv : g_prof[v].shape_ms) {
      const long M = kv.first >> 40, N = (kv.first >> 20) & 0xFFFFF, K = kv.first & 0xFFFFF;
      const double us = kv.second.second * 1e3 / kv.second.first;
      std::fprintf(stderr, "gemm %s %6ld x %6ld x %6ld  launches %5ld  avg %8.2f us  %7.1f TFLOP/s\n", names[v], M, N, K, kv.second.first, us,
                   2.0 * M * N * K / us / 1e6);
    }
  }
}
int aslp_gemm_profile_tile(int variant, char *buf, int buflen) {
  if (variant < 0 || variant > 3) return 0;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int best = 0;
  double fl = -1.0;
  for (auto &kv : g_prof[variant].cfg_flops)
    if (kv.second > fl) { fl = kv.second; best = kv.first; }
  const char *d = "";
  switch (best) {  // the tables of launch_variant (register-staged) and gemm_glds_launch (LDS-DMA)
    case 1: d = "gemm_f32_mfma 32x128x16, 4 waves, register-staged"; break;
    case 7: d = "gemm_f32_mfma 64x64x32, 4 waves, register-staged"; break;
    case 12: d = "gemm_f32_mfma 64x128x32, 8 waves, register-staged"; break;
    case 205: d = "gemm_f32_glds 32x128x32, 4 waves, LDS-DMA, 4 stages"; break;
    case 206: d = "gemm_f32_glds 32x64x32, 2 waves, LDS-DMA, 4 stages"; break;
    case 207: d = "gemm_f32_glds 64x64x32, 4 waves, LDS-DMA, 4 stages"; break;
    case 208: d = "gemm_f32_glds 64x128x32, 4 waves, LDS-DMA, 3 stages"; break;
    case 211: d = "gemm_f32_glds 128x128x32, 8 waves, LDS-DMA, 3 stages"; break;
    case 212: d = "gemm_f32_glds 64x128x32, 8 waves, LDS-DMA, 3 stages"; break;
    case 213: d = "gemm_f32_glds 64x128x32, 8 waves, LDS-DMA, 4 stages"; break;
    case 304: d = "gemm_s16_glds 32x64x64 halves, 2 waves, v_mfma_f32_32x32x16_f16 x3 on two-piece fp32 operands, LDS-DMA, 3 stages"; break;
    case 308: d = "gemm_s16_glds 64x128x64 halves, 4 waves, v_mfma_f32_32x32x16_f16 x3 on two-piece fp32 operands, LDS-DMA, 3 stages"; break;
    case 311: d = "gemm_s16_glds 128x128x64 halves, 4 waves, v_mfma_f32_32x32x16_f16 x3 on two-piece fp32 operands, LDS-DMA, 2 stages"; break;
    case 351: d = "gemm_s16_pc 128x128x64 halves, 4 consumer + 4 producer waves, v_mfma_f32_32x32x16_f16 x3 on two-piece fp32 operands, LDS-DMA, 2 stages"; break;
    case 328: d = "gemm_s16_ks128 128x128x32 halves, 4 waves, v_mfma_f32_32x32x16_f16 x3 on two-piece fp32 operands, transposing LDS reads, ring of 4"; break;
    default: d = "gemm_f32_mfma (devtools tile)"; break;
  }
  if (buf && buflen > 0) { std::strncpy(buf, d, buflen - 1); buf[buflen - 1] = 0; }
  return best;
}

}  // extern "C"
