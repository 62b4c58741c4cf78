// gemm_split16.hip -- fp32 GEMM carried on the fp16 matrix instruction (v_mfma_f32_32x32x16_f16) with fp32-equivalent operands.
//
// The fp32 MFMA rate of gfx950 is 256 FLOP/clk/CU (157 TFLOP/s) whatever the instruction shape; the fp16 instruction runs 16 x that.
// An fp32 value x is carried as TWO fp16 pieces behind a power-of-two scale s of its matrix (split16.h):  x s = hi + 2^-11 lo'.
// A product a b then is  hi_a hi_b + 2^-11 (hi_a lo'_b + lo'_a hi_b)  (+ 2^-22 lo'_a lo'_b, dropped: below the pieces' own rounding):
// three instructions per 16-wide k step, every partial product exact in the multiplier, accumulated in fp32 in two accumulators (main and
// cross terms) that are joined once in front of the epilogue.  Measured against double (tests/test_gemm_split16_gpu.py) the result is
// closer than the fp32 instruction's.
//
// Operands reach the kernel as PLANES in the layout of their matrix (split16.h): made once per tensor and step -- by the kernel that
// writes the tensor where that kernel knows a bound of it, else by split16_convert_kernel -- and read by every product the tensor
// takes part in, whichever index that product reduces over:
//   * reduction index contiguous ("KC": x in the forward product, dy in the in-diff, W in the forward product): LDS image
//     [rows][64 halves], 128-byte rows with gemm_glds.hip's chunk swizzle, fragments by ds_read_b128;
//   * reduction index = row index ("KS": dy and x in the weight gradient, W in the in-diff): LDS image [64 k][BR halves] exactly as the
//     rows lie in memory, fragments by TWO ds_read_b64_tr_b16 -- each 16-lane group reads a [4 k][16 columns] block and receives it
//     transposed, lane (column) p holding k .. k+3 (checked lane by lane on the device: devtools/micro/tr_read.hip).  The 64-byte
//     column groups of a k row are XOR-swizzled with the row's low bits so that the four rows a 32-lane group touches fall on four
//     different quarters of the 64 banks.
// Everything else is gemm_glds.hip's pipeline: LDS-DMA (global_load_lds_dwordx4) into NS stages, counted vmcnt, fragments double
// buffered in registers, the epilogues of gemm_common.h unchanged (the 32 x 32 fp16 instruction has the fp32 one's result layout).
//
// A/B: ASLP_GEMM_SPLIT_F16=0 keeps every product on the fp32 instruction (default: on).
#include <algorithm>
#include <atomic>

#include "gemm_common.h"
#include "scratch.h"
#include "split16.h"

#pragma clang diagnostic ignored "-Winline-asm"  // the DMA asm clobbers m0 on purpose

namespace aslp {
namespace {

thread_local int t_last_cfg_s16 = 0;   // tile the calling thread's latest split-fp16 product ran on: 311 = 128x128, 351 = the same with producer / consumer waves, 308 = 64x128, 328 = 128x128 both operands reduction-major
thread_local int t_last_parts = 0;   // per-wave maxima the calling thread's latest product left (aslp_gemm_last_parts)
constexpr int BKH = 64;       // halves per K tile
constexpr int KH = BKH / 16;  // instruction k steps per tile
typedef __attribute__((address_space(3))) char lds_char;
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4v lds_short4;

__device__ __forceinline__ int kc_swizzle(int row) { return (row >> 1) & 7; }   // as gemm_glds.hip
// KS image: XOR of the 64-byte column-group index of k row `krow` (BR halves per row)
template <int BR>
__device__ __forceinline__ int ks_swizzle(int krow) { return BR == 64 ? ((krow >> 1) & 1) : BR == 128 ? (krow & 3) : 0; }
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory", "m0");
}

// ---- largest finite |x| of up to two matrices (blockIdx.y): every workgroup leaves its own maximum in part[workgroup] -- no atomics,
// nothing to zero first; the conversion reduces the partials.  Rows are dealt to the workgroups in contiguous chunks; a row is
// covered by tw = 2^tw_log2 threads with 16 bytes each per pass (no division anywhere, every load of a thread independent).
struct MaxJob { const float *p; int rows, cols, ld; float *part; };
struct MaxJobs { MaxJob j[kS16MaxJobs]; };
__global__ void __launch_bounds__(256) s16_absmax_kernel(MaxJobs jobs, int tw_log2) {
  const MaxJob j = jobs.j[blockIdx.y];
  const int c4 = j.cols >> 2, tw = 1 << tw_log2, rpw = 256 >> tw_log2;
  const int tr = threadIdx.x >> tw_log2, tc = threadIdx.x & (tw - 1);
  const int per = (j.rows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(j.rows, r0 + per);
  float m = 0.f;
  for (int r = r0 + tr; r < r1; r += rpw) {
    const float *row = j.p + (long)r * j.ld;
#pragma unroll 4
    for (int c = tc; c < c4; c += tw) m = s16_absmax4(m, *reinterpret_cast<const float4 *>(row + 4 * c));
  }
  m = wave_max(m);
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) j.part[blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}
// the matrix maximum from the partials (every thread of a 256-thread workgroup gets it)
__device__ __forceinline__ float reduce_parts(const float *part, int nparts) {
  __shared__ float wm2[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm2[threadIdx.x >> 6] = m;
  __syncthreads();
  const float r = fmaxf(fmaxf(wm2[0], wm2[1]), fmaxf(wm2[2], wm2[3]));
  __syncthreads();   // (wm2 may be reused by a second reduction)
  return r;
}

// ---- bound of the weights after the next fused step (include/aslp_kernels.h aslp_weight_bound): one workgroup
struct BoundJob { const float *w_parts; int n_w; const float *c_parts; int n_c; const unsigned *slot_a, *slot_b; float k, alpha, beta, w_alpha, clip; unsigned *slot_out; };
__global__ void __launch_bounds__(256) s16_weight_bound_kernel(BoundJob j) {
  const float mw = reduce_parts(j.w_parts, j.n_w);
  const float mc = j.n_c > 0 ? reduce_parts(j.c_parts, j.n_c) : 0.f;
  if (threadIdx.x == 0) {
    const float ba = __uint_as_float(*j.slot_a), bb = __uint_as_float(*j.slot_b);
    float v = fabsf(j.alpha) * j.k * ba * bb + fabsf(j.beta) * mc;    // |alpha A^T B + beta C_old| <= alpha K max|a| max|b| + beta max|C_old|
    if (j.clip > 0.f) v = fminf(v, j.clip);
    *j.slot_out = __float_as_uint(mw + fabsf(j.w_alpha) * v);
  }
}

// ---- fp32 matrix -> planes in the matrix' own layout (padding written as zeros), up to two matrices per launch (blockIdx.y) --------
// Same dealing of rows; a thread converts 8 consecutive columns per pass (two 16-byte loads, one 16-byte store per plane).
struct ConvJob { const float *src; int ld_src; S16View pl; const float *part; int nparts; };
struct ConvJobs { ConvJob j[kS16MaxJobs]; };
__global__ void __launch_bounds__(256) split16_convert_kernel(ConvJobs jobs, int tw_log2) {
  const ConvJob j = jobs.j[blockIdx.y];
  const int k8 = j.pl.ld >> 3, tw = 1 << tw_log2, rpw = 256 >> tw_log2;   // ld is a multiple of 64
  const int tr = threadIdx.x >> tw_log2, tc = threadIdx.x & (tw - 1);
  const int rows_p = (j.pl.rows + kS16Pad - 1) / kS16Pad * kS16Pad;
  const int per = (rows_p + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(rows_p, r0 + per);
  // A workgroup's share of a minibatch-sized matrix is a few 8-column pieces per thread: their loads go out before the maximum is read, so
  // the kernel is one round trip to memory deep instead of one for the partial maxima and one per row pass.
  constexpr int kPre = 4;
  const int units = max(r1 - r0, 0) * k8;
  if (units <= kPre * 256) {
    float4 a[kPre][2];
#pragma unroll
    for (int i = 0; i < kPre; i++) {
      const int u = (int)threadIdx.x + i * 256, ur = u / k8, r = r0 + ur, c = u - ur * k8;
      const bool ok = u < units && r < j.pl.rows;
      const float *row = j.src + (long)r * j.ld_src;
      a[i][0] = ok && 8 * c < j.pl.cols ? *reinterpret_cast<const float4 *>(row + 8 * c) : float4{0.f, 0.f, 0.f, 0.f};
      a[i][1] = ok && 8 * c + 4 < j.pl.cols ? *reinterpret_cast<const float4 *>(row + 8 * c + 4) : float4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned mbits = __float_as_uint(reduce_parts(j.part, j.nparts));
    if (blockIdx.x == 0 && threadIdx.x == 0) *j.pl.slot = mbits;
    const float s = ldexpf(1.f, s16_exponent(mbits));
#pragma unroll
    for (int i = 0; i < kPre; i++) {
      const int u = (int)threadIdx.x + i * 256, ur = u / k8, r = r0 + ur, c = u - ur * k8;
      if (u >= units) break;
      half4 h0, l0, h1, l1;
      s16_split4(a[i][0], s, &h0, &l0);   // (padding: zeros split into zeros)
      s16_split4(a[i][1], s, &h1, &l1);
      *reinterpret_cast<half8 *>(j.pl.hi + (long)r * j.pl.ld + 8 * c) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      *reinterpret_cast<half8 *>(j.pl.lo + (long)r * j.pl.ld + 8 * c) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    return;
  }
  const unsigned mbits = __float_as_uint(reduce_parts(j.part, j.nparts));
  if (blockIdx.x == 0 && threadIdx.x == 0) *j.pl.slot = mbits;   // for the products (launched behind this kernel)
  const float s = ldexpf(1.f, s16_exponent(mbits));
  for (int r = r0 + tr; r < r1; r += rpw) {
    const float *row = j.src + (long)r * j.ld_src;
    const bool row_ok = r < j.pl.rows;
#pragma unroll 2
    for (int c = tc; c < k8; c += tw) {
      half4 h0 = {0, 0, 0, 0}, l0 = {0, 0, 0, 0}, h1 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
      if (row_ok && 8 * c < j.pl.cols) s16_split4(*reinterpret_cast<const float4 *>(row + 8 * c), s, &h0, &l0);          // cols % 4 == 0
      if (row_ok && 8 * c + 4 < j.pl.cols) s16_split4(*reinterpret_cast<const float4 *>(row + 8 * c + 4), s, &h1, &l1);
      *reinterpret_cast<half8 *>(j.pl.hi + (long)r * j.pl.ld + 8 * c) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      *reinterpret_cast<half8 *>(j.pl.lo + (long)r * j.pl.ld + 8 * c) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
}
// threads per row for `units` 16-byte (absmax) / 32-byte (convert) pieces per row
inline int tw_log2_for(int units) {
  int l = 0;
  while (l < 8 && (1 << l) < units) l++;
  return l;
}

// ---- shared by the product kernels ------------------------------------------------------------------------------------------------
// (EXTRA) the bound of the weights after the fused step: |W + w_alpha C| <= max |W| + |w_alpha| (|alpha| K max|a| max|b| + |beta| max |C_old|),
// the maxima from the previous step's per-workgroup partials.  Every wave forms it; workgroup 0 stores it for the planes' readers.
__device__ __forceinline__ float s16_weight_bound(const GemmArgs &g, const S16View &va, const S16View &vb, int lane) {
  if (g.ep.bound_w_parts == nullptr) return 0.f;   // uniform
  // (four independent loads per array and pass: a dependent load per element here once cost every weight-gradient launch 50 us)
  float mw = 0.f, mc = 0.f;
  const int n = g.ep.bound_n;
  const float *wp = g.ep.bound_w_parts, *cp = g.ep.bound_c_parts;
  for (int i = lane; i < n; i += 256) {
    const int i1 = i + 64, i2 = i + 128, i3 = i + 192;
    const float a0 = wp[i], a1 = i1 < n ? wp[i1] : 0.f, a2 = i2 < n ? wp[i2] : 0.f, a3 = i3 < n ? wp[i3] : 0.f;
    mw = fmaxf(fmaxf(mw, fmaxf(a0, a1)), fmaxf(a2, a3));
    if (cp != nullptr) {
      const float c0 = cp[i], c1 = i1 < n ? cp[i1] : 0.f, c2 = i2 < n ? cp[i2] : 0.f, c3 = i3 < n ? cp[i3] : 0.f;
      mc = fmaxf(fmaxf(mc, fmaxf(c0, c1)), fmaxf(c2, c3));
    }
  }
  mw = wave_max(mw);
  mc = wave_max(mc);
  float v = fabsf(g.alpha) * (float)g.K * __uint_as_float(*va.slot) * __uint_as_float(*vb.slot) + fabsf(g.beta) * mc;
  if (g.ep.clip > 0.f) v = fminf(v, g.ep.clip);
  const float w_bound = mw + fabsf(g.ep.w_alpha) * v;
  if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) *const_cast<unsigned *>(g.ep.planes.slot) = __float_as_uint(w_bound);
  return w_bound;
}

// Everything behind the K loop: join the two accumulators and undo the operand scales, the column sums of a reduction-major A
// (COLSUM), the column statistics, the epilogue (with the planes / maxima of its output when EXTRA).  (row0, col0): this wave's patch.
template <int TM, int TN, int NW, bool EXTRA, bool COLSUM>
__device__ __forceinline__ void s16_finish(const GemmArgs &g, const S16View &va, const S16View &vb, f32x16 (&acc)[TM][TN], const f32x16 (&accx)[TM][TN],
                                           const float (&asum)[TM], bool do_colsum, float w_bound, int row0, int col0, int lane, int wave, float *lds) {
  const int l31 = lane & 31, lh = lane >> 5;
  // 2^-(up_a + up_b) in two exact factors (either alone may leave fp32's range where their product with the accumulator does not)
  {
    const int e = -(s16_exponent(*va.slot) + s16_exponent(*vb.slot));
    const float s1 = ldexpf(1.f, e / 2), s2 = ldexpf(1.f, e - e / 2);
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[i][j][q] = (fmaf(accx[i][j][q], 0x1p-11f, acc[i][j][q]) * s1) * s2;
  }
  if constexpr (COLSUM) {
    if (do_colsum) {
      const float inv_a = ldexpf(1.f, -s16_exponent(*va.slot));
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const float s_all = (asum[i] + __shfl_xor(asum[i], 32, 64)) * inv_a;  // the two lane halves hold disjoint k subsets
        const int row = row0 + i * 32 + l31;
        if (lh == 0 && row < g.M) {
          float v = s_all;
          if (g.ep.colsum_beta != 0.0f) v += g.ep.colsum_beta * g.ep.colsum[row];
          g.ep.colsum[row] = v;
          if (g.ep.colsum_w) g.ep.colsum_w[row] += g.ep.colsum_w_alpha * v;
        }
      }
    }
  }
  if (g.ep.colstats != nullptr) gemm_colstats<TM, TN>(g, acc, row0, col0, l31, lh);  // uniform
  // planes of an output and per-workgroup maxima for the products that will read it (aslp_gemm_epilogue.planes / *_parts)
  EpiExtra xtra;
  if constexpr (EXTRA) {
    if (g.ep.planes_of != 0 && g.ep.planes.hi != nullptr)
      xtra.pscale = ldexpf(1.f, s16_exponent(g.ep.bound_w_parts != nullptr ? __float_as_uint(w_bound) : *g.ep.planes.slot));
  }
  if (g.wide_epilogue && gemm_epilogue_wide_ok(g)) {  // uniform
    __builtin_amdgcn_s_barrier();
    gemm_epilogue_wide<TM, TN>(g, acc, row0, col0, lane, lds + wave * 32 * kEpiPitch, xtra, EXTRA);
  } else {
    gemm_epilogue<TM, TN>(g, acc, row0, col0, l31, lh, xtra, false);   // (the host asks for planes / maxima only where the wide epilogue applies)
  }
  if (EXTRA && (g.ep.wmax_parts != nullptr || g.ep.cmax_parts != nullptr)) {   // one maximum per workgroup: the waves meet in LDS
    const float wmx = wave_max(xtra.wmax), cmx = wave_max(xtra.cmax);
    __builtin_amdgcn_s_barrier();   // every wave is done with its epilogue slice of the LDS
    if (lane == 0) { lds[2 * wave] = wmx; lds[2 * wave + 1] = cmx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float w = lds[0], c = lds[1];
#pragma unroll
      for (int q = 1; q < NW; q++) { w = fmaxf(w, lds[2 * q]); c = fmaxf(c, lds[2 * q + 1]); }
      const int idx = (int)blockIdx.z * (int)gridDim.x + (int)blockIdx.x;
      if (g.ep.wmax_parts) g.ep.wmax_parts[idx] = w;
      if (g.ep.cmax_parts) g.ep.cmax_parts[idx] = c;
    }
  }
}
// column sums of a reduction-major A from the fragments a wave multiplies anyway: sum over the 8 k of a fragment of hi + 2^-11 lo'
__device__ __forceinline__ float s16_frag_sum(float sum, const half8 hi, const half8 lo) {
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  const half2v one = {(h16)1.0f, (h16)1.0f}, eps = {(h16)0x1p-11f, (h16)0x1p-11f};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const half2v hv = {hi[2 * q], hi[2 * q + 1]}, lv = {lo[2 * q], lo[2 * q + 1]};
    sum = __builtin_amdgcn_fdot2(hv, one, sum, false);     // fp32 accumulation of exact fp16 values
    sum = __builtin_amdgcn_fdot2(lv, eps, sum, false);
  }
  return sum;
}

// ---- the product ---------------------------------------------------------------------------------------------------------------
struct S16Operands { S16View a, b, a1, b1; int kp; };   // a1 / b1: second product of a pair (blockIdx.z == 1)
// one of two views, member by member: assigning a whole struct under a condition makes hipcc park both in scratch memory (88 bytes per
// lane and a private segment on every product kernel)
__device__ __forceinline__ S16View s16_pick(bool second, const S16View &x, const S16View &y) {
  S16View v;
  v.hi = second ? +y.hi : +x.hi;
  v.lo = second ? +y.lo : +x.lo;
  v.ld = second ? +y.ld : +x.ld;
  v.rows = second ? +y.rows : +x.rows;
  v.cols = second ? +y.cols : +x.cols;
  v.slot = second ? +y.slot : +x.slot;
  return v;
}

// ABL (devtools/micro/s16_ablate.hip only; 0 in the library): 1 = no MFMA, 2 = no DMA, 4 = no LDS reads -- wrong results, for timing
// EXTRA: the epilogue also leaves planes / maxima of its output (aslp_gemm_epilogue.planes, *_parts); a variant of its own because the
// extra epilogue state costs the 128 x 128 tile its last registers.
template <int BM, int BN, int WGM, int WGN, int NS, bool A_KC, bool B_KC, int ABL = 0, bool EXTRA = false>
__global__ void __launch_bounds__(64 * WGM * WGN)
    __attribute__((amdgpu_waves_per_eu(1, (NS * 2 * (BM + BN) * 128 > 80 * 1024 && WGM * WGN <= 4) ? 1 : 2)))   // (LDS already limits those to one wave per SIMD: all 512 registers are theirs)
    gemm_s16_glds(GemmArgs g, S16Operands ops) {
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  // a stage, in bytes (every plane tile is 64 halves x BR rows whichever way it lies): A_hi | A_lo | B_hi | B_lo
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = 2 * (A_BYTES + B_BYTES);
  constexpr int SLOTS_A = BM / 8, SLOTS_B = BN / 8, SLOTS = 2 * (SLOTS_A + SLOTS_B);   // 1-KiB DMA units per tile
  static_assert(SLOTS % NW == 0, "DMA units must divide over the waves");
  constexpr int G = SLOTS / NW;
  constexpr int D = NS - 1;
  constexpr int RA = A_KC ? 1 : 2, RB = B_KC ? 1 : 2;            // LDS reads per fragment and plane
  constexpr int NRH = 2 * (TM * RA + TN * RB);                    // reads per instruction k step
  constexpr int NM = KH * 3 * TM * TN, NRD = KH * NRH, SB = NM / 2 - 1;
  constexpr int UNROLL = (NS % 2 == 0) ? NS : 2 * NS;
  static_assert(G <= 2 * (SB + 1), "not enough MFMA slots before the barrier");
  static_assert(A_KC || BM == 32 || BM == 64 || BM == 128, "KS image: 32, 64 or 128 columns");
  static_assert(B_KC || BN == 32 || BN == 64 || BN == 128, "KS image: 32, 64 or 128 columns");
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  lds_char *lds3 = (lds_char *)(__attribute__((address_space(3))) void *)lds;
  const char *ldsb = reinterpret_cast<const char *>(lds);

  const bool second = g.pair && blockIdx.z == 1;   // second product of a pair (uniform)
  if (second) { g.C = g.C1; g.ep = g.ep1; }
  const S16View va = s16_pick(second, ops.a, ops.a1), vb = s16_pick(second, ops.b, ops.b1);
  int tm, tn;
  xcd_tile<BM, BN>(g, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, lh = lane >> 5;
  // split-K (as gemm_glds.hip): this workgroup reduces over K chunk blockIdx.y only and leaves a plain, unscaled-back partial product
  // in C + chunk * split_stride (the host has emptied the epilogue; splitk_reduce_kernel adds the chunks in order and applies it)
  int k_first = 0, ktiles = ops.kp / BKH;
  if (g.split_k > 1) {
    k_first = (int)blockIdx.y * g.k_chunk;                       // a multiple of the K tile
    ktiles = min(g.k_chunk, ops.kp - k_first) / BKH;
    g.C += (long)blockIdx.y * g.split_stride;
  }

  // ---- DMA descriptors: unit = 1 KiB of one plane tile.  KC: 8 rows x 128 B.  KS: 64 / (BR / 8) k rows x 2 BR bytes.
  const h16 *src[G];
  int adv[G];          // halves per K tile
  unsigned dst_off[G];
  static_for<0, G>([&](auto U_) {
    constexpr int u = decltype(U_)::value;
    const int slot = wave + u * NW;  // wave-uniform
    // plane order inside a stage: A_hi [0, SLOTS_A), A_lo, B_hi [2 SLOTS_A, ...), B_lo.  The plane tiles' unit counts are multiples of
    // the wave count, so WHICH plane unit u of a wave belongs to is a compile-time fact: choosing va / vb members under a run-time
    // condition makes hipcc select between their ADDRESSES, which parks both views in scratch memory (a private segment per launch)
    static_assert(SLOTS_A % NW == 0 && SLOTS_B % NW == 0, "a wave's DMA unit must not straddle planes");
    constexpr bool is_a = u * NW < 2 * SLOTS_A;
    constexpr int s2c = is_a ? u * NW : u * NW - 2 * SLOTS_A, per = is_a ? SLOTS_A : SLOTS_B;
    constexpr bool lo_plane = s2c >= per;
    const int sr = (lo_plane ? s2c - per : s2c) + wave;   // unit within the plane tile
    const h16 *base;
    int v_ld, v_rows;
    if constexpr (is_a) { v_ld = va.ld; v_rows = va.rows; if constexpr (lo_plane) base = va.lo; else base = va.hi; }
    else { v_ld = vb.ld; v_rows = vb.rows; if constexpr (lo_plane) base = vb.lo; else base = vb.hi; }
    const int rows_p = (v_rows + kS16Pad - 1) / kS16Pad * kS16Pad;
    auto kc_src = [&](int first_row) {
      const int r = lane >> 3;
      const int c = (lane & 7) ^ kc_swizzle(sr * 8 + r);
      int row = first_row + sr * 8 + r;
      row = row < rows_p ? row : rows_p - 1;   // (a row of the padding or of another tile: feeds outputs that are not stored)
      adv[u] = BKH;
      return base + (long)row * v_ld + 8 * c + k_first;
    };
    auto ks_src = [&](int first_col, auto BR_) {
      constexpr int BR = decltype(BR_)::value, CPR = BR / 8;   // 16-byte chunks per k row
      const int krow = sr * (64 / CPR) + lane / CPR;
      const int c = (lane % CPR) ^ (4 * ks_swizzle<BR>(krow));
      int col = first_col + 8 * c;
      col = col + 8 <= v_ld ? col : v_ld - 8;   // (columns past the planes: outputs that are not stored)
      adv[u] = BKH * v_ld;
      return base + (long)(k_first + krow) * v_ld + col;
    };
    if (is_a) {
      if constexpr (A_KC) src[u] = kc_src(m0); else src[u] = ks_src(m0, std::integral_constant<int, BM>());
    } else {
      if constexpr (B_KC) src[u] = kc_src(n0); else src[u] = ks_src(n0, std::integral_constant<int, BN>());
    }
    dst_off[u] = slot * 1024;
  });
  auto dma_unit = [&](auto U_, auto ST_, int r) {
    constexpr int u = decltype(U_)::value, st = decltype(ST_)::value;
    if constexpr (!(ABL & 2)) glds16(src[u], __builtin_amdgcn_readfirstlane(lds_base + st * STAGE + dst_off[u]));
    src[u] += (r + 1 < ktiles) ? adv[u] : 0;   // requests past the last tile fetch it again into a stage nobody reads
  };

  // ---- fragments: per-lane byte offsets inside a plane tile
  //  KC: row * 128 + swizzled 16-byte chunk (2 h + lh);  KS: lane (p = lane & 15, column half g = (lane >> 4) & 1) of a 16-lane group
  //  addresses k row 8 lh + p / 4, columns 16 g + 4 (p & 3) .. + 3 of its 32-column fragment and receives column 16 g + p, k .. k + 3
  //  (KC keeps one offset per k step: the swizzle is an XOR; KS steps are plain additions that fold into the instruction's offset field)
  constexpr int AH = A_KC ? KH : 1, BH = B_KC ? KH : 1;
  int a_off[TM][AH], b_off[TN][BH];
  const int p16 = lane & 15, g16 = (lane >> 4) & 1;
#pragma unroll
  for (int t = 0; t < TM; t++) {
    if constexpr (A_KC) {
      const int row = wm * WM + t * 32 + l31;
#pragma unroll
      for (int h = 0; h < KH; h++) a_off[t][h] = row * 128 + (((2 * h + lh) ^ kc_swizzle(row)) << 4);
    } else {
      const int T = wm * TM + t, krow = 8 * lh + (p16 >> 2);
      a_off[t][0] = krow * (2 * BM) + 64 * (T ^ ks_swizzle<BM>(krow)) + 32 * g16 + 8 * (p16 & 3);
    }
  }
#pragma unroll
  for (int t = 0; t < TN; t++) {
    if constexpr (B_KC) {
      const int col = wn * WN + t * 32 + l31;
#pragma unroll
      for (int h = 0; h < KH; h++) b_off[t][h] = 2 * A_BYTES + col * 128 + (((2 * h + lh) ^ kc_swizzle(col)) << 4);
    } else {
      const int T = wn * TN + t, krow = 8 * lh + (p16 >> 2);
      b_off[t][0] = 2 * A_BYTES + krow * (2 * BN) + 64 * (T ^ ks_swizzle<BN>(krow)) + 32 * g16 + 8 * (p16 & 3);
    }
  }
  struct Frag {
    half8 ah[KH][TM], al[KH][TM], bh[KH][TN], bl[KH][TN];
  };
  // one LDS read: flat index r -> (k step h, operand, fragment t, plane, half of the fragment)
  auto read_unit = [&](auto ST_, Frag &f, auto R_) {
    constexpr int r = decltype(R_)::value, st = decltype(ST_)::value;
    if constexpr (ABL & 4) return;
    constexpr int h = r / NRH, q = r % NRH;
    constexpr bool is_a = q < 2 * TM * RA;
    constexpr int q2 = is_a ? q : q - 2 * TM * RA, RR = is_a ? RA : RB;
    constexpr int t = q2 / (2 * RR), w = q2 % (2 * RR), lo = w / RR, half = w % RR;
    constexpr bool kc = is_a ? A_KC : B_KC;
    constexpr int plane_bytes = is_a ? A_BYTES : B_BYTES, BR = is_a ? BM : BN;
    int base = st * STAGE + (lo ? plane_bytes : 0);
    if constexpr (is_a) base += a_off[t][kc ? h : 0]; else base += b_off[t][kc ? h : 0];
    if constexpr (kc) {
      const half8 v = *reinterpret_cast<const half8 *>(ldsb + base);
      if constexpr (ABL & 1) asm volatile("" ::"v"(v));
      if constexpr (is_a) { if constexpr (lo) f.al[h][t] = v; else f.ah[h][t] = v; }
      else { if constexpr (lo) f.bl[h][t] = v; else f.bh[h][t] = v; }
    } else {
      const half4 v = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4 *)(lds3 + base + (16 * h + 4 * half) * (2 * BR))));
      if constexpr (ABL & 1) asm volatile("" ::"v"(v));
      half8 *dst = is_a ? (lo ? &f.al[h][t] : &f.ah[h][t]) : (lo ? &f.bl[h][t] : &f.bh[h][t]);
      if constexpr (half == 0) dst->lo = v; else dst->hi = v;
    }
  };

  f32x16 acc[TM][TN], accx[TM][TN];   // hi hi; hi lo' + lo' hi
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) { acc[i][j][e] = 0.0f; accx[i][j][e] = 0.0f; }
  // optional column sums of a reduction-major A operand (the bias gradient on the weight-gradient product, as gemm_glds.hip): the
  // first column of tiles' wn == 0 waves add up the A pieces they multiply anyway -- sum_k (hi + 2^-11 lo'), unscaled at the end
  const bool do_colsum = !A_KC && g.ep.colsum != nullptr && tn == 0 && wn == 0;  // wave-uniform
  float asum[TM];
#pragma unroll
  for (int i = 0; i < TM; i++) asum[i] = 0.0f;
  auto colsum_unit = [&](const Frag &f) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const half2v one = {(h16)1.0f, (h16)1.0f}, eps = {(h16)0x1p-11f, (h16)0x1p-11f};
#pragma unroll
    for (int h = 0; h < KH; h++)
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const half2v hv = {f.ah[h][i][2 * q], f.ah[h][i][2 * q + 1]}, lv = {f.al[h][i][2 * q], f.al[h][i][2 * q + 1]};
          asum[i] = __builtin_amdgcn_fdot2(hv, one, asum[i], false);     // fp32 accumulation of exact fp16 values
          asum[i] = __builtin_amdgcn_fdot2(lv, eps, asum[i], false);
        }
  };
  auto mma_unit = [&](const Frag &f, auto M_) {
    constexpr int m = decltype(M_)::value;
    if constexpr (ABL & 1) return;
    constexpr int n = m % TN, i = (m / TN) % TM, j = (m / (TN * TM)) % 3, h = m / (TN * TM * 3);
    if constexpr (j == 0) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[h][i], f.bh[h][n], acc[i][n], 0, 0, 0);
    else if constexpr (j == 1) accx[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[h][i], f.bl[h][n], accx[i][n], 0, 0, 0);
    else accx[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[h][i], f.bh[h][n], accx[i][n], 0, 0, 0);
  };

  auto step = [&](auto I_, const Frag &fcur, Frag &fnxt, int t) {
    constexpr int I = decltype(I_)::value;
    using StReq = std::integral_constant<int, (I + D) % NS>;
    using StNxt = std::integral_constant<int, (I + 1) % NS>;
    static_for<0, NM>([&](auto S_) {
      constexpr int sidx = decltype(S_)::value;
      mma_unit(fcur, S_);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!A_KC && sidx == 0) {
        if (do_colsum) colsum_unit(fcur);
      }
      if constexpr (sidx <= SB) {
        static_for<sidx * G / (SB + 1), (sidx + 1) * G / (SB + 1)>([&](auto U_) { dma_unit(U_, StReq(), t + D); });
        if constexpr (sidx == SB) {
          wait_vmcnt<(D - 1) * G>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      } else {
        constexpr int NSL = NM - SB - 1;
        static_for<(sidx - SB - 1) * NRD / NSL, (sidx - SB) * NRD / NSL>([&](auto R_) { read_unit(StNxt(), fnxt, R_); });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  Frag f0, f1;
  if constexpr (ABL & 4) {   // fragments never read: give them defined (non-constant) contents
    half8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (h16)(float)(lane + e);
    static_for<0, KH>([&](auto H_) {
      constexpr int h = decltype(H_)::value;
#pragma unroll
      for (int t = 0; t < TM; t++) { f0.ah[h][t] = z; f0.al[h][t] = z; f1.ah[h][t] = z; f1.al[h][t] = z; }
#pragma unroll
      for (int t = 0; t < TN; t++) { f0.bh[h][t] = z; f0.bl[h][t] = z; f1.bh[h][t] = z; f1.bl[h][t] = z; }
    });
  }
  static_for<0, D>([&](auto T_) {
    constexpr int t = decltype(T_)::value;
    static_for<0, G>([&](auto U_) { dma_unit(U_, T_, t); });
  });
  float w_bound = 0.f;
  if constexpr (EXTRA) w_bound = s16_weight_bound(g, va, vb, lane);   // while the first tiles are on their way
  wait_vmcnt<(D - 1) * G>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  static_for<0, NRD>([&](auto R_) { read_unit(std::integral_constant<int, 0>(), f0, R_); });

  for (int t0 = 0; t0 < ktiles; t0 += UNROLL) {
    static_for<0, UNROLL>([&](auto I_) {
      constexpr int I = decltype(I_)::value;
      if (t0 + I < ktiles) {  // wave-uniform
        if constexpr (I % 2 == 0) step(I_, f0, f1, t0 + I);
        else step(I_, f1, f0, t0 + I);
      }
    });
  }
  wait_vmcnt<0>();

  static_assert(NW * 32 * kEpiPitch * (int)sizeof(float) <= NS * STAGE, "the waves' epilogue slices must fit into the operand LDS");
  s16_finish<TM, TN, NW, EXTRA, !A_KC>(g, va, vb, acc, accx, asum, do_colsum, w_bound, m0 + wm * WM, n0 + wn * WN, lane, wave, lds);
}

// ---- both operands reduction-major (the weight gradient dW = dy^T x), 128 x 128 tile ---------------------------------------------
// The generic kernel above keeps the fragments of a whole 64-deep K tile in registers, twice: with a 128 x 128 tile on four waves that is
// 256 registers of fragments beside 128 of accumulators, and the transposing reads' addresses push it into scratch memory.  Here the
// tile's depth is 32 (two instruction steps), the fragments are double buffered per instruction step (64 registers), and the LDS holds a
// ring of FOUR such half tiles (128 KB): the DMA runs three half tiles (~2300 matrix-pipe cycles) ahead of the reads.  Against the
// 64 x 128 tile the 128 x 128 one moves a third less through L2 -> LDS per flop and issues a third fewer LDS reads per MFMA (wave tile
// 64 x 64) -- the two things gemm_s16_glds waits for (devtools/micro/s16_ablate.hip).
//   half tile h, slot h % 4:  A_hi | A_lo | B_hi | B_lo, each [32 k][128 columns] halves = 8 KB, rows as they lie in memory
//   per half tile and wave: 8 DMA units (4 k rows x 256 B each), 2 x 12 MFMAs, 2 x 16 transposing reads, one barrier
template <bool EXTRA>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gemm_s16_ks128(GemmArgs g, S16Operands ops) {   // (128 KB of LDS: one workgroup per CU whatever the register count)
  constexpr int BM = 128, BN = 128, NW = 4, TM = 2, TN = 2, KT = 32, RING = 4;
  constexpr int PLANE = KT * BM * 2, SLOT = 4 * PLANE;      // bytes: 8 KB per plane, 32 KB per half tile
  constexpr int G = (4 * PLANE / 1024) / NW;                  // 8 DMA units per wave and half tile
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  lds_char *lds3 = (lds_char *)(__attribute__((address_space(3))) void *)lds;

  const bool second = g.pair && blockIdx.z == 1;   // second product of a pair (uniform)
  if (second) { g.C = g.C1; g.ep = g.ep1; }
  const S16View va = s16_pick(second, ops.a, ops.a1), vb = s16_pick(second, ops.b, ops.b1);
  int tm, tn;
  xcd_tile<BM, BN>(g, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1, lh = lane >> 5;
  const int htiles = ops.kp / KT;

  // ---- DMA descriptors: unit = 4 k rows x 256 B of one plane; plane p = unit / 8 (A_hi, A_lo, B_hi, B_lo), 8 units per plane
  const h16 *src[G];
  int adv[G];
  static_for<0, G>([&](auto U_) {
    constexpr int u = decltype(U_)::value;
    // (8 units per plane, 4 waves: the plane of a wave's unit u is a compile-time fact -- see gemm_s16_glds)
    constexpr int plane = (u * NW) >> 3;
    const int sub = (wave + u * NW) & 7;
    constexpr bool is_a = plane < 2, lo_plane = (plane & 1) != 0;
    const h16 *base;
    int v_ld;
    if constexpr (is_a) { v_ld = va.ld; if constexpr (lo_plane) base = va.lo; else base = va.hi; }
    else { v_ld = vb.ld; if constexpr (lo_plane) base = vb.lo; else base = vb.hi; }
    const int krow = sub * 4 + (lane >> 4);
    const int c = (lane & 15) ^ (4 * (krow & 3));   // the 64-byte column groups of a k row, XOR-swizzled with the row's low bits
    int col = (is_a ? m0 : n0) + 8 * c;
    col = col + 8 <= v_ld ? col : v_ld - 8;         // (columns past the planes: outputs that are not stored)
    src[u] = base + (long)krow * v_ld + col;
    adv[u] = KT * v_ld;
  });
  auto dma_half_tile = [&](int slot, int h) {   // slot: wave-uniform ring index
    static_for<0, G>([&](auto U_) {
      constexpr int u = decltype(U_)::value;
      glds16(src[u], __builtin_amdgcn_readfirstlane(lds_base + slot * SLOT + (wave + u * NW) * 1024));
      src[u] += (h + 1 < htiles) ? adv[u] : 0;   // requests past the last half tile fetch it again into a slot nobody reads
    });
  };
  auto dma_unit = [&](auto U_, int slot, int h) {
    constexpr int u = decltype(U_)::value;
    glds16(src[u], __builtin_amdgcn_readfirstlane(lds_base + slot * SLOT + (wave + u * NW) * 1024));
    src[u] += (h + 1 < htiles) ? adv[u] : 0;
  };

  // ---- fragment addresses (bytes inside a slot): lane (p = lane & 15, column half gg) addresses k row 8 lh + p / 4 of its 16-lane group's
  // [4 k][16 columns] block and receives column 16 gg + p, k .. k + 3 (ds_read_b64_tr_b16)
  const int p16 = lane & 15, gg = (lane >> 4) & 1, krow_l = 8 * lh + (p16 >> 2);
  int a_off[TM], b_off[TN];
#pragma unroll
  for (int t = 0; t < TM; t++) a_off[t] = krow_l * 256 + 64 * ((wm * TM + t) ^ (krow_l & 3)) + 32 * gg + 8 * (p16 & 3);
#pragma unroll
  for (int t = 0; t < TN; t++) b_off[t] = 2 * PLANE + krow_l * 256 + 64 * ((wn * TN + t) ^ (krow_l & 3)) + 32 * gg + 8 * (p16 & 3);
  struct Frag { half8 ah[TM], al[TM], bh[TN], bl[TN]; };   // one instruction step
  // read r of the 16 of an instruction step: operand, fragment, plane, half of the fragment
  auto read_unit = [&](int slot_base, auto KS_, Frag &f, auto R_) {
    constexpr int r = decltype(R_)::value, ks = decltype(KS_)::value;
    constexpr bool is_a = r < 8;
    constexpr int q = r & 7, t = q >> 2, lo = (q >> 1) & 1, half = q & 1;
    const int off = slot_base + (is_a ? a_off[t] : b_off[t]) + (lo ? PLANE : 0) + (16 * ks + 4 * half) * 256;
    const half4 v = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4 *)(lds3 + off)));
    half8 *dst = is_a ? (lo ? &f.al[t] : &f.ah[t]) : (lo ? &f.bl[t] : &f.bh[t]);
    if constexpr (half == 0) dst->lo = v; else dst->hi = v;
  };

  f32x16 acc[TM][TN], accx[TM][TN];   // hi hi; hi lo' + lo' hi
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) { acc[i][j][e] = 0.0f; accx[i][j][e] = 0.0f; }
  const bool do_colsum = g.ep.colsum != nullptr && tn == 0 && wn == 0;  // wave-uniform
  float asum[TM] = {0.f, 0.f};
  // MFMA m of the 12 of an instruction step: the four main products, then the four hi lo', then the four lo' hi (a cross accumulator is
  // met again four instructions later)
  auto mma_unit = [&](const Frag &f, auto M_) {
    constexpr int m = decltype(M_)::value, j = m >> 2, i = (m >> 1) & 1, n = m & 1;
    if constexpr (j == 0) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[n], acc[i][n], 0, 0, 0);
    else if constexpr (j == 1) accx[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[n], accx[i][n], 0, 0, 0);
    else accx[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[n], accx[i][n], 0, 0, 0);
  };

  // prologue: three half tiles on their way, the first one landed, its first step's fragments read
  Frag f0, f1;
  dma_half_tile(0, 0);
  dma_half_tile(1, 1);
  dma_half_tile(2, 2);
  float w_bound = 0.f;
  if constexpr (EXTRA) w_bound = s16_weight_bound(g, va, vb, lane);
  wait_vmcnt<2 * G>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  static_for<0, 16>([&](auto R_) { read_unit(0, std::integral_constant<int, 0>(), f0, R_); });

  for (int h = 0; h < htiles; h++) {
    const int slot = h & (RING - 1), slot_base = slot * SLOT;
    const int slot_req = (h + 3) & (RING - 1), slot_nxt = ((h + 1) & (RING - 1)) * SLOT;
    // step 0 of half tile h from f0: request half tile h + 3 (the slot of h - 1: every wave has passed the barrier behind its last
    // read of it), read step 1's fragments into f1
    static_for<0, 12>([&](auto S_) {
      constexpr int sidx = decltype(S_)::value;
      mma_unit(f0, S_);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (sidx == 0) {
        if (do_colsum) {
#pragma unroll
          for (int i = 0; i < TM; i++) asum[i] = s16_frag_sum(asum[i], f0.ah[i], f0.al[i]);
        }
      }
      if constexpr (sidx < G) dma_unit(S_, slot_req, h + 3);
      static_for<sidx * 16 / 12, (sidx + 1) * 16 / 12>([&](auto R_) { read_unit(slot_base, std::integral_constant<int, 1>(), f1, R_); });
      __builtin_amdgcn_sched_barrier(0);
    });
    // step 1 from f1: this wave's share of half tile h + 1 has landed (h + 2, h + 3 stay in flight), one barrier publishes it, then
    // its first step's fragments go into f0
    static_for<0, 12>([&](auto S_) {
      constexpr int sidx = decltype(S_)::value;
      mma_unit(f1, S_);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (sidx == 0) {
        if (do_colsum) {
#pragma unroll
          for (int i = 0; i < TM; i++) asum[i] = s16_frag_sum(asum[i], f1.ah[i], f1.al[i]);
        }
      }
      if constexpr (sidx == 3) {
        wait_vmcnt<2 * G>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if constexpr (sidx >= 4) {
        static_for<(sidx - 4) * 16 / 8, (sidx - 3) * 16 / 8>([&](auto R_) { read_unit(slot_nxt, std::integral_constant<int, 0>(), f0, R_); });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  wait_vmcnt<0>();
  static_assert(NW * 32 * kEpiPitch * (int)sizeof(float) <= RING * SLOT, "the waves' epilogue slices must fit into the operand LDS");
  s16_finish<TM, TN, NW, EXTRA, true>(g, va, vb, acc, accx, asum, do_colsum, w_bound, m0 + wm * 64, n0 + wn * 64, lane, wave, lds);
}

// One scalar load whose result is never read (it pulls a line into the L2 over the scalar cache's path).  The result lands in s100 whenever
// the line arrives; the kernels that use this are checked to need far fewer scalar registers than that (kernel-resource-usage), so nothing
// of the compiler's ever lives there.  (An output operand would be a register the compiler re-uses right behind the statement.)
__device__ __forceinline__ void s16_scalar_touch(unsigned long long addr) {
  asm volatile("s_load_dword s100, %0, 0x0" ::"s"(addr) : "s100");
}

// ---- producer / consumer waves (round 6) -------------------------------------------------------------------------------------------
// The kernels above run one wave per SIMD that does everything: it issues the K tile's LDS-DMA requests, the fragment reads and the
// matrix instructions from ONE in-order instruction stream.  The counters (profiles/r06_gemm_split16_pmc_*.txt) say what that costs:
// 40 % of the wave cycles are instruction-issue stalls (SQ_WAIT_INST_ANY) while the LDS is ~25 % busy with no bank conflicts and the
// matrix pipe ~45 % busy -- a global_load_lds that waits for the texture path to take it (one 1-KiB request per ~70 cycles and wave at the
// path's ~56 B/clk/CU) holds up the matrix instructions and reads behind it, and nothing else can issue on that SIMD.  Here a workgroup
// is EIGHT waves, two per SIMD: waves 0-3 (consumers) only read fragments and multiply, waves 4-7 (producers) only issue the LDS-DMA
// requests and wait for them; a producer parked on the texture path costs its SIMD nothing, the consumer beside it keeps issuing.  One
// workgroup barrier per K tile connects the two roles exactly as before (counted vmcnt on the producer side, then the barrier publishes
// the tile and frees the stage the consumers have just left).  Producers end behind the K loop; the epilogue's barriers then count the
// surviving consumer waves only.  Same instruction order per accumulator as the kernels above: results are bit-identical to theirs.
//   KT = 64 halves per K tile (KC operands need 128-byte rows for their LDS image) or 32 when both operands are reduction-major
//   fragments double-buffered per INSTRUCTION step (registers: at most 256 per wave with two waves per SIMD)
// ABL (devtools only): 1 = no MFMA, 2 = no DMA, 4 = no LDS reads -- wrong results, for timing
template <int BM, int BN, int KT, int NS, bool A_KC, bool B_KC, bool EXTRA, int ABL = 0, int OPT = 0>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm_s16_pc(GemmArgs g, S16Operands ops) {
  constexpr int NC = 4, NP = 4;   // consumer waves (2 x 2 wave tiles), producer waves
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int KHS = KT / 16;    // instruction k steps per K tile
  static_assert(KT == 64 || (KT == 32 && !A_KC && !B_KC), "an operand whose reduction index is contiguous needs 128-byte rows in its LDS image");
  static_assert(KHS % 2 == 0, "the two fragment sets alternate per step and every tile starts on the first");
  constexpr int A_BYTES = BM * KT * 2, B_BYTES = BN * KT * 2, STAGE = 2 * (A_BYTES + B_BYTES);   // A_hi | A_lo | B_hi | B_lo
  constexpr int SLOTS_A = A_BYTES / 1024, SLOTS_B = B_BYTES / 1024, SLOTS = 2 * (SLOTS_A + SLOTS_B);   // 1-KiB DMA units per plane tile
  static_assert(SLOTS_A % NP == 0 && SLOTS_B % NP == 0, "a producer's DMA unit must not straddle planes");
  constexpr int G = SLOTS / NP;
  constexpr int RA = A_KC ? 1 : 2, RB = B_KC ? 1 : 2;   // LDS reads per fragment and plane
  constexpr int NRH = 2 * (TM * RA + TN * RB);            // reads per instruction step
  constexpr int NMS = 3 * TM * TN;                        // matrix instructions per instruction step
  constexpr int BAR_AT = NMS >= 12 ? 3 : 0;               // behind which instruction of a tile's last step the barrier sits
  static_assert(!A_KC || true, "");
  static_assert(A_KC || BM == 64 || BM == 128, "KS image: 64 or 128 columns");
  static_assert(B_KC || BN == 64 || BN == 128, "KS image: 64 or 128 columns");
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  lds_char *lds3 = (lds_char *)(__attribute__((address_space(3))) void *)lds;
  const char *ldsb = reinterpret_cast<const char *>(lds);

  const bool second = g.pair && blockIdx.z == 1;   // second product of a pair (uniform)
  if (second) { g.C = g.C1; g.ep = g.ep1; }
  const S16View va = s16_pick(second, ops.a, ops.a1), vb = s16_pick(second, ops.b, ops.b1);
  int tm, tn;
  xcd_tile<BM, BN>(g, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int k_first = 0, ktiles = ops.kp / KT;
  if (g.split_k > 1) {   // (as gemm_s16_glds: this workgroup reduces over K chunk blockIdx.y and leaves a plain partial product)
    k_first = (int)blockIdx.y * g.k_chunk;
    ktiles = min(g.k_chunk, ops.kp - k_first) / KT;
    g.C += (long)blockIdx.y * g.split_stride;
  }

  if (wave >= NC) {
    // ================================================= producer =================================================
    const int pw = wave - NC;
    if constexpr (OPT & 2) __builtin_amdgcn_s_setprio(1);
    const h16 *src[G];
    int adv[G];
    static_for<0, G>([&](auto U_) {
      constexpr int u = decltype(U_)::value;
      constexpr bool is_a = u * NP < 2 * SLOTS_A;
      constexpr int s2c = is_a ? u * NP : u * NP - 2 * SLOTS_A, per = is_a ? SLOTS_A : SLOTS_B;
      constexpr bool lo_plane = s2c >= per;
      const int sr = (lo_plane ? s2c - per : s2c) + pw;   // unit within the plane tile
      const h16 *base;
      int v_ld, v_rows;
      if constexpr (is_a) { v_ld = va.ld; v_rows = va.rows; if constexpr (lo_plane) base = va.lo; else base = va.hi; }
      else { v_ld = vb.ld; v_rows = vb.rows; if constexpr (lo_plane) base = vb.lo; else base = vb.hi; }
      constexpr bool kc = is_a ? A_KC : B_KC;
      constexpr int BR = is_a ? BM : BN;
      const int first = is_a ? m0 : n0;
      if constexpr (kc) {   // 8 rows x 128 B per unit, 16-byte chunks XOR-swizzled on the source side
        const int rows_p = (v_rows + kS16Pad - 1) / kS16Pad * kS16Pad;
        const int r = lane >> 3, c = (lane & 7) ^ kc_swizzle(sr * 8 + r);
        int row = first + sr * 8 + r;
        row = row < rows_p ? row : rows_p - 1;   // (a row of the padding or of another tile: feeds outputs that are not stored)
        adv[u] = KT;
        src[u] = base + (long)row * v_ld + 8 * c + k_first;
      } else {              // 1024 / (2 BR) k rows x 2 BR bytes per unit
        constexpr int CPR = BR / 8;
        const int krow = sr * (64 / CPR) + lane / CPR;
        const int c = (lane % CPR) ^ (4 * ks_swizzle<BR>(krow));
        int col = first + 8 * c;
        col = col + 8 <= v_ld ? col : v_ld - 8;   // (columns past the planes: outputs that are not stored)
        adv[u] = KT * v_ld;
        src[u] = base + (long)(k_first + krow) * v_ld + col;
      }
    });
    // (OPT & 8 / 16, experiment) scalar loads that pull the 128-byte lines of a LATER K tile into this XCD's L2 -- over the scalar cache's
    // path, not the texture path the LDS-DMA requests are bound by; their results are never read
    constexpr int PD = 2;   // tiles ahead of the tile being requested
    auto pf_tile = [&](int r) {
      if constexpr (!(OPT & 24)) return;
      if (r >= ktiles) return;
      static_for<0, G>([&](auto U_) {
        constexpr int u = decltype(U_)::value;
        constexpr bool is_a = u * NP < 2 * SLOTS_A;
        constexpr int s2c = is_a ? u * NP : u * NP - 2 * SLOTS_A, per = is_a ? SLOTS_A : SLOTS_B;
        constexpr bool lo_plane = s2c >= per;
        const int sr = (lo_plane ? s2c - per : s2c) + pw;
        const h16 *base;
        int v_ld, v_rows;
        if constexpr (is_a) { v_ld = va.ld; v_rows = va.rows; if constexpr (lo_plane) base = va.lo; else base = va.hi; }
        else { v_ld = vb.ld; v_rows = vb.rows; if constexpr (lo_plane) base = vb.lo; else base = vb.hi; }
        constexpr bool kc = is_a ? A_KC : B_KC;
        constexpr int BR = is_a ? BM : BN;
        const int first = is_a ? m0 : n0;
        const int share = is_a ? (tn & 3) : (tm & 7);   // (OPT & 16) which of the lines the tiles of this XCD's block share is this workgroup's to fetch
        static_for<0, 8>([&](auto L_) {
          constexpr int l = decltype(L_)::value;
          if constexpr ((OPT & 16) != 0) { if ((l & (is_a ? 3 : 7)) != share) return; }
          const h16 *p;
          if constexpr (kc) {
            const int rows_p = (v_rows + kS16Pad - 1) / kS16Pad * kS16Pad;
            int row = first + sr * 8 + l;
            row = row < rows_p ? row : rows_p - 1;
            p = base + (long)row * v_ld + k_first + (long)r * KT;
          } else {
            constexpr int LPR = BR / 64;   // lines per k row
            const int krow = sr * (8 / LPR) + l / LPR;
            int col = first + 64 * (l % LPR);
            col = col + 64 <= v_ld ? col : v_ld - 64;
            p = base + ((long)k_first + (long)r * KT + krow) * v_ld + col;
          }
          const unsigned long long a = reinterpret_cast<unsigned long long>(p);
          const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)a), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
          const unsigned long long au = ((unsigned long long)hi32 << 32) | lo32;
          s16_scalar_touch(au);
        });
      });
    };
    auto dma_tile = [&](auto ST_, int r) {   // K tile r into stage ST
      constexpr int st = decltype(ST_)::value;
      static_for<0, G>([&](auto U_) {
        constexpr int u = decltype(U_)::value;
        if constexpr (!(ABL & 2)) glds16(src[u], __builtin_amdgcn_readfirstlane(lds_base + st * STAGE + (pw + u * NP) * 1024));
        if constexpr ((OPT & 4) != 0) src[u] += (r + 1 < ktiles) ? (((r + 1) & 3) == 0 ? -3 * adv[u] : adv[u]) : 0;   // (experiment: every request an L2 hit; wrong results)
        else src[u] += (r + 1 < ktiles) ? adv[u] : 0;   // requests past the last tile fetch it again into a stage nobody reads any more
      });
      pf_tile(r + PD);
    };
    static_for<0, NS - 1 + PD>([&](auto T_) { if constexpr (decltype(T_)::value >= 1) pf_tile(decltype(T_)::value); });
    static_for<0, NS - 1>([&](auto T_) { dma_tile(T_, decltype(T_)::value); });
    wait_vmcnt<G *(NS - 2)>();   // tile 0 has landed
    __builtin_amdgcn_s_barrier();
    for (int t0 = 0; t0 < ktiles; t0 += NS) {
      static_for<0, NS>([&](auto I_) {
        constexpr int I = decltype(I_)::value;
        if (t0 + I < ktiles) {   // uniform
          // the stage of tile t - 1: every consumer passed the barrier of tile t - 1 behind its last read of it
          dma_tile(std::integral_constant<int, (I + NS - 1) % NS>(), t0 + I + NS - 1);
          wait_vmcnt<G *(NS - 2)>();   // this wave's share of tile t + 1 has landed
          __builtin_amdgcn_s_barrier();
        }
      });
    }
    wait_vmcnt<0>();                 // (the surplus requests write into the LDS the epilogue is about to use)
    if constexpr ((OPT & 24) != 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    return;
  }

  // =================================================== consumer ===================================================
  if constexpr (OPT & 1) __builtin_amdgcn_s_setprio(1);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
  constexpr int AH = A_KC ? KHS : 1, BH = B_KC ? KHS : 1;
  int a_off[TM][AH], b_off[TN][BH];
  const int p16 = lane & 15, g16 = (lane >> 4) & 1;
#pragma unroll
  for (int t = 0; t < TM; t++) {
    if constexpr (A_KC) {
      const int row = wm * WM + t * 32 + l31;
#pragma unroll
      for (int h = 0; h < KHS; h++) a_off[t][h] = row * 128 + (((2 * h + lh) ^ kc_swizzle(row)) << 4);
    } else {
      const int T = wm * TM + t, krow = 8 * lh + (p16 >> 2);
      a_off[t][0] = krow * (2 * BM) + 64 * (T ^ ks_swizzle<BM>(krow)) + 32 * g16 + 8 * (p16 & 3);
    }
  }
#pragma unroll
  for (int t = 0; t < TN; t++) {
    if constexpr (B_KC) {
      const int col = wn * WN + t * 32 + l31;
#pragma unroll
      for (int h = 0; h < KHS; h++) b_off[t][h] = 2 * A_BYTES + col * 128 + (((2 * h + lh) ^ kc_swizzle(col)) << 4);
    } else {
      const int T = wn * TN + t, krow = 8 * lh + (p16 >> 2);
      b_off[t][0] = 2 * A_BYTES + krow * (2 * BN) + 64 * (T ^ ks_swizzle<BN>(krow)) + 32 * g16 + 8 * (p16 & 3);
    }
  }
  struct Frag { half8 ah[TM], al[TM], bh[TN], bl[TN]; };   // one instruction step
  // LDS read r of instruction step h of the tile in stage st: operand, fragment t, plane, half of the fragment
  auto read_unit = [&](auto ST_, auto H_, Frag &f, auto R_) {
    constexpr int r = decltype(R_)::value, st = decltype(ST_)::value, h = decltype(H_)::value;
    if constexpr (ABL & 4) return;
    constexpr bool is_a = r < 2 * TM * RA;
    constexpr int q2 = is_a ? r : r - 2 * TM * RA, RR = is_a ? RA : RB;
    constexpr int t = q2 / (2 * RR), w = q2 % (2 * RR), lo = w / RR, half = w % RR;
    constexpr bool kc = is_a ? A_KC : B_KC;
    constexpr int plane_bytes = is_a ? A_BYTES : B_BYTES, BR = is_a ? BM : BN;
    int base = st * STAGE + (lo ? plane_bytes : 0);
    if constexpr (is_a) base += a_off[t][kc ? h : 0]; else base += b_off[t][kc ? h : 0];
    if constexpr (kc) {
      const half8 v = *reinterpret_cast<const half8 *>(ldsb + base);
      if constexpr (is_a) { if constexpr (lo) f.al[t] = v; else f.ah[t] = v; }
      else { if constexpr (lo) f.bl[t] = v; else f.bh[t] = v; }
    } else {
      const half4 v = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4 *)(lds3 + base + (16 * h + 4 * half) * (2 * BR))));
      half8 *dst = is_a ? (lo ? &f.al[t] : &f.ah[t]) : (lo ? &f.bl[t] : &f.bh[t]);
      if constexpr (half == 0) dst->lo = v; else dst->hi = v;
    }
  };

  f32x16 acc[TM][TN], accx[TM][TN];   // hi hi; hi lo' + lo' hi
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) { acc[i][j][e] = 0.0f; accx[i][j][e] = 0.0f; }
  const bool do_colsum = !A_KC && g.ep.colsum != nullptr && tn == 0 && wn == 0;  // wave-uniform
  float asum[TM];
#pragma unroll
  for (int i = 0; i < TM; i++) asum[i] = 0.0f;
  // matrix instruction m of a step: all main products, then all hi lo', then all lo' hi (an accumulator is met again TM TN instructions later)
  auto mma_unit = [&](const Frag &f, auto M_) {
    constexpr int m = decltype(M_)::value;
    constexpr int n = m % TN, i = (m / TN) % TM, j = m / (TN * TM);
    if constexpr (ABL & 1) {   // the instruction's operands are waited for where it would issue, nothing more
      if constexpr (j == 0) asm volatile("" ::"v"(f.ah[i]), "v"(f.bh[n]));
      else if constexpr (j == 1) asm volatile("" ::"v"(f.bl[n]));
      else asm volatile("" ::"v"(f.al[i]));
      return;
    }
    if constexpr (j == 0) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[n], acc[i][n], 0, 0, 0);
    else if constexpr (j == 1) accx[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[n], accx[i][n], 0, 0, 0);
    else accx[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[n], accx[i][n], 0, 0, 0);
  };
  // instruction step h of the tile in stage st from fc; the next step's fragments go into fn -- behind the tile's barrier when they
  // belong to the next tile
  auto cstep = [&](auto ST_, auto H_, const Frag &fc, Frag &fn) {
    constexpr int st = decltype(ST_)::value, h = decltype(H_)::value;
    constexpr bool last = h == KHS - 1;
    using StN = std::integral_constant<int, last ? (st + 1) % NS : st>;
    using HN = std::integral_constant<int, last ? 0 : h + 1>;
    constexpr int first = last ? BAR_AT : 0, nslots = NMS - first;
    static_for<0, NMS>([&](auto S_) {
      constexpr int s = decltype(S_)::value;
      mma_unit(fc, S_);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!A_KC && s == 0) {
        if (do_colsum) {
#pragma unroll
          for (int i = 0; i < TM; i++) asum[i] = s16_frag_sum(asum[i], fc.ah[i], fc.al[i]);
        }
      }
      if constexpr (last && s == BAR_AT) {
        __builtin_amdgcn_s_barrier();   // the next tile has landed; the producers may refill this one's stage
        asm volatile("" ::: "memory");
      }
      if constexpr (s >= first) {
        static_for<(s - first) * NRH / nslots, (s - first + 1) * NRH / nslots>([&](auto R_) { read_unit(StN(), HN(), fn, R_); });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  Frag f0, f1;
  if constexpr (ABL & 4) {   // fragments never read: give them defined (non-constant) contents
    half8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (h16)(float)(lane + e);
#pragma unroll
    for (int t = 0; t < TM; t++) { f0.ah[t] = z; f0.al[t] = z; f1.ah[t] = z; f1.al[t] = z; }
#pragma unroll
    for (int t = 0; t < TN; t++) { f0.bh[t] = z; f0.bl[t] = z; f1.bh[t] = z; f1.bl[t] = z; }
  }
  float w_bound = 0.f;
  if constexpr (EXTRA) w_bound = s16_weight_bound(g, va, vb, lane);   // while the first tiles are on their way
  __builtin_amdgcn_s_barrier();   // tile 0 has landed
  asm volatile("" ::: "memory");
  static_for<0, NRH>([&](auto R_) { read_unit(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), f0, R_); });
  for (int t0 = 0; t0 < ktiles; t0 += NS) {
    static_for<0, NS>([&](auto I_) {
      constexpr int I = decltype(I_)::value;
      if (t0 + I < ktiles) {  // wave-uniform
        static_for<0, KHS>([&](auto H_) {
          if constexpr (decltype(H_)::value % 2 == 0) cstep(I_, H_, f0, f1);
          else cstep(I_, H_, f1, f0);
        });
      }
    });
  }
  __builtin_amdgcn_s_barrier();   // the producers' last requests have landed: the operand LDS is the epilogue's now
  asm volatile("" ::: "memory");
  static_assert(NC * 32 * kEpiPitch * (int)sizeof(float) <= NS * STAGE, "the waves' epilogue slices must fit into the operand LDS");
  s16_finish<TM, TN, NC, EXTRA, !A_KC>(g, va, vb, acc, accx, asum, do_colsum, w_bound, m0 + wm * WM, n0 + wn * WN, lane, wave, lds);
}

template <int BM, int BN, int KT, int NS, bool A_KC, bool B_KC, bool EXTRA, int ABL = 0, int OPT = 0>
void launch_s16_pc(GemmArgs &g, const S16Operands &ops) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  constexpr int lds_bytes = NS * 4 * (BM + BN) * KT;
  auto kern = gemm_s16_pc<BM, BN, KT, NS, A_KC, B_KC, EXTRA, ABL, OPT>;
  static bool attr_set = false;
  if (!attr_set) {
    ASLP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.split_k > 1 ? g.split_k : 1, g.pair ? 2 : 1), dim3(512), lds_bytes, cur_stream(), g, ops);
  t_last_parts = EXTRA ? g.tiles_m * g.tiles_n * (g.pair ? 2 : 1) : 0;
}

template <bool EXTRA>
void launch_s16_ks128(GemmArgs &g, const S16Operands &ops) {
  g.tiles_m = (g.M + 127) / 128;
  g.tiles_n = (g.N + 127) / 128;
  constexpr int lds_bytes = 4 * 4 * 32 * 128 * 2;   // 128 KB
  auto kern = gemm_s16_ks128<EXTRA>;
  static bool attr_set = false;
  if (!attr_set) {
    ASLP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, 1, g.pair ? 2 : 1), dim3(256), lds_bytes, cur_stream(), g, ops);
  t_last_parts = EXTRA ? g.tiles_m * g.tiles_n * (g.pair ? 2 : 1) : 0;
}

template <int BM, int BN, int WGM, int WGN, int NS, bool A_KC, bool B_KC, int ABL = 0, bool EXTRA = false>
void launch_s16(GemmArgs &g, const S16Operands &ops) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  constexpr int lds_bytes = NS * 2 * (BM + BN) * 128;
  auto kern = gemm_s16_glds<BM, BN, WGM, WGN, NS, A_KC, B_KC, ABL, EXTRA>;
  static bool attr_set = false;
  if (!attr_set) {
    if (lds_bytes > 48 * 1024)
      ASLP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.split_k > 1 ? g.split_k : 1, g.pair ? 2 : 1), dim3(64 * WGM * WGN), lds_bytes, cur_stream(), g, ops);
  t_last_parts = EXTRA ? g.tiles_m * g.tiles_n * (g.pair ? 2 : 1) : 0;
}
template <bool A_KC, bool B_KC>
bool launch_s16_layout(GemmArgs &g, const S16Operands &ops, int cfg) {
  // 128 x 128 where that still gives every CU a workgroup, else 64 x 128 (measured: 1024 x 2048 x 2048 52 against 70 us per call).  Only with
  // both operands reduction-contiguous: the transposing reads' address registers push the 128 x 128 tile past 512 registers (27-31 spilled),
  // and a kernel with a private segment pays ~1 ms per launch for it on this runtime.
  const bool extra = g.ep.planes_of != 0 || g.ep.wmax_parts != nullptr || g.ep.cmax_parts != nullptr || (g.pair && (g.ep1.planes_of != 0 || g.ep1.wmax_parts || g.ep1.cmax_parts));
  if constexpr (!A_KC && !B_KC) {
    // both operands reduction-major: the 128 x 128 kernel wherever its grid fills the chip about as well as the 64 x 128 one's --
    // rounds of 256 workgroups, a 128 x 128 round costing ~1.6 of a 64 x 128 one (measured on 2048 x 2048 x 1024)
    static const int ks128 = [] { const char *e = getenv("ASLP_GEMM_KS128"); return e ? atoi(e) : 1; }();   // A/B switch
    const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.pair ? 2 : 1), t64 = (long)((g.M + 63) / 64) * ((g.N + 127) / 128) * (g.pair ? 2 : 1);
    static const double round_cost = [] { const char *e = getenv("ASLP_GEMM_KS128_COST"); return e ? atof(e) : 1.6; }();   // (tuning aid)
    const double cost128 = round_cost * (double)((t128 + 255) / 256), cost64 = (double)((t64 + 255) / 256);
    if (ks128 && (cfg == 0 || cfg == 328) && (cfg == 328 || (t128 >= 200 && cost128 <= cost64)) && (g.N % 8) == 0 && (g.M % 8) == 0) {
      if (extra) launch_s16_ks128<true>(g, ops); else launch_s16_ks128<false>(g, ops);
      t_last_cfg_s16 = 328;
      return true;
    }
  }
  // 128 x 128 (cfg 311; both operands reduction-contiguous, nothing extra to leave) where its rounds of 256 workgroups cost no more than the
  // 64 x 128 tile's, a 128 x 128 round counted as two: level at 2048^3 ... 8192 x 2048 x 2048 (57.8 / 222.3 against 57.4 / 220.8 us from
  // prepared planes), 1 % ahead at 4096^3, and inside the LC-BLSTM step, whose layer products come as pairs of 1920 x 2048 (480 against 960
  // workgroups), worth 2.87 against 2.93-3.05 ms; not where it leaves a ragged last round (1920 x 3000 x 1024: 58.1 against 48.5 us)
  if (cfg != 304 && cfg != 305 && (cfg == 0 || !(A_KC && B_KC) || extra)) {   // (311 / 312 / 351 asked for by number stand when the product is KC / KC without extras)
    const long t128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.pair ? 2 : 1), t64 = (long)((g.M + 63) / 64) * ((g.N + 127) / 128) * (g.pair ? 2 : 1);
    static const int any128 = [] { const char *e = getenv("ASLP_GEMM_S16_128_ANY"); return e ? atoi(e) : 0; }();   // (tuning aid: 1 = every grid of >= 224 tiles, 2 = never)
    cfg = (!extra && A_KC && B_KC && t128 >= 224 && any128 != 2 && (any128 == 1 || 2 * ((t128 + 255) / 256) <= (t64 + 255) / 256)) ? 311 : 308;
  }
  // (Below 224 tiles -- cfg2's output layer, 1024 x 3000 x 2048: 192 workgroups of 128 x 128 in one round against 384 of 64 x 128 in two -- the
  // producer / consumer kernel wins only from operands that are hot in the L2: 66.5 against 74.2 us per call in a loop over one product,
  // 59.6 against 57.8 us inside the training step, where the weights' planes come from HBM and two stages hide less of that than three;
  // cfg2 0.756 against 0.754 ms per step, three alternations.  The floor stays.)
  // the producer / consumer kernel (gemm_s16_pc) where the 128 x 128 tile was chosen: 4096^3 383 against 422 us, same bits (devtools/micro/s16_pc.hip,
  // profiles/r06_gemm_s16_pc_micro.txt); ASLP_GEMM_S16_PC=0 keeps the one-role kernel (A/B switch)
  static const int pc_on = [] { const char *e = getenv("ASLP_GEMM_S16_PC"); return e ? atoi(e) : 1; }();
  if (cfg == 311 && pc_on && A_KC && B_KC && g.split_k <= 1) cfg = 351;
  switch (cfg) {
    case 311: if constexpr (A_KC && B_KC) launch_s16<128, 128, 2, 2, 2, true, true>(g, ops); break;
    case 312: if constexpr (A_KC && B_KC) launch_s16<128, 128, 2, 2, 2, true, true>(g, ops); cfg = 311; break;   // (forced: the one-role kernel whatever the switch says)
    case 351: if constexpr (A_KC && B_KC) launch_s16_pc<128, 128, 64, 2, true, true, false>(g, ops); break;
    case 304:   // 32 x 64, two waves: 256 workgroups for a 256 x 2048 output with the whole reduction in one launch
      if (extra) launch_s16<32, 64, 1, 2, 3, A_KC, B_KC, 0, true>(g, ops);
      else launch_s16<32, 64, 1, 2, 3, A_KC, B_KC>(g, ops);
      break;
    case 305:   // (tuning aid: the same with four stages)
      if constexpr (A_KC) { if (!extra) { launch_s16<32, 64, 1, 2, 4, A_KC, B_KC>(g, ops); break; } }
      return false;
    case 308:
      if (extra) launch_s16<64, 128, 2, 2, 3, A_KC, B_KC, 0, true>(g, ops);
      else launch_s16<64, 128, 2, 2, 3, A_KC, B_KC>(g, ops);
      break;
    default: return false;
  }
  t_last_cfg_s16 = cfg;
  return true;
}

int g_split16_override = -1;   // aslp_gemm_split16(): -1 = the environment decides
int g_split16_tile_override = -1;   // aslp_gemm_split16_tile(): -1 = ASLP_GEMM_SPLIT_F16_TILE / the heuristic

}  // namespace

bool gemm_split16_enabled() {
  static const bool on = !(getenv("ASLP_GEMM_SPLIT_F16") != nullptr && getenv("ASLP_GEMM_SPLIT_F16")[0] == '0');
  return g_split16_override >= 0 ? g_split16_override != 0 : on;
}

int s16_plane_ld(int cols) {
  // (an extra pitch of 64 or 128 halves on rows whose pitch is a multiple of 2 KB was measured: no effect on the layer products, 31.5 /
  //  31.6 / 31.8 us)
  return (cols + kS16Pad - 1) / kS16Pad * kS16Pad;
}
bool gemm_split16_serves(int M, int N, int K) {
  return gemm_split16_enabled() && M >= 128 && N >= 128 && K >= 64 && !((M | N | K) & 3);
}
static int g_keep_override = -1;
bool s16_keep_weight_planes() {
  static const bool on = !(getenv("ASLP_KEEP_WEIGHT_PLANES") != nullptr && getenv("ASLP_KEEP_WEIGHT_PLANES")[0] == '0');
  return g_keep_override >= 0 ? g_keep_override != 0 : on;
}
static std::atomic<long> g_param_epoch{1};
long s16_param_epoch() { return g_param_epoch.load(std::memory_order_relaxed); }
long s16_new_epoch() {
  static std::atomic<long> counter{0};
  return ++counter;
}
S16DiffTarget &s16_loss_diff_target() {
  static thread_local S16DiffTarget t;
  return t;
}
S16Epochs &s16_epochs() {
  static thread_local S16Epochs e;
  return e;
}

// ---- PlaneSet ----------------------------------------------------------------------------------------------------------------------
PlaneSet::~PlaneSet() {
  // (process teardown may already have unloaded the HIP runtime: errors are ignored)
  if (hi_) (void)hipFree(hi_);
  if (slot_) (void)hipFree(slot_);
}
bool PlaneSet::ReserveParts() {
  if (!slot_) {
    void *p = nullptr;
    if (hipMalloc(&p, 256 + sizeof(float) * kS16MaxParts) != hipSuccess) { set_error("PlaneSet: hipMalloc failed"); return false; }
    slot_ = static_cast<unsigned *>(p);
    parts_ = reinterpret_cast<float *>(static_cast<char *>(p) + 256);
    (void)hipMemsetAsync(p, 0, 256 + sizeof(float) * kS16MaxParts, cur_stream());
    host_bound_ = -1.f;
  }
  return true;
}
bool PlaneSet::Reserve(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return false;
  const int ld = s16_plane_ld(cols), rows_p = (rows + kS16Pad - 1) / kS16Pad * kS16Pad;
  const size_t need = (size_t)rows_p * ld;
  if (!ReserveParts()) return false;
  const bool reshape = rows != rows_ || cols != cols_;
  if (need > cap_) {
    if (hi_) {
      (void)hipStreamSynchronize(cur_stream());   // outstanding products may still read the old planes
      (void)hipFree(hi_);
      hi_ = lo_ = nullptr;
      cap_ = 0;
    }
    void *p = nullptr;
    if (hipMalloc(&p, 2 * need * sizeof(h16)) != hipSuccess) { set_error("PlaneSet: hipMalloc failed"); return false; }
    hi_ = static_cast<h16 *>(p);
    cap_ = need;
  }
  if (reshape) {
    lo_ = hi_ + need;
    rows_ = rows; cols_ = cols; ld_ = ld; rows_p_ = rows_p;
    (void)hipMemsetAsync(hi_, 0, 2 * need * sizeof(h16), cur_stream());   // the padding stays zero from here on
    Invalidate();
  }
  return true;
}
bool PlaneSet::SetBound(float bound) {
  if (!slot_) return false;
  if (bound != host_bound_) {
    (void)hipMemcpyAsync(slot_, &bound, sizeof(float), hipMemcpyHostToDevice, cur_stream());   // (pageable source: copied before the call returns)
    host_bound_ = bound;
  }
  return true;
}
bool PlaneSet::ConvertWithParts(const float *src, int rows, int cols, int stride, int nparts) {
  if ((cols & 3) || (stride & 3) || !aligned16(src) || nparts > kS16MaxParts) return false;
  if (!Reserve(rows, cols)) return false;
  host_bound_ = -1.f;
  ConvJobs js;
  js.j[0] = ConvJob{src, stride, View(), parts_, nparts};
  hipLaunchKernelGGL(split16_convert_kernel, dim3(std::min(rows_p_, 512), 1), dim3(256), 0, cur_stream(), js, tw_log2_for(ld_ >> 3));
  return true;
}
bool PlaneSet::ConvertFrom(const float *src, int rows, int cols, int stride) {
  if ((cols & 3) || (stride & 3) || !aligned16(src)) return false;
  if (!Reserve(rows, cols)) return false;
  {
    const CoopConvJob job = {src, stride, nullptr, 0, View(), nullptr, 0};
    if (coop_convert_launch(&job, 1)) { host_bound_ = -1.f; return true; }   // one launch instead of two
  }
  MaxJobs ms;
  ms.j[0] = MaxJob{src, rows, cols, stride, parts_};
  hipLaunchKernelGGL(s16_absmax_kernel, dim3(kS16ConvParts, 1), dim3(256), 0, cur_stream(), ms, tw_log2_for(cols >> 2));
  return ConvertWithParts(src, rows, cols, stride, kS16ConvParts);
}
// several matrices in one maximum launch and one conversion launch (the recurrent layers convert up to seven small tensors at a time)
bool PlaneSet::ConvertMany(const ConvertSpec *specs, int n, const SeqFillJob *fill, bool *fill_done) {
  if (fill_done) *fill_done = false;
  if (n <= 0 || n > kS16MaxJobs) return false;
  MaxJobs ms;
  ConvJobs js;
  int max_c4 = 0, max_k8 = 0, max_rows = 0;
  bool need_max = false;
  for (int i = 0; i < n; i++) {
    const ConvertSpec &c = specs[i];
    if (!c.planes || !c.src || (c.cols & 3) || (c.stride & 3) || !aligned16(c.src)) return false;
    if (!c.planes->Reserve(c.rows, c.cols)) return false;
    c.planes->host_bound_ = -1.f;
    const bool given = c.parts != nullptr && c.nparts > 0 && c.nparts <= kS16MaxParts;
    need_max = need_max || !given;
    ms.j[i] = MaxJob{c.src, c.rows, c.cols, c.stride, c.planes->parts_};
    js.j[i] = given ? ConvJob{c.src, c.stride, c.planes->View(), c.parts, c.nparts} : ConvJob{c.src, c.stride, c.planes->View(), c.planes->parts_, kS16ConvParts};
    max_c4 = std::max(max_c4, c.cols >> 2);
    max_k8 = std::max(max_k8, c.planes->ld_ >> 3);
    max_rows = std::max(max_rows, c.planes->rows_p_);
  }
  if (need_max) {   // one launch for the maxima and the planes where the matrices fit one resident grid
    CoopConvJob cj[kS16MaxJobs];
    for (int i = 0; i < n; i++) {
      const ConvertSpec &c = specs[i];
      const bool given = c.parts != nullptr && c.nparts > 0 && c.nparts <= kS16MaxParts;
      cj[i] = CoopConvJob{c.src, c.stride, nullptr, 0, c.planes->View(), given ? c.parts : nullptr, given ? c.nparts : 0};
    }
    if (coop_convert_launch(cj, n, fill)) {
      if (fill_done) *fill_done = fill != nullptr;
      return true;
    }
  }
  if (need_max) hipLaunchKernelGGL(s16_absmax_kernel, dim3(kS16ConvParts, n), dim3(256), 0, cur_stream(), ms, tw_log2_for(max_c4));
  hipLaunchKernelGGL(split16_convert_kernel, dim3(std::min(max_rows, 512), n), dim3(256), 0, cur_stream(), js, tw_log2_for(max_k8));
  return true;
}
const float *PlaneSet::OneBound() {
  static float *one = [] {
    float *p = nullptr;
    const float v = 1.0f;
    if (hipMalloc(&p, 256) != hipSuccess || hipMemcpy(p, &v, sizeof(v), hipMemcpyHostToDevice) != hipSuccess) { set_error("PlaneSet::OneBound: allocation failed"); return static_cast<float *>(nullptr); }
    return p;
  }();
  return one;
}

// C = epilogue(alpha op(A) op(B) + beta C) from planes.  a_kc: A is stored [M x K] (else [K x M]); b_kc: B is stored [N x K]
// (else [K x N]).  For a pair (g.pair) a1 / b1 are the second product's operands.  false: not eligible (nothing was launched).
// Column statistics and (transposed A) column sums are formed in the kernel.
bool gemm_split16_planes_launch(GemmArgs &g, bool a_kc, bool b_kc, const S16View &a, const S16View &b, const S16View *a1, const S16View *b1,
                                int cfg) {
  if (g.split_k > 1) return false;   // (the split is chosen here, not by the caller)
  if (g.M < 64 || g.N < 64 || g.K < 32) return false;
  if (g.pair && (!a1 || !b1)) return false;
  S16Operands ops;
  ops.a = a; ops.b = b;
  ops.a1 = a1 ? *a1 : a; ops.b1 = b1 ? *b1 : b;
  ops.kp = (g.K + BKH - 1) / BKH * BKH;
  // the planes must describe the operands of this product
  auto fits = [&](const S16View &v, bool kc, int outer) { return v.hi && (kc ? (v.rows == outer && v.cols == g.K) : (v.rows == g.K && v.cols == outer)); };
  if (!fits(ops.a, a_kc, g.M) || !fits(ops.b, b_kc, g.N) || !fits(ops.a1, a_kc, g.M) || !fits(ops.b1, b_kc, g.N)) return false;
  t_last_parts = 0;
  // planes / maxima of the output are written by the 16-byte epilogue only: where that does not apply the request is dropped (the
  // caller sees aslp_gemm_last_parts() == 0 and converts for itself)
  auto drop_extras = [](aslp_gemm_epilogue &ep) { ep.planes_of = 0; ep.wmax_parts = ep.cmax_parts = nullptr; ep.bound_w_parts = ep.bound_c_parts = nullptr; };
  if (!(g.wide_epilogue && gemm_epilogue_wide_ok(g))) drop_extras(g.ep);
  if (g.pair) {
    GemmArgs g1 = g;
    g1.C = g.C1; g1.ep = g.ep1;
    if (!(g.wide_epilogue && gemm_epilogue_wide_ok(g1))) drop_extras(g.ep1);
  }
  auto launch = [&](GemmArgs &ga, int c) {
    if (a_kc && b_kc) return launch_s16_layout<true, true>(ga, ops, c);
    if (a_kc && !b_kc) return launch_s16_layout<true, false>(ga, ops, c);
    if (!a_kc && !b_kc) return launch_s16_layout<false, false>(ga, ops, c);
    return launch_s16_layout<false, true>(ga, ops, c);
  };
  // A long reduction on a grid that cannot fill the chip (the minibatch-256 layer products: 64 tiles of 64 x 128 for 256 CUs): K is split
  // over blockIdx.y, the chunks' partial products are added in chunk order by gemm_glds.hip's second launch, which also runs the epilogue
  // (bias, clip, SGD step, activation output -- not planes / maxima / column statistics: such requests keep the single launch).
  const long tiles = (long)((g.M + 63) / 64) * ((g.N + 127) / 128) * (g.pair ? 2 : 1);
  // per-workgroup maxima go to arrays of kS16MaxParts floats (PlaneSet::Parts, the components' own): a grid with more workgroups than
  // that leaves none (the smallest tile of such a grid is 64 x 128), and planes of updated weights, whose bound is formed from maxima, go too
  if (tiles > kS16MaxParts) {
    auto drop_maxima = [](aslp_gemm_epilogue &ep) {
      if (ep.planes_of == 1) { ep.planes_of = 0; ep.bound_w_parts = ep.bound_c_parts = nullptr; }
      ep.wmax_parts = ep.cmax_parts = nullptr;
    };
    drop_maxima(g.ep);
    if (g.pair) drop_maxima(g.ep1);
  }
  const bool extras = g.ep.planes_of != 0 || g.ep.wmax_parts || g.ep.cmax_parts || g.ep.colstats || g.ep.colsum ||
                      (g.pair && (g.ep1.planes_of != 0 || g.ep1.wmax_parts || g.ep1.cmax_parts || g.ep1.colstats || g.ep1.colsum));
  static const int splitk_off = [] { const char *e = getenv("ASLP_GEMM_SPLITK"); return e && atoi(e) == 0; }();
  // (act_out planes asked for by a forward product are given up for the split: the consumer converts the small activation matrix itself)
  // (the second launch writes the planes of an activation output and the maxima of |C| itself: those two requests go with the split)
  const bool reduce_serves = !g.ep.wmax_parts && !g.ep.colstats && !g.ep.colsum && g.ep.planes_of != 1 && !g.pair;
  // ... unless 32 x 64 tiles fill the chip in ONE round with the whole reduction (256 x 2048 outputs: 15.2 us against 19.4 us for the
  // two launches of the split; the epilogue, with whatever it was asked to leave, stays in the product's launch)
  static const int small_off = [] { const char *e = getenv("ASLP_GEMM_S16_SMALL"); return e && atoi(e) == 0; }();   // A/B switch
  const long t32 = (long)((g.M + 31) / 32) * ((g.N + 63) / 64) * (g.pair ? 2 : 1);
  // (with planes / maxima to leave, from K = 256: there the 64 x 128 tile's launch, 64 workgroups each with a long epilogue, is the slower
  // one -- the 440-input layer of the minibatch-256 net 18.6 against 25.7 us; a plain product of that shape is faster on 64 x 128: 9.7 / 12.6)
  const bool leaves = g.ep.planes_of != 0 || g.ep.wmax_parts || g.ep.cmax_parts;
  static const int leaves_min_k = [] { const char *e = getenv("ASLP_GEMM_S16_SMALL_MINK"); return e ? atoi(e) : 256; }();   // (tuning aid)
  if (!small_off && cfg == 0 && tiles <= 64 && t32 <= 256 && g.K >= (leaves ? leaves_min_k : 1024)) return launch(g, 304);
  if (!splitk_off && (!extras || reduce_serves) && (cfg == 0 || cfg == 308) && tiles <= 128 && g.K >= 1024) {
    int split = (int)(256 / tiles);
    if (split > g.K / 256) split = g.K / 256;
    if (split > 8) split = 8;
    int chunk = ((ops.kp / BKH + split - 1) / split) * BKH;
    split = (ops.kp + chunk - 1) / chunk;
    if (split >= 2) {
      const long stride = (long)g.M * g.N;
      const int np = g.pair ? 2 : 1;
      float *part = static_cast<float *>(scratch(kScratchSplitK, sizeof(float) * (size_t)stride * split * np));
      if (part) {
        GemmArgs pg = g;
        pg.C = part; pg.ldc = g.N; pg.alpha = 1.0f; pg.beta = 0.0f; pg.ep = aslp_gemm_epilogue();
        pg.C1 = part + (size_t)stride * split; pg.ep1 = aslp_gemm_epilogue();
        pg.split_k = split; pg.k_chunk = chunk; pg.split_stride = stride;
        if (launch(pg, 308)) {
          GemmArgs r = g;
          r.split_k = 0;
          const int wgs = gemm_splitk_reduce(part, split, stride, r);
          t_last_parts = (r.ep.planes_of == 2 || r.ep.cmax_parts) ? wgs : 0;
          return true;
        }
      }
    }
  }
  return launch(g, cfg);
}

// The same with either operand given as fp32 only (pa / pb NULL): its planes are made in the call's scratch, by a maximum pass and a
// conversion in front of the product.
bool gemm_split16_launch(GemmArgs &g, bool a_kc, bool b_kc, int cfg, const S16View *pa, const S16View *pb) {
  if (g.pair || g.split_k > 1) return false;
  if (g_split16_tile_override >= 0) cfg = g_split16_tile_override;
  if (!gemm_split16_serves(g.M, g.N, g.K)) return false;
  if ((!pa && !(g.A && g.a_vec)) || (!pb && !(g.B && g.b_vec))) return false;
  if (pa && pb) return gemm_split16_planes_launch(g, a_kc, b_kc, *pa, *pb, nullptr, nullptr, cfg);
  auto pad = [](int x) { return (x + kS16Pad - 1) / kS16Pad * kS16Pad; };
  const int a_rows = a_kc ? g.M : g.K, a_cols = a_kc ? g.K : g.M, b_rows = b_kc ? g.N : g.K, b_cols = b_kc ? g.K : g.N;
  const size_t plane_a = pa ? 0 : (size_t)pad(a_rows) * s16_plane_ld(a_cols), plane_b = pb ? 0 : (size_t)pad(b_rows) * s16_plane_ld(b_cols);
  const size_t head = 256 + sizeof(float) * 2 * kS16ConvParts;
  const size_t bytes = head + sizeof(h16) * 2 * (plane_a + plane_b);
  unsigned char *buf = static_cast<unsigned char *>(scratch(kScratchSplit16, bytes));
  if (!buf) return false;
  unsigned *slots = reinterpret_cast<unsigned *>(buf);
  float *part = reinterpret_cast<float *>(buf + 256);
  h16 *ah = reinterpret_cast<h16 *>(buf + head), *al = ah + plane_a, *bh = al + plane_a, *bl = bh + plane_b;
  const S16View va = pa ? *pa : S16View{ah, al, s16_plane_ld(a_cols), a_rows, a_cols, slots};
  const S16View vb = pb ? *pb : S16View{bh, bl, s16_plane_ld(b_cols), b_rows, b_cols, slots + 1};
  const MaxJob ma = {g.A, a_rows, a_cols, g.lda, part}, mb = {g.B, b_rows, b_cols, g.ldb, part + kS16ConvParts};
  const ConvJob ca = {g.A, g.lda, va, part, kS16ConvParts}, cb = {g.B, g.ldb, vb, part + kS16ConvParts, kS16ConvParts};
  // One launch for maxima and planes where the matrices fit one resident grid (nn_fused.hip copy_planes_coop) -- which writes the matrix
  // region only: scratch planes are not zeroed, so only for operands without padding rows / columns.
  {
    auto unpadded = [&](const S16View &v) { return pad(v.rows) == v.rows && v.ld == v.cols; };
    CoopConvJob cj[2];
    int n = 0;
    bool ok = true;
    if (!pa) { cj[n++] = CoopConvJob{g.A, g.lda, nullptr, 0, va, nullptr, 0}; ok = ok && unpadded(va); }
    if (!pb) { cj[n++] = CoopConvJob{g.B, g.ldb, nullptr, 0, vb, nullptr, 0}; ok = ok && unpadded(vb); }
    if (ok && n > 0 && coop_convert_launch(cj, n)) return gemm_split16_planes_launch(g, a_kc, b_kc, va, vb, nullptr, nullptr, cfg);
  }
  // (matrices in one launch share the threads-per-row choice: the widest one's)
  MaxJobs ms;
  ConvJobs js;
  if (!pa && !pb) {
    ms.j[0] = ma; ms.j[1] = mb; js.j[0] = ca; js.j[1] = cb;
    hipLaunchKernelGGL(s16_absmax_kernel, dim3(kS16ConvParts, 2), dim3(256), 0, cur_stream(), ms, tw_log2_for(std::max(a_cols, b_cols) >> 2));
    hipLaunchKernelGGL(split16_convert_kernel, dim3(512, 2), dim3(256), 0, cur_stream(), js, tw_log2_for(std::max(va.ld, vb.ld) >> 3));
  } else {
    ms.j[0] = pa ? mb : ma;
    js.j[0] = pa ? cb : ca;
    hipLaunchKernelGGL(s16_absmax_kernel, dim3(kS16ConvParts, 1), dim3(256), 0, cur_stream(), ms, tw_log2_for(ms.j[0].cols >> 2));
    hipLaunchKernelGGL(split16_convert_kernel, dim3(512, 1), dim3(256), 0, cur_stream(), js, tw_log2_for(js.j[0].pl.ld >> 3));
  }
  return gemm_split16_planes_launch(g, a_kc, b_kc, va, vb, nullptr, nullptr, cfg);
}

int gemm_split16_last_parts() { return t_last_parts; }
int gemm_split16_last_tile() { return t_last_cfg_s16; }
void gemm_split16_reset_last_parts() { t_last_parts = 0; }
// most per-workgroup maxima a split-fp16 product of this output shape leaves (the caller's arrays must hold them)
int gemm_split16_max_parts(int M, int N) { return std::max(((M + 63) / 64) * ((N + 127) / 128), ((M + 31) / 32) * ((N + 63) / 64) <= 256 ? ((M + 31) / 32) * ((N + 63) / 64) : 0); }

}  // namespace aslp

extern "C" {
void aslp_keep_weight_planes(int on) { aslp::g_keep_override = on < 0 ? -1 : (on != 0); }
void aslp_params_changed(void) {
  aslp::join_side_stream();   // weight updates the calling thread's latest backward pass left running beside it (Nnet::Backpropagate): the caller is about to touch the parameters
  aslp::g_param_epoch.fetch_add(1, std::memory_order_relaxed);
}
void aslp_gemm_split16(int on) { aslp::g_split16_override = on < 0 ? -1 : (on != 0); }
void aslp_gemm_split16_tile(int cfg) { aslp::g_split16_tile_override = cfg < 0 ? -1 : cfg; }
int aslp_gemm_last_parts(void) { return aslp::gemm_split16_last_parts(); }
void aslp_weight_bound(const float *w_parts, int n_w, const float *c_parts, int n_c, const aslp_planes *a, const aslp_planes *b, int K, float alpha,
                       float beta, float w_alpha, float clip, aslp_planes *w_planes) {
  using namespace aslp;
  if (!w_parts || n_w <= 0 || !a || !b || !w_planes) { set_error("aslp_weight_bound: missing argument"); return; }
  BoundJob j = {w_parts, n_w, c_parts, c_parts ? n_c : 0, reinterpret_cast<const PlaneSet *>(a)->Slot(), reinterpret_cast<const PlaneSet *>(b)->Slot(),
                (float)K, alpha, beta, w_alpha, clip, reinterpret_cast<PlaneSet *>(w_planes)->Slot()};
  hipLaunchKernelGGL(s16_weight_bound_kernel, dim3(1), dim3(256), 0, cur_stream(), j);
  reinterpret_cast<PlaneSet *>(w_planes)->ForgetHostBound();
  check_launch("aslp_weight_bound");
}
void aslp_absmax_parts(const float *src, MatrixDim d, float *parts) {
  using namespace aslp;
  if (!src || !parts || (d.cols & 3) || (d.stride & 3) || !aligned16(src)) { set_error("aslp_absmax_parts: unsupported matrix"); return; }
  MaxJobs ms;
  ms.j[0] = MaxJob{src, d.rows, d.cols, d.stride, parts};
  hipLaunchKernelGGL(s16_absmax_kernel, dim3(kS16ConvParts, 1), dim3(256), 0, cur_stream(), ms, tw_log2_for(d.cols >> 2));
  check_launch("aslp_absmax_parts");
}
}
