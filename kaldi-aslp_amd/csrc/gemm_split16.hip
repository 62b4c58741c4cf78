// gemm_split16.hip -- fp32 GEMM carried on the fp16 matrix instruction (v_mfma_f32_32x32x16_f16) with fp32-equivalent operands.
//
// The fp32 MFMA rate of gfx950 is 256 FLOP/clk/CU (157 TFLOP/s) whatever the instruction shape; the fp16 instruction runs 16 x that.
// An fp32 value x is carried as TWO fp16 pieces behind a power-of-two scale s of its matrix (s puts the matrix's largest |x| in
// [2^13, 2^14)):  x s = hi + 2^-11 lo',  hi = fp16(x s),  lo' = fp16((x s - hi) 2^11)  -- 22 significant bits for every element down to
// 2^-28 of the matrix's largest (fp16 is normal over 29 binades), and an element below that contributes < 2^-39 of the largest either way.
// A product a b then is  hi_a hi_b + 2^-11 (hi_a lo'_b + lo'_a hi_b)  (+ 2^-22 lo'_a lo'_b, dropped: below the pieces' own rounding):
// three instructions per 16-wide k step, every partial product exact in the multiplier, accumulated in fp32 in two accumulators (main and
// cross terms) that are joined once in front of the epilogue.  Measured against double (tests/test_gemm_split16_gpu.py) the result is as
// close as the fp32 instruction's.
//
// Operands reach the kernel as PLANES: hi[rows][Kp] and lo'[rows][Kp] fp16, Kp = K rounded up to 64 (zeros behind K), always with the
// reduction index contiguous ("NT" form) -- split16_convert writes them from the fp32 operand in one pass, transposing through LDS
// where the operand is stored reduction-major.  Each (operand, orientation) of a training step is used by exactly one product, so the
// planes are scratch of the call.  The kernel itself is gemm_glds.hip's pipeline with both operands K-contiguous: LDS-DMA into
// 128-byte-row tiles (same swizzle), fragments double-buffered in registers, the epilogues of gemm_common.h unchanged (the 32 x 32 fp16
// instruction has the fp32 one's result layout).
//
// A/B: ASLP_GEMM_SPLIT_F16=1 routes eligible aslp_sgemm_ex products here (default off: the fp32 instruction stays the shipped path).
#include "gemm_common.h"
#include "scratch.h"

#pragma clang diagnostic ignored "-Winline-asm"  // the DMA asm clobbers m0 on purpose

namespace aslp {
namespace {

typedef _Float16 h16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
constexpr int BKH = 64;      // halves per K tile: 128-byte rows, the LDS geometry of gemm_glds.hip's K-contiguous tiles
constexpr int KH = BKH / 16;  // instruction k steps per tile

__device__ __forceinline__ int kc_swizzle(int row) { return (row >> 1) & 7; }   // as gemm_glds.hip
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory", "m0");
}

// the scale of a matrix from the bits of its largest |x| (0, inf and NaN leave it at 1): *up = the exponent added
__device__ __forceinline__ float scale_from_max_bits(unsigned bits, float *inv) {
  const float mx = __uint_as_float(bits);
  int e = 0;
  (void)frexpf(mx, &e);   // mx = f 2^e, f in [0.5, 1)
  const bool scaled = mx > 0.f && mx < 3.0e38f;
  int up = scaled ? 14 - e : 0;
  up = up > 120 ? 120 : (up < -120 ? -120 : up);
  *inv = ldexpf(1.f, -up);
  return ldexpf(1.f, up);
}

// ---- largest |x| of two operands (blockIdx.y): every workgroup leaves its own maximum in part[operand][workgroup] (kMaxParts each) --
// no atomics and nothing to zero first; the conversion kernel reduces the partials (and leaves the result in slots[] for the product)
constexpr int kMaxParts = 256;
struct MaxJob { const float *p; int rows, cols, ld; };
__global__ void __launch_bounds__(256) absmax2_kernel(MaxJob a, MaxJob b, float *part) {
  const MaxJob j = blockIdx.y ? b : a;
  const int c4 = j.cols >> 2;
  const long n = (long)j.rows * c4;
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int r = (int)(i / c4), c = (int)(i - (long)r * c4);
    const float4 v = *reinterpret_cast<const float4 *>(j.p + (long)r * j.ld + 4 * c);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  // NaN anywhere: fmaxf drops it; the product then carries it through the pieces (fp16(NaN) = NaN) like the fp32 kernel would
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.y * kMaxParts + blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}
// the matrix maximum from the partials (every thread of a 256-thread workgroup gets it)
__device__ __forceinline__ float reduce_parts(const float *part, int nparts) {
  __shared__ float wm2[4];
  float m = (int)threadIdx.x < nparts ? part[threadIdx.x] : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wm2[threadIdx.x >> 6] = m;
  __syncthreads();
  return fmaxf(fmaxf(wm2[0], wm2[1]), fmaxf(wm2[2], wm2[3]));
}

__device__ __forceinline__ void split4(const float4 v, float s, half4 *hi, half4 *lo) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const h16 h = (h16)x[i];
    (*hi)[i] = h;
    (*lo)[i] = (h16)((x[i] - (float)h) * 2048.f);
  }
}

// ---- fp32 operand -> planes, reduction index already contiguous: out row r = src row r --------------------------------------------
struct ConvJob { const float *src; int rows, cols, ld; h16 *hi, *lo; int kp; int transpose; };
__global__ void __launch_bounds__(256) split16_convert_kernel(ConvJob a, ConvJob b, const float *part, int nparts, unsigned *slots) {
  const ConvJob j = blockIdx.z ? b : a;
  float inv;
  const unsigned mbits = __float_as_uint(reduce_parts(part + blockIdx.z * kMaxParts, nparts));
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) slots[blockIdx.z] = mbits;   // for the product kernel (launched behind this one)
  const float s = scale_from_max_bits(mbits, &inv);
  if (!j.transpose) {
    const int k4 = j.kp >> 2, c4 = j.cols >> 2;
    const long n = (long)j.rows * k4;
    for (long i = (long)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < n; i += (long)gridDim.x * gridDim.y * 256) {
      const int r = (int)(i / k4), c = (int)(i - (long)r * k4);
      half4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
      if (c < c4) split4(*reinterpret_cast<const float4 *>(j.src + (long)r * j.ld + 4 * c), s, &hi, &lo);
      *reinterpret_cast<half4 *>(j.hi + (long)r * j.kp + 4 * c) = hi;
      *reinterpret_cast<half4 *>(j.lo + (long)r * j.kp + 4 * c) = lo;
    }
    return;
  }
  // transpose: out row = src column, out k = src row.  64 x 64 tiles through LDS: read along the source rows, write along k.
  __shared__ float tile[64][65];
  const int tiles_c = (j.cols + 63) >> 6, tiles_k = j.kp >> 6;
  for (int t = blockIdx.y * gridDim.x + blockIdx.x; t < tiles_c * tiles_k; t += gridDim.x * gridDim.y) {
    const int tk = t / tiles_c, tc = t - tk * tiles_c;
    const int k0 = tk * 64, c0 = tc * 64;
#pragma unroll
    for (int p = 0; p < 4; p++) {   // 64 source rows (k) x 16 float4
      const int kr = (threadIdx.x >> 4) + 16 * p, c = 4 * (threadIdx.x & 15);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + kr < j.rows && c0 + c < j.cols) v = *reinterpret_cast<const float4 *>(j.src + (long)(k0 + kr) * j.ld + c0 + c);   // cols % 4 == 0
      tile[kr][c] = v.x; tile[kr][c + 1] = v.y; tile[kr][c + 2] = v.z; tile[kr][c + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; p++) {   // 64 out rows (source columns) x 16 groups of 4 k
      const int orow = (threadIdx.x >> 4) + 16 * p, k = 4 * (threadIdx.x & 15);
      if (c0 + orow < j.cols) {
        const float4 v = make_float4(tile[k][orow], tile[k + 1][orow], tile[k + 2][orow], tile[k + 3][orow]);
        half4 hi, lo;
        split4(v, s, &hi, &lo);
        *reinterpret_cast<half4 *>(j.hi + (long)(c0 + orow) * j.kp + k0 + k) = hi;
        *reinterpret_cast<half4 *>(j.lo + (long)(c0 + orow) * j.kp + k0 + k) = lo;
      }
    }
    __syncthreads();
  }
}

// ---- the product ---------------------------------------------------------------------------------------------------------------
struct S16Planes { const h16 *ah, *al, *bh, *bl; int kp; const unsigned *slots; };

template <int BM, int BN, int WGM, int WGN, int NS>
__global__ void __launch_bounds__(64 * WGM * WGN) gemm_s16_glds(GemmArgs g, S16Planes pl) {
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  // a stage, in floats (128-byte rows = 32 floats): A_hi | A_lo | B_hi | B_lo
  constexpr int A_FLOATS = BM * 32, B_FLOATS = BN * 32, STAGE = 2 * (A_FLOATS + B_FLOATS);
  constexpr int SLOTS_A = BM / 8, SLOTS_B = BN / 8, SLOTS = 2 * (SLOTS_A + SLOTS_B);   // 1-KiB DMA units per tile
  static_assert(SLOTS % NW == 0, "DMA units must divide over the waves");
  constexpr int G = SLOTS / NW;
  constexpr int D = NS - 1;
  constexpr int NM = KH * 3 * TM * TN, NRD = KH * 2 * (TM + TN), SB = NM / 2 - 1;
  constexpr int UNROLL = (NS % 2 == 0) ? NS : 2 * NS;
  static_assert(G <= SB + 1, "not enough MFMA slots before the barrier");
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;

  int tm, tn;
  xcd_tile<BM, BN>(g, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, lh = lane >> 5;
  const int ktiles = pl.kp / BKH;

  // ---- DMA descriptors: unit = 8 rows x 128 B of one plane
  const h16 *src[G];
  unsigned dst_off[G];
  static_for<0, G>([&](auto U_) {
    constexpr int u = decltype(U_)::value;
    const int slot = wave + u * NW;  // wave-uniform
    const int r = lane >> 3;
    // plane order inside a stage: A_hi [0, SLOTS_A), A_lo, B_hi [2 SLOTS_A, ...), B_lo
    const bool is_a = slot < 2 * SLOTS_A;
    const int s2 = is_a ? slot : slot - 2 * SLOTS_A, per = is_a ? SLOTS_A : SLOTS_B;
    const bool lo_plane = s2 >= per;
    const int sr = lo_plane ? s2 - per : s2;   // 8-row group within the tile
    const int c = (lane & 7) ^ kc_swizzle(sr * 8 + r);
    int row = (is_a ? m0 : n0) + sr * 8 + r;
    const int lim = is_a ? g.M : g.N;
    row = row < lim ? row : lim - 1;
    const h16 *base = is_a ? (lo_plane ? pl.al : pl.ah) : (lo_plane ? pl.bl : pl.bh);
    src[u] = base + (long)row * pl.kp + 8 * c;
    dst_off[u] = slot * 1024;
  });
  auto dma_unit = [&](auto U_, auto ST_, int r) {
    constexpr int u = decltype(U_)::value, st = decltype(ST_)::value;
    glds16(src[u], __builtin_amdgcn_readfirstlane(lds_base + st * STAGE * 4 + dst_off[u]));
    src[u] += (r + 1 < ktiles) ? BKH : 0;   // requests past the last tile fetch it again into a stage nobody reads
  };

  // ---- fragments: per-lane float offsets inside a plane tile (row * 32 + swizzled 16-byte chunk * 4)
  int a_off[TM][KH], b_off[TN][KH];
#pragma unroll
  for (int t = 0; t < TM; t++)
#pragma unroll
    for (int h = 0; h < KH; h++) {
      const int row = wm * WM + t * 32 + l31;
      a_off[t][h] = row * 32 + (((2 * h + lh) ^ kc_swizzle(row)) << 2);
    }
#pragma unroll
  for (int t = 0; t < TN; t++)
#pragma unroll
    for (int h = 0; h < KH; h++) {
      const int col = wn * WN + t * 32 + l31;
      b_off[t][h] = 2 * A_FLOATS + col * 32 + (((2 * h + lh) ^ kc_swizzle(col)) << 2);
    }
  struct Frag {
    half8 ah[KH][TM], al[KH][TM], bh[KH][TN], bl[KH][TN];
  };
  auto read_unit = [&](auto ST_, Frag &f, auto R_) {
    constexpr int r = decltype(R_)::value, st = decltype(ST_)::value;
    constexpr int h = r / (2 * (TM + TN)), q = r % (2 * (TM + TN)), t = q >> 1, lo = q & 1;
    const float *stage = lds + st * STAGE;
    if constexpr (t < TM) {
      const half8 v = *reinterpret_cast<const half8 *>(stage + a_off[t][h] + (lo ? A_FLOATS : 0));
      if constexpr (lo) f.al[h][t] = v; else f.ah[h][t] = v;
    } else {
      constexpr int tb = t - TM;
      const half8 v = *reinterpret_cast<const half8 *>(stage + b_off[tb][h] + (lo ? B_FLOATS : 0));
      if constexpr (lo) f.bl[h][tb] = v; else f.bh[h][tb] = v;
    }
  };

  f32x16 acc[TM][TN], accx[TM][TN];   // hi hi; hi lo' + lo' hi
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) { acc[i][j][e] = 0.0f; accx[i][j][e] = 0.0f; }
  auto mma_unit = [&](const Frag &f, auto M_) {
    constexpr int m = decltype(M_)::value;
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
  static_for<0, D>([&](auto T_) {
    constexpr int t = decltype(T_)::value;
    static_for<0, G>([&](auto U_) { dma_unit(U_, T_, t); });
  });
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

  // join the two accumulators and undo the operand scales (exact powers of two) through alpha
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = fmaf(accx[i][j][e], 0x1p-11f, acc[i][j][e]);
  {
    float inv_a, inv_b;
    (void)scale_from_max_bits(pl.slots[0], &inv_a);
    (void)scale_from_max_bits(pl.slots[1], &inv_b);
    g.alpha *= inv_a * inv_b;
  }
  if (g.ep.colstats != nullptr) gemm_colstats<TM, TN>(g, acc, m0 + wm * WM, n0 + wn * WN, l31, lh);  // uniform
  static_assert(NW * 32 * kEpiPitch * (int)sizeof(float) <= NS * STAGE * (int)sizeof(float), "the waves' epilogue slices must fit into the operand LDS");
  if (g.wide_epilogue && gemm_epilogue_wide_ok(g)) {  // uniform
    __builtin_amdgcn_s_barrier();
    gemm_epilogue_wide<TM, TN>(g, acc, m0 + wm * WM, n0 + wn * WN, lane, lds + wave * 32 * kEpiPitch);
  } else {
    gemm_epilogue<TM, TN>(g, acc, m0 + wm * WM, n0 + wn * WN, l31, lh);
  }
}

template <int BM, int BN, int WGM, int WGN, int NS>
void launch_s16(GemmArgs &g, const S16Planes &pl) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  constexpr int lds_bytes = NS * 2 * (BM + BN) * 128;
  auto kern = gemm_s16_glds<BM, BN, WGM, WGN, NS>;
  static bool attr_set = false;
  if (!attr_set) {
    if (lds_bytes > 48 * 1024)
      ASLP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n), dim3(64 * WGM * WGN), lds_bytes, cur_stream(), g, pl);
}

}  // namespace

int g_split16_override = -1;   // aslp_gemm_split16(): -1 = the environment decides
bool gemm_split16_enabled() {
  static const bool on = getenv("ASLP_GEMM_SPLIT_F16") != nullptr && getenv("ASLP_GEMM_SPLIT_F16")[0] == '1';
  return g_split16_override >= 0 ? g_split16_override != 0 : on;
}

// C = epilogue(alpha op(A) op(B) + beta C) through fp16 planes.  a_kc: A is stored [M x K] (else [K x M]); b_kc: B is stored [N x K]
// (else [K x N]).  false: not eligible (the caller runs the fp32 kernels).  Column sums (ep.colsum) stay the caller's job.
bool gemm_split16_launch(GemmArgs &g, bool a_kc, bool b_kc, int cfg) {
  if (g.pair || g.split_k > 1) return false;
  if (g.M < 128 || g.N < 128 || g.K < 64 || !g.a_vec || !g.b_vec) return false;
  if ((g.M & 3) || (g.N & 3) || (g.K & 3)) return false;
  const int kp = (g.K + BKH - 1) / BKH * BKH;
  const size_t plane_a = (size_t)g.M * kp, plane_b = (size_t)g.N * kp;
  const size_t head = 256 + sizeof(float) * 2 * kMaxParts;
  const size_t bytes = head + sizeof(h16) * 2 * (plane_a + plane_b);
  unsigned char *buf = static_cast<unsigned char *>(scratch(kScratchSplit16, bytes));
  if (!buf) return false;
  unsigned *slots = reinterpret_cast<unsigned *>(buf);
  float *part = reinterpret_cast<float *>(buf + 256);
  h16 *ah = reinterpret_cast<h16 *>(buf + head), *al = ah + plane_a, *bh = al + plane_a, *bl = bh + plane_b;
  MaxJob ma = {g.A, a_kc ? g.M : g.K, a_kc ? g.K : g.M, g.lda}, mb = {g.B, b_kc ? g.N : g.K, b_kc ? g.K : g.N, g.ldb};
  hipLaunchKernelGGL(absmax2_kernel, dim3(kMaxParts, 2), dim3(256), 0, cur_stream(), ma, mb, part);
  ConvJob ca = {g.A, ma.rows, ma.cols, g.lda, ah, al, kp, a_kc ? 0 : 1}, cb = {g.B, mb.rows, mb.cols, g.ldb, bh, bl, kp, b_kc ? 0 : 1};
  hipLaunchKernelGGL(split16_convert_kernel, dim3(256, 4, 2), dim3(256), 0, cur_stream(), ca, cb, part, kMaxParts, slots);
  S16Planes pl = {ah, al, bh, bl, kp, slots};
  if (cfg == 0)   // 128 x 128 where that still gives every CU a workgroup, else 64 x 128 (measured: 1024 x 2048 x 2048 52 against 70 us per call)
    cfg = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) >= 224 ? 311 : 308;
  switch (cfg) {
    case 311: launch_s16<128, 128, 2, 2, 2>(g, pl); break;
    case 312: launch_s16<128, 128, 2, 4, 2>(g, pl); break;
    case 308: launch_s16<64, 128, 2, 2, 3>(g, pl); break;
    default: return false;
  }
  return true;
}

}  // namespace aslp

extern "C" void aslp_gemm_split16(int on) { aslp::g_split16_override = on < 0 ? -1 : (on != 0); }
