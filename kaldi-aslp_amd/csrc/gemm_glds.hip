// gemm_glds.hip -- fp32 MFMA GEMM whose operand tiles travel global -> LDS directly (LDS-DMA,
// global_load_lds_dwordx4), gfx950 only.
//
// Measured on MI355X (devtools/bench_gemm.py ablations): in the register-staged kernel of
// gemm.hip the ds_write_b128 pass that moves a K tile from VGPRs into LDS costs ~13 us of a 92 us layer GEMM
// (more than the barrier and the global loads together) -- LDS write bandwidth, not latency.  LDS-DMA removes
// that pass, the staging VGPRs and the k-tail select VALU: a wave instruction moves 64 x 16 B = 1 KiB from
// per-lane global addresses to a lane-linear LDS block, asynchronously, counted by vmcnt.
//   * K-contiguous operands ("KC"): LDS image [rows][32] floats, 128-B rows, unpadded (the DMA destination is
//     lane-linear).  Bank conflicts of the ds_read_b128 fragment reads (32 lanes x same 16-B column) are
//     removed by an XOR swizzle of the 16-B chunk index with ((row >> 1) & 7) (kc_swizzle below), applied to the SOURCE address of the
//     DMA and to the read address (both sides or neither).
//   * row-contiguous operands ("RC"): LDS image [32][rows], read with ds_read_b32, conflict-free as is.
//   * NS LDS stages; tile t+NS-1 is requested while tile t is multiplied from registers (fragments are double
//     buffered in VGPRs, read one tile ahead); each wave waits with a COUNTED s_waitcnt vmcnt for its own part
//     of tile t+1, then one raw s_barrier publishes it.  All non-MFMA work of a K tile is slotted between its
//     MFMAs (sched_barrier pins the order).
// A partial last K tile is handled by pointing the out-of-range DMA lanes at a zero buffer.
// Eligibility: K % 4 == 0, 16-byte aligned operands, leading dimensions % 4 == 0 (else gemm.hip's kernel).
#include "gemm_common.h"
#include "scratch.h"

#pragma clang diagnostic ignored "-Winline-asm"  // the DMA asm clobbers m0 on purpose

namespace aslp {
namespace {

constexpr int BK = 32, KH = BK / 8;

// source of the DMA lanes whose k index lies beyond K in the last, partial K tile: LDS receives zeros there, so the
// tail needs no masking anywhere else
__device__ float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// XOR swizzle of the 16-byte chunk index of a K-contiguous LDS row (128 B = half a 256-byte bank row).  ds_read_b128 is served in four
// fixed groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- and a group
// is conflict-free when its 16 rows land on 16 distinct 16-byte slots of the bank row, i.e. on distinct (row & 1, chunk) pairs.  With
// the chunk XORed by (row & 7) rows 0-3 met rows 24-27 and rows 12-15 met rows 20-23 in every group: a 2-way conflict on EVERY fragment
// read (SQ_LDS_BANK_CONFLICT: 4.3 M cycles per NT layer product, 9 % of the kernel's time, 2.2 M for NN, 0.07 M for TN which has no
// K-contiguous operand).  (row >> 1) & 7 gives the 8 even and the 8 odd rows of each group 8 distinct chunks.
__device__ __forceinline__ int kc_swizzle(int row) { return (row >> 1) & 7; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA through inline asm: hipcc does not see a VMEM op writing LDS, so it neither drains vmcnt(0) before the
// next ds_read (it does for the builtin: every LDS read "may alias" the DMA target) nor counts these in its own
// vmcnt bookkeeping -- the kernel waits with explicit counted s_waitcnt vmcnt(N) instead.
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory", "m0");
}

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC, int NS, bool TAIL>
__global__ void __launch_bounds__(64 * WGM * WGN) gemm_f32_glds(GemmArgs g) {
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
  constexpr int SLOTS_A = BM / 8, SLOTS_B = BN / 8;                 // 1-KiB DMA units per tile
  static_assert((SLOTS_A + SLOTS_B) % NW == 0, "DMA units must divide over the waves");
  constexpr int G = (SLOTS_A + SLOTS_B) / NW;                        // DMA instructions per wave per tile
  constexpr int D = NS - 1;                                           // tiles requested ahead
  constexpr int NM = KH * 4 * TM * TN, NRD = KH * (TM + TN), SB = NM / 2 - 1;
  constexpr int UNROLL = (NS % 2 == 0) ? NS : 2 * NS;                 // stage and fragment-buffer indices both static
  static_assert(G <= SB + 1, "not enough MFMA slots before the barrier");
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;

  if (g.pair && blockIdx.z == 1) {  // second product of a pair (uniform)
    g.A = g.A1; g.B = g.B1; g.C = g.C1; g.ep = g.ep1;
  }
  if (g.split_k > 1) {  // this workgroup reduces over K chunk blockIdx.y only and leaves a plain partial product
    const int k0 = (int)blockIdx.y * g.k_chunk;
    g.A += A_KC ? (long)k0 : (long)k0 * g.lda;
    g.B += B_KC ? (long)k0 : (long)k0 * g.ldb;
    g.K = min(g.k_chunk, g.K - k0);
    g.C += (long)blockIdx.y * g.split_stride;
  }
  int tm, tn;
  xcd_tile<BM, BN>(g, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, lh = lane >> 5;
  const int kfull = g.K / BK, ktail = TAIL ? g.K % BK : 0, ktiles = kfull + (ktail ? 1 : 0);

  // ---- DMA descriptors of this wave's G units: per-lane source pointer of tile 0, per-tile pointer step, LDS offset
  const float *src[G];  // tile 0, then advanced tile by tile up to the last full tile
  long kstep[G], tail_off[G];  // per-tile pointer step; (source of the partial last tile) - (source of the last full tile), in floats
  unsigned dst_off[G];  // bytes within a stage
  static_for<0, G>([&](auto U_) {
    constexpr int u = decltype(U_)::value;
    const int slot = wave + u * NW;  // wave-uniform
    if (slot < SLOTS_A) {
      dst_off[u] = slot * 1024;
      if (A_KC) {  // rows 8*slot .. +7, 128 B each; source chunk = dest chunk ^ swz(row)
        const int r = lane >> 3, c = (lane & 7) ^ kc_swizzle(slot * 8 + r);
        int row = m0 + slot * 8 + r;
        row = row < g.M ? row : g.M - 1;
        src[u] = g.A + (long)row * g.lda + 4 * c;
        tail_off[u] = (4 * c < ktail ? src[u] + (long)kfull * BK : g_zero16) - (src[u] + (long)max(kfull - 1, 0) * BK);
        kstep[u] = BK;
      } else {     // [32][BM]: 256 floats = 256 / BM k-rows
        const int f = slot * 256 + 4 * lane, k = f / BM;
        int r4 = m0 + f % BM;
        r4 = r4 + 3 < g.M ? r4 : g.M - 4;
        src[u] = g.A + (long)k * g.lda + r4;
        tail_off[u] = (k < ktail ? src[u] + (long)kfull * BK * g.lda : g_zero16) - (src[u] + (long)max(kfull - 1, 0) * BK * g.lda);
        kstep[u] = (long)BK * g.lda;
      }
    } else {
      const int sb = slot - SLOTS_A;
      dst_off[u] = (A_FLOATS + sb * 256) * 4;
      if (B_KC) {
        const int r = lane >> 3, c = (lane & 7) ^ kc_swizzle(sb * 8 + r);
        int row = n0 + sb * 8 + r;
        row = row < g.N ? row : g.N - 1;
        src[u] = g.B + (long)row * g.ldb + 4 * c;
        tail_off[u] = (4 * c < ktail ? src[u] + (long)kfull * BK : g_zero16) - (src[u] + (long)max(kfull - 1, 0) * BK);
        kstep[u] = BK;
      } else {
        const int f = sb * 256 + 4 * lane, k = f / BN;
        int r4 = n0 + f % BN;
        r4 = r4 + 3 < g.N ? r4 : g.N - 4;
        src[u] = g.B + (long)k * g.ldb + r4;
        tail_off[u] = (k < ktail ? src[u] + (long)kfull * BK * g.ldb : g_zero16) - (src[u] + (long)max(kfull - 1, 0) * BK * g.ldb);
        kstep[u] = (long)BK * g.ldb;
      }
    }
  });
  // request tile r of unit u into stage ST.  src[u] points at tile min(r, kfull-1); requests past the last tile
  // (pipeline tail) fetch that tile again into a stage nobody reads.
  auto dma_unit = [&](auto U_, auto ST_, int r) {
    constexpr int u = decltype(U_)::value, st = decltype(ST_)::value;
    const float *p = src[u];
    if constexpr (TAIL) p += (ktail && r >= kfull) ? tail_off[u] : 0;  // wave-uniform condition
    glds16(p, __builtin_amdgcn_readfirstlane(lds_base + st * STAGE * 4 + dst_off[u]));
    src[u] += (r + 1 < kfull) ? kstep[u] : 0;
  };

  // ---- fragments: per-lane LDS float offsets inside a stage, one per (sub-tile, k-octet) ----------------------------
  int a_off[TM][KH], b_off[TN][KH];
#pragma unroll
  for (int t = 0; t < TM; t++)
#pragma unroll
    for (int h = 0; h < KH; h++) {
      const int row = wm * WM + t * 32 + l31;
      a_off[t][h] = A_KC ? row * BK + (((2 * h + lh) ^ kc_swizzle(row)) << 2) : (h * 8 + lh * 4) * BM + row;
    }
#pragma unroll
  for (int t = 0; t < TN; t++)
#pragma unroll
    for (int h = 0; h < KH; h++) {
      const int col = wn * WN + t * 32 + l31;
      b_off[t][h] = A_FLOATS + (B_KC ? col * BK + (((2 * h + lh) ^ kc_swizzle(col)) << 2) : (h * 8 + lh * 4) * BN + col);
    }
  struct Frag {
    float a[KH][TM][4], b[KH][TN][4];
  };
  auto read_unit = [&](auto ST_, Frag &f, auto R_) {
    constexpr int r = decltype(R_)::value, st = decltype(ST_)::value;
    constexpr int h = r / (TM + TN), t = r % (TM + TN);
    const float *stage = lds + st * STAGE;
    if constexpr (t < TM) {
      if (A_KC) {
        float4 v = *reinterpret_cast<const float4 *>(stage + a_off[t][h]);
        f.a[h][t][0] = v.x; f.a[h][t][1] = v.y; f.a[h][t][2] = v.z; f.a[h][t][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) f.a[h][t][j] = stage[a_off[t][h] + j * BM];
      }
    } else {
      constexpr int tb = t - TM;
      if (B_KC) {
        float4 v = *reinterpret_cast<const float4 *>(stage + b_off[tb][h]);
        f.b[h][tb][0] = v.x; f.b[h][tb][1] = v.y; f.b[h][tb][2] = v.z; f.b[h][tb][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) f.b[h][tb][j] = stage[b_off[tb][h] + j * BN];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
  // optional column sums of a transposed A operand (bias gradient on the weight-gradient GEMM): the first column
  // of tiles' wn == 0 waves add up the A fragments they multiply anyway
  const bool do_colsum = !A_KC && g.ep.colsum != nullptr && tn == 0 && wn == 0;  // wave-uniform
  float asum[TM];
#pragma unroll
  for (int i = 0; i < TM; i++) asum[i] = 0.0f;
  auto mma_unit = [&](const Frag &f, auto M_) {
    constexpr int m = decltype(M_)::value;
    constexpr int n = m % TN, i = (m / TN) % TM, j = (m / (TN * TM)) % 4, h = m / (TN * TM * 4);
    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[h][i][j], f.b[h][n][j], acc[i][n], 0, 0, 0);
  };

  // K tile t (stage index I % NS) from fcur; request tile t+D into stage (I+D) % NS; after the barrier read the
  // fragments of tile t+1 (stage (I+1) % NS) into fnxt.  I = t mod UNROLL is a compile-time constant.
  auto step = [&](auto I_, const Frag &fcur, Frag &fnxt, int t) {
    constexpr int I = decltype(I_)::value;
    using StReq = std::integral_constant<int, (I + D) % NS>;
    using StNxt = std::integral_constant<int, (I + 1) % NS>;
    static_for<0, NM>([&](auto S_) {
      constexpr int sidx = decltype(S_)::value;
      mma_unit(fcur, S_);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!A_KC && sidx == 0) {
        if (do_colsum) {
#pragma unroll
          for (int h = 0; h < KH; h++)
#pragma unroll
            for (int i = 0; i < TM; i++) asum[i] += (fcur.a[h][i][0] + fcur.a[h][i][1]) + (fcur.a[h][i][2] + fcur.a[h][i][3]);
        }
      }
      if constexpr (sidx <= SB) {
        static_for<sidx * G / (SB + 1), (sidx + 1) * G / (SB + 1)>([&](auto U_) { dma_unit(U_, StReq(), t + D); });
        if constexpr (sidx == SB) {
          wait_vmcnt<(D - 1) * G>();  // this wave's share of tile t+1 has landed; tiles t+2.. stay in flight
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
  // prologue: request tiles 0 .. D-1 (stages 0 .. D-1), wait for tile 0, read its fragments
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
  wait_vmcnt<0>();  // drain the clamped tail requests before the LDS block is released

  if (g.ep.colstats != nullptr && g.split_k <= 1) gemm_colstats<TM, TN>(g, acc, m0 + wm * WM, n0 + wn * WN, l31, lh);  // uniform
  static_assert(NW * 32 * kEpiPitch * (int)sizeof(float) <= NS * STAGE * (int)sizeof(float), "the waves' epilogue slices must fit into the operand LDS");
  if (g.wide_epilogue && gemm_epilogue_wide_ok(g) && g.split_k <= 1) {  // uniform
    __builtin_amdgcn_s_barrier();  // every wave is past its last operand read: the LDS is free
    gemm_epilogue_wide<TM, TN>(g, acc, m0 + wm * WM, n0 + wn * WN, lane, lds + wave * 32 * kEpiPitch);
  } else {
    gemm_epilogue<TM, TN>(g, acc, m0 + wm * WM, n0 + wn * WN, l31, lh);
  }
  if constexpr (!A_KC) {
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const float s_all = asum[i] + __shfl_xor(asum[i], 32, 64);  // the two lane halves hold disjoint k subsets
        const int row = m0 + wm * WM + i * 32 + l31;
        if (lh == 0 && row < g.M) {
          float v = s_all;
          if (g.ep.colsum_beta != 0.0f) v += g.ep.colsum_beta * g.ep.colsum[row];
          g.ep.colsum[row] = v;
          if (g.ep.colsum_w) g.ep.colsum_w[row] += g.ep.colsum_w_alpha * v;
        }
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC, int NS, bool TAIL>
void launch_t(GemmArgs &g) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  constexpr int lds_bytes = NS * (BM + BN) * BK * (int)sizeof(float);
  auto kern = gemm_f32_glds<BM, BN, WGM, WGN, A_KC, B_KC, NS, TAIL>;
  static bool attr_set = false;
  if (!attr_set) {
    if (lds_bytes > 48 * 1024)
      ASLP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, g.split_k > 1 ? g.split_k : 1, g.pair ? 2 : 1), dim3(64 * WGM * WGN), lds_bytes, cur_stream(), g);
}

template <int BM, int BN, int WGM, int WGN, bool A_KC, bool B_KC, int NS>
void launch(GemmArgs &g) {
  if (g.K % BK == 0) launch_t<BM, BN, WGM, WGN, A_KC, B_KC, NS, false>(g);
  else launch_t<BM, BN, WGM, WGN, A_KC, B_KC, NS, true>(g);
}

template <bool A_KC, bool B_KC>
bool launch_cfg(GemmArgs &g, int cfg) {
  switch (cfg) {
    case 212: launch<64, 128, 2, 4, A_KC, B_KC, 3>(g); return true;   // 8 waves, 3 stages (72 KiB)
    case 213: launch<64, 128, 2, 4, A_KC, B_KC, 4>(g); return true;   // 4 stages (96 KiB)
    case 208: launch<64, 128, 2, 2, A_KC, B_KC, 3>(g); return true;   // 4 waves, 32 x 64 per wave
    case 207: launch<64, 64, 2, 2, A_KC, B_KC, 4>(g); return true;    // 4 waves, 2 workgroups per CU
    case 211: launch<128, 128, 2, 4, A_KC, B_KC, 3>(g); return true;  // 8 waves, 64 x 32 per wave
    case 205: launch<32, 128, 1, 4, A_KC, B_KC, 4>(g); return true;   // 4 waves, 32-row tile: twice the workgroups of 64 x 128
    case 206: launch<32, 64, 1, 2, A_KC, B_KC, 4>(g); return true;    // 2 waves, for grids the 64 x 64 tile cannot fill
    default: return false;
  }
}

}  // namespace

// Second half of a split-K product: C = epilogue(alpha * sum_s partial[s] + beta * C), partials added in chunk order.
// Covers what the split path accepts: alpha / beta, bias, the element-wise clip, the fused SGD step on W and the second (activation)
// output -- per element the scalar epilogue's arithmetic (gemm_common.h), on the chunk-ordered sum instead of one accumulator.
__global__ void __launch_bounds__(kBlock) splitk_reduce_kernel(const float *__restrict__ part, int split, long stride, GemmArgs g) {
  if (g.pair && blockIdx.y == 1) {  // the second product's partials follow the first's
    part += (long)split * stride;
    g.C = g.C1; g.ep = g.ep1;
  }
  const long n = (long)g.M * g.N;
  // what a split-fp16 product leaves for the products that read its output (aslp_gemm_epilogue.planes / cmax_parts): the planes of the
  // activation output under a bound known before the launch, one maximum of |C| per workgroup
  const float pscale = (g.ep.planes_of == 2 && g.ep.planes.hi != nullptr && g.ep.act_out != nullptr) ? ldexpf(1.f, s16_exponent(*g.ep.planes.slot)) : 0.f;
  float cmax = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / g.N), col = (int)(i - (long)row * g.N);
    float acc = part[i];
    for (int s = 1; s < split; s++) acc += part[i + s * stride];
    float *cp = g.C + (long)row * g.ldc + col;
    const float c_old = g.beta != 0.0f ? (g.ep.c_src ? g.ep.c_src[(long)row * g.ep.ld_c_src + col] : *cp) : 0.0f;
    const float w_old = g.ep.W ? g.ep.W[(long)row * g.ep.ldw + col] : 0.0f;
    const float bias = g.ep.bias ? g.ep.bias[col] : 0.0f;
    float v = fmaf(g.alpha, acc, fmaf(g.beta, c_old, bias));  // the epilogue's spelling (gemm_common.h)
    if (g.ep.clip > 0.0f) v = fminf(fmaxf(v, -g.ep.clip), g.ep.clip);
    *cp = v;
    cmax = fmaxf(cmax, epi_finite_abs(v));
    if (g.ep.W) g.ep.W[(long)row * g.ep.ldw + col] = fmaf(g.ep.w_alpha, v, w_old);
    if (g.ep.act_out) {
      const float a = g.ep.act == 1 ? sigmoid_ref(v) : g.ep.act == 2 ? tanh_ref(v) : g.ep.act == 3 ? fmaxf(v, 0.0f) : v;
      g.ep.act_out[(long)row * g.ep.ld_act + col] = a;
      if (pscale != 0.f) epi_plane_store(g.ep, pscale, a, row, col);
    }
  }
  if (g.ep.cmax_parts != nullptr) {   // (uniform)
    __shared__ float wmax[kBlock / 64];
    cmax = wave_max(cmax);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = cmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = wmax[0];
      for (int w = 1; w < kBlock / 64; w++) m = fmaxf(m, wmax[w]);
      g.ep.cmax_parts[(long)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
  }
}

// Long reductions on a grid that cannot fill the chip (the recurrent weight gradients and d_r of the LSTM family: K = T*S
// or 4C against 30-130 output tiles) are bound by the length of one workgroup's K loop, not by the MFMA rate: 36 us whether
// the output is 256 x 512 or 2048 x 512.  Splitting K over blockIdx.y shortens that loop; the partial products are summed in
// chunk order by a second launch, so results do not depend on scheduling.  Returns 0 when the product should not be split.
int pick_split_k(const GemmArgs &g, int bm, int bn) {
  static const int forced = [] { const char *e = getenv("ASLP_GEMM_SPLITK"); return e ? atoi(e) : -1; }();
  if (forced == 0) return 0;
  const aslp_gemm_epilogue &ep = g.ep;
  if (ep.colsum || ep.colstats || g.K < 1024) return 0;
  if (g.pair && (g.ep1.colsum || g.ep1.colstats)) return 0;
  const long tiles = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * (g.pair ? 2 : 1);
  static const int slots = [] { const char *e = getenv("ASLP_GEMM_SPLITK_SLOTS"); return e ? atoi(e) : 256; }();
  int split = (int)(slots / tiles);
  if (split > g.K / 256) split = g.K / 256;
  if (split > 8) split = 8;
  if (forced > 0 && split >= 2) split = forced;
  return split >= 2 ? split : 0;
}

template <bool A_KC, bool B_KC>
bool launch_split(GemmArgs &g, int cfg, int split) {
  int chunk = ((g.K + split - 1) / split + BK - 1) / BK * BK;
  split = (g.K + chunk - 1) / chunk;
  if (split < 2) return false;
  const long stride = (long)g.M * g.N;
  const int np = g.pair ? 2 : 1;
  float *part = static_cast<float *>(scratch(kScratchSplitK, sizeof(float) * (size_t)stride * split * np));
  if (!part) return false;
  GemmArgs p = g;
  p.C = part; p.ldc = g.N; p.alpha = 1.0f; p.beta = 0.0f; p.ep = aslp_gemm_epilogue();
  p.C1 = part + (size_t)stride * split; p.ep1 = aslp_gemm_epilogue();
  p.split_k = split; p.k_chunk = chunk; p.split_stride = stride;
  if (!launch_cfg<A_KC, B_KC>(p, cfg)) return false;
  GemmArgs r = g;
  r.split_k = 0;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(stride), np), dim3(kBlock), 0, cur_stream(), part, split, stride, r);
  return true;
}

int gemm_splitk_reduce(const float *part, int split, long stride, const GemmArgs &r) {
  const int grid = grid_for(stride);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid, r.pair ? 2 : 1), dim3(kBlock), 0, cur_stream(), part, split, stride, r);
  return grid * (r.pair ? 2 : 1);   // workgroups = per-workgroup maxima written when r.ep.cmax_parts is set
}

bool gemm_glds_launch(GemmArgs &g, bool a_kc, bool b_kc, int cfg, int *cfg_used) {
  *cfg_used = cfg;
  // (a column-sum request on a non-transposed A is the caller's job: see aslp_sgemm_ex)
  if (g.K < 4 || g.K % 4 != 0 || !g.a_vec || !g.b_vec) return false;
  if (!a_kc && (g.M % 4 != 0 || g.M < 4)) return false;
  if (!b_kc && (g.N % 4 != 0 || g.N < 4)) return false;
  if (cfg == 207) {
    if (const int split = pick_split_k(g, 64, 64)) {
      bool ok = false;
      if (a_kc && b_kc) ok = launch_split<true, true>(g, cfg, split);
      else if (a_kc && !b_kc) ok = launch_split<true, false>(g, cfg, split);
      else if (!a_kc && !b_kc) ok = launch_split<false, false>(g, cfg, split);
      else ok = launch_split<false, true>(g, cfg, split);
      if (ok) return true;
    }
  }
  if (a_kc && b_kc) return launch_cfg<true, true>(g, cfg);
  if (a_kc && !b_kc) return launch_cfg<true, false>(g, cfg);
  if (!a_kc && !b_kc) return launch_cfg<false, false>(g, cfg);
  return launch_cfg<false, true>(g, cfg);
}

}  // namespace aslp
