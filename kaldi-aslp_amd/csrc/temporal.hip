// temporal.hip -- depthwise temporal filters for gfx950: RowConvolution and CompactFsmn.
//
// The reference builds both out of generic matrix ops: RowConvolution runs one D x D GEMM per
// frame and keeps only its diagonal (nnet-row-convolution.cc:128-133, D^2 (K+1) flops for D (K+1)
// useful ones) plus ~8 elementwise launches per frame in backward; CompactFsmn materialises a
// T*(P+F+1) x D product matrix and row-sums it (nnet-cfsmn-component.h:191-201).  Both are really
// per-column FIR filters along time -- HBM/L2-bound, no matrix core involved.
//
// RowConvolution (tensors of tens of MB): two STREAMING kernels.  A lane owns two adjacent feature columns of one stream and walks
// down time; every element is loaded once per chunk of 64 frames (+ a halo of K rows) and meets all K + 1 taps in registers
// (v_pk_fma_f32 on column pairs; at 21 taps x 2 flop per 4 bytes the vector pipe is as close to its limit as the HBM is).
// Forward keeps a ring of K + 1 running outputs; backward keeps a ring of the last K + 1 out-diff rows, which is all that BOTH the
// in-diff and the tap gradients need -- one pass over `in` and `out_diff`, one write of `in_diff`.  Workgroup ids are laid out so
// that the chunks of one strip of columns follow each other on ONE XCD (the halo rows are then L2 hits).  Tap gradients: four
// streams are summed in LDS, then one partial per (chunk, stream group); a small second launch adds them in a fixed order (no float
// atomics) and takes the momentum + SGD step.
//
// CompactFsmn (one utterance, T x D of a few MB: launch-latency territory): tiles of 64 columns x 32 frames staged in LDS.  Forward: one
// launch.  Backward: one launch forms the in-diff (the reversed filter) and the tap gradients of its frames, a small second one their
// fixed-order sum over the chunks, the clip and the SGD step.
//
// Shapes outside what these serve (more than 32 taps in RowConvolution, odd widths, filters too long for the LDS tile) run on the
// simpler kernels kept at the end of each section: lanes along the feature dimension, a register window over TT frames per thread,
// tap gradients by a two-stage reduction.
#include "aslp_kernels.h"
#include "common.h"
#include "scratch.h"

namespace aslp {
namespace {

constexpr int TT = 8;  // frames per thread

// ---- CompactFsmn ---------------------------------------------------------------------------------------
// out[t][d] = src[t][d] + sum_j coef[row(j)][d] * src[t + j - pad][d], rows outside [0,T) count as 0.
// forward: row(j) = j, pad = P.  in-diff: row(j) = C-1-j, pad = F (the reversed filter, cfsmn.h:232-246).
__global__ void __launch_bounds__(kBlock) fsmn_filter(float *__restrict__ out, int ldo, const float *__restrict__ src, int lds,
                                                      const float *__restrict__ coef, int ldc, int D, int C, int pad, int reverse, int T) {
  const int d = blockIdx.x * kWave + threadIdx.x;
  const int t0 = (blockIdx.y * (kBlock / kWave) + threadIdx.y) * TT;
  if (d >= D || t0 >= T) return;
  float acc[TT], win[TT];
#pragma unroll
  for (int u = 0; u < TT; u++) {
    acc[u] = 0.0f;
    const int r = t0 + u - pad;
    win[u] = (r >= 0 && r < T) ? src[(long)r * lds + d] : 0.0f;
  }
#pragma unroll 4
  for (int j = 0; j < C; j++) {
    const float c = coef[(long)(reverse ? C - 1 - j : j) * ldc + d];
#pragma unroll
    for (int u = 0; u < TT; u++) acc[u] += win[u] * c;
#pragma unroll
    for (int u = 0; u < TT - 1; u++) win[u] = win[u + 1];
    const int r = t0 + TT + j - pad;
    win[TT - 1] = (r >= 0 && r < T) ? src[(long)r * lds + d] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < TT; u++)
    if (t0 + u < T) out[(long)(t0 + u) * ldo + d] = src[(long)(t0 + u) * lds + d] + acc[u];
}

// stage 1 of the tap gradient: partial[chunk][i][d] = sum_{t in chunk} in[t + i - P][d] * od[t][d]
__global__ void __launch_bounds__(kBlock) fsmn_coef_grad1(float *__restrict__ partial, const float *__restrict__ in, int ldi,
                                                          const float *__restrict__ od, int ldod, int D, int C, int P, int T, int rpc) {
  const int d = blockIdx.x * kWave + threadIdx.x;
  if (d >= D) return;
  const int ta = blockIdx.y * rpc, tb = min(T, ta + rpc);
  for (int i = threadIdx.y; i < C; i += kBlock / kWave) {
    float acc = 0.0f;
    int lo = max(ta, P - i), hi = min(tb, T + P - i);  // t + i - P in [0, T)
#pragma unroll 8
    for (int t = lo; t < hi; t++) acc += in[(long)(t + i - P) * ldi + d] * od[(long)t * ldod + d];
    partial[((long)blockIdx.y * C + i) * D + d] = acc;
  }
}
__global__ void __launch_bounds__(kBlock) fsmn_coef_grad2(float *__restrict__ corr, int ldc, const float *__restrict__ partial, int D, int C,
                                                          int chunks, float clip) {
  const long n = (long)C * D;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int i = idx / D, d = idx - (long)i * D;
    float s = 0.0f;
#pragma unroll 8
    for (int c = 0; c < chunks; c++) s += partial[((long)c * C + i) * D + d];
    if (clip > 0.0f) s = fminf(fmaxf(s, -clip), clip);
    corr[(long)i * ldc + d] = s;  // beta 0: no momentum in the reference (cfsmn.h:219)
  }
}

// ---- RowConvolution ----------------------------------------------------------------------------------------
// rows are t*S + s; frames past the stream's length repeat its last frame (row-convolution.cc:118-126)
__global__ void __launch_bounds__(kBlock) rowconv_fwd(float *__restrict__ out, int ldo, const float *__restrict__ in, int ldi,
                                                      const float *__restrict__ w, int D, int K, int T, int S,
                                                      const int32_t *__restrict__ seq_len) {
  const int d = blockIdx.x * kWave + threadIdx.x;
  const int s = blockIdx.z;
  const int t0 = (blockIdx.y * (kBlock / kWave) + threadIdx.y) * TT;
  if (d >= D || t0 >= T) return;
  const int L = min(seq_len[s], T);
  float acc[TT], win[TT];
#pragma unroll
  for (int u = 0; u < TT; u++) {
    acc[u] = 0.0f;
    const int r = min(t0 + u, L - 1);
    win[u] = r >= 0 ? in[((long)r * S + s) * ldi + d] : 0.0f;
  }
  for (int k = 0; k <= K; k++) {
    const float c = w[(long)d * (K + 1) + k];
#pragma unroll
    for (int u = 0; u < TT; u++) acc[u] += c * win[u];
#pragma unroll
    for (int u = 0; u < TT - 1; u++) win[u] = win[u + 1];
    const int r = min(t0 + TT + k, L - 1);
    win[TT - 1] = r >= 0 ? in[((long)r * S + s) * ldi + d] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < TT; u++)
    if (t0 + u < T) out[((long)(t0 + u) * S + s) * ldo + d] = (t0 + u < L) ? acc[u] : 0.0f;  // frames past the end stay 0
}

// in_diff[t] = sum_{k <= min(K,t)} w[d][k] * od[t-k] for t < L, 0 beyond; the part of the diff that fell
// on the replicated tail frames is dropped, as in the reference (:168-174)
__global__ void __launch_bounds__(kBlock) rowconv_bwd(float *__restrict__ in_diff, int ldid, const float *__restrict__ od, int ldod,
                                                      const float *__restrict__ w, int D, int K, int T, int S,
                                                      const int32_t *__restrict__ seq_len) {
  const int d = blockIdx.x * kWave + threadIdx.x;
  const int s = blockIdx.z;
  const int t0 = (blockIdx.y * (kBlock / kWave) + threadIdx.y) * TT;
  if (d >= D || t0 >= T) return;
  const int L = min(seq_len[s], T);
  float acc[TT], win[TT];
#pragma unroll
  for (int u = 0; u < TT; u++) {
    acc[u] = 0.0f;
    const int r = t0 + u;
    win[u] = r < L ? od[((long)r * S + s) * ldod + d] : 0.0f;
  }
  for (int k = 0; k <= K; k++) {
    const float c = w[(long)d * (K + 1) + k];
#pragma unroll
    for (int u = 0; u < TT; u++) acc[u] += c * win[u];
#pragma unroll
    for (int u = TT - 1; u > 0; u--) win[u] = win[u - 1];
    const int r = t0 - 1 - k;
    win[0] = r >= 0 && r < L ? od[((long)r * S + s) * ldod + d] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < TT; u++)
    if (t0 + u < T) in_diff[((long)(t0 + u) * S + s) * ldid + d] = (t0 + u < L) ? acc[u] : 0.0f;
}

// partial[chunk][k][d] = sum over the streams s and the frames t of the chunk (t < L_s) of
//   in[min(t+k, L_s-1)][s][d] * od[t][s][d].
// Each lane owns a feature column and slides a register window of KP input frames along time, so every input and
// diff element is loaded once (the straightforward form re-reads them K+1 times: 977 us -> tens of us at T=800, S=32,
// D=512, K=20).  The 4 waves of a workgroup split the streams; their sums are combined through LDS in wave order.
template <int KP>
__global__ void __launch_bounds__(kBlock) rowconv_wgrad1(float *__restrict__ partial, const float *__restrict__ in, int ldi,
                                                         const float *__restrict__ od, int ldod, int D, int K, int T, int S,
                                                         const int32_t *__restrict__ seq_len, int tc, int spg, int k0) {
  // k0: first tap of this launch (taps k0 .. k0 + KP - 1; more than 64 taps go in groups of 64, one launch each)
  __shared__ float red[kBlock / kWave][KP][kWave];
  const int x = threadIdx.x, y = threadIdx.y;
  const int d = blockIdx.x * kWave + x;
  const int ta = blockIdx.y * tc, tb0 = min(T, ta + tc);
  float acc[KP];
#pragma unroll
  for (int k = 0; k < KP; k++) acc[k] = 0.0f;
  if (d < D) {
    const int s_end = min(S, ((int)blockIdx.z + 1) * spg);
    for (int s = blockIdx.z * spg + y; s < s_end; s += kBlock / kWave) {
      const int L = min(seq_len[s], T), tb = min(tb0, L);
      if (ta >= tb) continue;
      float w[KP];
#pragma unroll
      for (int k = 0; k < KP; k++) w[k] = in[((long)min(ta + k0 + k, L - 1) * S + s) * ldi + d];
      for (int t = ta; t < tb; t++) {
        const float g = od[((long)t * S + s) * ldod + d];
#pragma unroll
        for (int k = 0; k < KP; k++) acc[k] += w[k] * g;
#pragma unroll
        for (int k = 0; k < KP - 1; k++) w[k] = w[k + 1];
        w[KP - 1] = in[((long)min(t + k0 + KP, L - 1) * S + s) * ldi + d];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KP; k++) red[y][k][x] = acc[k];
  __syncthreads();
  if (d < D)
    for (int k = y; k < KP && k0 + k <= K; k += kBlock / kWave) {
      float sum = red[0][k][x];
#pragma unroll
      for (int j = 1; j < kBlock / kWave; j++) sum += red[j][k][x];
      partial[(((long)blockIdx.z * gridDim.y + blockIdx.y) * (K + 1) + k0 + k) * D + d] = sum;
    }
}
__global__ void __launch_bounds__(kBlock) rowconv_wgrad2(float *__restrict__ w_diff, const float *__restrict__ partial, int D, int K,
                                                         int chunks) {
  const long n = (long)D * (K + 1);
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const int k = idx / D, d = idx - (long)k * D;
    float s = 0.0f;
#pragma unroll 8
    for (int c = 0; c < chunks; c++) s += partial[((long)c * (K + 1) + k) * D + d];
    w_diff[(long)d * (K + 1) + k] = s;
  }
}

// ---- RowConvolution, streaming -----------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kRcCols = 2 * kWave;   // feature columns per workgroup (two per lane)
constexpr int kRcStreams = kBlock / kWave;   // streams per workgroup (one per wave)

// workgroup id -> (strip of columns x stream group, chunk of frames): the chunks of a strip are consecutive ids ON ONE XCD
// (ids go round-robin over the 8 XCDs)
struct RcPlace { int chunk, dtile, sgroup; bool valid; };
__device__ __forceinline__ RcPlace rc_place(int nchunks, int dtiles, int nstrips) {
  const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
  RcPlace p;
  p.chunk = j % nchunks;
  const int strip = (j / nchunks) * 8 + xcd;
  p.valid = strip < nstrips;
  p.dtile = strip % dtiles;
  p.sgroup = strip / dtiles;
  return p;
}

// out[t] = sum_{k <= K} w[:,k] in[min(t + k, L - 1)] for t < L, 0 beyond.  KP >= K + 1 is the ring length; EXACT: KP == K + 1 (no
// tap is tested against K).  Row r meets tap k for output t = r - k, which lives in ring slot (t - ta) mod KP: static indices once
// the row loop is unrolled by KP.  An output is taken out of the ring KP - 1 rows after its own row (taps beyond K are not applied).
template <int KP, bool EXACT>
__global__ void __launch_bounds__(kBlock) rowconv_fwd_stream(float *__restrict__ out, int ldo, const float *__restrict__ in, int ldi,
                                                             const float *__restrict__ w, int D, int K, int T, int S,
                                                             const int32_t *__restrict__ seq_len, int tc, int nchunks, int dtiles, int nstrips) {
  const RcPlace pl = rc_place(nchunks, dtiles, nstrips);
  if (!pl.valid) return;
  const int d0 = (pl.dtile * kWave + threadIdx.x) * 2, s = pl.sgroup * kRcStreams + threadIdx.y;
  if (d0 >= D || s >= S) return;
  const int ta = pl.chunk * tc, tb = min(T, ta + tc);
  const int L = min(seq_len[s], T), tv = min(tb, L);
  const long rs_in = (long)S * ldi, rs_out = (long)S * ldo;
  float *op = out + (long)s * ldo + d0;
  if (tv > ta) {
    const float *ip = in + (long)s * ldi + d0;
    f2 wr[KP], acc[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) {
      wr[k] = (EXACT || k <= K) ? f2{w[(long)d0 * (K + 1) + k], w[(long)(d0 + 1) * (K + 1) + k]} : f2{0.f, 0.f};
      acc[k] = f2{0.f, 0.f};
    }
    const int rend = tv + KP - 1;
    for (int r0 = ta; r0 < rend; r0 += KP) {
      f2 x[KP];
#pragma unroll
      for (int i = 0; i < KP; i++) x[i] = *reinterpret_cast<const f2 *>(ip + (long)min(r0 + i, L - 1) * rs_in);   // the last frame repeats
#pragma unroll
      for (int i = 0; i < KP; i++) {
#pragma unroll
        for (int k = 0; k < KP; k++)
          if (EXACT || k <= K) acc[(i - k + KP) % KP] += wr[k] * x[i];
        const int t = r0 + i - (KP - 1);
        if (t >= ta && t < tv) *reinterpret_cast<f2 *>(op + (long)t * rs_out) = acc[(i + 1) % KP];
        acc[(i + 1) % KP] = f2{0.f, 0.f};
      }
    }
  }
  for (int t = max(tv, ta); t < tb; t++) *reinterpret_cast<f2 *>(op + (long)t * rs_out) = f2{0.f, 0.f};   // frames past the end stay 0
}

// One pass over rows r of [0, T + K): with the ring o[k] = out_diff[r - k] (0 outside [0, L))
//   in_diff[r] = sum_k w[:,k] o[k]                      (r < L; 0 beyond)
//   g[:,k]    += in[min(r, L - 1)] o[k]                 (every pair (t = r - k < L, k) exactly once over the chunks)
// then the tap sums of the workgroup's four streams meet in LDS and leave as ONE partial [K + 1][D] per (chunk, stream group).
template <int KP, bool EXACT>
__global__ void __launch_bounds__(kBlock) rowconv_bwd_fused(float *__restrict__ in_diff, int ldid, float *__restrict__ partial,
                                                            const float *__restrict__ in, int ldi, const float *__restrict__ od, int ldod,
                                                            const float *__restrict__ w, int D, int K, int T, int S,
                                                            const int32_t *__restrict__ seq_len, int tc, int nchunks, int dtiles, int nstrips,
                                                            int sgroups) {
  __shared__ float red[kRcStreams - 1][KP][kRcCols];
  const RcPlace pl = rc_place(nchunks, dtiles, nstrips);
  if (!pl.valid) return;   // (the whole workgroup)
  const int x = threadIdx.x, y = threadIdx.y;
  const int d0 = (pl.dtile * kWave + x) * 2, s = pl.sgroup * kRcStreams + y;
  const bool live = d0 < D && s < S;
  f2 g[KP];
#pragma unroll
  for (int k = 0; k < KP; k++) g[k] = f2{0.f, 0.f};
  if (live) {
    const int ra = pl.chunk * tc, rb = ra + tc;
    const int L = min(seq_len[s], T), re = min(rb, L + K);
    const long rs_in = (long)S * ldi, rs_od = (long)S * ldod, rs_id = (long)S * ldid;
    float *dp = in_diff + (long)s * ldid + d0;
    if (re > ra && L > 0) {
      const float *ip = in + (long)s * ldi + d0, *gp = od + (long)s * ldod + d0;
      f2 wr[KP], o[KP];
#pragma unroll
      for (int k = 0; k < KP; k++) wr[k] = (EXACT || k <= K) ? f2{w[(long)d0 * (K + 1) + k], w[(long)(d0 + 1) * (K + 1) + k]} : f2{0.f, 0.f};
      // rows ra - (KP - 1) ... ra - 1 in slots 0 ... KP - 2; row r of the loop goes to slot (KP - 1 + i) % KP
#pragma unroll
      for (int j = 0; j < KP - 1; j++) {
        const int r = ra - (KP - 1) + j;
        o[j] = (r >= 0 && r < L) ? *reinterpret_cast<const f2 *>(gp + (long)r * rs_od) : f2{0.f, 0.f};
      }
      o[KP - 1] = f2{0.f, 0.f};
      for (int r0 = ra; r0 < re; r0 += KP) {
        f2 xi[KP], oi[KP];
#pragma unroll
        for (int i = 0; i < KP; i++) {
          const int r = min(r0 + i, L - 1);
          xi[i] = *reinterpret_cast<const f2 *>(ip + (long)r * rs_in);
          oi[i] = *reinterpret_cast<const f2 *>(gp + (long)r * rs_od);
        }
#pragma unroll
        for (int i = 0; i < KP; i++) {
          const int r = r0 + i;
          if (r < re) {   // (uniform over the wave: one stream per wave)
            o[(KP - 1 + i) % KP] = r < L ? oi[i] : f2{0.f, 0.f};
            f2 idf = f2{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KP; k++)
              if (EXACT || k <= K) {
                const f2 ok = o[(KP - 1 + i - k + KP) % KP];
                g[k] += xi[i] * ok;
                idf += wr[k] * ok;
              }
            if (r < L) *reinterpret_cast<f2 *>(dp + (long)r * rs_id) = idf;
          }
        }
      }
    }
    for (int r = max(ra, L); r < min(rb, T); r++) *reinterpret_cast<f2 *>(dp + (long)r * rs_id) = f2{0.f, 0.f};
  }
  if (y > 0) {
#pragma unroll
    for (int k = 0; k < KP; k++) { red[y - 1][k][2 * x] = g[k].x; red[y - 1][k][2 * x + 1] = g[k].y; }
  }
  __syncthreads();
  if (y == 0 && d0 < D) {
    float *pp = partial + ((long)pl.chunk * sgroups + pl.sgroup) * (K + 1) * D + d0;
#pragma unroll
    for (int k = 0; k < KP; k++)
      if (EXACT || k <= K) {
        f2 sum = g[k];
#pragma unroll
        for (int j = 0; j < kRcStreams - 1; j++) { sum.x += red[j][k][2 * x]; sum.y += red[j][k][2 * x + 1]; }
        *reinterpret_cast<f2 *>(pp + (long)k * D) = sum;
      }
  }
}
// w_diff[d][k] = sum of the partials (a fixed order: wave y adds parts y, y + 16, ..., then the sixteen sums are added in wave order); with
// `update` the step of nnet-row-convolution.cc:178-186 rides along: w_corr = momentum w_corr + w_diff, w -= lr w_corr.
// One workgroup per (tap, 64 columns): sixteen waves, so that every thread has a dozen loads in flight instead of two hundred in a row.
constexpr int kFinishWaves = 16;
__global__ void __launch_bounds__(kWave * kFinishWaves) rowconv_wgrad_finish(float *__restrict__ w_diff, const float *__restrict__ partial, int D, int K,
                                                                            int nparts, float *__restrict__ w_corr, float *__restrict__ w, float mmt,
                                                                            float lr, int update) {
  __shared__ float red[kFinishWaves][kWave];
  const int x = threadIdx.x, y = threadIdx.y, d = blockIdx.x * kWave + x, k = blockIdx.y;
  float s = 0.0f;
  if (d < D) {
#pragma unroll 16
    for (int c = y; c < nparts; c += kFinishWaves) s += partial[((long)c * (K + 1) + k) * D + d];
  }
  red[y][x] = s;
  __syncthreads();
  if (y != 0 || d >= D) return;
#pragma unroll
  for (int j = 1; j < kFinishWaves; j++) s += red[j][x];
  const long o = (long)d * (K + 1) + k;
  w_diff[o] = s;
  if (update) {
    const float c = mmt * w_corr[o] + s;
    w_corr[o] = c;
    w[o] += -lr * c;
  }
}

// ---- CompactFsmn, LDS tiles ---------------------------------------------------------------------------------
// One utterance is a few MB, so the passes are made of latencies and of vector instructions, not of bandwidth: 800 x 512 x 61 multiply-adds
// are 0.6 us of the chip's vector pipes but 6 us of 56 CUs'.  Hence tiles of 64 columns x 32 frames -- 200 workgroups of four waves for one
// utterance -- whose rows reach LDS in ONE round trip to memory (every load of a thread is issued before its first LDS write), windows kept as
// rings in registers (no moves), and tap gradients left as one partial per chunk of frames.
constexpr int kFsmnWaves = kBlock / kWave;   // 4
constexpr int kFsmnFrames = kFsmnWaves * TT; // frames per tile: one block of TT per wave
constexpr int kFsmnSpare = 3 * TT;           // rows behind every LDS region that read-ahead may touch (zeros or stale: never used)

// One tap against a ring window of N registers, with the ring position a COMPILE-TIME constant (a loop index inside `win[(u + jj) % N]` is
// turned into selects by the compiler unless every level of the nest is unrolled; spelled out, it cannot be):
//   acc[u] += win[(u + JJ) % N] * c for all u, then the slot of the oldest element takes the incoming one
template <int JJ, int N>
__device__ __forceinline__ void ring_tap(float (&acc)[N], float (&win)[N], float c, float e) {
#pragma unroll
  for (int u = 0; u < N; u++) acc[u] += win[(u + JJ) % N] * c;
  win[JJ] = e;
}
// taps j0 ... j0 + 7 of a block, operands from registers: a full block straight through, the last block of a count that is not a multiple
// of 8 tap by tap (those below `limit`)
__device__ __forceinline__ void ring_block8(float (&acc)[8], float (&win)[8], const float (&c)[8], const float (&e)[8], int j0, int limit) {
  if (j0 + 8 <= limit) {
    ring_tap<0, 8>(acc, win, c[0], e[0]); ring_tap<1, 8>(acc, win, c[1], e[1]); ring_tap<2, 8>(acc, win, c[2], e[2]); ring_tap<3, 8>(acc, win, c[3], e[3]);
    ring_tap<4, 8>(acc, win, c[4], e[4]); ring_tap<5, 8>(acc, win, c[5], e[5]); ring_tap<6, 8>(acc, win, c[6], e[6]); ring_tap<7, 8>(acc, win, c[7], e[7]);
    return;
  }
  if (j0 + 0 < limit) ring_tap<0, 8>(acc, win, c[0], e[0]);
  if (j0 + 1 < limit) ring_tap<1, 8>(acc, win, c[1], e[1]);
  if (j0 + 2 < limit) ring_tap<2, 8>(acc, win, c[2], e[2]);
  if (j0 + 3 < limit) ring_tap<3, 8>(acc, win, c[3], e[3]);
  if (j0 + 4 < limit) ring_tap<4, 8>(acc, win, c[4], e[4]);
  if (j0 + 5 < limit) ring_tap<5, 8>(acc, win, c[5], e[5]);
  if (j0 + 6 < limit) ring_tap<6, 8>(acc, win, c[6], e[6]);
  if (j0 + 7 < limit) ring_tap<7, 8>(acc, win, c[7], e[7]);
}
static_assert(TT == 8, "ring_block8 is written for windows of 8");

// rows [r0, r0 + n) of a [T x D] matrix into an LDS tile of 64 columns (zeros outside the matrix; `spare` further rows of zeros), and -- with
// taps != nullptr -- the C taps (reversed: tap j is coef row C - 1 - j) into theirs, by the whole workgroup.  Up to 26 rows and 16 taps per
// thread and round (104 rows, 64 taps: the 100 rows of a 30 + 30-tap tile in one round); 32-bit element offsets from the scalar bases, one address register per
// load -- the launchers check that they fit.
constexpr int kStageU = 26, kStageTapsU = 16;
__device__ __forceinline__ void stage_rows_and_taps(float *__restrict__ tile, const float *__restrict__ src, int ld, int r0, int n, int spare, int T,
                                                    float *__restrict__ taps, const float *__restrict__ coef, int ldc, int C, bool reversed, bool col_ok, int d,
                                                    int x, int y) {
  for (int i0 = y, j0 = y; i0 < n + spare || (taps != nullptr && j0 < C); i0 += kFsmnWaves * kStageU, j0 += kFsmnWaves * kStageTapsU) {
    float v[kStageU], c[kStageTapsU];
#pragma unroll
    for (int u = 0; u < kStageU; u++) {
      const int i = i0 + u * kFsmnWaves, r = r0 + i;
      v[u] = (col_ok && i < n && r >= 0 && r < T) ? src[(unsigned)(r * ld + d)] : 0.0f;
    }
    if (taps != nullptr) {
#pragma unroll
      for (int u = 0; u < kStageTapsU; u++) {
        const int j = j0 + u * kFsmnWaves;
        c[u] = (col_ok && j < C) ? coef[(unsigned)((reversed ? C - 1 - j : j) * ldc + d)] : 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < kStageU; u++) {
      const int i = i0 + u * kFsmnWaves;
      if (i < n + spare) tile[i * kWave + x] = v[u];
    }
    if (taps != nullptr) {
#pragma unroll
      for (int u = 0; u < kStageTapsU; u++) {
        const int j = j0 + u * kFsmnWaves;
        if (j < C) taps[j * kWave + x] = c[u];
      }
    }
  }
}
// out[t] = tile[t + pad] + sum_j taps[j] tile[t + j] for the wave's blocks of TT frames; `tile` row i holds source row ta - pad + i.  The
// window is a ring: tap j = j0 + jj reads elements j ... j + TT - 1, element e sits in slot e % TT, and the slot of element j takes j + TT.
__device__ __forceinline__ void filter_tile(float *__restrict__ out, int ldo, const float *__restrict__ tile, const float *__restrict__ taps, int C, int pad,
                                            int ta, int tb, int d, int x, int y) {
  for (int t0 = y * TT; t0 < tb - ta; t0 += kFsmnWaves * TT) {
    float acc[TT], win[TT];
#pragma unroll
    for (int u = 0; u < TT; u++) { acc[u] = 0.0f; win[u] = tile[(t0 + u) * kWave + x]; }
    // taps and incoming elements of a block of TT taps are read one block AHEAD (16 LDS reads in flight under 64 multiply-adds): with one
    // wave per SIMD nothing else hides an LDS read that is used the moment it is issued
    float cn[TT], en[TT];
#pragma unroll
    for (int jj = 0; jj < TT; jj++) { cn[jj] = taps[jj * kWave + x]; en[jj] = tile[(t0 + TT + jj) * kWave + x]; }
    for (int j0 = 0; j0 < C; j0 += TT) {
      float c[TT], e[TT];
#pragma unroll
      for (int jj = 0; jj < TT; jj++) { c[jj] = cn[jj]; e[jj] = en[jj]; }
      // (reads past tap C - 1 land in the regions' spare rows -- kFsmnSpare of them behind each -- and are never used)
#pragma unroll
      for (int jj = 0; jj < TT; jj++) {
        const int j = j0 + TT + jj;
        cn[jj] = taps[j * kWave + x];
        en[jj] = tile[(t0 + TT + j) * kWave + x];
      }
      ring_block8(acc, win, c, e, j0, C);
    }
#pragma unroll
    for (int u = 0; u < TT; u++)
      if (ta + t0 + u < tb) out[(long)(ta + t0 + u) * ldo + d] = tile[(t0 + u + pad) * kWave + x] + acc[u];
  }
}

// out[t] = src[t] + sum_j coef[row(j)] src[t + j - pad] on a tile of 64 columns x kFsmnFrames frames.
// sm: [kFsmnFrames + C - 1 + kFsmnSpare][64] source rows ta - pad ..., then [C + kFsmnSpare][64] taps in application order.
__global__ void __launch_bounds__(kBlock) fsmn_filter_lds(float *__restrict__ out, int ldo, const float *__restrict__ src, int lds_,
                                                          const float *__restrict__ coef, int ldc, int D, int C, int pad, int reverse, int T) {
  extern __shared__ float sm[];
  const int x = threadIdx.x, y = threadIdx.y, d = blockIdx.x * kWave + x;
  const int ta = blockIdx.y * kFsmnFrames, tb = min(T, ta + kFsmnFrames), rows = kFsmnFrames + C - 1;
  float *tile = sm, *taps = sm + (long)(rows + kFsmnSpare) * kWave;   // (taps: C + kFsmnSpare rows)
  stage_rows_and_taps(tile, src, lds_, ta - pad, rows, TT, T, taps, coef, ldc, C, reverse != 0, d < D, d, x, y);
  __syncthreads();
  if (d < D) filter_tile(out, ldo, tile, taps, C, pad, ta, tb, d, x, y);
}

constexpr int kFsmnTapsPerThread = 8;   // (= the ring length of ring_block8)
// Backward of a tile of 64 columns x kFsmnFrames frames [ta, tb):
//   in_diff[t] = od[t] + sum_j coef[C-1-j] od[t + j - F]                  (cfsmn.h:232-249)
//   partial[chunk][i] = sum_{t in chunk} in[t + i - P] od[t]               (:213-219)
// fsmn_grad_finish adds the chunks' partials in a launch of its own, spread over the chip.  (Adding them in the last workgroup of a column
// tile to arrive, behind a ticket, was measured: the device-scope release in front of the ticket costs every workgroup an L2 write-back --
// 48 us per launch with every wave fencing, 23 with one thread per workgroup, against 16 + 4.5 for the two launches.)
// sm: [rows + kFsmnSpare][64] od rows ta - F ..., [rows + kFsmnSpare][64] in rows ta - P ... (rows = kFsmnFrames + C - 1), [C + kFsmnSpare][64] reversed taps.
__global__ void __launch_bounds__(kBlock) fsmn_backward_fused(float *__restrict__ in_diff, int ldid, float *__restrict__ partial,
                                                              const float *__restrict__ coef, int ldc, const float *__restrict__ in, int ldi,
                                                              const float *__restrict__ od, int ldod, int D, int C, int P, int F, int T) {
  extern __shared__ float sm[];
  const int x = threadIdx.x, y = threadIdx.y, d = blockIdx.x * kWave + x;
  const int chunk = blockIdx.y, ta = chunk * kFsmnFrames, tb = min(T, ta + kFsmnFrames), rows = kFsmnFrames + C - 1;
  float *odt = sm, *int_ = odt + (long)(rows + kFsmnSpare) * kWave, *taps = int_ + (long)(rows + kFsmnSpare) * kWave;   // (taps: C + kFsmnSpare rows)
  const bool col_ok = d < D;
  // the in rows are needed behind the in-diff only: their loads (up to 26 per thread; more go the plain way below) leave with the od tile's
  // and reach LDS when the in-diff is done
  float vin[kStageU];
#pragma unroll
  for (int u = 0; u < kStageU; u++) {
    const int i = y + u * kFsmnWaves, r = ta - P + i;
    vin[u] = (col_ok && i < rows && r >= 0 && r < T) ? in[(unsigned)(r * ldi + d)] : 0.0f;
  }
  stage_rows_and_taps(odt, od, ldod, ta - F, rows, TT, T, taps, coef, ldc, C, true, col_ok, d, x, y);
  __syncthreads();
  if (col_ok) filter_tile(in_diff, ldid, odt, taps, C, F, ta, tb, d, x, y);
#pragma unroll
  for (int u = 0; u < kStageU; u++) {
    const int i = y + u * kFsmnWaves;
    if (i < rows) int_[i * kWave + x] = vin[u];
  }
  for (int i = y + kStageU * kFsmnWaves; i < rows; i += kFsmnWaves) {
    const int r = ta - P + i;
    int_[i * kWave + x] = (col_ok && r >= 0 && r < T) ? in[(unsigned)(r * ldi + d)] : 0.0f;
  }
  __syncthreads();
  if (col_ok) {
    // tap gradients of this chunk: 8 taps per thread, a ring of the in tile sliding down the chunk's frames (frame tt meets in rows
    // tt + ib ... tt + ib + 7: element e in slot e % 8)
    constexpr int G = kFsmnTapsPerThread;
    for (int ib = y * G; ib < C; ib += kFsmnWaves * G) {
      float acc[G], win[G];
#pragma unroll
      for (int j = 0; j < G; j++) { acc[j] = 0.0f; win[j] = int_[(ib + j) * kWave + x]; }
      const int nt = tb - ta;
      float gn[G], wn[G];   // read one block of frames ahead, as in filter_tile
#pragma unroll
      for (int tj = 0; tj < G; tj++) { gn[tj] = odt[(tj + F) * kWave + x]; wn[tj] = int_[(tj + ib + G) * kWave + x]; }
      for (int t0 = 0; t0 < nt; t0 += G) {
        float gs[G], wi[G];
#pragma unroll
        for (int tj = 0; tj < G; tj++) { gs[tj] = gn[tj]; wi[tj] = wn[tj]; }
#pragma unroll
        for (int tj = 0; tj < G; tj++) {   // (past the chunk: spare rows, never used)
          const int tt = t0 + G + tj;
          gn[tj] = odt[(tt + F) * kWave + x];
          wn[tj] = int_[(tt + ib + G) * kWave + x];
        }
        ring_block8(acc, win, gs, wi, t0, nt);
      }
#pragma unroll
      for (int j = 0; j < G; j++)
        if (ib + j < C) partial[((long)chunk * C + ib + j) * D + d] = acc[j];
    }
  }
}
// corr[i] = clip(sum of the chunks' partials, in chunk order) and, with lr != 0, coef[i] -= lr corr[i].  A workgroup of sixteen waves per
// 64 columns x 16 taps: up to 32 partials per thread go out at once.
constexpr int kFsmnFinishTaps = 16;
__global__ void __launch_bounds__(kWave * kFsmnFinishTaps) fsmn_grad_finish(float *__restrict__ corr, int ldcc, float *__restrict__ coef, int ldc,
                                                                          const float *__restrict__ partial, int D, int C, int nchunks, float clip, float lr) {
  const int d = blockIdx.x * kWave + threadIdx.x, i = blockIdx.y * kFsmnFinishTaps + threadIdx.y;
  if (d >= D || i >= C) return;
  const float w_old = lr != 0.0f ? coef[(long)i * ldc + d] : 0.0f;
  float s = 0.0f;
  for (int c0 = 0; c0 < nchunks; c0 += 32) {
    float v[32];
#pragma unroll
    for (int c = 0; c < 32; c++) v[c] = c0 + c < nchunks ? partial[(unsigned)(((c0 + c) * C + i) * D + d)] : 0.0f;
#pragma unroll
    for (int c = 0; c < 32; c++) s += v[c];
  }
  if (clip > 0.0f) s = fminf(fmaxf(s, -clip), clip);
  corr[(long)i * ldcc + d] = s;
  if (lr != 0.0f) coef[(long)i * ldc + d] = w_old + -lr * s;
}

// rows per chunk so that (column tiles x chunks) is a few hundred blocks
int rows_per_chunk(int rows, int ctiles) {
  int want = (512 + ctiles - 1) / ctiles;
  int rpc = (rows + want - 1) / want;
  return rpc < 16 ? 16 : rpc;
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

// LDS budget of the tiled CompactFsmn kernels (of the CU's 160 KB): rows of 64 floats
static constexpr int kFsmnLdsRows = 512;   // 128 KB
static bool fsmn_fits32(long rows, long ld) { return rows * ld < (1L << 30); }   // element offsets of the tiled kernels are 32-bit

void aslp_fsmn_filter(float *out, int ldo, const float *src, int lds, const float *coef, int ldc, int D, int past, int future, int T,
                      int reverse) {
  if (T <= 0 || D <= 0) return;
  const int C = past + future + 1, pad = reverse ? future : past;
  const int rows = kFsmnFrames + C - 1 + kFsmnSpare + C + kFsmnSpare;
  if (rows <= kFsmnLdsRows && fsmn_fits32(T, lds) && fsmn_fits32(T, ldo) && fsmn_fits32(C, ldc)) {
    const size_t lds_bytes = sizeof(float) * (size_t)rows * kWave;
    static bool attr_set = false;
    if (!attr_set) {
      ASLP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fsmn_filter_lds), hipFuncAttributeMaxDynamicSharedMemorySize, kFsmnLdsRows * kWave * sizeof(float)));
      attr_set = true;
    }
    hipLaunchKernelGGL(fsmn_filter_lds, dim3((D + kWave - 1) / kWave, (T + kFsmnFrames - 1) / kFsmnFrames), dim3(kWave, kFsmnWaves), lds_bytes, cur_stream(), out, ldo,
                       src, lds, coef, ldc, D, C, pad, reverse, T);
  } else {   // a filter too long for the tile
    dim3 block(kWave, kBlock / kWave), grid((D + kWave - 1) / kWave, (T + TT * (kBlock / kWave) - 1) / (TT * (kBlock / kWave)));
    hipLaunchKernelGGL(fsmn_filter, grid, block, 0, cur_stream(), out, ldo, src, lds, coef, ldc, D, C, pad, reverse, T);
  }
  check_launch("aslp_fsmn_filter");
}

void aslp_fsmn_backward(float *in_diff, int ldid, float *coef_corr, int ldcc, float *coef, int ldc, const float *in, int ldi, const float *out_diff,
                        int ldod, int D, int past, int future, int T, float clip, float lr) {
  if (T <= 0 || D <= 0) return;
  const int C = past + future + 1, ctiles = (D + kWave - 1) / kWave, rows = kFsmnFrames + C - 1;
  const int nchunks = (T + kFsmnFrames - 1) / kFsmnFrames;
  const int lds_rows = 2 * (rows + kFsmnSpare) + C + kFsmnSpare;
  if (lds_rows > kFsmnLdsRows || !fsmn_fits32(T, ldi) || !fsmn_fits32(T, ldod) || !fsmn_fits32(T, ldid) || !fsmn_fits32(C, ldc) ||
      !fsmn_fits32((long)nchunks * C, D)) {   // a filter too long for the tile (or offsets beyond 32 bits): the separate kernels
    aslp_fsmn_coef_grad(coef_corr, ldcc, in, ldi, out_diff, ldod, D, past, future, T, clip);
    aslp_fsmn_filter(in_diff, ldid, out_diff, ldod, coef, ldc, D, past, future, T, 1);
    if (lr != 0.0f) {   // coef += -lr coef_corr
      MatrixDim d = {C, D, ldc};
      cudaF_add_mat(aslp_dim3{1, 1, 1}, aslp_dim3{1, 1, 1}, -lr, coef_corr, coef, d, ldcc, 0);
    }
    return;
  }
  float *partial = static_cast<float *>(scratch(kScratchMisc, sizeof(float) * (size_t)nchunks * C * D));
  if (!partial) { set_error("aslp_fsmn_backward: no scratch for the tap-gradient partials -- nothing was launched (no in_diff, no gradient, no step)"); return; }
  static bool attr_set = false;
  if (!attr_set) {
    ASLP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fsmn_backward_fused), hipFuncAttributeMaxDynamicSharedMemorySize, kFsmnLdsRows * kWave * sizeof(float)));
    attr_set = true;
  }
  hipLaunchKernelGGL(fsmn_backward_fused, dim3(ctiles, nchunks), dim3(kWave, kFsmnWaves), sizeof(float) * (size_t)lds_rows * kWave, cur_stream(), in_diff, ldid,
                     partial, coef, ldc, in, ldi, out_diff, ldod, D, C, past, future, T);
  hipLaunchKernelGGL(fsmn_grad_finish, dim3(ctiles, (C + kFsmnFinishTaps - 1) / kFsmnFinishTaps), dim3(kWave, kFsmnFinishTaps), 0, cur_stream(), coef_corr, ldcc,
                     coef, ldc, partial, D, C, nchunks, clip, lr);
  check_launch("aslp_fsmn_backward");
}

void aslp_fsmn_coef_grad(float *coef_corr, int ldc, const float *in, int ldi, const float *out_diff, int ldod, int D, int past, int future,
                         int T, float clip) {
  if (T <= 0 || D <= 0) return;
  const int C = past + future + 1, ctiles = (D + kWave - 1) / kWave;
  const int rpc = rows_per_chunk(T, ctiles), chunks = (T + rpc - 1) / rpc;
  float *partial = static_cast<float *>(scratch(kScratchMisc, sizeof(float) * (size_t)chunks * C * D));
  if (!partial) return;
  hipLaunchKernelGGL(fsmn_coef_grad1, dim3(ctiles, chunks), dim3(kWave, kBlock / kWave), 0, cur_stream(), partial, in, ldi, out_diff, ldod,
                     D, C, past, T, rpc);
  hipLaunchKernelGGL(fsmn_coef_grad2, dim3(grid_for((long)C * D)), dim3(kBlock), 0, cur_stream(), coef_corr, ldc, partial, D, C, chunks,
                     clip);
  check_launch("aslp_fsmn_coef_grad");
}

// the streaming RowConvolution kernels serve up to 32 taps on even widths with 8-byte aligned rows
static bool rowconv_stream_serves(int D, int K, int ld_a, int ld_b, const void *a, const void *b) {
  return K + 1 <= 32 && !(D & 1) && !(ld_a & 1) && !(ld_b & 1) && !(reinterpret_cast<uintptr_t>(a) & 7u) && !(reinterpret_cast<uintptr_t>(b) & 7u);
}
static constexpr int kRcChunk = 64;   // frames per chunk: the K halo rows cost (64 + K) / 64 of the reads

void aslp_rowconv_forward(float *out, int ldo, const float *in, int ldi, const float *w, int D, int K, int T, int S,
                          const int32_cuda *seq_len) {
  if (T <= 0 || D <= 0 || S <= 0) return;
  if (rowconv_stream_serves(D, K, ldo, ldi, out, in)) {
    const int dtiles = (D + kRcCols - 1) / kRcCols, sgroups = (S + kRcStreams - 1) / kRcStreams, nstrips = dtiles * sgroups;
    const int tc = kRcChunk, nchunks = (T + tc - 1) / tc;
    const dim3 grid(8 * ((nstrips + 7) / 8) * nchunks), block(kWave, kRcStreams);
#define ASLP_RC_FWD(KP, EXACT) hipLaunchKernelGGL((rowconv_fwd_stream<KP, EXACT>), grid, block, 0, cur_stream(), out, ldo, in, ldi, w, D, K, T, S, seq_len, tc, nchunks, dtiles, nstrips)
    if (K + 1 == 21) ASLP_RC_FWD(21, true);        // FutureContext 20 (the recipes' value): no tap is tested
    else if (K + 1 <= 4) ASLP_RC_FWD(4, false);
    else if (K + 1 <= 8) ASLP_RC_FWD(8, false);
    else if (K + 1 <= 16) ASLP_RC_FWD(16, false);
    else ASLP_RC_FWD(32, false);
#undef ASLP_RC_FWD
    check_launch("aslp_rowconv_forward");
    return;
  }
  dim3 block(kWave, kBlock / kWave), grid((D + kWave - 1) / kWave, (T + TT * (kBlock / kWave) - 1) / (TT * (kBlock / kWave)), S);
  hipLaunchKernelGGL(rowconv_fwd, grid, block, 0, cur_stream(), out, ldo, in, ldi, w, D, K, T, S, seq_len);
  check_launch("aslp_rowconv_forward");
}

void aslp_rowconv_backward_fused(float *in_diff, int ldid, float *w_diff, const float *in, int ldi, const float *out_diff, int ldod, float *w, int D,
                                 int K, int T, int S, const int32_cuda *seq_len, float *w_corr, float momentum, float learn_rate, int update) {
  if (T <= 0 || D <= 0 || S <= 0) return;
  if (update && !w_corr) { set_error("aslp_rowconv_backward_fused: update without w_corr"); return; }
  if (!(rowconv_stream_serves(D, K, ldid, ldi, in_diff, in) && !(ldod & 1) && !(reinterpret_cast<uintptr_t>(out_diff) & 7u))) {
    aslp_rowconv_backward(in_diff, ldid, out_diff, ldod, w, D, K, T, S, seq_len);
    aslp_rowconv_wgrad(w_diff, in, ldi, out_diff, ldod, D, K, T, S, seq_len);
    if (update) {   // nnet-row-convolution.cc:178-186
      const int n = D * (K + 1);
      cudaF_scale(aslp_dim3{1, 1, 1}, aslp_dim3{1, 1, 1}, w_corr, momentum, MatrixDim{1, n, n});
      aslp_vec_axpy(1.0f, w_diff, w_corr, n);
      aslp_vec_axpy(-learn_rate, w_corr, w, n);
    }
    return;
  }
  const int dtiles = (D + kRcCols - 1) / kRcCols, sgroups = (S + kRcStreams - 1) / kRcStreams, nstrips = dtiles * sgroups;
  const int tc = kRcChunk, nchunks = (T + K + tc - 1) / tc, nparts = nchunks * sgroups;
  float *partial = static_cast<float *>(scratch(kScratchMisc, sizeof(float) * (size_t)nparts * (K + 1) * D));
  if (!partial) { set_error("aslp_rowconv_backward_fused: no scratch for the tap-gradient partials -- nothing was launched (no in_diff, no gradient, no step)"); return; }
  const dim3 grid(8 * ((nstrips + 7) / 8) * nchunks), block(kWave, kRcStreams);
#define ASLP_RC_BWD(KP, EXACT) hipLaunchKernelGGL((rowconv_bwd_fused<KP, EXACT>), grid, block, 0, cur_stream(), in_diff, ldid, partial, in, ldi, out_diff, ldod, w, D, K, T, S, seq_len, tc, nchunks, dtiles, nstrips, sgroups)
  if (K + 1 == 21) ASLP_RC_BWD(21, true);
  else if (K + 1 <= 4) ASLP_RC_BWD(4, false);
  else if (K + 1 <= 8) ASLP_RC_BWD(8, false);
  else if (K + 1 <= 16) ASLP_RC_BWD(16, false);
  else ASLP_RC_BWD(32, false);
#undef ASLP_RC_BWD
  hipLaunchKernelGGL(rowconv_wgrad_finish, dim3((D + kWave - 1) / kWave, K + 1), dim3(kWave, kFinishWaves), 0, cur_stream(), w_diff, partial, D, K, nparts,
                     w_corr, w, momentum, learn_rate, update);
  check_launch("aslp_rowconv_backward_fused");
}

void aslp_rowconv_backward(float *in_diff, int ldid, const float *out_diff, int ldod, const float *w, int D, int K, int T, int S,
                           const int32_cuda *seq_len) {
  if (T <= 0 || D <= 0 || S <= 0) return;
  dim3 block(kWave, kBlock / kWave), grid((D + kWave - 1) / kWave, (T + TT * (kBlock / kWave) - 1) / (TT * (kBlock / kWave)), S);
  hipLaunchKernelGGL(rowconv_bwd, grid, block, 0, cur_stream(), in_diff, ldid, out_diff, ldod, w, D, K, T, S, seq_len);
  check_launch("aslp_rowconv_backward");
}

void aslp_rowconv_wgrad(float *w_diff, const float *in, int ldi, const float *out_diff, int ldod, int D, int K, int T, int S,
                        const int32_cuda *seq_len) {
  if (T <= 0 || D <= 0 || S <= 0) return;
  const int ctiles = (D + kWave - 1) / kWave;
  // one partial per (frame chunk, stream group): chunks of two window lengths keep the window fill at a third of the
  // loads; stream groups supply the rest of the ~1000 workgroups the chip wants
  const int kp = K + 1 <= 8 ? 8 : K + 1 <= 16 ? 16 : K + 1 <= 32 ? 32 : 64;
  const int tc = 2 * kp, chunks = (T + tc - 1) / tc, waves = kBlock / kWave;
  int sg = 1024 / (ctiles * chunks);
  sg = sg < 1 ? 1 : sg > (S + waves - 1) / waves ? (S + waves - 1) / waves : sg;
  const int spg = (S + sg - 1) / sg;
  sg = (S + spg - 1) / spg;
  float *partial = static_cast<float *>(scratch(kScratchMisc, sizeof(float) * (size_t)chunks * sg * (K + 1) * D));
  if (!partial) return;
  dim3 grid(ctiles, chunks, sg), block(kWave, waves);
  if (K + 1 <= 8) hipLaunchKernelGGL((rowconv_wgrad1<8>), grid, block, 0, cur_stream(), partial, in, ldi, out_diff, ldod, D, K, T, S, seq_len, tc, spg, 0);
  else if (K + 1 <= 16) hipLaunchKernelGGL((rowconv_wgrad1<16>), grid, block, 0, cur_stream(), partial, in, ldi, out_diff, ldod, D, K, T, S, seq_len, tc, spg, 0);
  else if (K + 1 <= 32) hipLaunchKernelGGL((rowconv_wgrad1<32>), grid, block, 0, cur_stream(), partial, in, ldi, out_diff, ldod, D, K, T, S, seq_len, tc, spg, 0);
  else   // 64 taps per launch: a register window longer than that spills; the reference has no limit on FutureContext
    for (int k0 = 0; k0 <= K; k0 += 64)
      hipLaunchKernelGGL((rowconv_wgrad1<64>), grid, block, 0, cur_stream(), partial, in, ldi, out_diff, ldod, D, K, T, S, seq_len, tc, spg, k0);
  hipLaunchKernelGGL(rowconv_wgrad2, dim3(grid_for((long)D * (K + 1))), dim3(kBlock), 0, cur_stream(), w_diff, partial, D, K, chunks * sg);
  check_launch("aslp_rowconv_wgrad");
}

}  // extern "C"
