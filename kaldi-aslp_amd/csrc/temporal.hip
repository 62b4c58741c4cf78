// temporal.hip -- depthwise temporal filters for gfx950: RowConvolution and CompactFsmn.
//
// The reference builds both out of generic matrix ops: RowConvolution runs one D x D GEMM per
// frame and keeps only its diagonal (nnet-row-convolution.cc:128-133, D^2 (K+1) flops for D (K+1)
// useful ones) plus ~8 elementwise launches per frame in backward; CompactFsmn materialises a
// T*(P+F+1) x D product matrix and row-sums it (nnet-cfsmn-component.h:191-201).  Both are really
// per-column FIR filters along time -- HBM/L2-bound, no matrix core involved -- so here each is one
// direct kernel per pass: lanes run along the feature dimension (coalesced rows), every thread
// slides a register window over TT consecutive frames, and the tap gradients use a deterministic
// two-stage reduction (partials per frame chunk, then a fixed-order sum; no float atomics).
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

void aslp_fsmn_filter(float *out, int ldo, const float *src, int lds, const float *coef, int ldc, int D, int past, int future, int T,
                      int reverse) {
  if (T <= 0 || D <= 0) return;
  const int C = past + future + 1;
  dim3 block(kWave, kBlock / kWave), grid((D + kWave - 1) / kWave, (T + TT * (kBlock / kWave) - 1) / (TT * (kBlock / kWave)));
  hipLaunchKernelGGL(fsmn_filter, grid, block, 0, cur_stream(), out, ldo, src, lds, coef, ldc, D, C, reverse ? future : past, reverse, T);
  check_launch("aslp_fsmn_filter");
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

void aslp_rowconv_forward(float *out, int ldo, const float *in, int ldi, const float *w, int D, int K, int T, int S,
                          const int32_cuda *seq_len) {
  if (T <= 0 || D <= 0 || S <= 0) return;
  dim3 block(kWave, kBlock / kWave), grid((D + kWave - 1) / kWave, (T + TT * (kBlock / kWave) - 1) / (TT * (kBlock / kWave)), S);
  hipLaunchKernelGGL(rowconv_fwd, grid, block, 0, cur_stream(), out, ldo, in, ldi, w, D, K, T, S, seq_len);
  check_launch("aslp_rowconv_forward");
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
