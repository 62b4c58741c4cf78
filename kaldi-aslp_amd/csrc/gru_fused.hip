// gru_fused.hip -- the GRU recurrence (GruStreams, nnet-gru-streams.h:238-430) as four launches per timestep.
//
// Inside a timestep the GRU has two products that depend on each other in each direction of time:
//   forward   zr(t) += h(t-1) W_zr_h^T  -> z, r = sigmoid, g = r .* h(t-1)   then   m(t) += g(t) W_m_g^T -> m = tanh, h(t)
//   backward  d_h(t) += DZR(t+1) W_zr_h -> d_h, d_m                           then   d_g(t) = d_m(t) W_m_g -> d_r, d_z
// The reference issues each as a skinny cuBLAS GEMM (M = S streams) plus several elementwise launches.  Here each product
// is fused with the gate arithmetic that consumes it: a workgroup owns a 32 x 32 tile (32 streams x 32 gate columns), its 4
// waves split K, operands come straight from global memory as 16-byte loads (both K-contiguous: the backward kernels read
// transposed copies of the two recurrent matrices, refreshed once per Backpropagate), partial tiles meet in LDS in wave
// order (deterministic), and the epilogue finishes every (stream, cell) of the tile.
#include "aslp_kernels.h"
#include "common.h"
#include "rnn_mfma.h"

namespace aslp {
namespace {

template <int NW = 4>
__device__ __forceinline__ float tile_sum(const float (*red)[32 * kPad], int row, int col) {
  float acc = red[0][row * kPad + col];
#pragma unroll
  for (int w = 1; w < NW; w++) acc += red[w][row * kPad + col];
  return acc;
}
constexpr int kBwd1Waves = 4;  // K = 2H; 8 waves (two rounds of loads instead of four) measured no faster

// columns of the tile: n < 16 -> z of cell c0 + n, n >= 16 -> r of cell c0 + n - 16
__global__ void __launch_bounds__(256) gru_step_fwd1(float *__restrict__ y, const float *__restrict__ yp, const float *__restrict__ w_zr_h, int ldw,
                                                     int ld, int S, int H, int no_product) {
  __shared__ float red[4][32 * kPad];
  const int c0 = blockIdx.x * 16, s0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  // epilogue operands, requested before the product so their latency hides under it
  // all loads of this thread's (stream, cell) pairs first, stores last: a store in between would fence the later loads behind
  // it (possible aliasing), i.e. one more round of memory latency per pair
  float xz[2], xr[2], hp[2];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int idx = threadIdx.x + 256 * p, sl = idx >> 4, cc = idx & 15;
    const int s = min(s0 + sl, S - 1), c = min(c0 + cc, H - 1);
    xz[p] = y[(long)s * ld + c]; xr[p] = y[(long)s * ld + H + c]; hp[p] = yp[(long)s * ld + 4 * H + c];
  }
  {
    const int gate = l31 >> 4, cell = min(c0 + (l31 & 15), H - 1);
    const float *arow = yp + (long)min(s0 + l31, S - 1) * ld + 4 * H;  // h(t-1)
    const float *brow = w_zr_h + (long)(gate * H + cell) * ldw;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nch = (H + 7) / 8, per = (nch + 3) / 4;
    if (!no_product) mfma_k_slices(acc, arow, brow, H, wave * per, min(nch, (wave + 1) * per), h);
    store_tile(red[wave], acc, lane);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int idx = threadIdx.x + 256 * p, sl = idx >> 4, cc = idx & 15;
    const int s = s0 + sl, c = c0 + cc;
    if (s >= S || c >= H) continue;
    float *ys = y + (long)s * ld;
    const float z = sigmoid_ref(xz[p] + tile_sum(red, sl, cc)), r = sigmoid_ref(xr[p] + tile_sum(red, sl, 16 + cc));
    ys[c] = z;
    ys[H + c] = r;
    ys[3 * H + c] = r * hp[p];
  }
}

__global__ void __launch_bounds__(256) gru_step_fwd2(float *__restrict__ y, const float *__restrict__ yp, const float *__restrict__ w_m_g, int ldw,
                                                     int ld, int S, int H) {
  __shared__ float red[4][32 * kPad];
  const int c0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  // epilogue operands, requested before the product so their latency hides under it
  float hp[4], zz[4], xm[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {  // loads first (see gru_step_fwd1)
    const int idx = threadIdx.x + 256 * p, sl = idx >> 5, cc = idx & 31;
    const int s = min(s0 + sl, S - 1), c = min(c0 + cc, H - 1);
    hp[p] = yp[(long)s * ld + 4 * H + c]; zz[p] = y[(long)s * ld + c]; xm[p] = y[(long)s * ld + 2 * H + c];
  }
  {
    const float *arow = y + (long)min(s0 + l31, S - 1) * ld + 3 * H;  // g(t)
    const float *brow = w_m_g + (long)min(c0 + l31, H - 1) * ldw;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nch = (H + 7) / 8, per = (nch + 3) / 4;
    mfma_k_slices(acc, arow, brow, H, wave * per, min(nch, (wave + 1) * per), h);
    store_tile(red[wave], acc, lane);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int idx = threadIdx.x + 256 * p, sl = idx >> 5, cc = idx & 31;
    const int s = s0 + sl, c = c0 + cc;
    if (s >= S || c >= H) continue;
    float *ys = y + (long)s * ld;
    const float m = tanh_ref(xm[p] + tile_sum(red, sl, cc));
    ys[2 * H + c] = m;
    ys[4 * H + c] = hp[p] - hp[p] * zz[p] + zz[p] * m;
  }
}

// d_h(t) += DZR(t+1) W_zr_h (K = 2H, w_t = W_zr_h^T [H x 2H]); then d_h, d_m (gru_bwd1's arithmetic)
__global__ void __launch_bounds__(64 * kBwd1Waves) gru_step_bwd1(float *__restrict__ d, const float *__restrict__ dn, const float *__restrict__ y,
                                                     const float *__restrict__ yn, const float *__restrict__ w_t, int ldwt, int ld, int S, int H,
                                                     int has_next) {
  __shared__ float red[kBwd1Waves][32 * kPad];
  const int c0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  // epilogue operands, requested before the product so their latency hides under it
  constexpr int NP = 1024 / (64 * kBwd1Waves);
  float v_dhn[NP], v_dh[NP], v_zn[NP], v_dgn[NP], v_rn[NP], v_m[NP], v_z[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) {  // loads first (see gru_step_fwd1)
    const int idx = threadIdx.x + 64 * kBwd1Waves * p, sl = idx >> 5, cc = idx & 31;
    const long o = (long)min(s0 + sl, S - 1) * ld;
    const int c = min(c0 + cc, H - 1);
    v_dhn[p] = dn[o + 4 * H + c]; v_dh[p] = d[o + 4 * H + c]; v_zn[p] = yn[o + c]; v_dgn[p] = dn[o + 3 * H + c]; v_rn[p] = yn[o + H + c];
    v_m[p] = y[o + 2 * H + c]; v_z[p] = y[o + c];
  }
  {
    const float *arow = dn + (long)min(s0 + l31, S - 1) * ld;  // [d_z | d_r](t+1)
    const float *brow = w_t + (long)min(c0 + l31, H - 1) * ldwt;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nch = (2 * H + 7) / 8, per = (nch + kBwd1Waves - 1) / kBwd1Waves;
    if (has_next) mfma_k_slices(acc, arow, brow, 2 * H, wave * per, min(nch, (wave + 1) * per), h);
    store_tile(red[wave], acc, lane);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const int idx = threadIdx.x + 64 * kBwd1Waves * p, sl = idx >> 5, cc = idx & 31;
    const int s = s0 + sl, c = c0 + cc;
    if (s >= S || c >= H) continue;
    const long o = (long)s * ld;
    const float dh = v_dh[p] + tile_sum<kBwd1Waves>(red, sl, cc) + v_dhn[p] - v_dhn[p] * v_zn[p] + v_dgn[p] * v_rn[p];
    d[o + 4 * H + c] = dh;
    d[o + 2 * H + c] = dtanh(v_m[p], dh * v_z[p]);
  }
}

// d_g(t) = d_m(t) W_m_g (K = H, w_t = W_m_g^T); then d_r, d_z (gru_bwd2's arithmetic)
__global__ void __launch_bounds__(256) gru_step_bwd2(float *__restrict__ d, const float *__restrict__ y, const float *__restrict__ yp,
                                                     const float *__restrict__ w_t, int ldwt, int ld, int S, int H) {
  __shared__ float red[4][32 * kPad];
  const int c0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  // epilogue operands, requested before the product so their latency hides under it
  float v_hp[4], v_dh[4], v_r[4], v_z[4], v_m[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {  // loads first (see gru_step_fwd1)
    const int idx = threadIdx.x + 256 * p, sl = idx >> 5, cc = idx & 31;
    const long o = (long)min(s0 + sl, S - 1) * ld;
    const int c = min(c0 + cc, H - 1);
    v_hp[p] = yp[o + 4 * H + c]; v_dh[p] = d[o + 4 * H + c]; v_r[p] = y[o + H + c]; v_z[p] = y[o + c]; v_m[p] = y[o + 2 * H + c];
  }
  {
    const float *arow = d + (long)min(s0 + l31, S - 1) * ld + 2 * H;  // d_m(t)
    const float *brow = w_t + (long)min(c0 + l31, H - 1) * ldwt;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nch = (H + 7) / 8, per = (nch + 3) / 4;
    mfma_k_slices(acc, arow, brow, H, wave * per, min(nch, (wave + 1) * per), h);
    store_tile(red[wave], acc, lane);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int idx = threadIdx.x + 256 * p, sl = idx >> 5, cc = idx & 31;
    const int s = s0 + sl, c = c0 + cc;
    if (s >= S || c >= H) continue;
    const long o = (long)s * ld;
    const float dg = tile_sum(red, sl, cc);
    d[o + 3 * H + c] = dg;
    d[o + H + c] = dsigm(v_r[p], dg * v_hp[p]);
    d[o + c] = dsigm(v_z[p], v_dh[p] * v_m[p] - v_dh[p] * v_hp[p]);
  }
}

bool gru_args_ok(const void *p0, const void *p1, const void *p2, int ld, int ldw, int H) {
  return H % 4 == 0 && ld % 4 == 0 && ldw % 4 == 0 && aligned16(p0) && aligned16(p1) && aligned16(p2);
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

int aslp_gru_step_supported(int H) { return H % 4 == 0; }

void aslp_gru_step_forward(float *y_cur, const float *y_prev, const float *w_zr_h, int ld_zr, const float *w_m_g, int ld_mg, int ld, int S, int H) {
  if (S <= 0 || H <= 0) return;
  if (!gru_args_ok(y_cur, y_prev, w_zr_h, ld, ld_zr, H) || ld_mg % 4 != 0 || !aligned16(w_m_g)) { set_error("aslp_gru_step_forward: unaligned operands"); return; }
  const int sb = (S + 31) / 32;
  hipLaunchKernelGGL(gru_step_fwd1, dim3((H + 15) / 16, sb), dim3(256), 0, cur_stream(), y_cur, y_prev, w_zr_h, ld_zr, ld, S, H, 0);
  hipLaunchKernelGGL(gru_step_fwd2, dim3((H + 31) / 32, sb), dim3(256), 0, cur_stream(), y_cur, y_prev, w_m_g, ld_mg, ld, S, H);
  check_launch("gru_step_forward");
}

void aslp_gru_step_backward(float *d_cur, const float *d_next, const float *y_cur, const float *y_next, const float *y_prev, const float *w_zr_h_t,
                            int ld_zr_t, const float *w_m_g_t, int ld_mg_t, int ld, int S, int H, int has_next) {
  if (S <= 0 || H <= 0) return;
  if (!gru_args_ok(d_cur, d_next, w_zr_h_t, ld, ld_zr_t, H) || ld_mg_t % 4 != 0 || !aligned16(w_m_g_t)) { set_error("aslp_gru_step_backward: unaligned operands"); return; }
  const int sb = (S + 31) / 32;
  hipLaunchKernelGGL(gru_step_bwd1, dim3((H + 31) / 32, sb), dim3(64 * kBwd1Waves), 0, cur_stream(), d_cur, d_next, y_cur, y_next, w_zr_h_t, ld_zr_t, ld, S, H,
                     has_next);
  hipLaunchKernelGGL(gru_step_bwd2, dim3((H + 31) / 32, sb), dim3(256), 0, cur_stream(), d_cur, y_cur, y_prev, w_m_g_t, ld_mg_t, ld, S, H);
  check_launch("gru_step_backward");
}

}  // extern "C"
