// rnn_fused.hip -- one launch per timestep for the LSTM recurrence (both directions at once).
//
// The reference runs, per timestep and direction, a skinny cuBLAS GEMM (S x 4C x R), ~18 elementwise
// launches and a second skinny GEMM (S x R x C) (nnet-blstm-projected-streams-lc.h:571-609); with
// M = S = 32 streams those GEMMs use a handful of CUs and are pure latency.  Here the recurrence is
// re-associated so that only ONE product sits on the sequential path:
//     gates(t) = [x(t) W_x^T + b]  +  m(t-1) W_eff^T,      W_eff = W_r W_rm   (4C x C;  = W_r without projection)
// and the projection r = m W_rm^T is taken out of the loop (one batched GEMM over all t).  The step
// kernel fuses that product with the whole gate block: a workgroup owns 8 cells x 32 streams, i.e. a
// 32 x 32 MFMA tile whose columns are the (g,i,f,o) pre-activations of its 8 cells; its 4 waves split
// K = C, operands go global -> VGPR as 16-B loads (both are K-contiguous, no LDS staging), partial
// tiles are summed through LDS in wave order (deterministic) and the 256 threads then each finish one
// (stream, cell): peepholes, sigmoid/tanh, cell clip, c/h/m -- one launch, C/8 x S/32 x ndir workgroups.
// W_eff's row slices are pinned to a workgroup index, hence to one XCD's L2, across the T launches.
// Backward mirrors it:  d_m(t) = dm_ext(t) + dGATES(t+1) W_eff  (K = 4C, split over 4 workgroups x 4 waves,
// partials to scratch) followed by the fused gate-block backward, which adds the partials in a fixed order.
#include "aslp_kernels.h"
#include "common.h"
#include "scratch.h"
#include "rnn_mfma.h"

namespace aslp {
namespace {

constexpr int kCB = 8;    // cells per forward workgroup (x 4 gates = 32 MFMA columns)
constexpr int kKQ = 8;    // K-split of the backward product across workgroups (one round of 8 loads per wave at C = 512; 4: 7.85, 8: 7.42, 16: 8.17 ms per LC step)
constexpr int kFwdU = 8;   // K-chunks in flight in the forward product (16: 7.60 vs 7.45 ms per LC step)
constexpr int kNW = 4;    // waves per step workgroup (K split 4 ways; 8 waves were measured slower: 8.15 vs 7.85 ms on the LC step)

template <bool CIFG>
__global__ void __launch_bounds__(64 * kNW) lstm_step_fwd(aslp_lstm_step a) {
  __shared__ float red[kNW][32 * kPad];
  constexpr int G = CIFG ? 3 : 4;
  const aslp_lstm_step_dir D = a.dir[blockIdx.z];
  const int C = a.C, S = a.S, ld = a.ld;
  const int GC = G * C, oc = GC, oh = GC + C, om = GC + 2 * C;
  const int c0 = blockIdx.x * kCB, s0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  // epilogue operands of this thread's (stream, cell), requested before the product so their latency hides under it
  const int sl = threadIdx.x / kCB, cc = threadIdx.x % kCB;
  const int s = s0 + sl, c = c0 + cc;
  const bool live = threadIdx.x < 256 && s < S && c < C;  // the first 4 waves finish the 256 (stream, cell) pairs
  constexpr int gi = 1, gf = CIFG ? 1 : 2, go = CIFG ? 2 : 3;
  float *ys = D.y_cur + (long)(live ? s : 0) * ld;
  const int cq = live ? c : 0;
  const float xg = ys[cq], xf = ys[gf * C + cq], xo = ys[go * C + cq], xi = CIFG ? 0.f : ys[gi * C + cq];
  const float cprev = D.y_prev[(long)(live ? s : 0) * ld + oc + cq];
  const float pf = D.peep_f[cq], po = D.peep_o[cq], pi = CIFG ? 0.f : D.peep_i[cq];
  const bool masked = D.seq_lengths && D.t > D.seq_lengths[live ? s : 0];
  {
    const int srow = min(s0 + l31, S - 1);
    const int gate = l31 / kCB, cell = c0 + (l31 % kCB);
    const bool nvalid = gate < G && cell < C;
    const float *arow = D.y_prev + (long)srow * ld + om;                       // m(t-1) of stream srow
    const float *brow = D.w + (long)(nvalid ? gate * C + cell : 0) * a.ldw;    // W_eff row of (gate, cell)
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nch = (C + 7) / 8, per = (nch + kNW - 1) / kNW;
    if (!D.no_product) mfma_k_slices<kFwdU>(acc, arow, brow, C, wave * per, min(nch, (wave + 1) * per), h);
    store_tile(red[wave], acc, lane);
  }
  __syncthreads();
  if (!live) return;
  float pre[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const int n = g * kCB + cc;
    float acc = red[0][sl * kPad + n];
#pragma unroll
    for (int w = 1; w < kNW; w++) acc += red[w][sl * kPad + n];
    pre[g] = acc;
  }
  if (masked) {  // nnet-blstm-projected-streams.h:654-657
    ys[c] = 0.f; ys[gf * C + c] = 0.f; ys[go * C + c] = 0.f; ys[oc + c] = 0.f; ys[oh + c] = 0.f; ys[om + c] = 0.f;
    if (!CIFG) ys[gi * C + c] = 0.f;
    return;
  }
  const float g = tanh_ref(xg + pre[0]);
  const float f = sigmoid_ref(xf + pre[gf] + cprev * pf);
  float cell;
  if (!CIFG) {
    const float i = sigmoid_ref(xi + pre[gi] + cprev * pi);
    ys[gi * C + c] = i;
    cell = g * i + cprev * f;
  } else {
    cell = -g * f + g + cprev * f;
  }
  cell = fminf(fmaxf(cell, -50.0f), 50.0f);
  const float hh = tanh_ref(cell);
  const float o = sigmoid_ref(xo + pre[go] + cell * po);
  ys[c] = g; ys[gf * C + c] = f; ys[go * C + c] = o; ys[oc + c] = cell; ys[oh + c] = hh; ys[om + c] = hh * o;
}

// partial[dir][kq][s][c] = sum over this workgroup's K-quarter of dGATES(next)[s][k] * W_eff^T[c][k]
template <int G>
__global__ void __launch_bounds__(64 * kNW) lstm_step_bwd_gemm(aslp_lstm_step a, float *__restrict__ partial) {
  __shared__ float red[kNW][32 * kPad];
  const aslp_lstm_step_dir D = a.dir[blockIdx.z];
  const int C = a.C, S = a.S, ld = a.ld, GC = G * C;
  const int c0 = blockIdx.x * 32, kq = blockIdx.y % kKQ, s0 = (blockIdx.y / kKQ) * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  {
    const float *arow = D.d_next + (long)min(s0 + l31, S - 1) * ld;          // dGATES(next), columns [0, GC)
    const float *brow = D.w + (long)min(c0 + l31, C - 1) * a.ldw;            // W_eff^T row of cell c0 + l31
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nch = (GC + 7) / 8, per = (nch + kNW * kKQ - 1) / (kNW * kKQ), part = kq * kNW + wave;
    mfma_k_slices(acc, arow, brow, GC, part * per, min(nch, (part + 1) * per), h);
    store_tile(red[wave], acc, lane);
  }
  __syncthreads();
  float *out = partial + ((long)(blockIdx.z * kKQ + kq) * S) * C;
  for (int e = threadIdx.x; e < 32 * 32; e += 64 * kNW) {
    const int sl = e >> 5, n = e & 31;
    if (s0 + sl < S && c0 + n < C) {
      float acc = red[0][sl * kPad + n];
#pragma unroll
      for (int w = 1; w < kNW; w++) acc += red[w][sl * kPad + n];
      out[(long)(s0 + sl) * C + c0 + n] = acc;
    }
  }
}

// gate-block backward of step t for every direction; d_m = dm_ext (already in the m column) + the K-split partials
template <bool CIFG>
__global__ void __launch_bounds__(kBlock) lstm_step_bwd_cell(aslp_lstm_step a, const float *__restrict__ partial, int with_partial) {
  constexpr int G = CIFG ? 3 : 4;
  const aslp_lstm_step_dir D = a.dir[blockIdx.y];
  const int C = a.C, S = a.S, ld = a.ld;
  const int GC = G * C, oc = GC, oh = GC + C, om = GC + 2 * C;
  const int og = 0, oi = C, of = CIFG ? C : 2 * C, oo = CIFG ? 2 * C : 3 * C;
  const int n = S * C;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int s = idx / C, c = idx - s * C;
    const long o_ = (long)s * ld;
    // every load of this (stream, cell) first, stores last: the compiler cannot move a load above a store through a pointer
    // that might alias it, and a store in the middle would split the loads into two dependent rounds of memory latency
    float dm = D.d_cur[o_ + om + c];
    float psum[kKQ];
    if (with_partial) {
      const float *p = partial + ((long)blockIdx.y * kKQ * S + s) * C + c;
#pragma unroll
      for (int q = 0; q < kKQ; q++) psum[q] = p[(long)q * S * C];
    }
    const float yo = D.y_cur[o_ + oo + c], yh = D.y_cur[o_ + oh + c], yg = D.y_cur[o_ + og + c], yf = D.y_cur[o_ + of + c];
    const float yi = CIFG ? 0.f : D.y_cur[o_ + oi + c];
    const float dn_c = D.d_next[o_ + oc + c], yn_f = D.y_next[o_ + of + c], dn_f = D.d_next[o_ + of + c];
    const float dn_i = CIFG ? 0.f : D.d_next[o_ + oi + c];
    const float pi = CIFG ? 0.f : D.peep_i[c], pf = D.peep_f[c], po = D.peep_o[c];
    const float cprev = D.y_prev[o_ + oc + c];
    if (with_partial) {
#pragma unroll
      for (int q = 0; q < kKQ; q++) dm += psum[q];
      D.d_cur[o_ + om + c] = dm;  // the reference's d_m (lc.h:793) -- kept for InfoGradient-style dumps
    }
    const float dh = dtanh(yh, dm * yo);
    const float dov = dsigm(yo, dm * yh);
    float dc = dh + dn_c * yn_f;
    if (!CIFG) dc += dn_i * pi;
    dc += dn_f * pf;
    dc += dov * po;
    D.d_cur[o_ + oh + c] = dh;
    D.d_cur[o_ + oo + c] = dov;
    D.d_cur[o_ + oc + c] = dc;
    if (!CIFG) {
      D.d_cur[o_ + of + c] = dsigm(yf, dc * cprev);
      D.d_cur[o_ + oi + c] = dsigm(yi, dc * yg);
      D.d_cur[o_ + og + c] = dtanh(yg, dc * yi);
    } else {
      D.d_cur[o_ + of + c] = dsigm(yf, dc * cprev - dc * yg);
      D.d_cur[o_ + og + c] = dtanh(yg, dc - dc * yf);
    }
  }
}

bool step_args_ok(const aslp_lstm_step *a, const char *who) {
  if (!a || a->ndir < 1 || a->ndir > 2 || a->S <= 0 || a->C <= 0 || (a->C & 3) || (a->ld & 3) || (a->ldw & 3)) {
    set_error(std::string(who) + ": bad arguments (needs 1..2 directions, C, ld, ldw multiples of 4)");
    return false;
  }
  return true;
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

void aslp_lstm_step_forward(const aslp_lstm_step *a) {
  if (!step_args_ok(a, "aslp_lstm_step_forward")) return;
  dim3 grid((a->C + kCB - 1) / kCB, (a->S + 31) / 32, a->ndir);
  if (a->cifg) hipLaunchKernelGGL((lstm_step_fwd<true>), grid, dim3(64 * kNW), 0, cur_stream(), *a);
  else hipLaunchKernelGGL((lstm_step_fwd<false>), grid, dim3(64 * kNW), 0, cur_stream(), *a);
  check_launch("aslp_lstm_step_forward");
}

void aslp_lstm_step_backward(const aslp_lstm_step *a) {
  if (!step_args_ok(a, "aslp_lstm_step_backward")) return;
  int with_partial = 0;
  for (int d = 0; d < a->ndir; d++) with_partial |= a->dir[d].has_next;
  float *partial = nullptr;
  if (with_partial) {
    partial = static_cast<float *>(scratch(kScratchMisc, sizeof(float) * (size_t)a->ndir * kKQ * a->S * a->C));
    if (!partial) return;
    dim3 grid((a->C + 31) / 32, kKQ * ((a->S + 31) / 32), a->ndir);
    if (a->cifg) hipLaunchKernelGGL((lstm_step_bwd_gemm<3>), grid, dim3(64 * kNW), 0, cur_stream(), *a, partial);
    else hipLaunchKernelGGL((lstm_step_bwd_gemm<4>), grid, dim3(64 * kNW), 0, cur_stream(), *a, partial);
  }
  dim3 grid(grid_for((long)a->S * a->C), a->ndir);
  if (a->cifg) hipLaunchKernelGGL((lstm_step_bwd_cell<true>), grid, dim3(kBlock), 0, cur_stream(), *a, partial, with_partial);
  else hipLaunchKernelGGL((lstm_step_bwd_cell<false>), grid, dim3(kBlock), 0, cur_stream(), *a, partial, with_partial);
  check_launch("aslp_lstm_step_backward");
}

}  // extern "C"
