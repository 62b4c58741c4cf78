// rnn_cells.hip -- fused per-timestep gate blocks of the LSTM family and GruStreams for gfx950.
//
// The reference runs each timestep as ~18 (forward) / ~22 (backward) separate elementwise
// CuMatrix launches on S x C tiles (nnet-blstm-projected-streams-lc.h:571-609, 783-835); here one
// launch per timestep does the whole gate block: every thread owns one (stream, cell) pair, reads
// its gate pre-activations / neighbours once and writes all derived values once (HBM/L2-bound,
// ~11 floats read + 7 written per cell forward).  Columns of the activation buffer follow the
// reference: [g | i | f | o | c | h | m | r] (CIFG: [g | f | o | c | h | m | r]), GRU [z | r | m | g | h].
#include "aslp_kernels.h"
#include "common.h"

namespace aslp {
namespace {

__device__ __forceinline__ float dsigm(float y, float d) { return d * y * (1.0f - y); }
__device__ __forceinline__ float dtanh(float y, float d) { return d * (1.0f - y * y); }

template <bool CIFG>
__global__ void __launch_bounds__(kBlock) lstm_cell_fwd(float *__restrict__ y, const float *__restrict__ yp, int ld, int S, int C,
                                                        const float *__restrict__ pi, const float *__restrict__ pf,
                                                        const float *__restrict__ po, const int32_t *__restrict__ seq_len, int t) {
  constexpr int G = CIFG ? 3 : 4;
  const int GC = G * C, oc = GC, oh = GC + C, om = GC + 2 * C;
  const int og = 0, oi = C, of = CIFG ? C : 2 * C, oo = CIFG ? 2 * C : 3 * C;
  const int n = S * C;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int s = idx / C, c = idx - s * C;
    float *ys = y + (long)s * ld;
    const float *ps = yp + (long)s * ld;
    if (seq_len && t > seq_len[s]) {  // nnet-blstm-projected-streams.h:654-657: zero the whole row
      ys[og + c] = 0.f; ys[of + c] = 0.f; ys[oo + c] = 0.f; ys[oc + c] = 0.f; ys[oh + c] = 0.f; ys[om + c] = 0.f;
      if (!CIFG) ys[oi + c] = 0.f;
      continue;
    }
    const float cprev = ps[oc + c];
    const float g = tanh_ref(ys[og + c]);
    const float f = sigmoid_ref(ys[of + c] + cprev * pf[c]);
    float cc;
    if (!CIFG) {
      const float i = sigmoid_ref(ys[oi + c] + cprev * pi[c]);
      ys[oi + c] = i;
      cc = g * i + cprev * f;
    } else {
      cc = -g * f + g + cprev * f;  // cifg.h:372-378
    }
    cc = fminf(fmaxf(cc, -50.0f), 50.0f);
    const float h = tanh_ref(cc);
    const float o = sigmoid_ref(ys[oo + c] + cc * po[c]);
    ys[og + c] = g; ys[of + c] = f; ys[oo + c] = o; ys[oc + c] = cc; ys[oh + c] = h; ys[om + c] = h * o;
  }
}

// d: diff block of step t (d_m already holds dL/dm), dn: diff block of the step processed just before
// (recursion-next), y / yn / yp: activation blocks of t, recursion-next and recursion-previous.
template <bool CIFG>
__global__ void __launch_bounds__(kBlock) lstm_cell_bwd(float *__restrict__ d, const float *__restrict__ dn, const float *__restrict__ y,
                                                        const float *__restrict__ yn, const float *__restrict__ yp, int ld, int S, int C,
                                                        const float *__restrict__ pi, const float *__restrict__ pf,
                                                        const float *__restrict__ po) {
  constexpr int G = CIFG ? 3 : 4;
  const int GC = G * C, oc = GC, oh = GC + C, om = GC + 2 * C;
  const int og = 0, oi = C, of = CIFG ? C : 2 * C, oo = CIFG ? 2 * C : 3 * C;
  const int n = S * C;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int s = idx / C, c = idx - s * C;
    const long o_ = (long)s * ld;
    const float dm = d[o_ + om + c];
    const float yo = y[o_ + oo + c], yh = y[o_ + oh + c], yg = y[o_ + og + c], yf = y[o_ + of + c];
    const float dh = dtanh(yh, dm * yo);
    const float dov = dsigm(yo, dm * yh);
    float dc = dh + dn[o_ + oc + c] * yn[o_ + of + c];
    if (!CIFG) dc += dn[o_ + oi + c] * pi[c];
    dc += dn[o_ + of + c] * pf[c];
    dc += dov * po[c];
    const float cprev = yp[o_ + oc + c];
    d[o_ + oh + c] = dh;
    d[o_ + oo + c] = dov;
    d[o_ + oc + c] = dc;
    if (!CIFG) {
      const float yi = y[o_ + oi + c];
      d[o_ + of + c] = dsigm(yf, dc * cprev);
      d[o_ + oi + c] = dsigm(yi, dc * yg);
      d[o_ + og + c] = dtanh(yg, dc * yi);
    } else {  // cifg.h:529-536
      d[o_ + of + c] = dsigm(yf, dc * cprev - dc * yg);
      d[o_ + og + c] = dtanh(yg, dc - dc * yf);
    }
  }
}

// GRU forward, part 1 (after zr += h(t-1) W_zr_h^T): z, r = sigmoid; g = r .* h(t-1)
__global__ void __launch_bounds__(kBlock) gru_fwd1(float *__restrict__ y, const float *__restrict__ yp, int ld, int S, int H) {
  const int n = S * H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int s = idx / H, c = idx - s * H;
    float *ys = y + (long)s * ld;
    const float z = sigmoid_ref(ys[c]), r = sigmoid_ref(ys[H + c]);
    ys[c] = z;
    ys[H + c] = r;
    ys[3 * H + c] = r * yp[(long)s * ld + 4 * H + c];
  }
}
// part 2 (after m += g W_m_g^T): m = tanh; h = h(t-1) - h(t-1) z + z m
__global__ void __launch_bounds__(kBlock) gru_fwd2(float *__restrict__ y, const float *__restrict__ yp, int ld, int S, int H) {
  const int n = S * H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int s = idx / H, c = idx - s * H;
    float *ys = y + (long)s * ld;
    const float hp = yp[(long)s * ld + 4 * H + c], z = ys[c];
    const float m = tanh_ref(ys[2 * H + c]);
    ys[2 * H + c] = m;
    ys[4 * H + c] = hp - hp * z + z * m;
  }
}
// backward part 1 (after d_h += DZR(t+1) W_zr_h): finish d_h, d_m
__global__ void __launch_bounds__(kBlock) gru_bwd1(float *__restrict__ d, const float *__restrict__ dn, const float *__restrict__ y,
                                                   const float *__restrict__ yn, int ld, int S, int H) {
  const int n = S * H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int s = idx / H, c = idx - s * H;
    const long o = (long)s * ld;
    const float dhn = dn[o + 4 * H + c];
    const float dh = d[o + 4 * H + c] + dhn - dhn * yn[o + c] + dn[o + 3 * H + c] * yn[o + H + c];
    d[o + 4 * H + c] = dh;
    d[o + 2 * H + c] = dtanh(y[o + 2 * H + c], dh * y[o + c]);
  }
}
// backward part 2 (after d_g = d_m W_m_g): d_r, d_z
__global__ void __launch_bounds__(kBlock) gru_bwd2(float *__restrict__ d, const float *__restrict__ y, const float *__restrict__ yp, int ld,
                                                   int S, int H) {
  const int n = S * H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int s = idx / H, c = idx - s * H;
    const long o = (long)s * ld;
    const float hp = yp[o + 4 * H + c], dh = d[o + 4 * H + c];
    d[o + H + c] = dsigm(y[o + H + c], d[o + 3 * H + c] * hp);
    d[o + c] = dsigm(y[o + c], dh * y[o + 2 * H + c] - dh * hp);
  }
}

// Vector gradients of a recurrent component in one launch: every job is a column reduction over the T*S rows of the diff buffer,
// optionally weighted element-wise by an activation block (the peephole gradients diag(d^T c)), followed by momentum, the
// element-wise clip and -- when the caller folds the update in -- the SGD step.  16 columns x 64 row groups per block (a
// 64-byte segment per row and wave quarter; 7C/16 blocks, enough to cover the chip at C = 512); the 64 partial sums of a
// column meet in LDS in a fixed order, so results do not depend on the launch.
constexpr int kVgCols = 16, kVgGroups = 64, kVgMaxJobs = 8;
static_assert(kVgGroups == 64, "the LDS fold below is written for 64 row groups");
struct VecGradJobs { aslp_rnn_vec_grad j[kVgMaxJobs]; int first_block[kVgMaxJobs + 1]; int njobs; };

__global__ void __launch_bounds__(kVgCols * kVgGroups) rnn_vec_grads_kernel(VecGradJobs jobs, int ldd, int rows, float mmt, float clip, float neg_lr) {
  __shared__ float part[kVgGroups][kVgCols];
  int k = 0;
  while (k + 1 < jobs.njobs && (int)blockIdx.x >= jobs.first_block[k + 1]) k++;
  const aslp_rnn_vec_grad job = jobs.j[k];
  const int cx = threadIdx.x & (kVgCols - 1), ry = threadIdx.x / kVgCols;
  const int col = ((int)blockIdx.x - jobs.first_block[k]) * kVgCols + cx;
  float acc = 0.0f;
  if (col < job.n) {
    const float *dp = job.d + col;
    if (job.x != nullptr) {
      const float *xp = job.x + col;
      int r = ry;
      for (; r + 3 * kVgGroups < rows; r += 4 * kVgGroups) {
        const float d0 = dp[(long)r * ldd], d1 = dp[(long)(r + kVgGroups) * ldd], d2 = dp[(long)(r + 2 * kVgGroups) * ldd], d3 = dp[(long)(r + 3 * kVgGroups) * ldd];
        const float x0 = xp[(long)r * job.ldx], x1 = xp[(long)(r + kVgGroups) * job.ldx], x2 = xp[(long)(r + 2 * kVgGroups) * job.ldx], x3 = xp[(long)(r + 3 * kVgGroups) * job.ldx];
        acc += d0 * x0; acc += d1 * x1; acc += d2 * x2; acc += d3 * x3;
      }
      for (; r < rows; r += kVgGroups) acc += dp[(long)r * ldd] * xp[(long)r * job.ldx];
    } else {
      int r = ry;
      for (; r + 3 * kVgGroups < rows; r += 4 * kVgGroups) {
        const float d0 = dp[(long)r * ldd], d1 = dp[(long)(r + kVgGroups) * ldd], d2 = dp[(long)(r + 2 * kVgGroups) * ldd], d3 = dp[(long)(r + 3 * kVgGroups) * ldd];
        acc += d0; acc += d1; acc += d2; acc += d3;
      }
      for (; r < rows; r += kVgGroups) acc += dp[(long)r * ldd];
    }
  }
  part[ry][cx] = acc;
  __syncthreads();
  // fold 64 -> 4 partials per column with all threads, then one thread per column finishes
  if (ry < 16) part[ry][cx] += part[ry + 16][cx] + part[ry + 32][cx] + part[ry + 48][cx];
  __syncthreads();
  if (ry < 4) part[ry][cx] += part[ry + 4][cx] + part[ry + 8][cx] + part[ry + 12][cx];
  __syncthreads();
  if (ry == 0 && col < job.n) {
    float v = (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]);
    if (mmt != 0.0f) v += mmt * job.corr[col];
    if (clip > 0.0f) v = v < -clip ? -clip : (v > clip ? clip : v);
    job.corr[col] = v;
    if (neg_lr != 0.0f) job.param[col] += neg_lr * v;
  }
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

void aslp_lstm_cell_forward(float *y_cur, const float *y_prev, int ld, int S, int C, int cifg, const float *peep_i, const float *peep_f,
                            const float *peep_o, const int32_cuda *seq_lengths, int t) {
  if (S <= 0 || C <= 0) return;
  const int g = grid_for((long)S * C);
  if (cifg) hipLaunchKernelGGL((lstm_cell_fwd<true>), dim3(g), dim3(kBlock), 0, cur_stream(), y_cur, y_prev, ld, S, C, peep_i, peep_f, peep_o, seq_lengths, t);
  else hipLaunchKernelGGL((lstm_cell_fwd<false>), dim3(g), dim3(kBlock), 0, cur_stream(), y_cur, y_prev, ld, S, C, peep_i, peep_f, peep_o, seq_lengths, t);
  check_launch("lstm_cell_forward");
}
void aslp_lstm_cell_backward(float *d_cur, const float *d_next, const float *y_cur, const float *y_next, const float *y_prev, int ld, int S,
                             int C, int cifg, const float *peep_i, const float *peep_f, const float *peep_o) {
  if (S <= 0 || C <= 0) return;
  const int g = grid_for((long)S * C);
  if (cifg) hipLaunchKernelGGL((lstm_cell_bwd<true>), dim3(g), dim3(kBlock), 0, cur_stream(), d_cur, d_next, y_cur, y_next, y_prev, ld, S, C, peep_i, peep_f, peep_o);
  else hipLaunchKernelGGL((lstm_cell_bwd<false>), dim3(g), dim3(kBlock), 0, cur_stream(), d_cur, d_next, y_cur, y_next, y_prev, ld, S, C, peep_i, peep_f, peep_o);
  check_launch("lstm_cell_backward");
}
void aslp_gru_forward1(float *y_cur, const float *y_prev, int ld, int S, int H) {
  if (S <= 0 || H <= 0) return;
  hipLaunchKernelGGL(gru_fwd1, dim3(grid_for((long)S * H)), dim3(kBlock), 0, cur_stream(), y_cur, y_prev, ld, S, H);
  check_launch("gru_forward1");
}
void aslp_gru_forward2(float *y_cur, const float *y_prev, int ld, int S, int H) {
  if (S <= 0 || H <= 0) return;
  hipLaunchKernelGGL(gru_fwd2, dim3(grid_for((long)S * H)), dim3(kBlock), 0, cur_stream(), y_cur, y_prev, ld, S, H);
  check_launch("gru_forward2");
}
void aslp_gru_backward1(float *d_cur, const float *d_next, const float *y_cur, const float *y_next, int ld, int S, int H) {
  if (S <= 0 || H <= 0) return;
  hipLaunchKernelGGL(gru_bwd1, dim3(grid_for((long)S * H)), dim3(kBlock), 0, cur_stream(), d_cur, d_next, y_cur, y_next, ld, S, H);
  check_launch("gru_backward1");
}
void aslp_gru_backward2(float *d_cur, const float *y_cur, const float *y_prev, int ld, int S, int H) {
  if (S <= 0 || H <= 0) return;
  hipLaunchKernelGGL(gru_bwd2, dim3(grid_for((long)S * H)), dim3(kBlock), 0, cur_stream(), d_cur, y_cur, y_prev, ld, S, H);
  check_launch("gru_backward2");
}

void aslp_rnn_vec_grads(const aslp_rnn_vec_grad *jobs, int njobs, int ldd, int rows, float mmt, float clip, float neg_lr) {
  if (njobs <= 0) return;
  if (njobs > kVgMaxJobs) {   // a launch carries kVgMaxJobs jobs in its arguments: longer lists go out in pieces (independent jobs)
    for (int k = 0; k < njobs; k += kVgMaxJobs) aslp_rnn_vec_grads(jobs + k, njobs - k < kVgMaxJobs ? njobs - k : kVgMaxJobs, ldd, rows, mmt, clip, neg_lr);
    return;
  }
  VecGradJobs a;
  int blocks = 0;
  a.njobs = 0;
  for (int k = 0; k < njobs; k++) {
    if (jobs[k].n <= 0) continue;
    a.j[a.njobs] = jobs[k];
    a.first_block[a.njobs++] = blocks;
    blocks += (jobs[k].n + kVgCols - 1) / kVgCols;
  }
  if (blocks == 0) return;
  for (int k = a.njobs; k <= kVgMaxJobs; k++) a.first_block[k] = blocks;
  for (int k = a.njobs; k < kVgMaxJobs; k++) a.j[k] = a.j[0];
  hipLaunchKernelGGL(rnn_vec_grads_kernel, dim3(blocks), dim3(kVgCols * kVgGroups), 0, cur_stream(), a, ldd, rows, mmt, clip, neg_lr);
  check_launch("rnn_vec_grads");
}

}  // extern "C"
