// ctc.hip -- CTC forward-backward for gfx950 behind the Warp-CTC C ABI (include/aslp_ctc.h).
//
// One workgroup (4 waves) per utterance runs the whole alpha recursion over T with the previous
// and current alpha columns ping-ponging in LDS (per-state log-sum-exp, no per-timestep kernel
// launches -- the reference's Eesen path launches one kernel per frame, ctc-loss.cc:160-165);
// a second kernel sweeps beta backwards, forms alpha*beta per state in LDS, reduces it by label
// and writes the gradient row.  Mathematically this is the reference CPU implementation
// (src/warp-ctc/include/detail/cpu_ctc.h:217-367) without its [start, end) pruning window:
// states outside the window have alpha or beta = -inf in the full recursion, so every product
// and the final log-likelihood are identical.
//
// Roofline: latency / LDS bound (a serial chain of T steps per utterance); HBM traffic is the
// algorithmic minimum 4*T*(2A + 2S) bytes per utterance (read probs, write grads, write+read alphas).
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "aslp_ctc.h"
#include "aslp_kernels.h"
#include "common.h"
#include "scratch.h"

namespace aslp {
namespace {

constexpr float kNegInf = -INFINITY;

// ctc_helper.h:49-60.  The reference's log_plus evaluates log1p(exp(-|a-b|)) + max(a,b) in DOUBLE (the unqualified
// libm calls promote) and rounds once to float; at |alpha| ~ 1e3 (long utterances) a float evaluation is off by up to
// an ulp of the SUM (6e-5), which shows as ~6e-5 relative gradient error.  The lattice is latency-bound, so the double
// transcendental is affordable and buys parity at the 1e-7 level.
__device__ __forceinline__ float log_plus(float p1, float p2) {
  if (p1 == kNegInf) return p2;
  if (p2 == kNegInf) return p1;
  return (float)(log1p(exp(-fabs((double)p1 - (double)p2))) + (double)fmaxf(p1, p2));
}

// The reference's host code calls std::log / std::exp on floats (glibc logf / expf: computed in double, rounded once,
// correctly rounded in all but ~1e-9 of the cases).  alpha / beta reach magnitudes of several 1e3 on long utterances, where
// one float ulp is 2-5e-4: unless the per-frame log-probabilities are BIT-identical to the reference's, the two lattices
// drift apart by a few ulps and the posteriors by ~1e-4 relative.  So the CTC path evaluates them the same way.
__device__ __forceinline__ float logf_cr(float x) { return (float)log((double)x); }
__device__ __forceinline__ float expf_cr(float x) { return (float)exp((double)x); }

// softmax of cpu_ctc.h:158-179 with the reference's exact arithmetic: max, exp(x - max) rounded to float, denominator
// summed in float in index order, float division.  One lane per row (the sum order is sequential by definition).
__global__ void __launch_bounds__(256) ctc_softmax_kernel(const float *__restrict__ acts, int ld, float *__restrict__ probs, int rows, int A) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float *x = acts + (long)r * ld;
  float *p = probs + (long)r * A;
  float mx = -INFINITY;
  for (int a = 0; a < A; a++) mx = fmaxf(mx, x[a]);
  float denom = 0.0f;
  for (int a = 0; a < A; a++) {
    const float e = expf_cr(x[a] - mx);
    p[a] = e;
    denom += e;
  }
  for (int a = 0; a < A; a++) p[a] = p[a] / denom;
}

// The same arithmetic with coalesced memory traffic: a workgroup stages TR rows in LDS (pitch odd: a lane walking its row hits a bank
// of its own), all threads take the element-wise parts (row maximum -- exact in any order --, exp(x - max), the division), and only
// the denominator, whose float sum is sequential BY DEFINITION (cpu_ctc.h:171-174), is walked by one lane per row, 64 rows side by side
// in a wave.  The lane-per-row kernel above read its row with a stride of A floats per lane: 0.33 TB/s on 25,600 x 128.
__global__ void __launch_bounds__(256) ctc_softmax_tile_kernel(const float *__restrict__ acts, int ld, float *__restrict__ probs, int rows, int A,
                                                               int TR, int pitch) {
  extern __shared__ float tile[];          // [TR][pitch] values, then [TR] row maxima / denominators
  float *rowv = tile + TR * pitch;
  const int r0 = blockIdx.x * TR, nr = min(TR, rows - r0), n = nr * A;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i / A, a = i - r * A;
    tile[r * pitch + a] = acts[(long)(r0 + r) * ld + a];
  }
  __syncthreads();
  // row maxima: 4 threads per row (any order gives the same maximum)
  {
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    float mx = -INFINITY;
    if (r < nr)
      for (int a = q; a < A; a += 4) mx = fmaxf(mx, tile[r * pitch + a]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    if (r < nr && q == 0) rowv[r] = mx;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i / A, a = i - r * A;
    tile[r * pitch + a] = expf_cr(tile[r * pitch + a] - rowv[r]);
  }
  __syncthreads();
  if (threadIdx.x < nr) {   // the denominator: float additions in index order
    const float *e = tile + threadIdx.x * pitch;
    float denom = 0.0f;
    for (int a = 0; a < A; a++) denom += e[a];
    rowv[threadIdx.x] = denom;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i / A, a = i - r * A;
    probs[(long)(r0 + r) * A + a] = tile[r * pitch + a] / rowv[r];
  }
}

struct UttInfo {
  int T, L, S, repeats, feasible;
  int lab_off;   // offset of this utterance's S entries in labels_w_blanks / next_same / first_of_label
};

// ---- emission log-probabilities ------------------------------------------------------------------------
// log p_t(label of state s) for every (utterance, frame, state), stored where alpha(t, s) and beta(t, s) will go.  The lattice
// recursions below then find their only frame-dependent input already in the workspace -- coalesced, requested one frame ahead --
// instead of a gather through the label table plus a double-precision logarithm on the sequential path of every timestep.
__global__ void __launch_bounds__(256) ctc_emit_kernel(const float *__restrict__ probs, float *__restrict__ alphas, float *__restrict__ betas,
                                                       const UttInfo *info, const int *__restrict__ lwb_all, int mb, int maxS, int maxT, int ldp,
                                                       int want_beta) {
  const int n = blockIdx.y, t = blockIdx.x;
  const UttInfo u = info[n];
  if (!u.feasible || t >= u.T) return;
  const float *pt = probs + ((long)t * mb + n) * ldp;
  const int *lab = lwb_all + u.lab_off;
  float *al = alphas + (long)n * maxS * maxT + (long)t * u.S, *be = betas + (long)n * maxS * maxT + (long)t * u.S;
  for (int s = threadIdx.x; s < u.S; s += blockDim.x) {
    const float lp = logf_cr(pt[lab[s]]);
    al[s] = lp;
    if (want_beta) be[s] = lp;
  }
}

// ---- alpha / beta ------------------------------------------------------------------------------------------
// One workgroup per utterance and direction walks its lattice over T with the previous and current columns ping-ponging in
// LDS (cpu_ctc.h:217-255 / :300-350 without the [start, end) window).  Every thread owns up to kLatSlots states
// (tid + k * blockDim); a step's emission terms are requested at its top and consumed at its end, so the only things on the
// sequential path are three LDS reads, two log_plus and the barrier.
constexpr int kLatSlots = 8;

// blockIdx.y == 0: alpha pass; blockIdx.y == 1: beta pass of utterance blockIdx.x (both at once: 2 x mb workgroups)
__global__ void __launch_bounds__(512) ctc_lattice_kernel(float *__restrict__ alphas, float *__restrict__ betas, const UttInfo *info,
                                                          const int *__restrict__ lwb_all, int mb, int maxS, int maxT, float *loglike) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x;
  const UttInfo u = info[n];
  const bool beta = blockIdx.y == 1;
  if (!u.feasible) {
    if (!beta && threadIdx.x == 0) loglike[n] = 0.0f;
    return;
  }
  const int S = u.S, T = u.T, nthr = blockDim.x, nslots = (S + nthr - 1) / nthr;  // nslots <= kLatSlots (host check), uniform
  float *c0 = smem, *c1 = smem + maxS;
  int *lab = reinterpret_cast<int *>(smem + 2 * maxS);
  for (int s = threadIdx.x; s < S; s += nthr) lab[s] = lwb_all[u.lab_off + s];
  __syncthreads();
  float *lat = (beta ? betas : alphas) + (long)n * maxS * maxT;   // holds the emission terms, becomes the lattice
  // per-slot constants: may this state take the skip transition (from s - 2 forward, from s + 2 backward)?
  bool skip[kLatSlots];
#pragma unroll
  for (int k = 0; k < kLatSlots; k++) {
    const int s = threadIdx.x + k * nthr;
    skip[k] = false;
    if (s < S) {
      const int l = lab[s];
      skip[k] = beta ? (s + 2 < S && l != 0 && l != lab[s + 2]) : (s >= 2 && l != 0 && l != lab[s - 2]);
    }
  }
  // first column
  const int tfirst = beta ? T - 1 : 0, dt = beta ? -1 : 1;
  float lp[kLatSlots];
#pragma unroll
  for (int k = 0; k < kLatSlots; k++) {
    const int s = threadIdx.x + k * nthr;
    lp[k] = s < S ? lat[(long)tfirst * S + s] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < kLatSlots; k++) {
    const int s = threadIdx.x + k * nthr;
    if (s < S) {
      const bool start = beta ? (s >= S - 2) : (s < 2);
      const float v = start ? lp[k] : kNegInf;
      c0[s] = v;
      lat[(long)tfirst * S + s] = v;
    }
  }
  __syncthreads();
  float *prev = c0, *cur = c1;
  for (int step = 1; step < T; step++) {
    const int t = tfirst + dt * step;
#pragma unroll
    for (int k = 0; k < kLatSlots; k++) {  // this frame's emission terms: in flight while the LDS reads and log_plus of the step run
      if (k >= nslots) break;
      const int s = threadIdx.x + k * nthr;
      lp[k] = s < S ? lat[(long)t * S + s] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < kLatSlots; k++) {
      if (k >= nslots) break;
      const int s = threadIdx.x + k * nthr;
      if (s < S) {
        float sum = prev[s];
        if (beta) {
          if (s + 1 < S) sum = log_plus(sum, prev[s + 1]);
          if (skip[k]) sum = log_plus(sum, prev[s + 2]);
        } else {
          if (s >= 1) sum = log_plus(sum, prev[s - 1]);
          if (skip[k]) sum = log_plus(sum, prev[s - 2]);
        }
        const float v = sum + lp[k];
        cur[s] = v;
        lat[(long)t * S + s] = v;
      }
    }
    __syncthreads();
    float *tmp = prev; prev = cur; cur = tmp;
  }
  if (!beta && threadIdx.x == 0) {
    float ll = kNegInf;
    if (S > 1) ll = log_plus(ll, prev[S - 2]);
    ll = log_plus(ll, prev[S - 1]);
    loglike[n] = ll;
  }
}

// ---- gradient ---------------------------------------------------------------------------------------------
// With both lattices in the workspace the per-frame work -- alpha*beta, the reduce-by-label and the gradient row
// (cpu_ctc.h:318-362) -- is independent across frames: one wave per (frame, utterance), all CUs busy.  The
// reduce-by-label keeps the reference's SEQUENTIAL order (ascending state index, every step a double log_plus
// rounded to float): at |alpha + beta| of several 1e3 a float ulp is 2-5e-4, so any other summation order moves
// the posteriors by ~1e-4 relative.  Lane 0 walks the blank states, the other lanes the chains of the non-blank
// labels (first occurrence -> next occurrence of the same label).
// The blank label owns every other state: its reduce-by-label is ONE chain of L + 1 sequential log_plus per frame.  Walked by a
// single lane of the frame's wave (as the other chains are) it made the gradient kernel FP64-issue bound at 1/64 lane
// utilisation (1.44 of 3.07 ms for 32 x <= 800 x 128).  Here the chains of 64 FRAMES run side by side in the lanes of one wave;
// the order inside every chain is unchanged (ascending state index, each step rounded to float).
__global__ void __launch_bounds__(64) ctc_blank_kernel(const float *__restrict__ alphas, const float *__restrict__ betas, const UttInfo *info,
                                                       int maxS, int maxT, float *__restrict__ blank_acc) {
  const int n = blockIdx.y, t = blockIdx.x * 64 + threadIdx.x;
  const UttInfo u = info[n];
  if (!u.feasible || t >= u.T) return;
  const int S = u.S;
  const float *al = alphas + (long)n * maxS * maxT + (long)t * S, *be = betas + (long)n * maxS * maxT + (long)t * S;
  float acc = kNegInf;
  int s = 0;
  for (; s + 14 < S; s += 16) {  // eight blank states per round: their loads are in flight together
    float ab[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ab[k] = al[s + 2 * k] + be[s + 2 * k];
#pragma unroll
    for (int k = 0; k < 8; k++) acc = log_plus(ab[k], acc);
  }
  for (; s < S; s += 2) acc = log_plus(al[s] + be[s], acc);
  blank_acc[(long)n * maxT + t] = acc;
}

constexpr int kGradWaves = 4;
__global__ void __launch_bounds__(64 * kGradWaves) ctc_grad_kernel(const float *__restrict__ probs, const float *__restrict__ alphas,
                                                                    const float *__restrict__ betas, float *__restrict__ grads, const UttInfo *info,
                                                                    const int *__restrict__ lwb_all, const int *__restrict__ next_all,
                                                                    const int *__restrict__ first_all, int A, int mb, int maxS, int maxT,
                                                                    const float *loglike, int ldg, int ldp, const float *__restrict__ blank_acc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.y;
  const UttInfo u = info[n];
  if (!u.feasible) return;
  const int S = u.S, T = u.T;
  int *lab = reinterpret_cast<int *>(smem);
  int *nxt = lab + maxS;
  int *fst = nxt + maxS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *ab = reinterpret_cast<float *>(fst + maxS) + wave * (maxS + A);
  float *out = ab + maxS;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    lab[s] = lwb_all[u.lab_off + s];
    nxt[s] = next_all[u.lab_off + s];
    fst[s] = first_all[u.lab_off + s];
  }
  __syncthreads();
  const float logZ = loglike[n];
  const long tstride = (long)ldp * mb, gstride = (long)ldg * mb;
  const float *p = probs + (long)n * ldp;
  float *g = grads + (long)n * ldg;
  const float *al = alphas + (long)n * maxS * maxT, *be = betas + (long)n * maxS * maxT;
  const int wstride = gridDim.x * kGradWaves;
  const int iters = (T + wstride - 1) / wstride;  // same trip count for every wave of the grid: block barriers are legal
  for (int it = 0; it < iters; it++) {
    const int t = it * wstride + blockIdx.x * kGradWaves + wave;
    const bool live = t < T;
    if (live) {
      for (int s = lane; s < S; s += 64) ab[s] = al[(long)t * S + s] + be[(long)t * S + s];
      for (int a = lane; a < A; a += 64) out[a] = kNegInf;
    }
    __syncthreads();
    if (live) {
      if (lane == 0) {
        out[0] = blank_acc[(long)n * maxT + t];   // ctc_blank_kernel
      } else {
        for (int s = 2 * (lane - 1) + 1; s < S; s += 2 * 63) {
          if (fst[s]) {
            float acc = ab[s];
            for (int q = nxt[s]; q >= 0; q = nxt[q]) acc = log_plus(ab[q], acc);
            out[lab[s]] = acc;
          }
        }
      }
    }
    __syncthreads();
    if (live) {
      const float *pt = p + t * tstride;
      for (int a = lane; a < A; a += 64) {
        const float pr = pt[a], o = out[a];
        float gv;
        if (o == 0.0f || o == kNegInf || pr == 0.0f) gv = pr;
        else gv = pr - expf_cr(o - logf_cr(pr) - logZ);
        g[t * gstride + a] = gv;
      }
    }
    __syncthreads();
  }
}

__global__ void neg_costs_kernel(const float *loglike, const UttInfo *info, float *costs, int mb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mb) costs[i] = info[i].feasible ? -loglike[i] : 0.0f;
}

struct Layout {
  size_t probs, alphas, betas, info, lwb, nxt, fst, loglike, costs, blank, total;
};
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
Layout make_layout(int A, int mb, int maxT, int maxS, size_t total_S) {
  Layout l;
  size_t o = 0;
  l.probs = o; o += align256(sizeof(float) * (size_t)maxT * mb * A);
  l.alphas = o; o += align256(sizeof(float) * (size_t)mb * maxS * maxT);
  l.betas = o; o += align256(sizeof(float) * (size_t)mb * maxS * maxT);
  l.info = o; o += align256(sizeof(UttInfo) * mb);
  l.lwb = o; o += align256(sizeof(int) * total_S);
  l.nxt = o; o += align256(sizeof(int) * total_S);
  l.fst = o; o += align256(sizeof(int) * total_S);
  l.loglike = o; o += align256(sizeof(float) * mb);
  l.costs = o; o += align256(sizeof(float) * mb);
  l.blank = o; o += align256(sizeof(float) * (size_t)mb * maxT);
  l.total = o;
  return l;
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

int get_warpctc_version(void) { return 2; }

const char *ctcGetStatusString(ctcStatus_t status) {
  switch (status) {
    case CTC_STATUS_SUCCESS: return "no error";
    case CTC_STATUS_MEMOPS_FAILED: return "cuda memcpy or memset failed";
    case CTC_STATUS_INVALID_VALUE: return "invalid value";
    case CTC_STATUS_EXECUTION_FAILED: return "execution failed";
    case CTC_STATUS_UNKNOWN_ERROR:
    default: return "unknown error";
  }
}

ctcStatus_t get_workspace_size(const int *const label_lengths, const int *const input_lengths, int alphabet_size, int minibatch,
                               struct ctcComputeInfo info, size_t *size_bytes) {
  if (label_lengths == nullptr || input_lengths == nullptr || size_bytes == nullptr || alphabet_size <= 0 || minibatch <= 0)
    return CTC_STATUS_INVALID_VALUE;  // ctc_entrypoint.cpp:92-98
  (void)info;
  int maxL = *std::max_element(label_lengths, label_lengths + minibatch);
  int maxT = *std::max_element(input_lengths, input_lengths + minibatch);
  if (maxT < 0 || maxL < 0) return CTC_STATUS_INVALID_VALUE;
  const int maxS = 2 * maxL + 1;
  *size_bytes = make_layout(alphabet_size, minibatch, maxT > 0 ? maxT : 1, maxS, (size_t)maxS * minibatch).total;
  return CTC_STATUS_SUCCESS;
}

static ctcStatus_t ctc_loss_impl(const float *const activations, int ld_acts, float *gradients, int ld_grads, const int *const flat_labels,
                                 const int *const label_lengths, const int *const input_lengths, int A, int mb, float *costs, void *workspace,
                                 struct ctcComputeInfo cinfo, bool acts_are_probs = false) {
  if (activations == nullptr || flat_labels == nullptr || label_lengths == nullptr || input_lengths == nullptr || costs == nullptr ||
      workspace == nullptr || A <= 0 || mb <= 0)
    return CTC_STATUS_INVALID_VALUE;  // ctc_entrypoint.cpp:46-54
  if (cinfo.loc == CTC_CPU) return CTC_STATUS_EXECUTION_FAILED;  // no CPU path in this library
  if (cinfo.loc != CTC_GPU) return CTC_STATUS_INVALID_VALUE;
  hipStream_t stream = reinterpret_cast<hipStream_t>(cinfo.stream);
  const int maxL = *std::max_element(label_lengths, label_lengths + mb);
  const int maxT = *std::max_element(input_lengths, input_lengths + mb);
  const int maxS = 2 * maxL + 1;
  const Layout lay = make_layout(A, mb, maxT > 0 ? maxT : 1, maxS, (size_t)maxS * mb);
  char *ws = static_cast<char *>(workspace);
  float *probs = reinterpret_cast<float *>(ws + lay.probs);
  float *alphas = reinterpret_cast<float *>(ws + lay.alphas);
  UttInfo *d_info = reinterpret_cast<UttInfo *>(ws + lay.info);
  int *d_lwb = reinterpret_cast<int *>(ws + lay.lwb), *d_nxt = reinterpret_cast<int *>(ws + lay.nxt), *d_fst = reinterpret_cast<int *>(ws + lay.fst);
  float *d_ll = reinterpret_cast<float *>(ws + lay.loglike), *d_costs = reinterpret_cast<float *>(ws + lay.costs);

  // host-side label preparation (cpu_ctc.h:123-154): blanks interleaved, repeat count, and for the
  // reduce-by-label the chain "next state with the same label"
  std::vector<UttInfo> h_info(mb);
  std::vector<int> h_lwb((size_t)maxS * mb, 0), h_nxt((size_t)maxS * mb, -1), h_fst((size_t)maxS * mb, 0);
  std::vector<int> last(A);
  int off = 0;
  for (int n = 0; n < mb; n++) {
    const int L = label_lengths[n], T = input_lengths[n], S = 2 * L + 1;
    const int *lab = flat_labels + off;
    off += L;
    int repeats = 0;
    for (int i = 1; i < L; i++) repeats += lab[i - 1] == lab[i];
    UttInfo &u = h_info[n];
    u.T = T; u.L = L; u.S = S; u.repeats = repeats; u.lab_off = n * maxS;
    u.feasible = (T > 0 && L + repeats <= T) ? 1 : 0;  // cpu_ctc.h:196-198
    int *lw = &h_lwb[(size_t)n * maxS], *nx = &h_nxt[(size_t)n * maxS], *fs = &h_fst[(size_t)n * maxS];
    std::fill(last.begin(), last.end(), -1);
    for (int i = 0; i < L; i++) {
      if (lab[i] < 0 || lab[i] >= A) return CTC_STATUS_INVALID_VALUE;
      lw[2 * i] = 0;
      lw[2 * i + 1] = lab[i];
      const int s = 2 * i + 1;
      if (last[lab[i]] < 0) fs[s] = 1; else nx[last[lab[i]]] = s;
      last[lab[i]] = s;
    }
    lw[S - 1] = 0;
  }
  if (hipMemcpyAsync(d_info, h_info.data(), sizeof(UttInfo) * mb, hipMemcpyHostToDevice, stream) != hipSuccess ||
      hipMemcpyAsync(d_lwb, h_lwb.data(), sizeof(int) * h_lwb.size(), hipMemcpyHostToDevice, stream) != hipSuccess ||
      hipMemcpyAsync(d_nxt, h_nxt.data(), sizeof(int) * h_nxt.size(), hipMemcpyHostToDevice, stream) != hipSuccess ||
      hipMemcpyAsync(d_fst, h_fst.data(), sizeof(int) * h_fst.size(), hipMemcpyHostToDevice, stream) != hipSuccess)
    return CTC_STATUS_MEMOPS_FAILED;
  // the host vectors die at return: the copies must have been consumed (pageable memcpyAsync stages
  // synchronously on ROCm, but do not rely on it)
  if (hipStreamSynchronize(stream) != hipSuccess) return CTC_STATUS_MEMOPS_FAILED;

  // softmax over the alphabet for every (t, n) row (cpu_ctc.h:158-179)
  hipStream_t saved = cur_stream();
  set_cur_stream(stream);
  MatrixDim d = {maxT * mb, A, A};
  int ldp = A;
  if (acts_are_probs) {  // Eesen convention: the caller already ran the Softmax component
    probs = const_cast<float *>(activations);
    ldp = ld_acts;
  } else if (maxT > 0) {
    {
      const int pitch = A | 1;
      int TR = (40 * 1024 / 4 - 64) / pitch;   // rows of a 40 KB tile
      TR = TR > 64 ? 64 : TR;
      static const bool old_kernel = getenv("ASLP_CTC_SOFTMAX_TILED") != nullptr && getenv("ASLP_CTC_SOFTMAX_TILED")[0] == '0';   // A/B switch
      if (TR >= 1 && !old_kernel)
        hipLaunchKernelGGL(ctc_softmax_tile_kernel, dim3((d.rows + TR - 1) / TR), dim3(256), sizeof(float) * (size_t)(TR * pitch + TR), stream, activations, ld_acts,
                           probs, d.rows, A, TR, pitch);
      else
        hipLaunchKernelGGL(ctc_softmax_kernel, dim3((d.rows + 255) / 256), dim3(256), 0, stream, activations, ld_acts, probs, d.rows, A);
    }
  }
  set_cur_stream(saved);

  const int Tl = maxT > 0 ? maxT : 1;
  const size_t lds_lattice = sizeof(float) * 2 * maxS + sizeof(int) * maxS;
  const size_t lds_grad = sizeof(int) * 3 * maxS + sizeof(float) * kGradWaves * ((size_t)maxS + A);
  if (lds_lattice > 160 * 1024 || lds_grad > 160 * 1024) return CTC_STATUS_INVALID_VALUE;  // alphabet + label length beyond one CU's LDS
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_lattice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  float *betas = reinterpret_cast<float *>(ws + lay.betas);
  // emission terms for every (utterance, frame, state) in one parallel pass, then the alpha and (when gradients are wanted)
  // beta lattices side by side: 2 x mb workgroups; as many threads as states (up to 512), kLatSlots states per thread beyond
  hipLaunchKernelGGL(ctc_emit_kernel, dim3(Tl, mb), dim3(256), 0, stream, probs, alphas, betas, d_info, d_lwb, mb, maxS, Tl, ldp,
                     gradients != nullptr ? 1 : 0);
  int lat_threads = 64;
  while (lat_threads < maxS && lat_threads < 512) lat_threads *= 2;
  if (maxS > lat_threads * kLatSlots) return CTC_STATUS_INVALID_VALUE;  // label sequences beyond 2047 symbols
  hipLaunchKernelGGL(ctc_lattice_kernel, dim3(mb, gradients != nullptr ? 2 : 1), dim3(lat_threads), lds_lattice, stream, alphas, betas, d_info, d_lwb,
                     mb, maxS, Tl, d_ll);
  if (gradients != nullptr) {
    // one wave per (frame, utterance); a few frames per wave so the label tables are staged once per workgroup
    int tchunks = (Tl + kGradWaves * 4 - 1) / (kGradWaves * 4);
    if (tchunks < 1) tchunks = 1;
    float *d_blank = reinterpret_cast<float *>(ws + lay.blank);
    hipLaunchKernelGGL(ctc_blank_kernel, dim3((Tl + 63) / 64, mb), dim3(64), 0, stream, alphas, betas, d_info, maxS, Tl, d_blank);
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(tchunks, mb), dim3(64 * kGradWaves), lds_grad, stream, probs, alphas, betas, gradients, d_info, d_lwb,
                       d_nxt, d_fst, A, mb, maxS, Tl, d_ll, ld_grads, ldp, d_blank);
  }
  hipLaunchKernelGGL(neg_costs_kernel, dim3((mb + 255) / 256), dim3(256), 0, stream, d_ll, d_info, d_costs, mb);
  if (hipGetLastError() != hipSuccess) return CTC_STATUS_EXECUTION_FAILED;
  if (hipMemcpyAsync(costs, d_costs, sizeof(float) * mb, hipMemcpyDeviceToHost, stream) != hipSuccess) return CTC_STATUS_MEMOPS_FAILED;
  if (hipStreamSynchronize(stream) != hipSuccess) return CTC_STATUS_EXECUTION_FAILED;
  return CTC_STATUS_SUCCESS;
}

ctcStatus_t compute_ctc_loss(const float *const activations, float *gradients, const int *const flat_labels, const int *const label_lengths,
                             const int *const input_lengths, int A, int mb, float *costs, void *workspace, struct ctcComputeInfo cinfo) {
  return ctc_loss_impl(activations, A, gradients, A, flat_labels, label_lengths, input_lengths, A, mb, costs, workspace, cinfo);
}

// Same computation on row-padded matrices (what the host engine holds): activation row (t*mb + n) starts
// at acts + (t*mb + n)*ld_acts, gradient rows likewise; the workspace comes from the library's grow-only
// scratch slot, so a training loop does no allocation, no de-striding copy and no row-wise copy-back
// (the reference's wrapper does all three per call, aslp-nnet/warp-ctc.cc:85-95,105-113,139-147).
ctcStatus_t aslp_ctc_loss_strided(const float *acts, int ld_acts, float *grads, int ld_grads, const int *flat_labels, const int *label_lengths,
                                  const int *input_lengths, int A, int mb, float *costs) {
  if (ld_acts < A || (grads != nullptr && ld_grads < A)) return CTC_STATUS_INVALID_VALUE;
  struct ctcComputeInfo info;
  info.loc = CTC_GPU;
  info.stream = reinterpret_cast<CUstream>(cur_stream());
  size_t bytes = 0;
  ctcStatus_t st = get_workspace_size(label_lengths, input_lengths, A, mb, info, &bytes);
  if (st != CTC_STATUS_SUCCESS) return st;
  void *ws = scratch(kScratchCtc, bytes);
  if (!ws) return CTC_STATUS_MEMOPS_FAILED;
  return ctc_loss_impl(acts, ld_acts, grads, ld_grads, flat_labels, label_lengths, input_lengths, A, mb, costs, ws, info);
}

// Eesen-style objective (Ctc::EvalParallel, aslp-nnet/ctc-loss.cc:115-227): `net_out` are POST-softmax
// probabilities; diff = y - posterior for frames t < frame_num[n] (the reference forms it as
// err.*y - y*rowsum(err.*y) from 2T per-row kernel launches plus an O(T*A*(2L+1)) error kernel,
// cu-kernels.cu:3276-3534), untouched elsewhere; pzx_host[n] = log p(z|x), -1e30 (the reference's
// log_zero_) when no alignment exists.  Same lattice kernels as compute_ctc_loss.
ctcStatus_t aslp_eesen_ctc_mseq(const float *net_out, int ld, float *diff, int ld_diff, const int *flat_labels, const int *label_lengths,
                                const int *frame_num, int A, int mb, float *pzx_host) {
  if (ld < A || ld_diff < A || diff == nullptr) return CTC_STATUS_INVALID_VALUE;
  struct ctcComputeInfo info;
  info.loc = CTC_GPU;
  info.stream = reinterpret_cast<CUstream>(cur_stream());
  size_t bytes = 0;
  ctcStatus_t st = get_workspace_size(label_lengths, frame_num, A, mb, info, &bytes);
  if (st != CTC_STATUS_SUCCESS) return st;
  void *ws = scratch(kScratchCtc, bytes);
  if (!ws) return CTC_STATUS_MEMOPS_FAILED;
  st = ctc_loss_impl(net_out, ld, diff, ld_diff, flat_labels, label_lengths, frame_num, A, mb, pzx_host, ws, info, true);
  if (st != CTC_STATUS_SUCCESS) return st;
  int off = 0;
  for (int n = 0; n < mb; n++) {  // costs -> log-likelihoods; infeasible alignments are log_zero_
    const int L = label_lengths[n];
    int repeats = 0;
    for (int i = 1; i < L; i++) repeats += flat_labels[off + i - 1] == flat_labels[off + i];
    off += L;
    const bool feasible = frame_num[n] > 0 && L + repeats <= frame_num[n];
    pzx_host[n] = feasible ? -pzx_host[n] : -1e30f;
    if (pzx_host[n] < -1e30f) pzx_host[n] = -1e30f;  // -inf (zero-probability path) is log_zero_ there
  }
  return CTC_STATUS_SUCCESS;
}

}  // extern "C"
