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

__device__ __forceinline__ float log_plus(float p1, float p2) {  // ctc_helper.h:49-60
  if (p1 == kNegInf) return p2;
  if (p2 == kNegInf) return p1;
  return log1pf(expf(-fabsf(p1 - p2))) + fmaxf(p1, p2);
}

struct UttInfo {
  int T, L, S, repeats, feasible;
  int lab_off;   // offset of this utterance's S entries in labels_w_blanks / next_same / first_of_label
};

// ---- alpha --------------------------------------------------------------------------------------------
// LDS: two alpha columns + the blank-augmented label sequence.
__global__ void __launch_bounds__(256) ctc_alpha_kernel(const float *__restrict__ probs, float *__restrict__ alphas, const UttInfo *info,
                                                        const int *__restrict__ lwb_all, int A, int mb, int maxS, int maxT, float *loglike, int ldp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x;
  const UttInfo u = info[n];
  if (!u.feasible) {
    if (threadIdx.x == 0) loglike[n] = 0.0f;
    return;
  }
  const int S = u.S, T = u.T;
  float *a0 = smem, *a1 = smem + maxS;
  int *lab = reinterpret_cast<int *>(smem + 2 * maxS);
  for (int s = threadIdx.x; s < S; s += blockDim.x) lab[s] = lwb_all[u.lab_off + s];
  __syncthreads();
  const float *p = probs + (long)n * ldp;  // time stride ldp*mb
  const long tstride = (long)ldp * mb;
  float *al = alphas + (long)n * maxS * maxT;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    float v = s < 2 ? logf(p[lab[s]]) : kNegInf;
    a0[s] = v;
    al[s] = v;
  }
  __syncthreads();
  float *prev = a0, *cur = a1;
  for (int t = 1; t < T; t++) {
    const float *pt = p + t * tstride;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
      const int l = lab[s];
      float sum = prev[s];
      if (s >= 1) sum = log_plus(sum, prev[s - 1]);
      if (s >= 2 && l != 0 && l != lab[s - 2]) sum = log_plus(sum, prev[s - 2]);
      float v = sum + logf(pt[l]);
      cur[s] = v;
      al[(long)t * S + s] = v;
    }
    __syncthreads();
    float *tmp = prev; prev = cur; cur = tmp;
  }
  if (threadIdx.x == 0) {
    float ll = kNegInf;
    if (S > 1) ll = log_plus(ll, prev[S - 2]);
    ll = log_plus(ll, prev[S - 1]);
    loglike[n] = ll;
  }
}

// ---- beta + gradient ------------------------------------------------------------------------------------
// LDS: beta ping-pong [2*maxS], alpha*beta [maxS], labels [maxS], next_same [maxS], out[A], red[4]
__global__ void __launch_bounds__(256) ctc_beta_grad_kernel(const float *__restrict__ probs, const float *__restrict__ alphas, float *__restrict__ grads,
                                                            const UttInfo *info, const int *__restrict__ lwb_all, const int *__restrict__ next_all,
                                                            const int *__restrict__ first_all, int A, int mb, int maxS, int maxT,
                                                            const float *loglike, int ldg, int ldp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.x;
  const UttInfo u = info[n];
  if (!u.feasible) return;
  const int S = u.S, T = u.T;
  float *b0 = smem, *b1 = smem + maxS, *ab = smem + 2 * maxS;
  int *lab = reinterpret_cast<int *>(smem + 3 * maxS);
  int *nxt = lab + maxS;
  int *fst = nxt + maxS;
  float *out = reinterpret_cast<float *>(fst + maxS);
  float *red = out + A;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    lab[s] = lwb_all[u.lab_off + s];
    nxt[s] = next_all[u.lab_off + s];
    fst[s] = first_all[u.lab_off + s];
  }
  __syncthreads();
  const float logZ = loglike[n];
  const long tstride = (long)ldp * mb;
  const float *p = probs + (long)n * ldp;
  float *g = grads + (long)n * ldg;
  const long gstride = (long)ldg * mb;
  const float *al = alphas + (long)n * maxS * maxT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *nextb = b0, *curb = b1;  // nextb = beta_{t+1}, curb = beta_t
  for (int t = T - 1; t >= 0; t--) {
    const float *pt = p + t * tstride;
    for (int a = threadIdx.x; a < A; a += blockDim.x) out[a] = kNegInf;
    // beta_t and alpha*beta
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
      const int l = lab[s];
      float v;
      if (t == T - 1) {
        v = (s >= S - 2) ? logf(pt[l]) : kNegInf;
      } else {
        float sum = nextb[s];
        if (s + 1 < S) sum = log_plus(sum, nextb[s + 1]);
        if (s + 2 < S && l != 0 && l != lab[s + 2]) sum = log_plus(sum, nextb[s + 2]);
        v = sum + logf(pt[l]);
      }
      curb[s] = v;
      ab[s] = al[(long)t * S + s] + v;
    }
    __syncthreads();
    // reduce by label: blanks (even s) by a block-wide log-sum-exp, every other label by the
    // thread owning its first occurrence walking the chain of later occurrences (ascending s,
    // the same order as the reference's sequential reduce-by-key, cpu_ctc.h:337-339)
    float bl = kNegInf;
    for (int s = 2 * threadIdx.x; s < S; s += 2 * blockDim.x) bl = log_plus(bl, ab[s]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bl = log_plus(bl, __shfl_xor(bl, o, 64));
    if (lane == 0) red[wave] = bl;
    for (int s = 2 * threadIdx.x + 1; s < S; s += 2 * blockDim.x) {
      if (fst[s]) {
        float acc = ab[s];
        for (int q = nxt[s]; q >= 0; q = nxt[q]) acc = log_plus(ab[q], acc);
        out[lab[s]] = acc;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[0] = log_plus(log_plus(red[0], red[1]), log_plus(red[2], red[3]));
    __syncthreads();
    // gradient row (cpu_ctc.h:352-362)
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
      const float pr = pt[a], o = out[a];
      float gv;
      if (o == 0.0f || o == kNegInf || pr == 0.0f) gv = pr;
      else gv = pr - expf(o - logf(pr) - logZ);
      g[t * gstride + a] = gv;
    }
    __syncthreads();
    float *tmp = nextb; nextb = curb; curb = tmp;
  }
}

__global__ void neg_costs_kernel(const float *loglike, const UttInfo *info, float *costs, int mb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mb) costs[i] = info[i].feasible ? -loglike[i] : 0.0f;
}

struct Layout {
  size_t probs, alphas, info, lwb, nxt, fst, loglike, costs, total;
};
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
Layout make_layout(int A, int mb, int maxT, int maxS, size_t total_S) {
  Layout l;
  size_t o = 0;
  l.probs = o; o += align256(sizeof(float) * (size_t)maxT * mb * A);
  l.alphas = o; o += align256(sizeof(float) * (size_t)mb * maxS * maxT);
  l.info = o; o += align256(sizeof(UttInfo) * mb);
  l.lwb = o; o += align256(sizeof(int) * total_S);
  l.nxt = o; o += align256(sizeof(int) * total_S);
  l.fst = o; o += align256(sizeof(int) * total_S);
  l.loglike = o; o += align256(sizeof(float) * mb);
  l.costs = o; o += align256(sizeof(float) * mb);
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
    cudaF_softmax_reduce(0, 0, probs, activations, d, ld_acts);
  }
  set_cur_stream(saved);

  const size_t lds_alpha = sizeof(float) * 2 * maxS + sizeof(int) * maxS;
  const size_t lds_beta = sizeof(float) * 3 * maxS + sizeof(int) * 3 * maxS + sizeof(float) * (A + 4);
  if (lds_beta > 160 * 1024) return CTC_STATUS_INVALID_VALUE;  // alphabet + label length beyond one CU's LDS
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_alpha_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_beta_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(ctc_alpha_kernel, dim3(mb), dim3(256), lds_alpha, stream, probs, alphas, d_info, d_lwb, A, mb, maxS, maxT > 0 ? maxT : 1, d_ll, ldp);
  if (gradients != nullptr)
    hipLaunchKernelGGL(ctc_beta_grad_kernel, dim3(mb), dim3(256), lds_beta, stream, probs, alphas, gradients, d_info, d_lwb, d_nxt, d_fst, A, mb,
                       maxS, maxT > 0 ? maxT : 1, d_ll, ld_grads, ldp);
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
