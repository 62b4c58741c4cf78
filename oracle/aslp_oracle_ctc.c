/*
 * aslp_oracle_ctc.c -- TEST INFRASTRUCTURE ONLY (see aslp_oracle.h).
 *
 * Plain-C restatement of the reference's Warp-CTC CPU path
 * (src/warp-ctc/include/detail/cpu_ctc.h:158-428, ctc_helper.h:11-60) and of the thin wrapper
 * logic of src/aslp-nnet/warp-ctc.cc (valid-row copy-back, +-1 clip, outlier filter, token
 * error rate).  Pinned: tests/test_oracle_ctc_cpu.py checks it against the fixtures produced by
 * the reference itself (oracle/gen_ctc_golden.cpp -> tests/golden/ctc_*.bin) and, when
 * oracle/_ref/libwarpctc_ref.so is present, against that library on random inputs.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "aslp_oracle.h"

#define BLANK 0 /* ctc_helper.h:11 */

static float log_plus(float p1, float p2) { /* ctc_helper.h:49-60 */
  if (p1 == -INFINITY) return p2;
  if (p2 == -INFINITY) return p1;
  /* unqualified log1p/exp/fabs in the reference resolve to the C (double) functions: the
   * float difference is widened, the correction term is computed in double and the sum is
   * rounded to float once */
  return (float)(log1p(exp(-fabs((double)(p1 - p2)))) + (double)(p1 < p2 ? p2 : p1));
}

/* cpu_ctc.h:123-154 */
static int setup_labels(const int *labels, int L, int S, int *lwb, int *e_inc, int *s_inc) {
  int e = 0, s = 0, repeats = 0;
  s_inc[s++] = 1;
  for (int i = 1; i < L; ++i) {
    if (labels[i - 1] == labels[i]) {
      s_inc[s++] = 1; s_inc[s++] = 1;
      e_inc[e++] = 1; e_inc[e++] = 1;
      ++repeats;
    } else {
      s_inc[s++] = 2;
      e_inc[e++] = 2;
    }
  }
  e_inc[e++] = 1;
  for (int i = 0; i < L; ++i) {
    lwb[2 * i] = BLANK;
    lwb[2 * i + 1] = labels[i];
  }
  lwb[S - 1] = BLANK;
  return repeats;
}

/* cpu_ctc.h:217-262; probs points at (t=0, this utterance), time stride = tstride */
static float compute_alphas(const float *probs, int repeats, int S, int T, const int *e_inc, const int *s_inc,
                            const int *labels, float *alphas, int tstride) {
  int start = (((S / 2) + repeats - T) < 0) ? 0 : 1, end = S > 1 ? 2 : 1;
  for (int i = start; i < end; ++i) alphas[i] = logf(probs[labels[i]]);
  for (int t = 1; t < T; ++t) {
    int remain = (S / 2) + repeats - (T - t);
    if (remain >= 0) start += s_inc[remain];
    if (t <= (S / 2) + repeats) end += e_inc[t - 1];
    int startloop = start;
    int idx1 = t * S, idx2 = (t - 1) * S, idx3 = t * tstride;
    if (start == 0) {
      alphas[idx1] = alphas[idx2] + logf(probs[BLANK + idx3]);
      startloop += 1;
    }
    for (int i = startloop; i < end; ++i) {
      float prev_sum = log_plus(alphas[i + idx2], alphas[(i - 1) + idx2]);
      if (labels[i] != BLANK && i != 1 && labels[i] != labels[i - 2]) prev_sum = log_plus(prev_sum, alphas[(i - 2) + idx2]);
      alphas[i + idx1] = prev_sum + logf(probs[labels[i] + idx3]);
    }
  }
  float loglike = -INFINITY;
  for (int i = start; i < end; ++i) loglike = log_plus(loglike, alphas[i + (T - 1) * S]);
  return loglike;
}

/* cpu_ctc.h:269-367 */
static float compute_betas_and_grad(float *grad, const float *probs, float log_partition, int repeats, int S, int T,
                                    const int *e_inc, const int *s_inc, const int *labels, float *alphas, float *betas,
                                    float *output, int A, int tstride) {
  int start = S > 1 ? (S - 2) : 0, end = (T > (S / 2) + repeats) ? S : S - 1;
  for (int i = 0; i < A; i++) output[i] = -INFINITY;
  for (int i = start; i < end; ++i) {
    betas[i] = logf(probs[labels[i] + (T - 1) * tstride]);
    alphas[i + (T - 1) * S] += betas[i];
    output[labels[i]] = log_plus(alphas[i + (T - 1) * S], output[labels[i]]);
  }
  for (int i = 0; i < A; ++i) {
    int idx3 = (T - 1) * tstride + i;
    if (output[i] == 0.0 || output[i] == -INFINITY || probs[idx3] == 0.0) grad[idx3] = probs[idx3];
    else grad[idx3] = probs[idx3] - expf(output[i] - logf(probs[idx3]) - log_partition);
  }
  for (int t = T - 2; t >= 0; --t) {
    int remain = (S / 2) + repeats - (T - t);
    if (remain >= -1) start -= s_inc[remain + 1];
    if (t < (S / 2) + repeats) end -= e_inc[t];
    int endloop = end == S ? end - 1 : end;
    int idx1 = t * S, idx3 = t * tstride;
    for (int i = 0; i < A; i++) output[i] = -INFINITY;
    for (int i = start; i < endloop; ++i) {
      float next_sum = log_plus(betas[i], betas[(i + 1)]);
      if (labels[i] != BLANK && i != (S - 2) && labels[i] != labels[i + 2]) next_sum = log_plus(next_sum, betas[(i + 2)]);
      betas[i] = next_sum + logf(probs[labels[i] + idx3]);
      alphas[i + idx1] += betas[i];
      output[labels[i]] = log_plus(alphas[i + idx1], output[labels[i]]);
    }
    if (end == S) {
      betas[(S - 1)] = betas[(S - 1)] + logf(probs[BLANK + idx3]);
      alphas[(S - 1) + idx1] += betas[(S - 1)];
      output[labels[S - 1]] = log_plus(alphas[S - 1 + idx1], output[labels[S - 1]]);
    }
    for (int i = 0; i < A; ++i) {
      if (output[i] == 0.0 || output[i] == -INFINITY || probs[idx3] == 0.0) grad[idx3] = probs[idx3];
      else grad[idx3] = probs[idx3] - expf(output[i] - logf(probs[idx3]) - log_partition);
      ++idx3;
    }
  }
  float loglike = -INFINITY;
  for (int i = start; i < end; ++i) loglike = log_plus(loglike, betas[i]);
  return loglike;
}

/* cpu_ctc.h:158-179 (softmax) + :369-428 (cost_and_grad).  acts/grads layout (t, n, p);
 * grads must be zeroed by the caller (cpu_ctc.h:268).  grads == NULL: scores only. */
int orc_ctc_cost_and_grad(const float *acts, float *grads, const int *flat_labels, const int *label_lengths,
                          const int *input_lengths, int A, int mb, float *costs) {
  int maxT = 0, maxL = 0;
  for (int i = 0; i < mb; i++) {
    if (input_lengths[i] > maxT) maxT = input_lengths[i];
    if (label_lengths[i] > maxL) maxL = label_lengths[i];
  }
  int maxS = 2 * maxL + 1;
  const int tstride = A * mb;
  float *probs = (float *)calloc((size_t)maxT * tstride, sizeof(float));
  for (int n = 0; n < mb; ++n)
    for (int c = 0; c < input_lengths[n]; ++c) {
      int off = (n + mb * c) * A;
      float mx = -INFINITY;
      for (int r = 0; r < A; ++r) mx = acts[r + off] > mx ? acts[r + off] : mx;
      float denom = 0.0f;
      for (int r = 0; r < A; ++r) denom += expf(acts[r + off] - mx);
      for (int r = 0; r < A; ++r) probs[r + off] = expf(acts[r + off] - mx) / denom;
    }
  float *alphas = (float *)malloc(sizeof(float) * (size_t)maxS * (maxT > 0 ? maxT : 1));
  float *betas = (float *)malloc(sizeof(float) * maxS);
  float *output = (float *)malloc(sizeof(float) * A);
  float *scratch_grad = grads ? NULL : (float *)malloc(sizeof(float) * (size_t)(maxT > 0 ? maxT : 1) * tstride);
  int *lwb = (int *)malloc(sizeof(int) * maxS), *e_inc = (int *)malloc(sizeof(int) * maxS), *s_inc = (int *)malloc(sizeof(int) * maxS);
  int lab_off = 0;
  for (int n = 0; n < mb; ++n) {
    const int T = input_lengths[n], L = label_lengths[n], S = 2 * L + 1;
    for (int i = 0; i < S * T; i++) alphas[i] = -INFINITY;
    for (int i = 0; i < S; i++) betas[i] = -INFINITY;
    int repeats = setup_labels(flat_labels + lab_off, L, S, lwb, e_inc, s_inc);
    lab_off += L;
    if (L + repeats > T) { /* :196-198 */
      costs[n] = 0.0f;
      continue;
    }
    float ll = compute_alphas(probs + n * A, repeats, S, T, e_inc, s_inc, lwb, alphas, tstride);
    if (grads)
      compute_betas_and_grad(grads + n * A, probs + n * A, ll, repeats, S, T, e_inc, s_inc, lwb, alphas, betas, output, A, tstride);
    costs[n] = -ll;
  }
  free(probs); free(alphas); free(betas); free(output); free(scratch_grad); free(lwb); free(e_inc); free(s_inc);
  return 0;
}

/* ---- src/aslp-nnet/warp-ctc.cc wrapper logic ---------------------------------------------------- */

/* warp-ctc.cc:288-349 StatAndAverageLossCheck (the active variant: warp-ctc.h:25 sets
 * WARP_CTC_GRAD_CHECK = WARP_CTC_AVG_LOSS_CHECK).  While fewer than stat_period/2 (= 250)
 * utterances have been seen everything is kept and accumulated; afterwards an utterance is kept
 * only if its cost is finite, in (0, 3000), and its per-frame loss lies within
 * mean +- 6*sqrt(sum_sq/n) (an RMS, not a standard deviation -- kept as is) of the running
 * window; the window restarts every stat_period kept utterances.  keep[n] = 0 means the
 * caller zeroes that utterance's diff rows.  `obj` accumulates kept costs. */
void orc_ctc_loss_filter(const float *costs, const int *frame_num, int mb, orc_ctc_filter_state *st, int *keep) {
  for (int s = 0; s < mb; s++) {
    keep[s] = 1;
    double loss_per_frame = costs[s] / frame_num[s];
    if (st->normal_num < st->stat_period / 2) {
      st->normal_num++;
      st->loss_sum += loss_per_frame;
      st->loss_sum_bak += loss_per_frame;
      st->loss_square_sum += loss_per_frame * loss_per_frame;
      st->loss_square_sum_bak += loss_per_frame * loss_per_frame;
      st->obj += costs[s];
    } else {
      double mean = st->loss_sum / st->normal_num;
      double sigma = sqrt(st->loss_square_sum / st->normal_num);
      if (isfinite(costs[s]) && (loss_per_frame >= (mean - 6 * sigma) && loss_per_frame <= (mean + 6 * sigma)) &&
          (costs[s] > 0 && costs[s] < 3000)) {
        st->normal_num++;
        st->loss_sum += loss_per_frame;
        st->loss_square_sum += loss_per_frame * loss_per_frame;
        st->obj += costs[s];
        if (st->normal_num == st->stat_period) {
          st->loss_sum -= st->loss_sum_bak;
          st->loss_square_sum -= st->loss_square_sum_bak;
          st->loss_sum_bak = st->loss_sum;
          st->loss_square_sum_bak = st->loss_square_sum;
          st->normal_num = st->stat_period / 2;
        }
      } else {
        keep[s] = 0;
      }
    }
    st->frames += frame_num[s];
  }
  st->sequences += mb;
}

/* warp-ctc.cc:487-526: greedy path (argmax per frame), collapse repeats, drop blanks, then
 * Levenshtein distance to the reference label sequence.  net_out is [T x A] for one utterance
 * with row stride ld.  Returns the edit distance; *hyp_len gets the hypothesis length. */
int orc_ctc_token_errors(const float *net_out, int ld, int T, int A, const int *ref, int ref_len, int *hyp_len) {
  int *hyp = (int *)malloc(sizeof(int) * (T > 0 ? T : 1));
  int n = 0, prev = -1;
  for (int t = 0; t < T; t++) {
    const float *row = net_out + (size_t)t * ld;
    int best = 0;
    for (int a = 1; a < A; a++)
      if (row[a] > row[best]) best = a;
    if (best != prev && best != BLANK) hyp[n++] = best;
    prev = best;
  }
  if (hyp_len) *hyp_len = n;
  int *d = (int *)malloc(sizeof(int) * (n + 1));
  for (int j = 0; j <= n; j++) d[j] = j;
  for (int i = 1; i <= ref_len; i++) {
    int prev_diag = d[0];
    d[0] = i;
    for (int j = 1; j <= n; j++) {
      int tmp = d[j];
      int sub = prev_diag + (ref[i - 1] != hyp[j - 1]);
      int del = d[j] + 1, ins = d[j - 1] + 1;
      d[j] = sub < del ? (sub < ins ? sub : ins) : (del < ins ? del : ins);
      prev_diag = tmp;
    }
  }
  int res = d[n];
  free(hyp);
  free(d);
  return res;
}

/* ---- Eesen CTC (Ctc::EvalParallel, aslp-nnet/ctc-loss.cc:115-227) --------------------------------------
 * GPU-only in the reference: restated from the device kernels aslp-cudamatrix/cu-kernels.cu:3276-3534 and
 * the log-domain helpers of ctc-utils.h:52-95 (float: log_zero = -1e30, exp_limit = 88.722839).
 * PARITY UNPINNED on its own (no CPU path, no tests in the reference); cross-checked in
 * tests/test_oracle_ctc_cpu.py against the pinned Warp-CTC restatement: the unclipped diff equals the
 * Warp-CTC gradient on the pre-softmax activations and -pzx equals its cost. */
#define E_LOG_ZERO (-1e30f)
#define E_LOG_INF (1e30f)
#define E_EXP_LIMIT (88.722839f)
#define E_MAX (3.4028235e+038f)
static float e_add(float a, float b) { return (a == E_LOG_ZERO || b == E_LOG_ZERO) ? E_LOG_ZERO : a + b; }
static float e_sub(float a, float b) { return a == E_LOG_ZERO ? E_LOG_ZERO : (b == E_LOG_ZERO ? E_LOG_INF : a - b); }
static float e_exp(float a) { return a <= E_LOG_ZERO ? 0.0f : (a >= E_EXP_LIMIT ? E_MAX : expf(a)); }
static float e_logaddexp(float a, float b) {
  if (b < a) return e_add(a, logf(1 + e_exp(e_sub(b, a))));
  return e_add(b, logf(1 + e_exp(e_sub(a, b))));
}
static double e_logaddexp_d(double a, double b) { /* host call in double, ctc-loss.cc:176 (double limits -1e100 / 709.78) */
  const double LZ = -1e100, LI = 1e100, EL = 709.78271289338397, MX = 1.7976931348623157e+308;
#define DSUB(x, y) ((x) == LZ ? LZ : ((y) == LZ ? LI : (x) - (y)))
#define DEXP(x) ((x) <= LZ ? 0.0 : ((x) >= EL ? MX : exp(x)))
#define DADD(x, y) (((x) == LZ || (y) == LZ) ? LZ : (x) + (y))
  if (b < a) return DADD(a, log(1 + DEXP(DSUB(b, a))));
  return DADD(b, log(1 + DEXP(DSUB(a, b))));
#undef DSUB
#undef DEXP
#undef DADD
}

/* net_out: softmax outputs [T*S x A] (row = t*S + s, leading dim ld); labels flat with label_lens[S];
 * frame_num[S].  diff [T*S x A] (leading dim ldd) receives ctc_err .* y - y * rowsum(ctc_err .* y), UNclipped,
 * zero rows past each utterance's end; pzx[S] receives log p(z|x). */
void orc_eesen_ctc_mseq(const float *net_out, int ld, int T, int S, int A, const int *flat_labels, const int *label_lens,
                        const int *frame_num, float *diff, int ldd, float *pzx) {
  int maxL = 0;
  for (int s = 0; s < S; s++)
    if (label_lens[s] > maxL) maxL = label_lens[s];
  const int E = 2 * maxL + 1;
  int *lab = (int *)malloc(sizeof(int) * (size_t)S * E);
  int *explen = (int *)malloc(sizeof(int) * S);
  for (int i = 0; i < S * E; i++) lab[i] = -1;
  int off = 0;
  for (int s = 0; s < S; s++) {
    explen[s] = 2 * label_lens[s] + 1;
    for (int l = 0; l < label_lens[s]; l++) {
      lab[s * E + 2 * l] = 0;
      lab[s * E + 2 * l + 1] = flat_labels[off + l];
    }
    lab[s * E + 2 * label_lens[s]] = 0;
    off += label_lens[s];
  }
  float *logp = (float *)malloc(sizeof(float) * (size_t)T * S * A);
  for (int r = 0; r < T * S; r++)
    for (int a = 0; a < A; a++) logp[(size_t)r * A + a] = logf(net_out[(size_t)r * ld + a]);
  float *alpha = (float *)malloc(sizeof(float) * (size_t)T * S * E), *beta = (float *)malloc(sizeof(float) * (size_t)T * S * E);
  for (size_t i = 0; i < (size_t)T * S * E; i++) alpha[i] = beta[i] = E_LOG_ZERO;
  for (int t = 0; t < T; t++) /* _compute_ctc_alpha_multiple_sequence */
    for (int s = 0; s < S; s++)
      for (int j = 0; j < E; j++) {
        float *a = alpha + ((size_t)t * S + s) * E + j;
        const int cls = lab[s * E + j];
        if (cls == -1 || t >= frame_num[s]) { *a = E_LOG_ZERO; continue; }
        const float p = logp[((size_t)t * S + s) * A + cls];
        const float *pa = alpha + ((size_t)(t - 1) * S + s) * E + j;
        if (t == 0) *a = j < 2 ? p : E_LOG_ZERO;
        else if (j > 1) {
          if (j % 2 == 0 || lab[s * E + j - 2] == cls) *a = e_add(p, e_logaddexp(pa[-1], pa[0]));
          else { float tmp = e_logaddexp(pa[-1], pa[0]); *a = e_add(p, e_logaddexp(pa[-2], tmp)); }
        } else if (j == 1) *a = e_add(p, e_logaddexp(pa[-1], pa[0]));
        else *a = e_add(p, pa[0]);
      }
  for (int t = T - 1; t >= 0; t--) /* _compute_ctc_beta_multiple_sequence */
    for (int s = 0; s < S; s++)
      for (int j = 0; j < E; j++) {
        float *b = beta + ((size_t)t * S + s) * E + j;
        const int cls = lab[s * E + j];
        if (cls == -1 || t >= frame_num[s]) { *b = E_LOG_ZERO; continue; }
        const float p = logp[((size_t)t * S + s) * A + cls];
        const float *nb = beta + ((size_t)(t + 1) * S + s) * E + j;
        const int LL = explen[s];
        if (t == frame_num[s] - 1) *b = j > LL - 3 ? p : E_LOG_ZERO;
        else if (j < LL - 2) {
          if (j % 2 == 0 || lab[s * E + j + 2] == cls) *b = e_add(p, e_logaddexp(nb[1], nb[0]));
          else { float tmp = e_logaddexp(nb[1], nb[0]); *b = e_add(p, e_logaddexp(nb[2], tmp)); }
        } else if (j == LL - 2) *b = e_add(p, e_logaddexp(nb[1], nb[0]));
        else *b = e_add(p, nb[0]);
      }
  for (int s = 0; s < S; s++) { /* ctc-loss.cc:169-177 */
    const int LL = explen[s], fn = frame_num[s];
    float tmp1 = alpha[((size_t)(fn - 1) * S + s) * E + LL - 1];
    float tmp2 = LL >= 2 ? alpha[((size_t)(fn - 1) * S + s) * E + LL - 2] : E_LOG_ZERO;
    pzx[s] = (float)e_logaddexp_d((double)tmp1, (double)tmp2);
  }
  for (int r = 0; r < T * S; r++) { /* _compute_ctc_error_multiple_sequence, then ctc-loss.cc:183-191 */
    const int s = r % S, t = r / S;
    float *d = diff + (size_t)r * ldd;
    for (int a = 0; a < A; a++) d[a] = 0.0f;
    if (t >= frame_num[s]) continue;
    float rowsum = 0.0f;
    for (int k = 0; k < A; k++) {
      float err = E_LOG_ZERO;
      for (int j = 0; j < E; j++) {
        if (lab[s * E + j] == -1) continue;
        if (lab[s * E + j] == k) err = e_logaddexp(err, e_add(alpha[(size_t)r * E + j], beta[(size_t)r * E + j]));
      }
      const float y = net_out[(size_t)r * ld + k];
      float val = e_exp(e_sub(err, e_add(pzx[s], y == 0 ? E_LOG_ZERO : 2 * logf(y))));
      float ce = -1.0f * val;
      ce *= y; /* ctc_err_.MulElements(net_out) */
      d[k] = ce;
      rowsum += ce;
    }
    for (int k = 0; k < A; k++) d[k] += -1.0f * (net_out[(size_t)r * ld + k] * rowsum);
  }
  free(lab); free(explen); free(logp); free(alpha); free(beta);
}

/* Ctc::StatAndAverageLossCheck (ctc-loss.cc:229-302): like the WarpCtc one, but the warm-up phase also
 * requires a finite cost in (0, 3000) -- and silently skips (without dropping the diff) when it is not. */
void orc_eesen_ctc_loss_filter(const float *costs, const int *frame_num, int mb, orc_ctc_filter_state *st, int *keep) {
  for (int s = 0; s < mb; s++) {
    keep[s] = 1;
    double loss_per_frame = costs[s] / frame_num[s];
    if (st->normal_num < st->stat_period / 2) {
      if (isfinite(costs[s]) && costs[s] > 0 && costs[s] < 3000) {
        st->normal_num++;
        st->loss_sum += loss_per_frame;
        st->loss_sum_bak += loss_per_frame;
        st->loss_square_sum += loss_per_frame * loss_per_frame;
        st->loss_square_sum_bak += loss_per_frame * loss_per_frame;
        st->obj += costs[s];
      }
    } else {
      double mean = st->loss_sum / st->normal_num;
      double sigma = sqrt(st->loss_square_sum / st->normal_num);
      if (isfinite(costs[s]) && (loss_per_frame >= (mean - 6 * sigma) && loss_per_frame <= (mean + 6 * sigma)) &&
          (costs[s] > 0 && costs[s] < 3000)) {
        st->normal_num++;
        st->loss_sum += loss_per_frame;
        st->loss_square_sum += loss_per_frame * loss_per_frame;
        st->obj += costs[s];
        if (st->normal_num == st->stat_period) {
          st->loss_sum -= st->loss_sum_bak;
          st->loss_square_sum -= st->loss_square_sum_bak;
          st->loss_sum_bak = st->loss_sum;
          st->loss_square_sum_bak = st->loss_square_sum;
          st->normal_num = st->stat_period / 2;
        }
      } else {
        keep[s] = 0;
      }
    }
    st->frames += frame_num[s];
  }
  st->sequences += mb;
}
