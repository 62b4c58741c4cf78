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
