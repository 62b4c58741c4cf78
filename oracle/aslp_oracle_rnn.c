/*
 * aslp_oracle_rnn.c -- TEST INFRASTRUCTURE ONLY (see aslp_oracle.h).
 *
 * Plain-C restatement of the reference's recurrent components (CPU path), one function per
 * direction, op by op in the reference's order:
 *   LstmProjectedStreams      nnet-lstm-projected-streams.h:313-617
 *   BLstmProjectedStreams     nnet-blstm-projected-streams.h:467-1036
 *   BLstmProjectedStreamsLC   nnet-blstm-projected-streams-lc.h:503-1110
 *   LstmCifgProjectedStreams  nnet-lstm-couple-if-projected-streams.h:300-600
 *   Lstm / BLstm              nnet-recurrent-component.cc:235-554, 912-1450
 *   GruStreams                nnet-gru-streams.h:238-450
 * Pinning: the reference ships no tests for these and its component headers cannot be compiled here (OpenFst),
 * but its CuMatrix library can (linked to the image's OpenBLAS): every gate block below -- forward buffer, backward buffer, input
 * diff, gradients, incl. the carried state, the backward-in-time direction and the length masking -- is checked against the same
 * op sequence issued on that library (oracle/gen_cumatrix_blas_golden.cpp -> tests/golden/cumatrix_blas_ops.bin,
 * tests/test_oracle_ref_blas_cpu.py).  The chunk / stream-reset bookkeeping stays restated from source (PARITY UNPINNED for that
 * part); tests/test_oracle_rnn_cpu.py checks backward against central differences of forward besides.
 *
 * Buffer layout as in the reference: rows (T+2)*S, row = t*S + s; columns
 *   LSTM: [g | i | f | o | c | h | m | r]  (CIFG: [g | f | o | c | h | m | r]; no r without projection)
 *   GRU : [z | r | m | g | h]
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "aslp_oracle.h"

static float sigm(float x) {
  if (x > 0.0) return 1.0 / (1.0 + expf(-x));
  float ex = expf(x);
  return ex / (ex + 1.0);
}
static float tanh_k(float x) {
  if (x > 0.0) {
    float ie = expf(-x);
    return -1.0 + 2.0 / (1.0 + ie * ie);
  }
  float ie = expf(x);
  return 1.0 - 2.0 / (1.0 + ie * ie);
}
static float dsigm(float y, float d) { return d * y * (1.0 - y); }
static float dtanh(float y, float d) { return d * (1.0 - (y * y)); }

/* p->... helpers */
static int gates(const orc_lstm_dir *p) { return p->cifg ? 3 : 4; }
int orc_lstm_width(const orc_lstm_dir *p) { return (gates(p) + 3) * p->C + (p->R > 0 ? p->R : 0); }

/* Forward pass of one direction over one batch.
 * in [T*S x D]; buf [(T+2)*S x width] (zeroed here like Resize(kSetZero)); if init_state != NULL it is
 * copied to the history row block (row block 0 for the forward-in-time direction).
 * reverse != 0: recursion runs t = T..1 reading t+1.  seq_len (may be NULL): rows with t > seq_len[s]
 * are zeroed after the step (BLstm* backward direction, nnet-blstm-projected-streams.h:654-657). */
void orc_lstm_forward(const orc_lstm_dir *p, const float *in, int ldi, int T, int S, int reverse, const float *init_state,
                      const int32_t *seq_len, float *buf) {
  const int C = p->C, R = p->R, D = p->D, G = gates(p), W = orc_lstm_width(p), GC = G * C;
  const int oc = GC, oh = GC + C, om = GC + 2 * C, orr = GC + 3 * C;
  const int og = 0, oi = p->cifg ? -1 : C, of = p->cifg ? C : 2 * C, oo = p->cifg ? 2 * C : 3 * C;
  const int rec = R > 0 ? R : C, orec = R > 0 ? orr : om; /* recurrent signal: r, or m without projection */
  memset(buf, 0, sizeof(float) * (size_t)(T + 2) * S * W);
  if (init_state) memcpy(buf + (size_t)(reverse ? (T + 1) : 0) * S * W, init_state, sizeof(float) * (size_t)S * W);
  /* x -> gates for all t at once, then bias (lc.h:553-556) */
  orc_add_mat_mat(buf + (size_t)S * W, T * S, GC, W, 1.0f, in, ldi, 0, p->w_x, D, 1, D, 0.0f);
  for (int r = S; r < (T + 1) * S; r++)
    for (int c = 0; c < GC; c++) buf[(size_t)r * W + c] += 1.0f * p->bias[c];
  for (int step = 0; step < T; step++) {
    const int t = reverse ? T - step : 1 + step, tp = reverse ? t + 1 : t - 1;
    float *y = buf + (size_t)t * S * W;
    const float *yp = buf + (size_t)tp * S * W;
    /* r(t-1) -> gates */
    orc_add_mat_mat(y, S, GC, W, 1.0f, yp + orec, W, 0, p->w_r, rec, 1, rec, 1.0f);
    for (int s = 0; s < S; s++) {
      float *ys = y + (size_t)s * W;
      const float *ps = yp + (size_t)s * W;
      for (int c = 0; c < C; c++) {
        float g, i = 0.0f, f, cc;
        if (!p->cifg) {
          ys[oi + c] += 1.0f * ps[oc + c] * p->peep_i[c];
          ys[of + c] += 1.0f * ps[oc + c] * p->peep_f[c];
          i = ys[oi + c] = sigm(ys[oi + c]);
          f = ys[of + c] = sigm(ys[of + c]);
          g = ys[og + c] = tanh_k(ys[og + c]);
          cc = 0.0f * ys[oc + c] + 1.0f * g * i;
          cc = 1.0f * cc + 1.0f * ps[oc + c] * f;
        } else { /* cifg.h:372-378: c = -g*f + g + c(t-1)*f */
          ys[of + c] += 1.0f * ps[oc + c] * p->peep_f[c];
          f = ys[of + c] = sigm(ys[of + c]);
          g = ys[og + c] = tanh_k(ys[og + c]);
          cc = 0.0f * ys[oc + c] + -1.0f * g * f;
          cc += 1.0f * g;
          cc = 1.0f * cc + 1.0f * ps[oc + c] * f;
        }
        if (cc < -50) cc = -50;
        if (cc > 50) cc = 50;
        ys[oc + c] = cc;
        float h = ys[oh + c] = tanh_k(cc);
        ys[oo + c] += 1.0f * cc * p->peep_o[c];
        float o = ys[oo + c] = sigm(ys[oo + c]);
        ys[om + c] = 0.0f * ys[om + c] + 1.0f * h * o;
      }
    }
    if (R > 0) orc_add_mat_mat(y + orr, S, R, W, 1.0f, y + om, W, 0, p->w_rm, C, 1, C, 0.0f);
    if (seq_len)
      for (int s = 0; s < S; s++)
        if (t > seq_len[s]) memset(y + (size_t)s * W, 0, sizeof(float) * W);
  }
}

/* Backward pass (BPTT "version 1", lc.h:762-835): dbuf [(T+2)*S x width] zeroed here; out_diff [T*S x rec]
 * (leading dim ldo) is the diff w.r.t. this direction's output (r, or m without projection).
 * in_diff += / = (beta) dGATES * w_x. */
void orc_lstm_backward(const orc_lstm_dir *p, const float *out_diff, int ldo, int T, int S, int reverse, const float *buf,
                       float *dbuf, float *in_diff, int ldid, float in_diff_beta) {
  const int C = p->C, R = p->R, D = p->D, G = gates(p), W = orc_lstm_width(p), GC = G * C;
  const int oc = GC, oh = GC + C, om = GC + 2 * C, orr = GC + 3 * C;
  const int og = 0, oi = p->cifg ? -1 : C, of = p->cifg ? C : 2 * C, oo = p->cifg ? 2 * C : 3 * C;
  const int rec = R > 0 ? R : C, orec = R > 0 ? orr : om;
  memset(dbuf, 0, sizeof(float) * (size_t)(T + 2) * S * W);
  for (int r = 0; r < T * S; r++) memcpy(dbuf + (size_t)(S + r) * W + orec, out_diff + (size_t)r * ldo, sizeof(float) * rec);
  for (int step = 0; step < T; step++) {
    /* the BPTT runs against the recursion direction */
    const int t = reverse ? 1 + step : T - step;
    const int tn = reverse ? t - 1 : t + 1; /* "next" in recursion order (already processed) */
    const int tp = reverse ? t + 1 : t - 1; /* "previous" in recursion order */
    float *d = dbuf + (size_t)t * S * W;
    const float *dn = dbuf + (size_t)tn * S * W;
    const float *y = buf + (size_t)t * S * W, *yn = buf + (size_t)tn * S * W, *yp = buf + (size_t)tp * S * W;
    /* d_rec += dGATES(next) * w_r */
    orc_add_mat_mat(d + orec, S, rec, W, 1.0f, dn, W, 0, p->w_r, rec, 0, GC, 1.0f);
    if (R > 0) orc_add_mat_mat(d + om, S, C, W, 1.0f, d + orr, W, 0, p->w_rm, C, 0, R, 0.0f);
    for (int s = 0; s < S; s++) {
      float *ds = d + (size_t)s * W;
      const float *dns = dn + (size_t)s * W, *ys = y + (size_t)s * W, *yns = yn + (size_t)s * W, *yps = yp + (size_t)s * W;
      for (int c = 0; c < C; c++) {
        float dm = ds[om + c];
        float dh = dtanh(ys[oh + c], 1.0f * dm * ys[oo + c]);
        float dov = dsigm(ys[oo + c], 1.0f * dm * ys[oh + c]);
        float dc = ds[oc + c] + 1.0f * dh;
        dc = 1.0f * dns[oc + c] * yns[of + c] + 1.0f * dc;
        if (!p->cifg) dc += 1.0f * dns[oi + c] * p->peep_i[c];
        dc += 1.0f * dns[of + c] * p->peep_f[c];
        dc += 1.0f * dov * p->peep_o[c];
        ds[oh + c] = dh;
        ds[oo + c] = dov;
        ds[oc + c] = dc;
        if (!p->cifg) {
          ds[of + c] = dsigm(ys[of + c], 1.0f * dc * yps[oc + c]);
          ds[oi + c] = dsigm(ys[oi + c], 1.0f * dc * ys[og + c]);
          ds[og + c] = dtanh(ys[og + c], 1.0f * dc * ys[oi + c]);
        } else { /* cifg.h:529-536 */
          float df = 1.0f * dc * yps[oc + c];
          df = -1.0f * dc * ys[og + c] + 1.0f * df;
          ds[of + c] = dsigm(ys[of + c], df);
          float dg = -1.0f * dc * ys[of + c];
          dg += 1.0f * dc;
          ds[og + c] = dtanh(ys[og + c], dg);
        }
      }
    }
  }
  orc_add_mat_mat(in_diff, T * S, D, ldid, 1.0f, dbuf + (size_t)S * W, W, 0, p->w_x, D, 0, GC, in_diff_beta);
}

static void clipv(float *v, size_t n, float c) {
  if (c <= 0.0f) return;
  for (size_t i = 0; i < n; i++) {
    if (v[i] < -c) v[i] = -c;
    if (v[i] > c) v[i] = c;
  }
}
static void diag_mat_mat(float *v, const float *A, const float *B, int rows, int ld, int cols, float beta) {
  /* v[c] = sum_r A[r][c] * B[r][c] + beta v[c]  (CuVector::AddDiagMatMat(1, A, kTrans, B, kNoTrans, beta)) */
  for (int c = 0; c < cols; c++) {
    float s = 0.0f;
    for (int r = 0; r < rows; r++) s += A[(size_t)r * ld + c] * B[(size_t)r * ld + c];
    v[c] = 1.0f * s + beta * v[c];
  }
}

/* Gradient accumulation with momentum + per-element clipping (lc.h:976-1058), into g->*. */
void orc_lstm_grads(const orc_lstm_dir *p, orc_lstm_dir *g, const float *in, int ldi, int T, int S, int reverse, const float *buf,
                    const float *dbuf, float mmt, float clip) {
  const int C = p->C, R = p->R, D = p->D, G = gates(p), W = orc_lstm_width(p), GC = G * C;
  const int oc = GC, om = GC + 2 * C, orr = GC + 3 * C;
  const int oi = p->cifg ? -1 : C, of = p->cifg ? C : 2 * C, oo = p->cifg ? 2 * C : 3 * C;
  const int rec = R > 0 ? R : C, orec = R > 0 ? orr : om;
  const float *dG = dbuf + (size_t)S * W; /* rows 1..T */
  const float *yprev = buf + (size_t)(reverse ? 2 : 0) * S * W; /* recursion-previous rows for t = 1..T */
  orc_add_mat_mat(g->w_x, GC, D, D, 1.0f, dG, W, 1, in, ldi, 0, T * S, mmt);
  orc_add_mat_mat(g->w_r, GC, rec, rec, 1.0f, dG, W, 1, yprev + orec, W, 0, T * S, mmt);
  for (int c = 0; c < GC; c++) {
    float s = 0.0f;
    for (int r = 0; r < T * S; r++) s += dG[(size_t)r * W + c];
    g->bias[c] = 1.0f * s + mmt * g->bias[c];
  }
  if (!p->cifg) diag_mat_mat(g->peep_i, dG + oi, yprev + oc, T * S, W, C, mmt);
  diag_mat_mat(g->peep_f, dG + of, yprev + oc, T * S, W, C, mmt);
  diag_mat_mat(g->peep_o, dG + oo, buf + (size_t)S * W + oc, T * S, W, C, mmt);
  if (R > 0) orc_add_mat_mat(g->w_rm, R, C, C, 1.0f, dG + orr, W, 1, buf + (size_t)S * W + om, W, 0, T * S, mmt);
  clipv(g->w_x, (size_t)GC * D, clip);
  clipv(g->w_r, (size_t)GC * rec, clip);
  clipv(g->bias, GC, clip);
  if (R > 0) clipv(g->w_rm, (size_t)R * C, clip);
  if (!p->cifg) clipv(g->peep_i, C, clip);
  clipv(g->peep_f, C, clip);
  clipv(g->peep_o, C, clip);
}

void orc_lstm_update(orc_lstm_dir *p, const orc_lstm_dir *g, float lr) { /* lc.h:1085-1110 */
  const int C = p->C, R = p->R, D = p->D, GC = gates(p) * C, rec = R > 0 ? R : C;
  for (size_t i = 0; i < (size_t)GC * D; i++) p->w_x[i] += -lr * g->w_x[i];
  for (size_t i = 0; i < (size_t)GC * rec; i++) p->w_r[i] += -lr * g->w_r[i];
  for (int i = 0; i < GC; i++) p->bias[i] += -lr * g->bias[i];
  for (int i = 0; i < C; i++) {
    if (!p->cifg) p->peep_i[i] += -lr * g->peep_i[i];
    p->peep_f[i] += -lr * g->peep_f[i];
    p->peep_o[i] += -lr * g->peep_o[i];
  }
  if (R > 0)
    for (size_t i = 0; i < (size_t)R * C; i++) p->w_rm[i] += -lr * g->w_rm[i];
}

/* ---- GruStreams (nnet-gru-streams.h:238-450) ----------------------------------------------------------- */
void orc_gru_forward(const orc_gru *p, const float *in, int ldi, int T, int S, const float *init_state, float *buf) {
  const int H = p->H, D = p->D, W = 5 * H;
  const int oz = 0, orr = H, om = 2 * H, og = 3 * H, oh = 4 * H;
  memset(buf, 0, sizeof(float) * (size_t)(T + 2) * S * W);
  if (init_state) memcpy(buf, init_state, sizeof(float) * (size_t)S * W);
  orc_add_mat_mat(buf + (size_t)S * W, T * S, 3 * H, W, 1.0f, in, ldi, 0, p->w_zrm_x, D, 1, D, 0.0f);
  for (int r = S; r < (T + 1) * S; r++)
    for (int c = 0; c < 3 * H; c++) buf[(size_t)r * W + c] += 1.0f * p->bias[c];
  for (int t = 1; t <= T; t++) {
    float *y = buf + (size_t)t * S * W;
    const float *yp = buf + (size_t)(t - 1) * S * W;
    orc_add_mat_mat(y, S, 2 * H, W, 1.0f, yp + oh, W, 0, p->w_zr_h, H, 1, H, 1.0f);
    for (int s = 0; s < S; s++)
      for (int c = 0; c < H; c++) {
        float *ys = y + (size_t)s * W;
        ys[oz + c] = sigm(ys[oz + c]);
        ys[orr + c] = sigm(ys[orr + c]);
        ys[og + c] = 0.0f * ys[og + c] + 1.0f * ys[orr + c] * yp[(size_t)s * W + oh + c];
      }
    orc_add_mat_mat(y + om, S, H, W, 1.0f, y + og, W, 0, p->w_m_g, H, 1, H, 1.0f);
    for (int s = 0; s < S; s++)
      for (int c = 0; c < H; c++) {
        float *ys = y + (size_t)s * W;
        float hp = yp[(size_t)s * W + oh + c];
        float m = ys[om + c] = tanh_k(ys[om + c]);
        float h = ys[oh + c] + 1.0f * hp;
        h = -1.0f * hp * ys[oz + c] + 1.0f * h;
        h = 1.0f * ys[oz + c] * m + 1.0f * h;
        ys[oh + c] = h;
      }
  }
}

void orc_gru_backward(const orc_gru *p, const float *out_diff, int ldo, int T, int S, const float *buf, float *dbuf, float *in_diff,
                      int ldid) {
  const int H = p->H, D = p->D, W = 5 * H;
  const int oz = 0, orr = H, om = 2 * H, og = 3 * H, oh = 4 * H;
  memset(dbuf, 0, sizeof(float) * (size_t)(T + 2) * S * W);
  for (int r = 0; r < T * S; r++) memcpy(dbuf + (size_t)(S + r) * W + oh, out_diff + (size_t)r * ldo, sizeof(float) * H);
  for (int t = T; t >= 1; t--) {
    float *d = dbuf + (size_t)t * S * W;
    const float *dn = dbuf + (size_t)(t + 1) * S * W;
    const float *y = buf + (size_t)t * S * W, *yn = buf + (size_t)(t + 1) * S * W, *yp = buf + (size_t)(t - 1) * S * W;
    orc_add_mat_mat(d + oh, S, H, W, 1.0f, dn, W, 0, p->w_zr_h, H, 0, 2 * H, 1.0f);
    for (int s = 0; s < S; s++)
      for (int c = 0; c < H; c++) {
        size_t o = (size_t)s * W;
        float dh = d[o + oh + c] + 1.0f * dn[o + oh + c];
        dh = -1.0f * dn[o + oh + c] * yn[o + oz + c] + 1.0f * dh;
        dh = 1.0f * dn[o + og + c] * yn[o + orr + c] + 1.0f * dh;
        d[o + oh + c] = dh;
        d[o + om + c] = dtanh(y[o + om + c], 1.0f * dh * y[o + oz + c]);
      }
    orc_add_mat_mat(d + og, S, H, W, 1.0f, d + om, W, 0, p->w_m_g, H, 0, H, 0.0f);
    for (int s = 0; s < S; s++)
      for (int c = 0; c < H; c++) {
        size_t o = (size_t)s * W;
        float hp = yp[o + oh + c], dh = d[o + oh + c];
        d[o + orr + c] = dsigm(y[o + orr + c], 1.0f * d[o + og + c] * hp);
        float dz = 1.0f * dh * y[o + om + c];
        dz = -1.0f * dh * hp + 1.0f * dz;
        d[o + oz + c] = dsigm(y[o + oz + c], dz);
      }
  }
  orc_add_mat_mat(in_diff, T * S, D, ldid, 1.0f, dbuf + (size_t)S * W, W, 0, p->w_zrm_x, D, 0, 3 * H, 0.0f);
}

void orc_gru_grads(const orc_gru *p, orc_gru *g, const float *in, int ldi, int T, int S, const float *buf, const float *dbuf, float mmt,
                   float clip) {
  const int H = p->H, D = p->D, W = 5 * H;
  const int om = 2 * H, og = 3 * H, oh = 4 * H;
  const float *dZ = dbuf + (size_t)S * W;
  orc_add_mat_mat(g->w_zrm_x, 3 * H, D, D, 1.0f, dZ, W, 1, in, ldi, 0, T * S, mmt);
  for (int c = 0; c < 3 * H; c++) {
    float s = 0.0f;
    for (int r = 0; r < T * S; r++) s += dZ[(size_t)r * W + c];
    g->bias[c] = 1.0f * s + mmt * g->bias[c];
  }
  orc_add_mat_mat(g->w_zr_h, 2 * H, H, H, 1.0f, dZ, W, 1, buf + oh, W, 0, T * S, mmt);
  orc_add_mat_mat(g->w_m_g, H, H, H, 1.0f, dZ + om, W, 1, buf + (size_t)S * W + og, W, 0, T * S, mmt);
  clipv(g->w_zrm_x, (size_t)3 * H * D, clip);
  clipv(g->w_zr_h, (size_t)2 * H * H, clip);
  clipv(g->bias, 3 * H, clip);
  clipv(g->w_m_g, (size_t)H * H, clip);
}

void orc_gru_update(orc_gru *p, const orc_gru *g, float lr) {
  const int H = p->H, D = p->D;
  for (size_t i = 0; i < (size_t)3 * H * D; i++) p->w_zrm_x[i] += -lr * g->w_zrm_x[i];
  for (size_t i = 0; i < (size_t)2 * H * H; i++) p->w_zr_h[i] += -lr * g->w_zr_h[i];
  for (size_t i = 0; i < (size_t)H * H; i++) p->w_m_g[i] += -lr * g->w_m_g[i];
  for (int i = 0; i < 3 * H; i++) p->bias[i] += -lr * g->bias[i];
}
