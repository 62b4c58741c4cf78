// nnet-recurrent.cpp -- LSTM family + GruStreams host logic (see nnet-recurrent.h).
#include "nnet-recurrent.h"

#include <algorithm>

#include "scratch.h"

#include <cstdlib>
#include <sstream>

namespace aslp {

namespace {
void CheckK() {
  char buf[512];
  if (aslp_get_last_error(buf, sizeof(buf))) ASLP_ERR << buf;
}
std::pair<BaseFloat *, int> MatParam(CuMatrix &m) { return std::make_pair(m.Data(), m.NumRows() * m.Stride()); }
std::pair<BaseFloat *, int> VecParam(CuVector &v) { return std::make_pair(v.Data(), v.Dim()); }
}  // namespace

// ---- one LSTM direction ---------------------------------------------------------------------------
void LstmDir::AllocCorr() {
  w_x_corr.Resize(GC(), D, kSetZero);
  w_r_corr.Resize(GC(), Rec(), kSetZero);
  bias_corr.Resize(GC(), kSetZero);
  if (!cifg) peep_i_corr.Resize(C, kSetZero);
  peep_f_corr.Resize(C, kSetZero);
  peep_o_corr.Resize(C, kSetZero);
  if (R > 0) w_rm_corr.Resize(R, C, kSetZero);
}

void LstmDir::InitRandom(float scale) {
  // matrices first, then vectors (e.g. nnet-blstm-projected-streams-lc.h:117-150)
  w_x.Resize(GC(), D, kUndefined);
  w_r.Resize(GC(), Rec(), kUndefined);
  InitMatParamUniform(w_x, scale);
  InitMatParamUniform(w_r, scale);
  if (R > 0) { w_rm.Resize(R, C, kUndefined); InitMatParamUniform(w_rm, scale); }
  bias.Resize(GC(), kUndefined);
  InitVecParamUniform(bias, scale);
  if (!cifg) { peep_i.Resize(C, kUndefined); InitVecParamUniform(peep_i, scale); }
  peep_f.Resize(C, kUndefined);
  peep_o.Resize(C, kUndefined);
  InitVecParamUniform(peep_f, scale);
  InitVecParamUniform(peep_o, scale);
  AllocCorr();
  eff_dirty = true;
}

void LstmDir::Read(std::istream &is, bool binary) {
  w_x.Read(is, binary);
  w_r.Read(is, binary);
  bias.Read(is, binary);
  if (!cifg) peep_i.Read(is, binary);
  peep_f.Read(is, binary);
  peep_o.Read(is, binary);
  if (R > 0) w_rm.Read(is, binary);
  ASLP_ASSERT(w_x.NumRows() == GC() && w_x.NumCols() == D);
  ASLP_ASSERT(w_r.NumRows() == GC() && w_r.NumCols() == Rec());
  ASLP_ASSERT(bias.Dim() == GC());
  ASLP_ASSERT(cifg || peep_i.Dim() == C);
  ASLP_ASSERT(peep_f.Dim() == C && peep_o.Dim() == C);
  if (R > 0) ASLP_ASSERT(w_rm.NumRows() == R && w_rm.NumCols() == C);
  AllocCorr();
  eff_dirty = true;
}

void LstmDir::Write(std::ostream &os, bool binary) const {
  w_x.Write(os, binary);
  w_r.Write(os, binary);
  bias.Write(os, binary);
  if (!cifg) peep_i.Write(os, binary);
  peep_f.Write(os, binary);
  peep_o.Write(os, binary);
  if (R > 0) w_rm.Write(os, binary);
}

int LstmDir::NumParams() const { return GC() * D + GC() * Rec() + GC() + (cifg ? 2 : 3) * C + R * C; }

void LstmDir::AppendParams(std::vector<BaseFloat> *w) const {
  AppendRowMajor(w_x, w);
  AppendRowMajor(w_r, w);
  AppendVector(bias, w);
  if (!cifg) AppendVector(peep_i, w);
  AppendVector(peep_f, w);
  AppendVector(peep_o, w);
  if (R > 0) AppendRowMajor(w_rm, w);
}

void LstmDir::AppendGpuParams(std::vector<std::pair<BaseFloat *, int>> *p) {
  aliased = true;
  p->push_back(MatParam(w_x));
  p->push_back(MatParam(w_r));
  p->push_back(VecParam(bias));
  if (!cifg) p->push_back(VecParam(peep_i));
  p->push_back(VecParam(peep_f));
  p->push_back(VecParam(peep_o));
  if (R > 0) p->push_back(MatParam(w_rm));
}

std::string LstmDir::Info(const char *pre) const {
  std::string s;
  s += std::string("\n  ") + pre + "w_x  " + MomentStatistics(w_x);
  s += std::string("\n  ") + pre + "w_r  " + MomentStatistics(w_r);
  s += std::string("\n  ") + pre + "bias  " + MomentStatistics(bias);
  if (!cifg) s += std::string("\n  ") + pre + "peephole_i_c  " + MomentStatistics(peep_i);
  s += std::string("\n  ") + pre + "peephole_f_c  " + MomentStatistics(peep_f);
  s += std::string("\n  ") + pre + "peephole_o_c  " + MomentStatistics(peep_o);
  if (R > 0) s += std::string("\n  ") + pre + "w_r_m  " + MomentStatistics(w_rm);
  return s;
}

void LstmDir::Forward(const CuMatrixBase &in, int T, int S, bool reverse, const CuMatrixBase *init_state, const CuArray<int32> *seq_len,
                      CuMatrix *buf) const {
  ASLP_ASSERT(in.NumRows() == T * S && in.NumCols() == D);
  buf->Resize((T + 2) * S, Width(), kSetZero);
  if (init_state) buf->RowRange(reverse ? (T + 1) * S : 0, S).CopyFromMat(*init_state);
  {  // x -> gates for every t in one GEMM, bias in its epilogue (lc.h:553-556)
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    ep.bias = bias.Data();
    CuSubMatrix gates(*buf, S, T * S, 0, GC());
    gates.AddMatMat(1.0, in, kNoTrans, w_x, kTrans, 0.0, &ep);
  }
  const int ld = buf->Stride();
  for (int step = 0; step < T; step++) {
    const int t = reverse ? T - step : 1 + step, tp = reverse ? t + 1 : t - 1;
    CuSubMatrix y_gates(*buf, t * S, S, 0, GC());
    CuSubMatrix rec_prev(*buf, tp * S, S, OffRec(), Rec());
    y_gates.AddMatMat(1.0, rec_prev, kNoTrans, w_r, kTrans, 1.0);  // r(t-1) -> gates (lc.h:575)
    aslp_lstm_cell_forward(buf->RowData(t * S), buf->RowData(tp * S), ld, S, C, cifg ? 1 : 0, cifg ? nullptr : peep_i.Data(), peep_f.Data(),
                           peep_o.Data(), seq_len ? seq_len->Data() : nullptr, t);
    if (R > 0) {  // m -> r (lc.h:608); masked rows have m = 0, hence r = 0
      CuSubMatrix y_r(*buf, t * S, S, OffRec(), R), y_m(*buf, t * S, S, OffM(), C);
      y_r.AddMatMat(1.0, y_m, kNoTrans, w_rm, kTrans, 0.0);
    }
  }
  CheckK();
}

void LstmDir::Backward(const CuMatrixBase &out_diff, int T, int S, bool reverse, const CuMatrix &buf, CuMatrix *dbuf, CuMatrixBase *in_diff,
                       float beta) const {
  ASLP_ASSERT(out_diff.NumRows() == T * S && out_diff.NumCols() == Rec());
  dbuf->Resize((T + 2) * S, Width(), kSetZero);
  CuSubMatrix(*dbuf, S, T * S, OffRec(), Rec()).CopyFromMat(out_diff);
  const int ld = dbuf->Stride();
  ASLP_ASSERT(ld == buf.Stride());
  for (int step = 0; step < T; step++) {
    const int t = reverse ? 1 + step : T - step;
    const int tn = reverse ? t - 1 : t + 1, tp = reverse ? t + 1 : t - 1;
    if (step > 0) {  // d_rec(t) += dGATES(next) * w_r (lc.h:791); at the first step the next block is all zero
      CuSubMatrix d_rec(*dbuf, t * S, S, OffRec(), Rec()), dn_gates(*dbuf, tn * S, S, 0, GC());
      d_rec.AddMatMat(1.0, dn_gates, kNoTrans, w_r, kNoTrans, 1.0);
    }
    if (R > 0) {  // d_m = d_r * w_r_m (lc.h:793)
      CuSubMatrix d_m(*dbuf, t * S, S, OffM(), C), d_r(*dbuf, t * S, S, OffRec(), R);
      d_m.AddMatMat(1.0, d_r, kNoTrans, w_rm, kNoTrans, 0.0);
    }
    aslp_lstm_cell_backward(dbuf->RowData(t * S), dbuf->RowData(tn * S), buf.RowData(t * S), buf.RowData(tn * S), buf.RowData(tp * S), ld, S, C,
                            cifg ? 1 : 0, cifg ? nullptr : peep_i.Data(), peep_f.Data(), peep_o.Data());
  }
  CheckK();
  if (in_diff) {
    CuSubMatrix d_gates(*dbuf, S, T * S, 0, GC());
    in_diff->AddMatMat(1.0, d_gates, kNoTrans, w_x, kNoTrans, beta);
  }
}

void LstmDir::VecGrads(int T, int S, bool reverse, const CuMatrix &buf, const CuMatrix &dbuf, float mmt, float clip, float lr_fold,
                       const aslp_lstm_seq *seq, int dir) {
  // bias and peephole gradients (:1005-1058): column sums of d_gates and of d_{i,f,o} .* c.  The persistent backward kernel has the
  // addends in registers and leaves per-chain sums (one small launch finishes them); otherwise one pass over the diff buffer.
  if (seq != nullptr && seq->grad_partial != nullptr) {
    aslp_lstm_seq_vec_grads(seq, dir, bias_corr.Data(), bias.Data(), cifg ? nullptr : peep_i_corr.Data(), cifg ? nullptr : peep_i.Data(),
                            peep_f_corr.Data(), peep_f.Data(), peep_o_corr.Data(), peep_o.Data(), mmt, clip, -lr_fold);
  } else {
    aslp_rnn_vec_grad jobs[4];
    const int n = VecGradJobs(S, reverse, buf, dbuf, jobs);
    aslp_rnn_vec_grads(jobs, n, dbuf.Stride(), T * S, mmt, clip, -lr_fold);
  }
}

void LstmDir::Grads(const CuMatrixBase &in, int T, int S, bool reverse, const CuMatrix &buf, const CuMatrix &dbuf, float mmt, float clip, float lr_fold,
                    const aslp_lstm_seq *seq, int dir) {
  // lc.h:976-1058: corr = grad + mmt * corr, then clip element-wise (the clip rides in the GEMM epilogue, and with it --
  // when the executor announced that Update follows -- the step W += -lr * corr of lc.h:1085-1110)
  const int prev0 = (reverse ? 2 : 0) * S;  // recursion-previous row block of t = 1
  CuSubMatrix d_gates(dbuf, S, T * S, 0, GC());
  auto wgrad = [&](CuMatrix &corr, CuMatrix &w, const CuMatrixBase &d, const CuMatrixBase &x) {
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    ep.clip = clip;
    if (lr_fold != 0.0f) { ep.W = w.Data(); ep.ldw = w.Stride(); ep.w_alpha = -lr_fold; }
    corr.AddMatMat(1.0, d, kTrans, x, kNoTrans, mmt, &ep);
  };
  wgrad(w_x_corr, w_x, d_gates, in);
  wgrad(w_r_corr, w_r, d_gates, CuSubMatrix(buf, prev0, T * S, OffRec(), Rec()));
  if (R > 0) wgrad(w_rm_corr, w_rm, CuSubMatrix(dbuf, S, T * S, OffRec(), R), CuSubMatrix(buf, S, T * S, OffM(), C));
  VecGrads(T, S, reverse, buf, dbuf, mmt, clip, lr_fold, seq, dir);
  CheckK();
  if (lr_fold != 0.0f) eff_dirty = true;
}

int LstmDir::VecGradJobs(int S, bool reverse, const CuMatrix &buf, const CuMatrix &dbuf, aslp_rnn_vec_grad *jobs) {
  const int prev0 = (reverse ? 2 : 0) * S;
  const BaseFloat *d0 = dbuf.RowData(S), *c_prev = buf.RowData(prev0) + OffC(), *c_cur = buf.RowData(S) + OffC();
  int n = 0;
  jobs[n++] = {d0, nullptr, 0, GC(), bias_corr.Data(), bias.Data()};
  if (!cifg) jobs[n++] = {d0 + OffI(), c_prev, buf.Stride(), C, peep_i_corr.Data(), peep_i.Data()};
  jobs[n++] = {d0 + OffF(), c_prev, buf.Stride(), C, peep_f_corr.Data(), peep_f.Data()};
  jobs[n++] = {d0 + OffO(), c_cur, buf.Stride(), C, peep_o_corr.Data(), peep_o.Data()};
  return n;
}

void LstmDir::Update(float lr) {  // lc.h:1085-1110
  w_x.AddMat(-lr, w_x_corr);
  w_r.AddMat(-lr, w_r_corr);
  bias.AddVec(-lr, bias_corr, 1.0);
  if (!cifg) peep_i.AddVec(-lr, peep_i_corr, 1.0);
  peep_f.AddVec(-lr, peep_f_corr, 1.0);
  peep_o.AddVec(-lr, peep_o_corr, 1.0);
  if (R > 0) w_rm.AddMat(-lr, w_rm_corr);
  eff_dirty = true;
}

// ---- fused-step path ------------------------------------------------------------------------------------
bool LstmDir::FusedOk() const {
  static const bool disabled = getenv("ASLP_LSTM_UNFUSED") != nullptr && getenv("ASLP_LSTM_UNFUSED")[0] == '1';
  return !disabled && C % 4 == 0;
}
void LstmDir::RefreshEff() const {
  if (!eff_dirty && !aliased) return;
  if (R > 0) {
    if (w_eff.NumRows() != GC() || w_eff.NumCols() != C) w_eff.Resize(GC(), C, kUndefined);
    w_eff.AddMatMat(1.0, w_r, kNoTrans, w_rm, kNoTrans, 0.0);
  }
  eff_dirty = false;
  eff_t_dirty = true;
}
// W_eff^T: only the one-launch-per-timestep backward path multiplies with it (the persistent kernel takes rows of W_eff)
void LstmDir::RefreshEffT() const {
  if (!eff_t_dirty && !aliased && w_eff_t.NumRows() == C) return;
  if (w_eff_t.NumRows() != C || w_eff_t.NumCols() != GC()) w_eff_t.Resize(C, GC(), kUndefined);
  if (R > 0) {
    w_eff_t.AddMatMat(1.0, w_rm, kTrans, w_r, kTrans, 0.0);
  } else {
    w_eff_t.CopyFromMatTrans(w_r);
  }
  eff_t_dirty = false;
}

void LstmDir::ForwardPrepare(const CuMatrixBase &in, int T, int S, bool reverse, const CuMatrixBase *init_state, CuMatrix *buf,
                             bool persistent, bool with_gemm) const {
  ASLP_ASSERT(in.NumRows() == T * S && in.NumCols() == D);
  if (with_gemm) RefreshEff();
  if (persistent) {  // row blocks 1..T start as "not yet published" (csrc/rnn_persistent.hip), the two boundary blocks as zero
    buf->Resize((T + 2) * S, Width(), kUndefined);
    aslp_lstm_seq_fill(buf->Data(), buf->Stride(), T, S, OffM(), C);
  } else {
    buf->Resize((T + 2) * S, Width(), kSetZero);
  }
  if (init_state) buf->RowRange(reverse ? (T + 1) * S : 0, S).CopyFromMat(*init_state);
  if (!with_gemm) return;
  aslp_gemm_epilogue ep = aslp_gemm_epilogue();
  ep.bias = bias.Data();
  CuSubMatrix gates(*buf, S, T * S, 0, GC());
  gates.AddMatMat(1.0, in, kNoTrans, w_x, kTrans, 0.0, &ep);
}

bool LstmDir::ForwardFinish(int T, int S, CuMatrix *buf, CuMatrixBase *out, int out_col) const {
  if (R <= 0) return false;
  CuSubMatrix y_r(*buf, S, T * S, OffRec(), R), y_m(*buf, S, T * S, OffM(), C);
  aslp_gemm_epilogue ep = aslp_gemm_epilogue();
  if (out) { ep.act_out = out->Data() + out_col; ep.ld_act = out->Stride(); ep.act = 0; }  // second store: the component's output block
  y_r.AddMatMat(1.0, y_m, kNoTrans, w_rm, kTrans, 0.0, &ep);  // m -> r for every t at once (lc.h:608)
  return out != nullptr;
}

void LstmDir::BackwardPrepare(const CuMatrixBase &out_diff, int T, int S, CuMatrix *dbuf, bool persistent, bool with_gemm) const {
  ASLP_ASSERT(out_diff.NumRows() == T * S && out_diff.NumCols() == Rec());
  if (persistent) {
    // the persistent backward kernel writes every gate / c / h / m entry of row blocks 1..T itself (d_r follows in BackwardFinish):
    // only the two boundary row blocks have to be zero -- 1 MB instead of a 29 MB memset per direction and layer
    dbuf->Resize((T + 2) * S, Width(), kUndefined);
    aslp_lstm_seq_fill(dbuf->Data(), dbuf->Stride(), T, S, 0, 0);
  } else {
    dbuf->Resize((T + 2) * S, Width(), kSetZero);
  }
  if (!with_gemm) return;
  CuSubMatrix d_m(*dbuf, S, T * S, OffM(), C);
  if (R > 0) d_m.AddMatMat(1.0, out_diff, kNoTrans, w_rm, kNoTrans, 0.0);  // the loss's share of d_m, all t at once
  else d_m.CopyFromMat(out_diff);
}

void LstmDir::BackwardFinish(const CuMatrixBase &out_diff, int T, int S, bool reverse, CuMatrix *dbuf, CuMatrixBase *in_diff, float beta,
                             bool with_dr) const {
  if (R > 0 && with_dr) {  // d_r(t) = out_diff(t) + dGATES(next) W_r (lc.h:791), needed by the W_rm gradient
    CuSubMatrix d_r(*dbuf, S, T * S, OffRec(), R);
    d_r.CopyFromMat(out_diff);
    CuSubMatrix d_gates_next(*dbuf, (reverse ? 0 : 2) * S, T * S, 0, GC());  // row block of step t's recursion-next
    d_r.AddMatMat(1.0, d_gates_next, kNoTrans, w_r, kNoTrans, 1.0);
  }
  if (in_diff) {
    CuSubMatrix d_gates(*dbuf, S, T * S, 0, GC());
    in_diff->AddMatMat(1.0, d_gates, kNoTrans, w_x, kNoTrans, beta);
  }
}

// ---- both directions at once ------------------------------------------------------------------------
namespace {
bool PairsOn() {
  static const bool off = getenv("ASLP_LSTM_PAIR") != nullptr && getenv("ASLP_LSTM_PAIR")[0] == '0';  // A/B switch
  return !off;
}
bool SameShape(const LstmDir &f, const LstmDir &b) { return PairsOn() && f.D == b.D && f.C == b.C && f.R == b.R && f.cifg == b.cifg; }
// Do the layer's batched products run on the fp16 instruction from prepared planes?  Every reduction extent they meet (T S, C, R, 4C)
// must be a multiple of 64 -- the planes of a buffer's column block or row range are windows, with neighbours instead of zero padding.
// Switch ASLP_LSTM_PLANES: 0 = the layer's batched products convert their operands call by call or run on the fp32 instruction as
// aslp_sgemm_pair_ex decides (pairs: the fp32 instruction), 1 = the forward products from planes the layer prepares, 2 = + the products in
// front of the backward recurrence, 3 (default) = all of them, 4 = as 3 but the weight gradients on the fp32 instruction where they are
// issued beside the recurrence below.  Measured at the end of round 4 (one box each, two alternations): devtools/bench_lc.py 32 30 (output
// layer 3000 wide) 2.97 / 3.00 ms with 0, 2.98 / 2.99 with 1, 3.13 / 3.08 with 2, 2.86 / 2.85 with 3, 3.07 / 3.07 with 4; bench.py's cfg3
// block: chunked 2.750 ms with 0 and 2.752-2.759 with 3, whole utterances + Warp-CTC 34.2 ms with 0 and **30.6 with 3** (the products of
// 25,600 rows are what the long step is made of).  Earlier in the round, before the product kernels lost their private segment, 3 and 0
// were level on the chunked step (2.872 / 2.858), which is why it shipped off for a while.
int PlanesLevel() {
  static const int level = [] { const char *e = getenv("ASLP_LSTM_PLANES"); return e ? atoi(e) : 3; }();
  return level;
}
bool PlanesUsable(const LstmDir &f, int T, int S) {
  return PlanesLevel() > 0 && gemm_split16_enabled() && f.R > 0 && (T * S) % 64 == 0 && f.C % 64 == 0 && f.R % 64 == 0 && T * S >= 128;
}
// The buffer preparation of the persistent kernels (aslp_lstm_seq_fill_pair) as a job for the conversion launch that follows it (split16.h
// SeqFillJob): one launch less per layer and pass (LC-BLSTM step: 8 of ~100).  The checks are aslp_lstm_seq_fill_pair's own; a job that
// fails them goes to that function, which names the problem.  A/B: ASLP_LSTM_FILL_ALONG=0.
bool FillRidesAlong(const SeqFillJob &j) {
  static const bool off = [] { const char *e = getenv("ASLP_LSTM_FILL_ALONG"); return e != nullptr && e[0] == '0'; }();
  auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (off || !j.buf0 || !j.buf1 || j.T <= 0 || j.S <= 0 || j.ld <= 0) return false;
  if ((j.ld & 3) || (j.col0 & 3) || (j.ncols & 3) || j.col0 < 0 || j.col0 + j.ncols > j.ld || !al(j.buf0) || !al(j.buf1)) return false;
  if (j.init && ((j.ld_init & 3) || (j.init_cols & 3) || j.init_cols > j.ld || j.init_cols > j.ld_init || !al(j.init))) return false;
  return true;
}
PlaneSet::ConvertSpec Spec(PlaneSet *ps, const CuMatrixBase &m) { return PlaneSet::ConvertSpec{ps, m.Data(), m.NumRows(), m.NumCols(), m.Stride()}; }
}  // namespace

void LstmDir::RefreshEffPair(const LstmDir &f, const LstmDir &b, const S16View *views) {
  const bool stale_f = f.eff_dirty || f.aliased, stale_b = b.eff_dirty || b.aliased;
  if (!(stale_f && stale_b && f.R > 0 && SameShape(f, b))) { f.RefreshEff(); b.RefreshEff(); return; }
  for (const LstmDir *p : {&f, &b})
    if (p->w_eff.NumRows() != p->GC() || p->w_eff.NumCols() != p->C) p->w_eff.Resize(p->GC(), p->C, kUndefined);
  AddMatMatPair(f.w_eff, b.w_eff, 1.0, f.w_r, b.w_r, kNoTrans, f.w_rm, b.w_rm, kNoTrans, 0.0, nullptr, nullptr, views);
  f.eff_dirty = b.eff_dirty = false;
  f.eff_t_dirty = b.eff_t_dirty = true;
}

void LstmDir::ForwardPreparePair(const LstmDir &f, const LstmDir &b, const CuMatrixBase &in, int T, int S, const CuMatrixBase *init_f,
                                 CuMatrix *fbuf, CuMatrix *bbuf, bool persistent, LstmPlanes *pl) {
  if (pl) pl->weights_ok = pl->in_ok = pl->m_ok = pl->od_ok = pl->dg_ok = false;
  if (!SameShape(f, b)) {
    f.ForwardPrepare(in, T, S, false, init_f, fbuf, persistent);
    b.ForwardPrepare(in, T, S, true, nullptr, bbuf, persistent);
    return;
  }
  // (with prepared planes W_eff = W_r W_rm waits for this step's conversion of the weights below and reads their planes)
  const bool eff_from_planes = pl && persistent && PlanesUsable(f, T, S) && PlanesLevel() >= 3;
  if (!eff_from_planes) RefreshEffPair(f, b);
  SeqFillJob fill = {};
  bool fill_pending = false;
  if (persistent && f.Width() % 4 == 0 && (!init_f || init_f->Stride() % 4 == 0)) {
    // both buffers in one launch: boundary row blocks zero (the forward direction's history block takes the carried state),
    // the m columns of row blocks 1..T "not yet published" (csrc/rnn_persistent.hip)
    ASLP_ASSERT(in.NumRows() == T * S && in.NumCols() == f.D);
    fbuf->Resize((T + 2) * S, f.Width(), kUndefined);
    bbuf->Resize((T + 2) * S, b.Width(), kUndefined);
    ASLP_ASSERT(fbuf->Stride() == bbuf->Stride());
    if (init_f) ASLP_ASSERT(init_f->NumRows() == S && init_f->NumCols() == f.Width());
    // (issued below: the conversion launch of this step takes it along where there is one)
    fill = SeqFillJob{fbuf->Data(), bbuf->Data(), fbuf->Stride(), T, S, f.OffM(), f.C, init_f ? init_f->Data() : nullptr,
                      init_f ? init_f->Stride() : 0, f.Width()};
    fill_pending = true;
  } else {
    f.ForwardPrepare(in, T, S, false, init_f, fbuf, persistent, false);
    b.ForwardPrepare(in, T, S, true, nullptr, bbuf, persistent, false);
  }
  aslp_gemm_epilogue ep_f = aslp_gemm_epilogue(), ep_b = aslp_gemm_epilogue();
  ep_f.bias = f.bias.Data();
  ep_b.bias = b.bias.Data();
  CuSubMatrix gates_f(*fbuf, S, T * S, 0, f.GC()), gates_b(*bbuf, S, T * S, 0, b.GC());
  if (pl && persistent && PlanesUsable(f, T, S)) {
    // one maximum + one conversion launch for the layer input and the six weight matrices of this step (the weights' planes also serve
    // the backward pass: the weights move only in GradsPair, after the last product that reads them)
    const bool in_ok = f.D % 64 == 0;   // (the first layer's 40-wide input stays below the split kernels' floor)
    PlaneSet::ConvertSpec sp[7] = {Spec(&pl->wrm[0], f.w_rm), Spec(&pl->wrm[1], b.w_rm), Spec(&pl->wr[0], f.w_r), Spec(&pl->wr[1], b.w_r),
                                   Spec(&pl->wx[0], f.w_x), Spec(&pl->wx[1], b.w_x), Spec(&pl->in, in)};
    bool filled = false;
    pl->weights_ok = PlaneSet::ConvertMany(sp, in_ok ? 7 : 4, fill_pending && FillRidesAlong(fill) ? &fill : nullptr, &filled);
    pl->in_ok = pl->weights_ok && in_ok;
    if (filled) fill_pending = false;
  }
  if (fill_pending)
    aslp_lstm_seq_fill_pair(fill.buf0, fill.buf1, fill.ld, fill.T, fill.S, fill.col0, fill.ncols, fill.init, fill.ld_init, fill.init_cols);
  if (eff_from_planes) {
    const S16View v[4] = {pl->wr[0].View(), pl->wr[1].View(), pl->wrm[0].View(), pl->wrm[1].View()};
    RefreshEffPair(f, b, pl->weights_ok ? v : nullptr);
  }
  if (pl && pl->in_ok) {
    const S16View v[4] = {pl->in.View(), pl->in.View(), pl->wx[0].View(), pl->wx[1].View()};
    AddMatMatPair(gates_f, gates_b, 1.0, in, in, kNoTrans, f.w_x, b.w_x, kTrans, 0.0, &ep_f, &ep_b, v);
  } else {
    AddMatMatPair(gates_f, gates_b, 1.0, in, in, kNoTrans, f.w_x, b.w_x, kTrans, 0.0, &ep_f, &ep_b);
  }
}

bool LstmDir::ForwardFinishPair(const LstmDir &f, const LstmDir &b, int T, int S, CuMatrix *fbuf, CuMatrix *bbuf, CuMatrixBase *out, LstmPlanes *pl) {
  if (f.R <= 0 || !SameShape(f, b)) {
    bool w = f.ForwardFinish(T, S, fbuf, out, 0);
    return b.ForwardFinish(T, S, bbuf, out, f.Rec()) && w;
  }
  CuSubMatrix r_f(*fbuf, S, T * S, f.OffRec(), f.R), m_f(*fbuf, S, T * S, f.OffM(), f.C);
  CuSubMatrix r_b(*bbuf, S, T * S, b.OffRec(), b.R), m_b(*bbuf, S, T * S, b.OffM(), b.C);
  aslp_gemm_epilogue ep_f = aslp_gemm_epilogue(), ep_b = aslp_gemm_epilogue();
  if (out) {  // second store: the component's output block [r_f | r_b]
    ep_f.act_out = out->Data(); ep_f.ld_act = out->Stride(); ep_f.act = 0;
    ep_b.act_out = out->Data() + f.Rec(); ep_b.ld_act = out->Stride(); ep_b.act = 0;
  }
  if (pl && pl->weights_ok) {   // m of both directions: also the W_rm gradient's operand.  m = o tanh(c): |m| < 1, no maximum pass
    PlaneSet::ConvertSpec sp[2] = {Spec(&pl->m[0], m_f), Spec(&pl->m[1], m_b)};
    static const bool known_off = getenv("ASLP_LSTM_KNOWN_BOUNDS") != nullptr && getenv("ASLP_LSTM_KNOWN_BOUNDS")[0] == '0';   // A/B switch
    if (!known_off && PlaneSet::OneBound() != nullptr)
      for (auto &c : sp) { c.parts = PlaneSet::OneBound(); c.nparts = 1; }
    pl->m_ok = PlaneSet::ConvertMany(sp, 2);
  }
  if (pl && pl->m_ok) {
    const S16View v[4] = {pl->m[0].View(), pl->m[1].View(), pl->wrm[0].View(), pl->wrm[1].View()};
    AddMatMatPair(r_f, r_b, 1.0, m_f, m_b, kNoTrans, f.w_rm, b.w_rm, kTrans, 0.0, &ep_f, &ep_b, v);
  } else {
    AddMatMatPair(r_f, r_b, 1.0, m_f, m_b, kNoTrans, f.w_rm, b.w_rm, kTrans, 0.0, &ep_f, &ep_b);  // m -> r for every t at once (lc.h:608)
  }
  return out != nullptr;
}

void LstmDir::BackwardPreparePair(const LstmDir &f, const LstmDir &b, const CuMatrixBase &od_f, const CuMatrixBase &od_b, int T, int S,
                                  CuMatrix *fdbuf, CuMatrix *bdbuf, bool persistent, LstmPlanes *pl) {
  if (f.R <= 0 || !SameShape(f, b)) {
    f.BackwardPrepare(od_f, T, S, fdbuf, persistent);
    b.BackwardPrepare(od_b, T, S, bdbuf, persistent);
    return;
  }
  SeqFillJob fill = {};
  bool fill_pending = false;
  if (persistent) {   // only the two boundary row blocks of each buffer have to be zero (BackwardPrepare)
    fdbuf->Resize((T + 2) * S, f.Width(), kUndefined);
    bdbuf->Resize((T + 2) * S, b.Width(), kUndefined);
    ASLP_ASSERT(fdbuf->Stride() == bdbuf->Stride());
    fill = SeqFillJob{fdbuf->Data(), bdbuf->Data(), fdbuf->Stride(), T, S, 0, 0, nullptr, 0, 0};
    fill_pending = true;
  } else {
    f.BackwardPrepare(od_f, T, S, fdbuf, persistent, false);
    b.BackwardPrepare(od_b, T, S, bdbuf, persistent, false);
  }
  CuSubMatrix dm_f(*fdbuf, S, T * S, f.OffM(), f.C), dm_b(*bdbuf, S, T * S, b.OffM(), b.C);
  if (pl && pl->weights_ok && persistent && PlanesLevel() >= 2) {   // (the weights' planes of this step's forward pass: the weights have not moved)
    PlaneSet::ConvertSpec sp[2] = {Spec(&pl->od[0], od_f), Spec(&pl->od[1], od_b)};
    bool filled = false;
    pl->od_ok = PlaneSet::ConvertMany(sp, 2, fill_pending && FillRidesAlong(fill) ? &fill : nullptr, &filled);
    if (filled) fill_pending = false;
  }
  if (fill_pending)
    aslp_lstm_seq_fill_pair(fill.buf0, fill.buf1, fill.ld, fill.T, fill.S, fill.col0, fill.ncols, fill.init, fill.ld_init, fill.init_cols);
  if (pl && pl->od_ok) {
    const S16View v[4] = {pl->od[0].View(), pl->od[1].View(), pl->wrm[0].View(), pl->wrm[1].View()};
    AddMatMatPair(dm_f, dm_b, 1.0, od_f, od_b, kNoTrans, f.w_rm, b.w_rm, kNoTrans, 0.0, nullptr, nullptr, v);
  } else {
    AddMatMatPair(dm_f, dm_b, 1.0, od_f, od_b, kNoTrans, f.w_rm, b.w_rm, kNoTrans, 0.0);  // the loss's share of d_m, all t at once
  }
}

// d_r(t) = out_diff(t) + dGATES(next) W_r (lc.h:791), needed by the W_rm gradient only
// (out_diff enters as the epilogue's beta term read from its own matrix: no copy into d_r first)
void LstmDir::BackwardDrPair(const LstmDir &f, const LstmDir &b, const CuMatrixBase &od_f, const CuMatrixBase &od_b, int T, int S, CuMatrix *fdbuf,
                             CuMatrix *bdbuf, LstmPlanes *pl) {
  CuSubMatrix dr_f(*fdbuf, S, T * S, f.OffRec(), f.R), dr_b(*bdbuf, S, T * S, b.OffRec(), b.R);
  CuSubMatrix next_f(*fdbuf, 2 * S, T * S, 0, f.GC()), next_b(*bdbuf, 0, T * S, 0, b.GC());  // row blocks of each step's recursion-next
  aslp_gemm_epilogue ep_f = aslp_gemm_epilogue(), ep_b = aslp_gemm_epilogue();
  ep_f.c_src = od_f.Data(); ep_f.ld_c_src = od_f.Stride();
  ep_b.c_src = od_b.Data(); ep_b.ld_c_src = od_b.Stride();
  if (pl && pl->dg_ok) {
    const S16View v[4] = {pl->dg[0].Window(2 * S, T * S, 0, f.GC()), pl->dg[1].Window(0, T * S, 0, b.GC()), pl->wr[0].View(), pl->wr[1].View()};
    AddMatMatPair(dr_f, dr_b, 1.0, next_f, next_b, kNoTrans, f.w_r, b.w_r, kNoTrans, 1.0, &ep_f, &ep_b, v);
  } else {
    AddMatMatPair(dr_f, dr_b, 1.0, next_f, next_b, kNoTrans, f.w_r, b.w_r, kNoTrans, 1.0, &ep_f, &ep_b);
  }
}

bool LstmDir::BackwardFinishPair(const LstmDir &f, const LstmDir &b, const CuMatrixBase &od_f, const CuMatrixBase &od_b, int T, int S,
                                 CuMatrix *fdbuf, CuMatrix *bdbuf, CuMatrixBase *in_diff, LstmPlanes *pl, bool with_dr) {
  if (f.R <= 0 || !SameShape(f, b)) {
    f.BackwardFinish(od_f, T, S, false, fdbuf, in_diff, 0.0, true);
    b.BackwardFinish(od_b, T, S, true, bdbuf, in_diff, 1.0, true);
    return false;
  }
  if (pl && pl->od_ok && PlanesLevel() >= 3) {   // the dGATES columns of both diff buffers, boundary row blocks included: the products below read shifted row ranges of them
    CuSubMatrix dga_f(*fdbuf, 0, (T + 2) * S, 0, f.GC()), dga_b(*bdbuf, 0, (T + 2) * S, 0, b.GC());
    PlaneSet::ConvertSpec sp[2] = {Spec(&pl->dg[0], dga_f), Spec(&pl->dg[1], dga_b)};
    if (pl->dg_parts[0] != nullptr && pl->dg_nparts > 0) {   // the persistent backward launch left the gate diffs' per-workgroup maxima
      sp[0].parts = pl->dg_parts[0]; sp[1].parts = pl->dg_parts[1];
      sp[0].nparts = sp[1].nparts = pl->dg_nparts;
    }
    pl->dg_ok = PlaneSet::ConvertMany(sp, 2);
  }
  if (with_dr) BackwardDrPair(f, b, od_f, od_b, T, S, fdbuf, bdbuf, pl);
  if (!in_diff) return !with_dr;
  // in_diff = dGATES_f W_x,f + dGATES_b W_x,b.  Two products into one output cannot share a launch; as a pair into (in_diff, scratch)
  // followed by one addition they can, and the [T*S x D] output alone does not fill the chip (240 tiles for D = 512).  The sum is
  // rounded once either way (the accumulating epilogue computes acc + C as one rounded addition, gemm_common.h).
  CuSubMatrix dg_f(*fdbuf, S, T * S, 0, f.GC()), dg_b(*bdbuf, S, T * S, 0, b.GC());
  static thread_local CuMatrix scratch;
  if (scratch.NumRows() != in_diff->NumRows() || scratch.NumCols() != in_diff->NumCols() || scratch.Stride() != in_diff->Stride())
    scratch.Resize(in_diff->NumRows(), in_diff->NumCols(), kUndefined);
  if (scratch.Stride() != in_diff->Stride()) {   // (a view with a foreign stride: keep the two accumulating launches)
    f.BackwardFinish(od_f, T, S, false, fdbuf, in_diff, 0.0, false);
    b.BackwardFinish(od_b, T, S, true, bdbuf, in_diff, 1.0, false);
    return !with_dr;
  }
  if (pl && pl->dg_ok && pl->in_ok) {
    const S16View v[4] = {pl->dg[0].Window(S, T * S, 0, f.GC()), pl->dg[1].Window(S, T * S, 0, b.GC()), pl->wx[0].View(), pl->wx[1].View()};
    AddMatMatPair(*in_diff, scratch, 1.0, dg_f, dg_b, kNoTrans, f.w_x, b.w_x, kNoTrans, 0.0, nullptr, nullptr, v);
  } else {
    AddMatMatPair(*in_diff, scratch, 1.0, dg_f, dg_b, kNoTrans, f.w_x, b.w_x, kNoTrans, 0.0);
  }
  in_diff->AddMat(1.0, scratch);
  return !with_dr;
}

void LstmDir::GradsPair(LstmDir &f, LstmDir &b, const CuMatrixBase &in, int T, int S, const CuMatrix &fbuf, const CuMatrix &bbuf,
                        const CuMatrix &fdbuf, const CuMatrix &bdbuf, float mmt, float clip, float lr_fold, const aslp_lstm_seq *seq, LstmPlanes *pl) {
  if (!SameShape(f, b)) {
    f.Grads(in, T, S, false, fbuf, fdbuf, mmt, clip, lr_fold, seq, 0);
    b.Grads(in, T, S, true, bbuf, bdbuf, mmt, clip, lr_fold, seq, 1);
    return;
  }
  // lc.h:976-1058, as in Grads(): corr = grad + mmt * corr, clipped, and (folded) W += -lr * corr
  auto wgrad = [&](CuMatrix &corr_f, CuMatrix &corr_b, CuMatrix &w_f, CuMatrix &w_b, const CuMatrixBase &d_f, const CuMatrixBase &d_b,
                   const CuMatrixBase &x_f, const CuMatrixBase &x_b, const S16View *views) {
    aslp_gemm_epilogue ep_f = aslp_gemm_epilogue(), ep_b = aslp_gemm_epilogue();
    ep_f.clip = ep_b.clip = clip;
    if (lr_fold != 0.0f) {
      ep_f.W = w_f.Data(); ep_f.ldw = w_f.Stride(); ep_f.w_alpha = -lr_fold;
      ep_b.W = w_b.Data(); ep_b.ldw = w_b.Stride(); ep_b.w_alpha = -lr_fold;
    }
    AddMatMatPair(corr_f, corr_b, 1.0, d_f, d_b, kTrans, x_f, x_b, kNoTrans, mmt, &ep_f, &ep_b, views);
  };
  CuSubMatrix dg_f(fdbuf, S, T * S, 0, f.GC()), dg_b(bdbuf, S, T * S, 0, b.GC());
  CuSubMatrix dr_f(fdbuf, S, T * S, f.OffRec(), f.R > 0 ? f.R : f.C), dr_b(bdbuf, S, T * S, b.OffRec(), b.R > 0 ? b.R : b.C);
  // planes: dGATES (made for the in-diff products) and m (made for the projection) are there; r (whole column block: the operand is a
  // shifted row range of it) and d_r come with one more conversion launch pair
  // (level 4: the weight gradients stay on the fp32-instruction kernels when they are issued beside the recurrence below -- those fit on a CU next
  // to a persistent workgroup (128 registers per wave, 73 KB of LDS), the split kernels (340 registers, 144 KB) wait until it has left)
  bool side_ok = false;
  if (pl && pl->dg_ok && pl->m_ok && f.R > 0 && !(PlanesLevel() == 4 && on_side_stream())) {
    CuSubMatrix ra_f(fbuf, 0, (T + 2) * S, f.OffRec(), f.R), ra_b(bbuf, 0, (T + 2) * S, b.OffRec(), b.R);
    PlaneSet::ConvertSpec sp[4] = {Spec(&pl->r[0], ra_f), Spec(&pl->r[1], ra_b), Spec(&pl->dr[0], dr_f), Spec(&pl->dr[1], dr_b)};
    side_ok = PlaneSet::ConvertMany(sp, 4);
  }
  const S16View dgv[2] = {side_ok ? pl->dg[0].Window(S, T * S, 0, f.GC()) : S16View(), side_ok ? pl->dg[1].Window(S, T * S, 0, b.GC()) : S16View()};
  if (side_ok && pl->in_ok) {
    const S16View v[4] = {dgv[0], dgv[1], pl->in.View(), pl->in.View()};
    wgrad(f.w_x_corr, b.w_x_corr, f.w_x, b.w_x, dg_f, dg_b, in, in, v);
  } else {
    wgrad(f.w_x_corr, b.w_x_corr, f.w_x, b.w_x, dg_f, dg_b, in, in, nullptr);
  }
  {
    const S16View v[4] = {dgv[0], dgv[1], side_ok ? pl->r[0].Window(0, T * S, 0, f.R) : S16View(), side_ok ? pl->r[1].Window(2 * S, T * S, 0, b.R) : S16View()};
    wgrad(f.w_r_corr, b.w_r_corr, f.w_r, b.w_r, dg_f, dg_b, CuSubMatrix(fbuf, 0, T * S, f.OffRec(), f.Rec()),
          CuSubMatrix(bbuf, 2 * S, T * S, b.OffRec(), b.Rec()), side_ok ? v : nullptr);
  }
  if (f.R > 0) {
    const S16View v[4] = {side_ok ? pl->dr[0].View() : S16View(), side_ok ? pl->dr[1].View() : S16View(), side_ok ? pl->m[0].View() : S16View(),
                          side_ok ? pl->m[1].View() : S16View()};
    wgrad(f.w_rm_corr, b.w_rm_corr, f.w_rm, b.w_rm, dr_f, dr_b, CuSubMatrix(fbuf, S, T * S, f.OffM(), f.C), CuSubMatrix(bbuf, S, T * S, b.OffM(), b.C),
          side_ok ? v : nullptr);
  }
  if (seq != nullptr && seq->grad_partial != nullptr) {   // the persistent backward launch left the sums: one small finishing launch
    auto vec8 = [](LstmDir &p, float **v) {
      v[0] = p.bias_corr.Data(); v[1] = p.bias.Data();
      v[2] = p.cifg ? nullptr : p.peep_i_corr.Data(); v[3] = p.cifg ? nullptr : p.peep_i.Data();
      v[4] = p.peep_f_corr.Data(); v[5] = p.peep_f.Data(); v[6] = p.peep_o_corr.Data(); v[7] = p.peep_o.Data();
    };
    float *vf[8], *vb[8];
    vec8(f, vf);
    vec8(b, vb);
    aslp_lstm_seq_vec_grads2(seq, vf, vb, mmt, clip, -lr_fold);
  } else {   // bias and peephole gradients of both directions: one launch over the two diff buffers
    aslp_rnn_vec_grad jobs[8];
    int n = f.VecGradJobs(S, false, fbuf, fdbuf, jobs);
    n += b.VecGradJobs(S, true, bbuf, bdbuf, jobs + n);
    ASLP_ASSERT(fdbuf.Stride() == bdbuf.Stride());
    aslp_rnn_vec_grads(jobs, n, fdbuf.Stride(), T * S, mmt, clip, -lr_fold);
  }
  CheckK();
  if (lr_fold != 0.0f) f.eff_dirty = b.eff_dirty = true;
}

// ---- the component family ---------------------------------------------------------------------------
LstmFamily::LstmFamily(int32 di, int32 dout, const Config &cfg)
    : RecurrentBase(di, dout), cfg_(cfg), ncell_(0), nrecur_(0), nstream_(0), chunk_size_(0), clip_gradient_(0.0), do_stream_reset_(false) {
  const int per_dir = cfg.bidir ? dout / 2 : dout;
  if (cfg.proj) nrecur_ = per_dir;       // e.g. lc.h:61-62: ncell_(0), nrecur_(output_dim/2)
  else ncell_ = per_dir;                 // nnet-recurrent-component.h:32,110
}

void LstmFamily::InitData(std::istream &is) {
  float param_scale = 0.02;
  std::string token;
  while (!is.eof()) {
    ReadToken(is, false, &token);
    if (cfg_.cell_dim_token && token == "<CellDim>") ReadBasicType(is, false, &ncell_);
    else if (token == "<ClipGradient>") ReadBasicType(is, false, &clip_gradient_);
    else if (token == "<ParamScale>") ReadBasicType(is, false, &param_scale);
    else ASLP_ERR << "Unknown token " << token << ", a typo in config?"
                  << (cfg_.cell_dim_token ? " (CellDim|ClipGradient|ParamScale)" : " (ClipGradient|ParamScale)");
    is >> std::ws;
  }
  ASLP_ASSERT(ncell_ > 0);
  f_.Configure(input_dim_, ncell_, cfg_.proj ? nrecur_ : 0, cfg_.cifg);
  f_.InitRandom(param_scale);
  if (cfg_.bidir) {
    b_.Configure(input_dim_, ncell_, cfg_.proj ? nrecur_ : 0, cfg_.cifg);
    b_.InitRandom(param_scale);
  }
  ASLP_ASSERT(clip_gradient_ >= 0.0);
}

void LstmFamily::ReadData(std::istream &is, bool binary) {
  if (cfg_.cell_dim_token) { ExpectToken(is, binary, "<CellDim>"); ReadBasicType(is, binary, &ncell_); }
  ExpectToken(is, binary, "<ClipGradient>");
  ReadBasicType(is, binary, &clip_gradient_);
  f_.Configure(input_dim_, ncell_, cfg_.proj ? nrecur_ : 0, cfg_.cifg);
  f_.Read(is, binary);
  if (cfg_.bidir) {
    b_.Configure(input_dim_, ncell_, cfg_.proj ? nrecur_ : 0, cfg_.cifg);
    b_.Read(is, binary);
  }
}

void LstmFamily::WriteData(std::ostream &os, bool binary) const {
  if (cfg_.cell_dim_token) { WriteToken(os, binary, "<CellDim>"); WriteBasicType(os, binary, ncell_); }
  WriteToken(os, binary, "<ClipGradient>");
  WriteBasicType(os, binary, clip_gradient_);
  f_.Write(os, binary);
  if (cfg_.bidir) b_.Write(os, binary);
}

int32 LstmFamily::NumParams() const { return f_.NumParams() * (cfg_.bidir ? 2 : 1); }

void LstmFamily::GetParams(std::vector<BaseFloat> *w) const {
  w->clear();
  f_.AppendParams(w);
  if (cfg_.bidir) b_.AppendParams(w);
}

void LstmFamily::GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) {
  params->clear();
  f_.AppendGpuParams(params);
  if (cfg_.bidir) b_.AppendGpuParams(params);
}

std::string LstmFamily::Info() const {
  std::string s = f_.Info(cfg_.bidir ? "f_" : "");
  if (cfg_.bidir) s += b_.Info("b_");
  return s;
}

void LstmFamily::ResetLstmStreams(const std::vector<int32> &stream_reset_flag) {
  if (!cfg_.stream_reset) return;
  if (nstream_ == 0) {  // first call tells the number of streams (lc.h:477-483)
    nstream_ = stream_reset_flag.size();
    prev_state_.Resize(nstream_, f_.Width(), kSetZero);
    ASLP_LOG << "Running training with " << nstream_ << " streams.";
  }
  ASLP_ASSERT(prev_state_.NumRows() == (int)stream_reset_flag.size());
  for (size_t s = 0; s < stream_reset_flag.size(); s++)
    if (stream_reset_flag[s] == 1) prev_state_.RowRange(s, 1).SetZero();
}

void LstmFamily::SetSeqLengths(const std::vector<int32> &sequence_lengths) {
  if (cfg_.bidir && !cfg_.lc) {  // nnet-blstm-projected-streams.h:81-83
    sequence_lengths_ = sequence_lengths;
    seq_len_dev_.CopyFromVec(sequence_lengths);
  } else {  // whole-sentence training of the carried-state variants (nnet-lstm-projected-streams.h:308-311)
    nstream_ = sequence_lengths.size();
    prev_state_.Resize(nstream_, f_.Width(), kSetZero);
  }
}

void LstmFamily::PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {
  const bool carried = !cfg_.bidir || cfg_.lc;
  bool out_written = false;
  int32 S;
  if (carried) {
    if (nstream_ == 0) {  // nnet-forward: one stream, state reset per utterance (lc.h:505-512)
      do_stream_reset_ = true;
      nstream_ = 1;
      prev_state_.Resize(nstream_, f_.Width(), kSetZero);
      ASLP_LOG << "Running nnet-forward with per-utterance LSTM-state reset";
    }
    if (do_stream_reset_) prev_state_.SetZero();
    S = nstream_;
  } else {
    S = sequence_lengths_.size();  // nnet-blstm-projected-streams.h:469
  }
  ASLP_ASSERT(S > 0);
  ASLP_ASSERT(in.NumRows() % S == 0);
  const int32 T = in.NumRows() / S;
  const int rec = f_.Rec();
  if (f_.FusedOk()) {
    // fused recurrence: batched x-part, the recurrence itself as ONE persistent launch for all timesteps and directions
    // (csrc/rnn_persistent.hip) -- or, where that kernel does not apply, one launch per timestep --, batched projection
    aslp_lstm_seq q = aslp_lstm_seq();
    q.ndir = cfg_.bidir ? 2 : 1; q.ld = 4; q.ldw = 4; q.T = T; q.S = S; q.C = ncell_; q.cifg = cfg_.cifg ? 1 : 0;
    // one launch serves 8 chains of 8 streams (32 streams of a bidirectional layer, 64 of a unidirectional one); more streams go in windows
    const int per_launch = 64 / q.ndir, nwin = (S + per_launch - 1) / per_launch;
    if (nwin > 1) q.s_count = per_launch;
    const bool persistent = aslp_lstm_seq_supported(&q, 0) != 0;
    last_persistent_ = persistent;
    if (cfg_.bidir) LstmDir::ForwardPreparePair(f_, b_, in, T, S, carried ? &prev_state_ : nullptr, &f_buf_, &b_buf_, persistent, planes_.get());
    else f_.ForwardPrepare(in, T, S, false, carried ? &prev_state_ : nullptr, &f_buf_, persistent);
    // The carried history holds r(0) = m(0) W_rm^T formed with the weights of the PREVIOUS batch (the reference
    // recurs on the stored r, lc.h:575); m(0) W_eff^T would silently re-project it with the updated W_rm.  The persistent
    // kernel forms r(0) W_r^T itself at its first step where it can; otherwise a 32-row product adds it here.
    const bool first_in_kernel = carried && cfg_.proj && persistent && aslp_lstm_seq_first_product_supported_for(f_.R, ncell_) != 0 &&
                                 f_.w_r.Stride() % 4 == 0 && f_.OffRec() % 4 == 0;
    if (carried && cfg_.proj && !first_in_kernel) {
      CuSubMatrix y_gates(f_buf_, S, S, 0, f_.GC()), r0(f_buf_, 0, S, f_.OffRec(), f_.R);
      y_gates.AddMatMat(1.0, r0, kNoTrans, f_.w_r, kTrans, 1.0);
    }
    if (persistent) {
      q.ld = f_buf_.Stride(); q.ldw = f_.Weff().Stride();
      for (int d = 0; d < q.ndir; d++) {
        const LstmDir &p = d == 0 ? f_ : b_;
        q.dir[d].y = (d == 0 ? f_buf_ : b_buf_).Data();
        q.dir[d].w = p.Weff().Data();
        q.dir[d].peep_i = cfg_.cifg ? nullptr : p.peep_i.Data();
        q.dir[d].peep_f = p.peep_f.Data();
        q.dir[d].peep_o = p.peep_o.Data();
        q.dir[d].seq_lengths = (d == 1 && !cfg_.lc) ? seq_len_dev_.Data() : nullptr;
        q.dir[d].reverse = d;
        q.dir[d].skip_first_product = (d == 0 && carried && cfg_.proj && !first_in_kernel) ? 1 : 0;
        if (d == 0 && first_in_kernel) {
          q.dir[d].w_first = f_.w_r.Data(); q.dir[d].ldw_first = f_.w_r.Stride();
          q.dir[d].k_first = f_.R; q.dir[d].col_first = f_.OffRec();
        }
      }
      ASLP_ASSERT(!cfg_.bidir || b_buf_.Stride() == f_buf_.Stride());
      RegionScope timed("lstm_recurrence_fwd");
      for (int w = 0; w < nwin; w++) {
        if (nwin > 1) { q.s_begin = w * per_launch; q.s_count = std::min(per_launch, S - q.s_begin); }
        aslp_lstm_seq_forward(&q);
      }
    } else {
      aslp_lstm_step a = aslp_lstm_step();
      a.ndir = cfg_.bidir ? 2 : 1; a.ld = f_buf_.Stride(); a.S = S; a.C = ncell_; a.cifg = cfg_.cifg ? 1 : 0;
      a.ldw = f_.Weff().Stride();
      for (int d = 0; d < a.ndir; d++) {
        const LstmDir &p = d == 0 ? f_ : b_;
        a.dir[d].w = p.Weff().Data();
        a.dir[d].peep_i = cfg_.cifg ? nullptr : p.peep_i.Data();
        a.dir[d].peep_f = p.peep_f.Data();
        a.dir[d].peep_o = p.peep_o.Data();
        a.dir[d].seq_lengths = (d == 1 && !cfg_.lc) ? seq_len_dev_.Data() : nullptr;
      }
      RegionScope timed("lstm_recurrence_fwd");
      for (int step = 0; step < T; step++) {
        const int tf = 1 + step, tb = T - step;
        a.dir[0].no_product = (step == 0 && carried && cfg_.proj) ? 1 : 0;
        a.dir[0].y_cur = f_buf_.RowData(tf * S); a.dir[0].y_prev = f_buf_.RowData((tf - 1) * S); a.dir[0].t = tf;
        if (cfg_.bidir) { a.dir[1].y_cur = b_buf_.RowData(tb * S); a.dir[1].y_prev = b_buf_.RowData((tb + 1) * S); a.dir[1].t = tb; }
        aslp_lstm_step_forward(&a);
      }
    }
    CheckK();
    // with a projection the GEMM that forms r(t) for all t also writes it into this component's output block
    if (cfg_.bidir) out_written = LstmDir::ForwardFinishPair(f_, b_, T, S, &f_buf_, &b_buf_, out, planes_.get());
    else out_written = f_.ForwardFinish(T, S, &f_buf_, out, 0);
  } else {
    f_.Forward(in, T, S, false, carried ? &prev_state_ : nullptr, nullptr, &f_buf_);
    if (cfg_.bidir) b_.Forward(in, T, S, true, nullptr, cfg_.lc ? nullptr : &seq_len_dev_, &b_buf_);
  }
  if (carried) {
    // next batch starts from the last frame (nnet-lstm-projected-streams.h:432); the latency-controlled
    // BLSTM from the last frame of the chunk proper, not of its right context (lc.h:629)
    const int row_block = cfg_.lc ? chunk_size_ : T;
    ASLP_ASSERT(row_block <= T + 1);
    prev_state_.CopyFromMat(f_buf_.RowRange(row_block * S, S));
  }
  if (out_written) {
  } else if (cfg_.bidir) {
    CuSubMatrix(*out, 0, T * S, 0, rec).CopyFromMat(CuSubMatrix(f_buf_, S, T * S, f_.OffRec(), rec));
    CuSubMatrix(*out, 0, T * S, rec, rec).CopyFromMat(CuSubMatrix(b_buf_, S, T * S, b_.OffRec(), rec));
  } else {
    out->CopyFromMat(CuSubMatrix(f_buf_, S, T * S, f_.OffRec(), rec));
  }
}

void LstmFamily::BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {
  if (TakeInDiffUnused()) in_diff = nullptr;   // first layer of a training step: only the gradients are wanted
  const bool carried = !cfg_.bidir || cfg_.lc;
  const int32 S = carried ? nstream_ : (int32)sequence_lengths_.size();
  ASLP_ASSERT(S > 0 && in.NumRows() % S == 0);
  const int32 T = in.NumRows() / S;
  const int rec = f_.Rec();
  const BaseFloat mmt = opts_.momentum;
  const bool folded = TakeFoldHint();
  const bool aside = TakeGradsAside() && folded;   // the gradients (and d_r with them) go beside the recurrence of the layer below
  bool dr_deferred = false;
  if (f_.FusedOk()) {
    CuSubMatrix od_f(out_diff, 0, T * S, 0, rec), od_b(out_diff, 0, T * S, cfg_.bidir ? rec : 0, rec);
    aslp_lstm_seq q = aslp_lstm_seq();
    q.ndir = cfg_.bidir ? 2 : 1; q.ld = 4; q.ldw = 4; q.T = T; q.S = S; q.C = ncell_; q.cifg = cfg_.cifg ? 1 : 0;
    const int per_launch = 64 / q.ndir, nwin = (S + per_launch - 1) / per_launch;   // stream windows as in PropagateFnc
    if (nwin > 1) q.s_count = per_launch;
    const bool persistent = aslp_lstm_seq_supported(&q, 1) != 0;
    if (cfg_.bidir) LstmDir::BackwardPreparePair(f_, b_, od_f, od_b, T, S, &f_dbuf_, &b_dbuf_, persistent, planes_.get());
    else f_.BackwardPrepare(od_f, T, S, &f_dbuf_, persistent);
    ASLP_ASSERT(f_dbuf_.Stride() == f_buf_.Stride());
    if (persistent) {
      q.ld = f_dbuf_.Stride(); q.ldw = f_.Weff().Stride();
      static const bool vec_fused_off = getenv("ASLP_LSTM_VEC_FUSED") != nullptr && getenv("ASLP_LSTM_VEC_FUSED")[0] == '0';   // A/B switch
      LstmPlanes *lp = cfg_.bidir ? planes_.get() : nullptr;
      if (lp) { lp->dg_parts[0] = lp->dg_parts[1] = nullptr; lp->dg_nparts = 0; }
      static const bool known_off = getenv("ASLP_LSTM_KNOWN_BOUNDS") != nullptr && getenv("ASLP_LSTM_KNOWN_BOUNDS")[0] == '0';   // A/B switch
      if (lp && !known_off && nwin == 1 && PlanesLevel() >= 3) {   // ... and the per-workgroup maxima of the gate diffs, for their planes
        if (dmax_parts_.Dim() != 512) dmax_parts_.Resize(512);
        q.dmax_parts[0] = dmax_parts_.Data(); q.dmax_parts[1] = dmax_parts_.Data() + 256;
      }
      if (!vec_fused_off && nwin == 1) {   // the kernel also leaves the sums the bias / peephole gradients are made of (8 chains x 7 quantities x C)
        if (grad_partial_.NumRows() != 16 * 7 || grad_partial_.NumCols() != ncell_) grad_partial_.Resize(16 * 7, ncell_, kUndefined);   // <= 16 chains
        q.grad_partial = grad_partial_.Data();
        q.grad_ld = grad_partial_.Stride();
      }
      for (int d = 0; d < q.ndir; d++) {
        const LstmDir &p = d == 0 ? f_ : b_;
        q.dir[d].y = (d == 0 ? f_buf_ : b_buf_).Data();
        q.dir[d].d = (d == 0 ? f_dbuf_ : b_dbuf_).Data();
        q.dir[d].w = p.Weff().Data();   // the persistent backward kernel multiplies with ROWS of W_eff (csrc/rnn_persistent.hip)
        q.dir[d].peep_i = cfg_.cifg ? nullptr : p.peep_i.Data();
        q.dir[d].peep_f = p.peep_f.Data();
        q.dir[d].peep_o = p.peep_o.Data();
        q.dir[d].reverse = d;
      }
      RegionScope timed("lstm_recurrence_bwd");
      for (int w = 0; w < nwin; w++) {
        if (nwin > 1) { q.s_begin = w * per_launch; q.s_count = std::min(per_launch, S - q.s_begin); }
        aslp_lstm_seq_backward(&q);
      }
      if (lp && q.dmax_parts[0] && aslp_lstm_seq_last_dmax() > 0) {
        lp->dg_parts[0] = q.dmax_parts[0]; lp->dg_parts[1] = q.dmax_parts[1];
        lp->dg_nparts = aslp_lstm_seq_last_dmax();
      }
      if (q.grad_partial) { vec_seq_ = q; vec_seq_valid_ = true; }
    } else {
      aslp_lstm_step a = aslp_lstm_step();
      a.ndir = cfg_.bidir ? 2 : 1; a.ld = f_dbuf_.Stride(); a.S = S; a.C = ncell_; a.cifg = cfg_.cifg ? 1 : 0;
      f_.RefreshEffT();
      if (cfg_.bidir) b_.RefreshEffT();
      a.ldw = f_.w_eff_t.Stride();
      for (int d = 0; d < a.ndir; d++) {
        const LstmDir &p = d == 0 ? f_ : b_;
        a.dir[d].w = p.w_eff_t.Data();
        a.dir[d].peep_i = cfg_.cifg ? nullptr : p.peep_i.Data();
        a.dir[d].peep_f = p.peep_f.Data();
        a.dir[d].peep_o = p.peep_o.Data();
      }
      RegionScope timed("lstm_recurrence_bwd");
      for (int step = 0; step < T; step++) {
        const int tf = T - step, tb = 1 + step;  // BPTT runs against each direction's recursion
        a.dir[0].d_cur = f_dbuf_.RowData(tf * S); a.dir[0].d_next = f_dbuf_.RowData((tf + 1) * S);
        a.dir[0].y_cur = f_buf_.RowData(tf * S); a.dir[0].y_next = f_buf_.RowData((tf + 1) * S); a.dir[0].y_prev = f_buf_.RowData((tf - 1) * S);
        a.dir[0].has_next = step > 0;
        if (cfg_.bidir) {
          a.dir[1].d_cur = b_dbuf_.RowData(tb * S); a.dir[1].d_next = b_dbuf_.RowData((tb - 1) * S);
          a.dir[1].y_cur = b_buf_.RowData(tb * S); a.dir[1].y_next = b_buf_.RowData((tb - 1) * S); a.dir[1].y_prev = b_buf_.RowData((tb + 1) * S);
          a.dir[1].has_next = step > 0;
        }
        aslp_lstm_step_backward(&a);
      }
    }
    CheckK();
    // (Running the backward direction's batched products beside the forward direction's on the side stream was tried: no gain,
    // 4.289 vs 4.285 ms per LC step -- the small products do not overlap usefully -- so everything stays on one stream.)
    // d_r feeds the W_rm gradient only: where the gradients go beside the recurrence of the layer below, it goes with them instead of
    // standing in front of this layer's in-diff (37 us per layer of the LC-BLSTM step's main stream).  A/B: ASLP_LSTM_DR_ASIDE=0.
    static const bool dr_aside_off = getenv("ASLP_LSTM_DR_ASIDE") != nullptr && getenv("ASLP_LSTM_DR_ASIDE")[0] == '0';
    if (cfg_.bidir) dr_deferred = LstmDir::BackwardFinishPair(f_, b_, od_f, od_b, T, S, &f_dbuf_, &b_dbuf_, in_diff, planes_.get(), !(aside && !dr_aside_off));
    else {
      dr_deferred = aside && !dr_aside_off && f_.R > 0;
      f_.BackwardFinish(od_f, T, S, false, &f_dbuf_, in_diff, 0.0, !dr_deferred);
    }
  } else {
    f_.Backward(CuSubMatrix(out_diff, 0, T * S, 0, rec), T, S, false, f_buf_, &f_dbuf_, in_diff, 0.0);
    if (cfg_.bidir) b_.Backward(CuSubMatrix(out_diff, 0, T * S, rec, rec), T, S, true, b_buf_, &b_dbuf_, in_diff, 1.0);
  }
  const BaseFloat lr_fold = folded ? opts_.learn_rate : 0.0f;
  const aslp_lstm_seq *seq = vec_seq_valid_ ? &vec_seq_ : nullptr;
  vec_seq_valid_ = false;
  auto grads = [&]() {
    if (dr_deferred && cfg_.bidir) {
      CuSubMatrix od_f2(out_diff, 0, T * S, 0, rec), od_b2(out_diff, 0, T * S, rec, rec);
      LstmDir::BackwardDrPair(f_, b_, od_f2, od_b2, T, S, &f_dbuf_, &b_dbuf_, planes_.get());
    } else if (dr_deferred) {
      f_.BackwardFinish(CuSubMatrix(out_diff, 0, T * S, 0, rec), T, S, false, &f_dbuf_, nullptr, 0.0, true);   // (d_r alone)
    }
    if (cfg_.bidir && f_.FusedOk()) LstmDir::GradsPair(f_, b_, in, T, S, f_buf_, b_buf_, f_dbuf_, b_dbuf_, mmt, clip_gradient_, lr_fold, seq, planes_.get());
    else {
      f_.Grads(in, T, S, false, f_buf_, f_dbuf_, mmt, clip_gradient_, lr_fold, seq, 0);
      if (cfg_.bidir) b_.Grads(in, T, S, true, b_buf_, b_dbuf_, mmt, clip_gradient_, lr_fold, seq, 1);
    }
  };
  // The weight gradients (and their folded SGD steps) read this layer's activations and diffs only: with layers below, the executor lets them
  // run on the side stream, where their workgroups share the CUs with the NEXT layer's persistent recurrence -- a latency chain that leaves
  // the CU's issue slots, LDS and registers mostly idle.  Everything they write (corr, W) is next read after the executor's join.
  // (Only with the SGD step folded into them: a separate Update() would follow on the main stream and read corr too early.)
  if (aside) {
    SideStreamScope aside_scope;
    grads();
  } else {
    grads();
  }
}

void LstmFamily::Update(const CuMatrixBase &, const CuMatrixBase &) {
  if (SkipFoldedUpdate()) return;
  const BaseFloat lr = opts_.learn_rate;
  f_.Update(lr);
  if (cfg_.bidir) b_.Update(lr);
}

// ---- GruStreams ---------------------------------------------------------------------------------------
void GruStreams::AllocCorr() {
  const int H = output_dim_;
  w_zrm_x_corr_.Resize(3 * H, input_dim_, kSetZero);
  w_zr_h_corr_.Resize(2 * H, H, kSetZero);
  w_m_g_corr_.Resize(H, H, kSetZero);
  bias_corr_.Resize(3 * H, kSetZero);
}

void GruStreams::InitData(std::istream &is) {  // nnet-gru-streams.h:68-109
  float param_scale = 0.02;
  std::string token;
  while (!is.eof()) {
    ReadToken(is, false, &token);
    if (token == "<ClipGradient>") ReadBasicType(is, false, &clip_gradient_);
    else if (token == "<ParamScale>") ReadBasicType(is, false, &param_scale);
    else ASLP_ERR << "Unknown token " << token << ", a typo in config?"
                  << " (ClipGradient|ParamScale)";
    is >> std::ws;
  }
  const int H = output_dim_;
  w_zrm_x_.Resize(3 * H, input_dim_, kUndefined);
  w_zr_h_.Resize(2 * H, H, kUndefined);
  w_m_g_.Resize(H, H, kUndefined);
  InitMatParamUniform(w_zrm_x_, param_scale);
  InitMatParamUniform(w_zr_h_, param_scale);
  InitMatParamUniform(w_m_g_, param_scale);
  bias_.Resize(3 * H, kUndefined);
  InitVecParamUniform(bias_, param_scale);
  AllocCorr();
  ASLP_ASSERT(clip_gradient_ >= 0.0);
}

void GruStreams::ReadData(std::istream &is, bool binary) {  // :111-125
  ExpectToken(is, binary, "<ClipGradient>");
  ReadBasicType(is, binary, &clip_gradient_);
  w_zrm_x_.Read(is, binary);
  w_zr_h_.Read(is, binary);
  w_m_g_.Read(is, binary);
  bias_.Read(is, binary);
  const int H = output_dim_;
  ASLP_ASSERT(w_zrm_x_.NumRows() == 3 * H && w_zrm_x_.NumCols() == input_dim_);
  ASLP_ASSERT(w_zr_h_.NumRows() == 2 * H && w_zr_h_.NumCols() == H);
  ASLP_ASSERT(w_m_g_.NumRows() == H && w_m_g_.NumCols() == H);
  ASLP_ASSERT(bias_.Dim() == 3 * H);
  AllocCorr();
}

void GruStreams::WriteData(std::ostream &os, bool binary) const {  // :127-136
  WriteToken(os, binary, "<ClipGradient>");
  WriteBasicType(os, binary, clip_gradient_);
  w_zrm_x_.Write(os, binary);
  w_zr_h_.Write(os, binary);
  w_m_g_.Write(os, binary);
  bias_.Write(os, binary);
}

int32 GruStreams::NumParams() const {
  const int H = output_dim_;
  return 3 * H * input_dim_ + 2 * H * H + H * H + 3 * H;
}

void GruStreams::GetParams(std::vector<BaseFloat> *w) const {
  w->clear();
  AppendRowMajor(w_zrm_x_, w);
  AppendRowMajor(w_zr_h_, w);
  AppendRowMajor(w_m_g_, w);
  AppendVector(bias_, w);
}

void GruStreams::GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) {
  params->clear();
  params->push_back(MatParam(w_zrm_x_));
  params->push_back(MatParam(w_zr_h_));
  params->push_back(MatParam(w_m_g_));
  params->push_back(VecParam(bias_));
}

void GruStreams::SetSeqLengths(const std::vector<int32> &sequence_lengths) {  // :216-219
  nstream_ = sequence_lengths.size();
  prev_state_.Resize(nstream_, 5 * output_dim_, kSetZero);
}

void GruStreams::ResetLstmStreams(const std::vector<int32> &stream_reset_flag) {  // :221-236
  if (nstream_ == 0) {
    nstream_ = stream_reset_flag.size();
    prev_state_.Resize(nstream_, 5 * output_dim_, kSetZero);
    ASLP_LOG << "Running training with " << nstream_ << " streams.";
  }
  ASLP_ASSERT(prev_state_.NumRows() == (int)stream_reset_flag.size());
  for (size_t s = 0; s < stream_reset_flag.size(); s++)
    if (stream_reset_flag[s] == 1) prev_state_.RowRange(s, 1).SetZero();
}

void GruStreams::PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {  // :238-321
  const int H = output_dim_;
  if (nstream_ == 0) {
    do_stream_reset_ = true;
    nstream_ = 1;
    prev_state_.Resize(nstream_, 5 * H, kSetZero);
    ASLP_LOG << "Runing nnet-forward with per-utterance GRU-state reset";
  }
  if (do_stream_reset_) prev_state_.SetZero();
  ASLP_ASSERT(in.NumRows() % nstream_ == 0);
  const int32 T = in.NumRows() / nstream_, S = nstream_;
  static const bool unfused = getenv("ASLP_LSTM_UNFUSED") != nullptr && getenv("ASLP_LSTM_UNFUSED")[0] == '1';
  // the whole recurrence as one persistent launch (csrc/rnn_persistent.hip) where it applies, else four launches per timestep
  aslp_gru_seq q = aslp_gru_seq();
  q.T = T; q.S = S; q.H = H; q.ld = (5 * H + 15) & ~15;
  const int per_launch = 64, nwin = (S + per_launch - 1) / per_launch;   // 8 chains of 8 streams per launch; more streams go in windows
  if (nwin > 1) q.s_count = per_launch;
  const bool persistent = !unfused && aslp_gru_seq_supported(&q, 0) != 0;
  if (persistent) {  // g and h of row blocks 1..T start as "not yet published", the two boundary blocks as zero
    buf_.Resize((T + 2) * S, 5 * H, kUndefined);
    aslp_lstm_seq_fill(buf_.Data(), buf_.Stride(), T, S, 3 * H, 2 * H);
  } else {
    buf_.Resize((T + 2) * S, 5 * H, kSetZero);
  }
  buf_.RowRange(0, S).CopyFromMat(prev_state_);
  {
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    ep.bias = bias_.Data();
    CuSubMatrix zrm(buf_, S, T * S, 0, 3 * H);
    zrm.AddMatMat(1.0, in, kNoTrans, w_zrm_x_, kTrans, 0.0, &ep);
  }
  const int ld = buf_.Stride();
  if (persistent) {
    q.y = buf_.Data(); q.ld = ld;
    q.w_zr = w_zr_h_.Data(); q.ldw_zr = w_zr_h_.Stride();
    q.w_m = w_m_g_.Data(); q.ldw_m = w_m_g_.Stride();
    RegionScope timed("gru_recurrence_fwd");
    for (int w = 0; w < nwin; w++) {
      if (nwin > 1) { q.s_begin = w * per_launch; q.s_count = std::min(per_launch, S - q.s_begin); }
      aslp_gru_seq_forward(&q);
    }
  }
  const bool fused = !persistent && !unfused && aslp_gru_step_supported(H);
  for (int t = 1; t <= T && fused; t++)
    aslp_gru_step_forward(buf_.RowData(t * S), buf_.RowData((t - 1) * S), w_zr_h_.Data(), w_zr_h_.Stride(), w_m_g_.Data(), w_m_g_.Stride(), ld, S, H);
  for (int t = 1; t <= T && !fused && !persistent; t++) {
    CuSubMatrix y_zr(buf_, t * S, S, 0, 2 * H), h_prev(buf_, (t - 1) * S, S, 4 * H, H);
    y_zr.AddMatMat(1.0, h_prev, kNoTrans, w_zr_h_, kTrans, 1.0);
    aslp_gru_forward1(buf_.RowData(t * S), buf_.RowData((t - 1) * S), ld, S, H);
    CuSubMatrix y_m(buf_, t * S, S, 2 * H, H), y_g(buf_, t * S, S, 3 * H, H);
    y_m.AddMatMat(1.0, y_g, kNoTrans, w_m_g_, kTrans, 1.0);
    aslp_gru_forward2(buf_.RowData(t * S), buf_.RowData((t - 1) * S), ld, S, H);
  }
  CheckK();
  prev_state_.CopyFromMat(buf_.RowRange(T * S, S));
  out->CopyFromMat(CuSubMatrix(buf_, S, T * S, 4 * H, H));
}

void GruStreams::BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {  // :323-430
  const int H = output_dim_;
  if (TakeInDiffUnused()) in_diff = nullptr;
  ASLP_ASSERT(nstream_ > 0 && in.NumRows() % nstream_ == 0);
  const int32 T = in.NumRows() / nstream_, S = nstream_;
  static const bool unfused = getenv("ASLP_LSTM_UNFUSED") != nullptr && getenv("ASLP_LSTM_UNFUSED")[0] == '1';
  aslp_gru_seq q = aslp_gru_seq();
  q.T = T; q.S = S; q.H = H; q.ld = buf_.Stride();
  const int per_launch = 64, nwin = (S + per_launch - 1) / per_launch;
  if (nwin > 1) q.s_count = per_launch;
  const bool persistent = !unfused && aslp_gru_seq_supported(&q, 1) != 0;
  if (persistent) {  // d_z, d_r, d_m of row blocks 1..T start as "not yet published" (the kernel writes d_g, d_h's loss share follows), boundary blocks zero
    dbuf_.Resize((T + 2) * S, 5 * H, kUndefined);
    aslp_lstm_seq_fill(dbuf_.Data(), dbuf_.Stride(), T, S, 0, 3 * H);
  } else {
    dbuf_.Resize((T + 2) * S, 5 * H, kSetZero);
  }
  CuSubMatrix(dbuf_, S, T * S, 4 * H, H).CopyFromMat(out_diff);
  const int ld = dbuf_.Stride();
  ASLP_ASSERT(ld == buf_.Stride());
  const bool fused = !persistent && !unfused && aslp_gru_step_supported(H);
  if (fused || persistent) {  // the backward products read W (not W^T): keep K-contiguous transposed copies, refreshed here once per call
    if (w_zr_h_t_.NumRows() != H) { w_zr_h_t_.Resize(H, 2 * H, kUndefined); w_m_g_t_.Resize(H, H, kUndefined); }
    w_zr_h_t_.CopyFromMatTrans(w_zr_h_);
    w_m_g_t_.CopyFromMatTrans(w_m_g_);
  }
  if (persistent) {
    q.y = buf_.Data(); q.d = dbuf_.Data();
    q.w_zr = w_zr_h_t_.Data(); q.ldw_zr = w_zr_h_t_.Stride();
    q.w_m = w_m_g_t_.Data(); q.ldw_m = w_m_g_t_.Stride();
    RegionScope timed("gru_recurrence_bwd");
    for (int w = 0; w < nwin; w++) {
      if (nwin > 1) { q.s_begin = w * per_launch; q.s_count = std::min(per_launch, S - q.s_begin); }
      aslp_gru_seq_backward(&q);
    }
  }
  if (fused) {
    for (int t = T; t >= 1; t--)
      aslp_gru_step_backward(dbuf_.RowData(t * S), dbuf_.RowData((t + 1) * S), buf_.RowData(t * S), buf_.RowData((t + 1) * S),
                             buf_.RowData((t - 1) * S), w_zr_h_t_.Data(), w_zr_h_t_.Stride(), w_m_g_t_.Data(), w_m_g_t_.Stride(), ld, S, H, t < T);
  }
  for (int t = T; t >= 1 && !fused && !persistent; t--) {
    if (t < T) {
      CuSubMatrix d_h(dbuf_, t * S, S, 4 * H, H), dn_zr(dbuf_, (t + 1) * S, S, 0, 2 * H);
      d_h.AddMatMat(1.0, dn_zr, kNoTrans, w_zr_h_, kNoTrans, 1.0);
    }
    aslp_gru_backward1(dbuf_.RowData(t * S), dbuf_.RowData((t + 1) * S), buf_.RowData(t * S), buf_.RowData((t + 1) * S), ld, S, H);
    CuSubMatrix d_g(dbuf_, t * S, S, 3 * H, H), d_m(dbuf_, t * S, S, 2 * H, H);
    d_g.AddMatMat(1.0, d_m, kNoTrans, w_m_g_, kNoTrans, 0.0);
    aslp_gru_backward2(dbuf_.RowData(t * S), buf_.RowData(t * S), buf_.RowData((t - 1) * S), ld, S, H);
  }
  CheckK();
  CuSubMatrix d_zrm(dbuf_, S, T * S, 0, 3 * H);
  if (in_diff) in_diff->AddMatMat(1.0, d_zrm, kNoTrans, w_zrm_x_, kNoTrans, 0.0);
  // gradients with momentum, clipped element-wise (:432-455); with the executor's fold hint the step of :457-466 rides along
  const BaseFloat mmt = opts_.momentum;
  const BaseFloat lr_fold = TakeFoldHint() ? opts_.learn_rate : 0.0f;
  auto wgrad = [&](CuMatrix &corr, CuMatrix &w, const CuMatrixBase &d, const CuMatrixBase &x) {
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    ep.clip = clip_gradient_;
    if (lr_fold != 0.0f) { ep.W = w.Data(); ep.ldw = w.Stride(); ep.w_alpha = -lr_fold; }
    corr.AddMatMat(1.0, d, kTrans, x, kNoTrans, mmt, &ep);
  };
  wgrad(w_zrm_x_corr_, w_zrm_x_, d_zrm, in);
  wgrad(w_zr_h_corr_, w_zr_h_, CuSubMatrix(dbuf_, S, T * S, 0, 2 * H), CuSubMatrix(buf_, 0, T * S, 4 * H, H));
  wgrad(w_m_g_corr_, w_m_g_, CuSubMatrix(dbuf_, S, T * S, 2 * H, H), CuSubMatrix(buf_, S, T * S, 3 * H, H));
  aslp_rnn_vec_grad job = {dbuf_.RowData(S), nullptr, 0, 3 * H, bias_corr_.Data(), bias_.Data()};
  aslp_rnn_vec_grads(&job, 1, dbuf_.Stride(), T * S, mmt, clip_gradient_, -lr_fold);
  CheckK();
}

void GruStreams::Update(const CuMatrixBase &, const CuMatrixBase &) {  // :457-466
  if (SkipFoldedUpdate()) return;
  const BaseFloat lr = opts_.learn_rate;
  w_zrm_x_.AddMat(-lr, w_zrm_x_corr_);
  w_zr_h_.AddMat(-lr, w_zr_h_corr_);
  w_m_g_.AddMat(-lr, w_m_g_corr_);
  bias_.AddVec(-lr, bias_corr_, 1.0);
}

}  // namespace aslp
