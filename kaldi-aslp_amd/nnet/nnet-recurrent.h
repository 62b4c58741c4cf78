// nnet-recurrent.h -- the LSTM family and GruStreams on the gfx950 gate-block kernels.
//
// One parameterised engine (LstmDir = one direction) serves the six reference components:
//   Lstm / BLstm                  nnet-recurrent-component.{h,cc}:28-215 / 26-1450   (no projection)
//   LstmProjectedStreams          nnet-lstm-projected-streams.h
//   BLstmProjectedStreams         nnet-blstm-projected-streams.h      (length masking, backward dir only)
//   BLstmProjectedStreamsLC       nnet-blstm-projected-streams-lc.h   (carried state saved at chunk_size)
//   LstmCifgProjectedStreams      nnet-lstm-couple-if-projected-streams.h  (i = 1 - f)
// Same buffers ([(T+2)S x width], row = t*S + s, columns g|i|f|o|c|h|m|r), same file formats,
// same GetParams / GetGpuParams tensor order, same quirks (cell clip +-50, per-element gradient
// clipping after momentum, masking only in the backward direction, LC state taken from row chunk_size).
// Per timestep: one skinny GEMM (recurrent), ONE fused gate-block kernel (the reference: ~18-22
// elementwise launches), one projection GEMM.
#pragma once
#include <memory>

#include <string>
#include <utility>
#include <vector>

#include "split16.h"
#include "nnet-component.h"

namespace aslp {

class RecurrentBase : public UpdatableComponent {
 public:
  RecurrentBase(int32 di, int32 dout) : UpdatableComponent(di, dout) {}
  bool GradientInBackprop() const { return true; }
  bool LatencyBoundPasses() const { return true; }
  void InDiffUnusedInNextBackprop() { in_diff_unused_ = true; }
  void FoldNextUpdateIntoBackprop() { fold_update_ = CanFoldUpdate(); }
  void GradientsBesideLowerLayers() { grads_aside_ = true; }
  virtual bool HasStreamReset() const { return false; }   // answers Nnet::ResetLstmStreams (nnet-nnet.cc:473-496)
  virtual bool HasSeqLengths() const { return false; }    // answers Nnet::SetSeqLengths   (nnet-nnet.cc:498-530)
  virtual void ResetLstmStreams(const std::vector<int32> &) {}
  virtual void SetSeqLengths(const std::vector<int32> &) {}

 protected:
  virtual bool CanFoldUpdate() const { return false; }
  // BackpropagateFnc of a folding subclass: true once per hint, and the following Update() then returns at once
  bool TakeFoldHint() { if (!fold_update_) return false; fold_update_ = false; update_done_ = true; return true; }
  bool SkipFoldedUpdate() { if (!update_done_) return false; update_done_ = false; return true; }

 private:
  bool fold_update_ = false, update_done_ = false;
 protected:
  bool TakeInDiffUnused() { const bool v = in_diff_unused_; in_diff_unused_ = false; return v; }
  bool TakeGradsAside() { const bool v = grads_aside_; grads_aside_ = false; return v; }
 private:
  bool in_diff_unused_ = false, grads_aside_ = false;
};

struct LstmDir {
  int D = 0, C = 0, R = 0;  // input dim, cells, projection dim (0: none)
  bool cifg = false;
  CuMatrix w_x, w_r, w_rm, w_x_corr, w_r_corr, w_rm_corr;
  // W_eff = W_r W_rm [GC x C] (with projection) and W_eff^T [C x GC]: the only products on the sequential path
  // of the fused recurrence (csrc/rnn_fused.hip); refreshed lazily after every parameter change
  mutable CuMatrix w_eff, w_eff_t;
  mutable bool eff_dirty = true, eff_t_dirty = true;
  bool aliased = false;  // GetGpuParams handed the tensors out (model sync may rewrite them at any time)
  CuVector bias, peep_i, peep_f, peep_o, bias_corr, peep_i_corr, peep_f_corr, peep_o_corr;

  int G() const { return cifg ? 3 : 4; }
  int GC() const { return G() * C; }
  int Rec() const { return R > 0 ? R : C; }
  int Width() const { return (G() + 3) * C + R; }
  int OffC() const { return GC(); }
  int OffM() const { return GC() + 2 * C; }
  int OffRec() const { return R > 0 ? GC() + 3 * C : OffM(); }
  int OffI() const { return C; }
  int OffF() const { return cifg ? C : 2 * C; }
  int OffO() const { return cifg ? 2 * C : 3 * C; }

  void Configure(int d, int c, int r, bool is_cifg) { D = d; C = c; R = r; cifg = is_cifg; }
  void AllocCorr();
  void InitRandom(float scale);
  void Read(std::istream &is, bool binary);
  void Write(std::ostream &os, bool binary) const;
  int NumParams() const;
  void AppendParams(std::vector<BaseFloat> *w) const;
  void AppendGpuParams(std::vector<std::pair<BaseFloat *, int>> *p);
  std::string Info(const char *prefix) const;

  // buf: activations [(T+2)S x Width]; init_state (S x Width, may be NULL) seeds the history row block
  void Forward(const CuMatrixBase &in, int T, int S, bool reverse, const CuMatrixBase *init_state, const CuArray<int32> *seq_len,
               CuMatrix *buf) const;
  // out_diff: [T*S x Rec] diff w.r.t. this direction's output; in_diff = dGATES*w_x + beta*in_diff
  void Backward(const CuMatrixBase &out_diff, int T, int S, bool reverse, const CuMatrix &buf, CuMatrix *dbuf, CuMatrixBase *in_diff,
                float beta) const;
  // lr_fold != 0: the step param += -lr_fold * corr is taken in the epilogues of the gradient kernels
  // seq (may be NULL): the persistent backward launch that already left this direction's (index dir) bias / peephole sums
  void Grads(const CuMatrixBase &in, int T, int S, bool reverse, const CuMatrix &buf, const CuMatrix &dbuf, float mmt, float clip, float lr_fold,
             const aslp_lstm_seq *seq = nullptr, int dir = 0);
  void VecGrads(int T, int S, bool reverse, const CuMatrix &buf, const CuMatrix &dbuf, float mmt, float clip, float lr_fold, const aslp_lstm_seq *seq, int dir);
  void Update(float lr);
  int VecGradJobs(int S, bool reverse, const CuMatrix &buf, const CuMatrix &dbuf, aslp_rnn_vec_grad *jobs);
  double ParamSum() const {   // on the device (Nnet::Check)
    double s = w_x.Sum() + w_r.Sum() + (double)bias.Sum() + (double)peep_f.Sum() + (double)peep_o.Sum();
    if (!cifg) s += (double)peep_i.Sum();
    if (R > 0) s += w_rm.Sum();
    return s;
  }  // bias + peephole gradient jobs (<= 4)

  // ---- fused-step path: the per-timestep loop lives in LstmFamily (all directions share a launch) ----
  bool FusedOk() const;  // C % 4 == 0 (16-byte operand loads) and not disabled by ASLP_LSTM_UNFUSED=1 (A/B switch for tests)
  void RefreshEff() const;
  void RefreshEffT() const;
  // Both directions of a bidirectional layer: every batched product below exists twice with the same shape, and most of them
  // (K or N = R, or a [R x C] output) cannot fill the chip alone -- they go out as pairs, one launch each (AddMatMatPair).
  // `with_gemm = false` on the single-direction methods leaves out the product the *Pair function then issues for both.
  static void RefreshEffPair(const LstmDir &f, const LstmDir &b, const struct S16View *views = nullptr);   // views: planes of w_r (f, b), w_rm (f, b)
  // pl (may be NULL): where the layer keeps the fp16 planes of these products' operands (LstmPlanes below)
  static void ForwardPreparePair(const LstmDir &f, const LstmDir &b, const CuMatrixBase &in, int T, int S, const CuMatrixBase *init_f,
                                 CuMatrix *fbuf, CuMatrix *bbuf, bool persistent, struct LstmPlanes *pl = nullptr);
  static bool ForwardFinishPair(const LstmDir &f, const LstmDir &b, int T, int S, CuMatrix *fbuf, CuMatrix *bbuf, CuMatrixBase *out,
                                struct LstmPlanes *pl = nullptr);
  static void BackwardPreparePair(const LstmDir &f, const LstmDir &b, const CuMatrixBase &od_f, const CuMatrixBase &od_b, int T, int S,
                                  CuMatrix *fdbuf, CuMatrix *bdbuf, bool persistent, struct LstmPlanes *pl = nullptr);
  // with_dr = false: d_r (which only the W_rm gradient reads) is left to BackwardDrPair -- the caller issues that with the gradients, beside
  // the recurrence of the layer below, instead of in front of this layer's in-diff on the main stream.  Returns true if d_r was left out.
  static bool BackwardFinishPair(const LstmDir &f, const LstmDir &b, const CuMatrixBase &od_f, const CuMatrixBase &od_b, int T, int S,
                                 CuMatrix *fdbuf, CuMatrix *bdbuf, CuMatrixBase *in_diff, struct LstmPlanes *pl = nullptr, bool with_dr = true);
  static void BackwardDrPair(const LstmDir &f, const LstmDir &b, const CuMatrixBase &od_f, const CuMatrixBase &od_b, int T, int S,
                             CuMatrix *fdbuf, CuMatrix *bdbuf, struct LstmPlanes *pl);
  static void GradsPair(LstmDir &f, LstmDir &b, const CuMatrixBase &in, int T, int S, const CuMatrix &fbuf, const CuMatrix &bbuf,
                        const CuMatrix &fdbuf, const CuMatrix &bdbuf, float mmt, float clip, float lr_fold, const aslp_lstm_seq *seq = nullptr,
                        struct LstmPlanes *pl = nullptr);
  const CuMatrixBase &Weff() const { return R > 0 ? static_cast<const CuMatrixBase &>(w_eff) : w_r; }
  void ForwardPrepare(const CuMatrixBase &in, int T, int S, bool reverse, const CuMatrixBase *init_state, CuMatrix *buf, bool persistent,
                      bool with_gemm = true) const;
  // returns true if the projected output r(1..T) was also stored at out[:, out_col ...] (only with a projection)
  bool ForwardFinish(int T, int S, CuMatrix *buf, CuMatrixBase *out, int out_col) const;                                   // batched projection
  void BackwardPrepare(const CuMatrixBase &out_diff, int T, int S, CuMatrix *dbuf, bool persistent, bool with_gemm = true) const;  // dm_ext
  // with_dr: also form d_r (needed by the W_rm gradient); in_diff == NULL: only that
  void BackwardFinish(const CuMatrixBase &out_diff, int T, int S, bool reverse, CuMatrix *dbuf, CuMatrixBase *in_diff, float beta, bool with_dr) const;
};

// The fp16 planes (csrc/split16.h) of the tensors a bidirectional layer's batched products read, one set per layer: the layer input, the
// out-diff halves, the weights, and the m / r / dGATES / d_r column blocks of the activation and diff buffers (windows of those planes
// serve the products that read a shifted row range).  Made by a few multi-matrix conversions per pass; *_ok say what this step has.
struct LstmPlanes {
  PlaneSet in, od[2], wx[2], wrm[2], wr[2], m[2], r[2], dg[2], dr[2];
  bool weights_ok = false, in_ok = false, m_ok = false, od_ok = false, dg_ok = false;
  const float *dg_parts[2] = {nullptr, nullptr};   // per-workgroup maxima of the gate diffs left by this pass' persistent backward launch
  int dg_nparts = 0;
};
struct LstmPlanesHolder {   // a copied component starts without planes
  LstmPlanesHolder() = default;
  LstmPlanesHolder(const LstmPlanesHolder &) {}
  LstmPlanesHolder &operator=(const LstmPlanesHolder &) { return *this; }
  LstmPlanes *get() { if (!p) p.reset(new LstmPlanes()); return p.get(); }
  std::unique_ptr<LstmPlanes> p;
};

// Shared implementation; the concrete classes below only fix the configuration.
class LstmFamily : public RecurrentBase {
 public:
  struct Config {
    bool bidir, proj, cifg, lc;
    bool cell_dim_token;   // file / config carries <CellDim>
    bool stream_reset;     // listed in Nnet::ResetLstmStreams
    bool seq_lengths;      // listed in Nnet::SetSeqLengths
  };
  LstmFamily(int32 di, int32 dout, const Config &cfg);
  bool CanFoldUpdate() const { return true; }

  void InitData(std::istream &is);
  void ReadData(std::istream &is, bool binary);
  void WriteData(std::ostream &os, bool binary) const;
  int32 NumParams() const;
  double ParamSum() const { return f_.ParamSum() + (cfg_.bidir ? b_.ParamSum() : 0.0); }
  void GetParams(std::vector<BaseFloat> *w) const;
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params);
  std::string Info() const;

  // a pass of launches per timestep is a chain of tiny dependent kernels (see Component::LatencyBoundPasses); one persistent launch is not
  bool LatencyBoundPasses() const { return !last_persistent_; }
  bool PersistentRecurrence() const { return last_persistent_; }
  bool HasStreamReset() const { return cfg_.stream_reset; }
  bool HasSeqLengths() const { return cfg_.seq_lengths; }
  void ResetLstmStreams(const std::vector<int32> &stream_reset_flag);
  void SetSeqLengths(const std::vector<int32> &sequence_lengths);
  void SetChunkSize(int chunk_size) { chunk_size_ = chunk_size; }

  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out);
  void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &out, const CuMatrixBase &out_diff, CuMatrixBase *in_diff);
  void Update(const CuMatrixBase &input, const CuMatrixBase &diff);

  // test access: activation buffers of the last Propagate
  const CuMatrix &ForwardBuffer(int dir) const { return dir == 0 ? f_buf_ : b_buf_; }

 protected:
  Config cfg_;
  int32 ncell_, nrecur_, nstream_, chunk_size_;
  BaseFloat clip_gradient_;
  bool do_stream_reset_;
  LstmDir f_, b_;
  CuMatrix prev_state_;                 // carried state of the forward-in-time direction
  std::vector<int32> sequence_lengths_; // BLstm* masking
  CuArray<int32> seq_len_dev_;
  CuMatrix f_buf_, b_buf_, f_dbuf_, b_dbuf_;
  LstmPlanesHolder planes_;   // made on first use
  bool last_persistent_ = false;   // the last Propagate ran the recurrence as ONE persistent launch (else: launches per timestep)
  CuMatrix grad_partial_;          // per-chain bias / peephole gradient sums of the persistent backward launch (aslp_lstm_seq.grad_partial)
  CuVector dmax_parts_;            // 2 x 256 per-workgroup maxima of the gate diffs from the same launch (aslp_lstm_seq.dmax_parts)
  aslp_lstm_seq vec_seq_ = aslp_lstm_seq();   // that launch's arguments, for aslp_lstm_seq_vec_grads
  bool vec_seq_valid_ = false;
};

#define ASLP_LSTM_CLASS(Name, Type, ...)                                                   \
  class Name : public LstmFamily {                                                         \
   public:                                                                                 \
    Name(int32 di, int32 dout) : LstmFamily(di, dout, Config{__VA_ARGS__}) {}              \
    Component *Copy() const { return new Name(*this); }                                    \
    ComponentType GetType() const { return Type; }                                         \
  }
//                                                      bidir  proj   cifg   lc     celltok reset  seqlen
ASLP_LSTM_CLASS(Lstm, kLstm,                             false, false, false, false, false, true,  true);
ASLP_LSTM_CLASS(BLstm, kBLstm,                           true,  false, false, false, false, false, true);
ASLP_LSTM_CLASS(LstmProjectedStreams, kLstmProjectedStreams, false, true, false, false, true, true, true);
ASLP_LSTM_CLASS(BLstmProjectedStreams, kBLstmProjectedStreams, true, true, false, false, true, false, true);
ASLP_LSTM_CLASS(BLstmProjectedStreamsLC, kBLstmProjectedStreamsLC, true, true, false, true, true, true, false);
ASLP_LSTM_CLASS(LstmCifgProjectedStreams, kLstmCifgProjectedStreams, false, true, true, false, true, true, true);
#undef ASLP_LSTM_CLASS

// GruStreams (nnet-gru-streams.h:39-482)
class GruStreams : public RecurrentBase {
 public:
  GruStreams(int32 di, int32 dout) : RecurrentBase(di, dout), nstream_(0), clip_gradient_(0.0), do_stream_reset_(false) {}
  Component *Copy() const { return new GruStreams(*this); }
  ComponentType GetType() const { return kGruStreams; }
  bool CanFoldUpdate() const { return true; }
  void InitData(std::istream &is);
  void ReadData(std::istream &is, bool binary);
  void WriteData(std::ostream &os, bool binary) const;
  int32 NumParams() const;
  void GetParams(std::vector<BaseFloat> *w) const;
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params);
  bool HasStreamReset() const { return true; }
  bool HasSeqLengths() const { return true; }
  void ResetLstmStreams(const std::vector<int32> &stream_reset_flag);
  void SetSeqLengths(const std::vector<int32> &sequence_lengths);
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out);
  void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &out, const CuMatrixBase &out_diff, CuMatrixBase *in_diff);
  void Update(const CuMatrixBase &input, const CuMatrixBase &diff);

 private:
  void AllocCorr();
  int32 nstream_;
  BaseFloat clip_gradient_;
  bool do_stream_reset_;
  CuMatrix prev_state_, w_zrm_x_, w_zr_h_, w_m_g_, w_zrm_x_corr_, w_zr_h_corr_, w_m_g_corr_;
  CuVector bias_, bias_corr_;
  CuMatrix buf_, dbuf_;
  CuMatrix w_zr_h_t_, w_m_g_t_;  // transposed recurrent matrices for the fused backward step
};

}  // namespace aslp
