// nnet-loss.h -- Xent / Mse / MultiTaskLoss of the host engine.
// Same interface and Report() strings as src/aslp-nnet/nnet-loss.h:35-218 / nnet-loss.cc (the
// bash schedulers grep these lines).  `Posterior` is hmm/posterior.h's type.
// Difference in mechanism only: one fused kernel per Eval and NO blocking host reads per
// minibatch -- the five statistics accumulate in device memory (double) and are fetched when
// a report or the hourly progress line needs them.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "cu-matrix.h"
#include "posterior.h"

namespace aslp {


class LossItf {
 public:
  virtual ~LossItf() {}
  virtual void Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const CuMatrixBase &target, CuMatrix *diff) = 0;
  virtual void Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const Posterior &target, CuMatrix *diff) = 0;
  virtual void Eval(const CuMatrixBase &net_out, const Posterior &target, CuMatrix *diff) {
    std::vector<BaseFloat> w(target.size(), 1.0f);
    Eval(w, net_out, target, diff);
  }
  virtual std::string Report() = 0;
  virtual BaseFloat AvgLoss() = 0;
};

void PosteriorToMatrix(const Posterior &post, int32 num_cols, CuMatrix *mat);  // nnet-utils.h:160-177

class Xent : public LossItf {
 public:
  Xent();
  void Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const CuMatrixBase &target, CuMatrix *diff);
  void Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const Posterior &target, CuMatrix *diff);
  using LossItf::Eval;
  // device-resident variants (no host data on the step path): weights / labels already on the GPU
  void EvalLabels(const CuVectorBase &frame_weights, const CuMatrixBase &net_out, const CuArray<int32> &labels, CuMatrix *diff);
  // `acts` are the activations in front of the network's final Softmax; the posteriors are formed inside the loss kernel
  // fw_max: the largest |frame weight| if the caller knows it (> 0): the diff's bound, which lets the kernel leave the diff's fp16 planes
  // for the layer product that reads it (csrc/split16.h)
  void EvalLabelsPreSoftmax(const CuVectorBase &frame_weights, const CuMatrixBase &acts, const int32 *labels_dev, CuMatrix *diff, float fw_max = -1.0f);
  void EvalLabels(const CuVectorBase &frame_weights, const CuMatrixBase &net_out, const int32 *labels_dev, CuMatrix *diff, float fw_max = -1.0f);
  // Eval(frame_weights, net_out, Posterior) for callers that hold the executor's buffers (Nnet::PropagateForLoss):
  // `loss_input` is the network output, or the activations in front of its final Softmax when `pre_softmax`
  void EvalOnLossInput(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &loss_input, bool pre_softmax, const Posterior &target,
                       CuMatrix *diff);
  std::string Report();
  BaseFloat AvgLoss() { Fetch(); return (loss_ - entropy_) / frames_; }
  // raw accumulators {frames, correct, loss, entropy, likelyhood}
  void GetStats(double out[5]) { Fetch(); out[0] = frames_; out[1] = correct_; out[2] = loss_; out[3] = entropy_; out[4] = likelyhood_; }

 private:
  void Fetch();
  void AfterEval(int rows);
  double frames_, correct_, loss_, entropy_, likelyhood_;
  double frames_progress_, loss_progress_, entropy_progress_, likelyhood_progress_;
  double rows_since_progress_;
  std::vector<float> loss_vec_;
  CuVector frame_weights_;
  CuMatrix tgt_mat_;
  CuArray<int32> labels_;
  CuVectorD stats_;  // device accumulators
  bool dirty_;
  // The label-target evaluations (the step path) leave their per-row statistics here, batch after batch, and the sum into stats_ is made
  // for all of them at once when the room is used up or the accumulators are read (Fetch): same bits, one launch less per step.
  double *PendingRowStats(int32 rows);   // room for the next batch's [rows x 5], or NULL: too large to defer
  void FlushPending();
  CuVectorD pending_;
  int32 pending_rows_ = 0, pending_batches_ = 0, pending_cap_ = 0;
};

class Mse : public LossItf {
 public:
  Mse() : frames_(0.0), loss_(0.0), frames_progress_(0.0), loss_progress_(0.0), num_tgt_(0) {}
  void Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const CuMatrixBase &target, CuMatrix *diff);
  void Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const Posterior &target, CuMatrix *diff);
  using LossItf::Eval;
  std::string Report();
  BaseFloat AvgLoss() { return loss_ / frames_; }

 private:
  double frames_, loss_, frames_progress_, loss_progress_;
  std::vector<float> loss_vec_;
  CuVector frame_weights_;
  CuMatrix tgt_mat_, diff_pow_2_;
  int num_tgt_;
};

class MultiTaskLoss : public LossItf {
 public:
  MultiTaskLoss() {}
  ~MultiTaskLoss() { for (LossItf *l : loss_vec_) delete l; }
  void InitFromString(const std::string &s);
  void Eval(const std::vector<BaseFloat> &, const CuMatrixBase &, const CuMatrixBase &, CuMatrix *) { ASLP_ERR << "This is not supposed to be called!"; }
  void Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const Posterior &target, CuMatrix *diff);
  using LossItf::Eval;
  std::string Report();
  BaseFloat AvgLoss();

 private:
  std::vector<LossItf *> loss_vec_;
  std::vector<int32> loss_dim_;
  std::vector<BaseFloat> loss_weights_;
  std::vector<int32> loss_dim_offset_;
  CuMatrix tgt_mat_;
};

}  // namespace aslp
