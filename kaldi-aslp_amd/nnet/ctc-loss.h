// ctc-loss.h -- the Eesen-style Ctc loss of the host engine (src/aslp-nnet/ctc-loss.{h,cc}).
// Input is the Softmax output; Eval (one sequence) / EvalParallel (padded multi-sequence batch, row =
// t*num_seq + s), ErrorRate / ErrorRateMSeq, the abnormal-loss filter (stat_period 100 here) and the
// Report() string follow the reference.  Mechanism: the lattice runs in two launches over all sequences
// (aslp_eesen_ctc_mseq) instead of 2T row kernels + the O(T*A*(2L+1)) error kernel + 6 elementwise passes.
#pragma once
#include <string>
#include <vector>

#include "cu-matrix.h"

namespace aslp {

#define SUM_LOSS_CHECK 0
#define AVG_LOSS_CHECK 1
#define NONE_LOSS_CHECK 2
#define CTC_GRAD_CHECK 1  // ctc-loss.h:37: the reference ships with the average-loss check compiled in

class Ctc {
 public:
  Ctc();
  void Eval(const CuMatrixBase &net_out, const std::vector<int32> &label, CuMatrix *diff);
  void EvalParallel(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const CuMatrixBase &net_out,
                    std::vector<std::vector<int32>> &label, CuMatrix *diff);
  void ErrorRate(const CuMatrixBase &net_out, const std::vector<int32> &label, float *err, std::vector<int32> *hyp);
  void ErrorRateMSeq(const std::vector<int> &frame_num_utt, const CuMatrixBase &net_out, std::vector<std::vector<int>> &label);
  void SetReportStep(int32 report_step) { report_step_ = report_step; }
  std::string Report();
  float NumErrorTokens() const { return error_num_; }
  int32 NumRefTokens() const { return ref_num_; }
  void StatOnly(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host, CuMatrix *diff);
  void StatAndLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host,
                        CuMatrix *diff);
  void StatAndAverageLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt,
                               const std::vector<float> &pzx_host, CuMatrix *diff);
  const std::vector<float> &LastCosts() const { return last_costs_; }  // -log p(z|x) per sequence of the last call
  double Obj() const { return obj_; }
  int32 Frames() const { return frames_; }
  int32 Sequences() const { return sequences_num_; }

 private:
  void ProgressReport();
  void AccumulateErrors(const std::vector<int32> &ref, const std::vector<int32> &hyp, int32 *err);
  int32 frames_, sequences_num_, ref_num_;
  float error_num_;
  int32 frames_progress_, ref_num_progress_;
  float error_num_progress_;
  int32 sequences_progress_;
  double obj_progress_;
  int32 report_step_;
  double obj_;
  double loss_sum_, loss_square_sum_, loss_sum_bak_, loss_square_sum_bak_;
  int32 normal_num_, stat_period_;
  std::vector<float> last_costs_;
};

}  // namespace aslp
