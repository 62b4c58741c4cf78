// ctc-loss.h -- the Eesen-style Ctc loss of the host engine (src/aslp-nnet/ctc-loss.{h,cc}).
// Input is the Softmax output; Eval (one sequence) / EvalParallel (padded multi-sequence batch, row =
// t*num_seq + s), ErrorRate / ErrorRateMSeq, the abnormal-loss filter (window of 100 utterances here; ctc-cost-book.h) and the
// Report() string follow the reference.  Mechanism: the lattice runs in two launches over all sequences
// (aslp_eesen_ctc_mseq) instead of 2T row kernels + the O(T*A*(2L+1)) error kernel + 6 elementwise passes.
#pragma once
#include <string>
#include <vector>

#include "ctc-cost-book.h"
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
  void SetReportStep(int32 report_step) { book_.SetReportStep(report_step); }
  std::string Report() { return book_.Report(); }
  float NumErrorTokens() const { return book_.NumErrorTokens(); }
  int32 NumRefTokens() const { return book_.NumRefTokens(); }
  // the reference's three bookkeeping variants (ctc-loss.cc:229-302, 304-329, 331-344), over the shared book
  void StatOnly(const std::vector<std::string> &, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host, CuMatrix *) {
    book_.AcceptAll(frame_num_utt, pzx_host);
  }
  void StatAndLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host,
                        CuMatrix *diff) {
    book_.DropOutOfRange(utt, frame_num_utt, pzx_host, diff);
  }
  void StatAndAverageLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt,
                               const std::vector<float> &pzx_host, CuMatrix *diff) {
    book_.DropOutliers(utt, frame_num_utt, pzx_host, diff, /*sane_only_during_warmup=*/true);
  }
  const std::vector<float> &LastCosts() const { return last_costs_; }  // -log p(z|x) per sequence of the last call
  double Obj() const { return book_.Obj(); }
  int32 Frames() const { return book_.Frames(); }
  int32 Sequences() const { return book_.Sequences(); }

 private:
  CtcCostBook book_;   // window of 100 utterances (ctc-loss.cc: stat_period_)
  std::vector<float> last_costs_;
};

}  // namespace aslp
