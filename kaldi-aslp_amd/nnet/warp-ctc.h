// warp-ctc.h -- the WarpCtc loss wrapper of the host engine (src/aslp-nnet/warp-ctc.{h,cc}).
// Same public interface, statistics, abnormal-loss filter and Report() string as the reference
// (the bash schedulers grep "TOKEN_ACCURACY >>").  Mechanism: the loss runs directly on the
// row-padded network output and writes straight into `diff` (aslp_ctc_loss_strided), so the
// reference's per-call cudaMalloc x3, row-wise de-stride copy and per-frame copy-back
// (warp-ctc.cc:85-95, 105-113, 139-147) disappear.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "ctc-cost-book.h"
#include "cu-matrix.h"

namespace aslp {

#define WARP_CTC_SUM_LOSS_CHECK 0
#define WARP_CTC_AVG_LOSS_CHECK 1
#define WARP_CTC_NONE_LOSS_CHECK 3
#define WARP_CTC_GRAD_CHECK 1  // the reference ships with the average-loss check compiled in (warp-ctc.h:27)

class WarpCtc {
 public:
  WarpCtc();
  void Eval(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const CuMatrixBase &net_out,
            const std::vector<std::vector<int32>> &labels, CuMatrix *diff);
  void EvalGpu(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const CuMatrixBase &net_out,
               const std::vector<std::vector<int32>> &labels, CuMatrix *diff);
  // No host fallback exists in this library: fails loudly (the reference's EvalCpu ran cpu_ctc.h on the host)
  void EvalCpu(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const CuMatrixBase &net_out,
               const std::vector<std::vector<int32>> &labels, CuMatrix *diff);
  void ErrorRate(const std::vector<int> &frame_num_utt, const CuMatrixBase &net_out, std::vector<std::vector<int>> &label);
  void SetReportStep(int32 report_step) { book_.SetReportStep(report_step); }
  std::string Report() { return book_.Report(); }
  float NumErrorTokens() const { return book_.NumErrorTokens(); }
  int32 NumRefTokens() const { return book_.NumRefTokens(); }
  void SetUseGpu(bool use_gpu) { use_gpu_ = use_gpu; }
  // the reference's three bookkeeping variants (warp-ctc.cc:288-365, 446-470, 472-485), over the shared book
  void StatOnly(const std::vector<std::string> &, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host, CuMatrix *) {
    book_.AcceptAll(frame_num_utt, pzx_host);
  }
  void StatAndLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host,
                        CuMatrix *diff) {
    book_.DropOutOfRange(utt, frame_num_utt, pzx_host, diff);
  }
  void StatAndAverageLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt,
                               const std::vector<float> &pzx_host, CuMatrix *diff) {
    book_.DropOutliers(utt, frame_num_utt, pzx_host, diff, /*sane_only_during_warmup=*/false);
  }
  // accessors for tests / the C ABI
  const std::vector<float> &LastCosts() const { return last_costs_; }
  double Obj() const { return book_.Obj(); }
  int32 Frames() const { return book_.Frames(); }
  int32 Sequences() const { return book_.Sequences(); }

 private:
  bool use_gpu_;
  CtcCostBook book_;   // window of 500 utterances (warp-ctc.cc: stat_period_)
  std::vector<float> last_costs_;
};

// util/edit-distance-inl.h:80-155 (counts insertions, deletions, substitutions of hyp against ref)
int32 LevenshteinEditDistance(const std::vector<int32> &ref, const std::vector<int32> &hyp, int32 *ins, int32 *del, int32 *sub);

}  // namespace aslp
