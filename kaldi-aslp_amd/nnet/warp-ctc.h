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
  void SetReportStep(int32 report_step) { report_step_ = report_step; }
  std::string Report();
  float NumErrorTokens() const { return error_num_; }
  int32 NumRefTokens() const { return ref_num_; }
  void SetUseGpu(bool use_gpu) { use_gpu_ = use_gpu; }
  void StatOnly(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host, CuMatrix *diff);
  void StatAndLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host,
                        CuMatrix *diff);
  void StatAndAverageLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt,
                               const std::vector<float> &pzx_host, CuMatrix *diff);
  // accessors for tests / the C ABI
  const std::vector<float> &LastCosts() const { return last_costs_; }
  double Obj() const { return obj_; }
  int32 Frames() const { return frames_; }
  int32 Sequences() const { return sequences_num_; }

 private:
  void ProgressReport();
  int32 frames_, sequences_num_, ref_num_;
  float error_num_;
  int32 frames_progress_, ref_num_progress_;
  float error_num_progress_;
  int32 sequences_progress_;
  double obj_progress_;
  int32 report_step_;
  double obj_;
  bool use_gpu_;
  // Running statistics of the per-frame cost of the utterances accepted so far, over a sliding window of `period` utterances that
  // restarts from its younger half when full (what the reference keeps in loss_sum_ / loss_sum_bak_ / normal_num_, warp-ctc.cc:288-365)
  struct CostWindow {
    explicit CostWindow(int32 period_) : period(period_) {}
    bool WarmingUp() const { return count < period / 2; }
    double Mean() const { return sum / count; }
    double RootMeanSquare() const { return sqrt(sum_sq / count); }   // the reference's "sigma": no mean subtracted
    void Add(double x) {
      const bool warm = WarmingUp();
      count++;
      sum += x;
      sum_sq += x * x;
      if (warm) { young_sum += x; young_sum_sq += x * x; }   // the first half window is what survives the first restart
      if (count == period) {   // keep the younger half
        sum -= young_sum;
        sum_sq -= young_sum_sq;
        young_sum = sum;
        young_sum_sq = sum_sq;
        count = period / 2;
      }
    }
    int32 period, count = 0;
    double sum = 0.0, sum_sq = 0.0, young_sum = 0.0, young_sum_sq = 0.0;
  };
  CostWindow window_;
  // one utterance into the totals and the progress counters (obj only when it is kept)
  void Count(int32 frames, bool kept, double obj);
  void CountBatch(int32 num_sequence);
  std::vector<float> last_costs_;
};

// util/edit-distance-inl.h:80-155 (counts insertions, deletions, substitutions of hyp against ref)
int32 LevenshteinEditDistance(const std::vector<int32> &ref, const std::vector<int32> &hyp, int32 *ins, int32 *del, int32 *sub);

}  // namespace aslp
