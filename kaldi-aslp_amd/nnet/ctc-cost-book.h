// ctc-cost-book.h -- what happens to the per-utterance costs of a CTC batch, shared by the two losses (WarpCtc: warp-ctc.h, Ctc: ctc-loss.h).
// Behaviour of the reference's three bookkeeping variants (src/aslp-nnet/warp-ctc.cc:288-365 / 446-485 and ctc-loss.cc:229-344): which
// utterances train and count, which are dropped with a zero diff, what is logged, the totals behind Report() and the progress line.
// The two reference classes carry the same text twice with different window lengths (500 utterances for WarpCtc, 100 for Ctc); here it
// exists once.
#pragma once
#include <cmath>
#include <sstream>
#include <string>
#include <vector>

#include "cu-matrix.h"

namespace aslp {

// Running statistics of the per-frame cost of the utterances accepted so far, over a sliding window of `period` utterances that
// restarts from its younger half when full.
struct CostWindow {
  explicit CostWindow(int32 period_) : period(period_) {}
  bool WarmingUp() const { return count < period / 2; }
  double Mean() const { return sum / count; }
  double MeanSquare() const { return sum_sq / count; }
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

class CtcCostBook {
 public:
  explicit CtcCostBook(int32 window_period) : window_(window_period) {}

  // ---- three policies over one batch; diff rows are (t, stream)-interleaved, row = t * num_sequence + s ----
  // no test at all
  void AcceptAll(const std::vector<int32> &frames, const std::vector<float> &cost) {
    for (size_t s = 0; s < frames.size(); s++) CountUtterance(frames[s], true, cost[s]);
    CountBatch(frames.size());
  }
  // only the absolute test: a cost outside [0, 3000] neither trains nor counts
  void DropOutOfRange(const std::vector<std::string> &utt, const std::vector<int32> &frames, const std::vector<float> &cost, CuMatrix *diff) {
    const int32 n = frames.size();
    for (int s = 0; s < n; s++) {
      const bool keep = !(cost[s] > CostCeiling() || cost[s] < 0);
      if (!keep) {
        ASLP_WARN << "Sequences " << NameOf(utt, s) << " obj is abnormal(" << cost[s] << "), drop it's diff and stat";
        ZeroUtterance(diff, s, n, frames[s]);
      }
      CountUtterance(frames[s], keep, cost[s]);
    }
    CountBatch(n);
  }
  // every utterance's cost per frame is held against the window of the accepted ones: outside mean +- 6 "sigma" (or non-finite, or outside
  // (0, 3000)) it neither trains nor counts.  While the window warms up nothing is dropped; with sane_only_during_warmup only finite costs
  // inside (0, 3000) enter the window and the totals then (Ctc's variant, ctc-loss.cc:236-247), without it every cost does (WarpCtc's,
  // warp-ctc.cc:296-307).  A diff that is not finite afterwards is zeroed as a whole.
  void DropOutliers(const std::vector<std::string> &utt, const std::vector<int32> &frames, const std::vector<float> &cost, CuMatrix *diff,
                    bool sane_only_during_warmup) {
    const int32 n = frames.size();
    for (int s = 0; s < n; s++) {
      const double c = cost[s], per_frame = c / frames[s];
      const bool sane = std::isfinite(cost[s]) && c > 0 && c < CostCeiling();
      if (window_.WarmingUp()) {
        const bool enters = sane || !sane_only_during_warmup;
        if (enters) window_.Add(per_frame);
        CountUtterance(frames[s], enters, c);
        continue;
      }
      const double mean = window_.Mean(), spread = 6 * window_.RootMeanSquare();
      const bool keep = sane && per_frame >= mean - spread && per_frame <= mean + spread;
      if (keep) {
        window_.Add(per_frame);
      } else {
        ASLP_WARN << "Sequences " << NameOf(utt, s) << " obj is abnormal(sum " << cost[s] << " per_frame " << per_frame << " mean " << window_.Mean()
                  << " sigma " << window_.MeanSquare() << "), drop it's diff and stat";
        ZeroUtterance(diff, s, n, frames[s]);
      }
      CountUtterance(frames[s], keep, c);
    }
    if (!std::isfinite(diff->Sum())) {
      ASLP_WARN << "DIFF FINITE: nan or inf ocurred in the diff, ignore";
      diff->SetZero();
    }
    CountBatch(n);
  }

  // one sequence outside any batch policy (Ctc::Eval)
  void CountSingle(int32 frames, double obj) { CountUtterance(frames, true, obj); CountBatch(1); }
  void CountTokens(int32 errors, int32 ref_tokens) {
    error_num_ += errors; error_num_progress_ += errors;
    ref_num_ += ref_tokens; ref_num_progress_ += ref_tokens;
  }

  // the reference's progress line (warp-ctc.cc:188-203, ctc-loss.cc:213-226), every report_step sequences
  void ProgressReport() {
    if (sequences_progress_ < report_step_) return;
    ASLP_LOG << "Progress " << sequences_num_ << " sequences (" << frames_ / (100.0 * 3600) << "Hr):"
             << " Obj(log[Pzx]) = " << obj_progress_ / sequences_progress_ << " Obj(frame) = " << obj_progress_ / frames_progress_
             << " TokenAcc = " << 100.0 * (1.0 - error_num_progress_ / ref_num_progress_) << " %";
    sequences_progress_ = frames_progress_ = ref_num_progress_ = 0;
    obj_progress_ = 0.0;
    error_num_progress_ = 0;
  }
  std::string Report() const {   // warp-ctc.cc:531-538, ctc-loss.cc:426-432 (the schedulers grep "TOKEN_ACCURACY >>")
    std::ostringstream oss;
    oss << " Obj(log[Pzx]) = " << obj_ / sequences_num_ << " Obj(frame) = " << obj_ / frames_ << " TOKEN_ACCURACY >> "
        << 100.0 * (1.0 - error_num_ / ref_num_) << " % <<";
    return oss.str();
  }
  void SetReportStep(int32 n) { report_step_ = n; }
  double Obj() const { return obj_; }
  int32 Frames() const { return frames_; }
  int32 Sequences() const { return sequences_num_; }
  float NumErrorTokens() const { return error_num_; }
  int32 NumRefTokens() const { return ref_num_; }
  const CostWindow &Window() const { return window_; }

 private:
  static double CostCeiling() { return 3000.0; }   // a sequence cost outside (0, 3000) is never trusted
  static const std::string &NameOf(const std::vector<std::string> &utt, int s) {
    static const std::string unknown("?");
    return s < (int)utt.size() ? utt[s] : unknown;
  }
  static void ZeroUtterance(CuMatrix *diff, int s, int num_sequence, int frames) {
    for (int t = 0; t < frames; t++) diff->RowRange(t * num_sequence + s, 1).SetZero();
  }
  // one utterance into the totals and the progress counters (its cost only when it is kept)
  void CountUtterance(int32 frames, bool kept, double obj) {
    if (kept) { obj_ += obj; obj_progress_ += obj; }
    frames_ += frames;
    frames_progress_ += frames;
  }
  void CountBatch(int32 num_sequence) {
    sequences_progress_ += num_sequence;
    sequences_num_ += num_sequence;
  }
  CostWindow window_;
  int32 frames_ = 0, sequences_num_ = 0, ref_num_ = 0;
  float error_num_ = 0;
  int32 frames_progress_ = 0, ref_num_progress_ = 0, sequences_progress_ = 0;
  float error_num_progress_ = 0;
  double obj_ = 0.0, obj_progress_ = 0.0;
  int32 report_step_ = 100;
};

}  // namespace aslp
