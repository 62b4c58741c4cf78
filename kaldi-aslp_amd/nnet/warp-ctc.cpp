// warp-ctc.cpp -- WarpCtc loss wrapper (see warp-ctc.h; reference src/aslp-nnet/warp-ctc.cc).
#include "warp-ctc.h"
#include "scratch.h"

#include <cmath>
#include <sstream>

#include "aslp_ctc.h"

namespace aslp {

WarpCtc::WarpCtc()
    : frames_(0), sequences_num_(0), ref_num_(0), error_num_(0), frames_progress_(0), ref_num_progress_(0), error_num_progress_(0),
      sequences_progress_(0), obj_progress_(0.0), report_step_(100), obj_(0), use_gpu_(true), window_(500) {}

void WarpCtc::Eval(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const CuMatrixBase &net_out,
                   const std::vector<std::vector<int32>> &labels, CuMatrix *diff) {  // warp-ctc.cc:33-46
  ASLP_ASSERT(diff != NULL);
  if (use_gpu_) EvalGpu(utt, frame_num_utt, net_out, labels, diff);
  else EvalCpu(utt, frame_num_utt, net_out, labels, diff);
}

void WarpCtc::EvalCpu(const std::vector<std::string> &, const std::vector<int32> &, const CuMatrixBase &, const std::vector<std::vector<int32>> &,
                      CuMatrix *) {
  ASLP_ERR << "WarpCtc::EvalCpu: this library has no host CTC path (device only); call SetUseGpu(true)";
}

void WarpCtc::EvalGpu(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const CuMatrixBase &net_out,
                      const std::vector<std::vector<int32>> &labels, CuMatrix *diff) {  // warp-ctc.cc:48-204
  diff->Resize(net_out.NumRows(), net_out.NumCols());  // zeroed: frames past an utterance's end keep a zero diff
  const int minibatch = frame_num_utt.size();
  ASLP_ASSERT(minibatch > 0 && (int)labels.size() >= minibatch);
  std::vector<int> flat_labels, label_lengths;
  int max_frames = 0;
  for (int i = 0; i < minibatch; i++) {
    const std::vector<int> &l = labels[i];
    for (size_t j = 0; j < l.size(); j++) ASLP_ASSERT(l[j] < net_out.NumCols());
    flat_labels.insert(flat_labels.end(), l.begin(), l.end());
    label_lengths.push_back(l.size());
    if (frame_num_utt[i] > max_frames) max_frames = frame_num_utt[i];
  }
  ASLP_ASSERT(max_frames * minibatch <= net_out.NumRows());
  if (flat_labels.empty()) flat_labels.push_back(0);  // keep the pointer valid for all-empty label sets
  last_costs_.assign(minibatch, 0.0f);
  ctcStatus_t st;
  {
    RegionScope timed("ctc_loss");
    st = aslp_ctc_loss_strided(net_out.Data(), net_out.Stride(), diff->Data(), diff->Stride(), flat_labels.data(),
                               label_lengths.data(), frame_num_utt.data(), net_out.NumCols(), minibatch, last_costs_.data());
  }
  if (st != CTC_STATUS_SUCCESS) ASLP_ERR << "Error: compute_ctc_loss: " << ctcGetStatusString(st);

  bool checked_finite = false;
#if WARP_CTC_GRAD_CHECK == WARP_CTC_SUM_LOSS_CHECK
  StatAndLossCheck(utt, frame_num_utt, last_costs_, diff);
#elif WARP_CTC_GRAD_CHECK == WARP_CTC_AVG_LOSS_CHECK
  StatAndAverageLossCheck(utt, frame_num_utt, last_costs_, diff);
  checked_finite = true;  // that check already zeroed a non-finite diff; clipping cannot make it non-finite again
#else
  StatOnly(utt, frame_num_utt, last_costs_, diff);
#endif
  diff->ApplyFloor(-1.0);  // :171-173
  diff->ApplyCeiling(1.0);
  if (!checked_finite) {
    double grad_sum = diff->Sum();
    ASLP_ASSERT(std::isfinite(grad_sum));
  }
  ProgressReport();
}

void WarpCtc::ProgressReport() {   // the reference's line, warp-ctc.cc:188-203
  if (sequences_progress_ < report_step_) return;
  ASLP_LOG << "Progress " << sequences_num_ << " sequences (" << frames_ / (100.0 * 3600) << "Hr):"
           << " Obj(log[Pzx]) = " << obj_progress_ / sequences_progress_ << " Obj(frame) = " << obj_progress_ / frames_progress_
           << " TokenAcc = " << 100.0 * (1.0 - error_num_progress_ / ref_num_progress_) << " %";
  sequences_progress_ = frames_progress_ = ref_num_progress_ = 0;
  obj_progress_ = 0.0;
  error_num_progress_ = 0;
}

// ---- what happens to the costs of a batch: three policies over the same bookkeeping ---------------------------------------------------
// (behaviour of warp-ctc.cc:288-365 / 446-470 / 472-485: which utterances count, which are dropped, what is logged)
void WarpCtc::Count(int32 frames, bool kept, double obj) {
  if (kept) { obj_ += obj; obj_progress_ += obj; }
  frames_ += frames;
  frames_progress_ += frames;
}
void WarpCtc::CountBatch(int32 num_sequence) {
  sequences_progress_ += num_sequence;
  sequences_num_ += num_sequence;
}
namespace {
// the rows of utterance s in a (t, stream)-interleaved matrix
void DropUtterance(CuMatrix *diff, int s, int num_sequence, int frames) {
  for (int t = 0; t < frames; t++) diff->RowRange(t * num_sequence + s, 1).SetZero();
}
const std::string &NameOf(const std::vector<std::string> &utt, int s) {
  static const std::string unknown("?");
  return s < (int)utt.size() ? utt[s] : unknown;
}
const double kCostCeiling = 3000.0;   // a sequence cost outside (0, 3000) is never trusted
}  // namespace

// every utterance's cost per frame is held against the window of the accepted ones: outside mean +- 6 "sigma" (or non-finite, or
// outside (0, 3000)) it neither trains nor counts; the first half window is accepted unseen
void WarpCtc::StatAndAverageLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt,
                                      const std::vector<float> &pzx_host, CuMatrix *diff) {
  const int32 num_sequence = frame_num_utt.size();
  for (int s = 0; s < num_sequence; s++) {
    const double cost = pzx_host[s], per_frame = cost / frame_num_utt[s];
    bool keep = true;
    if (!window_.WarmingUp()) {
      const double mean = window_.Mean(), spread = 6 * window_.RootMeanSquare();
      keep = std::isfinite(pzx_host[s]) && per_frame >= mean - spread && per_frame <= mean + spread && cost > 0 && cost < kCostCeiling;
    }
    if (keep) {
      window_.Add(per_frame);
    } else {
      ASLP_WARN << "Sequences " << NameOf(utt, s) << " obj is abnormal(sum " << pzx_host[s] << " per_frame " << per_frame << " mean "
                << window_.Mean() << " sigma " << window_.sum_sq / window_.count << "), drop it's diff and stat";
      DropUtterance(diff, s, num_sequence, frame_num_utt[s]);
    }
    Count(frame_num_utt[s], keep, cost);
  }
  if (!std::isfinite(diff->Sum())) {
    ASLP_WARN << "DIFF FINITE: nan or inf ocurred in the diff, ignore";
    diff->SetZero();
  }
  CountBatch(num_sequence);
}

// only the absolute test: a cost outside [0, 3000] is dropped
void WarpCtc::StatAndLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt,
                               const std::vector<float> &pzx_host, CuMatrix *diff) {
  const int32 num_sequence = frame_num_utt.size();
  for (int s = 0; s < num_sequence; s++) {
    const bool keep = !(pzx_host[s] > kCostCeiling || pzx_host[s] < 0);
    if (!keep) {
      ASLP_WARN << "Sequences " << NameOf(utt, s) << " obj is abnormal(" << pzx_host[s] << "), drop it's diff and stat";
      DropUtterance(diff, s, num_sequence, frame_num_utt[s]);
    }
    Count(frame_num_utt[s], keep, pzx_host[s]);
  }
  CountBatch(num_sequence);
}

// no test at all
void WarpCtc::StatOnly(const std::vector<std::string> &, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host,
                       CuMatrix *) {
  const int32 num_sequence = frame_num_utt.size();
  for (int s = 0; s < num_sequence; s++) Count(frame_num_utt[s], true, pzx_host[s]);
  CountBatch(num_sequence);
}

void WarpCtc::ErrorRate(const std::vector<int> &frame_num_utt, const CuMatrixBase &net_out, std::vector<std::vector<int>> &label) {  // :487-529
  CuArray<int32> maxid;
  net_out.FindRowMaxId(&maxid);
  std::vector<int32> data;
  maxid.CopyToVec(&data);
  const int32 num_seq = frame_num_utt.size();
  for (int32 s = 0; s < num_seq; s++) {
    const int32 num_frame = frame_num_utt[s];
    // best path: collapse repeats, then drop blanks (label 0)
    std::vector<int32> hyp_seq;
    int32 prev = -1;
    for (int32 f = 0; f < num_frame; f++) {
      const int32 id = data[f * num_seq + s];
      if (f == 0 || id != prev) {
        if (id != 0) hyp_seq.push_back(id);
      }
      prev = id;
    }
    int32 ins, del, sub;
    const int32 err = LevenshteinEditDistance(label[s], hyp_seq, &ins, &del, &sub);
    error_num_ += err;
    ref_num_ += label[s].size();
    error_num_progress_ += err;
    ref_num_progress_ += label[s].size();
  }
}

std::string WarpCtc::Report() {  // :531-538
  std::ostringstream oss;
  oss << " Obj(log[Pzx]) = " << obj_ / sequences_num_ << " Obj(frame) = " << obj_ / frames_ << " TOKEN_ACCURACY >> "
      << 100.0 * (1.0 - error_num_ / ref_num_) << " % <<";
  return oss.str();
}

int32 LevenshteinEditDistance(const std::vector<int32> &ref, const std::vector<int32> &hyp, int32 *ins, int32 *del, int32 *sub) {
  // Row-by-row dynamic programme over hyp positions, carrying the error-type counts of the best
  // alignment; tie-breaking as util/edit-distance-inl.h:104-118 (substitution only if strictly best,
  // then deletion if strictly better than insertion).
  struct Cell { int32 i, d, s, cost; };
  const size_t R = ref.size();
  std::vector<Cell> prev(R + 1), cur(R + 1);
  for (size_t r = 0; r <= R; r++) prev[r] = Cell{0, (int32)r, 0, (int32)r};
  for (size_t h = 1; h <= hyp.size(); h++) {
    cur[0] = prev[0];
    cur[0].i++;
    cur[0].cost++;
    for (size_t r = 1; r <= R; r++) {
      const bool diff_sym = hyp[h - 1] != ref[r - 1];
      const int32 c_ins = prev[r].cost + 1, c_del = cur[r - 1].cost + 1, c_sub = prev[r - 1].cost + (diff_sym ? 1 : 0);
      if (c_sub < c_ins && c_sub < c_del) {
        cur[r] = prev[r - 1];
        if (diff_sym) cur[r].s++;
        cur[r].cost = c_sub;
      } else if (c_del < c_ins) {
        cur[r] = cur[r - 1];
        cur[r].d++;
        cur[r].cost = c_del;
      } else {
        cur[r] = prev[r];
        cur[r].i++;
        cur[r].cost = c_ins;
      }
    }
    prev.swap(cur);
  }
  *ins = prev[R].i; *del = prev[R].d; *sub = prev[R].s;
  return prev[R].cost;
}

}  // namespace aslp
