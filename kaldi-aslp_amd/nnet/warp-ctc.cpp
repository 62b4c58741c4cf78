// warp-ctc.cpp -- WarpCtc loss wrapper (see warp-ctc.h; reference src/aslp-nnet/warp-ctc.cc).
#include "warp-ctc.h"
#include "scratch.h"

#include <cmath>
#include <sstream>

#include "aslp_ctc.h"

namespace aslp {

WarpCtc::WarpCtc() : use_gpu_(true), book_(500) {}

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
  book_.ProgressReport();
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
    book_.CountTokens(LevenshteinEditDistance(label[s], hyp_seq, &ins, &del, &sub), label[s].size());
  }
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
