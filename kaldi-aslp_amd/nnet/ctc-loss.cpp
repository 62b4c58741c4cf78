// ctc-loss.cpp -- Eesen-style Ctc loss (see ctc-loss.h; reference src/aslp-nnet/ctc-loss.cc).
#include "ctc-loss.h"

#include <cmath>
#include <sstream>

#include "aslp_ctc.h"
#include "warp-ctc.h"  // LevenshteinEditDistance

namespace aslp {

Ctc::Ctc() : book_(100) {}

namespace {
// runs the lattice; diff is zeroed first, costs[s] = -log p(z|x)
void RunLattice(const CuMatrixBase &net_out, const std::vector<int32> &frame_num_utt, const std::vector<std::vector<int32>> &label,
                CuMatrix *diff, std::vector<float> *costs) {
  diff->Resize(net_out.NumRows(), net_out.NumCols());
  const int num_sequence = frame_num_utt.size();
  ASLP_ASSERT(num_sequence > 0 && net_out.NumRows() % num_sequence == 0);  // ctc-loss.cc:124
  ASLP_ASSERT((int)label.size() >= num_sequence);
  std::vector<int> flat, lens;
  for (int s = 0; s < num_sequence; s++) {
    for (size_t l = 0; l < label[s].size(); l++)
      if (label[s][l] >= net_out.NumCols()) ASLP_ERR << "label gt outdim " << label[s][l] << " " << net_out.NumCols();  // :141-143
    flat.insert(flat.end(), label[s].begin(), label[s].end());
    lens.push_back(label[s].size());
    ASLP_ASSERT(frame_num_utt[s] * num_sequence <= net_out.NumRows());
  }
  if (flat.empty()) flat.push_back(0);
  costs->assign(num_sequence, 0.0f);
  ctcStatus_t st = aslp_eesen_ctc_mseq(net_out.Data(), net_out.Stride(), diff->Data(), diff->Stride(), flat.data(), lens.data(),
                                       frame_num_utt.data(), net_out.NumCols(), num_sequence, costs->data());
  if (st != CTC_STATUS_SUCCESS) ASLP_ERR << "Ctc: lattice computation failed: " << ctcGetStatusString(st);
  for (float &c : *costs) c = -c;  // pzx.Scale(-1), ctc-loss.cc:200
}
}  // namespace

void Ctc::Eval(const CuMatrixBase &net_out, const std::vector<int32> &label, CuMatrix *diff) {  // ctc-loss.cc:31-109
  std::vector<int32> frames(1, net_out.NumRows());
  std::vector<std::vector<int32>> labels(1, label);
  RunLattice(net_out, frames, labels, diff, &last_costs_);
  diff->ApplyFloor(-1.0);
  diff->ApplyCeiling(1.0);
  double pzx = -last_costs_[0];
  if (pzx < -10000) pzx = -10000;  // :85-86
  if (pzx > 10000) pzx = 10000;
  book_.CountSingle(net_out.NumRows(), -pzx);
  book_.ProgressReport();
}

void Ctc::EvalParallel(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt, const CuMatrixBase &net_out,
                       std::vector<std::vector<int32>> &label, CuMatrix *diff) {  // ctc-loss.cc:115-227
  RunLattice(net_out, frame_num_utt, label, diff, &last_costs_);
#if CTC_GRAD_CHECK == SUM_LOSS_CHECK
  StatAndLossCheck(utt, frame_num_utt, last_costs_, diff);
#elif CTC_GRAD_CHECK == AVG_LOSS_CHECK
  StatAndAverageLossCheck(utt, frame_num_utt, last_costs_, diff);
#else
  StatOnly(utt, frame_num_utt, last_costs_, diff);
#endif
  diff->ApplyFloor(-1.0);  // :210-211
  diff->ApplyCeiling(1.0);
  book_.ProgressReport();
}

// best path of one sequence: argmax ids with stride `step` starting at `first`, repeats collapsed, blanks dropped
static void BestPath(const std::vector<int32> &ids, int first, int step, int frames, std::vector<int32> *hyp) {
  hyp->clear();
  int32 prev = -1;
  for (int f = 0; f < frames; f++) {
    const int32 id = ids[first + f * step];
    if ((f == 0 || id != prev) && id != 0) hyp->push_back(id);
    prev = id;
  }
}

void Ctc::ErrorRate(const CuMatrixBase &net_out, const std::vector<int32> &label, float *err_rate, std::vector<int32> *hyp) {  // :346-383
  CuArray<int32> maxid;
  net_out.FindRowMaxId(&maxid);
  std::vector<int32> data;
  maxid.CopyToVec(&data);
  BestPath(data, 0, 1, data.size(), hyp);
  int32 ins, del, sub;
  const int32 err = LevenshteinEditDistance(label, *hyp, &ins, &del, &sub);
  book_.CountTokens(err, label.size());
  *err_rate = (100.0 * err) / label.size();
}

void Ctc::ErrorRateMSeq(const std::vector<int> &frame_num_utt, const CuMatrixBase &net_out, std::vector<std::vector<int>> &label) {  // :385-424
  CuArray<int32> maxid;
  net_out.FindRowMaxId(&maxid);
  std::vector<int32> data;
  maxid.CopyToVec(&data);
  const int32 num_seq = frame_num_utt.size();
  std::vector<int32> hyp;
  for (int32 s = 0; s < num_seq; s++) {
    BestPath(data, s, num_seq, frame_num_utt[s], &hyp);
    int32 ins, del, sub;
    book_.CountTokens(LevenshteinEditDistance(label[s], hyp, &ins, &del, &sub), label[s].size());
  }
}

}  // namespace aslp
