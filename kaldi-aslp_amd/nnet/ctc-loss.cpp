// ctc-loss.cpp -- Eesen-style Ctc loss (see ctc-loss.h; reference src/aslp-nnet/ctc-loss.cc).
#include "ctc-loss.h"

#include <cmath>
#include <sstream>

#include "aslp_ctc.h"
#include "warp-ctc.h"  // LevenshteinEditDistance

namespace aslp {

Ctc::Ctc()
    : frames_(0), sequences_num_(0), ref_num_(0), error_num_(0), frames_progress_(0), ref_num_progress_(0), error_num_progress_(0),
      sequences_progress_(0), obj_progress_(0.0), report_step_(100), obj_(0), loss_sum_(0), loss_square_sum_(0), loss_sum_bak_(0),
      loss_square_sum_bak_(0), normal_num_(0), stat_period_(100) {}

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
  obj_ += -pzx;
  obj_progress_ += -pzx;
  sequences_progress_ += 1;
  sequences_num_ += 1;
  frames_progress_ += net_out.NumRows();
  frames_ += net_out.NumRows();
  ProgressReport();
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
  ProgressReport();
}

void Ctc::ProgressReport() {
  if (sequences_progress_ >= report_step_) {
    ASLP_LOG << "Progress " << sequences_num_ << " sequences (" << frames_ / (100.0 * 3600) << "Hr):"
             << " Obj(log[Pzx]) = " << obj_progress_ / sequences_progress_ << " Obj(frame) = " << obj_progress_ / frames_progress_
             << " TokenAcc = " << 100.0 * (1.0 - error_num_progress_ / ref_num_progress_) << " %";
    sequences_progress_ = 0;
    frames_progress_ = 0;
    obj_progress_ = 0.0;
    error_num_progress_ = 0;
    ref_num_progress_ = 0;
  }
}

void Ctc::StatAndAverageLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt,
                                  const std::vector<float> &pzx_host, CuMatrix *diff) {  // :229-302
  const int32 num_sequence = frame_num_utt.size();
  for (int s = 0; s < num_sequence; s++) {
    if (normal_num_ < stat_period_ / 2) {  // warm-up: only sane costs enter the statistics; nothing is dropped
      if (std::isfinite(pzx_host[s]) && pzx_host[s] > 0 && pzx_host[s] < 3000) {
        normal_num_++;
        const double loss_per_frame = pzx_host[s] / frame_num_utt[s];
        loss_sum_ += loss_per_frame;
        loss_sum_bak_ += loss_per_frame;
        loss_square_sum_ += loss_per_frame * loss_per_frame;
        loss_square_sum_bak_ += loss_per_frame * loss_per_frame;
        obj_ += pzx_host[s];
        obj_progress_ += pzx_host[s];
      }
    } else {
      const double loss_per_frame = pzx_host[s] / frame_num_utt[s];
      const double mean = loss_sum_ / normal_num_;
      const double sigma = sqrt(loss_square_sum_ / normal_num_);
      if (std::isfinite(pzx_host[s]) && (loss_per_frame >= (mean - 6 * sigma) && loss_per_frame <= (mean + 6 * sigma)) &&
          (pzx_host[s] > 0 && pzx_host[s] < 3000)) {
        normal_num_++;
        loss_sum_ += loss_per_frame;
        loss_square_sum_ += loss_per_frame * loss_per_frame;
        obj_ += pzx_host[s];
        obj_progress_ += pzx_host[s];
        if (normal_num_ == stat_period_) {
          loss_sum_ -= loss_sum_bak_;
          loss_square_sum_ -= loss_square_sum_bak_;
          loss_sum_bak_ = loss_sum_;
          loss_square_sum_bak_ = loss_square_sum_;
          normal_num_ = stat_period_ / 2;
        }
      } else {
        ASLP_WARN << "Sequences " << (s < (int)utt.size() ? utt[s] : std::string("?")) << " obj is abnormal(sum " << pzx_host[s]
                  << " per_frame " << loss_per_frame << " mean " << loss_sum_ / normal_num_ << " sigma " << loss_square_sum_ / normal_num_
                  << "), drop it's diff and stat";
        for (int t = 0; t < frame_num_utt[s]; t++) diff->RowRange(t * num_sequence + s, 1).SetZero();
      }
    }
    frames_ += frame_num_utt[s];
    frames_progress_ += frame_num_utt[s];
  }
  double grad_sum = diff->Sum();
  if (!std::isfinite(grad_sum)) {
    ASLP_WARN << "DIFF FINITE: nan or inf ocurred in the diff, ignore";
    diff->SetZero();
  }
  sequences_progress_ += num_sequence;
  sequences_num_ += num_sequence;
}

void Ctc::StatAndLossCheck(const std::vector<std::string> &utt, const std::vector<int32> &frame_num_utt,
                           const std::vector<float> &pzx_host, CuMatrix *diff) {  // :304-329
  const int32 num_sequence = frame_num_utt.size();
  for (int s = 0; s < num_sequence; s++) {
    if (pzx_host[s] > 3000 || pzx_host[s] < 0) {
      ASLP_WARN << "Sequences " << (s < (int)utt.size() ? utt[s] : std::string("?")) << " obj is abnormal(" << pzx_host[s]
                << "), drop it's diff and stat";
      for (int t = 0; t < frame_num_utt[s]; t++) diff->RowRange(t * num_sequence + s, 1).SetZero();
    } else {
      obj_ += pzx_host[s];
      obj_progress_ += pzx_host[s];
    }
    frames_ += frame_num_utt[s];
    frames_progress_ += frame_num_utt[s];
  }
  sequences_progress_ += num_sequence;
  sequences_num_ += num_sequence;
}

void Ctc::StatOnly(const std::vector<std::string> &, const std::vector<int32> &frame_num_utt, const std::vector<float> &pzx_host,
                   CuMatrix *) {  // :331-344
  const int32 num_sequence = frame_num_utt.size();
  for (int s = 0; s < num_sequence; s++) {
    obj_ += pzx_host[s];
    obj_progress_ += pzx_host[s];
    frames_progress_ += frame_num_utt[s];
    frames_ += frame_num_utt[s];
  }
  sequences_progress_ += num_sequence;
  sequences_num_ += num_sequence;
}

void Ctc::AccumulateErrors(const std::vector<int32> &ref, const std::vector<int32> &hyp, int32 *err) {
  int32 ins, del, sub;
  *err = LevenshteinEditDistance(ref, hyp, &ins, &del, &sub);
  error_num_ += *err;
  ref_num_ += ref.size();
  error_num_progress_ += *err;
  ref_num_progress_ += ref.size();
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
  int32 err;
  AccumulateErrors(label, *hyp, &err);
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
    int32 err;
    AccumulateErrors(label[s], hyp, &err);
  }
}

std::string Ctc::Report() {  // :426-432
  std::ostringstream oss;
  oss << " Obj(log[Pzx]) = " << obj_ / sequences_num_ << " Obj(frame) = " << obj_ / frames_ << " TOKEN_ACCURACY >> "
      << 100.0 * (1.0 - error_num_ / ref_num_) << " % <<";
  return oss.str();
}

}  // namespace aslp
