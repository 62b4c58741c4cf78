// nnet-loss.cpp -- follows src/aslp-nnet/nnet-loss.cc (line cited per function).
#include "nnet-loss.h"
#include "split16.h"

#include <cmath>
#include <map>

namespace aslp {

static void CheckK() {
  char buf[512];
  if (aslp_get_last_error(buf, sizeof(buf))) ASLP_ERR << buf;
}

void PosteriorToMatrix(const Posterior &post, int32 num_cols, CuMatrix *mat) {
  // nnet-utils.h:160-177: zero matrix, m(t, col) = weight (assignment: a later duplicate wins)
  int32 num_rows = post.size();
  mat->Resize(num_rows, num_cols, kSetZero);
  std::vector<int32> rows, cols;
  std::vector<float> vals;
  for (int32 t = 0; t < num_rows; t++) {
    std::map<int32, float> last;
    for (size_t i = 0; i < post[t].size(); i++) {
      int32 col = post[t][i].first;
      if (col >= num_cols) ASLP_ERR << "Out-of-bound Posterior element with index " << col << ", higher than number of columns " << num_cols;
      last[col] = post[t][i].second;
    }
    for (auto &kv : last) { rows.push_back(t); cols.push_back(kv.first); vals.push_back(kv.second); }
  }
  if (rows.empty()) return;
  CuArray<int32> r(rows), c(cols);
  CuVector v;
  HostVector hv;
  hv.data = vals;
  v = hv;
  aslp_scatter_add(mat->Data(), mat->Dim(), r.Data(), c.Data(), v.Data(), (int)rows.size());
  CheckK();
}

Xent::Xent()
    : frames_(0.0), correct_(0.0), loss_(0.0), entropy_(0.0), likelyhood_(0.0), frames_progress_(0.0), loss_progress_(0.0),
      entropy_progress_(0.0), likelyhood_progress_(0.0), rows_since_progress_(0.0), dirty_(false) {
  stats_.Resize(5, kSetZero);
}

double *Xent::PendingRowStats(int32 rows) {
  static const int32 kMaxBatches = 32, kMaxRows = 1 << 16;
  if (rows != pending_rows_) {
    FlushPending();
    pending_rows_ = rows;
    pending_cap_ = std::min(kMaxBatches, kMaxRows / std::max(rows, 1));
    if (pending_cap_ >= 2 && pending_.Dim() < pending_cap_ * rows * 5) pending_.Resize(pending_cap_ * rows * 5, kUndefined);
  }
  if (pending_cap_ < 2) return nullptr;
  if (pending_batches_ == pending_cap_) FlushPending();
  return pending_.Data() + (size_t)(pending_batches_++) * rows * 5;
}
void Xent::FlushPending() {
  if (pending_batches_ > 0) aslp_xent_sum_rowstats(pending_.Data(), pending_rows_, pending_batches_, stats_.Data());
  pending_batches_ = 0;
}

void Xent::Fetch() {
  if (!dirty_) return;
  FlushPending();
  double h[5];
  stats_.CopyToHost(h);
  stats_.SetZero();
  dirty_ = false;
  if (!std::isfinite(h[2])) ASLP_ERR << "Xent: cross-entropy is not finite";    // nnet-loss.cc:124-126
  if (!std::isfinite(h[3])) ASLP_ERR << "Xent: target entropy is not finite";
  if (!std::isfinite(h[4])) ASLP_ERR << "Xent: likelihood is not finite";
  frames_ += h[0]; correct_ += h[1]; loss_ += h[2]; entropy_ += h[3]; likelyhood_ += h[4];
  frames_progress_ += h[0]; loss_progress_ += h[2]; entropy_progress_ += h[3]; likelyhood_progress_ += h[4];
}

void Xent::AfterEval(int rows) {
  dirty_ = true;
  // progressive loss reporting (nnet-loss.cc:134-155): every 1h of frames.  The frame count is
  // only known on the device, so the row count gates the (rare) fetch.
  static const int32 progress_step = 3600 * 100;
  rows_since_progress_ += rows;
  if (rows_since_progress_ > progress_step) {
    Fetch();
    if (frames_progress_ > progress_step) {
      ASLP_LOG << "ProgressLoss[last " << static_cast<int>(frames_progress_ / 100 / 3600) << "h of "
               << static_cast<int>(frames_ / 100 / 3600) << "h]: " << likelyhood_progress_ / frames_progress_ << " (Likelyhood) "
               << (loss_progress_ - entropy_progress_) / frames_progress_ << " (Xent)";
      loss_vec_.push_back((loss_progress_ - entropy_progress_) / frames_progress_);
      frames_progress_ = 0; loss_progress_ = 0.0; entropy_progress_ = 0.0; likelyhood_progress_ = 0.0;
      rows_since_progress_ = 0;
    }
  }
}

void Xent::Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const CuMatrixBase &targets, CuMatrix *diff) {
  // nnet-loss.cc:63-156
  ASLP_ASSERT(net_out.NumCols() == targets.NumCols());
  ASLP_ASSERT(net_out.NumRows() == targets.NumRows());
  ASLP_ASSERT(net_out.NumRows() == (int)frame_weights.size());
  for (BaseFloat w : frame_weights) ASLP_ASSERT(std::isfinite(w));
  HostVector hv;
  hv.data = frame_weights;
  frame_weights_ = hv;
  diff->Resize(net_out.NumRows(), net_out.NumCols(), kUndefined);
  FlushPending();   // (the accumulators take the batches in evaluation order)
  aslp_xent_eval(net_out.Data(), net_out.Dim(), targets.Data(), targets.Stride(), nullptr, frame_weights_.Data(), diff->Data(),
                 diff->Stride(), stats_.Data());
  CheckK();
  AfterEval(net_out.NumRows());
}

void Xent::Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const Posterior &post, CuMatrix *diff) {
  // nnet-loss.cc:159-172.  One-hot posteriors (the normal alignment case) skip the dense
  // [frames x pdfs] target matrix: the fused kernel takes the label directly.
  int32 num_frames = net_out.NumRows(), num_pdf = net_out.NumCols();
  ASLP_ASSERT(num_frames == (int32)post.size());
  bool one_hot = true;
  std::vector<int32> labels(num_frames);
  for (int32 t = 0; t < num_frames && one_hot; t++) {
    if (post[t].size() != 1 || post[t][0].second != 1.0f) one_hot = false;
    else {
      labels[t] = post[t][0].first;
      if (labels[t] >= num_pdf) ASLP_ERR << "Out-of-bound Posterior element with index " << labels[t] << ", higher than number of columns " << num_pdf;
      if (labels[t] < 0) one_hot = false;
    }
  }
  if (!one_hot) {
    PosteriorToMatrix(post, num_pdf, &tgt_mat_);
    Eval(frame_weights, net_out, tgt_mat_, diff);
    return;
  }
  ASLP_ASSERT(num_frames == (int)frame_weights.size());
  HostVector hv;
  hv.data = frame_weights;
  frame_weights_ = hv;
  labels_ = labels;
  EvalLabels(frame_weights_, net_out, labels_, diff);
}

// Xent with one label per frame.  On posteriors |diff| = |y - t| w <= max w, known before the launch when the caller knows its frame
// weights' maximum (fw_max > 0).  If the network asked for the diff's planes (Nnet::LossDiff -> s16_loss_diff_target) the kernel
// writes them beside the diff.
static void XentLabels(const CuMatrixBase &in, bool softmax, const int32 *labels_dev, const CuVectorBase &fw, float fw_max, CuMatrix *diff,
                       double *stats, double *rowstats_room) {
  S16DiffTarget t = s16_loss_diff_target();
  s16_loss_diff_target() = S16DiffTarget();
  aslp_planes_out po = aslp_planes_out();
  // (only where the kernel forms the posteriors itself: what a caller hands in as "net_out" need not be posteriors -- a linear output
  //  layer -- and |y - t| <= 1 is then nobody's promise; the consumer of the diff converts it with a measured maximum instead)
  PlaneSet *ps = (softmax && t.planes && t.diff == diff->Data() && fw_max > 0.0f && std::isfinite(fw_max)) ? t.planes : nullptr;
  if (ps && ps->Reserve(in.NumRows(), in.NumCols()) && ps->SetBound(fw_max)) aslp_planes_as_output(reinterpret_cast<const aslp_planes *>(ps), &po);
  else ps = nullptr;
  const int planes = rowstats_room
                         ? aslp_xent_eval_rows(in.Data(), in.Dim(), labels_dev, fw.Data(), diff->Data(), diff->Stride(), rowstats_room, softmax ? 1 : 0, ps ? &po : nullptr)
                         : aslp_xent_eval_p(in.Data(), in.Dim(), labels_dev, fw.Data(), diff->Data(), diff->Stride(), stats, softmax ? 1 : 0, ps ? &po : nullptr);
  if (planes && ps) ps->Tag(diff->Data(), diff->Stride(), t.epoch);
}

void Xent::EvalLabels(const CuVectorBase &fw, const CuMatrixBase &net_out, const CuArray<int32> &labels, CuMatrix *diff) {
  ASLP_ASSERT(fw.Dim() == net_out.NumRows() && labels.Dim() == net_out.NumRows());
  diff->Resize(net_out.NumRows(), net_out.NumCols(), kUndefined);
  FlushPending();
  aslp_xent_eval(net_out.Data(), net_out.Dim(), nullptr, 0, labels.Data(), fw.Data(), diff->Data(), diff->Stride(), stats_.Data());
  CheckK();
  AfterEval(net_out.NumRows());
}

void Xent::EvalLabels(const CuVectorBase &fw, const CuMatrixBase &net_out, const int32 *labels_dev, CuMatrix *diff, float fw_max) {
  ASLP_ASSERT(fw.Dim() == net_out.NumRows());
  diff->Resize(net_out.NumRows(), net_out.NumCols(), kUndefined);
  XentLabels(net_out, false, labels_dev, fw, fw_max, diff, stats_.Data(), PendingRowStats(net_out.NumRows()));
  CheckK();
  AfterEval(net_out.NumRows());
}
void Xent::EvalLabelsPreSoftmax(const CuVectorBase &fw, const CuMatrixBase &acts, const int32 *labels_dev, CuMatrix *diff, float fw_max) {
  ASLP_ASSERT(fw.Dim() == acts.NumRows());
  diff->Resize(acts.NumRows(), acts.NumCols(), kUndefined);
  if (!aslp_softmax_xent_supported(acts.NumCols())) ASLP_ERR << "Softmax + Xent in one pass: unsupported number of classes " << acts.NumCols();
  XentLabels(acts, true, labels_dev, fw, fw_max, diff, stats_.Data(), PendingRowStats(acts.NumRows()));
  CheckK();
  AfterEval(acts.NumRows());
}

void Xent::EvalOnLossInput(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &in, bool pre_softmax, const Posterior &post,
                           CuMatrix *diff) {
  if (!pre_softmax) { Eval(frame_weights, in, post, diff); return; }
  const int32 num_frames = in.NumRows(), num_pdf = in.NumCols();
  ASLP_ASSERT(num_frames == (int32)post.size() && num_frames == (int32)frame_weights.size());
  for (BaseFloat w : frame_weights) ASLP_ASSERT(std::isfinite(w));
  bool one_hot = true;
  std::vector<int32> labels(num_frames);
  for (int32 t = 0; t < num_frames && one_hot; t++) {
    if (post[t].size() != 1 || post[t][0].second != 1.0f || post[t][0].first < 0) one_hot = false;
    else {
      labels[t] = post[t][0].first;
      if (labels[t] >= num_pdf) ASLP_ERR << "Out-of-bound Posterior element with index " << labels[t] << ", higher than number of columns " << num_pdf;
    }
  }
  HostVector hv;
  hv.data = frame_weights;
  frame_weights_ = hv;
  diff->Resize(num_frames, num_pdf, kUndefined);
  if (one_hot) {
    labels_ = labels;
    float fw_max = 0.0f;
    for (BaseFloat w : frame_weights) fw_max = std::max(fw_max, std::fabs(w));
    XentLabels(in, true, labels_.Data(), frame_weights_, fw_max, diff, stats_.Data(), PendingRowStats(num_frames));
  } else {
    PosteriorToMatrix(post, num_pdf, &tgt_mat_);
    FlushPending();
    aslp_softmax_xent_eval(in.Data(), in.Dim(), tgt_mat_.Data(), tgt_mat_.Stride(), nullptr, frame_weights_.Data(), diff->Data(), diff->Stride(),
                           stats_.Data(), nullptr, 0);
  }
  CheckK();
  AfterEval(num_frames);
}

std::string Xent::Report() {  // nnet-loss.cc:175-199
  Fetch();
  std::ostringstream oss;
  if (0 == frames_) {
    oss << "AvgLoss: " << 0.0 << " (Xent), " << "Likelyhood: " << 0.0 << " " << "Frame: " << 0 << std::endl;
    oss << "FRAME_ACCURACY >> " << 0.0 << "% <<" << std::endl;
  } else {
    oss << "AvgLoss: " << (loss_ - entropy_) / frames_ << " (Xent), " << "Likelyhood: " << (likelyhood_) / frames_ << " "
        << "Frame: " << frames_ << std::endl;
    if (correct_ >= 0.0) oss << "FRAME_ACCURACY >> " << 100.0 * correct_ / frames_ << "% <<" << std::endl;
  }
  return oss.str();
}

void Mse::Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const CuMatrixBase &target, CuMatrix *diff) {
  // nnet-loss.cc:205-258
  ASLP_ASSERT(net_out.NumCols() == target.NumCols());
  ASLP_ASSERT(net_out.NumRows() == target.NumRows());
  ASLP_ASSERT(net_out.NumRows() == (int)frame_weights.size());
  double fsum = 0.0;
  for (BaseFloat w : frame_weights) fsum += w;
  ASLP_ASSERT(std::isfinite(fsum));
  int32 num_frames = fsum;
  ASLP_ASSERT(num_frames >= 0.0);
  HostVector hv;
  hv.data = frame_weights;
  frame_weights_ = hv;
  *diff = net_out;
  diff->AddMat(-1.0, target);
  diff->MulRowsVec(frame_weights_);
  diff_pow_2_ = *diff;
  diff_pow_2_.MulElements(diff_pow_2_);
  diff_pow_2_.MulRowsVec(frame_weights_);
  num_tgt_ = diff_pow_2_.NumCols();
  double mean_square_error = 0.5 * diff_pow_2_.Sum();
  ASLP_ASSERT(std::isfinite(mean_square_error));
  loss_ += mean_square_error;
  frames_ += num_frames;
  static const int32 progress_step = 3600 * 100;
  frames_progress_ += num_frames;
  loss_progress_ += mean_square_error;
  if (frames_progress_ > progress_step) {
    ASLP_LOG << "ProgressLoss[last " << static_cast<int>(frames_progress_ / 100 / 3600) << "h of " << static_cast<int>(frames_ / 100 / 3600)
             << "h]: " << loss_progress_ / frames_progress_ << " (Mse)";
    loss_vec_.push_back(loss_progress_ / frames_progress_);
    frames_progress_ = 0;
    loss_progress_ = 0.0;
  }
}
void Mse::Eval(const std::vector<BaseFloat> &frame_weights, const CuMatrixBase &net_out, const Posterior &post, CuMatrix *diff) {
  ASLP_ASSERT(net_out.NumRows() == (int32)post.size());
  PosteriorToMatrix(post, net_out.NumCols(), &tgt_mat_);
  Eval(frame_weights, net_out, tgt_mat_, diff);
}
std::string Mse::Report() {  // nnet-loss.cc:277-290
  BaseFloat root_mean_square = sqrt(loss_ / frames_ / num_tgt_);
  std::ostringstream oss;
  oss << "AvgLoss: " << loss_ / frames_ << " (Mse), " << "[RMS " << root_mean_square << ", frames " << frames_ << "]" << std::endl;
  return oss.str();
}

void MultiTaskLoss::InitFromString(const std::string &s) {  // nnet-loss.cc:296-339
  std::vector<std::string> v;
  SplitStringToVector(s, ",:", false, &v);
  ASLP_ASSERT((v.size() - 1) % 3 == 0);
  ASLP_ASSERT(v[0] == "multitask");
  for (size_t i = 1; i < v.size(); i += 3) {
    if (v[i] == "xent") loss_vec_.push_back(new Xent());
    else if (v[i] == "mse") loss_vec_.push_back(new Mse());
    else ASLP_ERR << "Unknown objective function code : " << v[i];
    int32 dim;
    if (!ConvertStringToInteger(v[i + 1], &dim)) ASLP_ERR << "Cannot convert 'dim' " << v[i + 1] << " to integer!";
    loss_dim_.push_back(dim);
    BaseFloat weight;
    if (!ConvertStringToReal(v[i + 2], &weight)) ASLP_ERR << "Cannot convert 'weight' " << v[i + 2] << " to integer!";
    ASLP_ASSERT(weight >= 0.0);
    loss_weights_.push_back(weight);
  }
  loss_dim_offset_.assign(loss_dim_.size() + 1, 0);
  for (size_t i = 1; i <= loss_dim_.size(); i++) loss_dim_offset_[i] = loss_dim_offset_[i - 1] + loss_dim_[i - 1];
  ASLP_ASSERT(loss_vec_.size() > 0);
}
void MultiTaskLoss::Eval(const std::vector<BaseFloat> &fw, const CuMatrixBase &net_out, const Posterior &post, CuMatrix *diff) {
  int32 num_frames = net_out.NumRows(), num_output = net_out.NumCols();  // nnet-loss.cc:341-368
  ASLP_ASSERT(num_frames == (int32)post.size());
  ASLP_ASSERT(num_output == loss_dim_offset_.back());
  PosteriorToMatrix(post, num_output, &tgt_mat_);
  diff->Resize(num_frames, num_output);
  CuMatrix diff_aux;
  for (size_t i = 0; i < loss_vec_.size(); i++) {
    loss_vec_[i]->Eval(fw, net_out.ColRange(loss_dim_offset_[i], loss_dim_[i]), tgt_mat_.ColRange(loss_dim_offset_[i], loss_dim_[i]), &diff_aux);
    diff_aux.Scale(loss_weights_[i]);
    diff->ColRange(loss_dim_offset_[i], loss_dim_[i]).CopyFromMat(diff_aux);
  }
}
std::string MultiTaskLoss::Report() {  // nnet-loss.cc:370-393
  BaseFloat overall_loss = AvgLoss();
  std::ostringstream oss;
  oss << "MultiTaskLoss, with " << loss_vec_.size() << " parallel loss functions." << std::endl;
  for (size_t i = 0; i < loss_vec_.size(); i++) oss << "Loss " << i + 1 << ", " << loss_vec_[i]->Report() << std::endl;
  // (the reference streams both vectors through nnet-utils.h:42-45: every element followed by one blank, no brackets)
  oss << "Loss (OVERALL), " << "AvgLoss: " << overall_loss << " (MultiTaskLoss), " << "weights ";
  for (BaseFloat w : loss_weights_) oss << w << " ";
  oss << ", values ";
  for (LossItf *l : loss_vec_) oss << l->AvgLoss() << " ";
  oss << std::endl;
  return oss.str();
}
BaseFloat MultiTaskLoss::AvgLoss() {  // nnet-loss.cc:395-406
  BaseFloat ans(0.0);
  for (size_t i = 0; i < loss_vec_.size(); i++) {
    BaseFloat val = loss_weights_[i] * loss_vec_[i]->AvgLoss();
    if (!std::isfinite(val)) {
      ASLP_WARN << "Loss " << i + 1 << ", has bad objective function value '" << val << "', using 0.0 instead.";
      val = 0.0;
    }
    ans += val;
  }
  return ans;
}

}  // namespace aslp
