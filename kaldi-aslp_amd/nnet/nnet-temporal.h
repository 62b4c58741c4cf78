// nnet-temporal.h -- RowConvolution (nnet-row-convolution.{h,cc}) and CompactFsmn
// (nnet-cfsmn-component.h) on the direct depthwise kernels of csrc/temporal.hip.
// Same config tokens, file formats, parameter order and update rules as the reference; the
// reorder / product scratch matrices of the reference (in_buf_, conv_buf_, aux_mat_, ...) do not exist.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "nnet-component.h"

namespace aslp {

class RowConvolution : public UpdatableComponent {
 public:
  RowConvolution(int32 di, int32 dout) : UpdatableComponent(di, dout), future_ctx_(0) {
    if (input_dim_ != output_dim_) ASLP_ERR << "RowConvolution layer input dim and output dimmust be equal";
  }
  Component *Copy() const { return new RowConvolution(*this); }
  ComponentType GetType() const { return kRowConvolution; }
  bool GradientInBackprop() const { return true; }
  void FoldNextUpdateIntoBackprop() { fold_update_ = true; }   // the tap gradients' finishing launch then also takes the SGD step
  void SetSeqLengths(const std::vector<int32> &sequence_lengths) {  // row-convolution.h:39-41
    sequence_lengths_ = sequence_lengths;
    seq_len_dev_.CopyFromVec(sequence_lengths);
  }
  void InitData(std::istream &is) {  // row-convolution.cc:15-44
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<FutureContext>") ReadBasicType(is, false, &future_ctx_);
      else ASLP_ERR << "Unknown token " << token << ", a typo in config?"
                    << " (FutureContext)";
      is >> std::ws;
    }
    ASLP_ASSERT(future_ctx_ > 0);
    HostMatrix mat(input_dim_, future_ctx_ + 1);
    for (auto &x : mat.data) x = 1.0 * RandGauss();
    SetWeights(mat);
  }
  void ReadData(std::istream &is, bool binary) {  // :46-56
    ExpectToken(is, binary, "<FutureContext>");
    ReadBasicType(is, binary, &future_ctx_);
    HostMatrix mat;
    mat.Read(is, binary);
    ASLP_ASSERT(mat.rows == input_dim_ && mat.cols == future_ctx_ + 1);
    SetWeights(mat);
  }
  void WriteData(std::ostream &os, bool binary) const {  // :58-62
    WriteToken(os, binary, "<FutureContext>");
    WriteBasicType(os, binary, future_ctx_);
    HostMatrix mat(input_dim_, future_ctx_ + 1);
    w_.CopyToHost(mat.data.data());
    mat.Write(os, binary);
  }
  int32 NumParams() const { return w_.Dim(); }
  void GetParams(std::vector<BaseFloat> *w) const { w->clear(); AppendVector(w_, w); }
  // the reference hands NumRows * NumCols floats to the all-reduce (:74-77); w_ is stored dense
  // here so that count covers exactly the weights
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) {
    params->clear();
    params->push_back(std::make_pair(w_.Data(), w_.Dim()));
  }
  std::string Info() const { return std::string("  ") + "\n  w_ " + MomentStatistics(w_); }
  std::string InfoGradient() const {
    return std::string("  ") + "\n w_diff_ " + MomentStatistics(w_diff_) + "\n w_corr_ " + MomentStatistics(w_corr_);
  }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {  // :105-135
    const int32 S = sequence_lengths_.size();
    ASLP_ASSERT(S > 0 && in.NumRows() % S == 0);
    const int32 T = in.NumRows() / S;
    aslp_rowconv_forward(out->Data(), out->Stride(), in.Data(), in.Stride(), w_.Data(), input_dim_, future_ctx_, T, S, seq_len_dev_.Data());
  }
  void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {  // :137-176
    const int32 S = sequence_lengths_.size();
    ASLP_ASSERT(S > 0 && in.NumRows() % S == 0);
    const int32 T = in.NumRows() / S;
    const bool fold = fold_update_;
    fold_update_ = false;
    update_done_ = fold;
    aslp_rowconv_backward_fused(in_diff->Data(), in_diff->Stride(), w_diff_.Data(), in.Data(), in.Stride(), out_diff.Data(), out_diff.Stride(), w_.Data(),
                                input_dim_, future_ctx_, T, S, seq_len_dev_.Data(), w_corr_.Data(), opts_.momentum, opts_.learn_rate, fold ? 1 : 0);
    // a launch that did not happen (its scratch could not be had) has applied no step and written no in_diff: the step must not count as
    // done, and the caller hears about it here rather than at some later kernel's check
    try { CheckKernelError(); } catch (...) { update_done_ = false; throw; }
  }
  void Update(const CuMatrixBase &, const CuMatrixBase &) {  // :178-186
    if (update_done_) { update_done_ = false; return; }   // (rode in BackpropagateFnc's finishing launch)
    w_corr_.AddVec(1.0, w_diff_, opts_.momentum);
    w_.AddVec(-opts_.learn_rate, w_corr_, 1.0);
  }

 private:
  void SetWeights(const HostMatrix &mat) {
    w_.Resize(mat.rows * mat.cols, kUndefined);
    w_.CopyFromHost(mat.data.data(), mat.rows * mat.cols);
    w_diff_.Resize(w_.Dim(), kSetZero);
    w_corr_.Resize(w_.Dim(), kSetZero);
  }
  std::vector<int32> sequence_lengths_;
  CuArray<int32> seq_len_dev_;
  int32 future_ctx_;
  bool fold_update_ = false, update_done_ = false;
  CuVector w_, w_diff_, w_corr_;  // [D x (K+1)] row-major, unpadded
};

class CompactFsmn : public UpdatableComponent {
 public:
  CompactFsmn(int32 di, int32 dout)
      : UpdatableComponent(di, dout), max_frames_(3000), learn_rate_coef_(1.0), past_context_(0), future_context_(0), clip_gradient_(0.0) {}
  Component *Copy() const { return new CompactFsmn(*this); }
  ComponentType GetType() const { return kCompactFsmn; }
  bool GradientInBackprop() const { return true; }
  void FoldNextUpdateIntoBackprop() { fold_update_ = true; }   // the backward launch then also takes the SGD step
  void InitData(std::istream &is) {  // cfsmn.h:53-89
    int past_context = 30, future_context = 30;
    float learn_rate_coef = 1.0, vec_coef_mean = 0.0, vec_coef_range = 1.0;
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<PastContext>") ReadBasicType(is, false, &past_context);
      else if (token == "<FutureContext>") ReadBasicType(is, false, &future_context);
      else if (token == "<LearnRateCoef>") ReadBasicType(is, false, &learn_rate_coef);
      else if (token == "<VecCoefMean>") ReadBasicType(is, false, &vec_coef_mean);
      else if (token == "<VecCoefRange>") ReadBasicType(is, false, &vec_coef_range);
      else if (token == "<ClipGradient>") ReadBasicType(is, false, &clip_gradient_);
      else ASLP_ERR << "Unknown token " << token << ", a type in config?"
                    << " (PastContext|FutureContext|VecCoefMean|VecCoefRange|LearnRateCoef)";
      is >> std::ws;
    }
    const int32 num_row = past_context + future_context + 1, num_col = input_dim_;
    vec_coef_.Resize(num_row, num_col, kUndefined);
    InitMatParamUniform(vec_coef_, 0.5 * sqrt(6.0 / (num_col + num_row)));
    vec_coef_corr_.Resize(num_row, num_col);
    past_context_ = past_context;
    future_context_ = future_context;
    learn_rate_coef_ = learn_rate_coef;
  }
  void ReadData(std::istream &is, bool binary) {  // :91-108
    ExpectToken(is, binary, "<PastContext>"); ReadBasicType(is, binary, &past_context_);
    ExpectToken(is, binary, "<FutureContext>"); ReadBasicType(is, binary, &future_context_);
    ExpectToken(is, binary, "<LearnRateCoef>"); ReadBasicType(is, binary, &learn_rate_coef_);
    vec_coef_.Read(is, binary);
    vec_coef_corr_.Resize(vec_coef_.NumRows(), vec_coef_.NumCols());
    ASLP_ASSERT(vec_coef_.NumCols() == input_dim_);
    ASLP_ASSERT(vec_coef_.NumRows() == past_context_ + future_context_ + 1);
  }
  void WriteData(std::ostream &os, bool binary) const {  // :110-121
    WriteToken(os, binary, "<PastContext>"); WriteBasicType(os, binary, past_context_);
    WriteToken(os, binary, "<FutureContext>"); WriteBasicType(os, binary, future_context_);
    WriteToken(os, binary, "<LearnRateCoef>"); WriteBasicType(os, binary, learn_rate_coef_);
    vec_coef_.Write(os, binary);
  }
  int32 NumParams() const { return vec_coef_.NumRows() * vec_coef_.NumCols(); }
  void SetMaxSeqLength(int32 max_len) { max_frames_ = max_len; }
  void GetParams(std::vector<BaseFloat> *w) const { w->clear(); AppendRowMajor(vec_coef_, w); }
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) {  // :146-149
    params->clear();
    params->push_back(std::make_pair(vec_coef_.Data(), vec_coef_.NumRows() * vec_coef_.Stride()));
  }
  std::string Info() const { return std::string("\n vector_coefficient") + MomentStatistics(vec_coef_); }
  std::string InfoGradient() const {
    std::ostringstream o;
    o << "\n vector_coefficient_grad" << MomentStatistics(vec_coef_corr_) << ", learn-rate-coef" << learn_rate_coef_;
    return o.str();
  }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {  // :170-202
    const int32 T = in.NumRows();
    ASLP_ASSERT(T <= max_frames_);
    ASLP_ASSERT(in.NumCols() == vec_coef_.NumCols());
    aslp_fsmn_filter(out->Data(), out->Stride(), in.Data(), in.Stride(), vec_coef_.Data(), vec_coef_.Stride(), input_dim_, past_context_,
                     future_context_, T, 0);
  }
  void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {  // :204-256
    const int32 T = in.NumRows();
    ASLP_ASSERT(T <= max_frames_);
    ASLP_ASSERT(in.NumCols() == vec_coef_.NumCols());
    const bool fold = fold_update_;
    fold_update_ = false;
    update_done_ = fold;
    aslp_fsmn_backward(in_diff->Data(), in_diff->Stride(), vec_coef_corr_.Data(), vec_coef_corr_.Stride(), vec_coef_.Data(), vec_coef_.Stride(), in.Data(),
                       in.Stride(), out_diff.Data(), out_diff.Stride(), input_dim_, past_context_, future_context_, T, clip_gradient_,
                       fold ? opts_.learn_rate * learn_rate_coef_ : 0.0f);
    try { CheckKernelError(); } catch (...) { update_done_ = false; throw; }   // (as RowConvolution: a failed launch has applied no step)
  }
  void Update(const CuMatrixBase &, const CuMatrixBase &) {  // :258-262
    if (update_done_) { update_done_ = false; return; }   // (rode in BackpropagateFnc's launch)
    vec_coef_.AddMat(-opts_.learn_rate * learn_rate_coef_, vec_coef_corr_);
  }

 private:
  CuMatrix vec_coef_, vec_coef_corr_;
  int32 max_frames_;
  BaseFloat learn_rate_coef_;
  int32 past_context_, future_context_;
  BaseFloat clip_gradient_;
  bool fold_update_ = false, update_done_ = false;
};

}  // namespace aslp
