// nnet-conv.h -- the front-end components of the CNN / cFSMN recipes (aslp_scripts/aslp_nnet/run_cnn*.sh, run_*ctc_cnn_*.sh,
// run_cfsmn*.sh) on the gfx950 kernels: LinearTransform (nnet-linear-transform.h), ConvolutionalComponent
// (nnet-convolutional-component.h), MaxPoolingComponent (nnet-max-pooling-component.h), LengthNormComponent (nnet-various.h:327-363),
// PnormComponent / MaxoutComponent (nnet-activation.h:300-377).  Reference file:line for each method is cited in place.
#pragma once
#include <cmath>

#include "kaldi-io.h"
#include "nnet-component.h"

namespace aslp {

// ---- LinearTransform (nnet-linear-transform.h:33-186): an AffineTransform without bias ---------------------------------------
class LinearTransform : public UpdatableComponent {
 public:
  LinearTransform(int32 dim_in, int32 dim_out)
      : UpdatableComponent(dim_in, dim_out), linearity_(dim_out, dim_in), linearity_corr_(dim_out, dim_in), learn_rate_coef_(1.0) {}
  Component *Copy() const { return new LinearTransform(*this); }
  ComponentType GetType() const { return kLinearTransform; }

  void InitData(std::istream &is) {  // :45-87
    float param_stddev = 0.1, learn_rate_coef = 1.0;
    std::string read_matrix_file, token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<ParamStddev>") ReadBasicType(is, false, &param_stddev);
      else if (token == "<LearnRateCoef>") ReadBasicType(is, false, &learn_rate_coef);
      else if (token == "<ReadMatrix>") ReadToken(is, false, &read_matrix_file);
      else ASLP_ERR << "Unknown token " << token << ", a typo in config?" << " (ParamStddev|ReadMatrix|LearnRateCoef)";
      is >> std::ws;
    }
    if (read_matrix_file != "") {
      bool binary;
      Input in(read_matrix_file, &binary);
      linearity_.Read(in.Stream(), binary);
      in.Close();
      ASLP_LOG << "Loaded <LinearTransform> matrix from file : " << read_matrix_file;
    } else {
      HostMatrix mat(output_dim_, input_dim_);
      for (int32 r = 0; r < output_dim_; r++)
        for (int32 c = 0; c < input_dim_; c++) mat(r, c) = param_stddev * RandGauss();
      linearity_ = mat;
    }
    learn_rate_coef_ = learn_rate_coef;
    ASLP_ASSERT(linearity_.NumRows() == output_dim_);
    ASLP_ASSERT(linearity_.NumCols() == input_dim_);
    linearity_corr_.Resize(output_dim_, input_dim_);
  }
  void ReadData(std::istream &is, bool binary) {  // :89-98
    ExpectToken(is, binary, "<LearnRateCoef>");
    ReadBasicType(is, binary, &learn_rate_coef_);
    linearity_.Read(is, binary);
    ASLP_ASSERT(linearity_.NumRows() == output_dim_);
    ASLP_ASSERT(linearity_.NumCols() == input_dim_);
    linearity_corr_.Resize(output_dim_, input_dim_);
  }
  void WriteData(std::ostream &os, bool binary) const {  // :100-104
    WriteToken(os, binary, "<LearnRateCoef>");
    WriteBasicType(os, binary, learn_rate_coef_);
    linearity_.Write(os, binary);
  }
  int32 NumParams() const { return linearity_.NumRows() * linearity_.NumCols(); }
  void GetParams(std::vector<BaseFloat> *w) const { w->clear(); AppendRowMajor(linearity_, w); }
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) {  // :114-117
    params->clear();
    params->push_back(std::make_pair(linearity_.Data(), linearity_.NumRows() * linearity_.Stride()));
  }
  double ParamSum() const { return linearity_.Sum(); }
  std::string Info() const { return std::string("\n  linearity") + MomentStatistics(linearity_); }
  std::string InfoGradient() const {
    std::ostringstream o;
    o << "\n  linearity_grad" << MomentStatistics(linearity_corr_) << ", lr-coef " << learn_rate_coef_;
    return o.str();
  }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { out->AddMatMat(1.0, in, kNoTrans, linearity_, kTrans, 0.0); }  // :127-130
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {
    in_diff->AddMatMat(1.0, out_diff, kNoTrans, linearity_, kNoTrans, 0.0);  // :132-136
  }
  void Update(const CuMatrixBase &input, const CuMatrixBase &diff) {  // :139-160
    const BaseFloat lr = opts_.learn_rate, mmt = opts_.momentum, l2 = opts_.l2_penalty, l1 = opts_.l1_penalty;
    const int32 num_frames = input.NumRows();
    const bool plain = (l2 == 0.0 && l1 == 0.0);
    // gradient incl. momentum; with no regulariser between gradient and step the step W += -lr * coef * W_corr rides in the epilogue
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    if (plain) { ep.W = linearity_.Data(); ep.ldw = linearity_.Stride(); ep.w_alpha = -lr * learn_rate_coef_; }
    linearity_corr_.AddMatMat(1.0, diff, kTrans, input, kNoTrans, mmt, &ep);
    if (!plain) {
      if (l2 != 0.0) linearity_.AddMat(-lr * l2 * num_frames, linearity_);
      if (l1 != 0.0) cu::RegularizeL1(&linearity_, &linearity_corr_, lr * l1 * num_frames, lr);
      linearity_.AddMat(-lr * learn_rate_coef_, linearity_corr_);
    }
  }
  const CuMatrixBase &GetLinearity() const { return linearity_; }
  void SetLinearity(const CuMatrixBase &l) { ASLP_ASSERT(SameDim(l, linearity_)); linearity_.CopyFromMat(l); }
  const CuMatrixBase &GetLinearityCorr() const { return linearity_corr_; }

 private:
  CuMatrix linearity_, linearity_corr_;
  BaseFloat learn_rate_coef_;
};

// ---- ConvolutionalComponent (nnet-convolutional-component.h:65-490) -------------------------------------------------------------
// 1-D convolution along the frequency axis of spliced frames.  The reference gathers P = num_patches column blocks per frame and
// runs P products per pass (:326-331, :399-407, :443-450).  Here the patches of frame n are P consecutive rows of one matrix, so
// each pass is ONE product over N * P rows:
//   forward   out_rows[(n P + p)][f]      = bias[f] + patches[(n P + p)] . filters[f]          (bias in the epilogue)
//   backward  patch_diff[(n P + p)]       = out_diff_rows[(n P + p)] x filters;  in_diff = gather-sum of patch_diff (one pass)
//   update    filters_grad = out_diff_rows^T x patches (K = N P), bias_grad = its column sums, both SGD steps in the epilogue
// out / out_diff [N x P F] are viewed as [N P x F] in place when their rows are unpadded (stride == P F: every recipe's shape),
// through a contiguous copy otherwise.
class ConvolutionalComponent : public UpdatableComponent {
 public:
  ConvolutionalComponent(int32 dim_in, int32 dim_out)
      : UpdatableComponent(dim_in, dim_out), patch_dim_(0), patch_step_(0), patch_stride_(0), learn_rate_coef_(1.0), bias_learn_rate_coef_(1.0),
        max_norm_(0.0) {}
  Component *Copy() const { return new ConvolutionalComponent(*this); }
  ComponentType GetType() const { return kConvolutionalComponent; }

  void InitData(std::istream &is) {  // :96-165
    BaseFloat bias_mean = -2.0, bias_range = 2.0, param_stddev = 0.1, norm_init_scale = 1.0;
    bool gauss_init = true;
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<NormInit>") { ReadBasicType(is, false, &norm_init_scale); gauss_init = false; }
      else if (token == "<ParamStddev>") ReadBasicType(is, false, &param_stddev);
      else if (token == "<BiasMean>") ReadBasicType(is, false, &bias_mean);
      else if (token == "<BiasRange>") ReadBasicType(is, false, &bias_range);
      else if (token == "<PatchDim>") ReadBasicType(is, false, &patch_dim_);
      else if (token == "<PatchStep>") ReadBasicType(is, false, &patch_step_);
      else if (token == "<PatchStride>") ReadBasicType(is, false, &patch_stride_);
      else if (token == "<LearnRateCoef>") ReadBasicType(is, false, &learn_rate_coef_);
      else if (token == "<BiasLearnRateCoef>") ReadBasicType(is, false, &bias_learn_rate_coef_);
      else if (token == "<MaxNorm>") ReadBasicType(is, false, &max_norm_);
      else ASLP_ERR << "Unknown token " << token << ", a typo in config?" << " (ParamStddev|BiasMean|BiasRange|PatchDim|PatchStep|PatchStride)";
      is >> std::ws;
    }
    CheckGeometry(true);
    const int32 num_filters = NumFilters(), filter_dim = FilterDim();
    if (!gauss_init) {
      filters_.Resize(num_filters, filter_dim);
      bias_.Resize(num_filters);
      float scale = norm_init_scale * sqrt(6.0 / (num_filters + filter_dim));
      InitMatParamUniform(filters_, scale);
      InitVecParamUniform(bias_, scale);
    } else {
      HostMatrix mat(num_filters, filter_dim);
      for (int32 r = 0; r < num_filters; r++)
        for (int32 c = 0; c < filter_dim; c++) mat(r, c) = param_stddev * RandGauss();
      filters_ = mat;
      HostVector vec(num_filters);
      for (int32 i = 0; i < num_filters; i++) vec.data[i] = bias_mean + (RandUniform() - 0.5) * bias_range;
      bias_ = vec;
    }
  }
  void ReadData(std::istream &is, bool binary) {  // :167-210
    ExpectToken(is, binary, "<PatchDim>"); ReadBasicType(is, binary, &patch_dim_);
    ExpectToken(is, binary, "<PatchStep>"); ReadBasicType(is, binary, &patch_step_);
    ExpectToken(is, binary, "<PatchStride>"); ReadBasicType(is, binary, &patch_stride_);
    ExpectToken(is, binary, "<LearnRateCoef>"); ReadBasicType(is, binary, &learn_rate_coef_);
    ExpectToken(is, binary, "<BiasLearnRateCoef>"); ReadBasicType(is, binary, &bias_learn_rate_coef_);
    ExpectToken(is, binary, "<MaxNorm>"); ReadBasicType(is, binary, &max_norm_);
    ExpectToken(is, binary, "<Filters>"); filters_.Read(is, binary);
    ExpectToken(is, binary, "<Bias>"); bias_.Read(is, binary);
    CheckGeometry(false);
    ASLP_ASSERT(NumFilters() == filters_.NumRows());
    ASLP_ASSERT(NumFilters() == bias_.Dim());
    ASLP_ASSERT(FilterDim() == filters_.NumCols());
  }
  void WriteData(std::ostream &os, bool binary) const {  // :212-236
    WriteToken(os, binary, "<PatchDim>"); WriteBasicType(os, binary, patch_dim_);
    WriteToken(os, binary, "<PatchStep>"); WriteBasicType(os, binary, patch_step_);
    WriteToken(os, binary, "<PatchStride>"); WriteBasicType(os, binary, patch_stride_);
    WriteToken(os, binary, "<LearnRateCoef>"); WriteBasicType(os, binary, learn_rate_coef_);
    WriteToken(os, binary, "<BiasLearnRateCoef>"); WriteBasicType(os, binary, bias_learn_rate_coef_);
    WriteToken(os, binary, "<MaxNorm>"); WriteBasicType(os, binary, max_norm_);
    WriteToken(os, binary, "<Filters>"); filters_.Write(os, binary);
    WriteToken(os, binary, "<Bias>"); bias_.Write(os, binary);
  }
  int32 NumParams() const { return filters_.NumRows() * filters_.NumCols() + bias_.Dim(); }
  void GetParams(std::vector<BaseFloat> *w) const { w->clear(); AppendRowMajor(filters_, w); AppendVector(bias_, w); }
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) {  // :249-253
    params->clear();
    params->push_back(std::make_pair(filters_.Data(), filters_.NumRows() * filters_.Stride()));
    params->push_back(std::make_pair(bias_.Data(), bias_.Dim()));
  }
  std::string Info() const { return std::string("\n  filters") + MomentStatistics(filters_) + "\n  bias" + MomentStatistics(bias_); }
  std::string InfoGradient() const {
    std::ostringstream o;
    o << "\n  filters_grad" << MomentStatistics(filters_grad_) << ", lr-coef " << learn_rate_coef_ << ", max-norm " << max_norm_ << "\n  bias_grad"
      << MomentStatistics(bias_grad_) << ", lr-coef " << bias_learn_rate_coef_;
    return o.str();
  }

  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {  // :268-332
    const int32 P = NumPatches(), F = filters_.NumRows(), K = filters_.NumCols(), N = in.NumRows();
    if (patches_.NumRows() != N * P || patches_.NumCols() != K) patches_.Resize(N * P, K, kUndefined);
    aslp_conv_gather_patches(patches_.Data(), patches_.Stride(), in.Data(), in.Dim(), P, NumSplice(), patch_dim_, patch_step_, patch_stride_);
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    ep.bias = bias_.Data();   // tgt.AddVecToRows(1.0, bias_, 0.0) then AddMatMat(..., 1.0) (:327-330)
    if (out->Stride() == P * F) {
      CuSubMatrix rows(out->Data(), N * P, F, F);
      rows.AddMatMat(1.0, patches_, kNoTrans, filters_, kTrans, 0.0, &ep);
    } else {
      CuSubMatrix rows = Scratch(&rows_tmp_, N * P, F);
      rows.AddMatMat(1.0, patches_, kNoTrans, filters_, kTrans, 0.0, &ep);
      out->CopyFromMat(CuSubMatrix(rows.Data(), N, P * F, P * F));
    }
  }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {  // :390-422
    const int32 P = NumPatches(), F = filters_.NumRows(), K = filters_.NumCols(), N = out_diff.NumRows();
    if (patch_diffs_.NumRows() != N * P || patch_diffs_.NumCols() != K) patch_diffs_.Resize(N * P, K, kUndefined);
    CuSubMatrix od = DiffRows(out_diff, N, P, F);
    patch_diffs_.AddMatMat(1.0, od, kNoTrans, filters_, kNoTrans, 0.0);
    aslp_conv_in_diff(in_diff->Data(), in_diff->Dim(), patch_diffs_.Data(), patch_diffs_.Stride(), P, NumSplice(), patch_dim_, patch_step_, patch_stride_);
  }
  void Update(const CuMatrixBase &input, const CuMatrixBase &diff) {  // :425-470
    const int32 P = NumPatches(), F = filters_.NumRows(), K = filters_.NumCols(), N = diff.NumRows();
    const BaseFloat lr = opts_.learn_rate;
    ASLP_ASSERT(patches_.NumRows() == N * P);   // the patches of the Propagate this Update belongs to (:447)
    if (filters_grad_.NumRows() != F || filters_grad_.NumCols() != K) filters_grad_.Resize(F, K, kUndefined);
    if (bias_grad_.Dim() != F) bias_grad_.Resize(F, kUndefined);
    CuSubMatrix od = DiffRows(diff, N, P, F);
    // filters_grad_ = sum_p diff_p^T patch_p, bias_grad_ = sum_p colsum(diff_p) (gradient reset every call, no momentum: :437-438),
    // filters_ += -lr coef filters_grad_, bias_ += -lr bias_coef bias_grad_ (:456-457): one product with both steps in its epilogue
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    ep.W = filters_.Data(); ep.ldw = filters_.Stride(); ep.w_alpha = -lr * learn_rate_coef_;
    ep.colsum = bias_grad_.Data(); ep.colsum_beta = 0.0f; ep.colsum_w = bias_.Data(); ep.colsum_w_alpha = -lr * bias_learn_rate_coef_;
    filters_grad_.AddMatMat(1.0, od, kTrans, patches_, kNoTrans, 0.0, &ep);
    if (max_norm_ > 0.0) aslp_max_norm_rows(filters_.Data(), filters_.Dim(), max_norm_);  // :460-470
  }
  const CuMatrixBase &GetFilters() const { return filters_; }
  const CuVectorBase &GetBias() const { return bias_; }
  const CuMatrixBase &GetFiltersGrad() const { return filters_grad_; }
  const CuVectorBase &GetBiasGrad() const { return bias_grad_; }

 private:
  int32 NumSplice() const { return input_dim_ / patch_stride_; }
  int32 NumPatches() const { return 1 + (patch_stride_ - patch_dim_) / patch_step_; }
  int32 FilterDim() const { return NumSplice() * patch_dim_; }
  int32 NumFilters() const { return output_dim_ / NumPatches(); }
  void CheckGeometry(bool log) {  // :121-137, :192-205
    ASLP_ASSERT(patch_dim_ > 0 && patch_step_ > 0 && patch_stride_ > 0);
    ASLP_ASSERT(input_dim_ % patch_stride_ == 0);
    ASLP_ASSERT((patch_stride_ - patch_dim_) % patch_step_ == 0);
    ASLP_ASSERT(output_dim_ % NumPatches() == 0);
    if (log) {
      ASLP_LOG << "num_splice " << NumSplice();
      ASLP_LOG << "num_patches " << NumPatches();
      ASLP_LOG << "filter_dim " << FilterDim();
      ASLP_LOG << "num_filters " << NumFilters();
    }
  }
  // an unpadded [rows x cols] scratch matrix (leading dimension == cols)
  static CuSubMatrix Scratch(CuMatrix *buf, int32 rows, int32 cols) {
    if (buf->NumRows() != 1 || buf->NumCols() != rows * cols) buf->Resize(1, rows * cols, kUndefined);
    return CuSubMatrix(buf->Data(), rows, cols, cols);
  }
  // m [N x P F] as [N P x F]: in place when the rows of m are unpadded, else through a contiguous copy
  CuSubMatrix DiffRows(const CuMatrixBase &m, int32 N, int32 P, int32 F) {
    if (m.Stride() == P * F) return CuSubMatrix(const_cast<BaseFloat *>(m.Data()), N * P, F, F);
    CuSubMatrix flat = Scratch(&rows_tmp_, N, P * F);
    flat.CopyFromMat(m);
    return CuSubMatrix(flat.Data(), N * P, F, F);
  }

  int32 patch_dim_, patch_step_, patch_stride_;
  CuMatrix filters_;
  CuVector bias_;
  CuMatrix filters_grad_;
  CuVector bias_grad_;
  BaseFloat learn_rate_coef_, bias_learn_rate_coef_, max_norm_;
  CuMatrix patches_;       // the reference's vectorized_feature_patches_, one patch per ROW: [N P x filter_dim]
  CuMatrix patch_diffs_;   // feature_patch_diffs_, same layout
  CuMatrix rows_tmp_;
};

// ---- MaxPoolingComponent (nnet-max-pooling-component.h:39-169) ----------------------------------------------------------------
class MaxPoolingComponent : public Component {
 public:
  MaxPoolingComponent(int32 dim_in, int32 dim_out) : Component(dim_in, dim_out), pool_size_(0), pool_step_(0), pool_stride_(0) {}
  Component *Copy() const { return new MaxPoolingComponent(*this); }
  ComponentType GetType() const { return kMaxPoolingComponent; }
  void InitData(std::istream &is) {  // :54-67
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<PoolSize>") ReadBasicType(is, false, &pool_size_);
      else if (token == "<PoolStep>") ReadBasicType(is, false, &pool_step_);
      else if (token == "<PoolStride>") ReadBasicType(is, false, &pool_stride_);
      else ASLP_ERR << "Unknown token " << token << ", a typo in config?" << " (PoolSize|PoolStep|PoolStride)";
      is >> std::ws;
    }
    ASLP_ASSERT(pool_size_ != 0 && pool_step_ != 0 && pool_stride_ != 0);
    CheckGeometry();   // (the reference checks these only when it reads a model, :82-90; a proto that violates them cannot train either)
  }
  void ReadData(std::istream &is, bool binary) {  // :69-91
    ExpectToken(is, binary, "<PoolSize>"); ReadBasicType(is, binary, &pool_size_);
    ExpectToken(is, binary, "<PoolStep>"); ReadBasicType(is, binary, &pool_step_);
    ExpectToken(is, binary, "<PoolStride>"); ReadBasicType(is, binary, &pool_stride_);
    CheckGeometry();
  }
  void WriteData(std::ostream &os, bool binary) const {  // :93-99
    WriteToken(os, binary, "<PoolSize>"); WriteBasicType(os, binary, pool_size_);
    WriteToken(os, binary, "<PoolStep>"); WriteBasicType(os, binary, pool_step_);
    WriteToken(os, binary, "<PoolStride>"); WriteBasicType(os, binary, pool_stride_);
  }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {  // :101-116
    aslp_max_pool_forward(out->Data(), out->Stride(), in.Data(), in.Dim(), pool_size_, pool_step_, pool_stride_);
  }
  void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &out, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {  // :118-162
    aslp_max_pool_backward(in_diff->Data(), in_diff->Stride(), in.Data(), in.Dim(), out.Data(), out.Stride(), out_diff.Data(), out_diff.Stride(), pool_size_,
                           pool_step_, pool_stride_);
  }

 private:
  void CheckGeometry() const {
    ASLP_ASSERT(pool_size_ > 0 && pool_step_ > 0 && pool_stride_ > 0);
    ASLP_ASSERT(input_dim_ % pool_stride_ == 0);
    const int32 num_patches = input_dim_ / pool_stride_;
    ASLP_ASSERT((num_patches - pool_size_) % pool_step_ == 0);
    const int32 num_pools = 1 + (num_patches - pool_size_) / pool_step_;
    ASLP_ASSERT(output_dim_ == num_pools * pool_stride_);
  }
  int32 pool_size_, pool_step_, pool_stride_;
};

// ---- LengthNormComponent (nnet-various.h:327-363) -------------------------------------------------------------------------------
class LengthNormComponent : public Component {
 public:
  LengthNormComponent(int32 dim_in, int32 dim_out) : Component(dim_in, dim_out) {}
  Component *Copy() const { return new LengthNormComponent(*this); }
  ComponentType GetType() const { return kLengthNormComponent; }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {  // :338-352: one pass (sum of squares per row, scale, write)
    if (row_scales_.Dim() != in.NumRows()) row_scales_.Resize(in.NumRows(), kUndefined);
    aslp_length_norm_forward(out->Data(), out->Stride(), in.Data(), in.Dim(), row_scales_.Data());
  }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {  // :354-358
    in_diff->CopyFromMat(out_diff);
    in_diff->MulRowsVec(row_scales_);  // "diff_by_x(s * x) = s"
  }

 private:
  CuVector row_scales_;
};

// ---- PnormComponent / MaxoutComponent (nnet-activation.h:300-377) -------------------------------------------------------------------
class PnormComponent : public Component {
 public:
  PnormComponent(int32 dim_in, int32 dim_out) : Component(dim_in, dim_out), p_(2.0) {}
  Component *Copy() const { return new PnormComponent(*this); }
  ComponentType GetType() const { return kPnormComponent; }
  void InitData(std::istream &is) {  // :315-327
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<P>") ReadBasicType(is, false, &p_);
      else ASLP_ERR << "Unknown token " << token << ", a typo in config?" << " (P)";
      is >> std::ws;
    }
    ASLP_ASSERT(p_ != 0);
    ASLP_ASSERT(input_dim_ % output_dim_ == 0);
  }
  void ReadData(std::istream &is, bool binary) { ExpectToken(is, binary, "<P>"); ReadBasicType(is, binary, &p_); }   // :329-333
  void WriteData(std::ostream &os, bool binary) const { WriteToken(os, binary, "<P>"); WriteBasicType(os, binary, p_); }  // :335-339
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {  // :341-343 GroupPnorm
    ASLP_ASSERT(in.NumCols() % out->NumCols() == 0);
    cudaF_group_pnorm(aslp_dim3(), aslp_dim3(), out->Data(), in.Data(), out->Dim(), in.Stride(), in.NumCols() / out->NumCols(), p_);
  }
  void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &out, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {  // :345-349
    aslp_group_pnorm_backward(in_diff->Data(), in_diff->Stride(), in.Data(), in.Dim(), out.Data(), out.Stride(), out_diff.Data(), out_diff.Stride(),
                              in.NumCols() / out.NumCols(), p_);
  }

 private:
  BaseFloat p_;
};

class MaxoutComponent : public Component {   // :355-375 (no marker maps to it in the reference: "<Maxout>" reads as a Pnorm, nnet-component.cc:79-80)
 public:
  MaxoutComponent(int32 dim_in, int32 dim_out) : Component(dim_in, dim_out) { ASLP_ASSERT(dim_in % dim_out == 0); }
  Component *Copy() const { return new MaxoutComponent(*this); }
  ComponentType GetType() const { return kMaxoutComponent; }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {
    cudaF_group_max(aslp_dim3(), aslp_dim3(), out->Data(), in.Data(), out->Dim(), in.Stride(), in.NumCols() / out->NumCols());
  }
  void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &out, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {
    aslp_group_max_backward(in_diff->Data(), in_diff->Stride(), in.Data(), in.Dim(), out.Data(), out.Stride(), out_diff.Data(), out_diff.Stride(),
                            in.NumCols() / out.NumCols());
  }
};

}  // namespace aslp
