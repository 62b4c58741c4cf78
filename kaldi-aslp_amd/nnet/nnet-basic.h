// nnet-basic.h -- feed-forward components of the hot path on the gfx950 kernels:
// InputLayer / OutputLayer / ScaleLayer (nnet-io.h), AffineTransform (nnet-affine-transform.h),
// Sigmoid / Tanh / ReLU / Softmax / BlockSoftmax (nnet-activation.h), Splice / Copy / AddShift /
// Rescale (nnet-various.h), BatchNormalization (nnet-batch-normalization.h).
// Reference file:line for each method is cited in place.
#pragma once
#include <algorithm>
#include <cmath>

#include <memory>

#include "nnet-component.h"
#include "split16.h"

namespace aslp {

// fp16 planes of one operand of a component's products (csrc/split16.h), made on first use; a copied component starts without any
struct PlaneHolder {
  PlaneHolder() = default;
  PlaneHolder(const PlaneHolder &) {}
  PlaneHolder &operator=(const PlaneHolder &) { return *this; }
  PlaneSet &get() { if (!p) p.reset(new PlaneSet()); return *p; }
  // the planes of m for a product about to be issued: reused when the executor's epoch says m is unchanged since they were made, else
  // converted now (maximum pass + conversion pass).  NULL: the product converts for itself / runs on the fp32 instruction.
  const PlaneSet *Of(const CuMatrixBase &m, long epoch) {
    PlaneSet &ps = get();
    if (epoch != 0 && ps.ValidFor(m.Data(), m.NumRows(), m.NumCols(), m.Stride(), epoch)) return &ps;
    const int np = ps.PartsFor(m.Data(), epoch);   // maxima left by the kernel that wrote m: no maximum pass
    const bool ok = np > 0 ? ps.ConvertWithParts(m.Data(), m.NumRows(), m.NumCols(), m.Stride(), np)
                           : ps.ConvertFrom(m.Data(), m.NumRows(), m.NumCols(), m.Stride());
    if (!ok) return nullptr;
    ps.Tag(m.Data(), m.Stride(), epoch);
    return &ps;
  }
  std::unique_ptr<PlaneSet> p;
};

// ---- graph endpoints (nnet-io.h:19-102) -------------------------------------------------------
class InputLayer : public Component {
 public:
  InputLayer(int32 di, int32 dout) : Component(di, dout) { ASLP_ASSERT(di == dout); }
  Component *Copy() const { return new InputLayer(*this); }
  ComponentType GetType() const { return kInputLayer; }
  bool BackpropIsCopy() const { return true; }
  // Executor peephole (one-shot): the AffineTransform that alone reads the next Propagate's output wants its fp16 planes (csrc/split16.h):
  // the copy then makes them in the same launch (aslp_copy_mat_planes) instead of a maximum pass and a conversion pass behind it
  void ProduceOutputPlanes(PlaneHolder *h) { out_planes_ = h; }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {
    PlaneHolder *oh = out_planes_;
    out_planes_ = nullptr;
    if (oh && oh->get().Reserve(in.NumRows(), in.NumCols())) {
      aslp_planes_out po = aslp_planes_out();
      aslp_planes_as_output(reinterpret_cast<const aslp_planes *>(&oh->get()), &po);
      if (aslp_copy_mat_planes(out->Data(), out->Dim(), in.Data(), in.Stride(), &po)) {
        oh->get().ForgetHostBound();
        oh->get().Tag(out->Data(), out->Stride(), s16_epochs().fwd);
        return;
      }
    }
    out->CopyFromMat(in);
  }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) { id->CopyFromMat(od); }

 private:
  PlaneHolder *out_planes_ = nullptr;
};
class OutputLayer : public Component {
 public:
  OutputLayer(int32 di, int32 dout) : Component(di, dout) { ASLP_ASSERT(di == dout); }
  Component *Copy() const { return new OutputLayer(*this); }
  ComponentType GetType() const { return kOutputLayer; }
  bool PropagateIsCopy() const { return true; }
  bool BackpropIsCopy() const { return true; }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { out->CopyFromMat(in); }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) { id->CopyFromMat(od); }
};
class ScaleLayer : public Component {
 public:
  ScaleLayer(int32 di, int32 dout) : Component(di, dout), scale_(1.0) { ASLP_ASSERT(di == dout); }
  Component *Copy() const { return new ScaleLayer(*this); }
  ComponentType GetType() const { return kScaleLayer; }
  void InitData(std::istream &is) { ExpectToken(is, false, "<Scale>"); ReadBasicType(is, false, &scale_); }
  void ReadData(std::istream &is, bool binary) { ExpectToken(is, binary, "<Scale>"); ReadBasicType(is, binary, &scale_); }
  void WriteData(std::ostream &os, bool binary) const { WriteToken(os, binary, "<Scale>"); WriteBasicType(os, binary, scale_); }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { out->CopyFromMat(in); out->Scale(scale_); }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) { id->CopyFromMat(od); id->Scale(scale_); }
  BaseFloat Scale() const { return scale_; }
 private:
  BaseFloat scale_;
};

// ---- AffineTransform (nnet-affine-transform.h:34-286) --------------------------------------------
class AffineTransform : public UpdatableComponent {
 public:
  AffineTransform(int32 dim_in, int32 dim_out)
      : UpdatableComponent(dim_in, dim_out), linearity_(dim_out, dim_in), bias_(dim_out), linearity_corr_(dim_out, dim_in),
        bias_corr_(dim_out), learn_rate_coef_(1.0), bias_learn_rate_coef_(1.0), max_norm_(0.0) {}
  Component *Copy() const {
    AffineTransform *c = new AffineTransform(*this);
    c->WeightsWritten();   // (planes are not copied)
    c->stats_request_ = nullptr;
    c->in_diff_maxima_ = nullptr;
    return c;
  }
  ComponentType GetType() const { return kAffineTransform; }

  void InitData(std::istream &is) {  // :61-127
    float bias_mean = -2.0, bias_range = 2.0, param_stddev = 0.1;
    float learn_rate_coef = 1.0, bias_learn_rate_coef = 1.0, max_norm = 0.0, norm_init_scale = 1.0;
    bool gauss_init = true;
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<NormInit>") { ReadBasicType(is, false, &norm_init_scale); gauss_init = false; }
      else if (token == "<ParamStddev>") ReadBasicType(is, false, &param_stddev);
      else if (token == "<BiasMean>") ReadBasicType(is, false, &bias_mean);
      else if (token == "<BiasRange>") ReadBasicType(is, false, &bias_range);
      else if (token == "<LearnRateCoef>") ReadBasicType(is, false, &learn_rate_coef);
      else if (token == "<BiasLearnRateCoef>") ReadBasicType(is, false, &bias_learn_rate_coef);
      else if (token == "<MaxNorm>") ReadBasicType(is, false, &max_norm);
      else ASLP_ERR << "Unknown token " << token << ", a typo in config?"
                    << " (ParamStddev|BiasMean|BiasRange|LearnRateCoef|BiasLearnRateCoef)";
      is >> std::ws;
    }
    if (!gauss_init) {  // Glorot-Bengio
      float scale = norm_init_scale * sqrt(6.0 / (linearity_.NumRows() + linearity_.NumCols()));
      InitMatParamUniform(linearity_, scale);
      InitVecParamUniform(bias_, scale);
    } else {
      HostMatrix mat(output_dim_, input_dim_);
      for (int32 r = 0; r < output_dim_; r++)
        for (int32 c = 0; c < input_dim_; c++) mat(r, c) = param_stddev * RandGauss();
      linearity_ = mat;
      HostVector vec(output_dim_);
      for (int32 i = 0; i < output_dim_; i++) vec.data[i] = bias_mean + (RandUniform() - 0.5) * bias_range;
      bias_ = vec;
    }
    learn_rate_coef_ = learn_rate_coef;
    bias_learn_rate_coef_ = bias_learn_rate_coef;
    max_norm_ = max_norm;
    WeightsWritten();
  }
  void ReadData(std::istream &is, bool binary) {  // :129-155
    if ('<' == Peek(is, binary)) {
      ExpectToken(is, binary, "<LearnRateCoef>");
      ReadBasicType(is, binary, &learn_rate_coef_);
      ExpectToken(is, binary, "<BiasLearnRateCoef>");
      ReadBasicType(is, binary, &bias_learn_rate_coef_);
    }
    if ('<' == Peek(is, binary)) { ExpectToken(is, binary, "<MaxNorm>"); ReadBasicType(is, binary, &max_norm_); }
    if ('<' == Peek(is, binary)) { float tmp; ExpectToken(is, binary, "<ClipGradient>"); ReadBasicType(is, binary, &tmp); }
    linearity_.Read(is, binary);
    bias_.Read(is, binary);
    ASLP_ASSERT(linearity_.NumRows() == output_dim_);
    ASLP_ASSERT(linearity_.NumCols() == input_dim_);
    ASLP_ASSERT(bias_.Dim() == output_dim_);
    linearity_corr_.Resize(output_dim_, input_dim_);
    bias_corr_.Resize(output_dim_);
    WeightsWritten();
  }
  void WriteData(std::ostream &os, bool binary) const {  // :157-167
    WriteToken(os, binary, "<LearnRateCoef>"); WriteBasicType(os, binary, learn_rate_coef_);
    WriteToken(os, binary, "<BiasLearnRateCoef>"); WriteBasicType(os, binary, bias_learn_rate_coef_);
    WriteToken(os, binary, "<MaxNorm>"); WriteBasicType(os, binary, max_norm_);
    linearity_.Write(os, binary);
    bias_.Write(os, binary);
  }
  int32 NumParams() const { return linearity_.NumRows() * linearity_.NumCols() + bias_.Dim(); }
  void GetParams(std::vector<BaseFloat> *w) const { w->clear(); AppendRowMajor(linearity_, w); AppendVector(bias_, w); }
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) {  // :166-170
    params->clear();
    params->push_back(std::make_pair(linearity_.Data(), linearity_.NumRows() * linearity_.Stride()));
    params->push_back(std::make_pair(bias_.Data(), bias_.Dim()));
  }
  std::string Info() const { return std::string("\n  linearity") + MomentStatistics(linearity_) + "\n  bias" + MomentStatistics(bias_); }
  std::string InfoGradient() const {
    std::ostringstream o;
    o << "\n  linearity_grad" << MomentStatistics(linearity_corr_) << ", lr-coef " << learn_rate_coef_ << ", max-norm " << max_norm_
      << "\n  bias_grad" << MomentStatistics(bias_corr_) << ", lr-coef " << bias_learn_rate_coef_;
    return o.str();
  }

  double ParamSum() const { return linearity_.Sum() + (double)bias_.Sum(); }   // on the device
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {
    // :186-191  out = bias (beta 0); out += in * W^T.  One GEMM with the bias in the epilogue.
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    ep.bias = bias_.Data();
    if (stats_request_) {  // a BatchNormalization is the only consumer: the epilogue also leaves its column statistics (one-shot)
      const int groups = (in.NumRows() + 31) / 32, ld = (output_dim_ + 3) & ~3;
      if (stats_request_->Dim() != 3 * groups * ld) stats_request_->Resize(3 * groups * ld, kUndefined);
      ep.colstats = stats_request_->Data();
      ep.colstats_ld = ld;
      stats_request_ = nullptr;
    }
    const PlaneSet *pa = nullptr, *pb = nullptr;
    ForwardPlanes(in, &pa, &pb);
    out->AddMatMat(1.0, in, kNoTrans, linearity_, kTrans, 0.0, &ep, pa, pb);
  }
  // Every tensor of this layer takes part in two products -- the input in the forward product and the weight gradient, the out-diff in
  // the in-diff and the weight gradient, the weights in the forward product and the in-diff -- so its fp16 planes (csrc/split16.h) are
  // made once per step and kept by the component; the executor's epochs (nnet-nnet.cpp) say when a buffer is still what it was.
  void ForwardPlanes(const CuMatrixBase &in, const PlaneSet **pa, const PlaneSet **pb) {
    if (!gemm_split16_serves(in.NumRows(), output_dim_, input_dim_)) return;
    *pa = in_planes_.Of(in, s16_epochs().fwd);
    *pb = WeightPlanes();
  }
  // The planes of the weights: the ones the last weight-gradient product's epilogue wrote beside the updated weights (Update below), as
  // long as nothing else has touched the weights since; else made now and good for this step.
  const PlaneSet *WeightPlanes() {
    if (w_kept_ && !aliased_silently_ && w_kept_param_epoch_ == s16_param_epoch()) return &w_planes_.get();
    w_kept_ = false;
    return w_planes_.Of(linearity_, s16_epochs().fwd);
  }
  void WeightsWritten() { w_kept_ = false; maxima_valid_ = false; w_planes_.get().Invalidate(); }
  void ParamsAliased(bool announces) { aliased_silently_ = !announces; }
  // Executor peepholes (nnet-nnet.cpp): the component that writes this layer's input / out-diff leaves their planes (or the maxima the
  // conversion needs) here
  PlaneHolder &InputPlanes() { return in_planes_; }
  PlaneHolder &DiffPlanes() { return diff_planes_; }
  // Executor peephole (nnet-nnet.cpp): the next Propagate also forms the per-column statistics of its output for the
  // BatchNormalization behind it, in `buf` as aslp_gemm_epilogue.colstats lays them out (groups = ceil(rows / 32), ld = dim rounded to 4)
  void RequestOutputStats(CuVectorD *buf) { stats_request_ = buf; }
  // Executor peephole (nnet-nnet.cpp): a Sigmoid that is this component's only consumer gets its output from the same
  // GEMM (second store of the epilogue, sigmoid of the value just written to `out`): same bits as the separate launch.
  // act_planes: the AffineTransform that alone reads sigmoid_out wants its fp16 planes (bound 1): the same epilogue writes them
  void PropagateWithSigmoid(const CuMatrixBase &in, CuMatrix *out, CuMatrix *sigmoid_out, PlaneHolder *act_planes = nullptr) {
    ASLP_ASSERT(in.NumCols() == input_dim_);
    if (out->NumRows() != in.NumRows() || out->NumCols() != output_dim_) out->Resize(in.NumRows(), output_dim_, kUndefined);
    if (sigmoid_out->NumRows() != in.NumRows() || sigmoid_out->NumCols() != output_dim_) sigmoid_out->Resize(in.NumRows(), output_dim_, kUndefined);
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    ep.bias = bias_.Data();
    ep.act_out = sigmoid_out->Data(); ep.ld_act = sigmoid_out->Stride(); ep.act = 1;
    const PlaneSet *pa = nullptr, *pb = nullptr;
    ForwardPlanes(in, &pa, &pb);
    PlaneSet *ap = (act_planes && pa && pb) ? &act_planes->get() : nullptr;
    if (ap && ap->Reserve(in.NumRows(), output_dim_) && ap->SetBound(1.0f)) {
      aslp_planes_as_output(reinterpret_cast<const aslp_planes *>(ap), &ep.planes);
      ep.planes_of = 2;
    } else {
      ap = nullptr;
    }
    out->AddMatMat(1.0, in, kNoTrans, linearity_, kTrans, 0.0, &ep, pa, pb);
    if (ap && aslp_gemm_last_parts() > 0) ap->Tag(sigmoid_out->Data(), sigmoid_out->Stride(), s16_epochs().fwd);   // (the split-fp16 kernel ran)
  }
  // Executor peephole (one-shot): a Sigmoid's backward pass reads the next BackpropagateFnc's in-diff and wants the per-workgroup maxima
  // of it (the scale of ITS result's planes follows from them); they are left in `h`'s maxima array, tagged with the in-diff
  void LeaveInDiffMaxima(PlaneHolder *h) { in_diff_maxima_ = h; }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {
    PlaneHolder *mx = in_diff_maxima_;
    in_diff_maxima_ = nullptr;
    const PlaneSet *pa = nullptr, *pb = nullptr;
    if (gemm_split16_serves(out_diff.NumRows(), input_dim_, output_dim_)) {
      pa = diff_planes_.Of(out_diff, s16_epochs().bwd);
      pb = WeightPlanes();   // (the weights have not moved since the forward pass: Update comes after this)
    }
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    const bool want_max = mx && pa && pb && gemm_split16_max_parts(out_diff.NumRows(), input_dim_) <= kS16MaxParts && mx->get().ReserveParts();
    if (want_max) ep.cmax_parts = mx->get().Parts();
    in_diff->AddMatMat(1.0, out_diff, kNoTrans, linearity_, kNoTrans, 0.0, want_max ? &ep : nullptr, pa, pb);  // :193-197
    if (want_max) {
      const int n = aslp_gemm_last_parts();
      if (n > 0) mx->get().TagParts(in_diff->Data(), n, s16_epochs().bwd);
    }
  }
  void Update(const CuMatrixBase &input, const CuMatrixBase &diff) {  // :200-245
    const BaseFloat lr = opts_.learn_rate * learn_rate_coef_;
    const BaseFloat lr_bias = opts_.learn_rate * bias_learn_rate_coef_;
    const BaseFloat mmt = opts_.momentum, l2 = opts_.l2_penalty, l1 = opts_.l1_penalty;
    const int32 num_frames = input.NumRows();
    const bool plain = (l2 == 0.0 && l1 == 0.0);
    // gradient (sums over frames, incl. momentum); when no regulariser sits between the
    // gradient and the step, the SGD step W += -lr * W_corr rides in the GEMM epilogue.
    aslp_gemm_epilogue ep = aslp_gemm_epilogue();
    if (plain) { ep.W = linearity_.Data(); ep.ldw = linearity_.Stride(); ep.w_alpha = -lr; }
    // bias_corr_ = colsum(diff) + mmt * bias_corr_ and the step bias_ += -lr_bias * bias_corr_ ride on the same GEMM:
    // its transposed A operand IS `diff`, so the first column of tiles sums the fragments it multiplies anyway
    ep.colsum = bias_corr_.Data(); ep.colsum_beta = mmt; ep.colsum_w = bias_.Data(); ep.colsum_w_alpha = -lr_bias;
    const PlaneSet *pa = nullptr, *pb = nullptr;
    if (gemm_split16_serves(output_dim_, input_dim_, num_frames)) {
      pa = diff_planes_.Of(diff, s16_epochs().bwd);
      pb = in_planes_.Of(input, s16_epochs().fwd);
    }
    // The epilogue that writes the updated weights also writes their fp16 planes for the next step's forward and in-diff products
    // (csrc/split16.h).  Their scale must be known before the first tile is written: a bound of max |W - lr W_corr| that every
    // workgroup forms from the maxima of |W| and |W_corr| the previous step's epilogue left (the first time: one pass over each),
    // the operands' bounds and K.
    const bool keep = s16_keep_weight_planes() && plain && max_norm_ <= 0.0 && !aliased_silently_ && pa && pb && gemm_split16_serves(num_frames, output_dim_, input_dim_) &&
                      gemm_split16_max_parts(output_dim_, input_dim_) <= kS16MaxParts && w_planes_.get().Reserve(output_dim_, input_dim_);
    if (keep) {
      for (int i = 0; i < 2; i++) {
        if (w_maxima_[i].Dim() != kS16MaxParts) { w_maxima_[i].Resize(kS16MaxParts); maxima_valid_ = false; }
        if (c_maxima_[i].Dim() != kS16MaxParts) { c_maxima_[i].Resize(kS16MaxParts); maxima_valid_ = false; }
      }
      if (!maxima_valid_ || maxima_param_epoch_ != s16_param_epoch()) {
        aslp_absmax_parts(linearity_.Data(), linearity_.Dim(), w_maxima_[cur_].Data());
        aslp_absmax_parts(linearity_corr_.Data(), linearity_corr_.Dim(), c_maxima_[cur_].Data());
        n_maxima_ = 256;
      }
      aslp_planes_as_output(reinterpret_cast<const aslp_planes *>(&w_planes_.get()), &ep.planes);
      w_planes_.get().ForgetHostBound();   // (the kernel writes the slot)
      ep.planes_of = 1;
      ep.bound_w_parts = w_maxima_[cur_].Data();   // the kernel forms the bound itself before its first tile
      ep.bound_c_parts = c_maxima_[cur_].Data();
      ep.bound_n = n_maxima_;
      ep.wmax_parts = w_maxima_[1 - cur_].Data();
      ep.cmax_parts = c_maxima_[1 - cur_].Data();
    }
    linearity_corr_.AddMatMat(1.0, diff, kTrans, input, kNoTrans, mmt, &ep, pa, pb);
    const int left = keep ? aslp_gemm_last_parts() : 0;
    if (left > 0) {   // the split-fp16 kernel ran: planes and maxima are those of the weights as they are now
      cur_ = 1 - cur_;
      n_maxima_ = left;
      maxima_valid_ = true;
      maxima_param_epoch_ = w_kept_param_epoch_ = s16_param_epoch();
      w_kept_ = true;
    } else {
      WeightsWritten();
    }
    if (!plain) {
      if (l2 != 0.0) linearity_.AddMat(-lr * l2 * num_frames, linearity_);
      if (l1 != 0.0) cu::RegularizeL1(&linearity_, &linearity_corr_, lr * l1 * num_frames, lr);
      linearity_.AddMat(-lr, linearity_corr_);
    }
    if (max_norm_ > 0.0) { aslp_max_norm_rows(linearity_.Data(), linearity_.Dim(), max_norm_); WeightsWritten(); }  // :231-243
  }
  const CuVectorBase &GetBias() const { return bias_; }
  void SetBias(const CuVectorBase &bias) { ASLP_ASSERT(bias.Dim() == bias_.Dim()); bias_.CopyFromVec(bias); }
  const CuMatrixBase &GetLinearity() const { return linearity_; }
  void SetLinearity(const CuMatrixBase &l) { ASLP_ASSERT(SameDim(l, linearity_)); linearity_.CopyFromMat(l); WeightsWritten(); }
  const CuVectorBase &GetBiasCorr() const { return bias_corr_; }
  const CuMatrixBase &GetLinearityCorr() const { return linearity_corr_; }

 private:
  CuVectorD *stats_request_ = nullptr;
  PlaneHolder *in_diff_maxima_ = nullptr;
  PlaneHolder in_planes_, diff_planes_, w_planes_;
  // weights' planes kept from step to step (Update): valid flag, the parameter epoch they belong to, per-workgroup maxima of |W| and
  // |W_corr| in two alternating arrays (one is read by the bound kernel while the epilogue fills the other)
  bool w_kept_ = false, maxima_valid_ = false, aliased_silently_ = false;
  long w_kept_param_epoch_ = -1, maxima_param_epoch_ = -1;
  CuVector w_maxima_[2], c_maxima_[2];
  int n_maxima_ = 0, cur_ = 0;
  CuMatrix linearity_;
  CuVector bias_;
  CuMatrix linearity_corr_;
  CuVector bias_corr_;
  BaseFloat learn_rate_coef_, bias_learn_rate_coef_, max_norm_;
};

// ---- activations (nnet-activation.h) -----------------------------------------------------------------
class Softmax : public Component {
 public:
  Softmax(int32 di, int32 dout) : Component(di, dout) {}
  Component *Copy() const { return new Softmax(*this); }
  ComponentType GetType() const { return kSoftmax; }
  bool BackpropIsCopy() const { return true; }  // :51-59
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { out->ApplySoftMaxPerRow(in); }
  // :51-59 backward is a plain copy: out_diff already is (y - t)
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) { id->CopyFromMat(od); }
};

class BlockSoftmax : public Component {  // :64-143
 public:
  BlockSoftmax(int32 di, int32 dout) : Component(di, dout) {}
  Component *Copy() const { return new BlockSoftmax(*this); }
  ComponentType GetType() const { return kBlockSoftmax; }
  void InitData(std::istream &is) {
    std::string token, dims_str;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<BlockDims>") is >> dims_str;
      else ASLP_ERR << "Unknown token " << token << ", a typo in config? (BlockDims)";
      is >> std::ws;
    }
    if (!SplitStringToIntegers(dims_str, ",:", false, &block_dims)) ASLP_ERR << "Invalid block-dims " << dims_str;
    SetOffsets();
  }
  void ReadData(std::istream &is, bool binary) { ReadIntegerVector(is, binary, &block_dims); SetOffsets(); }
  void WriteData(std::ostream &os, bool binary) const { WriteIntegerVector(os, binary, block_dims); }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {
    for (size_t bl = 0; bl < block_dims.size(); bl++) {
      CuSubMatrix in_bl = in.ColRange(block_offset[bl], block_dims[bl]);
      CuSubMatrix out_bl = out->ColRange(block_offset[bl], block_dims[bl]);
      out_bl.ApplySoftMaxPerRow(in_bl);
    }
  }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) {
    id->CopyFromMat(od);
    for (size_t bl = 0; bl < block_dims.size(); bl++) {  // :118-131 zero rows whose block-sum is non-zero
      CuSubMatrix diff_bl = id->ColRange(block_offset[bl], block_dims[bl]);
      CuVector row_sum(diff_bl.NumRows());
      row_sum.AddColSumMat(1.0, diff_bl, 0.0);
      CuVector mask(row_sum);
      mask.Scale(-1.0);
      mask.Add(1.0);
      diff_bl.MulRowsVec(mask);
    }
  }
  std::vector<int32> block_dims, block_offset;
 private:
  void SetOffsets() {
    block_offset.assign(block_dims.size() + 1, 0);
    for (size_t i = 0; i < block_dims.size(); i++) block_offset[i + 1] = block_offset[i] + block_dims[i];
    ASLP_ASSERT(OutputDim() == block_offset.back());
  }
};

class Sigmoid : public Component {
 public:
  Sigmoid(int32 di, int32 dout) : Component(di, dout) {}
  Component *Copy() const { Sigmoid *c = new Sigmoid(*this); c->in_diff_planes_ = nullptr; return c; }
  ComponentType GetType() const { return kSigmoid; }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { out->Sigmoid(in); }
  // Executor peephole (one-shot): the AffineTransform that reads the next BackpropagateFnc's in-diff as ITS out-diff wants the fp16 planes
  // of it (csrc/split16.h), and `h` already holds the maxima of this pass' out-diff (AffineTransform::LeaveInDiffMaxima above):
  // |od y (1 - y)| <= max |od| / 4, so the same launch writes in-diff and planes.
  void ProduceInDiffPlanes(PlaneHolder *h) { in_diff_planes_ = h; }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &out, const CuMatrixBase &od, CuMatrixBase *id) {
    PlaneHolder *h = in_diff_planes_;
    in_diff_planes_ = nullptr;
    const long epoch = s16_epochs().bwd;
    const int np = h ? h->get().PartsFor(od.Data(), epoch) : 0;
    if (np > 0 && h->get().Reserve(id->NumRows(), id->NumCols())) {
      aslp_planes_out po = aslp_planes_out();
      aslp_planes_as_output(reinterpret_cast<const aslp_planes *>(&h->get()), &po);
      h->get().ForgetHostBound();   // (the kernel stores the bound)
      if (aslp_diff_sigmoid_p(id->Data(), od.Data(), out.Data(), id->Dim(), od.Stride(), out.Stride(), h->get().Parts(), np, &po))
        h->get().Tag(id->Data(), id->Stride(), epoch);
      char err[512];
      if (aslp_get_last_error(err, sizeof(err))) ASLP_ERR << err;
      return;
    }
    id->DiffSigmoid(out, od);
  }

 private:
  PlaneHolder *in_diff_planes_ = nullptr;
};
class Tanh : public Component {
 public:
  Tanh(int32 di, int32 dout) : Component(di, dout) {}
  Component *Copy() const { return new Tanh(*this); }
  ComponentType GetType() const { return kTanh; }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { out->Tanh(in); }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &out, const CuMatrixBase &od, CuMatrixBase *id) { id->DiffTanh(out, od); }
};
class Dropout : public Component {  // nnet-activation.h:203-273
 public:
  Dropout(int32 di, int32 dout) : Component(di, dout), dropout_retention_(0.5), calls_(0) {}
  Component *Copy() const { return new Dropout(*this); }
  ComponentType GetType() const { return kDropout; }
  void InitData(std::istream &is) {
    is >> std::ws;
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<DropoutRetention>") ReadBasicType(is, false, &dropout_retention_);
      else ASLP_ERR << "Unknown token " << token << ", a typo in config?" << " (DropoutRetention)";
      is >> std::ws;
    }
    ASLP_ASSERT(dropout_retention_ > 0.0 && dropout_retention_ <= 1.0);
  }
  void ReadData(std::istream &is, bool binary) {
    if ('<' == Peek(is, binary)) {
      ExpectToken(is, binary, "<DropoutRetention>");
      ReadBasicType(is, binary, &dropout_retention_);
    }
    ASLP_ASSERT(dropout_retention_ > 0.0 && dropout_retention_ <= 1.0);
  }
  void WriteData(std::ostream &os, bool binary) const {
    WriteToken(os, binary, "<DropoutRetention>");
    WriteBasicType(os, binary, dropout_retention_);
  }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {
    if (dropout_mask_.NumRows() != in.NumRows() || dropout_mask_.NumCols() != in.NumCols()) dropout_mask_.Resize(in.NumRows(), in.NumCols(), kUndefined);
    // a fresh mask per call: the key is drawn from the engine's seeded generator once, then advanced by the call count
    if (calls_ == 0) seed_ = ((unsigned long long)(unsigned)Rand() << 32) ^ (unsigned)Rand();
    aslp_dropout_forward(out->Data(), out->Stride(), in.Data(), in.Dim(), dropout_mask_.Data(), dropout_mask_.Stride(), dropout_retention_,
                         seed_ + 0xD1B54A32D192ED03ull * calls_);
    calls_++;
  }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) {
    ASLP_ASSERT(SameDim(od, dropout_mask_));
    aslp_dropout_backward(id->Data(), id->Stride(), od.Data(), od.Dim(), dropout_mask_.Data(), dropout_mask_.Stride(), dropout_retention_);
  }
  BaseFloat GetDropoutRetention() const { return dropout_retention_; }
  void SetDropoutRetention(BaseFloat dr) {
    dropout_retention_ = dr;
    ASLP_ASSERT(dropout_retention_ > 0.0 && dropout_retention_ <= 1.0);
  }
  const CuMatrixBase &Mask() const { return dropout_mask_; }

 private:
  CuMatrix dropout_mask_;
  BaseFloat dropout_retention_;
  unsigned long long seed_ = 0, calls_;
};
class ReLU : public Component {  // :281-298
 public:
  ReLU(int32 di, int32 dout) : Component(di, dout) {}
  Component *Copy() const { return new ReLU(*this); }
  ComponentType GetType() const { return kReLU; }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { out->CopyFromMat(in); out->ApplyFloor(0.0); }
  void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) {
    aslp_diff_relu(id->Data(), in.Data(), od.Data(), id->Dim(), in.Stride(), od.Stride());
  }
};

// ---- Splice / Copy (nnet-various.h:43-330) ----------------------------------------------------------------
inline void ParseBuildVector(std::istream &is, std::vector<int32> *out, const char *what) {
  // nnet-various.h:54-107: <ReadVector> [ ... ] | <BuildVector> a:b a:s:b c </BuildVector>
  std::vector<std::vector<int32>> build_vector;
  std::string token;
  while (!is.eof()) {
    ReadToken(is, false, &token);
    if (token == "<ReadVector>") {
      ReadIntegerVector(is, false, out);
    } else if (token == "<BuildVector>") {
      while (!is.eof()) {
        std::string s;
        ReadToken(is, false, &s);
        if (s == "</BuildVector>") break;
        std::vector<int32> v;
        SplitStringToIntegers(s, ":", false, &v);
        build_vector.push_back(v);
      }
    } else {
      ASLP_ERR << "Unknown token " << token << ", a typo in config? (ReadVector|BuildVector)";
    }
    is >> std::ws;
  }
  for (auto &b : build_vector) {
    switch (b.size()) {
      case 1: out->push_back(b[0]); break;
      case 2: {
        ASLP_ASSERT(b[0] <= b[1]);
        for (int32 j = b[0]; j <= b[1]; j++) out->push_back(j);
      } break;
      case 3: {
        int32 mn = b[0], step = b[1], mx = b[2];
        ASLP_ASSERT((mn <= mx && step > 0) || (mn >= mx && step < 0));
        for (int32 j = mn; j <= mx; j += step) out->push_back(j);
      } break;
      default: ASLP_ERR << "Error parsing <BuildVector> of " << what;
    }
  }
}

class Splice : public Component {
 public:
  Splice(int32 di, int32 dout) : Component(di, dout) {}
  Component *Copy() const { return new Splice(*this); }
  ComponentType GetType() const { return kSplice; }
  void InitData(std::istream &is) {
    std::vector<int32> fo;
    ParseBuildVector(is, &fo, "Splice");
    frame_offsets_ = fo;
    ASLP_ASSERT(frame_offsets_.Dim() * InputDim() == OutputDim());
  }
  void ReadData(std::istream &is, bool binary) {
    std::vector<int32> fo;
    ReadIntegerVector(is, binary, &fo);
    frame_offsets_ = fo;
    ASLP_ASSERT(frame_offsets_.Dim() * InputDim() == OutputDim());
  }
  void WriteData(std::ostream &os, bool binary) const {
    std::vector<int32> fo;
    frame_offsets_.CopyToVec(&fo);
    WriteIntegerVector(os, binary, fo);
  }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { cu::Splice(in, frame_offsets_, out); }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) {
    // :143-175: one gather-sum kernel instead of n_splice CopyRows + AddMat passes
    aslp_splice_backward(id->Data(), id->Dim(), od.Data(), od.Stride(), frame_offsets_.Data(), frame_offsets_.Dim());
  }
 protected:
  CuArray<int32> frame_offsets_;
};

class CopyComponent : public Component {
 public:
  CopyComponent(int32 di, int32 dout) : Component(di, dout) {}
  Component *Copy() const { return new CopyComponent(*this); }
  ComponentType GetType() const { return kCopy; }
  void InitData(std::istream &is) {
    std::vector<int32> idx;
    ParseBuildVector(is, &idx, "Copy");
    for (auto &v : idx) --v;  // matlab indexing in the config (:253-255)
    for (int32 v : idx) ASLP_ASSERT(v >= 0 && v < InputDim());
    copy_from_indices_ = idx;
    ASLP_ASSERT(copy_from_indices_.Dim() == OutputDim());
  }
  void ReadData(std::istream &is, bool binary) {
    std::vector<int32> idx;
    ReadIntegerVector(is, binary, &idx);
    for (auto &v : idx) --v;
    copy_from_indices_ = idx;
    ASLP_ASSERT(copy_from_indices_.Dim() == OutputDim());
  }
  void WriteData(std::ostream &os, bool binary) const {
    std::vector<int32> idx;
    copy_from_indices_.CopyToVec(&idx);
    for (auto &v : idx) ++v;
    WriteIntegerVector(os, binary, idx);
  }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { cu::Copy(in, copy_from_indices_, out); }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &, CuMatrixBase *id) {
    static bool warned = false;  // :300-308 "Not implemented!": zero diff
    if (!warned) { ASLP_WARN << __func__ << "Not implemented!"; warned = true; }
    id->SetZero();
  }
 protected:
  CuArray<int32> copy_from_indices_;
};

// ---- AddShift / Rescale (nnet-various.h:365-590) ----------------------------------------------------------
class AddShift : public UpdatableComponent {
 public:
  AddShift(int32 di, int32 dout) : UpdatableComponent(di, dout), shift_data_(di), learn_rate_coef_(1.0) {}
  Component *Copy() const { return new AddShift(*this); }
  ComponentType GetType() const { return kAddShift; }
  void InitData(std::istream &is) {
    float init_param = 0.0;
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<InitParam>") ReadBasicType(is, false, &init_param);
      else if (token == "<LearnRateCoef>") ReadBasicType(is, false, &learn_rate_coef_);
      else ASLP_ERR << "Unknown token " << token << ", a typo in config? (InitParam)";
      is >> std::ws;
    }
    shift_data_.Resize(InputDim(), kSetZero);
    shift_data_.Set(init_param);
  }
  void ReadData(std::istream &is, bool binary) {
    if ('<' == Peek(is, binary)) { ExpectToken(is, binary, "<LearnRateCoef>"); ReadBasicType(is, binary, &learn_rate_coef_); }
    shift_data_.Read(is, binary);
  }
  void WriteData(std::ostream &os, bool binary) const {
    WriteToken(os, binary, "<LearnRateCoef>"); WriteBasicType(os, binary, learn_rate_coef_);
    shift_data_.Write(os, binary);
  }
  int32 NumParams() const { return shift_data_.Dim(); }
  void GetParams(std::vector<BaseFloat> *w) const { w->clear(); AppendVector(shift_data_, w); }
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *p) { p->clear(); p->push_back(std::make_pair(shift_data_.Data(), shift_data_.Dim())); }
  std::string Info() const { return std::string("\n  shift_data") + MomentStatistics(shift_data_); }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { out->CopyFromMat(in); out->AddVecToRows(1.0, shift_data_, 1.0); }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) { id->CopyFromMat(od); }
  void Update(const CuMatrixBase &, const CuMatrixBase &diff) {
    const BaseFloat lr = opts_.learn_rate;
    shift_data_grad_.Resize(InputDim(), kSetZero);
    shift_data_grad_.AddRowSumMat(1.0, diff, 0.0);
    shift_data_.AddVec(-lr * learn_rate_coef_, shift_data_grad_);
  }
 protected:
  CuVector shift_data_, shift_data_grad_;
  BaseFloat learn_rate_coef_;
};

class Rescale : public UpdatableComponent {
 public:
  Rescale(int32 di, int32 dout) : UpdatableComponent(di, dout), scale_data_(di), learn_rate_coef_(1.0) {}
  Component *Copy() const { return new Rescale(*this); }
  ComponentType GetType() const { return kRescale; }
  void InitData(std::istream &is) {
    float init_param = 0.0;
    std::string token;
    while (!is.eof()) {
      ReadToken(is, false, &token);
      if (token == "<InitParam>") ReadBasicType(is, false, &init_param);
      else if (token == "<LearnRateCoef>") ReadBasicType(is, false, &learn_rate_coef_);
      else ASLP_ERR << "Unknown token " << token << ", a typo in config? (InitParam)";
      is >> std::ws;
    }
    scale_data_.Resize(InputDim(), kSetZero);
    scale_data_.Set(init_param);
  }
  void ReadData(std::istream &is, bool binary) {
    if ('<' == Peek(is, binary)) { ExpectToken(is, binary, "<LearnRateCoef>"); ReadBasicType(is, binary, &learn_rate_coef_); }
    scale_data_.Read(is, binary);
  }
  void WriteData(std::ostream &os, bool binary) const {
    WriteToken(os, binary, "<LearnRateCoef>"); WriteBasicType(os, binary, learn_rate_coef_);
    scale_data_.Write(os, binary);
  }
  int32 NumParams() const { return scale_data_.Dim(); }
  void GetParams(std::vector<BaseFloat> *w) const { w->clear(); AppendVector(scale_data_, w); }
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *p) { p->clear(); p->push_back(std::make_pair(scale_data_.Data(), scale_data_.Dim())); }
  std::string Info() const { return std::string("\n  scale_data") + MomentStatistics(scale_data_); }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) { out->CopyFromMat(in); out->MulColsVec(scale_data_); }
  void BackpropagateFnc(const CuMatrixBase &, const CuMatrixBase &, const CuMatrixBase &od, CuMatrixBase *id) { id->CopyFromMat(od); id->MulColsVec(scale_data_); }
  void Update(const CuMatrixBase &input, const CuMatrixBase &diff) {
    const BaseFloat lr = opts_.learn_rate;
    scale_data_grad_.Resize(InputDim(), kSetZero);
    CuMatrix gradient_aux(diff);
    gradient_aux.MulElements(input);
    scale_data_grad_.AddRowSumMat(1.0, gradient_aux, 0.0);
    scale_data_.AddVec(-lr * learn_rate_coef_, scale_data_grad_);
  }
 protected:
  CuVector scale_data_, scale_data_grad_;
  BaseFloat learn_rate_coef_;
};

// ---- BatchNormalization (nnet-batch-normalization.h:32-298) ----------------------------------------------------
class BatchNormalization : public UpdatableComponent {
 public:
  BatchNormalization(int32 di, int32 dout)
      : UpdatableComponent(di, dout), var_floor_(0.0000001), num_acc_frames_(0), acc_cleaned_(false) {}
  Component *Copy() const { return new BatchNormalization(*this); }
  ComponentType GetType() const { return kBatchNormalization; }
  bool GradientInBackprop() const { return true; }
  void InitData(std::istream &) {  // :45-54
    num_acc_frames_ = 0;
    scale_.Resize(output_dim_); scale_.Set(1.0);
    shift_.Resize(output_dim_); shift_.SetZero();
    ASLP_ASSERT(output_dim_ > 0 && input_dim_ > 0);
    acc_means_.Resize(output_dim_, kSetZero);
    acc_vars_.Resize(output_dim_, kSetZero);
    AllocAux();
  }
  void ReadData(std::istream &is, bool binary) {  // :56-94
    ExpectToken(is, binary, "<NumAccFrames>");
    ReadBasicType(is, binary, &num_acc_frames_);
    acc_means_.Read(is, binary);
    acc_vars_.Read(is, binary);
    shift_.Read(is, binary);
    scale_.Read(is, binary);
    ASLP_ASSERT(acc_means_.Dim() == acc_vars_.Dim());
    ASLP_ASSERT(acc_means_.Dim() == shift_.Dim());
    ASLP_ASSERT(acc_means_.Dim() == scale_.Dim());
    AllocAux();
    mean_vec_.SetZero();
    var_vec_.Set(1.0);
    if (num_acc_frames_ <= 0.0) return;
    float var_floor = 1e-10;
    int D = acc_means_.Dim();
    std::vector<double> am(D), av(D);
    acc_means_.CopyToHost(am.data());
    acc_vars_.CopyToHost(av.data());
    std::vector<float> mh(D), vh(D);
    for (int32 d = 0; d < D; d++) {
      BaseFloat mean = am[d] / num_acc_frames_;
      BaseFloat var = av[d] / num_acc_frames_ - mean * mean;
      if (var <= var_floor) { ASLP_WARN << "Very small variance " << var << " flooring to " << var_floor; var = var_floor; }
      mh[d] = mean;
      vh[d] = 1.0 / sqrt(var + var_floor_);
    }
    mean_vec_.CopyFromHost(mh.data(), D);
    var_vec_.CopyFromHost(vh.data(), D);
  }
  void WriteData(std::ostream &os, bool binary) const {  // :96-103
    WriteToken(os, binary, "<NumAccFrames>");
    WriteBasicType(os, binary, num_acc_frames_);
    acc_means_.Write(os, binary);
    acc_vars_.Write(os, binary);
    shift_.Write(os, binary);
    scale_.Write(os, binary);
  }
  int32 NumParams() const { return shift_.Dim() + scale_.Dim(); }
  void GetParams(std::vector<BaseFloat> *w) const { w->clear(); AppendVector(shift_, w); AppendVector(scale_, w); }
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) {  // :115-119
    params->clear();
    params->push_back(std::make_pair(shift_.Data(), shift_.Dim()));
    params->push_back(std::make_pair(scale_.Data(), scale_.Dim()));
  }
  double *GetAccStats(std::vector<std::pair<double *, int>> *params) {  // :122-127
    params->clear();
    params->push_back(std::make_pair(acc_means_.Data(), acc_means_.Dim()));
    params->push_back(std::make_pair(acc_vars_.Data(), acc_vars_.Dim()));
    return &num_acc_frames_;
  }
  std::string Info() const { return std::string("\n  batch_normaliztion"); }
  void CleanAccs() { acc_means_.SetZero(); acc_vars_.SetZero(); num_acc_frames_ = 0; }

  void FeedforwardFnc(const CuMatrixBase &in, CuMatrixBase *out) {  // :139-175
    if (num_acc_frames_ <= 0) {  // local statistics
      aslp_bn_forward(in.Data(), in.Dim(), out->Data(), out->Stride(), nullptr, 0, scale_.Data(), shift_.Data(), mean_vec_.Data(),
                      var_vec_.Data(), nullptr, nullptr, var_floor_);
    } else {  // global statistics prepared by ReadData
      aslp_bn_apply(in.Data(), in.Dim(), out->Data(), out->Stride(), mean_vec_.Data(), var_vec_.Data(), scale_.Data(), shift_.Data());
    }
  }
  void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) {  // :177-220
    if (!acc_cleaned_) { acc_cleaned_ = true; CleanAccs(); }
    BaseFloat *xhat = XhatFor(in);
    if (ForwardFromStats(in, out->Data(), out->Stride(), nullptr, 0)) return;
    aslp_bn_forward(in.Data(), in.Dim(), out->Data(), out->Stride(), xhat, XsharpO_.Stride(), scale_.Data(), shift_.Data(),
                    mean_vec_.Data(), var_vec_.Data(), acc_means_.Data(), acc_vars_.Data(), var_floor_);
    num_acc_frames_ += in.NumRows();
  }
  void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &, const CuMatrixBase &out_diff, CuMatrixBase *in_diff) {  // :222-277
    Backward(in, out_diff, in_diff ? in_diff->Data() : nullptr, in_diff ? in_diff->Stride() : 0, nullptr, 0);
  }
  // the SGD step of scale / shift is then taken inside the backward statistics finalize (see nnet-component.h)
  void FoldNextUpdateIntoBackprop() { fold_update_ = true; }
  // Executor peephole (nnet-nnet.cpp): a Sigmoid that is this component's only consumer is folded into the write
  // pass (forward) and into the statistics / write passes (backward); the BN output itself is never materialised.
  // Executor peephole: the producer of the next Propagate's input (an AffineTransform) has left the column statistics of that
  // input in `stats` (AffineTransform::RequestOutputStats); one-shot.
  void UseInputStats(const CuVectorD *stats) { input_stats_ = stats; }
  // Executor peepholes (one-shot): the AffineTransform that alone reads the next PropagateWithSigmoid's output wants its fp16 planes
  // (csrc/split16.h; sigmoid outputs: bound 1, written by the same launch); the AffineTransform whose out-diff the next
  // BackpropagateWithSigmoid writes wants the maxima its conversion needs.
  void ProduceOutputPlanes(PlaneHolder *h) { out_planes_ = h; }
  void LeaveDiffMaxima(PlaneHolder *h) { diff_maxima_ = h; }
  void PropagateWithSigmoid(const CuMatrixBase &in, CuMatrix *sigmoid_out) {
    ASLP_ASSERT(in.NumCols() == input_dim_);
    if (!acc_cleaned_) { acc_cleaned_ = true; CleanAccs(); }
    BaseFloat *xhat = XhatFor(in);
    if (sigmoid_out->NumRows() != in.NumRows() || sigmoid_out->NumCols() != output_dim_) sigmoid_out->Resize(in.NumRows(), output_dim_, kUndefined);
    if (ForwardFromStats(in, nullptr, 0, sigmoid_out->Data(), sigmoid_out->Stride())) return;
    aslp_bn_forward_act(in.Data(), in.Dim(), nullptr, 0, xhat, XsharpO_.Stride(), scale_.Data(), shift_.Data(), mean_vec_.Data(),
                        var_vec_.Data(), acc_means_.Data(), acc_vars_.Data(), var_floor_, sigmoid_out->Data(), sigmoid_out->Stride());
    num_acc_frames_ += in.NumRows();
  }
  void BackpropagateWithSigmoid(const CuMatrixBase &in, const CuMatrixBase &sigmoid_out, const CuMatrixBase &sigmoid_out_diff, CuMatrix *in_diff) {
    ASLP_ASSERT(SameDim(sigmoid_out, sigmoid_out_diff) && sigmoid_out.NumCols() == output_dim_);
    if (in_diff && (in_diff->NumRows() != in.NumRows() || in_diff->NumCols() != input_dim_)) in_diff->Resize(in.NumRows(), input_dim_, kUndefined);
    Backward(in, sigmoid_out_diff, in_diff ? in_diff->Data() : nullptr, in_diff ? in_diff->Stride() : 0, sigmoid_out.Data(),
             sigmoid_out.Stride());
  }
  void Update(const CuMatrixBase &, const CuMatrixBase &) {  // :280-284
    if (update_done_) { update_done_ = false; return; }
    const BaseFloat lr = opts_.learn_rate;
    aslp_vec_axpy2(-lr, dscale_.Data(), scale_.Data(), dshift_.Data(), shift_.Data(), scale_.Dim());
  }
  CuVector &Scale() { return scale_; }
  CuVector &Shift() { return shift_; }
  double NumAccFrames() const { return num_acc_frames_; }
  CuVectorD &AccMeans() { return acc_means_; }
  CuVectorD &AccVars() { return acc_vars_; }

 private:
  const CuVectorD *input_stats_ = nullptr;
  PlaneHolder *out_planes_ = nullptr, *diff_maxima_ = nullptr;
  void Backward(const CuMatrixBase &in, const CuMatrixBase &out_diff, BaseFloat *in_diff, int32 id_stride, const BaseFloat *act_y, int32 act_stride) {
    PlaneHolder *dm = diff_maxima_;
    diff_maxima_ = nullptr;
    if (fold_update_) {
      fold_update_ = false;
      update_done_ = true;
      aslp_planes_out po = aslp_planes_out();
      if (dm && in_diff && dm->get().Reserve(in.NumRows(), input_dim_)) {
        aslp_planes_as_output(reinterpret_cast<const aslp_planes *>(&dm->get()), &po);
      }
      aslp_bn_backward_step_p(in.Dim(), out_diff.Data(), out_diff.Stride(), xhat_kept_ ? XsharpO_.Data() : nullptr, XsharpO_.Stride(), scale_.Data(),
                              shift_.Data(), var_vec_.Data(), dscale_.Data(), dshift_.Data(), opts_.momentum, opts_.learn_rate, in_diff, id_stride, act_y,
                              act_stride, in.Data(), mean_vec_.Data(), &po);
      if (po.planes_written) {   // in_diff's planes came out of the launch itself (bound: the matrix maximum, found by its workgroups)
        dm->get().ForgetHostBound();
        dm->get().Tag(in_diff, id_stride, s16_epochs().bwd);
      } else if (po.nparts > 0) {
        dm->get().TagParts(in_diff, po.nparts, s16_epochs().bwd);
      }
    } else {
      aslp_bn_backward_act(in.Data(), in.Dim(), out_diff.Data(), out_diff.Stride(), xhat_kept_ ? XsharpO_.Data() : nullptr, XsharpO_.Stride(), scale_.Data(),
                           mean_vec_.Data(), var_vec_.Data(), dscale_.Data(), dshift_.Data(), opts_.momentum, in_diff, id_stride, act_y, act_stride);
    }
  }
  // statistics handed over by the producer (UseInputStats): the launch that only streams the input once; false = not served
  bool ForwardFromStats(const CuMatrixBase &in, BaseFloat *out, int32 out_stride, BaseFloat *act, int32 act_stride) {
    const CuVectorD *st = input_stats_;
    input_stats_ = nullptr;
    PlaneHolder *oh = out_planes_;
    out_planes_ = nullptr;
    if (st == nullptr || xhat_kept_) return false;
    const int groups = (in.NumRows() + 31) / 32, ld = (input_dim_ + 3) & ~3;
    if (st->Dim() != 3 * groups * ld) return false;
    aslp_planes_out po = aslp_planes_out();
    PlaneSet *ps = (oh && act) ? &oh->get() : nullptr;
    if (ps && ps->Reserve(in.NumRows(), output_dim_) && ps->SetBound(1.0f)) aslp_planes_as_output(reinterpret_cast<const aslp_planes *>(ps), &po);
    else ps = nullptr;
    if (!aslp_bn_forward_stats_p(in.Data(), in.Dim(), out, out_stride, scale_.Data(), shift_.Data(), mean_vec_.Data(), var_vec_.Data(), acc_means_.Data(),
                                 acc_vars_.Data(), var_floor_, act, act_stride, st->Data(), groups, ld, ps ? &po : nullptr))
      return false;
    if (ps) ps->Tag(act, act_stride, s16_epochs().fwd);
    num_acc_frames_ += in.NumRows();
    return true;
  }
  // The normalised copy X# of the input (the reference's XsharpO_, nnet-batch-normalization.h:188) is only materialised where
  // the kernels need it: the single-launch panel kernels form it again from the layer input in the backward pass.
  BaseFloat *XhatFor(const CuMatrixBase &in) {
    xhat_kept_ = !(aslp_bn_panel_supported(in.NumRows(), output_dim_) && in.Stride() % 4 == 0 && (reinterpret_cast<uintptr_t>(in.Data()) & 15u) == 0);
    if (!xhat_kept_) return nullptr;
    if (XsharpO_.NumRows() != in.NumRows() || XsharpO_.NumCols() != output_dim_) XsharpO_.Resize(in.NumRows(), output_dim_, kUndefined);
    return XsharpO_.Data();
  }
  void AllocAux() {
    mean_vec_.Resize(output_dim_); var_vec_.Resize(output_dim_);
    dscale_.Resize(output_dim_); dshift_.Resize(output_dim_);
  }
  CuMatrix XsharpO_;
  CuVector mean_vec_, var_vec_, scale_, dscale_, shift_, dshift_;
  BaseFloat var_floor_;
  CuVectorD acc_means_, acc_vars_;
  double num_acc_frames_;
  bool acc_cleaned_;
  bool fold_update_ = false, update_done_ = false, xhat_kept_ = true;
};

}  // namespace aslp
