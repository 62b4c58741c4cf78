// nnet-component.h -- Component / UpdatableComponent interface of the host engine.
//
// Same public surface as the reference (src/aslp-nnet/nnet-component.h:45-347): type enum and
// marker table, Propagate / Backpropagate wrappers that check dims and size the output,
// virtual PropagateFnc / BackpropagateFnc / Update, graph bookkeeping (id, name, inputs,
// offsets), Init (one <NnetProto> line), Read / Write (nnet file format).  Error behaviour:
// every KALDI_ERR / KALDI_ASSERT of the reference is a std::runtime_error here.
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "cu-matrix.h"

namespace aslp {

class OptionsItf;
struct NnetTrainOptions {  // nnet-trnopts.h:29-47
  BaseFloat learn_rate, momentum, l2_penalty, l1_penalty;
  NnetTrainOptions() : learn_rate(0.008), momentum(0.0), l2_penalty(0.0), l1_penalty(0.0) {}
  void Register(OptionsItf *opts);   // nnet-trnopts.h:42-47 (defined in data-reader.h, beside the option strings)
};

class Component {
 public:
  typedef enum {  // nnet-component.h:51-107 (same numeric values)
    kUnknown = 0x0,
    kUpdatableComponent = 0x0100,
    kAffineTransform,
    kLinearTransform,
    kConvolutionalComponent,
    kConvolutional2DComponent,
    kLstmProjectedStreams,
    kBLstmProjectedStreams,
    kActivationFunction = 0x0200,
    kSoftmax,
    kBlockSoftmax,
    kSigmoid,
    kTanh,
    kDropout,
    kReLU,
    kLengthNormComponent,
    kTranform = 0x0400,
    kRbm,
    kSplice,
    kCopy,
    kTranspose,
    kBlockLinearity,
    kAddShift,
    kRescale,
    kKlHmm = 0x0800,
    kSentenceAveragingComponent,
    kSimpleSentenceAveragingComponent,
    kAveragePoolingComponent,
    kAveragePooling2DComponent,
    kMaxPoolingComponent,
    kMaxPooling2DComponent,
    kFramePoolingComponent,
    kParallelComponent,
    kBatchNormalization = 0x0f00,
    kInputLayer,
    kOutputLayer,
    kScaleLayer,
    kLstm,
    kBLstm,
    kRowConvolution,
    kBLstmProjectedStreamsLC,
    kGruStreams,
    kLstmCifgProjectedStreams,
    kCompactFsmn,
    kPnormComponent,
    kMaxoutComponent
  } ComponentType;
  struct key_value {
    const ComponentType key;
    const char *value;
  };
  static const struct key_value kMarkerMap[];
  static const char *TypeToMarker(ComponentType t);
  static ComponentType MarkerToType(const std::string &s);

  Component(int32 input_dim, int32 output_dim) : input_dim_(input_dim), output_dim_(output_dim), id_(-1) {}
  virtual ~Component() {}
  virtual Component *Copy() const = 0;
  virtual ComponentType GetType() const = 0;
  virtual bool IsUpdatable() const { return false; }

  int32 InputDim() const { return input_dim_; }
  int32 OutputDim() const { return output_dim_; }
  int32 Id() const { return id_; }
  int32 GetId() const { return id_; }
  void SetId(int id) { id_ = id; }
  void SetName(const std::string &name) { name_ = name; }
  const std::string &GetName() const { return name_; }
  const std::vector<int32> &GetInput() const { return input_; }
  void SetInput(const std::vector<int32> &input) { input_ = input; }
  void SetInputName(const std::vector<std::string> &n) { input_name_ = n; }
  const std::vector<std::string> &GetInputName() const { return input_name_; }
  void SetMonoInput(int id) {
    input_.assign(1, id);
    offset_.assign(1, 0);
  }
  const std::vector<int32> &GetOffset() const { return offset_; }
  void SetOffset(const std::vector<int32> &offset) { offset_ = offset; }

  virtual void Feedforward(const CuMatrixBase &in, CuMatrix *out);
  void Propagate(const CuMatrixBase &in, CuMatrix *out);
  void Backpropagate(const CuMatrixBase &in, const CuMatrixBase &out, const CuMatrixBase &out_diff, CuMatrix *in_diff);

  static Component *Init(const std::string &conf_line);
  static Component *Read(std::istream &is, bool binary);
  void Write(std::ostream &os, bool binary) const;
  void WriteStandard(std::ostream &os, bool binary) const;

  virtual std::string Info() const { return ""; }
  virtual std::string InfoGradient() const { return ""; }

  // True if BackpropagateFnc writes every element of in_diff (so the wrapper need not zero it
  // first; the reference always zeroes, nnet-component.h:335).  All components here do.
  virtual bool BackpropOverwritesInDiff() const { return true; }
  // Pure-copy layers: the executor may pass the buffer through instead of copying it.
  virtual bool PropagateIsCopy() const { return false; }
  virtual bool BackpropIsCopy() const { return false; }
  // True if the parameter gradients are formed inside BackpropagateFnc (the recurrent and temporal
  // components, e.g. lc.h:976-1058) rather than in Update: the executor must then never skip the call.
  virtual bool GradientInBackprop() const { return false; }
  // true for components whose passes are long chains of tiny dependent kernels (the recurrences): the executor then keeps
  // everything on one stream -- a second active stream costs such a chain more than the overlap gives (measured on the
  // LC-BLSTM net: 7.85 ms/step on one stream, 8.61 with the output layer's weight gradient on a side stream)
  virtual bool LatencyBoundPasses() const { return false; }
  // true for a recurrent component whose last Propagate ran its recurrence as one persistent launch (csrc/rnn_persistent.hip)
  virtual bool PersistentRecurrence() const { return false; }
  // The executor calls this right before a Backpropagate that it follows with Update (Nnet::Backpropagate always does).
  // A component that forms its gradients inside Backpropagate may then take the SGD step there too (in the epilogue of
  // the gradient kernels) and treat the next Update call as a no-op; the default ignores the hint.
  virtual void FoldNextUpdateIntoBackprop() {}
  // Hint in front of a Backpropagate of a component that forms its gradients there (GradientInBackprop): the executor has work below this
  // component that does not depend on them, so they may be issued on the library's side stream (csrc/scratch.h SideStreamScope; the
  // executor joins it at the end of the pass).  One-shot, like the fold hint; the default ignores it.
  virtual void GradientsBesideLowerLayers() {}
  // The executor calls this right before a Backpropagate whose in-diff nobody will read (the component is fed by the network
  // input only and the caller asked for no input diff): a component that forms its gradients inside Backpropagate is still
  // called, but may leave the in-diff product out.  One-shot: taken back by the component once used; the default ignores it.
  virtual void InDiffUnusedInNextBackprop() {}

 protected:
  virtual void FeedforwardFnc(const CuMatrixBase &in, CuMatrixBase *out) { PropagateFnc(in, out); }
  virtual void PropagateFnc(const CuMatrixBase &in, CuMatrixBase *out) = 0;
  virtual void BackpropagateFnc(const CuMatrixBase &in, const CuMatrixBase &out, const CuMatrixBase &out_diff,
                                CuMatrixBase *in_diff) = 0;
  virtual void InitData(std::istream &is) {}
  virtual void ReadData(std::istream &is, bool binary) {}
  virtual void WriteData(std::ostream &os, bool binary) const {}

  int32 input_dim_, output_dim_, id_;
  std::string name_;
  std::vector<std::string> input_name_;
  std::vector<int32> input_, offset_;

 private:
  static Component *NewComponentOfType(ComponentType t, int32 input_dim, int32 output_dim);
};

class UpdatableComponent : public Component {
 public:
  UpdatableComponent(int32 input_dim, int32 output_dim) : Component(input_dim, output_dim) {}
  bool IsUpdatable() const { return true; }
  virtual int32 NumParams() const = 0;
  virtual void GetParams(std::vector<BaseFloat> *params) const = 0;
  // Sum of all parameters, for Nnet::Check()'s inf / nan test (nnet-nnet.cc:812-818 sums a host copy of every weight: 108 MB through
  // pageable memory for cfg2, 0.1 s inside the training tool's timer).  Components whose tensors are large add them up on the device.
  virtual double ParamSum() const {
    std::vector<BaseFloat> w;
    GetParams(&w);
    double s = 0.0;
    for (BaseFloat v : w) s += v;
    return s;
  }
  // (device pointer, number of floats incl. row padding) per tensor, in the reference's order
  virtual void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) = 0;
  // The GetGpuParams pointers went to a writer outside the component.  announces: that writer calls aslp_params_changed() after every write
  // (the native sync workers do); otherwise nothing derived from the parameters may be kept from step to step any more.
  virtual void ParamsAliased(bool announces) {}
  virtual void Update(const CuMatrixBase &input, const CuMatrixBase &diff) = 0;
  virtual void SetTrainOptions(const NnetTrainOptions &opts) { opts_ = opts; }
  const NnetTrainOptions &GetTrainOptions() const { return opts_; }
  virtual void InitData(std::istream &is) = 0;

 protected:
  NnetTrainOptions opts_;
};

// helpers shared by the component zoo
void AppendRowMajor(const CuMatrixBase &m, std::vector<BaseFloat> *out);  // CopyRowsFromMat
void AppendVector(const CuVectorBase &v, std::vector<BaseFloat> *out);
void InitMatParamUniform(CuMatrix &m, float scale);  // uniform [-scale, scale] (e.g. lc.h:77-81)
void InitVecParamUniform(CuVector &v, float scale);

}  // namespace aslp
