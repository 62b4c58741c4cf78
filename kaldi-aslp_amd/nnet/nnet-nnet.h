// nnet-nnet.h -- the Nnet graph executor of the host engine.
// Same public API as the reference's kaldi::aslp_nnet::Nnet (src/aslp-nnet/nnet-nnet.h:38-193).
#pragma once
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "nnet-component.h"

namespace aslp {

class Nnet {
 public:
  Nnet() {}
  Nnet(const Nnet &other);
  Nnet &operator=(const Nnet &other);
  ~Nnet();

  void Propagate(const CuMatrixBase &in, CuMatrix *out);
  void Propagate(const std::vector<const CuMatrixBase *> &in, std::vector<CuMatrix *> *out);
  void Backpropagate(const CuMatrixBase &out_diff, CuMatrix *in_diff);
  void Backpropagate(const std::vector<const CuMatrixBase *> &out_diff, std::vector<CuMatrix *> *in_diff);
  void Feedforward(const CuMatrixBase &in, CuMatrix *out);
  void Feedforward(const std::vector<const CuMatrixBase *> &in, std::vector<CuMatrix *> *out);
  void GetComponentTime();

  int32 InputDim() const;
  int32 OutputDim() const;
  int32 NumInput() const { return input_.size(); }
  int32 NumOutput() const { return output_.size(); }
  int32 NumComponents() const { return components_.size(); }
  const Component &GetComponent(int32 c) const;
  Component &GetComponent(int32 c);
  void SetComponent(int32 c, Component *component);
  void AppendComponent(Component *dynamically_allocated_comp);
  void AppendNnet(const Nnet &nnet_to_append);
  void RemoveComponent(int32 c);
  void RemoveLastComponent() { RemoveComponent(NumComponents() - 1); }

  // per-component forward outputs / backward output-diffs (reference: PropagateBuffer())
  const CuMatrixBase &OutputBuffer(int32 c) const;
  const CuMatrixBase &OutputDiffBuffer(int32 c) const;
  const CuMatrixBase &InputDiffBuffer(int32 c) const;

  int32 NumParams() const;
  void GetParams(std::vector<BaseFloat> *wei_copy) const;
  void GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params);
  // whoever holds the GetGpuParams pointers calls aslp_params_changed() after writing through them (the native sync workers)
  void ParamWritersAnnounce();
  void GetAccStats(std::vector<double *> *acc_params, std::vector<std::pair<double *, int>> *data_params);

  void ResetLstmStreams(const std::vector<int32> &stream_reset_flag);
  void SetSeqLengths(const std::vector<int32> &sequence_lengths);
  void SetChunkSize(int chunk_size);
  void SetDropoutRetention(BaseFloat r);  // nnet-nnet.cc:454-464

  void Init(const std::string &config_file);
  void InitFromString(const std::string &proto_text);  // same grammar, from memory
  void Read(const std::string &file);
  void Read(std::istream &in, bool binary);
  void Write(const std::string &file, bool binary) const;
  void Write(std::ostream &out, bool binary) const;
  void WriteStandard(const std::string &file, bool binary) const;
  void WriteStandard(std::ostream &out, bool binary) const;
  void WriteDotFile(std::ostream &out) const;

  std::string Info() const;
  std::string InfoGradient() const;
  std::string InfoPropagate() const;
  std::string InfoBackPropagate() const;
  void Check() const;
  void Destroy();

  void SetTrainOptions(const NnetTrainOptions &opts);
  const NnetTrainOptions &GetTrainOptions() const { return opts_; }
  void AutoComplete();
  void AssignComponentId(std::vector<Component *> &components);
  void SortComponent(std::vector<Component *> &components);

  // Engine switch (not in the reference): when true (default) a component whose only input is the
  // full output of its producer reads that buffer in place instead of "zero + AddMat" copying it
  // (nnet-nnet.cc:86-95), and its in-diff is written straight into the producer's out-diff
  // buffer when it is the producer's only consumer (nnet-nnet.cc:133-144).  Values are identical.
  void SetLinkAliasing(bool on) { alias_links_ = on; }
  // Engine switch (not in the reference): fold a Sigmoid whose only producer is a BatchNormalization (and which is
  // that component's only consumer) into the BatchNormalization kernels.  Values are identical; the intermediate
  // BN output / Sigmoid in-diff buffers are then not materialised (OutputBuffer() on them throws).
  void SetLayerFusion(bool on) { fuse_layers_ = on; }
  // Engine switch (not in the reference): issue AffineTransform::Update (weight-gradient GEMM with the SGD step in its
  // epilogue) on a side stream, ordered after the component's own Backpropagate, so that it shares the chip with the
  // backward pass of the layers below instead of sitting in its critical path; Backpropagate() returns with the main
  // stream waiting for all of them.  Values are identical (no kernel changes, no reduction order changes).  Not applied to
  // nets with recurrent components (Component::LatencyBoundPasses).
  void SetUpdateOverlap(bool on) { overlap_updates_ = on; }

  // Step-path entry points for callers that evaluate the loss on the device right away (the train-step C API): the
  // network output is read, and the loss's diff is written, in the executor's own buffers, which saves the two
  // full-matrix copies of Propagate(in, &out) / Backpropagate(diff, NULL).  With `fold_softmax` a final Softmax that
  // is the single network output is left to the loss kernel (LossInputIsPreSoftmax() tells whether that happened).
  void PropagateForLoss(const CuMatrixBase &in, bool fold_softmax);
  const CuMatrixBase &LossInput() const;
  bool LossInputIsPreSoftmax() const { return softmax_folded_; }
  CuMatrix *LossDiff(int32 num_frames);
  void BackpropagateFromLossDiff();

 private:
  void InitStream(std::istream &is);
  void InitInputOutput();
  bool IsDirectLink(int32 i) const;  // single input, offset 0, full width
  bool IsFinalSoftmax(int32 i) const;
  int32 FusedSigmoidOf(int32 i) const;  // index of the Sigmoid folded into BatchNormalization i, or -1
  int32 BatchNormOf(int32 i) const;      // index of the BatchNormalization that is AffineTransform i's only consumer (direct link), or -1
  int32 AffineSigmoidOf(int32 i) const;
  int32 AffineConsumerOf(int32 c) const;  // index of the AffineTransform that alone reads component c's output (direct link), or -1  // index of the Sigmoid whose forward pass rides in AffineTransform i's GEMM, or -1

  std::vector<Component *> components_;
  std::vector<int32> input_, output_;
  std::vector<int32> num_consumers_;
  std::vector<std::pair<std::string, double>> propagate_time_, back_propagate_time_;
  std::vector<CuMatrix> input_buf_, output_buf_, input_diff_buf_, output_diff_buf_;
  std::vector<CuVectorD> bn_stats_buf_;  // per AffineTransform in front of a BatchNormalization: column statistics of its output
  std::vector<const CuMatrixBase *> in_view_;       // what component i actually reads as input
  std::vector<const CuMatrixBase *> in_diff_view_;  // where component i's in-diff went
  std::vector<const CuMatrixBase *> out_view_;      // component i's forward output (own buffer, or its input for copy layers)
  std::vector<const CuMatrixBase *> out_diff_view_; // component i's out-diff (may have been handed on to the producer)
  NnetTrainOptions opts_;
  bool alias_links_ = true;
  bool fuse_layers_ = true;
  bool overlap_updates_ = true;
  bool fold_softmax_request_ = false, softmax_folded_ = false, diff_in_place_ = false;
  // weight updates issued on the side stream by the latest Backpropagate and not waited for yet (JoinUpdates(): the main stream waits);
  // the next Propagate joins in front of component first_after_updates_
  // (host_wait: the HOST waits -- for callers that free what the updates touch.)  The marker is an event on the side stream of the thread that
  // ran Backpropagate, so a Write / GetParams / Destroy from another host thread waits for the same work.
  void JoinUpdates(bool host_wait = false) const;
  mutable bool updates_pending_ = false;
  mutable void *updates_mark_ = nullptr;
  int32 first_after_updates_ = 0;
  // GetGpuParams handed the parameters' device pointers out and nobody promised aslp_params_changed() for them (the reference's own
  // workers: itf.h:26-42 has no such call): Backpropagate then returns with the main stream already waiting for the weight updates, so
  // that whatever the holder of the pointers enqueues behind it -- on that stream or through the null stream -- finds the step complete
  bool params_out_silently_ = false, silent_writer_logged_ = false;
  long next_bwd_epoch_ = 0;  // drawn by LossDiff() for the backward pass that will read the diff the loss is about to write
  long fwd_epoch_ = 0;  // csrc/split16.h: the forward pass whose buffers are still in place (operand planes made from them may be reused)
};

}  // namespace aslp
