// nnet-nnet.cpp -- graph executor.  Follows src/aslp-nnet/nnet-nnet.cc (cited per function).
#include "nnet-nnet.h"
#include "split16.h"

#include <cstdlib>
#include <chrono>
#include <fstream>

#include "nnet-basic.h"
#include "scratch.h"
#include "kaldi-io.h"
#include "nnet-recurrent.h"
#include "nnet-temporal.h"

namespace aslp {

Nnet::Nnet(const Nnet &other) { *this = other; }
Nnet &Nnet::operator=(const Nnet &other) {  // nnet-nnet.cc:41-65
  if (this == &other) return *this;
  Destroy();
  for (int32 i = 0; i < other.NumComponents(); i++) components_.push_back(other.GetComponent(i).Copy());
  SetTrainOptions(other.opts_);
  alias_links_ = other.alias_links_;
  fuse_layers_ = other.fuse_layers_;
  overlap_updates_ = other.overlap_updates_;
  InitInputOutput();
  Check();
  return *this;
}
Nnet::~Nnet() { Destroy(); }

bool Nnet::IsDirectLink(int32 i) const {
  const std::vector<int32> &in = components_[i]->GetInput(), &off = components_[i]->GetOffset();
  return in.size() == 1 && in[0] >= 0 && off[0] == 0 && components_[in[0]]->OutputDim() == components_[i]->InputDim();
}

int32 Nnet::FusedSigmoidOf(int32 i) const {
  if (!fuse_layers_ || !alias_links_ || components_[i]->GetType() != Component::kBatchNormalization || num_consumers_[i] != 1) return -1;
  for (int32 j = i + 1; j < NumComponents(); j++) {
    const std::vector<int32> &inp = components_[j]->GetInput();
    if (inp.size() == 1 && inp[0] == i)
      return (components_[j]->GetType() == Component::kSigmoid && IsDirectLink(j)) ? j : -1;
  }
  return -1;
}
int32 Nnet::BatchNormOf(int32 i) const {
  if (!fuse_layers_ || !alias_links_ || components_[i]->GetType() != Component::kAffineTransform || num_consumers_[i] != 1) return -1;
  for (int32 j = i + 1; j < NumComponents(); j++) {
    const std::vector<int32> &inp = components_[j]->GetInput();
    if (inp.size() == 1 && inp[0] == i)
      return (components_[j]->GetType() == Component::kBatchNormalization && IsDirectLink(j)) ? j : -1;
  }
  return -1;
}
int32 Nnet::AffineSigmoidOf(int32 i) const {
  if (!fuse_layers_ || !alias_links_ || components_[i]->GetType() != Component::kAffineTransform || num_consumers_[i] != 1) return -1;
  for (int32 j = i + 1; j < NumComponents(); j++) {
    const std::vector<int32> &inp = components_[j]->GetInput();
    if (inp.size() == 1 && inp[0] == i)
      return (components_[j]->GetType() == Component::kSigmoid && IsDirectLink(j)) ? j : -1;
  }
  return -1;
}
int32 Nnet::AffineConsumerOf(int32 c) const {
  if (!fuse_layers_ || !alias_links_ || num_consumers_[c] != 1) return -1;
  for (int32 j = c + 1; j < NumComponents(); j++) {
    const std::vector<int32> &inp = components_[j]->GetInput();
    if (inp.size() == 1 && inp[0] == c) return (components_[j]->GetType() == Component::kAffineTransform && IsDirectLink(j)) ? j : -1;
  }
  return -1;
}
const CuMatrixBase &Nnet::OutputBuffer(int32 c) const {
  if (FusedSigmoidOf(c) >= 0) ASLP_ERR << "output of component " << c << " is not materialised (fused into the Sigmoid behind it); SetLayerFusion(false)";
  if (softmax_folded_ && (IsFinalSoftmax(c) || c == output_[0]))
    ASLP_ERR << "output of component " << c << " is not materialised (the last step left the final Softmax to the loss kernel); SetLayerFusion(false)";
  return *out_view_[c];
}
const CuMatrixBase &Nnet::OutputDiffBuffer(int32 c) const {
  if (FusedSigmoidOf(c) >= 0) ASLP_ERR << "out-diff of component " << c << " is not materialised (fused into the Sigmoid behind it); SetLayerFusion(false)";
  return *out_diff_view_[c];
}

void Nnet::Propagate(const std::vector<const CuMatrixBase *> &in, std::vector<CuMatrix *> *out) {  // nnet-nnet.cc:70-106
  ASLP_ASSERT(in.size() == input_.size());
  int num_frame = in[0]->NumRows();
  // every buffer of the forward pass is rewritten from here on: operand planes of the previous pass are stale; the ones made
  // during this pass stay usable until the next forward pass (the backward pass reads the same buffers)
  fwd_epoch_ = s16_new_epoch();
  S16EpochScope plane_scope(fwd_epoch_, 0);
  for (size_t i = 0; i < input_.size(); i++) in_view_[input_[i]] = in[i];  // InputLayer reads the caller's matrix
  softmax_folded_ = false;
  std::vector<char> done(components_.size(), 0);  // Sigmoids already produced by the BatchNormalization in front of them
  for (int32 i = 0; i < (int32)components_.size(); i++) {
    if (done[i]) continue;
    // weight updates of the previous backward pass may still be running on the side stream (Backpropagate): the main stream waits for
    // them in front of the first component that touches what they read or write
    if (i >= first_after_updates_) JoinUpdates();
    if (components_[i]->GetType() != Component::kInputLayer) {
      const std::vector<int32> &input_idx = components_[i]->GetInput();
      const std::vector<int32> &offset = components_[i]->GetOffset();
      ASLP_ASSERT(input_idx.size() == offset.size());
      if (alias_links_ && IsDirectLink(i)) {
        in_view_[i] = out_view_[input_idx[0]];
      } else {  // :86-95 zeroed buffer, links added in (this is how branch splice / sum works)
        input_buf_[i].Resize(num_frame, components_[i]->InputDim(), kSetZero);
        for (size_t j = 0; j < input_idx.size(); j++) {
          int out_len = components_[input_idx[j]]->OutputDim();
          input_buf_[i].ColRange(offset[j], out_len).AddMat(1.0, *out_view_[input_idx[j]]);
        }
        in_view_[i] = &input_buf_[i];
      }
    }
    Timer tim1;
    const int32 fs = FusedSigmoidOf(i), as = AffineSigmoidOf(i);
    {
      // AffineTransform whose only consumer is a BatchNormalization: the GEMM's epilogue leaves the column statistics of its output
      // (32-row partial sums), and the normalisation then streams that output once instead of reading it for statistics first
      static const bool stats_off = getenv("ASLP_BN_FROM_STATS") != nullptr && getenv("ASLP_BN_FROM_STATS")[0] == '0';  // A/B switch
      const int32 bn = stats_off ? -1 : BatchNormOf(i);
      if (bn >= 0) {
        dynamic_cast<AffineTransform *>(components_[i])->RequestOutputStats(&bn_stats_buf_[i]);
        dynamic_cast<BatchNormalization *>(components_[bn])->UseInputStats(&bn_stats_buf_[i]);
      }
    }
    if (as >= 0) {  // AffineTransform + Sigmoid forward in one GEMM; both outputs exist, the backward passes stay separate
      const int32 ac = AffineConsumerOf(as);   // the layer product that reads the activations gets their fp16 planes from the same epilogue
      PlaneHolder *ap = (ac >= 0 && gemm_split16_serves(num_frame, components_[ac]->OutputDim(), components_[ac]->InputDim()))
                            ? &dynamic_cast<AffineTransform *>(components_[ac])->InputPlanes() : nullptr;
      dynamic_cast<AffineTransform *>(components_[i])->PropagateWithSigmoid(*in_view_[i], &output_buf_[i], &output_buf_[as], ap);
      out_view_[i] = &output_buf_[i];
      in_view_[as] = &output_buf_[i];
      out_view_[as] = &output_buf_[as];
      done[as] = 1;
    } else if (fs >= 0) {  // BatchNormalization + Sigmoid in one statistics pass and one write pass
      const int32 ac = AffineConsumerOf(fs);   // ... which also leaves the activations' fp16 planes for the layer product that reads them
      if (ac >= 0 && gemm_split16_serves(num_frame, components_[ac]->OutputDim(), components_[ac]->InputDim()))
        dynamic_cast<BatchNormalization *>(components_[i])->ProduceOutputPlanes(&dynamic_cast<AffineTransform *>(components_[ac])->InputPlanes());
      dynamic_cast<BatchNormalization *>(components_[i])->PropagateWithSigmoid(*in_view_[i], &output_buf_[fs]);
      out_view_[i] = &output_buf_[fs];
      out_view_[fs] = &output_buf_[fs];
      in_view_[fs] = &output_buf_[fs];  // never read: Sigmoid's backward only needs its output
      done[fs] = 1;
    } else if (alias_links_ && components_[i]->PropagateIsCopy() && components_[i]->GetType() != Component::kInputLayer) {
      out_view_[i] = in_view_[i];  // pure copy layer: pass the buffer through
    } else if (fold_softmax_request_ && IsFinalSoftmax(i)) {
      out_view_[i] = in_view_[i];  // the loss kernel forms the posteriors itself
      softmax_folded_ = true;
    } else {
      if (components_[i]->GetType() == Component::kInputLayer) {   // the copy of the input also makes the planes the first layer product reads
        const int32 ac = AffineConsumerOf(i);
        if (ac >= 0 && gemm_split16_serves(num_frame, components_[ac]->OutputDim(), components_[ac]->InputDim()))
          dynamic_cast<InputLayer *>(components_[i])->ProduceOutputPlanes(&dynamic_cast<AffineTransform *>(components_[ac])->InputPlanes());
      }
      components_[i]->Propagate(*in_view_[i], &output_buf_[i]);
      out_view_[i] = &output_buf_[i];
    }
    propagate_time_[i].first = Component::TypeToMarker(components_[i]->GetType());
    propagate_time_[i].second += tim1.Elapsed();
  }
  JoinUpdates();
  if (out != NULL)
    for (size_t i = 0; i < output_.size(); i++) *((*out)[i]) = *out_view_[output_[i]];
  // the caller's input may go away: Update() of the consumers reads InputLayer's OUTPUT copy
  for (size_t i = 0; i < input_.size(); i++) in_view_[input_[i]] = &output_buf_[input_[i]];
}

// Softmax whose posteriors are the single network output (directly, or through the auto-added OutputLayer)
bool Nnet::IsFinalSoftmax(int32 i) const {
  if (!fuse_layers_ || !alias_links_ || output_.size() != 1 || components_[i]->GetType() != Component::kSoftmax) return false;
  if (!aslp_softmax_xent_supported(components_[i]->OutputDim())) return false;
  const int32 o = output_[0];
  if (o == i) return num_consumers_[i] == 0;
  return num_consumers_[i] == 1 && num_consumers_[o] == 0 && components_[o]->GetType() == Component::kOutputLayer && IsDirectLink(o) &&
         components_[o]->GetInput()[0] == i;
}
void Nnet::SetDropoutRetention(BaseFloat r) {  // nnet-nnet.cc:454-464
  for (int32 c = 0; c < NumComponents(); c++) {
    if (GetComponent(c).GetType() == Component::kDropout) {
      Dropout &comp = dynamic_cast<Dropout &>(GetComponent(c));
      BaseFloat r_old = comp.GetDropoutRetention();
      comp.SetDropoutRetention(r);
      ASLP_LOG << "Setting dropout-retention in component " << c << " from " << r_old << " to " << r;
    }
  }
}
void Nnet::PropagateForLoss(const CuMatrixBase &in, bool fold_softmax) {
  ASLP_ASSERT(input_.size() == 1 && output_.size() == 1);
  std::vector<const CuMatrixBase *> in_vec(1, &in);
  fold_softmax_request_ = fold_softmax;
  try { Propagate(in_vec, NULL); } catch (...) { fold_softmax_request_ = false; throw; }
  fold_softmax_request_ = false;
}
const CuMatrixBase &Nnet::LossInput() const {
  ASLP_ASSERT(output_.size() == 1 && out_view_[output_[0]] != NULL);
  return *out_view_[output_[0]];
}
CuMatrix *Nnet::LossDiff(int32 num_frames) {
  ASLP_ASSERT(output_.size() == 1);
  CuMatrix *d = &output_diff_buf_[output_[0]];
  d->Resize(num_frames, components_[output_[0]]->OutputDim(), kUndefined);
  // Does this very buffer reach an AffineTransform as its out-diff (through identity backward passes that hand the buffer on:
  // OutputLayer, the final Softmax)?  Then the loss may leave the diff's fp16 planes with that layer (csrc/split16.h).
  S16DiffTarget &t = s16_loss_diff_target();
  t = S16DiffTarget();
  next_bwd_epoch_ = 0;
  if (fuse_layers_ && alias_links_) {
    int32 c = output_[0];
    while (c >= 0 && components_[c]->GetType() != Component::kAffineTransform) {
      const bool hands_on = components_[c]->BackpropIsCopy() && components_[c]->GetType() != Component::kInputLayer && IsDirectLink(c) &&
                            num_consumers_[components_[c]->GetInput()[0]] == 1;
      c = hands_on ? components_[c]->GetInput()[0] : -1;
    }
    if (c >= 0 && (c == output_[0] || num_consumers_[c] == 1)) {
      AffineTransform *at = dynamic_cast<AffineTransform *>(components_[c]);
      if (gemm_split16_serves(at->OutputDim(), at->InputDim(), num_frames) && d->NumCols() == at->OutputDim()) {
        next_bwd_epoch_ = s16_new_epoch();
        t.planes = &at->DiffPlanes().get();
        t.epoch = next_bwd_epoch_;
        t.diff = d->Data();
      }
    }
  }
  return d;
}
void Nnet::BackpropagateFromLossDiff() {
  ASLP_ASSERT(output_.size() == 1);
  std::vector<const CuMatrixBase *> od(1, &output_diff_buf_[output_[0]]);
  diff_in_place_ = true;
  try { Backpropagate(od, NULL); } catch (...) { diff_in_place_ = false; throw; }
  diff_in_place_ = false;
}

void Nnet::Backpropagate(const std::vector<const CuMatrixBase *> &out_diff, std::vector<CuMatrix *> *in_diff) {  // :108-154
  ASLP_ASSERT(out_diff.size() == output_.size());
  JoinUpdates();
  int num_frame = out_diff[0]->NumRows();
  const int32 N = NumComponents();
  // the forward buffers as Propagate left them; every diff buffer new (the loss's diff may already carry planes under the epoch
  // LossDiff drew for this pass)
  const long bwd_epoch = (diff_in_place_ && next_bwd_epoch_ != 0) ? next_bwd_epoch_ : s16_new_epoch();
  next_bwd_epoch_ = 0;
  s16_loss_diff_target() = S16DiffTarget();
  S16EpochScope plane_scope(fwd_epoch_, bwd_epoch);
  // which producers receive their out-diff by accumulation (need a zeroed buffer, :112-114)
  std::vector<char> direct(N, 0);
  for (int32 i = 0; i < N; i++) {
    if (components_[i]->GetType() == Component::kInputLayer) continue;
    if (alias_links_ && IsDirectLink(i) && num_consumers_[components_[i]->GetInput()[0]] == 1) direct[i] = 1;
  }
  for (int32 i = 0; i < N; i++) {
    bool fed_direct = false;
    for (int32 c = i + 1; c < N && !fed_direct; c++)
      if (direct[c] && components_[c]->GetInput()[0] == i) fed_direct = true;
    if (diff_in_place_ && i == output_[0]) continue;  // already holds the loss's diff
    if (!fed_direct) output_diff_buf_[i].Resize(num_frame, components_[i]->OutputDim(), kSetZero);
  }
  if (!diff_in_place_)
    for (size_t i = 0; i < output_.size(); i++) output_diff_buf_[output_[i]].CopyFromMat(*(out_diff[i]));
  for (int32 i = 0; i < N; i++) out_diff_view_[i] = &output_diff_buf_[i];
  const bool want_in_diff = (in_diff != NULL);
  bool overlap_updates = overlap_updates_;
  // Recurrent layers whose passes are launches per timestep keep everything on one stream (Component::LatencyBoundPasses).  Persistent
  // recurrences (one launch per pass) do not: the weight gradients of the layer above then run on the side stream BESIDE the recurrence
  // below, whose workgroups leave most of every CU idle (measured on cfg3: 3.00 -> 2.88 ms per step; the recurrence itself stretches
  // from 190 to ~300 us while 165 us of GEMMs share its CUs, so a third of the gradient time is hidden).  A/B: ASLP_LSTM_SIDE_GRADS=0.
  static const bool lstm_side_off = getenv("ASLP_LSTM_SIDE_GRADS") != nullptr && getenv("ASLP_LSTM_SIDE_GRADS")[0] == '0';
  bool recurrent_net = false;
  for (int32 i = 0; i < N; i++) {
    if (components_[i]->LatencyBoundPasses()) overlap_updates = false;
    if (components_[i]->PersistentRecurrence()) recurrent_net = true;
  }
  if (recurrent_net && lstm_side_off) overlap_updates = false;
  std::vector<int32> fused_sigmoid(N, -1);   // BN index -> its folded Sigmoid
  std::vector<char> folded(N, 0);            // Sigmoids handled by their BatchNormalization
  for (int32 i = 0; i < N; i++) {
    fused_sigmoid[i] = FusedSigmoidOf(i);
    if (fused_sigmoid[i] >= 0) folded[fused_sigmoid[i]] = 1;
  }
  int32 lowest_updatable = -1;
  for (int32 i = 0; i < N && lowest_updatable < 0; i++)
    if (components_[i]->IsUpdatable()) lowest_updatable = i;
  static const bool lowest_on_side = getenv("ASLP_LOWEST_UPDATE_ON_SIDE") != nullptr && getenv("ASLP_LOWEST_UPDATE_ON_SIDE")[0] == '1';  // A/B switch
  if (lowest_on_side) lowest_updatable = -1;
  // How many of the LOWEST updatable components keep their update on the main stream.  The side stream starts late (behind the output layer's
  // in-diff product) with the largest update and every update carries an HBM-bound epilogue that does not shrink with the minibatch, so it
  // ends behind the main stream and the next forward pass waits for it: at minibatch 256 the five-hidden-layer net's main stream stood idle
  // for ~55 of 408 us.  With the lowest TWO AffineTransforms' updates on the main stream the two streams end together there: 0.410 -> 0.394 ms
  // per step at minibatch 256; at minibatch 512 and 1024 one is the better choice (0.547 against 0.552, 0.742 against 0.745 ms; with
  // BatchNormalization at 1024: 0.746 against 0.777), and so it is for nets with fewer than four AffineTransforms and for recurrent nets.
  // A/B: ASLP_UPDATES_ON_MAIN=n.
  static const int on_main_env = [] { const char *e = getenv("ASLP_UPDATES_ON_MAIN"); return e ? atoi(e) : -1; }();
  int32 main_below = lowest_updatable;   // updates of updatable components with index <= main_below stay on the main stream
  {
    auto is_affine = [&](int32 i) { return components_[i]->GetType() == Component::kAffineTransform && components_[i]->IsUpdatable(); };
    int affines = 0;
    for (int32 i = 0; i < N; i++) affines += is_affine(i) ? 1 : 0;
    const int on_main_n = on_main_env >= 0 ? on_main_env : ((affines >= 4 && !recurrent_net && num_frame <= 384) ? 2 : 1);
    int seen = 0;
    for (int32 i = 0; i < N && seen < on_main_n; i++)
      if (is_affine(i)) { main_below = i; seen++; }
    if (lowest_updatable < 0 || on_main_n == 0) main_below = -1;
    else if (main_below < lowest_updatable) main_below = lowest_updatable;
  }
  for (int32 i = N - 1; i >= 0; i--) {
    if (folded[i]) { in_diff_view_[i] = NULL; continue; }
    Timer tim2;
    const bool is_input = components_[i]->GetType() == Component::kInputLayer;
    // The reference also back-propagates into the network input (:124-125); that in-diff is
    // observable only through `in_diff`, so it is skipped when the caller passes NULL.
    bool feeds_only_input = !is_input;
    if (!is_input)
      for (int32 p : components_[i]->GetInput())
        if (components_[p]->GetType() != Component::kInputLayer) feeds_only_input = false;
    const bool skip_backprop = !want_in_diff && (is_input || feeds_only_input) && !components_[i]->GradientInBackprop();
    CuMatrix *target = &input_diff_buf_[i];
    if (!is_input && direct[i]) target = &output_diff_buf_[components_[i]->GetInput()[0]];
    if (!skip_backprop) {
      if (!is_input && direct[i] && components_[i]->BackpropIsCopy()) {
        // identity backward into the only consumer slot: hand the buffer over instead of copying
        target->Swap(&output_diff_buf_[i]);
        out_diff_view_[i] = target;
      } else if (fused_sigmoid[i] >= 0) {
        const int32 fs = fused_sigmoid[i];
        if (direct[i] && components_[components_[i]->GetInput()[0]]->GetType() == Component::kAffineTransform) {
          // the in-diff is the out-diff of the AffineTransform in front: its conversion takes the scale from maxima this launch leaves
          AffineTransform *at = dynamic_cast<AffineTransform *>(components_[components_[i]->GetInput()[0]]);
          if (gemm_split16_serves(at->OutputDim(), at->InputDim(), num_frame))
            dynamic_cast<BatchNormalization *>(components_[i])->LeaveDiffMaxima(&at->DiffPlanes());
        }
        dynamic_cast<BatchNormalization *>(components_[i])->FoldNextUpdateIntoBackprop();
        dynamic_cast<BatchNormalization *>(components_[i])->BackpropagateWithSigmoid(*in_view_[i], output_buf_[fs], output_diff_buf_[fs], target);
      } else {
        // AffineTransform <- Sigmoid <- AffineTransform (a sigmoid layer without BatchNormalization): the upper product leaves the maxima
        // of its in-diff, from which the Sigmoid's backward pass knows the scale of the lower layer's out-diff planes before it writes them
        if (fuse_layers_ && alias_links_ && !is_input && direct[i]) {
          const int32 below = components_[i]->GetInput()[0];
          const int32 sig = components_[i]->GetType() == Component::kAffineTransform ? below : i;
          if (components_[sig]->GetType() == Component::kSigmoid && !folded[sig] && direct[sig]) {
            const int32 lower = components_[sig]->GetInput()[0];
            if (components_[lower]->GetType() == Component::kAffineTransform) {
              AffineTransform *lo = dynamic_cast<AffineTransform *>(components_[lower]);
              if (gemm_split16_serves(lo->OutputDim(), lo->InputDim(), num_frame)) {
                if (sig == i) dynamic_cast<Sigmoid *>(components_[i])->ProduceInDiffPlanes(&lo->DiffPlanes());
                else dynamic_cast<AffineTransform *>(components_[i])->LeaveInDiffMaxima(&lo->DiffPlanes());
              }
            }
          }
        }
        if (fuse_layers_) components_[i]->FoldNextUpdateIntoBackprop();
        if (overlap_updates && components_[i]->PersistentRecurrence() && i != lowest_updatable) components_[i]->GradientsBesideLowerLayers();
        if (!want_in_diff && (is_input || feeds_only_input)) components_[i]->InDiffUnusedInNextBackprop();  // here for its gradients only
        components_[i]->Backpropagate(*in_view_[i], *out_view_[i], output_diff_buf_[i], target);
      }
      in_diff_view_[i] = target;
    }
    if (components_[i]->IsUpdatable()) {
      UpdatableComponent *uc = dynamic_cast<UpdatableComponent *>(components_[i]);
      // (the lowest updatable component has nothing below it to run beside: on the main stream it spares the step boundary a
      //  cross-stream event wait, ~10 us before the next forward pass can start)
      if (overlap_updates && components_[i]->GetType() == Component::kAffineTransform && i > main_below) {
        SideStreamScope side;  // after this component's Backpropagate (which reads the weights), beside everything below it
        uc->Update(*in_view_[i], output_diff_buf_[i]);
      } else {
        uc->Update(*in_view_[i], output_diff_buf_[i]);
      }
    }
    back_propagate_time_[i].first = Component::TypeToMarker(components_[i]->GetType());
    back_propagate_time_[i].second += tim2.Elapsed();
    if (!is_input && !direct[i] && !skip_backprop) {  // :133-144 scatter-add to the producers
      const std::vector<int32> &input_idx = components_[i]->GetInput();
      const std::vector<int32> &offset = components_[i]->GetOffset();
      for (size_t j = 0; j < input_idx.size(); j++) {
        ASLP_ASSERT(input_idx[j] >= 0 && input_idx[j] <= NumComponents());
        int out_len = components_[input_idx[j]]->OutputDim();
        output_diff_buf_[input_idx[j]].AddMat(1.0, input_diff_buf_[i].ColRange(offset[j], out_len));
      }
    }
  }
  // The main stream has to wait for the side stream's updates before anything reads the weights -- or overwrites what the updates still
  // read: the forward buffers and planes that are the weight-gradient products' operands.  Feed-forward nets put that wait off into the
  // next forward pass: everything in front of the first component that produces an operand of a side-stream update (or is such a
  // component itself) runs beside the last updates instead of behind them -- the input copy, its conversion and the lowest layer's product,
  // 34 of the 40 us for which the main stream used to sit idle at the end of a cfg2 step.  Every other way to the parameters (the
  // accessors below, the C API, aslp_params_changed() of the sync workers) joins first.  A/B: ASLP_LATE_JOIN=0.
  static const bool late_join_off = getenv("ASLP_LATE_JOIN") != nullptr && getenv("ASLP_LATE_JOIN")[0] == '0';
  first_after_updates_ = 0;
  if (params_out_silently_ && !silent_writer_logged_) {
    silent_writer_logged_ = true;
    ASLP_LOG << "parameters are aliased through GetGpuParams by a writer that does not announce its writes: every training step ends "
                "joined with its weight updates and the weights' fp16 planes are made anew in every step "
                "(ParamWritersAnnounce() + aslp_params_changed() after each write lift both)";
  }
  if (overlap_updates && !recurrent_net && !late_join_off && !params_out_silently_ && NULL == in_diff) {
    int32 first = N;
    for (int32 i = 0; i < N; i++) {
      if (!(components_[i]->GetType() == Component::kAffineTransform && components_[i]->IsUpdatable() && i > main_below)) continue;
      first = std::min(first, i);
      for (int32 p : components_[i]->GetInput()) first = std::min(first, p);   // its input buffer (and the planes made of it) is an operand
    }
    // a component folded into its producer's launch is written by that launch
    for (int32 i = 0; i < first; i++)
      if (fused_sigmoid[i] >= first || AffineSigmoidOf(i) >= first) { first = i; i = -1; }
    first_after_updates_ = first;
    const int marked = side_stream_mark(&updates_mark_);
    updates_pending_ = marked > 0;
    if (marked < 0) join_side_stream();   // no marker to wait for later: the join is made here (the error is set)
  } else {
    join_side_stream();
  }
  if (NULL == in_diff) return;
  for (size_t i = 0; i < input_.size(); i++)
    if ((*in_diff)[i] != NULL) *((*in_diff)[i]) = input_diff_buf_[input_[i]];
}

void Nnet::JoinUpdates(bool host_wait) const {
  if (!updates_pending_) return;
  updates_pending_ = false;
  side_stream_mark_wait(updates_mark_, host_wait);   // (whichever host thread this is)
  join_side_stream();                                // the issuing thread's own bookkeeping; a no-op elsewhere
}

void Nnet::Feedforward(const std::vector<const CuMatrixBase *> &in, std::vector<CuMatrix *> *out) {  // :156-189
  JoinUpdates();
  ASLP_ASSERT(NULL != out);
  ASLP_ASSERT(in.size() == input_.size());
  int num_frame = in[0]->NumRows();
  fwd_epoch_ = s16_new_epoch();   // (the forward buffers are rewritten: see Propagate)
  S16EpochScope plane_scope(fwd_epoch_, 0);
  for (size_t i = 0; i < input_.size(); i++) in_view_[input_[i]] = in[i];
  for (int32 i = 0; i < (int32)components_.size(); i++) {
    if (components_[i]->GetType() != Component::kInputLayer) {
      const std::vector<int32> &input_idx = components_[i]->GetInput();
      const std::vector<int32> &offset = components_[i]->GetOffset();
      if (alias_links_ && IsDirectLink(i)) {
        in_view_[i] = out_view_[input_idx[0]];
      } else {
        input_buf_[i].Resize(num_frame, components_[i]->InputDim(), kSetZero);
        for (size_t j = 0; j < input_idx.size(); j++) {
          int out_len = components_[input_idx[j]]->OutputDim();
          input_buf_[i].ColRange(offset[j], out_len).AddMat(1.0, *out_view_[input_idx[j]]);
        }
        in_view_[i] = &input_buf_[i];
      }
    }
    if (alias_links_ && components_[i]->PropagateIsCopy() && components_[i]->GetType() != Component::kInputLayer) {
      out_view_[i] = in_view_[i];
    } else {
      components_[i]->Feedforward(*in_view_[i], &output_buf_[i]);
      out_view_[i] = &output_buf_[i];
    }
  }
  for (size_t i = 0; i < output_.size(); i++) *((*out)[i]) = *out_view_[output_[i]];
  for (size_t i = 0; i < input_.size(); i++) in_view_[input_[i]] = &output_buf_[input_[i]];
}

void Nnet::Propagate(const CuMatrixBase &in, CuMatrix *out) {  // :191-204
  ASLP_ASSERT(NULL != out);
  if (NumComponents() == 0) { (*out) = in; return; }
  ASLP_ASSERT(input_.size() == 1);
  ASLP_ASSERT(output_.size() == 1);
  std::vector<const CuMatrixBase *> in_vec(1, &in);
  std::vector<CuMatrix *> out_vec(1, out);
  Propagate(in_vec, &out_vec);
}
void Nnet::Backpropagate(const CuMatrixBase &out_diff, CuMatrix *in_diff) {  // :206-216
  if (NumComponents() == 0) { (*in_diff) = out_diff; return; }
  ASLP_ASSERT(input_.size() == 1);
  ASLP_ASSERT(output_.size() == 1);
  std::vector<const CuMatrixBase *> od(1, &out_diff);
  if (in_diff == NULL) { Backpropagate(od, NULL); return; }
  std::vector<CuMatrix *> id(1, in_diff);
  Backpropagate(od, &id);
}
void Nnet::Feedforward(const CuMatrixBase &in, CuMatrix *out) {  // :218-232
  ASLP_ASSERT(NULL != out);
  if (NumComponents() == 0) { out->Resize(in.NumRows(), in.NumCols()); out->CopyFromMat(in); return; }
  ASLP_ASSERT(input_.size() == 1);
  ASLP_ASSERT(output_.size() == 1);
  std::vector<const CuMatrixBase *> in_vec(1, &in);
  std::vector<CuMatrix *> out_vec(1, out);
  Feedforward(in_vec, &out_vec);
}

const CuMatrixBase &Nnet::InputDiffBuffer(int32 c) const {
  ASLP_ASSERT(in_diff_view_[c] != NULL);
  return *in_diff_view_[c];
}

int32 Nnet::OutputDim() const { ASLP_ASSERT(!components_.empty()); return components_.back()->OutputDim(); }
int32 Nnet::InputDim() const { ASLP_ASSERT(!components_.empty()); return components_.front()->InputDim(); }
const Component &Nnet::GetComponent(int32 c) const { ASLP_ASSERT(static_cast<size_t>(c) < components_.size()); JoinUpdates(); return *(components_[c]); }
Component &Nnet::GetComponent(int32 c) { ASLP_ASSERT(static_cast<size_t>(c) < components_.size()); JoinUpdates(); return *(components_[c]); }

void Nnet::SetComponent(int32 c, Component *component) {
  JoinUpdates();
  ASLP_ASSERT(static_cast<size_t>(c) < components_.size());
  delete components_[c];
  components_[c] = component;
  InitInputOutput();
  Check();
}
void Nnet::AppendComponent(Component *comp) {  // :262-272
  JoinUpdates();
  components_.push_back(comp);
  for (int32 i = 0; i < (int32)components_.size(); i++) {
    components_[i]->SetId(i);
    components_[i]->SetMonoInput(i - 1);
  }
  InitInputOutput();
}
void Nnet::AppendNnet(const Nnet &other) {
  for (int32 i = 0; i < other.NumComponents(); i++) AppendComponent(other.GetComponent(i).Copy());
  InitInputOutput();
  Check();
}
void Nnet::RemoveComponent(int32 c) {
  JoinUpdates();
  ASLP_ASSERT(c < NumComponents());
  Component *ptr = components_[c];
  components_.erase(components_.begin() + c);
  delete ptr;
  InitInputOutput();
  Check();
}

void Nnet::GetParams(std::vector<BaseFloat> *wei_copy) const {  // :296-311
  JoinUpdates();
  wei_copy->clear();
  for (size_t i = 0; i < components_.size(); i++)
    if (components_[i]->IsUpdatable()) {
      std::vector<BaseFloat> c_params;
      dynamic_cast<UpdatableComponent &>(*components_[i]).GetParams(&c_params);
      wei_copy->insert(wei_copy->end(), c_params.begin(), c_params.end());
    }
  ASLP_ASSERT((int32)wei_copy->size() == NumParams());
}
void Nnet::GetGpuParams(std::vector<std::pair<BaseFloat *, int>> *params) {  // :314-325
  JoinUpdates();
  ASLP_ASSERT(params != NULL);
  params->clear();
  for (size_t i = 0; i < components_.size(); i++)
    if (components_[i]->IsUpdatable()) {
      std::vector<std::pair<BaseFloat *, int>> c_params;
      dynamic_cast<UpdatableComponent &>(*components_[i]).GetGpuParams(&c_params);
      dynamic_cast<UpdatableComponent &>(*components_[i]).ParamsAliased(false);
      params->insert(params->end(), c_params.begin(), c_params.end());
    }
  params_out_silently_ = true;   // until ParamWritersAnnounce()
  aslp_params_changed();   // (the caller may already hold pointers from an earlier call)
}
void Nnet::ParamWritersAnnounce() {
  for (size_t i = 0; i < components_.size(); i++)
    if (components_[i]->IsUpdatable()) dynamic_cast<UpdatableComponent &>(*components_[i]).ParamsAliased(true);
  params_out_silently_ = false;
}
void Nnet::GetAccStats(std::vector<double *> *acc_params, std::vector<std::pair<double *, int>> *data_params) {  // :327-342
  JoinUpdates();
  ASLP_ASSERT(acc_params != NULL && data_params != NULL);
  acc_params->clear();
  data_params->clear();
  for (size_t i = 0; i < components_.size(); i++)
    if (GetComponent(i).GetType() == Component::kBatchNormalization) {
      BatchNormalization &bn = dynamic_cast<BatchNormalization &>(GetComponent(i));
      std::vector<std::pair<double *, int>> c_params;
      double *acc_ptr = bn.GetAccStats(&c_params);
      acc_params->push_back(acc_ptr);
      data_params->insert(data_params->end(), c_params.begin(), c_params.end());
    }
}
int32 Nnet::NumParams() const {
  int32 n = 0;
  for (size_t i = 0; i < components_.size(); i++)
    if (components_[i]->IsUpdatable()) n += dynamic_cast<UpdatableComponent *>(components_[i])->NumParams();
  return n;
}

void Nnet::ResetLstmStreams(const std::vector<int32> &flags) {  // :473-496
  for (int32 c = 0; c < NumComponents(); c++) {
    RecurrentBase *r = dynamic_cast<RecurrentBase *>(components_[c]);
    if (r && r->HasStreamReset()) r->ResetLstmStreams(flags);
  }
}
void Nnet::SetSeqLengths(const std::vector<int32> &lens) {  // :498-530
  for (int32 c = 0; c < NumComponents(); c++) {
    if (RecurrentBase *r = dynamic_cast<RecurrentBase *>(components_[c])) { if (r->HasSeqLengths()) r->SetSeqLengths(lens); continue; }
    if (RowConvolution *rc = dynamic_cast<RowConvolution *>(components_[c])) rc->SetSeqLengths(lens);
  }
  (void)lens;
}
void Nnet::SetChunkSize(int chunk_size) {  // :532-539
  for (int32 c = 0; c < NumComponents(); c++)
    if (BLstmProjectedStreamsLC *l = dynamic_cast<BLstmProjectedStreamsLC *>(components_[c])) l->SetChunkSize(chunk_size);
}

void Nnet::AutoComplete() {
  // a plain chain of components (a prototype without <Name> / <Input>) becomes a graph: an InputLayer in front, an OutputLayer behind,
  // node k reading node k - 1 (what nnet-nnet.cc:541-568 produces)
  ASLP_ASSERT(!components_.empty());
  const int32 in_dim = components_.front()->InputDim(), out_dim = components_.back()->OutputDim();
  for (Component *c : components_) ASLP_ASSERT(c->Id() < 0);   // nobody has numbered them yet
  components_.insert(components_.begin(), new InputLayer(in_dim, in_dim));
  components_.push_back(new OutputLayer(out_dim, out_dim));
  for (size_t k = 0; k < components_.size(); k++) {
    components_[k]->SetId((int32)k);
    components_[k]->SetMonoInput((int32)k - 1);   // -1: the network input
  }
}

void Nnet::InitStream(std::istream &is) {  // :570-612
  std::string conf_line, token;
  bool simple_net = true;
  while (!is.eof()) {
    ASLP_ASSERT(is.good());
    std::getline(is, conf_line);
    if (conf_line == "") continue;
    ASLP_VLOG(1) << conf_line;
    std::istringstream ls(conf_line);
    ls >> std::ws >> token;
    if (ls.fail()) continue;  // whitespace-only line
    if (token == "<NnetProto>" || token == "</NnetProto>") continue;
    if (token[0] == '#') continue;  // comment line (extension; the reference would raise "Unknown marker")
    if (token == "<StructureType>") {
      ls >> std::ws >> token;
      if (token == "graph") simple_net = false;
      else if (token == "simple") simple_net = true;
      else ASLP_ERR << "The net's structure must be simple or graphi!";
      continue;
    }
    components_.push_back(Component::Init(conf_line + "\n"));
  }
  if (!simple_net) {
    AssignComponentId(components_);
    SortComponent(components_);
  }
  if (simple_net) AutoComplete();
  InitInputOutput();
  Check();
}
void Nnet::Init(const std::string &file) {
  Input in(file);  // extended filename: file, "-", "cmd |"
  InitStream(in.Stream());
}
void Nnet::InitFromString(const std::string &proto) {
  std::istringstream in(proto);
  InitStream(in);
}

void Nnet::Read(const std::string &file) {  // :615-625
  bool binary;
  Input in(file, &binary);
  Read(in.Stream(), binary);
  if (NumComponents() == 0) ASLP_WARN << "The network '" << file << "' is empty.";
}
void Nnet::Read(std::istream &is, bool binary) {  // :628-646
  Component *comp;
  while (NULL != (comp = Component::Read(is, binary))) {
    int id = comp->Id();
    if (id >= (int)components_.size()) components_.resize(id + 1, NULL);
    if (components_[id] != NULL) ASLP_ERR << "Component id " << id << " already be taken" << "the id must be unique";
    components_[id] = comp;
  }
  opts_.learn_rate = 0.0;  // :643 reset learn rate
  InitInputOutput();
  Check();
}
void Nnet::Write(const std::string &file, bool binary) const {
  Output out(file, binary, true);
  Write(out.Stream(), binary);
  if (!out.Close()) ASLP_ERR << "Error closing output stream " << PrintableWxfilename(file);
}
void Nnet::Write(std::ostream &os, bool binary) const {  // :654-663
  JoinUpdates();
  Check();
  WriteToken(os, binary, "<Nnet>");
  if (binary == false) os << std::endl;
  for (int32 i = 0; i < NumComponents(); i++) components_[i]->Write(os, binary);
  WriteToken(os, binary, "</Nnet>");
  if (binary == false) os << std::endl;
}
void Nnet::WriteStandard(const std::string &file, bool binary) const {  // :695-699 (sic: the reference calls Write here)
  Output out(file, binary, true);
  Write(out.Stream(), binary);
  if (!out.Close()) ASLP_ERR << "Error closing output stream " << PrintableWxfilename(file);
}
void Nnet::WriteStandard(std::ostream &os, bool binary) const {  // :701-712
  JoinUpdates();
  Check();
  WriteToken(os, binary, "<Nnet>");
  if (binary == false) os << std::endl;
  for (int32 i = 0; i < NumComponents(); i++) {
    if (components_[i]->GetType() == Component::kInputLayer || components_[i]->GetType() == Component::kOutputLayer) continue;
    components_[i]->WriteStandard(os, binary);
  }
  WriteToken(os, binary, "</Nnet>");
  if (binary == false) os << std::endl;
}
void Nnet::WriteDotFile(std::ostream &os) const {  // :665-693
  os << "digraph net{" << std::endl;
  os << "rankdir=BT" << std::endl;
  os << "node[shape = box; height = 1; width = 3; fontsize = 40];" << std::endl;
  os << "edge[minlen = 1 ]" << std::endl;
  for (int32 i = 0; i < NumComponents(); i++) {
    std::string name = components_[i]->GetName();
    int id = components_[i]->Id();
    const std::vector<int32> &input = components_[i]->GetInput(), &offset = components_[i]->GetOffset();
    if (name != "") os << id << " [label = " << "\"" << name << "\"]" << std::endl;
    else os << id << " [label = " << "\"" << Component::TypeToMarker(components_[i]->GetType()) << "\"]" << std::endl;
    if (input.size() == 1 && input[0] == -1) continue;
    for (size_t j = 0; j < input.size(); j++)
      os << "\t" << input[j] << " -> " << id << " [label = " << offset[j] << "; fontsize = 40]" << std::endl;
  }
  os << "}" << std::endl;
}

std::string Nnet::Info() const {  // :714-737
  JoinUpdates();
  std::ostringstream ostr;
  ostr << "num-components " << NumComponents() << std::endl;
  ostr << "input-dim " << InputDim() << std::endl;
  ostr << "output-dim " << OutputDim() << std::endl;
  ostr << "number-of-parameters " << static_cast<float>(NumParams()) / 1e6 << " millions" << std::endl;
  for (int32 i = 0; i < NumComponents(); i++) {
    ostr << "component " << i + 1 << " : " << Component::TypeToMarker(components_[i]->GetType()) << ", input-dim "
         << components_[i]->InputDim() << ", output-dim " << components_[i]->OutputDim() << ", id " << components_[i]->Id();
    const std::vector<int32> &input_idx = components_[i]->GetInput(), &offset = components_[i]->GetOffset();
    ostr << ", input ";
    for (size_t j = 0; j < input_idx.size(); j++) ostr << input_idx[j] << ":" << offset[j] << ",";
    ostr << "  " << components_[i]->Info() << std::endl;
  }
  return ostr.str();
}
std::string Nnet::InfoGradient() const {
  JoinUpdates();
  std::ostringstream ostr;
  ostr << "### Gradient stats :\n";
  for (int32 i = 0; i < NumComponents(); i++)
    ostr << "Component " << i + 1 << " : " << Component::TypeToMarker(components_[i]->GetType()) << ", "
         << components_[i]->InfoGradient() << std::endl;
  return ostr.str();
}
std::string Nnet::InfoPropagate() const {
  std::ostringstream ostr;
  ostr << "### Forward propagation buffer content :\n";
  ostr << "[0] output of <Input> " << MomentStatistics(*in_view_[0]) << std::endl;
  for (int32 i = 0; i < NumComponents(); i++)
    ostr << "[" << 1 + i << "] output of " << Component::TypeToMarker(components_[i]->GetType()) << MomentStatistics(*out_view_[i]) << std::endl;
  return ostr.str();
}
std::string Nnet::InfoBackPropagate() const {
  std::ostringstream ostr;
  ostr << "### Backward propagation buffer content :\n";
  ostr << "[0] diff of <Input> " << MomentStatistics(output_diff_buf_[0]) << std::endl;
  for (int32 i = 0; i < NumComponents(); i++)
    ostr << "[" << 1 + i << "] diff-output of " << Component::TypeToMarker(components_[i]->GetType()) << MomentStatistics(*out_diff_view_[i]) << std::endl;
  return ostr.str();
}

void Nnet::Check() const {  // :776-819
  JoinUpdates();
  if (input_.size() < 1) ASLP_ERR << "Must have at least one InputLayer";
  if (output_.size() < 1) ASLP_ERR << "Must have at least one OutputLayer";
  for (int i = 0; i < NumComponents(); i++) {
    if (components_[i] == NULL) ASLP_ERR << "Component id must be consistant, but have no id " << i;
    if (components_[i]->Id() != i) ASLP_ERR << "Component id not equal index id, May be error in Read";
  }
  for (int i = 0; i < NumComponents(); i++) {
    if (components_[i]->GetType() == Component::kInputLayer) continue;
    const std::vector<int32> &input_idx = components_[i]->GetInput(), &offset = components_[i]->GetOffset();
    ASLP_ASSERT(input_idx.size() == offset.size());
    for (size_t j = 0; j < input_idx.size(); j++) {
      int idx = input_idx[j];
      if (idx < 0 || idx >= NumComponents()) ASLP_ERR << "Component " << i << " has an invalid input id " << idx;
      if (components_[idx]->Id() >= components_[i]->Id())
        ASLP_ERR << "Input id must be less than Component id, case " << " <Id> " << i << " <Input> " << idx;
      int32 out_dim = components_[idx]->OutputDim();
      if (offset[j] + out_dim > components_[i]->InputDim())
        ASLP_ERR << "Component " << idx << " outputdim + offset must be less than " << "offset " << offset[j] << " " << "outdim "
                 << out_dim << " " << "Component " << i << " inputdim";
    }
  }
  double sum = 0.0;
  for (int i = 0; i < NumComponents(); i++)
    if (components_[i]->IsUpdatable()) sum += dynamic_cast<const UpdatableComponent *>(components_[i])->ParamSum();
  if (std::isinf(sum)) ASLP_ERR << "'inf' in network parameters (weight explosion, try lower learning rate?)";
  if (std::isnan(sum)) ASLP_ERR << "'nan' in network parameters (try lower learning rate?)";
}

void Nnet::Destroy() {  // :822-832
  JoinUpdates(true);
  side_stream_mark_free(updates_mark_);
  updates_mark_ = nullptr;
  params_out_silently_ = false;   // (pointers handed out earlier die with the components)
  for (int32 i = 0; i < NumComponents(); i++) delete components_[i];
  components_.resize(0);
  input_buf_.resize(0);
  input_diff_buf_.resize(0);
  output_buf_.resize(0);
  output_diff_buf_.resize(0);
  bn_stats_buf_.resize(0);
  in_view_.resize(0);
  in_diff_view_.resize(0);
  out_view_.resize(0);
  out_diff_view_.resize(0);
}

void Nnet::SetTrainOptions(const NnetTrainOptions &opts) {
  JoinUpdates();
  opts_ = opts;
  for (int32 l = 0; l < NumComponents(); l++)
    if (GetComponent(l).IsUpdatable()) dynamic_cast<UpdatableComponent &>(GetComponent(l)).SetTrainOptions(opts_);
}

void Nnet::InitInputOutput() {  // :845-870
  input_.clear();
  output_.clear();
  for (int i = 0; i < NumComponents(); i++) {
    if (components_[i] == NULL) continue;
    if (components_[i]->GetType() == Component::kInputLayer) input_.push_back(components_[i]->Id());
    else if (components_[i]->GetType() == Component::kOutputLayer) output_.push_back(components_[i]->Id());
  }
  input_buf_.resize(NumComponents());
  output_buf_.resize(NumComponents());
  input_diff_buf_.resize(NumComponents());
  output_diff_buf_.resize(NumComponents());
  bn_stats_buf_.resize(NumComponents());
  in_view_.assign(NumComponents(), NULL);
  in_diff_view_.assign(NumComponents(), NULL);
  out_view_.assign(NumComponents(), NULL);
  out_diff_view_.assign(NumComponents(), NULL);
  for (int i = 0; i < NumComponents(); i++) { out_view_[i] = &output_buf_[i]; out_diff_view_[i] = &output_diff_buf_[i]; }
  num_consumers_.assign(NumComponents(), 0);
  for (int i = 0; i < NumComponents(); i++) {
    if (components_[i] == NULL || components_[i]->GetType() == Component::kInputLayer) continue;
    for (int32 p : components_[i]->GetInput())
      if (p >= 0 && p < NumComponents()) num_consumers_[p]++;
  }
  propagate_time_.assign(NumComponents(), std::make_pair(std::string(), 0.0));
  back_propagate_time_.assign(NumComponents(), std::make_pair(std::string(), 0.0));
}

void Nnet::GetComponentTime() {  // :872-884
  for (size_t i = 0; i < propagate_time_.size(); i++) {
    ASLP_LOG << propagate_time_[i].first << ": Propagate time " << propagate_time_[i].second << "s, " << "Back-Propagate time "
             << back_propagate_time_[i].second << "s, " << "total time " << propagate_time_[i].second + back_propagate_time_[i].second << "s";
    propagate_time_[i].second = 0.0;
    back_propagate_time_[i].second = 0.0;
  }
}

void Nnet::AssignComponentId(std::vector<Component *> &comp) {
  // Ids are positions in a topological order of the named graph, found by taking ready nodes from a STACK (the node that became ready
  // last is numbered next; ready nodes of one step are stacked in prototype order) -- the order nnet-nnet.cc:886-949 assigns, which the
  // stored <Input> vectors of existing models rely on.  Then every <Input> name is replaced by its producer's id ("-1" = network input).
  const int32 n = (int32)comp.size();
  std::map<std::string, int32> index_of;
  for (int32 i = 0; i < n; i++)
    if (!index_of.insert(std::make_pair(comp[i]->GetName(), i)).second) ASLP_ERR << "Two components are named " << comp[i]->GetName();
  std::vector<std::vector<int32>> readers(n);   // producer -> the nodes that read it, one entry per link, in prototype order
  std::vector<int32> waiting(n, 0);             // links of a node whose producer has no id yet
  for (int32 i = 0; i < n; i++) {
    const std::vector<std::string> &in = comp[i]->GetInputName();
    if (in.size() == 1 && in[0] == "-1") continue;   // reads the network input only
    for (const std::string &name : in) {
      if (name == comp[i]->GetName()) ASLP_ERR << "The input of component " << comp[i]->GetName() << "include itself, Please check it!";
      waiting[i]++;
      const auto it = index_of.find(name);
      if (it != index_of.end()) readers[it->second].push_back(i);   // (a link to a name nobody has keeps its node waiting: reported as a cycle below)
    }
  }
  std::vector<int32> ready;
  for (int32 i = 0; i < n; i++)
    if (waiting[i] == 0) ready.push_back(i);
  int32 next_id = 0;
  while (!ready.empty()) {
    const int32 u = ready.back();
    ready.pop_back();
    comp[u]->SetId(next_id++);
    std::vector<int32> now_ready;
    for (int32 r : readers[u])
      if (--waiting[r] == 0) now_ready.push_back(r);
    std::sort(now_ready.begin(), now_ready.end());   // prototype order
    ready.insert(ready.end(), now_ready.begin(), now_ready.end());
  }
  if (next_id != n) ASLP_ERR << "The graph has a cycle";
  for (int32 i = 0; i < n; i++) {
    const std::vector<std::string> &in = comp[i]->GetInputName();
    std::vector<int32> ids(in.size(), 0);
    for (size_t j = 0; j < in.size(); j++) {
      if (in[j] == "-1") { ids[j] = -1; continue; }
      const auto it = index_of.find(in[j]);
      if (it != index_of.end()) ids[j] = comp[it->second]->GetId();
    }
    comp[i]->SetInput(ids);
  }
}
void Nnet::SortComponent(std::vector<Component *> &comp) {  // :951-958
  std::vector<Component *> tmp(comp.size());
  for (size_t i = 0; i < comp.size(); i++) tmp[comp[i]->GetId()] = comp[i];
  comp.swap(tmp);
}

}  // namespace aslp
