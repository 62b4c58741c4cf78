// data-reader.h -- FrameDataReader (src/aslp-nnet/data-reader.{h,cc}:12-195): features from a sequential table, targets
// looked up by utterance key, both pushed through frame-level randomizers sharing ONE shuffle mask per cache fill, handed
// out minibatch by minibatch.  Utterances without targets are skipped with the reference's warning; a length mismatch is
// an error.  The feature cache lives in HBM (MatrixRandomizer), the Posterior cache on the host.
#pragma once
#include <string>
#include <vector>

#include "kaldi-table.h"
#include "nnet-component.h"
#include "nnet-randomizer.h"
#include "parse-options.h"

namespace aslp {

inline void RegisterRandomizerOptions(NnetDataRandomizerOptions *o, OptionsItf *opts) {  // nnet-randomizer.h:43-47
  opts->Register("randomizer-size", &o->randomizer_size,
                 "Capacity of randomizer, length of concatenated utterances which are used for frame-level shuffling (in frames, affects memory consumption, max 8000000).");
  opts->Register("randomizer-seed", &o->randomizer_seed, "Seed value for srand, sets fixed order of frame-level shuffling");
  opts->Register("minibatch-size", &o->minibatch_size, "Size of a minibatch.");
}
inline void RegisterTrainOptions(NnetTrainOptions *o, OptionsItf *opts) {  // nnet-trnopts.h:42-47
  opts->Register("learn-rate", &o->learn_rate, "Learning rate");
  opts->Register("momentum", &o->momentum, "Momentum");
  opts->Register("l2-penalty", &o->l2_penalty, "L2 penalty (weight decay)");
  opts->Register("l1-penalty", &o->l1_penalty, "L1 penalty (promote sparsity)");
}

typedef StdVectorRandomizer<std::vector<std::pair<int32, BaseFloat>>> PosteriorRandomizer;

class FrameDataReader {
 public:
  FrameDataReader(const std::string &feature_rspecifier, const std::string &targets_rspecifier, const NnetDataRandomizerOptions &rand_opts,
                  bool randomize = true)
      : feature_reader_(feature_rspecifier), targets_reader_(targets_rspecifier), feature_randomizer_(rand_opts),
        targets_randomizer_(rand_opts), rand_opts_(rand_opts), randomize_(randomize), read_done_(false), num_no_tgt_(0), num_done_(0) {
    randomizer_mask_.Init(rand_opts_);
  }
  bool Done() { return read_done_ && feature_randomizer_.Done(); }
  // false: the remaining frames do not fill a minibatch (the reference drops them, data-reader.cc:176-190)
  bool ReadData(const CuMatrixBase **feat, const Posterior **targets) {
    ASLP_ASSERT(feat != NULL && targets != NULL);
    if (Done()) ASLP_ERR << "Already read done";
    if (feature_randomizer_.Done()) FillRandomizer();
    if (Done()) return false;
    *feat = &feature_randomizer_.Value();
    feature_randomizer_.Next();
    tgt_ = targets_randomizer_.Value();
    *targets = &tgt_;
    targets_randomizer_.Next();
    return true;
  }
  int32 NumUtterances() const { return num_done_; }
  int32 NumMissingTargets() const { return num_no_tgt_; }

 private:
  void FillRandomizer() {  // data-reader.cc:66-128
    while (true) {
      if (feature_randomizer_.IsFull()) break;
      if (feature_reader_.Done()) { read_done_ = true; break; }
      const std::string utt = feature_reader_.Key();
      ASLP_VLOG(3) << "Reading " << utt;
      if (!targets_reader_.HasKey(utt)) {
        ASLP_WARN << utt << ", missing targets";
        num_no_tgt_++;
      } else {
        const HostMatrix &mat = feature_reader_.Value();
        const Posterior &targets = targets_reader_.Value(utt);
        if ((int32)targets.size() != mat.rows) ASLP_ERR << "feature and target dim must match";
        cu_mat_ = mat;
        feature_randomizer_.AddData(cu_mat_);
        targets_randomizer_.AddData(targets);
        num_done_++;
      }
      feature_reader_.Next();
    }
    // the reference always shuffles here (its --randomize flag is only echoed in the log, aslp-nnet-train-frame.cc:41,136);
    // `randomize_ == false` keeps the frame order (identity mask) for the tools that expose a working switch
    const int32 n = feature_randomizer_.NumFrames();
    if (n == 0) return;
    if (randomize_) {
      const std::vector<int32> &mask = randomizer_mask_.Generate(n);
      feature_randomizer_.Randomize(mask);
      targets_randomizer_.Randomize(mask);
    }
  }
  SequentialBaseFloatMatrixReader feature_reader_;
  RandomAccessPosteriorReader targets_reader_;
  RandomizerMask randomizer_mask_;
  MatrixRandomizer feature_randomizer_;
  PosteriorRandomizer targets_randomizer_;
  NnetDataRandomizerOptions rand_opts_;
  bool randomize_, read_done_;
  int32 num_no_tgt_, num_done_;
  CuMatrix cu_mat_;
  Posterior tgt_;
};

}  // namespace aslp
