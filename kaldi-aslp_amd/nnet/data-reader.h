// data-reader.h -- FrameDataReader (src/aslp-nnet/data-reader.{h,cc}:12-195): features from a sequential table, targets
// looked up by utterance key, both pushed through frame-level randomizers sharing ONE shuffle mask per cache fill, handed
// out minibatch by minibatch.  Utterances without targets are skipped with the reference's warning; a length mismatch is
// an error.  The feature cache lives in HBM (MatrixRandomizer), the Posterior cache on the host.
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cu-device.h"
#include "kaldi-table.h"
#include "nnet-component.h"
#include "nnet-nnet.h"
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

inline void NnetTrainOptions::Register(OptionsItf *opts) { RegisterTrainOptions(this, opts); }              // the reference's spelling
inline void NnetDataRandomizerOptions::Register(OptionsItf *opts) { RegisterRandomizerOptions(this, opts); }

typedef StdVectorRandomizer<std::vector<std::pair<int32, BaseFloat>>> PosteriorRandomizer;

// A background thread parses the two tables (file / pipe reads, Posterior parsing, the copy of each feature matrix into
// page-locked memory) while the main thread trains; the main thread's share of a cache fill is one async H2D copy per
// utterance plus the shuffle.  The utterance order, the cache arithmetic and the shuffle masks are exactly those of the
// single-threaded reference loop.
class FrameDataReader {
 public:
  FrameDataReader(const std::string &feature_rspecifier, const std::string &targets_rspecifier, const NnetDataRandomizerOptions &rand_opts,
                  bool randomize = true)
      : feature_rspecifier_(feature_rspecifier), targets_rspecifier_(targets_rspecifier), feature_randomizer_(rand_opts),
        targets_randomizer_(rand_opts), rand_opts_(rand_opts), randomize_(randomize), read_done_(false), num_no_tgt_(0), num_done_(0),
        stop_(false), queued_frames_(0) {
    randomizer_mask_.Init(rand_opts_);
    // open the tables here so that a bad rspecifier fails in the constructor, like the reference's readers
    feature_reader_.reset(new SequentialBaseFloatMatrixReader(feature_rspecifier_));
    targets_reader_.reset(new RandomAccessPosteriorReader(targets_rspecifier_));
    producer_ = std::thread(&FrameDataReader::Produce, this);
  }
  ~FrameDataReader() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (producer_.joinable()) producer_.join();
    for (auto &it : in_flight_) { it.second->Wait(); PinnedFree(it.first.ptr); }
    for (auto &blk : free_blocks_) PinnedFree(blk.ptr);
    for (auto &item : queue_) PinnedFree(item.block.ptr);
  }
  bool Done() { return read_done_ && feature_randomizer_.Done(); }
  // false: the remaining frames do not fill a minibatch (the reference drops them, data-reader.cc:176-190)
  bool ReadData(const CuMatrixBase **feat, const Posterior **targets) {
    ASLP_ASSERT(feat != NULL && targets != NULL);
    if (Done()) ASLP_ERR << "Already read done";
    if (feature_randomizer_.Done()) FillRandomizer();
    if (Done()) {
      // The reference leaves *feat / *targets alone here, and they point at its randomizers' own minibatch copies (nnet-randomizer.cc:93-98):
      // a caller that does not look at the result -- the reference's worker mains, aslp-nnet-train-frame-worker.cc:147 -- runs one more step
      // on the LAST minibatch.  The engine's Value() is a view into the cache, which the fill above may have rearranged, so the last minibatch
      // was set aside when the fill began and the pointers are moved onto that copy: the same values the reference's caller sees.
      if (have_last_) { *feat = &last_feat_copy_; *targets = &last_tgt_copy_; }
      return false;
    }
    *feat = &feature_randomizer_.Value();
    feature_randomizer_.Next();
    cur_begin_ = targets_randomizer_.Begin();
    *targets = &targets_randomizer_.Value();
    targets_randomizer_.Next();
    last_feat_ = *feat;
    last_tgt_ = *targets;
    if (!read_done_) Prefetch(false);
    return true;
  }
  // The minibatch ReadData() handed out last, as device-resident labels: when every target frame of the current cache is ONE pdf of weight 1
  // (alignments through ali-to-post: the normal case) the labels of the whole cache were uploaded once at the refill, and a training loop
  // can hand `*labels_dev` (one int32 per frame of the minibatch) to Xent::EvalLabelsPreSoftmax with unit frame weights instead of sending
  // labels and weights to the device in every step -- the same kernel on the same numbers.  *max_label: the largest label in the cache
  // (the caller checks it against the net's output width).  false: soft or weighted targets in this cache; use the Posterior.
  bool MinibatchLabels(const int32 **labels_dev, int32 *max_label) const {
    if (!cache_labels_ok_) return false;
    *labels_dev = cache_labels_.Data() + cur_begin_;
    *max_label = cache_label_max_;
    return true;
  }
  int32 NumUtterances() const { return num_done_; }
  int32 NumMissingTargets() const { return num_no_tgt_; }

 private:
  struct Block { float *ptr = nullptr; size_t cap = 0; };
  struct Item {
    std::string key, error;
    bool end = false, has_targets = false;
    int32 rows = 0, cols = 0;
    Block block;
    Posterior targets;
  };

  // ---- producer thread ------------------------------------------------------------------------------------
  Block TakeBlock(size_t floats) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t i = 0; i < free_blocks_.size(); i++)
        if (free_blocks_[i].cap >= floats) { Block b = free_blocks_[i]; free_blocks_.erase(free_blocks_.begin() + i); return b; }
    }
    Block b;
    b.cap = (floats + (1u << 18) - 1) & ~(size_t)((1u << 18) - 1);  // 1 MiB granules: a handful of sizes, recycled
    b.ptr = static_cast<float *>(PinnedAlloc(sizeof(float) * b.cap));
    return b;
  }
  void Push(Item &&item) {
    std::unique_lock<std::mutex> lk(mu_);
    // bounded look-ahead: about two caches worth of frames
    cv_.wait(lk, [&] { return stop_ || queued_frames_ <= 2 * (int64_t)rand_opts_.randomizer_size; });
    if (stop_) { PinnedFree(item.block.ptr); return; }
    queued_frames_ += item.rows;
    queue_.push_back(std::move(item));
    lk.unlock();
    cv_.notify_all();
  }
  // the feature table's binary float matrices are read straight into page-locked blocks (host-matrix.h HostMatrixSink)
  static float *SinkTake(void *ctx, int rows, int cols) {
    FrameDataReader *self = static_cast<FrameDataReader *>(ctx);
    self->sunk_ = self->TakeBlock((size_t)rows * cols);
    self->sunk_valid_ = true;
    return self->sunk_.ptr;
  }
  void Produce() {
    struct SinkScope {   // (this thread reads nothing but the feature table's matrices)
      explicit SinkScope(FrameDataReader *r) { host_matrix_sink().take = &FrameDataReader::SinkTake; host_matrix_sink().ctx = r; }
      ~SinkScope() { host_matrix_sink() = HostMatrixSink(); }
    } sink_scope(this);
    // ASLP_READER_PROFILE=1: where this thread's time goes (seconds in the table reader's Next(), the targets' look-up + copy, Push)
    static const bool prof = getenv("ASLP_READER_PROFILE") != nullptr && getenv("ASLP_READER_PROFILE")[0] == '1';
    double t_next = 0.0, t_tgt = 0.0, t_push = 0.0;
    Timer tp;
    struct Report {
      const bool &on; const double &a, &b, &c;
      ~Report() { if (on) ASLP_LOG << "FrameDataReader thread: feature table " << a << " s, targets " << b << " s, hand-over " << c << " s"; }
    } report{prof, t_next, t_tgt, t_push};
    try {
      CuDevice::Instantiate().BindThread();
      for (; !feature_reader_->Done(); tp.Reset(), feature_reader_->Next(), t_next += tp.Elapsed()) {
        {
          std::lock_guard<std::mutex> lk(mu_);
          if (stop_) return;
        }
        Item item;
        item.key = feature_reader_->Key();
        ASLP_VLOG(3) << "Reading " << item.key;
        const HostMatrix &mat = feature_reader_->Value();   // (loaded by the table reader's Next(): on this thread from the second object on)
        item.rows = mat.rows;
        item.cols = mat.cols;
        tp.Reset();
        item.has_targets = targets_reader_->HasKey(item.key);
        const bool sunk = sunk_valid_ && mat.data.empty();   // (else: another encoding, or the table reader had the object before the sink was there)
        if (item.has_targets) {
          item.targets = targets_reader_->Value(item.key);
          t_tgt += tp.Elapsed();
          if (sunk) {
            item.block = sunk_;
          } else {
            item.block = TakeBlock(mat.data.size());
            std::memcpy(item.block.ptr, mat.data.data(), sizeof(float) * mat.data.size());
          }
        } else if (sunk) {
          std::lock_guard<std::mutex> lk(mu_);
          free_blocks_.push_back(sunk_);
        }
        sunk_valid_ = false;
        tp.Reset();
        Push(std::move(item));
        t_push += tp.Elapsed();
      }
    } catch (const std::exception &e) {
      Item item;
      item.error = e.what();
      Push(std::move(item));
      return;
    }
    Item last;
    last.end = true;
    Push(std::move(last));
  }

  // ---- consumer ---------------------------------------------------------------------------------------------
  void RecycleBlocks() {
    for (size_t i = 0; i < in_flight_.size();) {
      if (in_flight_[i].second->Done()) {
        std::lock_guard<std::mutex> lk(mu_);
        free_blocks_.push_back(in_flight_[i].first);
        in_flight_.erase(in_flight_.begin() + i);
      } else {
        i++;
      }
    }
  }
  // Moves parsed utterances from the reader thread's queue into the STAGED next cache (uploads on the randomizer's copy
  // lane, nnet-randomizer.h) until that cache is full or the table ends.  Non-blocking calls (one per ReadData) take what
  // has arrived, so by the time the current cache runs out the next one is already in HBM; the blocking call finishes it.
  // The loop body is the reference's, data-reader.cc:72-118, with "add to the randomizer" replaced by "add to the stage".
  void Prefetch(bool block) {
    if (stage_closed_) return;
    if (!feature_randomizer_.Staging()) {
      feature_randomizer_.StageBegin();
      stage_added_ = 0;
      stage_end_seen_ = false;
    }
    RecycleBlocks();
    while (true) {
      if (feature_randomizer_.StageFull()) { stage_closed_ = true; break; }
      Item item;
      {
        Timer tw;
        std::unique_lock<std::mutex> lk(mu_);
        if (!block && queue_.empty()) return;
        cv_.wait(lk, [&] { return !queue_.empty(); });
        item = std::move(queue_.front());
        queue_.pop_front();
        queued_frames_ -= item.rows;
        lk.unlock();
        cv_.notify_all();
        if (block) t_wait_ += tw.Elapsed();
      }
      if (!item.error.empty()) throw std::runtime_error(item.error);
      if (item.end) { stage_end_seen_ = true; stage_closed_ = true; break; }
      if (!item.has_targets) {
        ASLP_WARN << item.key << ", missing targets";
        num_no_tgt_++;
        continue;
      }
      if ((int32)item.targets.size() != item.rows) ASLP_ERR << "feature and target dim must match";
      std::unique_ptr<StreamMarker> marker(new StreamMarker);
      feature_randomizer_.StageAddPinned(item.block.ptr, item.rows, item.cols, marker.get());
      in_flight_.push_back(std::make_pair(item.block, std::move(marker)));
      stage_targets_.push_back(std::move(item.targets));
      num_done_++;
      stage_added_++;
    }
  }
  void FillRandomizer() {  // data-reader.cc:66-128
    Timer fill_timer;
    static const bool prof = getenv("ASLP_READER_PROFILE") != nullptr && getenv("ASLP_READER_PROFILE")[0] == '1';
    Timer tp;
    t_wait_ = 0.0;
    if (last_feat_ != nullptr) {   // (once per cache, not per step: one minibatch of rows device to device and its targets; see ReadData)
      last_feat_copy_ = *last_feat_;
      last_tgt_copy_ = *last_tgt_;
      have_last_ = true;
    }
    Prefetch(true);
    const double t_prefetch = tp.Elapsed();
    tp.Reset();
    feature_randomizer_.StageCommit();
    for (auto &t : stage_targets_) targets_randomizer_.AddData(std::move(t));
    stage_targets_.clear();
    const double t_commit = tp.Elapsed();
    tp.Reset();
    const int32 added = stage_added_;
    read_done_ = stage_end_seen_;
    stage_closed_ = false;
    CuDevice::Instantiate().AccuProfile("FrameDataReader: waiting for the reader thread", t_wait_);
    // the reference always shuffles here (its --randomize flag is only echoed in the log, aslp-nnet-train-frame.cc:41,136);
    // `randomize_ == false` keeps the frame order (identity mask) for the tools that expose a working switch
    const int32 n = feature_randomizer_.NumFrames();
    // Nothing new arrived (the previous fill stopped on "cache full" exactly at the last utterance): what is left is less
    // than a minibatch and is dropped.  The reference shuffles here regardless and dies on its own data_begin_ == 0 check.
    cache_labels_ok_ = false;
    if (n == 0 || added == 0) return;
    ASLP_ASSERT(n == targets_randomizer_.NumFrames());
    double t_mask = 0.0, t_feat = 0.0;
    if (randomize_) {
      Timer ts;
      const std::vector<int32> &mask = randomizer_mask_.Generate(n);
      t_mask = ts.Elapsed();
      ts.Reset();
      feature_randomizer_.Randomize(mask);
      t_feat = ts.Elapsed();
      targets_randomizer_.Randomize(mask);
    }
    {   // one label per frame and unit weights throughout the (shuffled) cache?  then the labels go up once
      const auto &frames = targets_randomizer_.Cache();
      host_labels_.resize(n);
      int32 mx = -1;
      bool one_hot = true;
      for (int32 i = 0; i < n && one_hot; i++) {
        const auto &fr = frames[i];
        if (fr.size() != 1 || fr[0].second != 1.0f || fr[0].first < 0) one_hot = false;
        else { host_labels_[i] = fr[0].first; mx = std::max(mx, fr[0].first); }
      }
      if (one_hot) {
        cache_labels_ = host_labels_;
        cache_label_max_ = mx;
        cache_labels_ok_ = true;
      }
    }
    CuDevice::Instantiate().AccuProfile("FrameDataReader::FillRandomizer (host, total)", fill_timer.Elapsed());
    const double t_shuffle = tp.Elapsed();
    tp.Reset();
    if (!read_done_) Prefetch(false);  // opens the next stage: whatever the reader thread already has goes up behind the shuffle
    if (prof) ASLP_LOG << "FillRandomizer: finish the stage " << t_prefetch * 1e3 << " ms (waited " << t_wait_ * 1e3 << "), commit + targets " << t_commit * 1e3
                       << ", shuffle " << t_shuffle * 1e3 << " (mask " << t_mask * 1e3 << ", features " << t_feat * 1e3 << "), open the next stage " << tp.Elapsed() * 1e3 << " ms; frames " << n;
  }

  std::string feature_rspecifier_, targets_rspecifier_;
  std::unique_ptr<SequentialBaseFloatMatrixReader> feature_reader_;  // used by the producer thread only
  std::unique_ptr<RandomAccessPosteriorReader> targets_reader_;      // used by the producer thread only
  RandomizerMask randomizer_mask_;
  MatrixRandomizer feature_randomizer_;
  PosteriorRandomizer targets_randomizer_;
  NnetDataRandomizerOptions rand_opts_;
  bool randomize_, read_done_;
  int32 num_no_tgt_, num_done_;
  CuArray<int32> cache_labels_;       // labels of the whole current cache (MinibatchLabels)
  std::vector<int32> host_labels_;
  bool cache_labels_ok_ = false;
  int32 cache_label_max_ = -1, cur_begin_ = 0;
  const CuMatrixBase *last_feat_ = nullptr;   // what ReadData handed out last, and the copy of it made when the next fill begins
  const Posterior *last_tgt_ = nullptr;
  CuMatrix last_feat_copy_;
  Posterior last_tgt_copy_;
  bool have_last_ = false;
  Block sunk_;                 // the block the sink handed out for the object the table reader holds now (producer thread only)
  bool sunk_valid_ = false;
  std::thread producer_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_;
  int64_t queued_frames_;
  std::deque<Item> queue_;
  std::vector<Block> free_blocks_;
  std::vector<std::pair<Block, std::unique_ptr<StreamMarker>>> in_flight_;
  // the cache being staged
  bool stage_closed_ = false, stage_end_seen_ = false;
  int32 stage_added_ = 0;
  double t_wait_ = 0.0;
  std::vector<Posterior> stage_targets_;
};

// ---- multi-input / multi-output FrameDataReader (data-reader.cc:12-150, the vector constructor) --------------------
// N feature tables read in lock step (same keys in the same order, same number of frames), M target tables looked up by
// key; every stream has its own randomizer, all shuffled with ONE mask per cache fill.  Single-threaded like the reference.
class MimoFrameDataReader {
 public:
  MimoFrameDataReader(const std::vector<std::string> &feature_rspecifiers, const std::vector<std::string> &targets_rspecifiers,
                      const NnetDataRandomizerOptions &rand_opts)
      : num_input_(feature_rspecifiers.size()), num_output_(targets_rspecifiers.size()), rand_opts_(rand_opts), read_done_(false) {
    ASLP_ASSERT(num_input_ > 0 && num_output_ > 0);
    for (int i = 0; i < num_input_; i++) {
      feature_readers_.emplace_back(new SequentialBaseFloatMatrixReader(feature_rspecifiers[i]));
      feature_randomizers_.emplace_back(new MatrixRandomizer(rand_opts_));
    }
    for (int i = 0; i < num_output_; i++) {
      targets_readers_.emplace_back(new RandomAccessPosteriorReader(targets_rspecifiers[i]));
      targets_randomizers_.emplace_back(new PosteriorRandomizer(rand_opts_));
    }
    randomizer_mask_.Init(rand_opts_);
  }
  bool Done() { return read_done_ && feature_randomizers_[0]->Done(); }
  // false when what is left does not fill a minibatch (the reference hands out a short read here and trips its own checks)
  bool ReadData(std::vector<const CuMatrixBase *> *input, std::vector<const Posterior *> *output) {
    ASLP_ASSERT(input != NULL && output != NULL);
    input->resize(num_input_);
    output->resize(num_output_);
    if (Done()) ASLP_ERR << "Already read done";
    if (feature_randomizers_[0]->Done()) FillRandomizer();
    if (Done()) return false;
    for (int i = 0; i < num_input_; i++) { (*input)[i] = &feature_randomizers_[i]->Value(); feature_randomizers_[i]->Next(); }
    for (int i = 0; i < num_output_; i++) { (*output)[i] = &targets_randomizers_[i]->Value(); targets_randomizers_[i]->Next(); }
    return true;
  }

 private:
  void FillRandomizer() {
    int32 added = 0;
    while (true) {
      if (feature_randomizers_[0]->IsFull()) break;
      if (feature_readers_[0]->Done()) {
        for (int i = 1; i < num_input_; i++) ASLP_ASSERT(feature_readers_[i]->Done());
        read_done_ = true;
        break;
      }
      const std::string utt = feature_readers_[0]->Key();
      for (int i = 1; i < num_input_; i++)
        if (utt != feature_readers_[i]->Key())
          ASLP_ERR << "all feature not in the same order" << "[0] " << utt << "[" << i << "] " << feature_readers_[i]->Key();
      bool all_have_target = true;
      for (int i = 0; i < num_output_; i++)
        if (!targets_readers_[i]->HasKey(utt)) { ASLP_WARN << utt << ", missing targets"; all_have_target = false; }
      if (all_have_target) {
        int32 num_frame = 0;
        for (int i = 0; i < num_input_; i++) {
          const HostMatrix &mat = feature_readers_[i]->Value();
          if (i == 0) num_frame = mat.rows;
          else if (mat.rows != num_frame) ASLP_ERR << "all feature dim not equal";
          cu_mat_ = mat;
          feature_randomizers_[i]->AddData(cu_mat_);
        }
        for (int i = 0; i < num_output_; i++) {
          const Posterior &targets = targets_readers_[i]->Value(utt);
          if ((int32)targets.size() != num_frame) ASLP_ERR << "feature and target dim must match";
          targets_randomizers_[i]->AddData(targets);
        }
        added++;
      }
      for (int i = 0; i < num_input_; i++) feature_readers_[i]->Next();
    }
    if (feature_randomizers_[0]->NumFrames() == 0 || added == 0) return;
    const std::vector<int32> &mask = randomizer_mask_.Generate(feature_randomizers_[0]->NumFrames());
    for (auto &r : feature_randomizers_) r->Randomize(mask);
    for (auto &r : targets_randomizers_) r->Randomize(mask);
  }
  int num_input_, num_output_;
  NnetDataRandomizerOptions rand_opts_;
  bool read_done_;
  std::vector<std::unique_ptr<SequentialBaseFloatMatrixReader>> feature_readers_;
  std::vector<std::unique_ptr<RandomAccessPosteriorReader>> targets_readers_;
  std::vector<std::unique_ptr<MatrixRandomizer>> feature_randomizers_;
  std::vector<std::unique_ptr<PosteriorRandomizer>> targets_randomizers_;
  RandomizerMask randomizer_mask_;
  CuMatrix cu_mat_;
};

// ---- SequenceDataReader (data-reader.h:48-100, data-reader.cc:178-340) -------------------------------------------
// num-stream utterances advance in parallel, batch-size frames per call, rows t*S + s; features are shifted by
// targets-delay frames against the targets (the last frame repeats), an exhausted stream is padded with its last frame /
// target under a zero mask and takes the next utterance (flag in GetNewUttFlags) at the next call.
struct SequenceDataReaderOptions {
  int32 batch_size, num_stream, drop_len, skip_width, targets_delay, length_tolerance;
  double frame_limit;
  SequenceDataReaderOptions() : batch_size(20), num_stream(100), drop_len(0), skip_width(1), targets_delay(5), length_tolerance(5), frame_limit(100000) {}
  void Register(OptionsItf *opts) {
    opts->Register("batch-size", &batch_size, "--LSTM-- BPTT batch_size");
    opts->Register("num-stream", &num_stream, "--LSTM-- BPTT multistream training");
    opts->Register("drop-len", &drop_len, "if Sentence frame length greater than drop_len,then drop it, default(0, no drop)");
    opts->Register("skip-width", &skip_width, "num of frame for one skip(default 1, no skip)");
    opts->Register("targets-delay", &targets_delay, "--LSTM-- BPTT targets delay");
    opts->Register("length-tolerance", &length_tolerance, "Allowed length difference of features/targets (frames),for the whole utterance training");
    opts->Register("frame-limit", &frame_limit, "Max number of frames to be processed for whole utterance training");
  }
};

class SequenceDataReader {
 public:
  SequenceDataReader(const std::string &feature_rspecifier, const std::string &targets_rspecifier, const SequenceDataReaderOptions &read_opts)
      : feature_reader_(feature_rspecifier), target_reader_(targets_rspecifier), read_opts_(read_opts), read_done_(false),
        keys_(read_opts.num_stream), feats_(read_opts.num_stream), targets_(read_opts.num_stream), curt_(read_opts.num_stream, 0),
        lent_(read_opts.num_stream, 0), new_utt_flags_(read_opts.num_stream, 0) {}
  bool Done() { return read_done_ && feature_reader_.Done(); }
  const std::vector<int32> &GetNewUttFlags() const { return new_utt_flags_; }
  // aslp-nnet-train-lstm-streams-skip.cc:160-225: that tool runs skip_width passes over the data, pass k keeping the frames
  // k, k + skip_width, ...; it also sends every utterance through a feature transform and counts what it had to leave out
  void SetSkipOffset(int32 offset) { skip_offset_ = offset; }
  void SetFeatureTransform(Nnet *transform) { transform_ = transform; }
  int32 NumNoTargets() const { return num_no_tgt_; }
  int32 NumLengthMismatch() const { return num_len_mismatch_; }
  // When every stream is exhausted the reference leaves `feat` as it was and hands out an all-zero mask (the caller still
  // runs one more step on it, which matters with momentum): same here.
  void ReadData(CuMatrix *feat, Posterior *target, std::vector<BaseFloat> *frame_mask) {
    ASLP_ASSERT(feat != NULL && target != NULL && frame_mask != NULL);
    if (Done()) ASLP_ERR << "Already read done!";
    AddNewUtt();
    FillBatchBuff(feat, target, frame_mask);
  }

 private:
  void AddNewUtt() {  // data-reader.cc:200-270
    for (int s = 0; s < read_opts_.num_stream; s++) {
      if (curt_[s] < lent_[s]) { new_utt_flags_[s] = 0; continue; }
      while (!feature_reader_.Done()) {
        const std::string key = feature_reader_.Key();
        const HostMatrix &raw = feature_reader_.Value();
        if (read_opts_.drop_len > 0 && raw.rows > read_opts_.drop_len) { ASLP_WARN << key << ", too long, droped"; feature_reader_.Next(); continue; }
        if (transform_ != NULL) {  // through the device and back: the batches are assembled on the host
          dev_in_ = raw;
          transform_->Feedforward(dev_in_, &dev_out_);
          dev_out_.CopyToMat(&transformed_);
        }
        const HostMatrix &mat = transform_ != NULL ? transformed_ : raw;
        if (!target_reader_.HasKey(key)) { ASLP_WARN << key << ", missing targets"; num_no_tgt_++; feature_reader_.Next(); continue; }
        const Posterior &target = target_reader_.Value(key);
        if (mat.rows != (int32)target.size()) {
          ASLP_WARN << key << ", length miss-match between feats and targers, skip";
          num_len_mismatch_++;
          feature_reader_.Next();
          continue;
        }
        const int32 skip_width = read_opts_.skip_width;
        if (skip_width > 1) {
          const int32 skip_len = mat.rows > skip_offset_ ? (mat.rows - 1 - skip_offset_) / skip_width + 1 : 0;
          feats_[s].Resize(skip_len, mat.cols);
          targets_[s].assign(skip_len, Posterior::value_type());
          for (int32 i = 0; i < skip_len; i++) {
            const size_t src = (size_t)i * skip_width + skip_offset_;
            std::copy(mat.data.begin() + src * mat.cols, mat.data.begin() + (src + 1) * mat.cols, feats_[s].data.begin() + (size_t)i * mat.cols);
            targets_[s][i] = target[src];
          }
        } else {
          feats_[s] = mat;
          targets_[s] = target;
        }
        keys_[s] = key;
        curt_[s] = 0;
        lent_[s] = feats_[s].rows;
        new_utt_flags_[s] = 1;
        feature_reader_.Next();
        break;
      }
    }
  }
  void FillBatchBuff(CuMatrix *feat, Posterior *target, std::vector<BaseFloat> *frame_mask) {  // :272-325
    const int32 S = read_opts_.num_stream, B = read_opts_.batch_size, delay = read_opts_.targets_delay;
    read_done_ = true;
    for (int s = 0; s < S; s++)
      if (curt_[s] < lent_[s]) { read_done_ = false; break; }
    const int32 feat_dim = feats_[0].cols;
    target->resize((size_t)B * S);
    frame_mask->assign((size_t)B * S, 0.0f);
    if (read_done_) return;
    host_.Resize(B * S, feat_dim);
    for (int t = 0; t < B; t++) {
      for (int s = 0; s < S; s++) {
        const size_t row = (size_t)t * S + s;
        if (lent_[s] == 0) { curt_[s]++; continue; }  // a stream that never got an utterance (the reference would index [-1])
        if (curt_[s] < lent_[s]) { (*frame_mask)[row] = 1.0f; (*target)[row] = targets_[s][curt_[s]]; }
        else { (*target)[row] = targets_[s][lent_[s] - 1]; }
        const int32 src = curt_[s] + delay < lent_[s] ? curt_[s] + delay : lent_[s] - 1;
        std::copy(feats_[s].data.begin() + (size_t)src * feat_dim, feats_[s].data.begin() + (size_t)(src + 1) * feat_dim,
                  host_.data.begin() + row * feat_dim);
        curt_[s]++;
      }
    }
    *feat = host_;
  }
  SequentialBaseFloatMatrixReader feature_reader_;
  RandomAccessPosteriorReader target_reader_;
  SequenceDataReaderOptions read_opts_;
  bool read_done_;
  std::vector<std::string> keys_;
  std::vector<HostMatrix> feats_;
  std::vector<Posterior> targets_;
  std::vector<int32> curt_, lent_, new_utt_flags_;
  HostMatrix host_;
  int32 skip_offset_ = 0, num_no_tgt_ = 0, num_len_mismatch_ = 0;
  Nnet *transform_ = NULL;
  CuMatrix dev_in_, dev_out_;
  HostMatrix transformed_;
};

}  // namespace aslp
