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
    if (Done()) return false;
    *feat = &feature_randomizer_.Value();
    feature_randomizer_.Next();
    *targets = &targets_randomizer_.Value();
    targets_randomizer_.Next();
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
  void Produce() {
    try {
      CuDevice::Instantiate().BindThread();
      for (; !feature_reader_->Done(); feature_reader_->Next()) {
        {
          std::lock_guard<std::mutex> lk(mu_);
          if (stop_) return;
        }
        Item item;
        item.key = feature_reader_->Key();
        ASLP_VLOG(3) << "Reading " << item.key;
        const HostMatrix &mat = feature_reader_->Value();
        item.rows = mat.rows;
        item.cols = mat.cols;
        item.has_targets = targets_reader_->HasKey(item.key);
        if (item.has_targets) {
          item.targets = targets_reader_->Value(item.key);
          item.block = TakeBlock(mat.data.size());
          std::memcpy(item.block.ptr, mat.data.data(), sizeof(float) * mat.data.size());
        }
        Push(std::move(item));
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
  void FillRandomizer() {  // data-reader.cc:66-128
    Timer fill_timer;
    double t_wait = 0.0;
    RecycleBlocks();
    while (true) {
      if (feature_randomizer_.IsFull()) break;
      Item item;
      {
        Timer tw;
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !queue_.empty(); });
        item = std::move(queue_.front());
        queue_.pop_front();
        queued_frames_ -= item.rows;
        lk.unlock();
        cv_.notify_all();
        t_wait += tw.Elapsed();
      }
      if (!item.error.empty()) throw std::runtime_error(item.error);
      if (item.end) { read_done_ = true; break; }
      if (!item.has_targets) {
        ASLP_WARN << item.key << ", missing targets";
        num_no_tgt_++;
        continue;
      }
      if ((int32)item.targets.size() != item.rows) ASLP_ERR << "feature and target dim must match";
      feature_randomizer_.AddDataPinned(item.block.ptr, item.rows, item.cols);
      std::unique_ptr<StreamMarker> marker(new StreamMarker);
      marker->Record();
      in_flight_.push_back(std::make_pair(item.block, std::move(marker)));
      targets_randomizer_.AddData(item.targets);
      num_done_++;
    }
    CuDevice::Instantiate().AccuProfile("FrameDataReader: waiting for the reader thread", t_wait);
    CuDevice::Instantiate().AccuProfile("FrameDataReader::FillRandomizer (host, total)", fill_timer.Elapsed());
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

  std::string feature_rspecifier_, targets_rspecifier_;
  std::unique_ptr<SequentialBaseFloatMatrixReader> feature_reader_;  // used by the producer thread only
  std::unique_ptr<RandomAccessPosteriorReader> targets_reader_;      // used by the producer thread only
  RandomizerMask randomizer_mask_;
  MatrixRandomizer feature_randomizer_;
  PosteriorRandomizer targets_randomizer_;
  NnetDataRandomizerOptions rand_opts_;
  bool randomize_, read_done_;
  int32 num_no_tgt_, num_done_;
  std::thread producer_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_;
  int64_t queued_frames_;
  std::deque<Item> queue_;
  std::vector<Block> free_blocks_;
  std::vector<std::pair<Block, std::unique_ptr<StreamMarker>>> in_flight_;
};

}  // namespace aslp
