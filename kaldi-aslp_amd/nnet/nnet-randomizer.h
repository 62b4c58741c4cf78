// nnet-randomizer.h -- frame-level shuffling caches of the training tools
// (src/aslp-nnet/nnet-randomizer.{h,cc}): RandomizerMask, MatrixRandomizer, VectorRandomizer,
// StdVectorRandomizer<T>.  Same interface and cache semantics (left-over rows move to the front,
// +1000-row growth, IsFull / Done / NumFrames) as the reference, whose unit test
// (nnet-randomizer-test.cc) is mirrored in tests/test_randomizer_gpu.py.
// Mechanism differences on the device side: Randomize() gathers into a second buffer and swaps
// (the reference first copies the whole cache, nnet-randomizer.cc:78), and Value() returns a view
// of the cache rows instead of copying the minibatch out (:93-98).
// MatrixRandomizer additionally has a STAGED refill (StageBegin / StageAddPinned / StageCommit): the
// utterances of the next cache are uploaded into the second buffer on a CopyLane while the minibatches
// of the current cache train; StageCommit() is the AddData() sequence of the reference collapsed into
// "left-over rows to the front of the staged buffer, swap" -- same rows in the same order, same
// data_begin_ / data_end_ afterwards.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <vector>

#include "cu-matrix.h"

namespace aslp {

class OptionsItf;
struct NnetDataRandomizerOptions {  // nnet-randomizer.h:34-50
  int32 randomizer_size, randomizer_seed, minibatch_size;
  NnetDataRandomizerOptions() : randomizer_size(32768), randomizer_seed(777), minibatch_size(256) {}
  void Register(OptionsItf *opts);   // nnet-randomizer.h:43-47 (defined in data-reader.h)
};

class RandomizerMask {  // :53-64
 public:
  RandomizerMask() {}
  explicit RandomizerMask(const NnetDataRandomizerOptions &conf) { Init(conf); }
  void Init(const NnetDataRandomizerOptions &conf) {
    ASLP_LOG << "Seeding by srand with : " << conf.randomizer_seed;
    SRand(conf.randomizer_seed);
  }
  // std::random_shuffle(begin, end) with the C library generator, written out (libstdc++
  // stl_algo.h: for i in [1, n): swap(v[i], v[rand() % (i + 1)])) because the algorithm left the
  // standard in C++17; the order is part of the reference's reproducibility contract.
  const std::vector<int32> &Generate(int32 mask_size) {
    mask_.resize(mask_size);
    for (int32 i = 0; i < mask_size; i++) mask_[i] = i;
    for (int32 i = 1; i < mask_size; i++) {
      int32 j = Rand() % (i + 1);
      if (i != j) std::swap(mask_[i], mask_[j]);
    }
    return mask_;
  }

 private:
  std::vector<int32> mask_;
};

// shared cache bookkeeping (data_begin_/data_end_ arithmetic of nnet-randomizer.cc:47-71)
class RandomizerBase {
 public:
  RandomizerBase() : data_begin_(0), data_end_(0) {}
  void Init(const NnetDataRandomizerOptions &conf) { conf_ = conf; }
  bool IsFull() { return ((data_begin_ == 0) && (data_end_ > conf_.randomizer_size)); }
  int32 NumFrames() { return data_end_; }
  bool Done() { return (data_end_ - data_begin_ < conf_.minibatch_size); }
  void Next() { data_begin_ += conf_.minibatch_size; }

 protected:
  // returns the number of left-over rows that have to move to the front (0: nothing to move)
  int32 BeginRefill() {
    if (data_begin_ == 0) return 0;
    ASLP_ASSERT(data_begin_ <= data_end_);
    const int32 leftover = data_end_ - data_begin_;
    ASLP_ASSERT(leftover < data_begin_);  // no overlap
    return leftover;
  }
  void CheckRandomize(size_t mask_size) const {
    ASLP_ASSERT(data_begin_ == 0);
    ASLP_ASSERT(data_end_ > 0);
    ASLP_ASSERT(data_end_ == (int32)mask_size);
    CheckCanProgress();
  }

 public:
  // A cache that counts as full (more than randomizer-size frames) and still holds less than one minibatch can never be consumed: the fill
  // loops of the frame tools -- the reference's included, aslp-nnet-train-simple.cc:185-263 -- then come back, find it full, read nothing and
  // spin forever (--randomizer-size smaller than --minibatch-size, found by devtools/r6_fuzz_refmains.py).  Said instead, here where every
  // such loop passes after a fill (Randomize) and by the tools themselves where --randomize=false skips it.
  void CheckCanProgress() const {
    if (data_begin_ == 0 && data_end_ > conf_.randomizer_size && data_end_ < conf_.minibatch_size)
      ASLP_ERR << "the cache is full at " << data_end_ << " frames (--randomizer-size=" << conf_.randomizer_size << ") and holds less than one minibatch (--minibatch-size="
               << conf_.minibatch_size << "): no step can be taken; make --randomizer-size at least --minibatch-size";
  }

 protected:
  void CheckValue() const { ASLP_ASSERT(data_end_ - data_begin_ >= conf_.minibatch_size); }
  NnetDataRandomizerOptions conf_;
  int32 data_begin_, data_end_;
};

class MatrixRandomizer : public RandomizerBase {  // :67-102
 public:
  MatrixRandomizer() : minibatch_(nullptr, 0, 0, 0) {}
  explicit MatrixRandomizer(const NnetDataRandomizerOptions &conf) : minibatch_(nullptr, 0, 0, 0) { Init(conf); }
  void AddData(const CuMatrixBase &m) {  // nnet-randomizer.cc:47-71
    CuSubMatrix dst = Append(m.NumRows(), m.NumCols());
    dst.CopyFromMat(m);
  }
  // same bookkeeping, rows taken straight from page-locked host memory (one async copy, no intermediate device matrix);
  // `src` must stay valid until the stream has passed this call
  void AddDataPinned(const float *src, int32 rows, int32 cols) {
    CuSubMatrix dst = Append(rows, cols);
    dst.CopyFromPinnedHost(src, cols);
  }

 private:
  CuSubMatrix Append(int32 m_rows, int32 m_cols) {
    if (data_.NumCols() == 0) data_.Resize(GrownRows(conf_.randomizer_size), m_cols);   // (a cache is "full" one utterance PAST randomizer_size: room for it from the start)
    if (data_begin_ > 0) {
      const int32 leftover = BeginRefill();
      if (leftover > 0) data_.RowRange(0, leftover).CopyFromMat(data_.RowRange(data_begin_, leftover));
      data_begin_ = 0;
      data_end_ = leftover;
      data_.RowRange(leftover, data_.NumRows() - leftover).SetZero();
    }
    if (data_.NumRows() < data_end_ + m_rows) {
      CuMatrix data_aux(data_);
      data_.Resize(GrownRows(data_end_ + m_rows), data_.NumCols());
      data_.RowRange(0, data_aux.NumRows()).CopyFromMat(data_aux);
    }
    ASLP_ASSERT(m_cols == data_.NumCols());
    CuSubMatrix dst = data_.RowRange(data_end_, m_rows);
    data_end_ += m_rows;
    return dst;
  }

 public:
  // rows a cache buffer grows to when `need` rows no longer fit: the reference's "+ 1000" (:60-64) rounded up to a multiple of 8192.  The two
  // buffers of a randomizer grow at different moments; with exact sizes nearly every refill met a row count the allocator had not seen
  // and paid a 58 MB hipMalloc for it -- 12-16 ms of idle GPU on every other refill of the cfg2 tool run.
  static int32 GrownRows(int32 need) { return (need + 1000 + 8191) / 8192 * 8192; }
  void Randomize(const std::vector<int32> &mask) {  // :73-88
    CheckRandomize(mask.size());
    mask_dev_.CopyFromVec(mask);
    // (the gather target only has to hold the cache's frames; it need not be as tall as data_)
    if (data_aux_.NumRows() < (int32)mask.size() || data_aux_.NumCols() != data_.NumCols())
      data_aux_.Resize(data_.NumRows(), data_.NumCols(), kUndefined);
    cu::Randomize(data_, mask_dev_, &data_aux_);  // rows [0, mask.size()); rows beyond hold no frames
    data_.Swap(&data_aux_);
  }
  const CuMatrixBase &Value() {  // :93-98
    CheckValue();
    minibatch_ = CuSubMatrix(data_, data_begin_, conf_.minibatch_size, 0, data_.NumCols());
    return minibatch_;
  }

  // ---- staged refill -------------------------------------------------------------------------------------------
  // The rows AddData() would find left over once every minibatch of the current cache has been handed out are known as
  // soon as the cache is shuffled: (data_end_ - data_begin_) % minibatch_size of them, the last rows of the cache.  The
  // stage reserves that many rows at its front and appends utterances behind them.
  void StageBegin() {
    ASLP_ASSERT(!staging_);
    if (!lane_) lane_.reset(new CopyLane);   // the copy stream exists only for randomizers that stage
    staging_ = true;
    stage_leftover_ = (data_end_ - data_begin_) % conf_.minibatch_size;
    stage_end_ = stage_leftover_;
    stage_zeroed_ = false;
    // the stage buffer is the gather source of the last Randomize(): the lane must not overwrite it before that ran
    lane_->LaneWaitsForStream();
  }
  bool Staging() const { return staging_; }
  bool StageFull() const { return stage_end_ > conf_.randomizer_size; }  // IsFull() of the cache being staged
  int32 StageFrames() const { return stage_end_; }
  // rows from page-locked memory, appended to the stage on the copy lane; `done` (may be NULL) is recorded behind the copy
  void StageAddPinned(const float *src, int32 rows, int32 cols, StreamMarker *done) {
    ASLP_ASSERT(staging_);
    if (data_aux_.NumCols() == 0) {
      data_aux_.Resize(std::max(GrownRows(conf_.randomizer_size), data_.NumRows()), cols, kUndefined);
      lane_->LaneWaitsForStream();  // the allocator may hand out a block the training stream has not finished with
    }
    ASLP_ASSERT(cols == data_aux_.NumCols());
    if (data_aux_.NumRows() < stage_end_ + rows) {  // the +1000-row growth of :60-64, on the stage
      lane_->Sync();
      CuMatrix grown(GrownRows(stage_end_ + rows), cols, kUndefined);
      if (stage_end_ > stage_leftover_)
        grown.RowRange(stage_leftover_, stage_end_ - stage_leftover_).CopyFromMat(data_aux_.RowRange(stage_leftover_, stage_end_ - stage_leftover_));
      data_aux_.Swap(&grown);
      lane_->LaneWaitsForStream();
      stage_zeroed_ = false;
    }
    if (!stage_zeroed_) {  // :57 zeroes everything behind the left-over rows; rows behind data_end_ are never read, kept for fidelity
      CuSubMatrix rest = data_aux_.RowRange(stage_end_, data_aux_.NumRows() - stage_end_);
      lane_->Zero(rest.Data(), sizeof(float) * (size_t)rest.NumRows() * rest.Stride());
      stage_zeroed_ = true;
    }
    CuSubMatrix dst = data_aux_.RowRange(stage_end_, rows);
    lane_->Upload(dst.Data(), dst.Stride(), src, cols, rows, cols);
    if (done) lane_->Record(done);
    stage_end_ += rows;
  }
  // The current cache must be Done() (or empty).  Afterwards the object is in the state the reference's AddData() calls
  // would have left it in: data_begin_ == 0, data_end_ == left-over + staged rows.
  void StageCommit() {
    ASLP_ASSERT(staging_);
    ASLP_ASSERT(data_end_ - data_begin_ == stage_leftover_);
    staging_ = false;
    if (stage_end_ == stage_leftover_) return;  // nothing was staged: the cache stays as it is, like a loop that added nothing
    lane_->StreamWaitsForLane();
    if (stage_leftover_ > 0) data_aux_.RowRange(0, stage_leftover_).CopyFromMat(data_.RowRange(data_begin_, stage_leftover_));
    data_.Swap(&data_aux_);
    data_begin_ = 0;
    data_end_ = stage_end_;
  }

 private:
  CuMatrix data_, data_aux_;
  std::unique_ptr<CopyLane> lane_;
  bool staging_ = false, stage_zeroed_ = false;
  int32 stage_leftover_ = 0, stage_end_ = 0;
  CuArray<int32> mask_dev_;
  CuSubMatrix minibatch_;
};

// host-side caches share one implementation (VectorRandomizer :105-140 is the float instance,
// Int32VectorRandomizer / PosteriorRandomizer :143-180 the std::vector<T> ones)
template <typename T>
class StdVectorRandomizer : public RandomizerBase {
 public:
  StdVectorRandomizer() {}
  explicit StdVectorRandomizer(const NnetDataRandomizerOptions &conf) { Init(conf); }
  void AddData(const std::vector<T> &v) {  // nnet-randomizer.cc:150-173
    if (data_.size() == 0) data_.resize(conf_.randomizer_size);
    if (data_begin_ > 0) {
      const int32 leftover = BeginRefill();
      if (leftover > 0) std::copy(data_.begin() + data_begin_, data_.begin() + data_begin_ + leftover, data_.begin());
      data_begin_ = 0;
      data_end_ = leftover;
    }
    if (data_.size() < data_end_ + v.size()) data_.resize(data_end_ + v.size() + 1000);
    std::copy(v.begin(), v.end(), data_.begin() + data_end_);
    data_end_ += v.size();
  }
  // the same bookkeeping for a caller that is done with `v`: the elements MOVE into the cache (a Posterior frame is a heap vector: a cache
  // fill of 32768 frames copied them one malloc at a time, twice, and a third time in Randomize -- 7.6 ms of host time per refill of the
  // cfg2 tool run, more than the lead the host has over the GPU there)
  void AddData(std::vector<T> &&v) {
    if (data_.size() == 0) data_.resize(conf_.randomizer_size);
    if (data_begin_ > 0) {
      const int32 leftover = BeginRefill();
      if (leftover > 0) std::move(data_.begin() + data_begin_, data_.begin() + data_begin_ + leftover, data_.begin());
      data_begin_ = 0;
      data_end_ = leftover;
    }
    if (data_.size() < data_end_ + v.size()) data_.resize(data_end_ + v.size() + 1000);
    std::move(v.begin(), v.end(), data_.begin() + data_end_);
    data_end_ += v.size();
    v.clear();
  }
  void Randomize(const std::vector<int32> &mask) {  // :175-187
    CheckRandomize(mask.size());
    // data_[i] = old data_[mask[i]]: the mask is a permutation of [0, mask.size()) (RandomizerMask::Generate), so every old element is
    // taken exactly once and may be moved instead of copied; an arbitrary index list (a caller's own) falls back to copies
    std::vector<char> seen(mask.size(), 0);
    bool permutation = true;
    for (size_t i = 0; i < mask.size() && permutation; i++) {
      const int32 m = mask[i];
      if (m < 0 || (size_t)m >= mask.size() || seen[m]) permutation = false;
      else seen[m] = 1;
    }
    if (!permutation) {
      std::vector<T> data_aux(data_);
      for (size_t i = 0; i < mask.size(); i++) data_.at(i) = data_aux.at(mask.at(i));
      return;
    }
    std::vector<T> data_aux(mask.size());
    for (size_t i = 0; i < mask.size(); i++) data_aux[i] = std::move(data_[mask[i]]);
    std::move(data_aux.begin(), data_aux.end(), data_.begin());
  }
  // the cache itself (frames [Begin(), NumFrames()) have not been handed out yet): for a reader that wants to look at a whole refill at once
  const std::vector<T> &Cache() const { return data_; }
  int32 Begin() const { return data_begin_; }
  const std::vector<T> &Value() {  // :194-201
    CheckValue();
    minibatch_.resize(conf_.minibatch_size);
    std::copy(data_.begin() + data_begin_, data_.begin() + data_begin_ + conf_.minibatch_size, minibatch_.begin());
    return minibatch_;
  }

 private:
  std::vector<T> data_, minibatch_;
};

typedef StdVectorRandomizer<BaseFloat> VectorRandomizer;
typedef StdVectorRandomizer<int32> Int32VectorRandomizer;

}  // namespace aslp
