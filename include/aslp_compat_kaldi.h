/* aslp_compat_kaldi.h -- source-level drop-in for the Component / Nnet C++ API (SURVEY 8b seam B4).
 *
 * A caller written against the reference -- `using namespace kaldi; using namespace kaldi::aslp_nnet;`, `CuMatrix<BaseFloat>`,
 * `KALDI_LOG`, `trn_opts.Register(&po)`, `#include "aslp-nnet/nnet-nnet.h"` -- compiles against THIS engine unchanged:
 *
 *   hipcc -x hip --offload-arch=gfx950 -std=c++17 -I include/kaldi_compat -I include \
 *         -I kaldi-aslp_amd/nnet -I kaldi-aslp_amd/util -I kaldi-aslp_amd/csrc \
 *         <reference>/src/aslp-nnetbin/aslp-nnet-train-frame.cc -L kaldi-aslp_amd -laslp_hip
 *
 * include/kaldi_compat/ holds one forwarding header per reference header path such a caller includes (base/kaldi-common.h,
 * util/common-utils.h, base/timer.h, aslp-cudamatrix/cu-device.h, aslp-nnet/{nnet-trnopts,nnet-nnet,nnet-loss,data-reader,...}.h); each
 * of them includes this file.  What replaces what (reference file:line):
 *   kaldi::aslp_nnet::Nnet / Component / UpdatableComponent      aslp-nnet/nnet-nnet.h:38-193, nnet-component.h:45-347
 *   kaldi::aslp_nnet::NnetTrainOptions (+ Register)              aslp-nnet/nnet-trnopts.h:29-63
 *   kaldi::aslp_nnet::LossItf / Xent / Mse / MultiTaskLoss       aslp-nnet/nnet-loss.h:35-218
 *   kaldi::aslp_nnet::FrameDataReader, NnetDataRandomizerOptions aslp-nnet/data-reader.h:25-63, nnet-randomizer.h:34-50
 *   kaldi::CuMatrix<Real> / CuMatrixBase<Real> / CuVector<Real>  aslp-cudamatrix/cu-matrix.h, cu-vector.h (Real = float: the engine is fp32)
 *   kaldi::CuDevice                                              aslp-cudamatrix/cu-device.h:43-151
 *   kaldi::ParseOptions, Timer, Posterior, the table typedefs    util/parse-options.h, base/timer.h, hmm/posterior.h, util/table-types.h
 *   KALDI_LOG / KALDI_WARN / KALDI_ERR / KALDI_VLOG / KALDI_ASSERT   base/kaldi-error.h (errors are std::runtime_error, as in the reference)
 * HAVE_CUDA is defined to 1: the `#if HAVE_CUDA==1` device-selection blocks of the reference's tools are the ones that apply.
 *   kaldi::Matrix<Real> / Vector<Real> / SubVector / SubMatrix   matrix/kaldi-matrix.h, kaldi-vector.h (aslp_compat_kaldi_matrix.h: over HostMatrix / std::vector<float>)
 *   kaldi::aslp_nnet::FrameDataReader (both constructors)        aslp-nnet/data-reader.h:23-47
 *   kaldi::aslp_nnet::Ctc, WarpCtc, AffineTransform              aslp-nnet/ctc-loss.h, warp-ctc.h, nnet-affine-transform.h
 *   fst::SymbolTable (ReadText, Find)                            OpenFst's symbol-table text format, as aslp-nnet-train-ctc.cc:81-85, 152-156 uses it
 *   ReadKaldiObject / WriteKaldiObject, KALDI_ISFINITE, g_kaldi_verbose_level, SplitStringToVector
 * Proof: tests/test_compat_kaldi_cpu.py compiles ALL 23 mains of the reference's aslp-nnetbin/ against this header (development container:
 * the reference tree does not travel); `make -C kaldi-aslp_amd refmains` links them into bin_ref/, and tests/test_tools_gpu.py runs every tool
 * test a second time on those binaries (test_every_reference_main_passes_the_tool_tests_of_its_name). */
#ifndef ASLP_COMPAT_KALDI_H_
#define ASLP_COMPAT_KALDI_H_

#ifndef HAVE_CUDA
#define HAVE_CUDA 1
#endif

#include <cstdint>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "cu-device.h"
#include "cu-matrix.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-basic.h"
#include "nnet-nnet.h"
#include "nnet-pdf-prior.h"
#include "nnet-randomizer.h"
#include "parse-options.h"
#include "posterior.h"
#include "ctc-loss.h"
#include "warp-ctc.h"
#include "aslp_compat_kaldi_matrix.h"

namespace kaldi {

typedef float BaseFloat;
typedef int16_t int16;
typedef int32_t int32;
typedef int64_t int64;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef uint64_t uint64;
typedef float float32;
typedef double double64;

using ::aslp::OptionsItf;
using ::aslp::ParseOptions;
using ::aslp::Input;    /* util/kaldi-io.h: extended filenames (files, pipes, offsets) */
using ::aslp::Output;
inline void SetVerboseLevel(int32 level) { ::aslp::g_verbose_level = level; }   /* base/kaldi-error.h */
inline int32 GetVerboseLevel() { return ::aslp::g_verbose_level; }
static int &g_kaldi_verbose_level = ::aslp::g_verbose_level;   /* (read directly by the tools' vlog blocks) */
using ::aslp::SplitStringToVector;   /* util/text-utils.h */
using ::aslp::RandGauss;             /* base/kaldi-math.h */
using ::aslp::RandUniform;
using ::aslp::Posterior;
using ::aslp::Timer;
using ::aslp::CuDevice;
using ::aslp::kSetZero;
using ::aslp::kUndefined;
using ::aslp::kCopyData;
using ::aslp::SequentialBaseFloatMatrixReader;
using ::aslp::RandomAccessBaseFloatMatrixReader;
using ::aslp::BaseFloatMatrixWriter;
using ::aslp::SequentialBaseFloatVectorReader;
using ::aslp::RandomAccessBaseFloatVectorReader;
using ::aslp::BaseFloatVectorWriter;
using ::aslp::RandomAccessBaseFloatReader;
using ::aslp::SequentialInt32VectorReader;
using ::aslp::RandomAccessInt32VectorReader;
using ::aslp::Int32VectorWriter;
using ::aslp::RandomAccessPosteriorReader;

/* the reference's matrix classes are templates over the element type; the engine computes in fp32 only */
template <typename Real> using CuMatrixBase = ::aslp::CuMatrixBase;
template <typename Real> using CuMatrix = ::aslp::CuMatrix;
template <typename Real> using CuSubMatrix = ::aslp::CuSubMatrix;
template <typename Real> using CuVectorBase = ::aslp::CuVectorBase;
template <typename Real> using CuVector = ::aslp::CuVector;
template <typename T> using CuArray = ::aslp::CuArray<T>;

namespace aslp_nnet {
using ::aslp::Component;
using ::aslp::UpdatableComponent;
using ::aslp::Nnet;
using ::aslp::NnetTrainOptions;
using ::aslp::NnetDataRandomizerOptions;
using ::aslp::RandomizerMask;
using ::aslp::MatrixRandomizer;
using ::aslp::Int32VectorRandomizer;
/* nnet-randomizer.h:120-141: Value() is a VectorBase in the reference (the engine hands out its std::vector<float>; this copies the minibatch's weights) */
class VectorRandomizer : public ::aslp::VectorRandomizer {
 public:
  VectorRandomizer() {}
  explicit VectorRandomizer(const NnetDataRandomizerOptions &conf) : ::aslp::VectorRandomizer(conf) {}
  const Vector<BaseFloat> &Value() { value_ = ::aslp::VectorRandomizer::Value(); return value_; }
 private:
  Vector<BaseFloat> value_;
};
using ::aslp::PosteriorRandomizer;
using ::aslp::LossItf;
using ::aslp::Xent;
using ::aslp::Mse;
using ::aslp::MultiTaskLoss;
using ::aslp::AffineTransform;
/* aslp-nnet/data-reader.h:23-47: ONE class with a single-table and a multi-table constructor.  The engine has a threaded reader for the first
 * (nnet/data-reader.h FrameDataReader) and a lock-step one for the second (MimoFrameDataReader); this is the reference's class over both. */
class FrameDataReader {
 public:
  FrameDataReader(const std::string &feature_rspecifier, const std::string &targets_rspecifier, const NnetDataRandomizerOptions &rand_opts)
      : one_(new ::aslp::FrameDataReader(feature_rspecifier, targets_rspecifier, rand_opts)) {}
  FrameDataReader(const std::vector<std::string> &feature_rspecifiers, const std::vector<std::string> &targets_rspecifiers,
                  const NnetDataRandomizerOptions &rand_opts)
      : many_(new ::aslp::MimoFrameDataReader(feature_rspecifiers, targets_rspecifiers, rand_opts)) {}
  bool ReadData(const ::aslp::CuMatrixBase **feat, const ::aslp::Posterior **targets) {
    ASLP_ASSERT(one_ != nullptr);
    return one_->ReadData(feat, targets);
  }
  /* (the reference returns nothing here; where no full minibatch is left its randomizers' own checks end the run, data-reader.cc:130-150 --
   * a caller that ignores the result must not train on the previous minibatch again, so that case is an error here as well) */
  void ReadData(std::vector<const ::aslp::CuMatrixBase *> *input, std::vector<const ::aslp::Posterior *> *output) {
    ASLP_ASSERT(many_ != nullptr);
    if (!many_->ReadData(input, output)) ASLP_ERR << "FrameDataReader::ReadData: no full minibatch left";
  }
  bool Done() { return one_ ? one_->Done() : many_->Done(); }
 private:
  std::unique_ptr< ::aslp::FrameDataReader> one_;
  std::unique_ptr< ::aslp::MimoFrameDataReader> many_;
};
using ::aslp::SequenceDataReaderOptions;
using ::aslp::SequenceDataReader;
using ::aslp::PdfPriorOptions;
using ::aslp::PdfPrior;
using ::aslp::Ctc;
using ::aslp::WarpCtc;
}  // namespace aslp_nnet

/* util/kaldi-io.h ReadKaldiObject / WriteKaldiObject: an object with Read(istream, binary) / Write(ostream, binary) from / to an extended filename */
template <class C> void ReadKaldiObject(const std::string &rxfilename, C *c) {
  bool binary;
  Input ki(rxfilename, &binary);
  c->Read(ki.Stream(), binary);
}
template <class C> void WriteKaldiObject(const C &c, const std::string &wxfilename, bool binary) {
  Output ko(wxfilename, binary);
  c.Write(ko.Stream(), binary);
}

}  // namespace kaldi

/* fst::SymbolTable as far as the reference's tools use it (aslp-nnet-train-ctc.cc:81-85, 152-156: token names in a verbose log): the text
 * format of OpenFst's symbol tables, "<symbol> <integer>" per line (OpenFst is an external library of the reference's build, not vendored) */
#include <fstream>
#include <map>
#include <sstream>
namespace fst {
class SymbolTable {
 public:
  static SymbolTable *ReadText(const std::string &filename) {
    std::ifstream is(filename);
    if (!is) return nullptr;
    SymbolTable *t = new SymbolTable();
    std::string line, sym;
    long long key;
    while (std::getline(is, line)) {
      std::istringstream ls(line);
      if (!(ls >> sym)) continue;
      if (!(ls >> key)) { delete t; return nullptr; }
      t->names_[key] = sym;
    }
    return t;
  }
  std::string Find(long long key) const { auto it = names_.find(key); return it == names_.end() ? std::string() : it->second; }
 private:
  std::map<long long, std::string> names_;
};
}  // namespace fst

/* The reference's tools seed the C library generator (`std::srand(seed)`, aslp-nnetbin/aslp-nnet-init.cc:56) and its components draw from
 * rand().  The engine draws from a private copy of that generator (nnet/base.h: inside a HIP process the global rand() state is not the
 * caller's alone), so a caller's srand must seed both.  `srand` becomes a name that exists in the global namespace AND in std. */
#include <cstdlib>
inline void aslp_compat_srand(unsigned seed) {
  (::srand)(seed);
  ::aslp::SRand(seed);
}
namespace std { using ::aslp_compat_srand; }
#define srand aslp_compat_srand

/* (the reference's aslp-nnet-train-simple-mpi.cc:37 spells `string::npos` at global scope: some header of its Kaldi put the name there) */
using std::string;

#define KALDI_LOG ASLP_LOG
#define KALDI_WARN ASLP_WARN
#define KALDI_ERR ASLP_ERR
#define KALDI_VLOG(v) ASLP_VLOG(v)
#define KALDI_ASSERT(cond) ASLP_ASSERT(cond)
#define KALDI_ISFINITE(x) std::isfinite(x)
#define KALDI_ISNAN(x) std::isnan(x)
#define KALDI_ISINF(x) std::isinf(x)

#endif  /* ASLP_COMPAT_KALDI_H_ */
