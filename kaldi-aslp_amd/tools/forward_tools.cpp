// forward_tools.cpp -- the inference tools of src/aslp-nnetbin (forward, forward-mimo, forward-blstm-lc, forward-skip): one entry function per tool
// (Main_<tool name with _ for ->), linked behind tools/main_stub.cpp into bin/<tool name>.
//
// The four tools are ONE utterance loop (ForwardTool::Run) around four ways of sending an utterance through the net: whole (optionally every
// skip-width-th frame with the outputs repeated), several input tables into a graph net, latency-controlled chunks, interleaved
// sub-sequences.  What they share -- flags, model / transform / prior set-up, the non-finite checks, the padded tail and dropped head of
// --time-shift, softmax / log / blank / prior post-processing, writer, progress and summary lines -- is written once; flags, usage and
// log strings are the reference's, byte for byte (decode.sh and the schedulers read them).
#include <cmath>
#include <functional>
#include <memory>

#include "cu-device.h"
#include "kaldi-table.h"
#include "nnet-nnet.h"
#include "nnet-pdf-prior.h"
#include "parse-options.h"

namespace {
using namespace aslp;

// min / max / finiteness of a device matrix in one download (CuMatrixBase::Min / Max / Sum in the reference)
struct MinMax {
  float mn = INFINITY, mx = -INFINITY;
  bool finite = true;
  bool ProbabilityLike() const { return mn >= 0.0 && mx <= 1.0; }
};
MinMax Stats(const CuMatrixBase &m) {
  HostMatrix h;
  m.CopyToMat(&h);
  MinMax s;
  for (float v : h.data) {
    if (!std::isfinite(v)) s.finite = false;
    if (v < s.mn) s.mn = v;
    if (v > s.mx) s.mx = v;
  }
  return s;
}
void RequireFinite(const HostMatrix &m, const char *what, const std::string &utt) {
  for (float v : m.data)
    if (!std::isfinite(v)) ASLP_ERR << "NaN or inf found in " << what << " for " << utt;
}
// --time-shift N (LSTM): the last input frame N more times at the tail ...
void RepeatLastFrame(HostMatrix *mat, int32 n) {
  if (n <= 0) return;
  const size_t cols = mat->cols, last = (size_t)(mat->rows - 1) * cols;
  mat->data.resize((size_t)(mat->rows + n) * cols);
  for (int32 r = 0; r < n; r++) std::copy(mat->data.begin() + last, mat->data.begin() + last + cols, mat->data.begin() + last + (size_t)(r + 1) * cols);
  mat->rows += n;
}
// ... and the first N output frames dropped
void DropFirstFrames(HostMatrix *mat, int32 n) {
  if (n <= 0) return;
  HostMatrix tail(mat->rows - n, mat->cols);
  std::copy(mat->data.begin() + (size_t)n * mat->cols, mat->data.end(), tail.data.begin());
  *mat = tail;
}

// which of the optional flags a tool has (the others keep their neutral defaults)
enum : unsigned { kSoftmaxBlankSkip = 1u, kTimeShift = 2u, kChunks = 4u, kManyInputs = 8u };

struct ForwardTool {
  const char *usage;
  unsigned has;
  bool apply_log_default;
  const char *done_unit;   // " files" -- the latency-controlled tool's summary line has no blank there

  // flags
  PdfPriorOptions prior_opts;
  std::string feature_transform, use_gpu = "no";
  bool no_softmax = false, apply_log = true, add_softmax = false;
  int32 time_shift = 0, skip_width = 0, chunk_size = 64, right_splice = 16;
  float scale_blank = 0.0;

  // session
  Nnet nnet_transf, nnet;
  std::unique_ptr<PdfPrior> pdf_prior;
  int64_t tot_t = 0;
  int32 num_done = 0;
  Timer time;

  ForwardTool(const char *usage_text, unsigned flags, bool log_default, const char *unit = " files")
      : usage(usage_text), has(flags), apply_log_default(log_default), done_unit(unit) { apply_log = log_default; }

  // flags and positional arguments; returns false after printing the usage (exit status 1, as the reference)
  bool Parse(ParseOptions *po, int argc, char *argv[], int min_args, bool exact) {
    prior_opts.Register(po);
    if (has & kChunks) {
      po->Register("chunk-size", &chunk_size, "---BLSTM--- Latency-controlled BPTT chunk size, must be same with training");
      po->Register("right-splice", &right_splice, "---BLSTM--- Latency-controlled BPTT right context size, must be same with training");
    }
    po->Register("feature-transform", &feature_transform, "Feature transform in front of main network (in nnet format)");
    po->Register("no-softmax", &no_softmax, "No softmax on MLP output (or remove it if found), the pre-softmax activations will be used as log-likelihoods, log-priors will be subtracted");
    po->Register("apply-log", &apply_log, "Transform MLP output to logscale");
    po->Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    if (has & kSoftmaxBlankSkip) po->Register("add-softmax", &add_softmax, "add softmax calulation for warp-ctc training");
    if (has & kTimeShift) po->Register("time-shift", &time_shift, "LSTM : repeat last input frame N-times, discrad N initial output frames.");
    if (has & kSoftmaxBlankSkip) {
      po->Register("scale-blank", &scale_blank, "scale the blank posterior for CTC decoding");
      po->Register("skip-width", &skip_width, "num of frame for one skip(default 0, not use skip)");
    }
    po->Read(argc, argv);
    if (exact ? po->NumArgs() != min_args : po->NumArgs() < min_args) { po->PrintUsage(); return false; }
    return true;
  }
  // device, model, transform, prior.  The reference defaults to the CPU here; this engine has none, so "no" (the default) selects a GPU like "yes"
  void Open(const std::string &model_filename) {
    if (has & kManyInputs) CheckFlags();   // (the multi-input tool complains before it touches the device)
    CuDevice::Instantiate().SelectGpuId(use_gpu == "no" ? "yes" : use_gpu);
    if (!(has & kManyInputs) && feature_transform != "") nnet_transf.Read(feature_transform);
    nnet.Read(model_filename);
    if (!(has & kManyInputs)) CheckFlags();
    pdf_prior.reset(new PdfPrior(prior_opts));
    nnet_transf.SetDropoutRetention(1.0);
    nnet.SetDropoutRetention(1.0);
  }
  void CheckFlags() const {
    if (apply_log && no_softmax) ASLP_ERR << "Cannot use both --apply-log=true --no-softmax=true, use only one of the two!";
  }
  // what every tool does to the network output of one utterance before it is written
  void PostProcess(const std::string &utt, CuMatrix *out, HostMatrix *host) {
    if (add_softmax) {
      CuMatrix pre(*out);
      out->ApplySoftMaxPerRow(pre);
    }
    MinMax st = Stats(*out);
    if (!(has & kManyInputs) && !st.finite) ASLP_ERR << "NaN or inf found in nn-output for " << utt;
    bool changed = false;
    if (apply_log) {
      if (!st.ProbabilityLike()) ASLP_WARN << utt << " Applying 'log' to data which don't seem to be probabilities (is there a softmax somwhere?)";
      out->Add(1e-20);  // avoid log(0)
      out->ApplyLog();
      changed = true;
    }
    if (scale_blank > 0.0) { out->ColRange(0, 1).Add(-scale_blank); changed = true; }
    if (prior_opts.class_frame_counts != "") {
      if (changed) st = Stats(*out);
      if (st.ProbabilityLike())
        ASLP_WARN << utt << " Subtracting log-prior on 'probability-like' data in range [0..1] (Did you forget --no-softmax=true or --apply-log=true ?)";
      pdf_prior->SubtractOnLogpost(out);
    }
    out->CopyToMat(host);
    DropFirstFrames(host, time_shift);
    RequireFinite(*host, "final output nn-output", utt);
  }
  void Count(int32 frames) {
    if (num_done % 100 == 0) {
      double time_now = time.Elapsed();
      ASLP_VLOG(1) << "After " << num_done << " utterances: time elapsed = " << time_now / 60 << " min; processed " << tot_t / time_now
                   << " frames per second.";
    }
    num_done++;
    tot_t += frames;
  }
  int Finish() {
    ASLP_LOG << "Done " << num_done << done_unit << " in " << time.Elapsed() / 60 << "min, (fps " << tot_t / time.Elapsed() << ")";
    if (g_verbose_level >= 1) CuDevice::Instantiate().PrintProfile();
    return num_done == 0 ? -1 : 0;
  }
  // One feature table in, one table out: through_net(transformed features, output) is the tool's own part.
  int Run(const std::string &feature_rspecifier, const std::string &feature_wspecifier,
          const std::function<void(const CuMatrix &, CuMatrix *)> &through_net) {
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    BaseFloatMatrixWriter feature_writer(feature_wspecifier);
    CuMatrix feats, feats_transf, nnet_out;
    HostMatrix host;
    time.Reset();
    for (; !feature_reader.Done(); feature_reader.Next()) {
      HostMatrix mat = feature_reader.Value();
      const std::string utt = feature_reader.Key();
      ASLP_VLOG(2) << "Processing utterance " << num_done + 1 << ", " << utt << ", " << mat.rows << "frm";
      RequireFinite(mat, "features", utt);
      const int32 in_rows = mat.rows;
      RepeatLastFrame(&mat, time_shift);
      feats = mat;
      nnet_transf.Feedforward(feats, &feats_transf);
      through_net(feats_transf, &nnet_out);
      PostProcess(utt, &nnet_out, &host);
      feature_writer.Write(utt, host);
      Count(in_rows);
    }
    return Finish();
  }
};

// the net sees the frames `rows` of `in` as one sequence; its outputs go to the rows `to_rows[i]` (several per input frame: repeated) of `out`
void ThroughNetOnRows(Nnet *nnet, const CuMatrix &in, const std::vector<int32> &rows, CuMatrix *sub_in, CuMatrix *sub_out) {
  CuArray<int32> cidx(rows);
  sub_in->Resize((int32)rows.size(), in.NumCols(), kUndefined);
  sub_in->CopyRows(in, cidx);
  nnet->SetSeqLengths(std::vector<int32>(1, (int32)rows.size()));
  nnet->Feedforward(*sub_in, sub_out);
}
}  // namespace

// ======================================================================================================================
// aslp-nnet-forward -- src/aslp-nnetbin/aslp-nnet-forward.cc: forward pass over a feature table, written as a table of
// (log-)posteriors / pre-softmax activations with log-priors subtracted, as decode.sh consumes them.  With --skip-width w
// only every w-th frame goes through the net and each output stands for the w frames behind it.
int Main_aslp_nnet_forward(int argc, char *argv[]) {
  try {
    ForwardTool t("Perform forward pass through Neural Network.\n"
                  "\n"
                  "Usage:  aslp-nnet-forward [options] <model-in> <feature-rspecifier> <feature-wspecifier>\n"
                  "e.g.: \n"
                  " aslp-nnet-forward nnet ark:features.ark ark:mlpoutput.ark\n",
                  kSoftmaxBlankSkip | kTimeShift, true);
    ParseOptions po(t.usage);
    if (!t.Parse(&po, argc, argv, 3, true)) exit(1);
    t.Open(po.GetArg(1));
    CuMatrix sub_in, sub_out;
    return t.Run(po.GetArg(2), po.GetArg(3), [&](const CuMatrix &in, CuMatrix *out) {
      const int32 n = in.NumRows(), w = t.skip_width;
      if (w <= 1) {
        t.nnet.SetSeqLengths(std::vector<int32>(1, n));
        t.nnet.Feedforward(in, out);
        return;
      }
      std::vector<int32> picked;
      for (int32 r = 0; r < n; r += w) picked.push_back(r);
      ThroughNetOnRows(&t.nnet, in, picked, &sub_in, &sub_out);
      out->Resize(n, sub_out.NumCols());
      for (int32 r = 0; r < n; r++) out->RowRange(r, 1).CopyFromMat(sub_out.RowRange(r / w, 1));
    });
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-forward-skip -- src/aslp-nnetbin/aslp-nnet-forward-skip.cc: aslp-nnet-forward for nets trained on every
// skip-width-th frame: the utterance is split into skip-width interleaved sub-sequences (frames off, off + w, off + 2w, ...),
// each goes through the net on its own and its outputs land back on the rows they came from.
int Main_aslp_nnet_forward_skip(int argc, char *argv[]) {
  try {
    ForwardTool t("Perform forward pass through Neural Network.\n"
                  "\n"
                  "Usage:  aslp-nnet-forward-skip [options] <model-in> <feature-rspecifier> <feature-wspecifier>\n"
                  "e.g.: \n"
                  " aslp-nnet-forward-skip nnet ark:features.ark ark:mlpoutput.ark\n",
                  kSoftmaxBlankSkip | kTimeShift, true);
    ParseOptions po(t.usage);
    if (!t.Parse(&po, argc, argv, 3, true)) exit(1);
    t.Open(po.GetArg(1));
    CuMatrix sub_in, sub_out;
    return t.Run(po.GetArg(2), po.GetArg(3), [&](const CuMatrix &in, CuMatrix *out) {
      const int32 n = in.NumRows(), w = t.skip_width;
      if (w < 1) ASLP_ERR << "--skip-width must be at least 1 (with the reference's default of 0 no frame is ever processed)";
      for (int32 off = 0; off < w && off < n; off++) {
        std::vector<int32> picked;
        for (int32 r = off; r < n; r += w) picked.push_back(r);
        ThroughNetOnRows(&t.nnet, in, picked, &sub_in, &sub_out);
        if (out->NumRows() != n || out->NumCols() != sub_out.NumCols()) out->Resize(n, sub_out.NumCols());
        for (size_t i = 0; i < picked.size(); i++) out->RowRange(picked[i], 1).CopyFromMat(sub_out.RowRange((int32)i, 1));
      }
    });
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-forward-blstm-lc -- src/aslp-nnetbin/aslp-nnet-forward-blstm-lc.cc: latency-controlled BLSTM inference.  Every
// utterance goes through the net in chunks of chunk-size frames followed by right-splice frames of look-ahead (one stream,
// history reset at the utterance start); only the chunk part of each output block is kept.
int Main_aslp_nnet_forward_blstm_lc(int argc, char *argv[]) {
  try {
    ForwardTool t("Perform forward pass for Latency Control BLSTM through Neural Network.\n"
                  "\n"
                  "Usage:  aslp-nnet-forward-blstm-lc [options] <model-in> <feature-rspecifier> <feature-wspecifier>\n"
                  "e.g.: \n"
                  " aslp-nnet-forward-blstm-lc nnet ark:features.ark ark:mlpoutput.ark\n",
                  kChunks, true, "files");
    ParseOptions po(t.usage);
    if (!t.Parse(&po, argc, argv, 3, true)) exit(1);
    t.Open(po.GetArg(1));
    t.nnet.SetChunkSize(t.chunk_size);
    // The reference forms its block length from the DEFAULTS of the two options -- `batch_size = chunk_size + right_splice` stands in front of
    // po.Read() (aslp-nnet-forward-blstm-lc.cc:49-58) -- so every run feeds blocks of 64 + 16 = 80 rows whatever --chunk-size /
    // --right-splice say: --chunk-size only decides how many rows of a block are kept (and SetChunkSize), the look-ahead is what is left of
    // the 80.  The recipes' decodes ran that way, so this tool does the same (tests/test_tools_gpu.py runs the reference's own main beside it).
    const int32 block = 64 + 16, feat_dim = t.nnet.InputDim(), out_dim = t.nnet.OutputDim();
    if (t.chunk_size > block) ASLP_ERR << "--chunk-size " << t.chunk_size << " exceeds the block of " << block << " rows (the reference's KALDI_ASSERT(len <= batch_size), :170, holds only up to there)";
    CuMatrix block_in, block_out;
    return t.Run(po.GetArg(2), po.GetArg(3), [&](const CuMatrix &in, CuMatrix *out) {
      t.nnet.ResetLstmStreams(std::vector<int32>(1, 1));
      const int32 n = in.NumRows();
      out->Resize(n, out_dim);
      block_in.Resize(block, feat_dim);  // zeroed once per utterance: a short last block keeps the previous block's tail rows (:139-152)
      for (int32 at = 0; at < n; at += t.chunk_size) {
        const int32 len = std::min(block, n - at), keep = std::min(t.chunk_size, n - at);
        block_in.RowRange(0, len).CopyFromMat(in.RowRange(at, len));
        t.nnet.Feedforward(block_in, &block_out);
        out->RowRange(at, keep).CopyFromMat(block_out.RowRange(0, keep));
      }
    });
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-forward-mimo -- src/aslp-nnetbin/aslp-nnet-forward-mimo.cc: forward pass through a graph net with several
// <InputLayer>s: one feature table per input (same keys, same order), the LAST output is written (the main task of a
// multi-task net).
int Main_aslp_nnet_forward_mimo(int argc, char *argv[]) {
  try {
    ForwardTool t("Perform forward pass through Neural Network.\n"
                  "\n"
                  "Usage:  aslp-nnet-forward-mimo [options] <model-in> <feature-rspecifier_1>...<feature_rspecifier_n> <feature-wspecifier>\n"
                  "e.g.: \n"
                  " aslp-nnet-forward-mimo nnet ark:features1.ark ark:features2.ark ark:mlpoutput.ark\n",
                  kTimeShift | kManyInputs, false);
    ParseOptions po(t.usage);
    if (!t.Parse(&po, argc, argv, 3, false)) exit(1);
    t.Open(po.GetArg(1));
    const int num_args = po.NumArgs(), num_input = t.nnet.NumInput(), num_output = t.nnet.NumOutput();
    ASLP_LOG << "Nnet num_input " << num_input << " num_output " << num_output;
    if (num_args != 1 + num_input + 1) { po.PrintUsage(); exit(1); }

    std::vector<std::unique_ptr<SequentialBaseFloatMatrixReader>> readers;
    for (int i = 0; i < num_input; i++) readers.emplace_back(new SequentialBaseFloatMatrixReader(po.GetArg(i + 2)));
    BaseFloatMatrixWriter feature_writer(po.GetArg(num_args));
    std::vector<CuMatrix> in_store(num_input), out_store(num_output);
    std::vector<const CuMatrixBase *> nnet_in(num_input);
    std::vector<CuMatrix *> nnet_outs(num_output);
    for (int i = 0; i < num_input; i++) nnet_in[i] = &in_store[i];
    for (int i = 0; i < num_output; i++) nnet_outs[i] = &out_store[i];
    HostMatrix host;
    t.time.Reset();
    for (; !readers[0]->Done(); ) {
      const std::string utt = readers[0]->Key();
      ASLP_VLOG(2) << "Processing " << utt;
      for (int i = 0; i < num_input; i++) {   // the tables are walked in step
        if (readers[i]->Done() || readers[i]->Key() != utt)
          ASLP_ERR << "Different key from the features " << utt << " " << (readers[i]->Done() ? std::string("<end of table>") : readers[i]->Key())
                   << " please check the order of feat scp";
        HostMatrix mat = readers[i]->Value();
        RequireFinite(mat, "features", utt);
        RepeatLastFrame(&mat, t.time_shift);
        in_store[i] = mat;
      }
      t.nnet.Feedforward(nnet_in, &nnet_outs);
      t.PostProcess(utt, &out_store[num_output - 1], &host);  // if multitask, only the last task is written
      feature_writer.Write(utt, host);
      t.Count(in_store[0].NumRows());
      for (auto &r : readers) r->Next();
    }
    for (int i = 1; i < num_input; i++) ASLP_ASSERT(readers[i]->Done());
    return t.Finish();
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
