// forward_tools.cpp -- the inference tools of src/aslp-nnetbin (forward, forward-mimo, forward-blstm-lc, forward-skip): one entry function per tool
// (Main_<tool name with _ for ->), linked behind tools/main_stub.cpp into bin/<tool name>.
#include <cmath>
#include <memory>

#include "cu-device.h"
#include "kaldi-table.h"
#include "nnet-nnet.h"
#include "nnet-pdf-prior.h"
#include "parse-options.h"

// ======================================================================================================================
// aslp-nnet-forward -- src/aslp-nnetbin/aslp-nnet-forward.cc: forward pass over a feature table, written as a table of
// (log-)posteriors / pre-softmax activations with log-priors subtracted, as decode.sh consumes them.
namespace {
// min / max / finiteness of a device matrix in one download of its statistics (CuMatrixBase::Min / Max / Sum in the reference)
struct MinMax { float mn, mx; bool finite; };
MinMax Stats(const aslp::CuMatrixBase &m) {
  aslp::HostMatrix h;
  m.CopyToMat(&h);
  MinMax s = {INFINITY, -INFINITY, true};
  for (float v : h.data) {
    if (!std::isfinite(v)) s.finite = false;
    if (v < s.mn) s.mn = v;
    if (v > s.mx) s.mx = v;
  }
  return s;
}
}  // namespace

int Main_aslp_nnet_forward(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform forward pass through Neural Network.\n"
        "\n"
        "Usage:  aslp-nnet-forward [options] <model-in> <feature-rspecifier> <feature-wspecifier>\n"
        "e.g.: \n"
        " aslp-nnet-forward nnet ark:features.ark ark:mlpoutput.ark\n";
    ParseOptions po(usage);
    PdfPriorOptions prior_opts;
    prior_opts.Register(&po);
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform in front of main network (in nnet format)");
    bool no_softmax = false;
    po.Register("no-softmax", &no_softmax, "No softmax on MLP output (or remove it if found), the pre-softmax activations will be used as log-likelihoods, log-priors will be subtracted");
    bool apply_log = true;
    po.Register("apply-log", &apply_log, "Transform MLP output to logscale");
    std::string use_gpu = "no";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    bool add_softmax = false;
    po.Register("add-softmax", &add_softmax, "add softmax calulation for warp-ctc training");
    int32 time_shift = 0;
    po.Register("time-shift", &time_shift, "LSTM : repeat last input frame N-times, discrad N initial output frames.");
    float scale_blank = 0.0;
    po.Register("scale-blank", &scale_blank, "scale the blank posterior for CTC decoding");
    int32 skip_width = 0;
    po.Register("skip-width", &skip_width, "num of frame for one skip(default 0, not use skip)");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); exit(1); }
    std::string model_filename = po.GetArg(1), feature_rspecifier = po.GetArg(2), feature_wspecifier = po.GetArg(3);

    // the reference defaults to the CPU here; this engine has none, so "no" (the default) selects a GPU like "yes"
    CuDevice::Instantiate().SelectGpuId(use_gpu == "no" ? "yes" : use_gpu);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);
    Nnet nnet;
    nnet.Read(model_filename);
    if (apply_log && no_softmax) ASLP_ERR << "Cannot use both --apply-log=true --no-softmax=true, use only one of the two!";
    PdfPrior pdf_prior(prior_opts);
    nnet_transf.SetDropoutRetention(1.0);
    nnet.SetDropoutRetention(1.0);

    int64_t tot_t = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    BaseFloatMatrixWriter feature_writer(feature_wspecifier);
    CuMatrix feats, feats_transf, nnet_out, skip_feat, skip_out;
    HostMatrix nnet_out_host;
    Timer time;
    int32 num_done = 0;
    for (; !feature_reader.Done(); feature_reader.Next()) {
      HostMatrix mat = feature_reader.Value();
      std::string utt = feature_reader.Key();
      ASLP_VLOG(2) << "Processing utterance " << num_done + 1 << ", " << utt << ", " << mat.rows << "frm";
      for (float v : mat.data)
        if (!std::isfinite(v)) ASLP_ERR << "NaN or inf found in features for " << utt;
      const int32 in_rows = mat.rows;
      if (time_shift > 0) {  // copy the last frame of LSTM input N-times
        const int32 last_row = mat.rows - 1, cols = mat.cols;
        mat.data.resize((size_t)(mat.rows + time_shift) * cols);
        for (int32 r = last_row + 1; r < last_row + 1 + time_shift; r++)
          std::copy(mat.data.begin() + (size_t)last_row * cols, mat.data.begin() + (size_t)(last_row + 1) * cols, mat.data.begin() + (size_t)r * cols);
        mat.rows += time_shift;
      }
      feats = mat;
      nnet_transf.Feedforward(feats, &feats_transf);
      std::vector<int32> frame_num_utt;
      if (skip_width > 1) {  // skip prediction: every skip_width-th frame goes through the net, outputs are repeated
        const int32 skip_len = (feats_transf.NumRows() - 1) / skip_width + 1;
        skip_feat.Resize(skip_len, feats_transf.NumCols());
        for (int32 i = 0; i < skip_len; i++) skip_feat.RowRange(i, 1).CopyFromMat(feats_transf.RowRange(i * skip_width, 1));
        frame_num_utt.push_back(skip_feat.NumRows());
        nnet.SetSeqLengths(frame_num_utt);
        nnet.Feedforward(skip_feat, &skip_out);
        nnet_out.Resize(feats_transf.NumRows(), skip_out.NumCols());
        for (int32 i = 0; i < skip_len; i++)
          for (int32 j = 0; j < skip_width; j++) {
            const int32 idx = i * skip_width + j;
            if (idx < nnet_out.NumRows()) nnet_out.RowRange(idx, 1).CopyFromMat(skip_out.RowRange(i, 1));
          }
      } else {
        frame_num_utt.push_back(feats_transf.NumRows());
        nnet.SetSeqLengths(frame_num_utt);
        nnet.Feedforward(feats_transf, &nnet_out);
      }
      if (add_softmax) {
        CuMatrix tmp_out(nnet_out);
        nnet_out.ApplySoftMaxPerRow(tmp_out);
      }
      MinMax st = Stats(nnet_out);
      if (!st.finite) ASLP_ERR << "NaN or inf found in nn-output for " << utt;
      if (apply_log) {
        if (!(st.mn >= 0.0 && st.mx <= 1.0))
          ASLP_WARN << utt << " Applying 'log' to data which don't seem to be probabilities (is there a softmax somwhere?)";
        nnet_out.Add(1e-20);  // avoid log(0)
        nnet_out.ApplyLog();
      }
      if (scale_blank > 0.0) nnet_out.ColRange(0, 1).Add(-scale_blank);
      if (prior_opts.class_frame_counts != "") {
        if (apply_log || scale_blank > 0.0) st = Stats(nnet_out);
        if (st.mn >= 0.0 && st.mx <= 1.0)
          ASLP_WARN << utt << " Subtracting log-prior on 'probability-like' data in range [0..1] (Did you forget --no-softmax=true or --apply-log=true ?)";
        pdf_prior.SubtractOnLogpost(&nnet_out);
      }
      nnet_out.CopyToMat(&nnet_out_host);
      if (time_shift > 0) {  // remove N first frames of LSTM output
        HostMatrix tmp(nnet_out_host.rows - time_shift, nnet_out_host.cols);
        std::copy(nnet_out_host.data.begin() + (size_t)time_shift * nnet_out_host.cols, nnet_out_host.data.end(), tmp.data.begin());
        nnet_out_host = tmp;
      }
      for (float v : nnet_out_host.data)
        if (!std::isfinite(v)) ASLP_ERR << "NaN or inf found in final output nn-output for " << utt;
      feature_writer.Write(feature_reader.Key(), nnet_out_host);
      if (num_done % 100 == 0) {
        double time_now = time.Elapsed();
        ASLP_VLOG(1) << "After " << num_done << " utterances: time elapsed = " << time_now / 60 << " min; processed " << tot_t / time_now
                     << " frames per second.";
      }
      num_done++;
      tot_t += in_rows;
    }
    ASLP_LOG << "Done " << num_done << " files in " << time.Elapsed() / 60 << "min, (fps " << tot_t / time.Elapsed() << ")";
    if (g_verbose_level >= 1) CuDevice::Instantiate().PrintProfile();
    if (num_done == 0) return -1;
    return 0;
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
  using namespace aslp;
  try {
    const char *usage =
        "Perform forward pass through Neural Network.\n"
        "\n"
        "Usage:  aslp-nnet-forward-mimo [options] <model-in> <feature-rspecifier_1>...<feature_rspecifier_n> <feature-wspecifier>\n"
        "e.g.: \n"
        " aslp-nnet-forward-mimo nnet ark:features1.ark ark:features2.ark ark:mlpoutput.ark\n";
    ParseOptions po(usage);
    PdfPriorOptions prior_opts;
    prior_opts.Register(&po);
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform in front of main network (in nnet format)");
    bool no_softmax = false;
    po.Register("no-softmax", &no_softmax, "No softmax on MLP output (or remove it if found), the pre-softmax activations will be used as log-likelihoods, log-priors will be subtracted");
    bool apply_log = false;
    po.Register("apply-log", &apply_log, "Transform MLP output to logscale");
    std::string use_gpu = "no";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    int32 time_shift = 0;
    po.Register("time-shift", &time_shift, "LSTM : repeat last input frame N-times, discrad N initial output frames.");
    po.Read(argc, argv);
    if (po.NumArgs() < 3) { po.PrintUsage(); exit(1); }
    if (apply_log && no_softmax) ASLP_ERR << "Cannot use both --apply-log=true --no-softmax=true, use only one of the two!";
    CuDevice::Instantiate().SelectGpuId(use_gpu == "no" ? "yes" : use_gpu);  // no host engine here, see aslp-nnet-forward

    const int num_args = po.NumArgs();
    std::string model_filename = po.GetArg(1), feature_wspecifier = po.GetArg(num_args);
    Nnet nnet;
    nnet.Read(model_filename);
    const int num_input = nnet.NumInput(), num_output = nnet.NumOutput();
    ASLP_LOG << "Nnet num_input " << num_input << " num_output " << num_output;
    if (num_args != 1 + num_input + 1) { po.PrintUsage(); exit(1); }
    PdfPrior pdf_prior(prior_opts);
    nnet.SetDropoutRetention(1.0);

    std::vector<std::unique_ptr<SequentialBaseFloatMatrixReader>> readers;
    for (int i = 0; i < num_input; i++) readers.emplace_back(new SequentialBaseFloatMatrixReader(po.GetArg(i + 2)));
    BaseFloatMatrixWriter feature_writer(feature_wspecifier);
    std::vector<CuMatrix> in_store(num_input), out_store(num_output);
    std::vector<const CuMatrixBase *> nnet_in(num_input);
    std::vector<CuMatrix *> nnet_outs(num_output);
    for (int i = 0; i < num_input; i++) nnet_in[i] = &in_store[i];
    for (int i = 0; i < num_output; i++) nnet_outs[i] = &out_store[i];
    HostMatrix nnet_out_host;
    int64_t tot_t = 0;
    Timer time;
    int32 num_done = 0;
    while (!readers[0]->Done()) {
      const std::string utt = readers[0]->Key();
      ASLP_VLOG(2) << "Processing " << utt;
      for (int i = 0; i < num_input; i++) {
        if (readers[i]->Done() || readers[i]->Key() != utt)
          ASLP_ERR << "Different key from the features " << utt << " " << (readers[i]->Done() ? std::string("<end of table>") : readers[i]->Key())
                   << " please check the order of feat scp";
        HostMatrix mat = readers[i]->Value();
        for (float v : mat.data)
          if (!std::isfinite(v)) ASLP_ERR << "NaN or inf found in features for " << utt;
        if (time_shift > 0) {
          const int32 last_row = mat.rows - 1, cols = mat.cols;
          mat.data.resize((size_t)(mat.rows + time_shift) * cols);
          for (int32 r = last_row + 1; r < last_row + 1 + time_shift; r++)
            std::copy(mat.data.begin() + (size_t)last_row * cols, mat.data.begin() + (size_t)(last_row + 1) * cols, mat.data.begin() + (size_t)r * cols);
          mat.rows += time_shift;
        }
        in_store[i] = mat;
      }
      nnet.Feedforward(nnet_in, &nnet_outs);
      CuMatrix &nnet_out = out_store[num_output - 1];  // if multitask, only the last task is written
      MinMax st = Stats(nnet_out);
      if (apply_log) {
        if (!(st.mn >= 0.0 && st.mx <= 1.0))
          ASLP_WARN << utt << " Applying 'log' to data which don't seem to be probabilities (is there a softmax somwhere?)";
        nnet_out.Add(1e-20);
        nnet_out.ApplyLog();
        st = Stats(nnet_out);
      }
      if (prior_opts.class_frame_counts != "") {
        if (st.mn >= 0.0 && st.mx <= 1.0)
          ASLP_WARN << utt << " Subtracting log-prior on 'probability-like' data in range [0..1] (Did you forget --no-softmax=true or --apply-log=true ?)";
        pdf_prior.SubtractOnLogpost(&nnet_out);
      }
      nnet_out.CopyToMat(&nnet_out_host);
      if (time_shift > 0) {
        HostMatrix tmp(nnet_out_host.rows - time_shift, nnet_out_host.cols);
        std::copy(nnet_out_host.data.begin() + (size_t)time_shift * nnet_out_host.cols, nnet_out_host.data.end(), tmp.data.begin());
        nnet_out_host = tmp;
      }
      for (float v : nnet_out_host.data)
        if (!std::isfinite(v)) ASLP_ERR << "NaN or inf found in final output nn-output for " << utt;
      feature_writer.Write(utt, nnet_out_host);
      if (num_done % 100 == 0) {
        double time_now = time.Elapsed();
        ASLP_VLOG(1) << "After " << num_done << " utterances: time elapsed = " << time_now / 60 << " min; processed " << tot_t / time_now
                     << " frames per second.";
      }
      num_done++;
      tot_t += in_store[0].NumRows();
      for (int i = 0; i < num_input; i++) readers[i]->Next();
    }
    for (int i = 1; i < num_input; i++) ASLP_ASSERT(readers[i]->Done());
    ASLP_LOG << "Done " << num_done << " files in " << time.Elapsed() / 60 << "min, (fps " << tot_t / time.Elapsed() << ")";
    if (g_verbose_level >= 1) CuDevice::Instantiate().PrintProfile();
    if (num_done == 0) return -1;
    return 0;
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
  using namespace aslp;
  try {
    const char *usage =
        "Perform forward pass for Latency Control BLSTM through Neural Network.\n"
        "\n"
        "Usage:  aslp-nnet-forward-blstm-lc [options] <model-in> <feature-rspecifier> <feature-wspecifier>\n"
        "e.g.: \n"
        " aslp-nnet-forward-blstm-lc nnet ark:features.ark ark:mlpoutput.ark\n";
    ParseOptions po(usage);
    PdfPriorOptions prior_opts;
    prior_opts.Register(&po);
    int32 chunk_size = 64;
    po.Register("chunk-size", &chunk_size, "---BLSTM--- Latency-controlled BPTT chunk size, must be same with training");
    int32 right_splice = 16;
    po.Register("right-splice", &right_splice, "---BLSTM--- Latency-controlled BPTT right context size, must be same with training");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform in front of main network (in nnet format)");
    bool no_softmax = false;
    po.Register("no-softmax", &no_softmax, "No softmax on MLP output (or remove it if found), the pre-softmax activations will be used as log-likelihoods, log-priors will be subtracted");
    bool apply_log = true;
    po.Register("apply-log", &apply_log, "Transform MLP output to logscale");
    std::string use_gpu = "no";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); exit(1); }
    std::string model_filename = po.GetArg(1), feature_rspecifier = po.GetArg(2), feature_wspecifier = po.GetArg(3);

    // the reference defaults to the CPU here; this engine has none, so "no" (the default) selects a GPU like "yes"
    CuDevice::Instantiate().SelectGpuId(use_gpu == "no" ? "yes" : use_gpu);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);
    Nnet nnet;
    nnet.Read(model_filename);
    if (apply_log && no_softmax) ASLP_ERR << "Cannot use both --apply-log=true --no-softmax=true, use only one of the two!";
    PdfPrior pdf_prior(prior_opts);
    nnet_transf.SetDropoutRetention(1.0);
    nnet.SetDropoutRetention(1.0);
    nnet.SetChunkSize(chunk_size);
    const int32 batch_size = chunk_size + right_splice;

    int64_t tot_t = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    BaseFloatMatrixWriter feature_writer(feature_wspecifier);
    CuMatrix feats, feats_transf, nnet_in, nnet_out, nnet_out_chunk;
    HostMatrix nnet_out_host;
    const int32 feat_dim = nnet.InputDim(), out_dim = nnet.OutputDim();
    Timer time;
    int32 num_done = 0;
    for (; !feature_reader.Done(); feature_reader.Next()) {
      HostMatrix mat = feature_reader.Value();
      std::string utt = feature_reader.Key();
      ASLP_VLOG(2) << "Processing utterance " << num_done + 1 << ", " << utt << ", " << mat.rows << "frm";
      for (float v : mat.data)
        if (!std::isfinite(v)) ASLP_ERR << "NaN or inf found in features for " << utt;
      const int32 in_rows = mat.rows;
      feats = mat;
      nnet_transf.Feedforward(feats, &feats_transf);
      std::vector<int32> reset_flags(1, 1);
      nnet.ResetLstmStreams(reset_flags);
      const int32 num_frames = feats_transf.NumRows();
      const int32 num_chunks = (num_frames - 1) / chunk_size + 1;
      nnet_out.Resize(num_frames, out_dim);
      nnet_in.Resize(batch_size, feat_dim);  // zeroed once per utterance: a short last block keeps the previous block's tail rows (:139-152)
      for (int32 i = 0; i < num_chunks; i++) {
        const int32 offset = i * chunk_size;
        const int32 len = offset + batch_size < num_frames ? batch_size : num_frames - offset;
        const int32 copy_len = offset + chunk_size < num_frames ? chunk_size : num_frames - offset;
        ASLP_ASSERT(len <= batch_size);
        nnet_in.RowRange(0, len).CopyFromMat(feats_transf.RowRange(offset, len));
        nnet.Feedforward(nnet_in, &nnet_out_chunk);
        nnet_out.RowRange(offset, copy_len).CopyFromMat(nnet_out_chunk.RowRange(0, copy_len));
      }
      MinMax st = Stats(nnet_out);
      if (!st.finite) ASLP_ERR << "NaN or inf found in nn-output for " << utt;
      if (apply_log) {
        if (!(st.mn >= 0.0 && st.mx <= 1.0))
          ASLP_WARN << utt << " Applying 'log' to data which don't seem to be probabilities (is there a softmax somwhere?)";
        nnet_out.Add(1e-20);  // avoid log(0)
        nnet_out.ApplyLog();
      }
      if (prior_opts.class_frame_counts != "") {
        if (apply_log) st = Stats(nnet_out);
        if (st.mn >= 0.0 && st.mx <= 1.0)
          ASLP_WARN << utt << " Subtracting log-prior on 'probability-like' data in range [0..1] (Did you forget --no-softmax=true or --apply-log=true ?)";
        pdf_prior.SubtractOnLogpost(&nnet_out);
      }
      nnet_out.CopyToMat(&nnet_out_host);
      for (float v : nnet_out_host.data)
        if (!std::isfinite(v)) ASLP_ERR << "NaN or inf found in final output nn-output for " << utt;
      feature_writer.Write(feature_reader.Key(), nnet_out_host);
      if (num_done % 100 == 0) {
        double time_now = time.Elapsed();
        ASLP_VLOG(1) << "After " << num_done << " utterances: time elapsed = " << time_now / 60 << " min; processed " << tot_t / time_now
                     << " frames per second.";
      }
      num_done++;
      tot_t += in_rows;
    }
    ASLP_LOG << "Done " << num_done << "files in " << time.Elapsed() / 60 << "min, (fps " << tot_t / time.Elapsed() << ")";
    if (g_verbose_level >= 1) CuDevice::Instantiate().PrintProfile();
    if (num_done == 0) return -1;
    return 0;
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
  using namespace aslp;
  try {
    const char *usage =
        "Perform forward pass through Neural Network.\n"
        "\n"
        "Usage:  aslp-nnet-forward-skip [options] <model-in> <feature-rspecifier> <feature-wspecifier>\n"
        "e.g.: \n"
        " aslp-nnet-forward-skip nnet ark:features.ark ark:mlpoutput.ark\n";
    ParseOptions po(usage);
    PdfPriorOptions prior_opts;
    prior_opts.Register(&po);
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform in front of main network (in nnet format)");
    bool no_softmax = false;
    po.Register("no-softmax", &no_softmax, "No softmax on MLP output (or remove it if found), the pre-softmax activations will be used as log-likelihoods, log-priors will be subtracted");
    bool apply_log = true;
    po.Register("apply-log", &apply_log, "Transform MLP output to logscale");
    std::string use_gpu = "no";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    bool add_softmax = false;
    po.Register("add-softmax", &add_softmax, "add softmax calulation for warp-ctc training");
    int32 time_shift = 0;
    po.Register("time-shift", &time_shift, "LSTM : repeat last input frame N-times, discrad N initial output frames.");
    float scale_blank = 0.0;
    po.Register("scale-blank", &scale_blank, "scale the blank posterior for CTC decoding");
    int32 skip_width = 0;
    po.Register("skip-width", &skip_width, "num of frame for one skip(default 0, not use skip)");
    po.Read(argc, argv);
    if (po.NumArgs() != 3) { po.PrintUsage(); exit(1); }
    std::string model_filename = po.GetArg(1), feature_rspecifier = po.GetArg(2), feature_wspecifier = po.GetArg(3);

    // the reference defaults to the CPU here; this engine has none, so "no" (the default) selects a GPU like "yes"
    CuDevice::Instantiate().SelectGpuId(use_gpu == "no" ? "yes" : use_gpu);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);
    Nnet nnet;
    nnet.Read(model_filename);
    if (apply_log && no_softmax) ASLP_ERR << "Cannot use both --apply-log=true --no-softmax=true, use only one of the two!";
    PdfPrior pdf_prior(prior_opts);
    nnet_transf.SetDropoutRetention(1.0);
    nnet.SetDropoutRetention(1.0);

    int64_t tot_t = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    BaseFloatMatrixWriter feature_writer(feature_wspecifier);
    CuMatrix feats, feats_transf, nnet_out, skip_feat, skip_out;
    HostMatrix nnet_out_host;
    Timer time;
    int32 num_done = 0;
    for (; !feature_reader.Done(); feature_reader.Next()) {
      HostMatrix mat = feature_reader.Value();
      std::string utt = feature_reader.Key();
      ASLP_VLOG(2) << "Processing utterance " << num_done + 1 << ", " << utt << ", " << mat.rows << "frm";
      for (float v : mat.data)
        if (!std::isfinite(v)) ASLP_ERR << "NaN or inf found in features for " << utt;
      const int32 in_rows = mat.rows;
      if (time_shift > 0) {  // copy the last frame of LSTM input N-times
        const int32 last_row = mat.rows - 1, cols = mat.cols;
        mat.data.resize((size_t)(mat.rows + time_shift) * cols);
        for (int32 r = last_row + 1; r < last_row + 1 + time_shift; r++)
          std::copy(mat.data.begin() + (size_t)last_row * cols, mat.data.begin() + (size_t)(last_row + 1) * cols, mat.data.begin() + (size_t)r * cols);
        mat.rows += time_shift;
      }
      feats = mat;
      nnet_transf.Feedforward(feats, &feats_transf);
      if (skip_width < 1) ASLP_ERR << "--skip-width must be at least 1 (with the reference's default of 0 no frame is ever processed)";
      for (int32 skip_offset = 0; skip_offset < skip_width; skip_offset++) {
        const int32 num_rows = feats_transf.NumRows();
        if (skip_offset >= num_rows) break;
        const int32 skip_len = (num_rows - 1 - skip_offset) / skip_width + 1;
        std::vector<int32> idx(skip_len);
        for (int32 i = 0; i < skip_len; i++) idx[i] = i * skip_width + skip_offset;
        CuArray<int32> cidx(idx);
        skip_feat.Resize(skip_len, feats_transf.NumCols(), kUndefined);
        skip_feat.CopyRows(feats_transf, cidx);
        std::vector<int32> frame_num_utt(1, skip_len);
        nnet.SetSeqLengths(frame_num_utt);
        nnet.Feedforward(skip_feat, &skip_out);
        if (nnet_out.NumRows() != num_rows || nnet_out.NumCols() != skip_out.NumCols()) nnet_out.Resize(num_rows, skip_out.NumCols());
        for (int32 i = 0; i < skip_len; i++) nnet_out.RowRange(idx[i], 1).CopyFromMat(skip_out.RowRange(i, 1));
      }
      if (add_softmax) {
        CuMatrix tmp_out(nnet_out);
        nnet_out.ApplySoftMaxPerRow(tmp_out);
      }
      MinMax st = Stats(nnet_out);
      if (!st.finite) ASLP_ERR << "NaN or inf found in nn-output for " << utt;
      if (apply_log) {
        if (!(st.mn >= 0.0 && st.mx <= 1.0))
          ASLP_WARN << utt << " Applying 'log' to data which don't seem to be probabilities (is there a softmax somwhere?)";
        nnet_out.Add(1e-20);  // avoid log(0)
        nnet_out.ApplyLog();
      }
      if (scale_blank > 0.0) nnet_out.ColRange(0, 1).Add(-scale_blank);
      if (prior_opts.class_frame_counts != "") {
        if (apply_log || scale_blank > 0.0) st = Stats(nnet_out);
        if (st.mn >= 0.0 && st.mx <= 1.0)
          ASLP_WARN << utt << " Subtracting log-prior on 'probability-like' data in range [0..1] (Did you forget --no-softmax=true or --apply-log=true ?)";
        pdf_prior.SubtractOnLogpost(&nnet_out);
      }
      nnet_out.CopyToMat(&nnet_out_host);
      if (time_shift > 0) {  // remove N first frames of LSTM output
        HostMatrix tmp(nnet_out_host.rows - time_shift, nnet_out_host.cols);
        std::copy(nnet_out_host.data.begin() + (size_t)time_shift * nnet_out_host.cols, nnet_out_host.data.end(), tmp.data.begin());
        nnet_out_host = tmp;
      }
      for (float v : nnet_out_host.data)
        if (!std::isfinite(v)) ASLP_ERR << "NaN or inf found in final output nn-output for " << utt;
      feature_writer.Write(feature_reader.Key(), nnet_out_host);
      if (num_done % 100 == 0) {
        double time_now = time.Elapsed();
        ASLP_VLOG(1) << "After " << num_done << " utterances: time elapsed = " << time_now / 60 << " min; processed " << tot_t / time_now
                     << " frames per second.";
      }
      num_done++;
      tot_t += in_rows;
    }
    ASLP_LOG << "Done " << num_done << " files in " << time.Elapsed() / 60 << "min, (fps " << tot_t / time.Elapsed() << ")";
    if (g_verbose_level >= 1) CuDevice::Instantiate().PrintProfile();
    if (num_done == 0) return -1;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
