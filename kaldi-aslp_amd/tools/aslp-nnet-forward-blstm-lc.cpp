// aslp-nnet-forward-blstm-lc -- src/aslp-nnetbin/aslp-nnet-forward-blstm-lc.cc: latency-controlled BLSTM inference.  Every
// utterance goes through the net in chunks of chunk-size frames followed by right-splice frames of look-ahead (one stream,
// history reset at the utterance start); only the chunk part of each output block is kept.
#include <cmath>

#include "cu-device.h"
#include "kaldi-table.h"
#include "nnet-nnet.h"
#include "nnet-pdf-prior.h"
#include "parse-options.h"

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

int main(int argc, char *argv[]) {
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
