// aslp-nnet-train-blstm-streams -- src/aslp-nnetbin/aslp-nnet-train-blstm-streams.cc: whole-utterance (B)LSTM training on
// senone targets.  Utterances are grouped num-stream at a time (fewer when frame-limit is hit), padded to the longest, rows
// t*S + s; padding frames carry zero features, empty targets and weight 0; Nnet::SetSeqLengths tells the recurrences where
// each stream ends; the learning rate is divided by the valid frames of the group.
#include <algorithm>

#include "cu-device.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"

int main(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of senones training by SGD.\n"
        "The updates are done per-utternace and by processing multiple utterances in parallel.\n"
        "\n"
        "Usage: aslp-nnet-train-blstm-streams [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-blstm-streams scp:feature.scp ark:labels.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    bool binary = true, crossvalidate = false;
    po.Register("binary", &binary, "Write model  in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (no backpropagation)");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform in Nnet format");
    int32 length_tolerance = 5;
    po.Register("length-tolerance", &length_tolerance, "Allowed length difference of features/targets (frames)");
    std::string frame_weights;
    po.Register("frame-weights", &frame_weights, "Per-frame weights to scale gradients (frame selection/weighting).");
    std::string objective_function = "xent";
    po.Register("objective-function", &objective_function, "Objective function : xent|mse");
    int32 num_stream = 4;
    po.Register("num-stream", &num_stream, "Number of sequences processed in parallel");
    double frame_limit = 100000;
    po.Register("frame-limit", &frame_limit, "Max number of frames to be processed");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    NnetDataRandomizerOptions rnd_opts;  // dummy randomizer options, to make the tool compatible with standard scripts
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool randomize = false;
    po.Register("randomize", &randomize, "Dummy option, for compatibility...");
    int32 report_period = 200;
    po.Register("report-period", &report_period, "Number of sentence for one report log, default(200)");
    int32 drop_len = 0;
    po.Register("drop-len", &drop_len, "if Sentence frame length greater than drop_len,then drop it, default(0, no drop)");
    int32 skip_width = 0;
    po.Register("skip-width", &skip_width, "num of frame for one skip(default 0, not use skip)");
    po.Read(argc, argv);
    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);
    CuDevice::Instantiate().SelectGpuId(use_gpu);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);
    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    const float norm_lr = trn_opts.learn_rate;
    int64_t total_frames = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    RandomAccessPosteriorReader targets_reader(targets_rspecifier);
    RandomAccessBaseFloatVectorReader weights_reader;
    if (frame_weights != "" && !weights_reader.Open(frame_weights)) ASLP_ERR << "cannot open " << frame_weights;
    Xent xent;
    Mse mse;
    CuMatrix feats, feats_transf, nnet_out, obj_diff;
    Timer time;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    std::vector<HostMatrix> feats_utt(num_stream);
    std::vector<Posterior> labels_utt(num_stream);
    std::vector<std::vector<BaseFloat>> weights_utt(num_stream);
    const int32 feat_dim = nnet_transf.NumComponents() > 0 ? nnet_transf.InputDim() : nnet.InputDim();
    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0, num_sentence = 0;
    while (1) {
      std::vector<int32> frame_num_utt;
      int32 sequence_index = 0, max_frame_num = 0;
      for (; !feature_reader.Done(); feature_reader.Next()) {
        std::string utt = feature_reader.Key();
        if (!targets_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing targets"; num_no_tgt_mat++; continue; }
        const HostMatrix &raw_mat = feature_reader.Value();
        if (drop_len > 0 && raw_mat.rows > drop_len) {
          ASLP_WARN << utt << ", too long, droped";
          feature_reader.Next();  // (sic) advanced here and by the loop header: the utterance after a dropped one is skipped too (:143)
          if (feature_reader.Done()) break;
          continue;
        }
        const Posterior &raw_targets = targets_reader.Value(utt);
        HostMatrix mat;
        Posterior targets;
        if (skip_width > 1) {
          const int32 skip_len = (raw_mat.rows - 1) / skip_width + 1;
          mat.Resize(skip_len, raw_mat.cols);
          targets.resize(skip_len);
          for (int32 i = 0; i < skip_len; i++) {
            std::copy(raw_mat.data.begin() + (size_t)i * skip_width * raw_mat.cols, raw_mat.data.begin() + (size_t)(i * skip_width + 1) * raw_mat.cols,
                      mat.data.begin() + (size_t)i * mat.cols);
            targets[i] = raw_targets[i * skip_width];
          }
        } else {
          mat = raw_mat;
          targets = raw_targets;
        }
        std::vector<BaseFloat> weights;
        if (frame_weights != "") weights = weights_reader.Value(utt).data;
        else weights.assign(mat.rows, 1.0f);
        {
          const int32 lens[3] = {mat.rows, (int32)targets.size(), (int32)weights.size()};
          const int32 mn = *std::min_element(lens, lens + 3), mx = *std::max_element(lens, lens + 3);
          if (mx - mn < length_tolerance) {
            if (mat.rows != mn) { mat.data.resize((size_t)mn * mat.cols); mat.rows = mn; }
            if ((int32)targets.size() != mn) targets.resize(mn);
            if ((int32)weights.size() != mn) weights.resize(mn);
          } else {
            ASLP_WARN << utt << ", length mismatch of targets " << targets.size() << " and features " << mat.rows;
            num_other_error++;
            continue;
          }
        }
        if (max_frame_num < mat.rows) max_frame_num = mat.rows;
        feats_utt[sequence_index] = mat;
        labels_utt[sequence_index] = targets;
        weights_utt[sequence_index] = weights;
        frame_num_utt.push_back(mat.rows);
        sequence_index++;
        if ((int32)frame_num_utt.size() == num_stream || frame_num_utt.size() * (double)max_frame_num > frame_limit) { feature_reader.Next(); break; }
      }
      const int32 S = frame_num_utt.size();
      if (S == 0) break;
      HostMatrix feat_mat_host(S * max_frame_num, feat_dim);
      Posterior target_host((size_t)S * max_frame_num);
      std::vector<BaseFloat> weight_host((size_t)S * max_frame_num, 0.0f);
      int32 num_valid_frame = 0;
      for (int s = 0; s < S; s++) {
        if (feats_utt[s].cols != feat_dim) ASLP_ERR << "feature dim " << feats_utt[s].cols << " vs network input " << feat_dim;
        for (int r = 0; r < frame_num_utt[s]; r++) {
          const size_t row = (size_t)r * S + s;
          std::copy(feats_utt[s].data.begin() + (size_t)r * feat_dim, feats_utt[s].data.begin() + (size_t)(r + 1) * feat_dim,
                    feat_mat_host.data.begin() + row * feat_dim);
          target_host[row] = labels_utt[s][r];
          weight_host[row] = weights_utt[s][r];
        }
        num_valid_frame += frame_num_utt[s];
      }
      feats = feat_mat_host;
      nnet_transf.Feedforward(feats, &feats_transf);
      nnet.SetSeqLengths(frame_num_utt);
      trn_opts.learn_rate = norm_lr / num_valid_frame;
      nnet.SetTrainOptions(trn_opts);
      if (!crossvalidate) nnet.Propagate(feats_transf, &nnet_out);
      else nnet.Feedforward(feats_transf, &nnet_out);
      if (objective_function == "xent") xent.Eval(weight_host, nnet_out, target_host, &obj_diff);
      else if (objective_function == "mse") mse.Eval(weight_host, nnet_out, target_host, &obj_diff);
      else ASLP_ERR << "Unknown objective function code : " << objective_function;
      if (!crossvalidate) nnet.Backpropagate(obj_diff, NULL);
      num_done += S;
      total_frames += feats_transf.NumRows();
      num_sentence += S;
      if (num_sentence >= report_period) {
        if (objective_function == "xent") ASLP_LOG << xent.Report();
        else if (objective_function == "mse") ASLP_LOG << mse.Report();
        num_sentence -= report_period;
      }
      if (feature_reader.Done()) break;
    }
    if (g_verbose_level >= 1) {
      ASLP_VLOG(1) << "### After " << total_frames << " frames,";
      ASLP_VLOG(1) << nnet.InfoPropagate();
      if (!crossvalidate) { ASLP_VLOG(1) << nnet.InfoBackPropagate(); ASLP_VLOG(1) << nnet.InfoGradient(); }
    }
    if (!crossvalidate) nnet.Write(target_model_filename, binary);
    StreamSync();
    ASLP_LOG << "Done " << num_done << " files, " << num_no_tgt_mat << " with no tgt_mats, " << num_other_error << " with other errors. "
             << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << (randomize ? "RANDOMIZED" : "NOT-RANDOMIZED") << ", "
             << time.Elapsed() / 60 << " min, fps" << total_frames / time.Elapsed() << "]";
    if (objective_function == "xent") ASLP_LOG << xent.Report();
    else if (objective_function == "mse") ASLP_LOG << mse.Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
