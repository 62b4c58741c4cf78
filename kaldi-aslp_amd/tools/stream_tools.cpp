// stream_tools.cpp -- the sequence training tools of src/aslp-nnetbin (lstm-streams, lstm-streams-skip, blstm-streams, blstm-parallel,
// blstm-streams-lc, warp-ctc-streams, ctc-streams, ctc): one entry function per tool (Main_<tool name with _ for ->), linked behind tools/main_stub.cpp into bin/<tool name>.
#include <algorithm>
#include <map>
#include <memory>

#include "ctc-loss.h"
#include "cu-device.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"
#include "warp-ctc.h"

// ======================================================================================================================
// aslp-nnet-train-lstm-streams -- src/aslp-nnetbin/aslp-nnet-train-lstm-streams.cc: multi-stream truncated-BPTT training
// of (projected / CIFG / GRU) LSTM nets fed by SequenceDataReader: batch-size frames of num-stream utterances per step,
// targets delayed by --targets-delay frames, history reset per stream when it takes a new utterance.
int Main_aslp_nnet_train_lstm_streams(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of LSTM training by Stochastic Gradient Descent.\n"
        "This version use pdf-posterior as targets, prepared typically by ali-to-post.\n"
        "The updates are done per-utterance, shuffling options are dummy for compatibility reason.\n"
        "\n"
        "Usage: aslp-nnet-train-lstm-streams [options] <feature-rspecifier> <targets-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-lstm-streams scp:feature.scp ark:posterior.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    NnetDataRandomizerOptions rnd_opts;
    RegisterRandomizerOptions(&rnd_opts, &po);
    SequenceDataReaderOptions read_opts;
    read_opts.Register(&po);
    bool binary = true, crossvalidate = false;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    std::string objective_function = "xent";
    po.Register("objective-function", &objective_function, "Objective function : xent|mse");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    int32 gpu_id = -1;
    po.Register("gpu-id", &gpu_id, "selected gpu id, if negative then select automaticly");
    bool randomize = false;
    po.Register("randomize", &randomize, "Dummy option, for compatibility...");
    int32 report_period = 200;
    po.Register("report-period", &report_period, "Number of sentence for one report log, default(200)");
    int32 dump_interval = 0;
    po.Register("dump-interval", &dump_interval, "---LSTM--- num utts between model dumping [ 0 == disabled ]");
    po.Read(argc, argv);
    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);
    if (gpu_id >= 0) CuDevice::Instantiate().SetGpuId(gpu_id);
    else CuDevice::Instantiate().SelectGpuId(use_gpu);

    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    int64_t total_frames = 0;
    int32 num_done = 0, num_sentence = 0;
    LossItf *loss = NULL;
    if (objective_function == "xent") loss = new Xent;
    else if (objective_function == "mse") loss = new Mse;
    else ASLP_ERR << "Unsupported objective function: " << objective_function;
    Timer time;
    RandomizerMask randomizer_mask(rnd_opts);   // unused by this tool, as in the reference (aslp-nnet-train-lstm-streams.cc:94), but its construction seeds the generator and says so in the log
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    SequenceDataReader reader(feature_rspecifier, targets_rspecifier, read_opts);
    CuMatrix nnet_out, obj_diff, nnet_in;
    std::vector<BaseFloat> frame_mask;
    Posterior nnet_tgt;
    while (!reader.Done()) {
      reader.ReadData(&nnet_in, &nnet_tgt, &frame_mask);
      if (nnet_in.NumRows() == 0) break;  // no usable utterance at all
      std::vector<int32> new_utt_flags = reader.GetNewUttFlags();
      nnet.ResetLstmStreams(new_utt_flags);
      if (!crossvalidate) nnet.Propagate(nnet_in, &nnet_out);
      else nnet.Feedforward(nnet_in, &nnet_out);
      loss->Eval(frame_mask, nnet_out, nnet_tgt, &obj_diff);
      if (!crossvalidate) nnet.Backpropagate(obj_diff, NULL);
      if (g_verbose_level >= 1 && total_frames == 0) {
        ASLP_VLOG(1) << "### After " << total_frames << " frames,";
        ASLP_VLOG(1) << nnet.InfoPropagate();
        if (!crossvalidate) { ASLP_VLOG(1) << nnet.InfoBackPropagate(); ASLP_VLOG(1) << nnet.InfoGradient(); }
      }
      int frame_progress = 0;
      for (BaseFloat m : frame_mask) frame_progress += (int)m;
      total_frames += frame_progress;
      int num_done_progress = 0;
      for (int32 f : new_utt_flags) num_done_progress += f;
      num_done += num_done_progress;
      num_sentence += num_done_progress;
      if (num_sentence >= report_period) { ASLP_LOG << loss->Report(); num_sentence -= report_period; }
      if ((num_done - num_done_progress) / 1000 != (num_done / 1000)) {
        double time_now = time.Elapsed();
        ASLP_VLOG(1) << "After " << num_done << " utterances: time elapsed = " << time_now / 60 << " min; processed "
                     << total_frames / time_now << " frames per second.";
        CuDevice::Instantiate().CheckGpuHealth();
      }
      if (dump_interval > 0 && (num_done - num_done_progress) / dump_interval != (num_done / dump_interval) && !crossvalidate)
        nnet.Write(target_model_filename + "_utt" + std::to_string(num_done), binary);
    }
    if (g_verbose_level >= 1) {
      ASLP_VLOG(1) << "### After " << total_frames << " frames,";
      ASLP_VLOG(1) << nnet.InfoPropagate();
      if (!crossvalidate) { ASLP_VLOG(1) << nnet.InfoBackPropagate(); ASLP_VLOG(1) << nnet.InfoGradient(); }
    }
    if (!crossvalidate) nnet.Write(target_model_filename, binary);
    StreamSync();
    ASLP_LOG << "Done " << num_done << " files, " << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", "
             << (randomize ? "RANDOMIZED" : "NOT-RANDOMIZED") << ", " << time.Elapsed() / 60 << " min, fps" << total_frames / time.Elapsed() << "]";
    ASLP_LOG << loss->Report();  // the reference calls Report() and drops the string (:222); the schedulers need the line
    delete loss;
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-lstm-streams-skip -- src/aslp-nnetbin/aslp-nnet-train-lstm-streams-skip.cc: the multi-stream LSTM trainer
// with frame skipping done as skip-width passes over the data, pass k training on frames k, k + skip-width, ... of every
// utterance (so each utterance is used skip-width times, once per phase).  Own stream options instead of the reader's.
int Main_aslp_nnet_train_lstm_streams_skip(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of LSTM training by Stochastic Gradient Descent.\n"
        "This version use pdf-posterior as targets, prepared typically by ali-to-post.\n"
        "The updates are done per-utterance, shuffling options are dummy for compatibility reason.\n"
        "Attention: one sentence will be devided into N part for skip training\n"
        "\n"
        "Usage: aslp-nnet-train-lstm-streams-skip [options] <feature-rspecifier> <targets-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-lstm-streams-skip scp:feature.scp ark:posterior.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    bool binary = true, crossvalidate = false;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform in Nnet format");
    std::string objective_function = "xent";
    po.Register("objective-function", &objective_function, "Objective function : xent|mse");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    SequenceDataReaderOptions read_opts;
    read_opts.num_stream = 4;
    po.Register("targets-delay", &read_opts.targets_delay, "---LSTM--- BPTT targets delay");
    po.Register("batch-size", &read_opts.batch_size, "---LSTM--- BPTT batch size");
    po.Register("num-stream", &read_opts.num_stream, "---LSTM--- BPTT multi-stream training");
    int32 dump_interval = 0;
    po.Register("dump-interval", &dump_interval, "---LSTM--- num utts between model dumping [ 0 == disabled ]");
    NnetDataRandomizerOptions rnd_opts;
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool randomize = false;
    po.Register("randomize", &randomize, "Dummy option, for compatibility...");
    int32 report_period = 200;
    po.Register("report-period", &report_period, "Number of sentence for one report log, default(200)");
    po.Register("drop-len", &read_opts.drop_len, "if Sentence frame length greater than drop_len,then drop it, default(0, no drop)");
    po.Register("skip-width", &read_opts.skip_width, "num of frame for one skip(default 0, no skip)");
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
    Xent xent;
    if (objective_function != "xent") ASLP_ERR << "Only Support xent objective function, but got" << objective_function;
    Timer time;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    int64_t total_frames = 0;
    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0, num_sentence = 0;
    CuMatrix nnet_in, nnet_out, obj_diff;
    std::vector<BaseFloat> frame_mask;
    Posterior nnet_tgt;
    const int32 passes = read_opts.skip_width > 1 ? read_opts.skip_width : 1;
    for (int32 skip_offset = 0; skip_offset < passes; skip_offset++) {  // every pass reopens both tables (:163-164)
      SequenceDataReader reader(feature_rspecifier, targets_rspecifier, read_opts);
      reader.SetSkipOffset(skip_offset);
      if (nnet_transf.NumComponents() > 0) reader.SetFeatureTransform(&nnet_transf);
      while (!reader.Done()) {
        reader.ReadData(&nnet_in, &nnet_tgt, &frame_mask);
        if (reader.Done()) break;  // all streams exhausted: the reference leaves the loop before touching the net (:238)
        std::vector<int32> new_utt_flags = reader.GetNewUttFlags();
        nnet.ResetLstmStreams(new_utt_flags);
        if (!crossvalidate) nnet.Propagate(nnet_in, &nnet_out);
        else nnet.Feedforward(nnet_in, &nnet_out);
        xent.Eval(frame_mask, nnet_out, nnet_tgt, &obj_diff);
        if (!crossvalidate) nnet.Backpropagate(obj_diff, NULL);
        int32 frame_progress = 0, num_done_progress = 0;
        for (BaseFloat m : frame_mask) frame_progress += (int32)m;
        for (int32 f : new_utt_flags) num_done_progress += f;
        total_frames += frame_progress;
        num_done += num_done_progress;
        num_sentence += num_done_progress;
        if (num_sentence >= report_period) { ASLP_LOG << xent.Report(); num_sentence -= report_period; }
        if (dump_interval > 0 && (num_done - num_done_progress) / dump_interval != num_done / dump_interval && !crossvalidate) {
          char nnet_name[512];
          snprintf(nnet_name, sizeof(nnet_name), "%s_utt%d", target_model_filename.c_str(), num_done);
          nnet.Write(nnet_name, binary);
        }
      }
      num_no_tgt_mat += reader.NumNoTargets();
      num_other_error += reader.NumLengthMismatch();
    }
    if (!crossvalidate) nnet.Write(target_model_filename, binary);
    StreamSync();
    ASLP_LOG << "Done " << num_done << " files, " << num_no_tgt_mat << " with no tgt_mats, " << num_other_error << " with other errors. "
             << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << (randomize ? "RANDOMIZED" : "NOT-RANDOMIZED") << ", "
             << time.Elapsed() / 60 << " min, fps" << total_frames / time.Elapsed() << "]";
    ASLP_LOG << xent.Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-blstm-streams -- src/aslp-nnetbin/aslp-nnet-train-blstm-streams.cc: whole-utterance (B)LSTM training on
// senone targets.  Utterances are grouped num-stream at a time (fewer when frame-limit is hit), padded to the longest, rows
// t*S + s; padding frames carry zero features, empty targets and weight 0; Nnet::SetSeqLengths tells the recurrences where
// each stream ends; the learning rate is divided by the valid frames of the group.
int Main_aslp_nnet_train_blstm_streams(int argc, char *argv[]) {
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
      if (g_verbose_level >= 2 && total_frames == 0) {   // 1st minibatch: from verbose level 2 here (aslp-nnet-train-blstm-streams.cc:289-296), though the lines are VLOG(1)
        ASLP_VLOG(1) << "### After " << total_frames << " frames,";
        ASLP_VLOG(1) << nnet.InfoPropagate();
        if (!crossvalidate) { ASLP_VLOG(1) << nnet.InfoBackPropagate(); ASLP_VLOG(1) << nnet.InfoGradient(); }
      }
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
             << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", "   // (no RANDOMIZED word here: aslp-nnet-train-blstm-streams.cc:329-334)
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

// ======================================================================================================================
// aslp-nnet-train-blstm-parallel -- src/aslp-nnetbin/aslp-nnet-train-blstm-parallel.cc: whole-utterance BLSTM training on up to
// num-stream utterances per step, the plain predecessor of aslp-nnet-train-blstm-streams: no feature transform (the flag is
// accepted and ignored), no frame weights -- padded frames score like any other (LossItf::Eval without weights, :211) -- no
// per-step learn-rate normalisation, utterances whose lengths differ from their targets' are left out.
int Main_aslp_nnet_train_blstm_parallel(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of Neural Network training by Stochastic Gradient Descent.\n"
        "This version use pdf-posterior as targets, prepared typically by ali-to-post.\n"
        "The updates are done per-utternace, shuffling options are dummy for compatibility reason.\n"
        "\n"
        "Usage: aslp-nnet-train-blstm-parallel [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-blstm-parallel scp:feature.scp ark:labels.ark nnet.init nnet.iter1\n";
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
    NnetDataRandomizerOptions rnd_opts;
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool randomize = false;
    po.Register("randomize", &randomize, "Dummy option, for compatibility...");
    int32 report_period = 200;
    po.Register("report-period", &report_period, "Number of sentence for one report log, default(200)");
    int32 drop_len = 0;
    po.Register("drop-len", &drop_len, "if Sentence frame length greater than drop_len,then drop it, default(0, no drop)");
    po.Read(argc, argv);
    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);
    CuDevice::Instantiate().SelectGpuId(use_gpu);

    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    RandomAccessPosteriorReader targets_reader(targets_rspecifier);
    std::unique_ptr<LossItf> loss;
    if (objective_function == "xent") loss.reset(new Xent);
    else if (objective_function == "mse") loss.reset(new Mse);
    else ASLP_ERR << "Unsupported objective function: " << objective_function;
    CuMatrix feats, nnet_out, obj_diff;
    Timer time;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    std::vector<HostMatrix> feats_utt(num_stream);
    std::vector<Posterior> labels_utt(num_stream);
    const int32 feat_dim = nnet.InputDim();
    int64_t total_frames = 0;
    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0, num_sentence = 0;
    while (1) {
      std::vector<int32> frame_num_utt;
      int32 max_frame_num = 0;
      for (; !feature_reader.Done(); feature_reader.Next()) {
        const std::string utt = feature_reader.Key();
        if (!targets_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing targets"; num_no_tgt_mat++; continue; }
        const HostMatrix &mat = feature_reader.Value();
        if (drop_len > 0 && mat.rows > drop_len) {
          ASLP_WARN << utt << ", too long, droped";
          feature_reader.Next();  // (sic) the loop header advances once more: the utterance after a dropped one goes too (:159)
          if (feature_reader.Done()) break;
          continue;
        }
        const Posterior &targets = targets_reader.Value(utt);
        if (mat.rows != (int32)targets.size()) { ASLP_WARN << utt << "feat and the target are not the same length, droped"; continue; }
        if (mat.cols != feat_dim) ASLP_ERR << "feature dim " << mat.cols << " vs network input " << feat_dim;
        if (max_frame_num < mat.rows) max_frame_num = mat.rows;
        feats_utt[frame_num_utt.size()] = mat;
        labels_utt[frame_num_utt.size()] = targets;
        frame_num_utt.push_back(mat.rows);
        if ((int32)frame_num_utt.size() == num_stream || frame_num_utt.size() * (double)max_frame_num > frame_limit) { feature_reader.Next(); break; }
      }
      const int32 S = frame_num_utt.size();
      if (S == 0) break;  // nothing usable left (the reference would run the net on an empty batch)
      HostMatrix feat_mat_host(S * max_frame_num, feat_dim);
      Posterior target_host((size_t)S * max_frame_num);
      for (int s = 0; s < S; s++)
        for (int r = 0; r < frame_num_utt[s]; r++) {
          const size_t row = (size_t)r * S + s;
          std::copy(feats_utt[s].data.begin() + (size_t)r * feat_dim, feats_utt[s].data.begin() + (size_t)(r + 1) * feat_dim,
                    feat_mat_host.data.begin() + row * feat_dim);
          target_host[row] = labels_utt[s][r];
        }
      feats = feat_mat_host;
      nnet.SetSeqLengths(frame_num_utt);
      if (!crossvalidate) nnet.Propagate(feats, &nnet_out);
      else nnet.Feedforward(feats, &nnet_out);
      loss->Eval(nnet_out, target_host, &obj_diff);
      if (!crossvalidate) nnet.Backpropagate(obj_diff, NULL);
      num_done += S;
      total_frames += feat_mat_host.rows;
      num_sentence += S;
      if (num_sentence >= report_period) { ASLP_LOG << loss->Report(); num_sentence -= report_period; }
      if (feature_reader.Done()) break;
    }
    if (!crossvalidate) nnet.Write(target_model_filename, binary);
    StreamSync();
    ASLP_LOG << "Done " << num_done << " files, " << num_no_tgt_mat << " with no tgt_mats, " << num_other_error << " with other errors. "
             << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << time.Elapsed() / 60 << " min, fps"
             << total_frames / time.Elapsed() << "]";
    ASLP_LOG << loss->Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-blstm-streams-lc -- src/aslp-nnetbin/aslp-nnet-train-blstm-streams-lc.cc: latency-controlled BLSTM
// training.  num-stream utterances advance in parallel; every step is a [ (chunk + right) * S x D ] batch, rows t*S + s,
// with a frame mask that is 1 on the chunk frames of live streams; each stream then rewinds by right-splice frames, an
// exhausted stream takes the next utterance and has its history reset (Nnet::ResetLstmStreams).
int Main_aslp_nnet_train_blstm_streams_lc(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of Latency Control BLSTM training by Stochastic Gradient Descent.\n"
        "This version use pdf-posterior as targets, prepared typically by ali-to-post.\n"
        "The updates are done per-utterance, shuffling options are dummy for compatibility reason.\n"
        "\n"
        "Usage: aslp-nnet-train-lstm-streams-lc [options] <feature-rspecifier> <targets-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-lstm-streams-lc scp:feature.scp ark:posterior.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    bool binary = true, crossvalidate = false;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform in Nnet format");
    std::string objective_function = "xent";
    po.Register("objective-function", &objective_function, "Objective function : xent|mse");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    int32 chunk_size = 64;
    po.Register("chunk-size", &chunk_size, "---BLSTM--- Latency-controlled BPTT chunk size");
    int32 right_splice = 16;
    po.Register("right-splice", &right_splice, "---BLSTM--- Latency-controlled BPTT right context size");
    int32 num_stream = 4;
    po.Register("num-stream", &num_stream, "---LSTM--- BPTT multi-stream training");
    int32 dump_interval = 0;
    po.Register("dump-interval", &dump_interval, "---LSTM--- num utts between model dumping [ 0 == disabled ]");
    NnetDataRandomizerOptions rnd_opts;  // dummy randomizer options, to make the tool compatible with standard scripts
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool randomize = false;
    po.Register("randomize", &randomize, "Dummy option, for compatibility...");
    int32 report_period = 200;
    po.Register("report-period", &report_period, "Number of sentence for one report log, default(200)");
    int32 drop_len = 0;
    po.Register("drop-len", &drop_len, "if Sentence frame length greater than drop_len,then drop it, default(0, no drop)");
    po.Read(argc, argv);
    const int32 batch_size = chunk_size + right_splice;
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
    nnet.SetChunkSize(chunk_size);

    int64_t total_frames = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    RandomAccessPosteriorReader target_reader(targets_rspecifier);
    Xent xent;
    Mse mse;
    Timer time;
    RandomizerMask randomizer_mask(rnd_opts);   // unused by this tool, as in the reference (aslp-nnet-train-blstm-streams-lc.cc:151), but its construction seeds the generator and says so in the log
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0, num_sentence = 0;
    std::vector<std::string> keys(num_stream);
    std::vector<HostMatrix> feats(num_stream);
    std::vector<Posterior> targets(num_stream);
    std::vector<int32> curt(num_stream, 0), lent(num_stream, 0), new_utt_flags(num_stream, 0);
    const int32 feat_dim = nnet.InputDim();
    std::vector<BaseFloat> frame_mask((size_t)batch_size * num_stream, 0.0f);
    HostMatrix feat(batch_size * num_stream, feat_dim);
    Posterior target((size_t)batch_size * num_stream);
    CuMatrix cu_in, feat_transf, cu_feat, nnet_out, obj_diff;

    while (1) {
      for (int s = 0; s < num_stream; s++) {  // feed exhausted streams with a new utterance
        if (curt[s] < lent[s]) { new_utt_flags[s] = 0; continue; }
        while (!feature_reader.Done()) {
          const std::string key = feature_reader.Key();
          const HostMatrix &mat = feature_reader.Value();
          if (drop_len > 0 && mat.rows > drop_len) { ASLP_WARN << key << ", too long, droped"; feature_reader.Next(); continue; }
          cu_in = mat;
          nnet_transf.Feedforward(cu_in, &feat_transf);
          if (!target_reader.HasKey(key)) { ASLP_WARN << key << ", missing targets"; num_no_tgt_mat++; feature_reader.Next(); continue; }
          const Posterior &tgt = target_reader.Value(key);
          if (feat_transf.NumRows() != (int32)tgt.size()) {
            ASLP_WARN << key << ", length miss-match between feats and targets, skip";
            num_other_error++;
            feature_reader.Next();
            continue;
          }
          keys[s] = key;
          feat_transf.CopyToMat(&feats[s]);
          targets[s] = tgt;
          curt[s] = 0;
          lent[s] = feats[s].rows;
          new_utt_flags[s] = 1;
          feature_reader.Next();
          break;
        }
      }
      int done = 1;
      for (int s = 0; s < num_stream; s++)
        if (curt[s] < lent[s]) done = 0;
      if (done) break;
      // fill a multi-stream batch: mask 1 = chunk frame of a live stream; padding rows are zero features with the
      // utterance's last target (masked out anyway); a stream that never got an utterance has no last target
      for (int t = 0; t < batch_size; t++) {
        for (int s = 0; s < num_stream; s++) {
          const size_t row = (size_t)t * num_stream + s;
          if (curt[s] < lent[s]) {
            frame_mask[row] = t >= chunk_size ? 0.0f : 1.0f;
            target[row] = targets[s][curt[s]];
            std::copy(feats[s].data.begin() + (size_t)curt[s] * feat_dim, feats[s].data.begin() + (size_t)(curt[s] + 1) * feat_dim,
                      feat.data.begin() + row * feat_dim);
          } else {
            frame_mask[row] = 0.0f;
            if (lent[s] > 0) target[row] = targets[s][lent[s] - 1];
            else target[row].clear();
            std::fill(feat.data.begin() + row * feat_dim, feat.data.begin() + (row + 1) * feat_dim, 0.0f);
          }
          curt[s]++;
        }
      }
      for (int s = 0; s < num_stream; s++) curt[s] = curt[s] - right_splice;
      nnet.ResetLstmStreams(new_utt_flags);
      cu_feat = feat;
      if (!crossvalidate) nnet.Propagate(cu_feat, &nnet_out);
      else nnet.Feedforward(cu_feat, &nnet_out);
      if (objective_function == "xent") xent.Eval(frame_mask, nnet_out, target, &obj_diff);
      else ASLP_ERR << "Unknown objective function code : " << objective_function;
      if (!crossvalidate) nnet.Backpropagate(obj_diff, NULL);
      if (g_verbose_level >= 1 && total_frames == 0) {
        ASLP_VLOG(1) << "### After " << total_frames << " frames,";
        ASLP_VLOG(1) << nnet.InfoPropagate();
        if (!crossvalidate) { ASLP_VLOG(1) << nnet.InfoBackPropagate(); ASLP_VLOG(1) << nnet.InfoGradient(); }
      }
      int frame_progress = 0;
      for (BaseFloat m : frame_mask) frame_progress += (int)m;
      total_frames += frame_progress;
      int num_done_progress = 0;
      for (int32 f : new_utt_flags) num_done_progress += f;
      num_done += num_done_progress;
      num_sentence += num_done_progress;
      if (num_sentence >= report_period) { ASLP_LOG << xent.Report(); num_sentence -= report_period; }
      if ((num_done - num_done_progress) / 10 != (num_done / 10)) {
        double time_now = time.Elapsed();
        ASLP_VLOG(1) << "After " << num_done << " utterances: time elapsed = " << time_now / 60 << " min; processed "
                     << total_frames / time_now << " frames per second.";
        CuDevice::Instantiate().CheckGpuHealth();
      }
      if (dump_interval > 0 && (num_done - num_done_progress) / dump_interval != (num_done / dump_interval) && !crossvalidate)
        nnet.Write(target_model_filename + "_utt" + std::to_string(num_done), binary);
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
    else ASLP_ERR << "Unknown objective function code : " << objective_function;
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-warp-ctc-streams -- src/aslp-nnetbin/aslp-nnet-train-warp-ctc-streams.cc: CTC training on whole
// utterances, num-stream at a time (fewer when frame-limit is hit), padded to the longest of the group, rows t*S + s;
// learning rate divided by the number of valid frames of the group; WarpCtc loss on the pre-softmax activations.
int Main_aslp_nnet_train_warp_ctc_streams(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of CTC training by SGD.\n"
        "The updates are done per-utterance and by processing multiple utterances in parallel.\n"
        "\n"
        "Usage: aslp-nnet-train-warp-ctc-streams [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        "aslp-nnet-train-warp-ctc-streams scp:feature.scp ark:labels.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    bool binary = true, crossvalidate = false;
    po.Register("binary", &binary, "Write model  in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (no backpropagation)");
    int32 num_stream = 5;
    po.Register("num-stream", &num_stream, "Number of sequences processed in parallel");
    double frame_limit = 100000;
    po.Register("frame-limit", &frame_limit, "Max number of frames to be processed");
    NnetDataRandomizerOptions rnd_opts;  // dummy randomizer options, to make the tool compatible with standard scripts
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool randomize = false;
    po.Register("randomize", &randomize, "Dummy option, for compatibility...");
    int32 report_step = 100;
    po.Register("report-step", &report_step, "Step (number of sequences) for status reporting");
    int32 report_period = 200;
    po.Register("report-period", &report_period, "Number of sentence for one report log, default(200)");
    int32 drop_len = 0;
    po.Register("drop-len", &drop_len, "if Sentence frame length greater than drop_len,then drop it, default(0, no drop)");
    int32 skip_width = 0;
    po.Register("skip-width", &skip_width, "num of frame for one skip(default 0, not use skip)");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    po.Read(argc, argv);
    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);

    CuDevice::Instantiate().SelectGpuId(use_gpu);
    Nnet net;
    net.Read(model_filename);
    net.SetTrainOptions(trn_opts);
    const float norm_lr = trn_opts.learn_rate;
    int64_t total_frames = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    RandomAccessInt32VectorReader targets_reader(targets_rspecifier);
    WarpCtc ctc;
    ctc.SetUseGpu(true);  // the loss runs on the device whatever --use-gpu says: there is no host CTC in this engine
    ctc.SetReportStep(report_step);
    CuMatrix net_in, net_out, obj_diff;
    Timer time;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    std::vector<HostMatrix> feats_utt(num_stream);
    std::vector<std::vector<int32>> labels_utt(num_stream);
    std::vector<std::string> key_utt(num_stream);
    const int32 feat_dim = net.InputDim();
    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0, num_sentence = 0;
    while (1) {
      std::vector<int32> frame_num_utt;
      int32 sequence_index = 0, max_frame_num = 0;
      for (; !feature_reader.Done(); feature_reader.Next()) {
        std::string utt = feature_reader.Key();
        if (!targets_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing targets"; num_no_tgt_mat++; continue; }
        const HostMatrix &raw_mat = feature_reader.Value();
        if (drop_len > 0 && raw_mat.rows > drop_len) { ASLP_WARN << utt << ", too long, droped"; continue; }
        HostMatrix mat;
        if (skip_width > 1) {
          const int32 skip_len = (raw_mat.rows - 1) / skip_width + 1;
          mat.Resize(skip_len, raw_mat.cols);
          for (int32 i = 0; i < skip_len; i++)
            std::copy(raw_mat.data.begin() + (size_t)i * skip_width * raw_mat.cols, raw_mat.data.begin() + (size_t)(i * skip_width + 1) * raw_mat.cols,
                      mat.data.begin() + (size_t)i * mat.cols);
        } else {
          mat = raw_mat;
        }
        if (max_frame_num < mat.rows) max_frame_num = mat.rows;
        feats_utt[sequence_index] = mat;
        labels_utt[sequence_index] = targets_reader.Value(utt);
        key_utt[sequence_index] = utt;
        frame_num_utt.push_back(mat.rows);
        sequence_index++;
        if ((int32)frame_num_utt.size() == num_stream || frame_num_utt.size() * (double)max_frame_num > frame_limit) { feature_reader.Next(); break; }
      }
      const int32 cur_sequence_num = frame_num_utt.size();
      if (cur_sequence_num == 0) break;  // nothing usable left (the reference would push an empty batch through the net)
      int32 num_valid_frame = 0;
      HostMatrix feat_mat_host(cur_sequence_num * max_frame_num, feat_dim);
      for (int s = 0; s < cur_sequence_num; s++) {
        if (feats_utt[s].cols != feat_dim) ASLP_ERR << key_utt[s] << ": feature dim " << feats_utt[s].cols << " vs network input " << feat_dim;
        for (int r = 0; r < frame_num_utt[s]; r++)
          std::copy(feats_utt[s].data.begin() + (size_t)r * feat_dim, feats_utt[s].data.begin() + (size_t)(r + 1) * feat_dim,
                    feat_mat_host.data.begin() + ((size_t)r * cur_sequence_num + s) * feat_dim);
        num_valid_frame += frame_num_utt[s];
      }
      net.SetSeqLengths(frame_num_utt);
      trn_opts.learn_rate = norm_lr / num_valid_frame;
      net.SetTrainOptions(trn_opts);
      net_in = feat_mat_host;
      if (!crossvalidate) net.Propagate(net_in, &net_out);
      else net.Feedforward(net_in, &net_out);
      std::vector<std::string> keys(key_utt.begin(), key_utt.begin() + cur_sequence_num);
      std::vector<std::vector<int32>> labels(labels_utt.begin(), labels_utt.begin() + cur_sequence_num);
      ctc.Eval(keys, frame_num_utt, net_out, labels, &obj_diff);
      ctc.ErrorRate(frame_num_utt, net_out, labels);
      if (!crossvalidate) net.Backpropagate(obj_diff, NULL);
      num_done += cur_sequence_num;
      total_frames += feat_mat_host.rows;
      num_sentence += cur_sequence_num;
      if (num_sentence >= report_period) { ASLP_LOG << ctc.Report(); num_sentence -= report_period; }
      if (feature_reader.Done()) break;
    }
    if (!crossvalidate) ASLP_LOG << net.InfoGradient();
    if (!crossvalidate) net.Write(target_model_filename, binary);
    StreamSync();
    ASLP_LOG << "Done " << num_done << " files, " << num_no_tgt_mat << " with no targets, " << num_other_error << " with other errors. "
             << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << time.Elapsed() / 60 << " min, fps"
             << total_frames / time.Elapsed() << "]";
    ASLP_LOG << ctc.Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-ctc-streams -- src/aslp-nnetbin/aslp-nnet-train-ctc-streams.cc: the Eesen-CTC twin of
// aslp-nnet-train-warp-ctc-streams (same grouping, padding and learning-rate normalisation): the net ends in a Softmax and
// Ctc::EvalParallel works on the posteriors (ctc-loss.cc:115-227); what run_ctc_*.sh call.
int Main_aslp_nnet_train_ctc_streams(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of CTC training by SGD.\n"
        "The updates are done per-utterance and by processing multiple utterances in parallel.\n"
        "\n"
        "Usage: aslp-nnet-train-ctc-streams [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        "aslp-nnet-train-ctc-streams scp:feature.scp ark:labels.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    bool binary = true, crossvalidate = false;
    po.Register("binary", &binary, "Write model  in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (no backpropagation)");
    int32 num_stream = 5;
    po.Register("num-stream", &num_stream, "Number of sequences processed in parallel");
    double frame_limit = 100000;
    po.Register("frame-limit", &frame_limit, "Max number of frames to be processed");
    NnetDataRandomizerOptions rnd_opts;  // dummy randomizer options, to make the tool compatible with standard scripts
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool randomize = false;
    po.Register("randomize", &randomize, "Dummy option, for compatibility...");
    int32 report_step = 100;
    po.Register("report-step", &report_step, "Step (number of sequences) for status reporting");
    int32 report_period = 200;
    po.Register("report-period", &report_period, "Number of sentence for one report log, default(200)");
    int32 drop_len = 0;
    po.Register("drop-len", &drop_len, "if Sentence frame length greater than drop_len,then drop it, default(0, no drop)");
    int32 skip_width = 0;
    po.Register("skip-width", &skip_width, "num of frame for one skip(default 0, not use skip)");
    std::string use_gpu = "yes";  // not registered as an option in this tool (:66)
    po.Read(argc, argv);
    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);

    CuDevice::Instantiate().SelectGpuId(use_gpu);
    Nnet net;
    net.Read(model_filename);
    net.SetTrainOptions(trn_opts);
    const float norm_lr = trn_opts.learn_rate;
    int64_t total_frames = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    RandomAccessInt32VectorReader targets_reader(targets_rspecifier);
    Ctc ctc;
    ctc.SetReportStep(report_step);
    CuMatrix net_in, net_out, obj_diff;
    Timer time;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    std::vector<HostMatrix> feats_utt(num_stream);
    std::vector<std::vector<int32>> labels_utt(num_stream);
    std::vector<std::string> key_utt(num_stream);
    const int32 feat_dim = net.InputDim();
    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0, num_sentence = 0;
    while (1) {
      std::vector<int32> frame_num_utt;
      int32 sequence_index = 0, max_frame_num = 0;
      for (; !feature_reader.Done(); feature_reader.Next()) {
        std::string utt = feature_reader.Key();
        if (!targets_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing targets"; num_no_tgt_mat++; continue; }
        const HostMatrix &raw_mat = feature_reader.Value();
        if (drop_len > 0 && raw_mat.rows > drop_len) { ASLP_WARN << utt << ", too long, droped"; continue; }
        HostMatrix mat;
        if (skip_width > 1) {
          const int32 skip_len = (raw_mat.rows - 1) / skip_width + 1;
          mat.Resize(skip_len, raw_mat.cols);
          for (int32 i = 0; i < skip_len; i++)
            std::copy(raw_mat.data.begin() + (size_t)i * skip_width * raw_mat.cols, raw_mat.data.begin() + (size_t)(i * skip_width + 1) * raw_mat.cols,
                      mat.data.begin() + (size_t)i * mat.cols);
        } else {
          mat = raw_mat;
        }
        if (max_frame_num < mat.rows) max_frame_num = mat.rows;
        feats_utt[sequence_index] = mat;
        labels_utt[sequence_index] = targets_reader.Value(utt);
        key_utt[sequence_index] = utt;
        frame_num_utt.push_back(mat.rows);
        sequence_index++;
        if ((int32)frame_num_utt.size() == num_stream || frame_num_utt.size() * (double)max_frame_num > frame_limit) { feature_reader.Next(); break; }
      }
      const int32 cur_sequence_num = frame_num_utt.size();
      if (cur_sequence_num == 0) break;  // nothing usable left (the reference would push an empty batch through the net)
      int32 num_valid_frame = 0;
      HostMatrix feat_mat_host(cur_sequence_num * max_frame_num, feat_dim);
      for (int s = 0; s < cur_sequence_num; s++) {
        if (feats_utt[s].cols != feat_dim) ASLP_ERR << key_utt[s] << ": feature dim " << feats_utt[s].cols << " vs network input " << feat_dim;
        for (int r = 0; r < frame_num_utt[s]; r++)
          std::copy(feats_utt[s].data.begin() + (size_t)r * feat_dim, feats_utt[s].data.begin() + (size_t)(r + 1) * feat_dim,
                    feat_mat_host.data.begin() + ((size_t)r * cur_sequence_num + s) * feat_dim);
        num_valid_frame += frame_num_utt[s];
      }
      net.SetSeqLengths(frame_num_utt);
      trn_opts.learn_rate = norm_lr / num_valid_frame;
      net.SetTrainOptions(trn_opts);
      net_in = feat_mat_host;
      if (!crossvalidate) net.Propagate(net_in, &net_out);
      else net.Feedforward(net_in, &net_out);
      std::vector<std::string> keys(key_utt.begin(), key_utt.begin() + cur_sequence_num);
      std::vector<std::vector<int32>> labels(labels_utt.begin(), labels_utt.begin() + cur_sequence_num);
      ctc.EvalParallel(keys, frame_num_utt, net_out, labels, &obj_diff);
      ctc.ErrorRateMSeq(frame_num_utt, net_out, labels);
      if (!crossvalidate) net.Backpropagate(obj_diff, NULL);
      num_done += cur_sequence_num;
      total_frames += feat_mat_host.rows;
      num_sentence += cur_sequence_num;
      if (num_sentence >= report_period) { ASLP_LOG << ctc.Report(); num_sentence -= report_period; }
      if (feature_reader.Done()) break;
    }
    if (!crossvalidate) ASLP_LOG << net.InfoGradient();
    if (!crossvalidate) net.Write(target_model_filename, binary);
    StreamSync();
    ASLP_LOG << "Done " << num_done << " files, " << num_no_tgt_mat << " with no targets, " << num_other_error << " with other errors. "
             << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << time.Elapsed() / 60 << " min, fps"
             << total_frames / time.Elapsed() << "]";
    ASLP_LOG << ctc.Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-ctc -- src/aslp-nnetbin/aslp-nnet-train-ctc.cc: CTC training (Eesen objective on the net's softmax output)
// one utterance per update; --token-symbol-table prints the greedy hypothesis of every utterance in token names.
int Main_aslp_nnet_train_ctc(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of CTC training by SGD.\n"
        "The updates are done per-utternace and by processing a single utterance at one time.\n"
        "\n"
        "Usage: aslp-nnet-train-ctc [options] <feature-rspecifier> <labels-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        "aslp-nnet-train-ctc scp:feature.scp ark:labels.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    bool binary = true, crossvalidate = false;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    std::string token_syms_filename;
    po.Register("token-symbol-table", &token_syms_filename, "Symbol table for tokens [for debug output]");
    int32 report_step = 100;
    po.Register("report-step", &report_step, "Step (number of sequences) for status reporting");
    int32 drop_len = 0;
    po.Register("drop-len", &drop_len, "if Sentence frame length greater than drop_len,then drop it, default(0, no drop)");
    po.Read(argc, argv);
    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);

    // token table in OpenFst's text form: "<symbol> <id>" per line
    std::map<int32, std::string> token_syms;
    if (token_syms_filename != "") {
      bool binary_in;
      Input ki;
      if (!ki.Open(token_syms_filename, &binary_in)) ASLP_ERR << "Could not read symbol table from file " << token_syms_filename;
      std::string sym;
      int64_t id;
      while (ki.Stream() >> sym >> id) token_syms[(int32)id] = sym;
      if (token_syms.empty()) ASLP_ERR << "Could not read symbol table from file " << token_syms_filename;
    }
    CuDevice::Instantiate().SelectGpuId("yes");  // the reference has --use-gpu commented out and always asks for a GPU

    Nnet net;
    net.Read(model_filename);
    net.SetTrainOptions(trn_opts);
    int64_t total_frames = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    RandomAccessInt32VectorReader targets_reader(targets_rspecifier);
    Ctc ctc;
    ctc.SetReportStep(report_step);
    CuMatrix feats, net_out, obj_diff;
    Timer time;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0;
    for (; !feature_reader.Done(); feature_reader.Next()) {
      const std::string utt = feature_reader.Key();
      if (!targets_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing targets"; num_no_tgt_mat++; continue; }
      const HostMatrix &mat = feature_reader.Value();
      if (drop_len > 0 && mat.rows > drop_len) { ASLP_WARN << utt << ", too long, droped"; continue; }
      const std::vector<int32> &targets = targets_reader.Value(utt);
      feats = mat;
      if (!crossvalidate) net.Propagate(feats, &net_out);
      else net.Feedforward(feats, &net_out);
      ctc.Eval(net_out, targets, &obj_diff);
      float err = 0.0;
      std::vector<int32> hyp;
      ctc.ErrorRate(net_out, targets, &err, &hyp);
      if (!crossvalidate) net.Backpropagate(obj_diff, NULL);
      if (!token_syms.empty()) {
        std::cerr << utt << ' ';
        for (int32 h : hyp) {
          auto it = token_syms.find(h);
          if (it == token_syms.end()) ASLP_ERR << "Token-id " << h << " not in symbol table.";
          std::cerr << it->second << " ";
        }
        std::cerr << '\n';
      }
      num_done++;
      total_frames += mat.rows;
    }
    if (!crossvalidate) ASLP_LOG << net.InfoGradient();
    if (!crossvalidate) net.Write(target_model_filename, binary);
    StreamSync();
    ASLP_LOG << "Done " << num_done << " files, " << num_no_tgt_mat << " with no targets, " << num_other_error << " with other errors. "
             << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << time.Elapsed() / 60 << " min, fps"
             << total_frames / time.Elapsed() << "]";
    ASLP_LOG << ctc.Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
