// aslp-nnet-train-ctc-streams -- src/aslp-nnetbin/aslp-nnet-train-ctc-streams.cc: the Eesen-CTC twin of
// aslp-nnet-train-warp-ctc-streams (same grouping, padding and learning-rate normalisation): the net ends in a Softmax and
// Ctc::EvalParallel works on the posteriors (ctc-loss.cc:115-227); what run_ctc_*.sh call.
#include "cu-device.h"
#include "data-reader.h"
#include "nnet-nnet.h"
#include "ctc-loss.h"

int main(int argc, char *argv[]) {
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
