// aslp-nnet-train-perutt -- src/aslp-nnetbin/aslp-nnet-train-perutt.cc: one update per utterance (the FSMN recipes,
// run_cfsmn.sh), learning rate divided by 1024 (:203), optional feature transform / frame weights / length tolerance.
#include <algorithm>

#include "cu-device.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"

int main(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of Neural Network training by Stochastic Gradient Descent.\n"
        "This version use pdf-posterior as targets, prepared typically by ali-to-post.\n"
        "The updates are done per-utterance, shuffling options are dummy for compatibility reason.\n"
        "\n"
        "Usage:  aslp-nnet-train-perutt [options] <feature-rspecifier> <targets-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-perutt scp:feature.scp ark:posterior.ark nnet.init nnet.iter1\n";
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
    int32 length_tolerance = 5;
    po.Register("length-tolerance", &length_tolerance, "Allowed length difference of features/targets (frames)");
    std::string frame_weights;
    po.Register("frame-weights", &frame_weights, "Per-frame weights to scale gradients (frame selection/weighting).");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    NnetDataRandomizerOptions rnd_opts;  // dummy randomizer options, to make the tool compatible with standard scripts
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool randomize = false;
    po.Register("randomize", &randomize, "Dummy option, for compatibility...");
    int32 report_period = 60000;
    po.Register("report-period", &report_period, "Number of frames for one report log, default(60000)");
    int32 drop_len = -1;
    po.Register("drop-len", &drop_len, "if sentence frame length greater than drop_len,if negative no drop");
    int32 gpu_id = -1;
    po.Register("gpu-id", &gpu_id, "selected gpu id, if negative then select automaticly");
    po.Read(argc, argv);
    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);
    if (gpu_id >= 0) CuDevice::Instantiate().SetGpuId(gpu_id);
    else CuDevice::Instantiate().SelectGpuId(use_gpu);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);
    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    const float norm_lr = trn_opts.learn_rate;
    int64_t total_frames = 0, report_frames = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    RandomAccessPosteriorReader targets_reader(targets_rspecifier);
    RandomAccessBaseFloatVectorReader weights_reader;
    if (frame_weights != "" && !weights_reader.Open(frame_weights)) ASLP_ERR << "cannot open " << frame_weights;
    Xent xent;
    Mse mse;
    CuMatrix feats, feats_transf, nnet_out, obj_diff;
    Timer time;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0;
    for (; !feature_reader.Done(); feature_reader.Next()) {
      std::string utt = feature_reader.Key();
      ASLP_VLOG(3) << "Reading " << utt;
      if (!targets_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing targets"; num_no_tgt_mat++; continue; }
      if (frame_weights != "" && !weights_reader.HasKey(utt)) {
        ASLP_WARN << utt << ", missing per-frame weights";
        num_other_error++;
        feature_reader.Next();  // (sic) the reference advances here AND in the loop header: the next utterance is skipped too (:146)
        if (feature_reader.Done()) break;
        continue;
      }
      HostMatrix mat = feature_reader.Value();
      Posterior targets = targets_reader.Value(utt);
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
      if (drop_len > 0 && mat.rows > drop_len) { ASLP_WARN << utt << ", length too long " << mat.rows << " drop it"; continue; }
      feats = mat;
      nnet_transf.Feedforward(feats, &feats_transf);
      trn_opts.learn_rate = norm_lr / 1024.0;
      nnet.SetTrainOptions(trn_opts);
      if (!crossvalidate) nnet.Propagate(feats_transf, &nnet_out);
      else nnet.Feedforward(feats_transf, &nnet_out);
      if (objective_function == "xent") xent.Eval(weights, nnet_out, targets, &obj_diff);
      else if (objective_function == "mse") mse.Eval(weights, nnet_out, targets, &obj_diff);
      else ASLP_ERR << "Unknown objective function code : " << objective_function;
      if (!crossvalidate) nnet.Backpropagate(obj_diff, NULL);
      if (g_verbose_level >= 1 && total_frames == 0) {
        ASLP_VLOG(1) << "### After " << total_frames << " frames,";
        ASLP_VLOG(1) << nnet.InfoPropagate();
        if (!crossvalidate) { ASLP_VLOG(1) << nnet.InfoBackPropagate(); ASLP_VLOG(1) << nnet.InfoGradient(); }
      }
      num_done++;
      total_frames += feats_transf.NumRows();
      report_frames += feats_transf.NumRows();
      if (report_frames >= report_period && report_period > 0) {
        if (objective_function == "xent") ASLP_LOG << xent.Report();
        if (!crossvalidate) nnet.GetComponentTime();
        report_frames -= report_period;
      }
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
