// frame_tools.cpp -- the frame-level training tools of src/aslp-nnetbin (train-frame, train-simple, train-mse, train-frame-mimo, train-perutt):
// one entry function per tool (Main_<tool name with _ for ->), linked behind tools/main_stub.cpp into bin/<tool name>.
#include <algorithm>

#include "cu-device.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"
#include "simple_sync.h"

// ======================================================================================================================
// aslp-nnet-train-frame -- src/aslp-nnetbin/aslp-nnet-train-frame.cc: one epoch of minibatch SGD (or cross-validation)
// over a feature table with pdf-posterior targets, fed by FrameDataReader.  Same flags, usage text, positional
// arguments and log lines (the bash schedulers grep "AvgLoss:" / "FRAME_ACCURACY").
int Main_aslp_nnet_train_frame(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of Neural Network training by mini-batch Stochastic Gradient Descent.\n"
        "It is same to aslp-nnet-train-simple, but use FrameDataReader to read feat and label.\n"
        "This version use pdf-posterior as targets, prepared typically by ali-to-post.\n"
        "Usage:  aslp-nnet-train-frame [options] <feature-rspecifier> <targets-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-frame scp:feature.scp ark:posterior.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    NnetDataRandomizerOptions rnd_opts;
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool binary = true, crossvalidate = false, randomize = true;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    po.Register("randomize", &randomize, "Perform the frame-level shuffling within the Cache::");
    std::string objective_function = "xent";
    po.Register("objective-function", &objective_function, "Objective function : xent|mse");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    double dropout_retention = 0.0;
    po.Register("dropout-retention", &dropout_retention, "number between 0..1, saying how many neurons to preserve (0.0 will keep original value");
    int32 report_period = -1;
    po.Register("report-period", &report_period, "Number of frames for one report log, default(-1, no report)");
    int32 gpu_id = -1;
    po.Register("gpu-id", &gpu_id, "selected gpu id, if negative then select automaticly");
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
    if (dropout_retention > 0.0) nnet.SetDropoutRetention(dropout_retention);
    if (crossvalidate) nnet.SetDropoutRetention(1.0);

    LossItf *loss = NULL;
    if (objective_function == "xent") loss = new Xent;
    else if (objective_function == "mse") loss = new Mse;
    else ASLP_ERR << "Unsupported objective function: " << objective_function;
    Xent *xent = dynamic_cast<Xent *>(loss);
    std::vector<BaseFloat> ones;
    CuVector ones_dev;

    Timer time;
    int64_t total_frames = 0, report_frames = 0;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    {
      // the reference shuffles every cache fill whatever --randomize says (it only echoes the flag, :41,136)
      FrameDataReader reader(feature_rspecifier, targets_rspecifier, rnd_opts);
      const CuMatrixBase *nnet_in;
      CuMatrix nnet_out, obj_diff;
      const Posterior *nnet_tgt;
      while (!reader.Done()) {
        Timer tr;
        const bool got = reader.ReadData(&nnet_in, &nnet_tgt);
        CuDevice::Instantiate().AccuProfile("host: FrameDataReader::ReadData", tr.Elapsed());
        if (!got) continue;
        if (!crossvalidate && xent != NULL) {
          // same three steps, in the executor's own buffers: no output / diff copies, final Softmax left to the loss kernel
          Timer t1;
          nnet.PropagateForLoss(*nnet_in, true);
          CuDevice::Instantiate().AccuProfile("host: Propagate (launches)", t1.Elapsed());
          t1.Reset();
          const int32 *labels_dev = nullptr;
          int32 max_label = -1;
          if (nnet.LossInputIsPreSoftmax() && reader.MinibatchLabels(&labels_dev, &max_label) && max_label < nnet.LossInput().NumCols()) {
            // alignment targets: the cache's labels are on the device since the refill, the frame weights are all 1 -- the loss kernel of
            // the branch below on the same numbers, without two uploads and a pass over the Posterior in every step
            if (ones_dev.Dim() != nnet_in->NumRows()) { ones_dev.Resize(nnet_in->NumRows(), kUndefined); ones_dev.Set(1.0); }
            xent->EvalLabelsPreSoftmax(ones_dev, nnet.LossInput(), labels_dev, nnet.LossDiff(nnet_in->NumRows()), 1.0f);
          } else {
            ones.assign(nnet_in->NumRows(), 1.0f);
            xent->EvalOnLossInput(ones, nnet.LossInput(), nnet.LossInputIsPreSoftmax(), *nnet_tgt, nnet.LossDiff(nnet_in->NumRows()));
          }
          CuDevice::Instantiate().AccuProfile("host: Xent::Eval (launches + label upload)", t1.Elapsed());
          t1.Reset();
          nnet.BackpropagateFromLossDiff();
          CuDevice::Instantiate().AccuProfile("host: Backpropagate (launches)", t1.Elapsed());
        } else {
          if (!crossvalidate) nnet.Propagate(*nnet_in, &nnet_out);
          else nnet.Feedforward(*nnet_in, &nnet_out);
          loss->Eval(nnet_out, *nnet_tgt, &obj_diff);
          if (!crossvalidate) nnet.Backpropagate(obj_diff, NULL);
        }
        total_frames += nnet_in->NumRows();
        report_frames += nnet_in->NumRows();
        if (report_period > 0 && report_frames >= report_period) {
          ASLP_LOG << loss->Report();
          report_frames -= report_period;
        }
      }
    }
    {
      Timer tw;
      StreamSync();
      CuDevice::Instantiate().AccuProfile("wait for the GPU after the last minibatch", tw.Elapsed());
      tw.Reset();
      if (!crossvalidate) nnet.Write(target_model_filename, binary);
      CuDevice::Instantiate().AccuProfile("Nnet::Write", tw.Elapsed());
    }
    ASLP_LOG << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << (randomize ? "RANDOMIZED" : "NOT-RANDOMIZED") << ", "
             << time.Elapsed() / 60 << " min, fps" << total_frames / time.Elapsed() << "]";
    ASLP_LOG << loss->Report();
    delete loss;
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-simple -- src/aslp-nnetbin/aslp-nnet-train-simple.cc: one epoch of minibatch SGD (or cross-validation)
// with the tool's own cache-fill loop: optional feature transform, per-frame and per-utterance weights, length
// tolerance, xent | mse | multitask objectives.  Same flags, usage text, positional arguments and log lines.
int Main_aslp_nnet_train_simple(int argc, char *argv[]) { return TrainSimpleWithSync(argc, argv, NULL); }

int TrainSimpleWithSync(int argc, char *argv[], SimpleSync *sync) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of Neural Network training by mini-batch Stochastic Gradient Descent.\n"
        "This version use pdf-posterior as targets, prepared typically by ali-to-post.\n"
        "Usage:  aslp-nnet-train-simple [options] <feature-rspecifier> <targets-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-simple scp:feature.scp ark:posterior.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    NnetDataRandomizerOptions rnd_opts;
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool binary = true, crossvalidate = false, randomize = true;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    po.Register("randomize", &randomize, "Perform the frame-level shuffling within the Cache::");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform in Nnet format");
    std::string objective_function = "xent";
    po.Register("objective-function", &objective_function, "Objective function : xent|mse");
    int32 length_tolerance = 5;
    po.Register("length-tolerance", &length_tolerance, "Allowed length difference of features/targets (frames)");
    std::string frame_weights;
    po.Register("frame-weights", &frame_weights, "Per-frame weights to scale gradients (frame selection/weighting).");
    std::string utt_weights;
    po.Register("utt-weights", &utt_weights, "Per-utterance weights (scalar applied to frame-weights).");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    double dropout_retention = 0.0;
    po.Register("dropout-retention", &dropout_retention, "number between 0..1, saying how many neurons to preserve (0.0 will keep original value");
    int32 report_period = 60000;
    po.Register("report-period", &report_period, "Number of frames for one report log, default(60000)");
    if (sync) sync->Register(&po);
    po.Read(argc, argv);
    if (po.NumArgs() != 4 - (crossvalidate ? 1 : 0)) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);

    if (sync) sync->Connect();  // picks the GPU of its rank and brings up the communicator before the model exists
    else CuDevice::Instantiate().SelectGpuId(use_gpu);

    Nnet nnet_transf;
    if (feature_transform != "") nnet_transf.Read(feature_transform);
    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    if (sync) sync->Init(&nnet, &feature_rspecifier);
    if (dropout_retention > 0.0) { nnet_transf.SetDropoutRetention(dropout_retention); nnet.SetDropoutRetention(dropout_retention); }
    if (crossvalidate) { nnet_transf.SetDropoutRetention(1.0); nnet.SetDropoutRetention(1.0); }

    int64_t total_frames = 0, report_frames = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier);
    RandomAccessPosteriorReader targets_reader(targets_rspecifier);
    RandomAccessBaseFloatVectorReader weights_reader;
    if (frame_weights != "" && !weights_reader.Open(frame_weights)) ASLP_ERR << "cannot open " << frame_weights;
    RandomAccessBaseFloatReader utt_weights_reader;
    if (utt_weights != "" && !utt_weights_reader.Open(utt_weights)) ASLP_ERR << "cannot open " << utt_weights;

    RandomizerMask randomizer_mask(rnd_opts);
    MatrixRandomizer feature_randomizer(rnd_opts);
    PosteriorRandomizer targets_randomizer(rnd_opts);
    VectorRandomizer weights_randomizer(rnd_opts);

    Xent xent;
    Mse mse;
    if (objective_function == "xent") ASLP_LOG << xent.Report();  // "Just for log analysis" (:145)
    MultiTaskLoss multitask;
    if (0 == objective_function.compare(0, 9, "multitask")) multitask.InitFromString(objective_function);

    CuMatrix feats, feats_transf, nnet_out, obj_diff;
    Timer time;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";

    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0;
    while (!feature_reader.Done()) {
      CuDevice::Instantiate().CheckGpuHealth();
      for (; !feature_reader.Done(); feature_reader.Next()) {  // fill the randomizer
        if (feature_randomizer.IsFull()) break;                // suspend, keep utt for next loop
        std::string utt = feature_reader.Key();
        ASLP_VLOG(3) << "Reading " << utt;
        if (!targets_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing targets"; num_no_tgt_mat++; continue; }
        if (frame_weights != "" && !weights_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing per-frame weights"; num_other_error++; continue; }
        if (utt_weights != "" && !utt_weights_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing per-utterance weight"; num_other_error++; continue; }
        HostMatrix mat = feature_reader.Value();
        Posterior targets = targets_reader.Value(utt);
        std::vector<BaseFloat> weights;
        if (frame_weights != "") weights = weights_reader.Value(utt).data;
        else weights.assign(mat.rows, 1.0f);
        if (utt_weights != "") {
          BaseFloat w = utt_weights_reader.Value(utt);
          ASLP_ASSERT(w >= 0.0);
          if (w == 0.0) continue;  // remove sentence from training
          for (BaseFloat &x : weights) x *= w;
        }
        {  // correct small length mismatch ... or drop sentence (:218-240)
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
        feats = mat;
        nnet_transf.Feedforward(feats, &feats_transf);
        ASLP_ASSERT(feats_transf.NumRows() == (int32)targets.size());
        feature_randomizer.AddData(feats_transf);
        targets_randomizer.AddData(targets);
        weights_randomizer.AddData(weights);
        num_done++;
        if (num_done % 5000 == 0) {
          double time_now = time.Elapsed();
          ASLP_VLOG(1) << "After " << num_done << " utterances: time elapsed = " << time_now / 60 << " min; processed "
                       << total_frames / time_now << " frames per second.";
        }
      }
      feature_randomizer.CheckCanProgress();   // (a full cache below one minibatch would be found full again and never read on: nnet-randomizer.h)
      if (!crossvalidate && randomize) {
        const std::vector<int32> &mask = randomizer_mask.Generate(feature_randomizer.NumFrames());
        feature_randomizer.Randomize(mask);
        targets_randomizer.Randomize(mask);
        weights_randomizer.Randomize(mask);
      }
      for (; !feature_randomizer.Done(); feature_randomizer.Next(), targets_randomizer.Next(), weights_randomizer.Next()) {
        const CuMatrixBase &nnet_in = feature_randomizer.Value();
        const Posterior &nnet_tgt = targets_randomizer.Value();
        const std::vector<BaseFloat> &frm_weights = weights_randomizer.Value();
        const bool fused = !crossvalidate && objective_function == "xent";
        if (fused) {  // same three steps in the executor's own buffers (no output / diff copies, Softmax left to the loss kernel)
          nnet.PropagateForLoss(nnet_in, true);
          xent.EvalOnLossInput(frm_weights, nnet.LossInput(), nnet.LossInputIsPreSoftmax(), nnet_tgt, nnet.LossDiff(nnet_in.NumRows()));
          nnet.BackpropagateFromLossDiff();
        } else if (!crossvalidate) nnet.Propagate(nnet_in, &nnet_out);
        else nnet.Feedforward(nnet_in, &nnet_out);
        if (fused) {
        } else if (objective_function == "xent") xent.Eval(frm_weights, nnet_out, nnet_tgt, &obj_diff);
        else if (objective_function == "mse") mse.Eval(frm_weights, nnet_out, nnet_tgt, &obj_diff);
        else if (0 == objective_function.compare(0, 9, "multitask")) multitask.Eval(frm_weights, nnet_out, nnet_tgt, &obj_diff);
        else ASLP_ERR << "Unknown objective function code : " << objective_function;
        if (!crossvalidate && !fused) nnet.Backpropagate(obj_diff, NULL);
        if (g_verbose_level >= 1 && total_frames == 0) {  // 1st minibatch : show what happens in network
          ASLP_VLOG(1) << "### After " << total_frames << " frames,";
          ASLP_VLOG(1) << nnet.InfoPropagate();
          if (!crossvalidate) { ASLP_VLOG(1) << nnet.InfoBackPropagate(); ASLP_VLOG(1) << nnet.InfoGradient(); }
        }
        if (g_verbose_level >= 2 && (total_frames / 25000) != ((total_frames + nnet_in.NumRows()) / 25000)) {
          ASLP_VLOG(2) << "### After " << total_frames << " frames,";
          ASLP_VLOG(2) << nnet.InfoPropagate();
          if (!crossvalidate) ASLP_VLOG(2) << nnet.InfoGradient();
        }
        total_frames += nnet_in.NumRows();
        report_frames += nnet_in.NumRows();
        if (report_frames >= report_period && report_period > 0) {
          if (objective_function == "xent") ASLP_LOG << xent.Report();
          report_frames -= report_period;
        }
        if (sync) sync->AfterMinibatch();
      }
    }
    if (sync) sync->Finish();
    if (g_verbose_level >= 1) {
      ASLP_VLOG(1) << "### After " << total_frames << " frames,";
      ASLP_VLOG(1) << nnet.InfoPropagate();
      if (!crossvalidate) { ASLP_VLOG(1) << nnet.InfoBackPropagate(); ASLP_VLOG(1) << nnet.InfoGradient(); }
    }
    if (!crossvalidate && (!sync || sync->WritesModel())) nnet.Write(target_model_filename, binary);
    StreamSync();
    ASLP_LOG << "Done " << num_done << " files, " << num_no_tgt_mat << " with no tgt_mats, " << num_other_error << " with other errors. "
             << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << (randomize ? "RANDOMIZED" : "NOT-RANDOMIZED") << ", "
             << time.Elapsed() / 60 << " min, fps" << total_frames / time.Elapsed() << "]";
    if (objective_function == "xent") ASLP_LOG << xent.Report();
    else if (objective_function == "mse") ASLP_LOG << mse.Report();
    else if (0 == objective_function.compare(0, 9, "multitask")) ASLP_LOG << multitask.Report();
    else ASLP_ERR << "Unknown objective function code : " << objective_function;
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-mse -- src/aslp-nnetbin/aslp-nnet-train-mse.cc: the train-simple loop for regression nets.  The targets
// are a second MATRIX table read in step with the features (same keys, same order -- anything else is an error), shuffled
// by a MatrixRandomizer under the same mask, scored by Mse.  The usage text still says train-simple, as in the reference.
int Main_aslp_nnet_train_mse(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of Neural Network training by mini-batch Stochastic Gradient Descent.\n"
        "This version use pdf-posterior as targets, prepared typically by ali-to-post.\n"
        "Usage:  aslp-nnet-train-simple [options] <feature-rspecifier> <targets-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-simple scp:feature.scp ark:posterior.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    NnetDataRandomizerOptions rnd_opts;
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool binary = true, crossvalidate = false, randomize = true;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    po.Register("randomize", &randomize, "Perform the frame-level shuffling within the Cache::");
    std::string feature_transform;
    po.Register("feature-transform", &feature_transform, "Feature transform in Nnet format");
    std::string objective_function = "mse";
    po.Register("objective-function", &objective_function, "Objective function : xent|mse");
    int32 length_tolerance = 5;
    po.Register("length-tolerance", &length_tolerance, "Allowed length difference of features/targets (frames)");
    std::string frame_weights;
    po.Register("frame-weights", &frame_weights, "Per-frame weights to scale gradients (frame selection/weighting).");
    std::string utt_weights;
    po.Register("utt-weights", &utt_weights, "Per-utterance weights (scalar applied to frame-weights).");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    double dropout_retention = 0.0;
    po.Register("dropout-retention", &dropout_retention, "number between 0..1, saying how many neurons to preserve (0.0 will keep original value");
    int32 report_period = 60000;
    po.Register("report-period", &report_period, "Number of frames for one report log, default(60000)");
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
    if (dropout_retention > 0.0) { nnet_transf.SetDropoutRetention(dropout_retention); nnet.SetDropoutRetention(dropout_retention); }
    if (crossvalidate) { nnet_transf.SetDropoutRetention(1.0); nnet.SetDropoutRetention(1.0); }

    int64_t total_frames = 0, report_frames = 0;
    SequentialBaseFloatMatrixReader feature_reader(feature_rspecifier), targets_reader(targets_rspecifier);
    RandomAccessBaseFloatVectorReader weights_reader;
    if (frame_weights != "" && !weights_reader.Open(frame_weights)) ASLP_ERR << "cannot open " << frame_weights;
    RandomAccessBaseFloatReader utt_weights_reader;
    if (utt_weights != "" && !utt_weights_reader.Open(utt_weights)) ASLP_ERR << "cannot open " << utt_weights;
    RandomizerMask randomizer_mask(rnd_opts);
    MatrixRandomizer feature_randomizer(rnd_opts), targets_randomizer(rnd_opts);
    VectorRandomizer weights_randomizer(rnd_opts);
    Mse mse;
    if (objective_function != "mse") ASLP_ERR << "Only support mse training";

    CuMatrix feats, feats_transf, tgt_dev, nnet_out, obj_diff;
    Timer time;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    int32 num_done = 0, num_no_tgt_mat = 0, num_other_error = 0;
    while (!feature_reader.Done()) {
      CuDevice::Instantiate().CheckGpuHealth();
      // both tables advance together, also past an utterance that is left out (:163)
      for (; !feature_reader.Done(); feature_reader.Next(), targets_reader.Next()) {
        if (feature_randomizer.IsFull()) break;
        const std::string utt = feature_reader.Key();
        if (targets_reader.Done() || utt != targets_reader.Key())
          ASLP_ERR << "feat and target not in the same order or not exist in target"
                   << "feat key " << utt << "target key " << (targets_reader.Done() ? std::string() : targets_reader.Key());
        ASLP_VLOG(3) << "Reading " << utt;
        if (frame_weights != "" && !weights_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing per-frame weights"; num_other_error++; continue; }
        if (utt_weights != "" && !utt_weights_reader.HasKey(utt)) { ASLP_WARN << utt << ", missing per-utterance weight"; num_other_error++; continue; }
        const HostMatrix &mat = feature_reader.Value();
        const HostMatrix &targets = targets_reader.Value();
        if (mat.rows != targets.rows) {
          ASLP_WARN << utt << " feat and target are not the same dim" << " feat " << mat.rows << " target " << targets.rows;
          continue;
        }
        std::vector<BaseFloat> weights;
        if (frame_weights != "") weights = weights_reader.Value(utt).data;
        else weights.assign(mat.rows, 1.0f);
        feats = mat;
        nnet_transf.Feedforward(feats, &feats_transf);
        ASLP_ASSERT(feats_transf.NumRows() == targets.rows);
        tgt_dev = targets;
        feature_randomizer.AddData(feats_transf);
        targets_randomizer.AddData(tgt_dev);
        weights_randomizer.AddData(weights);
        num_done++;
        if (num_done % 5000 == 0) {
          double time_now = time.Elapsed();
          ASLP_VLOG(1) << "After " << num_done << " utterances: time elapsed = " << time_now / 60 << " min; processed "
                       << total_frames / time_now << " frames per second.";
        }
      }
      feature_randomizer.CheckCanProgress();   // (a full cache below one minibatch would be found full again and never read on: nnet-randomizer.h)
      if (!crossvalidate && randomize) {
        const std::vector<int32> &mask = randomizer_mask.Generate(feature_randomizer.NumFrames());
        feature_randomizer.Randomize(mask);
        targets_randomizer.Randomize(mask);
        weights_randomizer.Randomize(mask);
      }
      for (; !feature_randomizer.Done(); feature_randomizer.Next(), targets_randomizer.Next(), weights_randomizer.Next()) {
        const CuMatrixBase &nnet_in = feature_randomizer.Value();
        const CuMatrixBase &nnet_tgt = targets_randomizer.Value();
        const std::vector<BaseFloat> &frm_weights = weights_randomizer.Value();
        if (!crossvalidate) nnet.Propagate(nnet_in, &nnet_out);
        else nnet.Feedforward(nnet_in, &nnet_out);
        mse.Eval(frm_weights, nnet_out, nnet_tgt, &obj_diff);
        if (!crossvalidate) nnet.Backpropagate(obj_diff, NULL);
        if (g_verbose_level >= 1 && total_frames == 0) {
          ASLP_VLOG(1) << "### After " << total_frames << " frames,";
          ASLP_VLOG(1) << nnet.InfoPropagate();
          if (!crossvalidate) { ASLP_VLOG(1) << nnet.InfoBackPropagate(); ASLP_VLOG(1) << nnet.InfoGradient(); }
        }
        if (g_verbose_level >= 2 && (total_frames / 25000) != ((total_frames + nnet_in.NumRows()) / 25000)) {
          ASLP_VLOG(2) << "### After " << total_frames << " frames,";
          ASLP_VLOG(2) << nnet.InfoPropagate();
          if (!crossvalidate) ASLP_VLOG(2) << nnet.InfoGradient();
        }
        total_frames += nnet_in.NumRows();
        report_frames += nnet_in.NumRows();
        if (report_frames >= report_period) { ASLP_LOG << mse.Report(); report_frames -= report_period; }
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
    ASLP_LOG << mse.Report();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-frame-mimo -- src/aslp-nnetbin/aslp-nnet-train-frame-mimo.cc: aslp-nnet-train-frame for graph nets with
// several inputs and / or outputs: one feature table per InputLayer, one target table and one objective ("xent:mse:...")
// per OutputLayer.
int Main_aslp_nnet_train_frame_mimo(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Perform one iteration of Neural Network training by mini-batch Stochastic Gradient Descent.\n"
        "It is same to aslp-nnet-train-frame, but the network has multi input or multi output.\n"
        "Attention: num input feat and target must match the input num and the output num of the nnet\n"
        "This version use pdf-posterior as targets, prepared typically by ali-to-post.\n"
        "Usage:  aslp-nnet-train-frame-mimo [options] <feature-rspecifier_1>...<feature_rspecifier_n> "
        "                   <targets-rspecifier_1>...<targets_rspecifier_n> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-frame-mimo scp:feature1.scp scp:feature2.scp "
        "                       ark:posterior1.ark ark:posterior2.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    NnetDataRandomizerOptions rnd_opts;
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool binary = true, crossvalidate = false, randomize = true;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("cross-validate", &crossvalidate, "Perform cross-validation (don't backpropagate)");
    po.Register("randomize", &randomize, "Perform the frame-level shuffling within the Cache::");
    std::string objective_function = "xent";
    po.Register("objective-function", &objective_function, "Objective function : xent|mse");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    double dropout_retention = 0.0;
    po.Register("dropout-retention", &dropout_retention, "number between 0..1, saying how many neurons to preserve (0.0 will keep original value");
    int32 report_period = -1;
    po.Register("report-period", &report_period, "Number of frames for one report log, default(-1, no report)");
    po.Read(argc, argv);
    if (po.NumArgs() == 0) { po.PrintUsage(); exit(1); }
    const int num_args = po.NumArgs();
    std::string model_filename, target_model_filename;
    if (!crossvalidate) {
      if (num_args < 2) { po.PrintUsage(); exit(1); }
      model_filename = po.GetArg(num_args - 1);
      target_model_filename = po.GetArg(num_args);
    } else {
      model_filename = po.GetArg(num_args);
    }
    CuDevice::Instantiate().SelectGpuId(use_gpu);
    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    const int num_input = nnet.NumInput(), num_output = nnet.NumOutput();
    ASLP_LOG << "Nnet num_input " << num_input << " num_output " << num_output;
    const int extra = !crossvalidate ? 2 : 1;
    if (num_args != num_input + num_output + extra) { po.PrintUsage(); exit(1); }
    std::vector<std::string> features, targets;
    for (int i = 0; i < num_input; i++) features.push_back(po.GetArg(i + 1));
    for (int i = 0; i < num_output; i++) targets.push_back(po.GetArg(i + num_input + 1));
    if (dropout_retention > 0.0) nnet.SetDropoutRetention(dropout_retention);
    if (crossvalidate) nnet.SetDropoutRetention(1.0);

    std::vector<std::unique_ptr<LossItf>> losses(num_output);
    std::vector<std::string> sub_string;
    SplitStringToVector(objective_function, ":", true, &sub_string);
    if ((int)sub_string.size() != num_output)
      ASLP_ERR << objective_function << "obj dim not match the nnet output layers num, need " << num_output << " obj function";
    for (int i = 0; i < num_output; i++) {
      if (sub_string[i] == "xent") losses[i].reset(new Xent);
      else if (sub_string[i] == "mse") losses[i].reset(new Mse);
      else ASLP_ERR << "Unsupported objective function: " << sub_string[i];
    }
    Timer time;
    int64_t total_frames = 0, report_frames = 0;
    ASLP_LOG << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << " STARTED";
    {
      MimoFrameDataReader reader(features, targets, rnd_opts);
      std::vector<const CuMatrixBase *> nnet_in;
      std::vector<const Posterior *> nnet_tgt;
      std::vector<std::unique_ptr<CuMatrix>> out_store, diff_store;
      std::vector<CuMatrix *> nnet_out, obj_diff;
      std::vector<const CuMatrixBase *> const_obj_diff;
      for (int i = 0; i < num_output; i++) {
        out_store.emplace_back(new CuMatrix);
        diff_store.emplace_back(new CuMatrix);
        nnet_out.push_back(out_store.back().get());
        obj_diff.push_back(diff_store.back().get());
        const_obj_diff.push_back(diff_store.back().get());
      }
      while (!reader.Done()) {
        if (!reader.ReadData(&nnet_in, &nnet_tgt)) continue;
        if (!crossvalidate) nnet.Propagate(nnet_in, &nnet_out);
        else nnet.Feedforward(nnet_in, &nnet_out);
        for (int i = 0; i < num_output; i++) losses[i]->Eval(*nnet_out[i], *nnet_tgt[i], obj_diff[i]);
        if (!crossvalidate) nnet.Backpropagate(const_obj_diff, NULL);
        total_frames += nnet_in[0]->NumRows();
        report_frames += nnet_in[0]->NumRows();
        if (report_period > 0 && report_frames >= report_period) {
          for (int i = 0; i < num_output; i++) { ASLP_LOG << "Obj " << "[" << i << "] " << sub_string[i]; ASLP_LOG << losses[i]->Report(); }
          report_frames -= report_period;
        }
      }
    }
    if (!crossvalidate) nnet.Write(target_model_filename, binary);
    StreamSync();
    ASLP_LOG << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << (randomize ? "RANDOMIZED" : "NOT-RANDOMIZED") << ", "
             << time.Elapsed() / 60 << " min, fps" << total_frames / time.Elapsed() << "]";
    for (int i = 0; i < num_output; i++) { ASLP_LOG << "Obj " << "[" << i << "] " << sub_string[i]; ASLP_LOG << losses[i]->Report(); }
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-perutt -- src/aslp-nnetbin/aslp-nnet-train-perutt.cc: one update per utterance (the FSMN recipes,
// run_cfsmn.sh), learning rate divided by 1024 (:203), optional feature transform / frame weights / length tolerance.
int Main_aslp_nnet_train_perutt(int argc, char *argv[]) {
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
    RandomizerMask randomizer_mask(rnd_opts);   // unused by this tool, as in the reference (aslp-nnet-train-perutt.cc:127), but its construction seeds the generator and says so in the log
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
