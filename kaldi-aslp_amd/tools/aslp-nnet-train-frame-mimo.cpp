// aslp-nnet-train-frame-mimo -- src/aslp-nnetbin/aslp-nnet-train-frame-mimo.cc: aslp-nnet-train-frame for graph nets with
// several inputs and / or outputs: one feature table per InputLayer, one target table and one objective ("xent:mse:...")
// per OutputLayer.
#include "cu-device.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"

int main(int argc, char *argv[]) {
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
