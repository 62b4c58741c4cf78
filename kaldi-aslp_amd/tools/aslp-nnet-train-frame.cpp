// aslp-nnet-train-frame -- src/aslp-nnetbin/aslp-nnet-train-frame.cc: one epoch of minibatch SGD (or cross-validation)
// over a feature table with pdf-posterior targets, fed by FrameDataReader.  Same flags, usage text, positional
// arguments and log lines (the bash schedulers grep "AvgLoss:" / "FRAME_ACCURACY").
#include "cu-device.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"

int main(int argc, char *argv[]) {
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
          ones.assign(nnet_in->NumRows(), 1.0f);
          xent->EvalOnLossInput(ones, nnet.LossInput(), nnet.LossInputIsPreSoftmax(), *nnet_tgt, nnet.LossDiff(nnet_in->NumRows()));
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
