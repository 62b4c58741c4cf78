// aslp-nnet-train-frame-worker -- src/aslp-parallelbin/aslp-nnet-train-frame-worker.cc: the data-parallel twin of
// aslp-nnet-train-frame.  One process per GPU, each on its own shard of the utterance list; every --sync-period frames
// the worker calls IWorker::Synchronize.  The reference is started by mpirun and syncs through MPI on host copies; this
// one syncs through RCCL on the device buffers and takes its rank from the launcher's environment (OMPI_COMM_WORLD_*,
// PMI_*, RANK / WORLD_SIZE) or from --rank / --num-workers, with --comm-file as the rendezvous point.
#include "cu-device.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"
#include "workers.h"

int main(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Parallel worker of aslp-nnet-train-frame, but don't do cross validation"
        "see aslp-nnet-train-frame for details\n"
        "Usage:  aslp-nnet-train-frame-worker [options] "
        "<feature-rspecifier> <targets-rspecifier> <model-in> [<model-out>]\n"
        "e.g.: \n"
        " aslp-nnet-train-frame-worker scp:feature.scp ark:posterior.ark nnet.init nnet.iter1\n";
    ParseOptions po(usage);
    NnetTrainOptions trn_opts;
    RegisterTrainOptions(&trn_opts, &po);
    NnetDataRandomizerOptions rnd_opts;
    RegisterRandomizerOptions(&rnd_opts, &po);
    bool binary = true, randomize = true;
    po.Register("binary", &binary, "Write output in binary mode");
    po.Register("randomize", &randomize, "Perform the frame-level shuffling within the Cache::");
    std::string objective_function = "xent";
    po.Register("objective-function", &objective_function, "Objective function : xent|mse");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    double dropout_retention = 0.0;
    po.Register("dropout-retention", &dropout_retention, "number between 0..1, saying how many neurons to preserve (0.0 will keep original value");
    int32 report_period = -1;
    po.Register("report-period", &report_period, "Number of frames for one report log, default(-1, no report)");
    std::string worker_type = "bsp";
    po.Register("worker-type", &worker_type, "Worker type(bsp | bmuf | easgd | asgd | masgd | sod)");
    float alpha = 0.5;
    po.Register("alpha", &alpha, "Moving rate alpha for easgd worker");
    float bmuf_momentum = 0.9;
    po.Register("bmuf-momentum", &bmuf_momentum, "momentum for bmuf worker");
    float bmuf_learn_rate = 1.0;
    po.Register("bmuf-learn-rate", &bmuf_learn_rate, "learn rate for bmuf worker");
    int32 sync_period = 25600;
    po.Register("sync-period", &sync_period, "number frames for every synchronization");
    int32 gpu_id = -1;
    po.Register("gpu-id", &gpu_id, "selected gpu id, if negative then select automaticly");
    int32 rank = -1, num_workers = -1;
    po.Register("rank", &rank, "Rank of this worker (default: from the launcher's environment)");
    po.Register("num-workers", &num_workers, "Number of workers (default: from the launcher's environment)");
    std::string comm_file = "";
    po.Register("comm-file", &comm_file, "Rendezvous file for the RCCL communicator (required with more than one worker)");
    po.Read(argc, argv);
    if (po.NumArgs() != 4) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3),
                target_model_filename = po.GetArg(4);

    RankFromEnvironment(&rank, &num_workers);
    if (gpu_id >= 0) CuDevice::Instantiate().SetGpuId(gpu_id);
    else if (num_workers > 1) CuDevice::Instantiate().SetGpuId(rank);  // one process per GPU of the node
    else CuDevice::Instantiate().SelectGpuId(use_gpu);
    // the communicator before the model: RCCL initialised after the first allocations / launches leaves every later step
    // slower on this stack (measured: 1.44 vs 2.39 ms/step on the cfg2 DNN)
    std::unique_ptr<Comm> comm(NewRcclComm(rank, num_workers, comm_file));

    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    if (dropout_retention > 0.0) nnet.SetDropoutRetention(dropout_retention);
    LossItf *loss = NULL;
    if (objective_function == "xent") loss = new Xent;
    else if (objective_function == "mse") loss = new Mse;
    else ASLP_ERR << "Unsupported objective function: " << objective_function;
    Xent *xent = dynamic_cast<Xent *>(loss);

    std::unique_ptr<IWorker> worker;
    if (worker_type == "bsp") worker.reset(new BspWorker(comm.get()));
    else if (worker_type == "bmuf") worker.reset(new BmufWorker(comm.get(), bmuf_learn_rate, bmuf_momentum));
    else ASLP_ERR << "Unsupported worker type: " << worker_type << " (this build has the collective workers bsp | bmuf)";
    std::vector<std::pair<BaseFloat *, int>> params;
    nnet.GetGpuParams(&params);
    worker->InitParam(params);
    ASLP_LOG << "Mpi cluster info total " << worker->NumNodes() << " worker rank " << worker->Rank();

    Timer time;
    int64_t total_frames = 0, report_frames = 0;
    int32 num_frames_since_last_sync = 0;
    ASLP_LOG << "TRAINING STARTED";
    {
      FrameDataReader reader(feature_rspecifier, targets_rspecifier, rnd_opts);
      const CuMatrixBase *nnet_in;
      CuMatrix nnet_out, obj_diff;
      const Posterior *nnet_tgt;
      std::vector<BaseFloat> ones;
      while (!reader.Done()) {
        if (!reader.ReadData(&nnet_in, &nnet_tgt)) continue;
        if (xent != NULL) {
          nnet.PropagateForLoss(*nnet_in, true);
          ones.assign(nnet_in->NumRows(), 1.0f);
          xent->EvalOnLossInput(ones, nnet.LossInput(), nnet.LossInputIsPreSoftmax(), *nnet_tgt, nnet.LossDiff(nnet_in->NumRows()));
          nnet.BackpropagateFromLossDiff();
        } else {
          nnet.Propagate(*nnet_in, &nnet_out);
          loss->Eval(nnet_out, *nnet_tgt, &obj_diff);
          nnet.Backpropagate(obj_diff, NULL);
        }
        total_frames += nnet_in->NumRows();
        report_frames += nnet_in->NumRows();
        num_frames_since_last_sync += nnet_in->NumRows();
        if (num_frames_since_last_sync > sync_period) {
          ASLP_VLOG(2) << "Worker " << worker->Rank() << " synchronize once";
          worker->Synchronize(num_frames_since_last_sync);
          num_frames_since_last_sync = 0;
        }
        if (report_period > 0 && report_frames >= report_period) { ASLP_LOG << loss->Report(); report_frames -= report_period; }
      }
    }
    worker->Stop();
    std::vector<double *> acc_params;
    std::vector<std::pair<double *, int>> data_params;
    nnet.GetAccStats(&acc_params, &data_params);
    worker->ReduceAccStat(acc_params, data_params);
    StreamSync();
    if (worker->IsMainNode()) nnet.Write(target_model_filename, binary);
    ASLP_LOG << loss->Report();
    ASLP_LOG << "[" << "TRAINING" << ", " << (randomize ? "RANDOMIZED" : "NOT-RANDOMIZED") << ", " << time.Elapsed() / 60 << " min, fps"
             << total_frames / time.Elapsed() << "]";
    delete loss;
    worker.reset();
    comm.reset();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}
