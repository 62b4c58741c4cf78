// aslp-nnet-train-lstm-stream-worker -- src/aslp-parallelbin/aslp-nnet-train-lstm-stream-worker.cc: the data-parallel twin
// of aslp-nnet-train-lstm-streams (one process per GPU on its own shard, IWorker::Synchronize every --sync-period valid
// frames; rank / rendezvous as in aslp-nnet-train-frame-worker).  Multi-stream truncated-BPTT training
// of (projected / CIFG / GRU) LSTM nets fed by SequenceDataReader: batch-size frames of num-stream utterances per step,
// targets delayed by --targets-delay frames, history reset per stream when it takes a new utterance.
#include "cu-device.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"
#include "workers.h"

int main(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Parallel worker of aslp-nnet-train-lstm-stream, but don't do cross validation"
        "see aslp-nnet-train-lstm-subsequence-stream for details\n"
        "Usage: aslp-nnet-train-lstm-stream-worker [options] "
        "<feature-rspecifier> <targets-respecifier> <model-in> <model-out>\n"
        "e.g.: \n"
        "aslp-nnet-train-lstm-stream-worker scp:feature.scp ark:posterior.ark nnet.init nnet.iter1\n";
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
    int32 rank = -1, num_workers = -1;
    po.Register("rank", &rank, "Rank of this worker (default: from the launcher's environment)");
    po.Register("num-workers", &num_workers, "Number of workers (default: from the launcher's environment)");
    std::string comm_file = "";
    po.Register("comm-file", &comm_file, "Rendezvous file for the RCCL communicator (required with more than one worker)");
    po.Read(argc, argv);
    if (crossvalidate) ASLP_ERR << "the worker tools train only (use aslp-nnet-train-lstm-streams --cross-validate=true)";
    if (po.NumArgs() != 4) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);
    RankFromEnvironment(&rank, &num_workers);
    if (gpu_id >= 0) CuDevice::Instantiate().SetGpuId(gpu_id);
    else if (num_workers > 1) CuDevice::Instantiate().SetGpuId(rank);
    else CuDevice::Instantiate().SelectGpuId(use_gpu);
    // the communicator before the model (see aslp-nnet-train-frame-worker)
    std::unique_ptr<Comm> comm(NewRcclComm(rank, num_workers, comm_file));

    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    int64_t total_frames = 0;
    int32 num_done = 0, num_sentence = 0;
    LossItf *loss = NULL;
    if (objective_function == "xent") loss = new Xent;
    else if (objective_function == "mse") loss = new Mse;
    else ASLP_ERR << "Unsupported objective function: " << objective_function;
    std::unique_ptr<IWorker> worker;
    if (worker_type == "bsp") worker.reset(new BspWorker(comm.get()));
    else if (worker_type == "bmuf") worker.reset(new BmufWorker(comm.get(), bmuf_learn_rate, bmuf_momentum));
    else ASLP_ERR << "Unsupported worker type: " << worker_type << " (this build has the collective workers bsp | bmuf)";
    std::vector<std::pair<BaseFloat *, int>> params;
    nnet.GetGpuParams(&params);
    worker->InitParam(params);
    ASLP_LOG << "Mpi cluster info total " << worker->NumNodes() << " worker rank " << worker->Rank();
    int32 num_frames_since_last_sync = 0;
    Timer time;
    ASLP_LOG << "TRAINING STARTED";
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
      num_frames_since_last_sync += frame_progress;
      if (num_frames_since_last_sync > sync_period) {
        ASLP_VLOG(2) << "Worker " << worker->Rank() << " synchronize once";
        worker->Synchronize(num_frames_since_last_sync);
        num_frames_since_last_sync = 0;
      }
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
    worker->Stop();
    {
      std::vector<double *> acc_params;
      std::vector<std::pair<double *, int>> data_params;
      nnet.GetAccStats(&acc_params, &data_params);
      worker->ReduceAccStat(acc_params, data_params);
    }
    StreamSync();
    if (worker->IsMainNode()) nnet.Write(target_model_filename, binary);
    ASLP_LOG << "Done " << num_done << " files, " << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", "
             << (randomize ? "RANDOMIZED" : "NOT-RANDOMIZED") << ", " << time.Elapsed() / 60 << " min, fps" << total_frames / time.Elapsed() << "]";
    ASLP_LOG << loss->Report();  // the reference calls Report() and drops the string (:222); the schedulers need the line
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
