// worker_tools.cpp -- the data-parallel worker tools of src/aslp-parallelbin (train-frame-worker, train-lstm-stream-worker,
// train-lc-blstm-streams-worker) on the native sync workers (parallel/workers.h): one entry function per tool, linked behind
// tools/main_stub.cpp into bin/<tool name>.
#include <hip/hip_runtime.h>
#include "cu-device.h"
#include "data-reader.h"
#include "nnet-loss.h"
#include "nnet-nnet.h"
#include "workers.h"
#include "../simple_sync.h"

namespace {
// the --worker-type switch of the three worker tools (aslp-nnet-train-frame-worker.cc:114-126).  With easgd / asgd / masgd rank 0
// of the group is aslp-nnet-train-server and these tools are ranks 1 .. N-1.
std::unique_ptr<aslp::IWorker> MakeWorker(const std::string &type, aslp::Comm *comm, float alpha, float bmuf_learn_rate, float bmuf_momentum,
                                          const aslp::OptimizerOption *optimizer_opts) {
  using namespace aslp;
  std::unique_ptr<IWorker> worker;
  if (type == "bsp") worker.reset(new BspWorker(comm));
  else if (type == "bmuf") worker.reset(new BmufWorker(comm, bmuf_learn_rate, bmuf_momentum));
  else if (type == "sod" && optimizer_opts) worker.reset(new SodWorker(comm, *optimizer_opts));
  else if (type == "easgd" || type == "asgd" || type == "masgd") {
    if (comm->NumNodes() < 2 || comm->Rank() == 0)
      ASLP_ERR << "worker type " << type << " needs aslp-nnet-train-server as rank 0 and the workers as ranks 1 .. N-1 (this is rank " << comm->Rank()
               << " of " << comm->NumNodes() << ")";
    if (type == "easgd") worker.reset(new EasgdWorker(comm, alpha));
    else worker.reset(new AsgdWorker(comm));
  } else {
    ASLP_ERR << "Unsupported worker type: " << type;
  }
  return worker;
}
}  // namespace

// ======================================================================================================================
// aslp-nnet-train-frame-worker -- src/aslp-parallelbin/aslp-nnet-train-frame-worker.cc: the data-parallel twin of
// aslp-nnet-train-frame.  One process per GPU, each on its own shard of the utterance list; every --sync-period frames
// the worker calls IWorker::Synchronize.  The reference is started by mpirun and syncs through MPI on host copies; this
// one syncs through RCCL on the device buffers and takes its rank from the launcher's environment (OMPI_COMM_WORLD_*,
// PMI_*, RANK / WORLD_SIZE) or from --rank / --num-workers, with --comm-file as the rendezvous point.
// The device of rank `rank` of `num_workers`: --gpu-id if given; one GPU per rank for the RCCL transport; over shared memory (where ranks
// may share GPUs) rank modulo the number of devices this box has; a single worker lets --use-gpu decide.
static void SelectWorkerDevice(int gpu_id, int rank, int num_workers, const std::string &comm_transport, const std::string &use_gpu) {
  using namespace aslp;
  if (gpu_id >= 0) { CuDevice::Instantiate().SetGpuId(gpu_id); return; }
  if (num_workers <= 1) { CuDevice::Instantiate().SelectGpuId(use_gpu); return; }
  std::string t = comm_transport;
  if (t.empty() && getenv("ASLP_COMM_TRANSPORT") != nullptr) t = getenv("ASLP_COMM_TRANSPORT");
  int ndev = 1;
  if (t == "shm" && (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)) ndev = 1;
  CuDevice::Instantiate().SetGpuId(t == "shm" ? rank % ndev : rank);
}

int Main_aslp_nnet_train_frame_worker(int argc, char *argv[]) {
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
    OptimizerOption optimizer_opts;
    optimizer_opts.Register(&po);
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
    std::string comm_transport = "";
    po.Register("comm-transport", &comm_transport, "rccl (one GPU per worker, default) | shm (workers may share a GPU: tensors staged through shared memory); default from ASLP_COMM_TRANSPORT");
    po.Read(argc, argv);
    if (po.NumArgs() != 4) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3),
                target_model_filename = po.GetArg(4);

    RankFromEnvironment(&rank, &num_workers);
    SelectWorkerDevice(gpu_id, rank, num_workers, comm_transport, use_gpu);
    // the communicator before the model: RCCL initialised after the first allocations / launches leaves every later step
    // slower on this stack (measured: 1.44 vs 2.39 ms/step on the cfg2 DNN)
    std::unique_ptr<Comm> comm(NewProcessComm(comm_transport, rank, num_workers, comm_file));

    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    if (dropout_retention > 0.0) nnet.SetDropoutRetention(dropout_retention);
    LossItf *loss = NULL;
    if (objective_function == "xent") loss = new Xent;
    else if (objective_function == "mse") loss = new Mse;
    else ASLP_ERR << "Unsupported objective function: " << objective_function;
    Xent *xent = dynamic_cast<Xent *>(loss);

    std::unique_ptr<IWorker> worker = MakeWorker(worker_type, comm.get(), alpha, bmuf_learn_rate, bmuf_momentum, &optimizer_opts);
    std::vector<std::pair<BaseFloat *, int>> params;
    nnet.GetGpuParams(&params);
    nnet.ParamWritersAnnounce();   // the sync workers call aslp_params_changed() after every exchange
    worker->InitParam(params);
    ASLP_LOG << "Mpi cluster info total " << worker->NumNodes() << " worker rank " << worker->Rank();

    Timer time;
    int64_t total_frames = 0, report_frames = 0;
    int32 num_frames_since_last_sync = 0;
    ASLP_LOG << "TRAINING STARTED";
    {
      FrameDataReader reader(feature_rspecifier, targets_rspecifier, rnd_opts);
      const CuMatrixBase *nnet_in = nullptr;
      CuMatrix nnet_out, obj_diff;
      const Posterior *nnet_tgt = nullptr;
      std::vector<BaseFloat> ones;
      while (!reader.Done()) {
        // The reference's worker does not look at what ReadData returns (aslp-nnet-train-frame-worker.cc:147; aslp-nnet-train-frame.cc:110-111
        // does): when the last cache fill of the data holds less than one minibatch, the loop body runs once more on the minibatch of the step before --
        // one more update, counted into the frames and the sync schedule.  Same here (the reader keeps that minibatch for the purpose).
        reader.ReadData(&nnet_in, &nnet_tgt);
        if (nnet_in == nullptr) continue;   // (no full minibatch in the whole input)
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

// ======================================================================================================================
// aslp-nnet-train-lstm-stream-worker -- src/aslp-parallelbin/aslp-nnet-train-lstm-stream-worker.cc: the data-parallel twin
// of aslp-nnet-train-lstm-streams (one process per GPU on its own shard, IWorker::Synchronize every --sync-period valid
// frames; rank / rendezvous as in aslp-nnet-train-frame-worker).  Multi-stream truncated-BPTT training
// of (projected / CIFG / GRU) LSTM nets fed by SequenceDataReader: batch-size frames of num-stream utterances per step,
// targets delayed by --targets-delay frames, history reset per stream when it takes a new utterance.
int Main_aslp_nnet_train_lstm_stream_worker(int argc, char *argv[]) {
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
    OptimizerOption optimizer_opts;
    optimizer_opts.Register(&po);
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
    std::string comm_transport = "";
    po.Register("comm-transport", &comm_transport, "rccl (one GPU per worker, default) | shm (workers may share a GPU: tensors staged through shared memory); default from ASLP_COMM_TRANSPORT");
    po.Read(argc, argv);
    if (crossvalidate) ASLP_ERR << "the worker tools train only (use aslp-nnet-train-lstm-streams --cross-validate=true)";
    if (po.NumArgs() != 4) { po.PrintUsage(); exit(1); }
    std::string feature_rspecifier = po.GetArg(1), targets_rspecifier = po.GetArg(2), model_filename = po.GetArg(3);
    std::string target_model_filename;
    if (!crossvalidate) target_model_filename = po.GetArg(4);
    RankFromEnvironment(&rank, &num_workers);
    SelectWorkerDevice(gpu_id, rank, num_workers, comm_transport, use_gpu);
    // the communicator before the model (see aslp-nnet-train-frame-worker)
    std::unique_ptr<Comm> comm(NewProcessComm(comm_transport, rank, num_workers, comm_file));

    Nnet nnet;
    nnet.Read(model_filename);
    nnet.SetTrainOptions(trn_opts);
    int64_t total_frames = 0;
    int32 num_done = 0, num_sentence = 0;
    LossItf *loss = NULL;
    if (objective_function == "xent") loss = new Xent;
    else if (objective_function == "mse") loss = new Mse;
    else ASLP_ERR << "Unsupported objective function: " << objective_function;
    RandomizerMask randomizer_mask(rnd_opts);   // unused by this tool, as in the reference (aslp-nnet-train-lstm-stream-worker.cc:111), but its construction seeds the generator and says so in the log
    std::unique_ptr<IWorker> worker = MakeWorker(worker_type, comm.get(), alpha, bmuf_learn_rate, bmuf_momentum, &optimizer_opts);
    std::vector<std::pair<BaseFloat *, int>> params;
    nnet.GetGpuParams(&params);
    nnet.ParamWritersAnnounce();   // the sync workers call aslp_params_changed() after every exchange
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
        ASLP_LOG << "Worker " << worker->Rank() << " synchronize once";
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

// ======================================================================================================================
// aslp-nnet-train-lc-blstm-streams-worker -- src/aslp-parallelbin/aslp-nnet-train-lc-blstm-streams-worker.cc: the
// data-parallel twin of aslp-nnet-train-blstm-streams-lc (one process per GPU on its own shard; IWorker::Synchronize every
// --sync-period valid frames; rank / rendezvous as in aslp-nnet-train-frame-worker).  Latency-controlled BLSTM
// training.  num-stream utterances advance in parallel; every step is a [ (chunk + right) * S x D ] batch, rows t*S + s,
// with a frame mask that is 1 on the chunk frames of live streams; each stream then rewinds by right-splice frames, an
// exhausted stream takes the next utterance and has its history reset (Nnet::ResetLstmStreams).
int Main_aslp_nnet_train_lc_blstm_streams_worker(int argc, char *argv[]) {
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
    po.Register("right_splice", &right_splice, "---BLSTM--- Latency-controlled BPTT right context size");
    int32 gpu_id = -1;
    po.Register("gpu-id", &gpu_id, "selected gpu id, if negative then select automaticly");
    std::string worker_type = "bsp";
    po.Register("worker-type", &worker_type, "Worker type(bsp | bmuf | easgd)");
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
    std::string comm_transport = "";
    po.Register("comm-transport", &comm_transport, "rccl (one GPU per worker, default) | shm (workers may share a GPU: tensors staged through shared memory); default from ASLP_COMM_TRANSPORT");
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

    if (crossvalidate) ASLP_ERR << "the worker tools train only (use aslp-nnet-train-blstm-streams-lc --cross-validate=true)";
    RankFromEnvironment(&rank, &num_workers);
    SelectWorkerDevice(gpu_id, rank, num_workers, comm_transport, use_gpu);
    // the communicator before the model: RCCL initialised after the first allocations / launches leaves every later step
    // slower on this stack (measured: 1.44 vs 2.39 ms/step on the cfg2 DNN)
    std::unique_ptr<Comm> comm(NewProcessComm(comm_transport, rank, num_workers, comm_file));
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
    RandomizerMask randomizer_mask(rnd_opts);   // unused by this tool, as in the reference (aslp-nnet-train-lc-blstm-streams-worker.cc:171), but its construction seeds the generator and says so in the log
    std::unique_ptr<IWorker> worker = MakeWorker(worker_type, comm.get(), alpha, bmuf_learn_rate, bmuf_momentum, nullptr);
    std::vector<std::pair<BaseFloat *, int>> params;
    nnet.GetGpuParams(&params);
    nnet.ParamWritersAnnounce();   // the sync workers call aslp_params_changed() after every exchange
    worker->InitParam(params);
    ASLP_LOG << "Mpi cluster info total " << worker->NumNodes() << " worker rank " << worker->Rank();
    int32 num_frames_since_last_sync = 0;
    Timer time;
    ASLP_LOG << "TRAINING STARTED";
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
      num_frames_since_last_sync += frame_progress;
      if (num_frames_since_last_sync > sync_period) {
        ASLP_LOG << "Worker " << worker->Rank() << " synchronize once";
        worker->Synchronize(num_frames_since_last_sync);
        num_frames_since_last_sync = 0;
      }
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
    worker->Stop();
    {
      std::vector<double *> acc_params;
      std::vector<std::pair<double *, int>> data_params;
      nnet.GetAccStats(&acc_params, &data_params);
      worker->ReduceAccStat(acc_params, data_params);
    }
    StreamSync();
    if (worker->IsMainNode()) nnet.Write(target_model_filename, binary);
    ASLP_LOG << "Done " << num_done << " files, " << num_no_tgt_mat << " with no tgt_mats, " << num_other_error << " with other errors. "
             << "[" << (crossvalidate ? "CROSS-VALIDATION" : "TRAINING") << ", " << (randomize ? "RANDOMIZED" : "NOT-RANDOMIZED") << ", "
             << time.Elapsed() / 60 << " min, fps" << total_frames / time.Elapsed() << "]";
    if (objective_function == "xent") ASLP_LOG << xent.Report();
    else if (objective_function == "mse") ASLP_LOG << mse.Report();
    else ASLP_ERR << "Unknown objective function code : " << objective_function;
    worker.reset();
    comm.reset();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-server -- src/aslp-parallelbin/aslp-nnet-train-server.cc: the parameter server of the easgd / asgd / masgd
// protocols, rank 0 of the group.  Holds the model on its own GPU, serves the workers in arrival order until each has
// reported that it is finished, joins the final BatchNormalization statistics reduction and writes the model.
int Main_aslp_nnet_train_server(int argc, char *argv[]) {
  using namespace aslp;
  try {
    const char *usage =
        "Parameter server for training, it can adapt all kinds of wokers,"
        "eg framewise, sequential and stream training\n"
        "Usage:  aslp-nnet-train-server [options] <model-in> <model-out>\n"
        "e.g.: \n"
        " aslp-nnet-train-server nnet.init nnet.out\n";
    ParseOptions po(usage);
    bool binary = true;
    po.Register("binary", &binary, "Write output in binary mode");
    std::string use_gpu = "yes";
    po.Register("use-gpu", &use_gpu, "yes|no|optional, only has effect if compiled with CUDA");
    std::string server_type = "easgd";
    po.Register("server-type", &server_type, "Server type(easgd | asgd)");
    float alpha = 0.5;
    po.Register("alpha", &alpha, "Moving rate alpha for easgd server");
    int32 sync_period = 1000;
    po.Register("sync-period", &sync_period, "Synchronization period for ASGD");
    int32 gpu_id = -1;
    po.Register("gpu-id", &gpu_id, "selected gpu id, if negative then select automaticly");
    float masgd_momentum = 0.9;
    po.Register("masgd-momentum", &masgd_momentum, "momentum for masgd");
    int32 rank = -1, num_workers = -1;
    po.Register("rank", &rank, "Rank of the server (default: from the launcher's environment; must be 0)");
    po.Register("num-workers", &num_workers, "Size of the group, server included (default: from the launcher's environment)");
    std::string comm_file = "";
    po.Register("comm-file", &comm_file, "Rendezvous file for the RCCL communicator");
    std::string comm_transport = "";
    po.Register("comm-transport", &comm_transport, "rccl (one GPU per worker, default) | shm (workers may share a GPU: tensors staged through shared memory); default from ASLP_COMM_TRANSPORT");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) { po.PrintUsage(); exit(1); }
    std::string model_filename = po.GetArg(1), target_model_filename = po.GetArg(2);
    RankFromEnvironment(&rank, &num_workers);
    if (rank != 0) ASLP_ERR << "the parameter server is rank 0 of the group (got rank " << rank << ")";
    SelectWorkerDevice(gpu_id, rank, num_workers, comm_transport, use_gpu);
    std::unique_ptr<Comm> comm(NewProcessComm(comm_transport, rank, num_workers, comm_file));

    Nnet nnet;
    nnet.Read(model_filename);
    std::unique_ptr<IServer> server;
    if (server_type == "easgd") server.reset(new EasgdServer(comm.get(), alpha));
    else if (server_type == "asgd") server.reset(new AsgdServer(comm.get(), alpha, sync_period));
    else if (server_type == "masgd") server.reset(new AsgdServer(comm.get(), 1.0f, sync_period, true, masgd_momentum));
    else ASLP_ERR << "Unsupported server type: " << server_type;
    std::vector<std::pair<BaseFloat *, int>> params;
    nnet.GetGpuParams(&params);
    nnet.ParamWritersAnnounce();   // the sync workers call aslp_params_changed() after every exchange
    server->InitParam(params);
    ASLP_LOG << "Mpi cluster info total " << server->NumNodes() << " server rank " << server->Rank();
    server->Run();
    std::vector<double *> acc_params;
    std::vector<std::pair<double *, int>> data_params;
    nnet.GetAccStats(&acc_params, &data_params);
    server->ReduceAccStat(acc_params, data_params);
    nnet.Write(target_model_filename, binary);
    StreamSync();
    CuDevice::Instantiate().PrintProfile();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << e.what();
    return -1;
  }
}

// ======================================================================================================================
// aslp-nnet-train-simple-mpi -- src/aslp-parallelbin/aslp-nnet-train-simple-mpi.cc: aslp-nnet-train-simple run as exactly two
// ranks that average their models every --sync-period minibatches (PairSync).  "JOB" in the feature rspecifier becomes the
// rank; rank 0 writes the model.  Usage text and flags are train-simple's (as in the reference) plus --sync-period.
namespace {
class PairSimpleSync : public SimpleSync {
 public:
  PairSimpleSync() : sync_period_(1), rank_(-1), num_workers_(-1), num_minibatch_(0) {}
  void Register(aslp::ParseOptions *po) {
    po->Register("sync-period", &sync_period_, "every n minibatch(sync_period) sync once");
    po->Register("rank", &rank_, "Rank of this process (default: from the launcher's environment)");
    po->Register("num-workers", &num_workers_, "Number of processes, must be 2 (default: from the launcher's environment)");
    po->Register("comm-file", &comm_file_, "Rendezvous file for the RCCL communicator");
  }
  void Connect() {
    using namespace aslp;
    RankFromEnvironment(&rank_, &num_workers_);
    if (num_workers_ != 2) ASLP_ERR << "num of jobs must be 2";
    SelectWorkerDevice(-1, rank_, num_workers_, "", "yes");   // one GPU per rank; over shared memory the two ranks may share one
    comm_.reset(NewProcessComm("", rank_, num_workers_, comm_file_));
    pair_.reset(new PairSync(comm_.get()));
  }
  void Init(aslp::Nnet *nnet, std::string *feature_rspecifier) {
    std::vector<std::pair<aslp::BaseFloat *, int>> params;
    nnet->GetGpuParams(&params);
    nnet->ParamWritersAnnounce();
    pair_->Init(params);
    const std::string rank = std::to_string(pair_->Rank());
    for (size_t pos = 0; (pos = feature_rspecifier->find("JOB", pos)) != std::string::npos; pos += rank.size()) feature_rspecifier->replace(pos, 3, rank);
    ASLP_LOG << "MPI Rank " << pair_->Rank();
    ASLP_LOG << "Train scp " << *feature_rspecifier;
  }
  void AfterMinibatch() {
    if (++num_minibatch_ % sync_period_ == 0) {
      ASLP_LOG << "MPI sync on " << num_minibatch_;
      pair_->Sync();
    }
  }
  void Finish() {  // keep answering the peer's exchanges until it is done as well (:362-368)
    pair_->SyncStatus();
    pair_->SetSelfDone();
    pair_->SyncStatus();
    do { pair_->Sync(); } while (!pair_->AllDone());
    pair_->SyncStatus();
  }
  bool WritesModel() const { return pair_->Rank() == 0; }

 private:
  aslp::int32 sync_period_, rank_, num_workers_;
  long num_minibatch_;
  std::string comm_file_;
  std::unique_ptr<aslp::Comm> comm_;
  std::unique_ptr<aslp::PairSync> pair_;
};
}  // namespace

int Main_aslp_nnet_train_simple_mpi(int argc, char *argv[]) {
  PairSimpleSync sync;
  return TrainSimpleWithSync(argc, argv, &sync);
}
