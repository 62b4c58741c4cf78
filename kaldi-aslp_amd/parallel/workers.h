// workers.h -- seam B6: the model-sync interface of src/aslp-parallel/itf.h:26-42 and the three collective workers,
// BspWorker (bsp-worker.cc:33-65), BmufWorker (bmuf-worker.cc:37-68) and SodWorker (sod-worker.cc:36-68, with the six
// solvers of optimizer.h), on a Comm (comm.h).  The workers alias the
// model's device memory (GetGpuParams) and never own it; the reference copies every tensor to the host, calls
// MPI_Allreduce per tensor and copies back -- here the tensors are reduced where they live, in one grouped collective.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "comm.h"

namespace aslp {

class IWorker {
 public:
  explicit IWorker(Comm *comm) : comm_(comm) {}
  virtual ~IWorker() {}
  virtual void InitParam(const std::vector<std::pair<BaseFloat *, int>> &params) = 0;
  // num_worker_samples: frames processed since the last synchronisation; false once every worker has run out of data
  virtual bool Synchronize(int num_worker_samples) = 0;
  virtual void Stop() = 0;  // called when this worker's data is finished: keeps joining the others' synchronisations
  int Rank() const { return comm_->Rank(); }
  int NumNodes() const { return comm_->NumNodes(); }
  int MainNode() const { return 0; }
  bool IsMainNode() const { return comm_->Rank() == 0; }
  void Barrier() { comm_->Barrier(); }
  // global BatchNormalization statistics at the end of an epoch (mpi-node.h:77-93)
  void ReduceAccStat(const std::vector<double *> &acc_params, const std::vector<std::pair<double *, int>> &data_params);

 protected:
  Comm *comm_;
};

class BspWorker : public IWorker {
 public:
  explicit BspWorker(Comm *comm) : IWorker(comm) {}
  void InitParam(const std::vector<std::pair<BaseFloat *, int>> &params) { params_ = params; }
  bool Synchronize(int num_worker_samples);
  void Stop();

 private:
  std::vector<std::pair<BaseFloat *, int>> params_;
};

class BmufWorker : public IWorker {
 public:
  BmufWorker(Comm *comm, float learn_rate, float momentum) : IWorker(comm), learn_rate_(learn_rate), momentum_(momentum) {}
  ~BmufWorker();
  void InitParam(const std::vector<std::pair<BaseFloat *, int>> &params);
  bool Synchronize(int num_worker_samples);
  void Stop();

 private:
  float learn_rate_, momentum_;
  std::vector<std::pair<BaseFloat *, int>> params_;
  std::vector<std::pair<BaseFloat *, int>> prev_, prev_grad_, grad_;  // w_g(t-1), d(t-1), work buffers (owned)
};

// optimizer.h:173-230: which solver SOD runs on the summed model deltas, with the reference's flag names and defaults
struct OptimizerOption {
  std::string solver = "momentum";
  float lr = 0.01f, momentum = 0.9f, adagrad_lr = 0.01f, rmsprop_lr = 0.001f, adam_lr = 0.001f, adadelta_gamma = 0.95f, adam_beta1 = 0.9f,
        adam_beta2 = 0.999f;
  template <class Opts>
  void Register(Opts *opts) {
    opts->Register("solver", &solver, "Optimizer solver(sgd | momentum | adagrad | adadelta | rmsprop | adam)");
    opts->Register("lr", &lr, "learning rate for (sgd | momentum) optimizer");
    opts->Register("sgd-momentum", &momentum, "momentum for (momentum) optimizer");
    opts->Register("adagrad-lr", &adagrad_lr, "learning rate for (adagrad) optimizer");
    opts->Register("rmsprop-lr", &rmsprop_lr, "learning rate for (rmsprop) optimizer");
    opts->Register("adam-lr", &adam_lr, "learning rate for (adam) optimizer");
    opts->Register("adadelta-gamma", &adadelta_gamma, "update factor for (adadelta) optimizer");
    opts->Register("adam-beta1", &adam_beta1, "update mean factor for (adam) optimizer");
    opts->Register("adam-beta2", &adam_beta2, "update variance factor for (adam) optimizer");
  }
};

// "Synchronous Optimize the Difference between global and local model" (sod-worker.h:17): the workers' model deltas since the
// last sync are summed and handed to a solver as if they were a gradient; the solver steps each worker's OWN current model
// (the models are not re-unified -- they stay apart by whatever separated them at the start, as in the reference).
class SodWorker : public IWorker {
 public:
  SodWorker(Comm *comm, const OptimizerOption &config);
  ~SodWorker();
  void InitParam(const std::vector<std::pair<BaseFloat *, int>> &params);
  bool Synchronize(int num_worker_samples);
  void Stop();

 private:
  OptimizerOption config_;
  int solver_, step_;  // step_: Adam's t (optimizer.h:150), one counter serves every tensor since all advance together
  std::vector<std::pair<BaseFloat *, int>> params_;
  std::vector<std::pair<BaseFloat *, int>> prev_, grad_, state1_, state2_;  // owned
};

// nnet-mpi-sync.cc:17-135 (NnetMpiSync, the synchroniser inside aslp-nnet-train-simple-mpi): exactly two ranks swap their whole
// models and their "my data is finished" flags; while both still train each keeps the average, a finished rank adopts the
// peer's model, and a rank whose peer has finished keeps its own.
class PairSync {
 public:
  explicit PairSync(Comm *comm);
  ~PairSync();
  void Init(const std::vector<std::pair<BaseFloat *, int>> &params);
  void Sync();
  void SetSelfDone() { self_done_ = 1; }
  bool SelfDone() const { return self_done_ > 0; }
  bool PeerDone() const { return peer_done_ > 0; }
  bool AllDone() const { return self_done_ > 0 && peer_done_ > 0; }
  int Rank() const { return comm_->Rank(); }
  void SyncStatus() const;  // the reference's three-column status line on stderr

 private:
  Comm *comm_;
  int peer_, self_done_, peer_done_;
  std::vector<std::pair<BaseFloat *, int>> params_, peer_params_;
};

// ---- server-based protocols: rank 0 is a parameter server (aslp-nnet-train-server), ranks 1.. are workers ----------------
enum { kMsgSynchronize = 0, kMsgFinished = 1 };  // itf.h:19-22

// easgd-worker.cc:37-80: the worker and the server swap models; x_w = (1 - alpha) x_w + alpha x_s
class EasgdWorker : public IWorker {
 public:
  EasgdWorker(Comm *comm, float alpha) : IWorker(comm), alpha_(alpha) {}
  ~EasgdWorker();
  void InitParam(const std::vector<std::pair<BaseFloat *, int>> &params);
  bool Synchronize(int num_worker_samples);
  void Stop();

 private:
  float alpha_;
  std::vector<std::pair<BaseFloat *, int>> params_, server_;  // server_: the server's model as received (owned)
};

// asgd-worker.cc:37-71 (also the worker of MasgdServer): sends its model DELTA since the last exchange, takes the server's model
class AsgdWorker : public IWorker {
 public:
  explicit AsgdWorker(Comm *comm) : IWorker(comm) {}
  ~AsgdWorker();
  void InitParam(const std::vector<std::pair<BaseFloat *, int>> &params);
  bool Synchronize(int num_worker_samples);
  void Stop();

 private:
  std::vector<std::pair<BaseFloat *, int>> params_, prev_, delta_;
};

class IServer : public IWorker {  // itf.h:44-50
 public:
  explicit IServer(Comm *comm) : IWorker(comm) {}
  bool Synchronize(int) { return false; }
  void Stop() {}
  virtual void Run() = 0;  // serves until every worker has sent kMsgFinished
};

// easgd-server.cc:37-86: x_s = (1 - alpha) x_s + alpha x_w, one worker at a time in arrival order
class EasgdServer : public IServer {
 public:
  EasgdServer(Comm *comm, float alpha) : IServer(comm), alpha_(alpha) {}
  ~EasgdServer();
  void InitParam(const std::vector<std::pair<BaseFloat *, int>> &params);
  void Run();

 private:
  float alpha_;
  std::vector<std::pair<BaseFloat *, int>> params_, worker_;
};

// asgd-server.cc:39-102 / masgd-server.cc:39-118 (per-worker momentum buffers, the LMASGD variant the reference compiles):
// x_s += alpha * delta  |  d_k = delta + momentum * d_k, x_s += d_k.  sync_period > 0: after that many exchanges the server
// stops answering until every running worker waits, then answers them all with the same model.
class AsgdServer : public IServer {
 public:
  AsgdServer(Comm *comm, float alpha, int sync_period, bool masgd = false, float momentum = 0.0f)
      : IServer(comm), alpha_(alpha), momentum_(momentum), sync_period_(sync_period), masgd_(masgd) {}
  ~AsgdServer();
  void InitParam(const std::vector<std::pair<BaseFloat *, int>> &params);
  void Run();

 private:
  float alpha_, momentum_;
  int sync_period_;
  bool masgd_;
  std::vector<std::pair<BaseFloat *, int>> params_, delta_;
  std::vector<std::vector<std::pair<BaseFloat *, int>>> diffs_;  // masgd: one per worker
};

}  // namespace aslp
