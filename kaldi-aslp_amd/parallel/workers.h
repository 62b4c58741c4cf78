// workers.h -- seam B6: the model-sync interface of src/aslp-parallel/itf.h:26-42 and the two collective workers,
// BspWorker (bsp-worker.cc:33-65) and BmufWorker (bmuf-worker.cc:37-68), on a Comm (comm.h).  The workers alias the
// model's device memory (GetGpuParams) and never own it; the reference copies every tensor to the host, calls
// MPI_Allreduce per tensor and copies back -- here the tensors are reduced where they live, in one grouped collective.
#pragma once
#include <memory>
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

}  // namespace aslp
