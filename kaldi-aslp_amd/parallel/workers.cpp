// workers.cpp -- see workers.h.
#include "workers.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <iostream>

#include "aslp_kernels.h"
#include "common.h"

namespace aslp {

static void Hip(hipError_t e, const char *what) { if (e != hipSuccess) ASLP_ERR << what << ": " << hipGetErrorString(e); }
static void CheckK() { char buf[512]; if (aslp_get_last_error(buf, sizeof(buf))) ASLP_ERR << buf; }
static const aslp_dim3 kD3 = {1, 1, 1};
static void Scale(float *v, int n, float a) { if (n) { MatrixDim d = {1, n, n}; cudaF_scale(kD3, kD3, v, a, d); } }
static void Copy(float *dst, const float *src, int n) {
  if (n) Hip(hipMemcpyAsync(dst, src, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, cur_stream()), "hipMemcpy D2D");
}

void IWorker::ReduceAccStat(const std::vector<double *> &acc_params, const std::vector<std::pair<double *, int>> &data_params) {
  comm_->Barrier();
  for (double *p : acc_params) comm_->AllReduceSumHost(p, 1);
  for (auto &d : data_params) comm_->AllReduceSum(d.first, (size_t)d.second);
  Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
}

// ---- BSP (bsp-worker.cc:33-65) ------------------------------------------------------------------------------------
bool BspWorker::Synchronize(int num_worker_samples) {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  int32 num_all_samples = num_worker_samples;
  comm_->AllReduceSumHost(&num_all_samples, 1);
  if (num_all_samples <= 0) { ASLP_LOG << "All worker finished their data"; return false; }
  const float factor = float(num_worker_samples) / num_all_samples;
  ASLP_ASSERT(factor >= 0.0 && factor <= 1.0);
  for (auto &p : params_) Scale(p.first, p.second, factor);   // theta_k * n_k / sum n
  CheckK();
  comm_->AllReduceSumMany(params_);                           // sum over workers, in HBM, one grouped collective
  return true;
}
void BspWorker::Stop() {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  ASLP_LOG << "Worker " << Rank() << "finished, waitting for others";
  while (Synchronize(0)) {}
}

// ---- BMUF (bmuf-worker.cc:37-68) ----------------------------------------------------------------------------------
static std::vector<std::pair<BaseFloat *, int>> AllocLike(const std::vector<std::pair<BaseFloat *, int>> &params, bool copy) {
  std::vector<std::pair<BaseFloat *, int>> out;
  for (auto &p : params) {
    void *d = nullptr;
    Hip(hipMalloc(&d, sizeof(float) * (size_t)(p.second > 0 ? p.second : 1)), "hipMalloc");
    if (copy) Copy(static_cast<float *>(d), p.first, p.second);
    else if (p.second) Hip(hipMemsetAsync(d, 0, sizeof(float) * (size_t)p.second, cur_stream()), "hipMemset");
    out.push_back(std::make_pair(static_cast<float *>(d), p.second));
  }
  return out;
}
void BmufWorker::InitParam(const std::vector<std::pair<BaseFloat *, int>> &params) {
  params_ = params;
  prev_ = AllocLike(params, true);        // w_g(t-1) = the initial model
  prev_grad_ = AllocLike(params, false);  // d(t-1) = 0
  grad_ = AllocLike(params, false);
}
BmufWorker::~BmufWorker() {
  for (auto *v : {&prev_, &prev_grad_, &grad_})
    for (auto &p : *v) (void)hipFree(p.first);
}
bool BmufWorker::Synchronize(int num_worker_samples) {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  int32 num_all_samples = num_worker_samples;
  comm_->AllReduceSumHost(&num_all_samples, 1);
  if (num_all_samples <= 0) { ASLP_LOG << "All worker finished their data"; return false; }
  for (size_t i = 0; i < params_.size(); i++) {  // 1. g = w(t) - w_g(t-1)
    Copy(grad_[i].first, params_[i].first, params_[i].second);
    aslp_vec_axpy(-1.0f, prev_[i].first, grad_[i].first, params_[i].second);
  }
  CheckK();
  comm_->AllReduceSumMany(grad_);                // 2./3. summed over workers (a sum, not a mean: :53)
  const float lr = (1.0 - momentum_) * learn_rate_;
  for (size_t i = 0; i < params_.size(); i++) {
    const int n = params_[i].second;
    Scale(grad_[i].first, n, lr);                                    // 4. d(t) = m d(t-1) + (1 - m) lr g
    aslp_vec_axpy(momentum_, prev_grad_[i].first, grad_[i].first, n);
    Copy(params_[i].first, prev_[i].first, n);                       // 5. w(t) = w_g(t-1) + d(t)
    aslp_vec_axpy(1.0f, grad_[i].first, params_[i].first, n);
    Copy(prev_[i].first, params_[i].first, n);                       // 6.
    Copy(prev_grad_[i].first, grad_[i].first, n);
  }
  CheckK();
  return true;
}
void BmufWorker::Stop() {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  ASLP_LOG << "Worker " << Rank() << "finished, waitting for others";
  while (Synchronize(0)) {}
}

// ---- SOD (sod-worker.cc:36-68) -----------------------------------------------------------------------------------
SodWorker::SodWorker(Comm *comm, const OptimizerOption &config) : IWorker(comm), config_(config), solver_(-1), step_(1) {
  const char *names[] = {"sgd", "momentum", "adagrad", "rmsprop", "adadelta", "adam"};
  for (int k = 0; k < 6; k++)
    if (config_.solver == names[k]) solver_ = k;
  if (solver_ < 0) ASLP_ERR << "Unknown solver type " << config_.solver;
}
void SodWorker::InitParam(const std::vector<std::pair<BaseFloat *, int>> &params) {
  params_ = params;
  prev_ = AllocLike(params, true);  // w(t-1) = the initial model
  grad_ = AllocLike(params, false);
  if (solver_ != ASLP_SOD_SGD) state1_ = AllocLike(params, false);
  if (solver_ == ASLP_SOD_ADADELTA || solver_ == ASLP_SOD_ADAM) state2_ = AllocLike(params, false);
}
SodWorker::~SodWorker() {
  for (auto *v : {&prev_, &grad_, &state1_, &state2_})
    for (auto &p : *v) (void)hipFree(p.first);
}
bool SodWorker::Synchronize(int num_worker_samples) {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  int32 num_all_samples = num_worker_samples;
  comm_->AllReduceSumHost(&num_all_samples, 1);
  if (num_all_samples <= 0) { ASLP_LOG << "All worker finished their data"; return false; }
  for (size_t i = 0; i < params_.size(); i++) aslp_vec_diff(grad_[i].first, prev_[i].first, params_[i].first, params_[i].second);  // 1. w(t-1) - w(t)
  CheckK();
  comm_->AllReduceSumMany(grad_);  // 2./3. summed over workers where they live
  aslp_sod_solver a = {};
  a.solver = solver_;
  a.lr = solver_ == ASLP_SOD_ADAGRAD ? config_.adagrad_lr : solver_ == ASLP_SOD_RMSPROP ? config_.rmsprop_lr : solver_ == ASLP_SOD_ADAM ? config_.adam_lr : config_.lr;
  a.momentum = config_.momentum;
  a.gamma = config_.adadelta_gamma;
  a.beta1 = config_.adam_beta1;
  a.beta2 = config_.adam_beta2;
  a.corr1 = 1.0 / (1 - pow(a.beta1, step_));  // optimizer.h:158-159
  a.corr2 = 1.0 / (1 - pow(a.beta2, step_));
  for (size_t i = 0; i < params_.size(); i++)  // 4. solver step on the local model, 5. which becomes the new reference point
    aslp_sod_solve(&a, grad_[i].first, params_[i].first, prev_[i].first, state1_.empty() ? nullptr : state1_[i].first,
                   state2_.empty() ? nullptr : state2_[i].first, params_[i].second);
  CheckK();
  step_++;
  return true;
}
void SodWorker::Stop() {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  ASLP_LOG << "Worker " << Rank() << "finished, waitting for others";
  while (Synchronize(0)) {}
}

// ---- EASGD (easgd-worker.cc:37-80, easgd-server.cc:37-86) --------------------------------------------------------
static void FreeAll(std::vector<std::pair<BaseFloat *, int>> *v) {
  for (auto &p : *v) (void)hipFree(p.first);
  v->clear();
}
// y = a * x + b * y per tensor (CuVectorBase::AddVec(a, x, b))
static void AddVec(const std::vector<std::pair<BaseFloat *, int>> &y, float a, const std::vector<std::pair<BaseFloat *, int>> &x, float b) {
  for (size_t i = 0; i < y.size(); i++) {
    if (b != 1.0f) Scale(y[i].first, y[i].second, b);
    aslp_vec_axpy(a, x[i].first, y[i].first, y[i].second);
  }
  CheckK();
}
void EasgdWorker::InitParam(const std::vector<std::pair<BaseFloat *, int>> &params) { params_ = params; server_ = AllocLike(params, false); }
EasgdWorker::~EasgdWorker() { FreeAll(&server_); }
bool EasgdWorker::Synchronize(int) {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  comm_->PostToServer(kMsgSynchronize);
  comm_->Exchange(MainNode(), params_, server_);          // both sides hand over their model as it is now
  AddVec(params_, alpha_, server_, 1.0f - alpha_);         // x_w = (1 - alpha) x_w + alpha x_s
  return true;
}
void EasgdWorker::Stop() {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  comm_->PostToServer(kMsgFinished);
  ASLP_LOG << "Worker " << Rank() << " finished";
}
void EasgdServer::InitParam(const std::vector<std::pair<BaseFloat *, int>> &params) { params_ = params; worker_ = AllocLike(params, false); }
EasgdServer::~EasgdServer() { FreeAll(&worker_); }
void EasgdServer::Run() {
  int num_running_workers = NumNodes() - 1;
  while (num_running_workers > 0) {
    int worker_rank;
    int32 msg;
    comm_->WaitFromWorker(&worker_rank, &msg);
    if (msg == kMsgFinished) {
      num_running_workers--;
      ASLP_LOG << "Worker " << worker_rank << " Finished ";
    } else if (msg == kMsgSynchronize) {
      comm_->Exchange(worker_rank, params_, worker_);
      AddVec(params_, alpha_, worker_, 1.0f - alpha_);     // x_s = (1 - alpha) x_s + alpha x_w
    } else {
      ASLP_WARN << "Unknown mpi msg type " << msg;
    }
  }
  Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
  ASLP_LOG << "All worker finished";
}

// ---- pairwise averaging (nnet-mpi-sync.cc:17-135) ------------------------------------------------------------------
PairSync::PairSync(Comm *comm) : comm_(comm), peer_(0), self_done_(0), peer_done_(0) {
  if (comm_->NumNodes() != 2) ASLP_ERR << "num of jobs must be 2";
  peer_ = (comm_->Rank() + 1) % 2;
}
PairSync::~PairSync() { FreeAll(&peer_params_); }
void PairSync::Init(const std::vector<std::pair<BaseFloat *, int>> &params) {
  params_ = params;
  peer_params_ = AllocLike(params, false);
  size_t total = 0;
  for (auto &p : params) total += p.second;
  ASLP_LOG << "num params " << params.size();
  ASLP_LOG << "total params " << total;
}
void PairSync::Sync() {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  int32 done[2] = {0, 0};
  done[comm_->Rank()] = self_done_;
  comm_->AllReduceSumHost(done, 2);  // both flags on both ranks
  peer_done_ = done[peer_];
  comm_->Exchange(peer_, params_, peer_params_);
  if (PeerDone()) return;              // the peer only listens now: keep training on the own model
  for (size_t i = 0; i < params_.size(); i++) {
    const int n = params_[i].second;
    if (SelfDone()) {
      Copy(params_[i].first, peer_params_[i].first, n);   // out of data: follow the peer
    } else {
      aslp_vec_axpy(1.0f, peer_params_[i].first, params_[i].first, n);  // (mine + peer's) / 2
      Scale(params_[i].first, n, 0.5f);
    }
  }
  CheckK();
  Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
}
void PairSync::SyncStatus() const {
  std::cerr << "self\tpeer\tall\n" << SelfDone() << "\t" << PeerDone() << "\t" << AllDone() << "\n";
}

// ---- ASGD / MASGD (asgd-worker.cc:37-71, asgd-server.cc:39-102, masgd-server.cc:39-118) -------------------------
void AsgdWorker::InitParam(const std::vector<std::pair<BaseFloat *, int>> &params) {
  params_ = params;
  prev_ = AllocLike(params, true);
  delta_ = AllocLike(params, false);
}
AsgdWorker::~AsgdWorker() { FreeAll(&prev_); FreeAll(&delta_); }
bool AsgdWorker::Synchronize(int) {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  comm_->PostToServer(kMsgSynchronize);
  for (size_t i = 0; i < params_.size(); i++) aslp_vec_diff(delta_[i].first, params_[i].first, prev_[i].first, params_[i].second);
  CheckK();
  comm_->Send(MainNode(), delta_);
  comm_->Recv(MainNode(), params_);  // returns when the server answers: at once, or at its periodic barrier
  for (size_t i = 0; i < params_.size(); i++) Copy(prev_[i].first, params_[i].first, params_[i].second);
  return true;
}
void AsgdWorker::Stop() {
  aslp_params_changed();   // the model is (about to be) written through the GetGpuParams pointers: planes kept of it are stale
  comm_->PostToServer(kMsgFinished);
  ASLP_LOG << "Worker " << Rank() << " finished";
}
void AsgdServer::InitParam(const std::vector<std::pair<BaseFloat *, int>> &params) {
  params_ = params;
  delta_ = AllocLike(params, false);
  if (masgd_)
    for (int w = 1; w < NumNodes(); w++) diffs_.push_back(AllocLike(params, false));
}
AsgdServer::~AsgdServer() {
  FreeAll(&delta_);
  for (auto &d : diffs_) FreeAll(&d);
}
void AsgdServer::Run() {
  int num_running_workers = NumNodes() - 1, synchronized_count = 0;
  std::vector<int> waited_worker;
  while (num_running_workers > 0) {
    int worker_rank;
    int32 msg;
    comm_->WaitFromWorker(&worker_rank, &msg);
    if (msg == kMsgFinished) {
      num_running_workers--;
      ASLP_LOG << "Worker " << worker_rank << " Finished ";
    } else if (msg == kMsgSynchronize) {
      ++synchronized_count;
      if (sync_period_ > 0 && synchronized_count >= sync_period_) waited_worker.push_back(worker_rank);
      comm_->Recv(worker_rank, delta_);
      if (masgd_) {
        AddVec(diffs_[worker_rank - 1], 1.0f, delta_, momentum_);  // d_k = delta + momentum d_k
        AddVec(params_, 1.0f, diffs_[worker_rank - 1], 1.0f);
      } else {
        AddVec(params_, alpha_, delta_, 1.0f);
      }
      if (synchronized_count < sync_period_ || sync_period_ <= 0) comm_->Send(worker_rank, params_);
    } else {
      ASLP_WARN << "Unknown mpi msg type " << msg;
    }
    if (sync_period_ > 0 && synchronized_count >= sync_period_ && (int)waited_worker.size() == num_running_workers && num_running_workers != 0) {
      for (int w : waited_worker) {
        ASLP_LOG << "Worker " << w << " synchronized!";
        comm_->Send(w, params_);
      }
      synchronized_count -= sync_period_;
      waited_worker.clear();
    }
  }
  Hip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
  ASLP_LOG << "All worker finished";
}

}  // namespace aslp
