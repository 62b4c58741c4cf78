// workers.cpp -- see workers.h.
#include "workers.h"

#include <hip/hip_runtime.h>

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
  ASLP_LOG << "Worker " << Rank() << "finished, waitting for others";
  while (Synchronize(0)) {}
}

}  // namespace aslp
