// cu-device.cpp -- see cu-device.h.
#include "cu-device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "cu-matrix.h"

namespace aslp {

void CuDevice::FinalizeActiveGpu() {  // cu-device.cc:226-260
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) ASLP_ERR << "hipGetDevice failed";
  active_gpu_id_ = dev;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess)
    ASLP_LOG << "The active GPU is [" << dev << "]: " << prop.name << "\t" << GetFreeMemory() << " version " << prop.major << "." << prop.minor;
}

std::string CuDevice::GetFreeMemory(int64_t *free_out, int64_t *total_out) const {
  size_t mem_free = 0, mem_total = 0;
  (void)hipMemGetInfo(&mem_free, &mem_total);
  if (free_out) *free_out = (int64_t)mem_free;
  if (total_out) *total_out = (int64_t)mem_total;
  std::ostringstream os;
  os << "free:" << mem_free / (1024 * 1024) << "M, used:" << (mem_total - mem_free) / (1024 * 1024) << "M, total:" << mem_total / (1024 * 1024)
     << "M, free/total:" << (mem_total ? mem_free / (float)mem_total : 0.0f);
  return os.str();
}

void CuDevice::SelectGpuId(const std::string &use_gpu) {
  if (use_gpu != "yes" && use_gpu != "no" && use_gpu != "optional" && use_gpu != "wait")
    ASLP_ERR << "Please choose : --use-gpu=yes|no|optional|wait, passed '" << use_gpu << "'";
  if (Enabled()) ASLP_ERR << "There is already an active GPU " << active_gpu_id_ << ", cannot change it on the fly!";
  if (use_gpu == "no") ASLP_ERR << "--use-gpu=no: this build has no CPU compute path (MI355X-only engine)";
  int num_gpus = 0;
  hipError_t e = hipGetDeviceCount(&num_gpus);
  if (e != hipSuccess || num_gpus == 0)
    ASLP_ERR << "No HIP GPU detected!" << (use_gpu == "optional" ? " (--use-gpu=optional cannot fall back: no CPU compute path)" : "");
  // the device with the largest proportion of free memory (SelectGpuIdAuto, cu-device.cc:300-390)
  int best = -1;
  float best_ratio = -1.0f;
  for (int n = 0; n < num_gpus; n++) {
    if (hipSetDevice(n) != hipSuccess) { (void)hipGetLastError(); continue; }
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess || mem_total == 0) { (void)hipGetLastError(); continue; }
    const float ratio = mem_free / (float)mem_total;
    ASLP_LOG << "hipSetDevice(" << n << "): free:" << mem_free / (1024 * 1024) << "M, total:" << mem_total / (1024 * 1024) << "M, free/total:" << ratio;
    if (ratio > best_ratio) { best_ratio = ratio; best = n; }
  }
  if (best < 0) ASLP_ERR << "Error acquiring GPU.";
  if (hipSetDevice(best) != hipSuccess) ASLP_ERR << "hipSetDevice(" << best << ") failed";
  ASLP_LOG << "Selected device: " << best << " (automatically)";
  FinalizeActiveGpu();
}

void CuDevice::SetGpuId(int32 gpu_id) {  // cu-device.cc:203-224
  if (Enabled()) ASLP_ERR << "There is already an active GPU " << active_gpu_id_ << ", cannot change it on the fly!";
  int num_gpus = 0;
  if (hipGetDeviceCount(&num_gpus) != hipSuccess || num_gpus == 0) ASLP_ERR << "No HIP GPU detected!";
  if (gpu_id < 0 || gpu_id >= num_gpus) ASLP_ERR << "Invalid gpu id " << gpu_id << ", number of GPUs is " << num_gpus;
  if (hipSetDevice(gpu_id) != hipSuccess) ASLP_ERR << "hipSetDevice(" << gpu_id << ") failed";
  ASLP_LOG << "Selected device: " << gpu_id << " (manually)";
  FinalizeActiveGpu();
}

void CuDevice::BindThread() const {
  if (active_gpu_id_ >= 0 && hipSetDevice(active_gpu_id_) != hipSuccess) ASLP_ERR << "hipSetDevice(" << active_gpu_id_ << ") failed";
}

void CuDevice::PrintProfile() {  // cu-device.cc:440-470
  if (!Enabled() && profile_map_.empty()) return;
  std::ostringstream os;
  os << "-----\n[cudevice profile]\n";
  std::vector<std::pair<double, std::string>> pairs;
  for (auto &kv : profile_map_) pairs.push_back(std::make_pair(kv.second, kv.first));
  std::sort(pairs.begin(), pairs.end());
  size_t max_print = 15, start = pairs.size() > max_print ? pairs.size() - max_print : 0;
  for (size_t i = start; i < pairs.size(); i++) os << pairs[i].second << "\t" << pairs[i].first << "s\n";
  os << "-----";
  ASLP_LOG << os.str();
  ASLP_LOG << "Memory used: " << GetFreeMemory();
}

void CuDevice::CheckGpuHealth() {
  if (!Enabled()) return;
  const int n = 64;
  HostMatrix a(n, n), b(n, n), c;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) { a(i, j) = (float)((i + 2 * j) % 7) - 3.0f; b(i, j) = (float)((3 * i + j) % 5) - 2.0f; }
  CuMatrix A, B, C(n, n);
  A = a;
  B = b;
  C.AddMatMat(1.0f, A, kNoTrans, B, kTrans, 0.0f);
  C.CopyToMat(&c);
  for (int i = 0; i < n; i += 9)
    for (int j = 0; j < n; j += 7) {
      float ref = 0.0f;
      for (int k = 0; k < n; k++) ref += a(i, k) * b(j, k);
      if (std::fabs(ref - c(i, j)) > 1e-3f) ASLP_ERR << "GPU health check failed: product mismatch at (" << i << "," << j << "): " << c(i, j) << " vs " << ref;
    }
}

}  // namespace aslp
