// cu-device.h -- seam B3: the slice of CuDevice (src/aslp-cudamatrix/cu-device.h:43-151, cu-device.cc:95-260) the tools
// call: SelectGpuId("yes|no|optional|wait"), SetGpuId(n) (the ASLP addition, cu-device.cc:203-224), Enabled(),
// AccuProfile / PrintProfile, CheckGpuHealth.  There is no CPU engine behind this library, so "no" -- and "optional"
// without a usable GPU -- is an error here, stated as such, instead of a silent CPU run.
#pragma once
#include <map>
#include <string>

#include "base.h"

namespace aslp {

class CuDevice {
 public:
  static CuDevice &Instantiate() { static CuDevice d; return d; }
  void SelectGpuId(const std::string &use_gpu);
  void SetGpuId(int32 gpu_id);
  bool Enabled() const { return active_gpu_id_ >= 0; }
  int32 ActiveGpuId() const { return active_gpu_id_; }
  void AccuProfile(const std::string &key, double time) { profile_map_[key] += time; }
  void PrintProfile();
  void ResetProfile() { profile_map_.clear(); }
  std::string GetFreeMemory(int64_t *free = NULL, int64_t *total = NULL) const;
  void BindThread() const;  // a new host thread starts on device 0: make it use the active GPU
  void CheckGpuHealth();  // runs a small GEMM + copy on the device and checks the result (cu-device.cc:482-510)

 private:
  CuDevice() : active_gpu_id_(-1) {}
  void FinalizeActiveGpu();
  int32 active_gpu_id_;
  std::map<std::string, double> profile_map_;
};

}  // namespace aslp
