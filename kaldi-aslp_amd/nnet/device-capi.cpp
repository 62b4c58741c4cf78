// device-capi.cpp -- seam B3 (include/aslp_device.h): C view of CuDevice and the allocator.
#include <cstring>

#include "aslp_device.h"
#include "cu-device.h"
#include "cu-matrix.h"

using namespace aslp;

static thread_local std::string t_dev_err;
#define DEV_BEGIN try {
#define DEV_END                      \
  }                                  \
  catch (const std::exception &e) {  \
    t_dev_err = e.what();            \
    return 1;                        \
  }                                  \
  return 0;

extern "C" {

const char *aslp_device_last_error(void) { return t_dev_err.c_str(); }
int aslp_device_select_gpu_id(const char *use_gpu) { DEV_BEGIN CuDevice::Instantiate().SelectGpuId(use_gpu ? use_gpu : ""); DEV_END }
int aslp_device_set_gpu_id(int gpu_id) { DEV_BEGIN CuDevice::Instantiate().SetGpuId(gpu_id); DEV_END }
int aslp_device_enabled(void) { return CuDevice::Instantiate().Enabled() ? 1 : 0; }
int aslp_device_active_gpu_id(void) { return CuDevice::Instantiate().ActiveGpuId(); }
void *aslp_device_malloc(size_t size) {
  try { return DeviceAlloc(size); } catch (const std::exception &e) { t_dev_err = e.what(); return nullptr; }
}
void *aslp_device_malloc_pitch(size_t row_bytes, size_t num_rows, size_t *pitch) {
  const size_t p = (row_bytes + 63) / 64 * 64;  // the engine's row padding: 16 floats (cu-matrix.cpp), rows 64-byte aligned
  if (pitch) *pitch = p;
  try { return DeviceAlloc(p * num_rows); } catch (const std::exception &e) { t_dev_err = e.what(); return nullptr; }
}
void aslp_device_free(void *ptr) { DeviceFree(ptr); }
void aslp_device_accu_profile(const char *function_name, double seconds) { CuDevice::Instantiate().AccuProfile(function_name ? function_name : "", seconds); }
void aslp_device_print_profile(void) { try { CuDevice::Instantiate().PrintProfile(); } catch (...) {} }
void aslp_device_reset_profile(void) { CuDevice::Instantiate().ResetProfile(); }
int aslp_device_check_gpu_health(void) { DEV_BEGIN CuDevice::Instantiate().CheckGpuHealth(); DEV_END }
int aslp_device_get_free_memory(char *buf, int buflen, long long *free_bytes, long long *total_bytes) {
  DEV_BEGIN
  int64_t f = 0, t = 0;
  const std::string s = CuDevice::Instantiate().GetFreeMemory(&f, &t);
  if (buf && buflen > 0) { std::strncpy(buf, s.c_str(), buflen - 1); buf[buflen - 1] = 0; }
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  DEV_END
}

}  // extern "C"
