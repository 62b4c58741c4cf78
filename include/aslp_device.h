/*
 * aslp_device.h -- seam B3 of SURVEY.md §8b: a C view of the reference's CuDevice singleton
 * (src/aslp-cudamatrix/cu-device.h:43-151, cu-device.cc) -- device selection behind --use-gpu / --gpu-id, the allocator the
 * CuMatrix / CuVector constructors call, the per-function profile and the health check -- over the engine's own
 * implementation (kaldi-aslp_amd/nnet/cu-device.cpp, the caching allocator of nnet/cu-matrix.cpp).
 * The reference's methods throw through KALDI_ERR; these return 0 on success and non-zero on error with the message in
 * aslp_device_last_error().  There is no CPU engine behind this library: "no", and "optional" without a usable GPU, fail.
 */
#ifndef ASLP_DEVICE_H_
#define ASLP_DEVICE_H_
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

const char *aslp_device_last_error(void);
int aslp_device_select_gpu_id(const char *use_gpu);   /* CuDevice::SelectGpuId("yes|no|optional|wait")  cu-device.h:67, cu-device.cc:95-201 */
int aslp_device_set_gpu_id(int gpu_id);               /* CuDevice::SetGpuId (ASLP)                      cu-device.h:69, cu-device.cc:203-224 */
int aslp_device_enabled(void);                        /* CuDevice::Enabled                              cu-device.h:72-74 */
int aslp_device_active_gpu_id(void);                  /* CuDevice::ActiveGpuId                          cu-device.h:63 */
/* CuDevice::Malloc / MallocPitch / Free (cu-device.h:52-59 -> CuMemoryAllocator): device memory from the caching allocator.
 * MallocPitch rounds the row size up like the reference's pitch allocation (rows start 64-byte aligned); *pitch in bytes. */
void *aslp_device_malloc(size_t size);
void *aslp_device_malloc_pitch(size_t row_bytes, size_t num_rows, size_t *pitch);
void aslp_device_free(void *ptr);
void aslp_device_accu_profile(const char *function_name, double seconds);   /* CuDevice::AccuProfile  cu-device.h:80 */
void aslp_device_print_profile(void);                                       /* CuDevice::PrintProfile cu-device.h:81 */
void aslp_device_reset_profile(void);                                       /* CuDevice::ResetProfile cu-device.h:83 */
int aslp_device_check_gpu_health(void);                                     /* CuDevice::CheckGpuHealth cu-device.h:104 */
/* "free:...M, used:...M, total:...M, free/total:..." like CuDevice::GetFreeMemory (cu-device.h:93) */
int aslp_device_get_free_memory(char *buf, int buflen, long long *free_bytes, long long *total_bytes);

#ifdef __cplusplus
}
#endif
#endif
