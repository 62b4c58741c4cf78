// runtime.cpp -- per-thread stream, sticky error slot and device scratch for libaslp_hip.so.
// Replaces the slice of CuDevice (src/aslp-cudamatrix/cu-device.h:43-151) the hot path needs.
#include <mutex>
#include <string>
#include <cstring>

#include "aslp_kernels.h"
#include "common.h"
#include "scratch.h"

namespace aslp {

static thread_local hipStream_t t_stream = nullptr;
static std::mutex g_err_mu;
static std::string g_err;

hipStream_t cur_stream() { return t_stream; }
void set_cur_stream(hipStream_t s) { t_stream = s; }

void set_error(const std::string &msg) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  if (g_err.empty()) g_err = msg;  // keep the first one: it is the cause
}
bool has_error() {
  std::lock_guard<std::mutex> lk(g_err_mu);
  return !g_err.empty();
}

// One grow-only scratch arena per slot.  Kernels on one stream run in order, so a slot can
// be reused by the next op on the same stream without synchronisation (the reference's
// caching allocator exists for the same reason: cu-allocator.h:67-70).
static void *g_scratch[kNumScratch] = {nullptr};
static size_t g_scratch_bytes[kNumScratch] = {0};
static std::mutex g_scratch_mu;

void *scratch(int slot, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (bytes > g_scratch_bytes[slot]) {
    if (g_scratch[slot]) {
      // outstanding kernels may still use the old block
      (void)hipStreamSynchronize(cur_stream());
      (void)hipFree(g_scratch[slot]);
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      set_error(std::string("hipMalloc(scratch): ") + hipGetErrorString(e));
      g_scratch[slot] = nullptr;
      g_scratch_bytes[slot] = 0;
      return nullptr;
    }
    g_scratch[slot] = p;
    g_scratch_bytes[slot] = want;
  }
  return g_scratch[slot];
}

static unsigned *g_tickets = nullptr;
static int g_ticket_count = 0;
unsigned *tickets(int count) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (count > g_ticket_count) {
    if (g_tickets) {
      (void)hipStreamSynchronize(cur_stream());
      (void)hipFree(g_tickets);
    }
    const int want = count < 4096 ? 4096 : count * 2;
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned) * want) != hipSuccess || hipMemset(p, 0, sizeof(unsigned) * want) != hipSuccess) {
      set_error("hipMalloc(tickets) failed");
      g_tickets = nullptr;
      g_ticket_count = 0;
      return nullptr;
    }
    g_tickets = static_cast<unsigned *>(p);
    g_ticket_count = want;
  }
  return g_tickets;
}

}  // namespace aslp

extern "C" {

void aslp_set_stream(void *s) { aslp::set_cur_stream(reinterpret_cast<hipStream_t>(s)); }
void *aslp_get_stream(void) { return reinterpret_cast<void *>(aslp::cur_stream()); }

int aslp_get_last_error(char *buf, int buflen) {
  std::lock_guard<std::mutex> lk(aslp::g_err_mu);
  if (aslp::g_err.empty()) {
    if (buf && buflen > 0) buf[0] = 0;
    return 0;
  }
  if (buf && buflen > 0) {
    std::strncpy(buf, aslp::g_err.c_str(), buflen - 1);
    buf[buflen - 1] = 0;
  }
  aslp::g_err.clear();
  return 1;
}

int aslp_device_sync(void) {
  hipError_t e = hipStreamSynchronize(aslp::cur_stream());
  if (e != hipSuccess) {
    aslp::set_error(std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return 1;
  }
  return 0;
}

const char *aslp_version(void) { return "aslp-hip 0.1 (gfx950)"; }

}  // extern "C"
