// runtime.cpp -- per-thread stream, sticky error slot and device scratch for libaslp_hip.so.
// Replaces the slice of CuDevice (src/aslp-cudamatrix/cu-device.h:43-151) the hot path needs.
#include <atomic>
#include <map>
#include <mutex>
#include <vector>
#include <string>
#include <cstring>
#include <sys/syscall.h>
#include <unistd.h>

#include "aslp_kernels.h"
#include "common.h"
#include "scratch.h"

namespace aslp {

static thread_local hipStream_t t_stream = nullptr;
static std::mutex g_err_mu;
static std::string g_err;

hipStream_t cur_stream() { return t_stream; }
void set_cur_stream(hipStream_t s) { t_stream = s; }

// Error words that a running kernel raises on its own (mapped host memory): looked at whenever the error slot is read.
namespace {
struct AsyncErr { const volatile unsigned *word; unsigned seen; std::string what; };
std::vector<AsyncErr> g_async_errs;
void poll_async_errors_locked() {
  for (auto &a : g_async_errs) {
    const unsigned v = *a.word;
    if (v != a.seen) { a.seen = v; if (g_err.empty()) g_err = a.what; }
  }
}
}  // namespace
void register_async_error_word(const volatile unsigned *host_word, const char *what) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_async_errs.push_back({host_word, *host_word, what});
}

// A fresh mapped host word a kernel can bump (system-scope atomic) to report a failure of its own; returns the device pointer.
unsigned *new_async_error_word(const char *what) {
  unsigned *host = nullptr, *dev = nullptr;
  if (hipHostMalloc(&host, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
  *host = 0;
  if (hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), host, 0) != hipSuccess) return nullptr;
  register_async_error_word(host, what);
  return dev;
}

void set_error(const std::string &msg) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  if (g_err.empty()) g_err = msg;  // keep the first one: it is the cause
}
bool has_error() {
  std::lock_guard<std::mutex> lk(g_err_mu);
  poll_async_errors_locked();
  return !g_err.empty();
}

// One grow-only scratch arena per (host thread, stream bank, slot).  Kernels on one stream run in order, so a slot can
// be reused by the next op on the same stream without synchronisation (the reference's caching allocator exists for
// the same reason: cu-allocator.h:67-70).  The arenas are thread_local like the stream they serve: two host threads
// (ThreadComm ranks, a multi-threaded API user) launch on different streams and must never hand the same partial-sum
// buffer to their column reductions / split-K GEMMs / CTC lattices.
static thread_local int t_bank = 0;  // 0 = main stream, 1 = side stream
namespace {
struct ScratchArenas {
  void *blk[2][kNumScratch] = {{nullptr}};
  size_t cap[2][kNumScratch] = {{0}};
  ~ScratchArenas() {
    // worker threads give their blocks back; the process' main thread (tid == pid) leaves them to process teardown,
    // which may already have unloaded the HIP runtime when thread_local destructors run
    if ((long)syscall(SYS_gettid) == (long)getpid()) return;
    for (auto &bank : blk)
      for (void *&p : bank)
        if (p) { (void)hipFree(p); p = nullptr; }
  }
};
}  // namespace
static thread_local ScratchArenas t_arenas;

void *scratch(int slot, size_t bytes) {
  void *&blk = t_arenas.blk[t_bank][slot];
  size_t &cap = t_arenas.cap[t_bank][slot];
  if (bytes > cap) {
    if (blk) {
      // outstanding kernels of this thread's stream may still use the old block
      (void)hipStreamSynchronize(cur_stream());
      (void)hipFree(blk);
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes + bytes / 2;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      set_error(std::string("hipMalloc(scratch): ") + hipGetErrorString(e));
      blk = nullptr;
      cap = 0;
      return nullptr;
    }
    blk = p;
    cap = want;
  }
  return blk;
}

// ---- side stream: fork / join by events, no host synchronisation --------------------------------------------
static thread_local hipStream_t t_side = nullptr;
static thread_local hipEvent_t t_fork_ev = nullptr, t_join_ev = nullptr;
static thread_local hipStream_t t_side_parent = nullptr;  // main stream the pending side work has to rejoin
static thread_local bool t_side_dirty = false;
static thread_local unsigned long t_side_seq = 0;          // side-stream scopes opened by this thread so far
static thread_local unsigned long long t_mark_gen = 0;     // generation of the latest marker this thread recorded, and the scope count it covers
static thread_local unsigned long t_mark_seq = 0;

static bool side_ready() {
  if (t_side) return true;
  // lowest priority: when both streams have workgroups to place, the main stream's (the critical path) go first
  int prio_low = 0, prio_high = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
  if (hipStreamCreateWithPriority(&t_side, hipStreamNonBlocking, prio_low) != hipSuccess ||
      hipEventCreateWithFlags(&t_fork_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&t_join_ev, hipEventDisableTiming) != hipSuccess) {
    set_error("side stream: cannot create stream / events");
    t_side = nullptr;
    return false;
  }
  return true;
}

SideStreamScope::SideStreamScope() : saved_(nullptr), active_(false) {
  if (t_bank != 0 || !side_ready()) return;  // nested scope or no stream: stay where we are
  hipStream_t main = cur_stream();
  if (t_side_dirty && t_side_parent != main) join_side_stream();  // pending work belongs to another main stream
  if (hipEventRecord(t_fork_ev, main) != hipSuccess || hipStreamWaitEvent(t_side, t_fork_ev, 0) != hipSuccess) {
    set_error("side stream: fork failed");
    return;
  }
  saved_ = main;
  t_side_parent = main;
  set_cur_stream(t_side);
  t_bank = 1;
  t_side_dirty = true;
  t_side_seq++;
  active_ = true;
}
SideStreamScope::~SideStreamScope() {
  if (!active_) return;
  set_cur_stream(static_cast<hipStream_t>(saved_));
  t_bank = 0;
}
bool on_side_stream() { return t_bank == 1; }
void join_side_stream() {
  if (!t_side_dirty) return;
  t_side_dirty = false;
  if (hipEventRecord(t_join_ev, t_side) != hipSuccess || hipStreamWaitEvent(t_side_parent, t_join_ev, 0) != hipSuccess)
    set_error("side stream: join failed");
}

// A marker is an event plus a generation number drawn from a process-wide counter at every record: "is this my own latest marker" is
// decided by the number, never by the address (an address can come back from the allocator after a net has been destroyed, and another
// thread may have recorded into it since).
namespace {
struct SideMark { hipEvent_t ev = nullptr; unsigned long long gen = 0; };
std::atomic<unsigned long long> g_mark_gen{0};
}  // namespace
int side_stream_mark(void **ev) {
  if (!t_side_dirty || !t_side || !ev) return 0;
  SideMark *m = static_cast<SideMark *>(*ev);
  if (!m) {
    m = new SideMark();
    if (hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) != hipSuccess) { delete m; set_error("side stream: cannot create a marker event"); return -1; }
    *ev = m;
  }
  if (hipEventRecord(m->ev, t_side) != hipSuccess) { set_error("side stream: marker record failed"); return -1; }
  m->gen = ++g_mark_gen;
  t_mark_gen = m->gen;
  t_mark_seq = t_side_seq;
  return 1;
}
void side_stream_mark_wait(void *ev, bool host) {
  SideMark *m = static_cast<SideMark *>(ev);
  if (!m || !m->ev) return;
  const hipError_t e = host ? hipEventSynchronize(m->ev) : hipStreamWaitEvent(cur_stream(), m->ev, 0);
  if (e != hipSuccess) { set_error(std::string("side stream: marker wait failed: ") + hipGetErrorString(e)); return; }
  // This thread's own latest marker with no side-stream scope opened since: the marker stands behind ALL pending side work, and the wait
  // just made IS the join with the parent stream -- join_side_stream() need not put a second event wait into that stream (a wait on another
  // stream's event costs the waiting stream several microseconds of idle time even when the event has long fired).
  if (m->gen != 0 && m->gen == t_mark_gen && t_mark_seq == t_side_seq && t_bank == 0 && (host || cur_stream() == t_side_parent)) t_side_dirty = false;
}
void side_stream_mark_free(void *ev) {
  SideMark *m = static_cast<SideMark *>(ev);
  if (!m) return;
  if (m->ev) (void)hipEventDestroy(m->ev);
  delete m;
}

// ---- host threads that launch kernels whose workgroups wait for ALL workgroups of their launch (scratch.h) --------------------------
namespace {
std::atomic<int> g_grid_wide_threads{0};
struct GridWideRegistrar {
  bool on = false;
  ~GridWideRegistrar() { if (on) g_grid_wide_threads.fetch_sub(1); }
};
}  // namespace
void register_grid_wide_thread() {
  static thread_local GridWideRegistrar r;
  if (!r.on) { r.on = true; g_grid_wide_threads.fetch_add(1); }
}
int grid_wide_threads() { return g_grid_wide_threads.load(); }

// ---- named region timers (bench.py's cfg3 block): HIP events on the launch stream around a host-side region ------------
namespace {
struct Region {
  double ms = 0.0;
  long count = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
std::mutex g_region_mu;
std::map<std::string, Region> g_regions;
std::vector<hipEvent_t> g_region_pool;
bool g_region_on = false;
hipEvent_t region_event() {
  if (!g_region_pool.empty()) { hipEvent_t e = g_region_pool.back(); g_region_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
void region_drain(Region &r) {
  for (auto &ev : r.pending) {
    float ms = 0.f;
    if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) r.ms += ms;
    g_region_pool.push_back(ev.first);
    g_region_pool.push_back(ev.second);
  }
  r.pending.clear();
}
}  // namespace

RegionScope::RegionScope(const char *name) : name_(nullptr), e0_(nullptr) {
  if (!g_region_on) return;
  std::lock_guard<std::mutex> lk(g_region_mu);
  name_ = name;
  e0_ = region_event();
  (void)hipEventRecord(static_cast<hipEvent_t>(e0_), cur_stream());
}
RegionScope::~RegionScope() {
  if (!name_) return;
  std::lock_guard<std::mutex> lk(g_region_mu);
  hipEvent_t e1 = region_event();
  (void)hipEventRecord(e1, cur_stream());
  Region &r = g_regions[name_];
  r.count++;
  r.pending.emplace_back(static_cast<hipEvent_t>(e0_), e1);
}

}  // namespace aslp

extern "C" {

void aslp_region_profile(int enable) {
  std::lock_guard<std::mutex> lk(aslp::g_region_mu);
  aslp::g_region_on = enable != 0;
}
void aslp_region_reset(void) {
  std::lock_guard<std::mutex> lk(aslp::g_region_mu);
  for (auto &kv : aslp::g_regions) aslp::region_drain(kv.second);
  aslp::g_regions.clear();
}
long aslp_region_get(const char *name, double *ms) {
  std::lock_guard<std::mutex> lk(aslp::g_region_mu);
  auto it = aslp::g_regions.find(name ? name : "");
  if (it == aslp::g_regions.end()) { if (ms) *ms = 0.0; return 0; }
  aslp::region_drain(it->second);
  if (ms) *ms = it->second.ms;
  return it->second.count;
}

void aslp_set_stream(void *s) { aslp::set_cur_stream(reinterpret_cast<hipStream_t>(s)); }
void *aslp_get_stream(void) { return reinterpret_cast<void *>(aslp::cur_stream()); }

int aslp_get_last_error(char *buf, int buflen) {
  std::lock_guard<std::mutex> lk(aslp::g_err_mu);
  aslp::poll_async_errors_locked();
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
