// scratch.h -- grow-only device scratch slots and the side stream (see runtime.cpp).
#pragma once
#include <stddef.h>
namespace aslp {
enum { kScratchReduce = 0, kScratchReduce2 = 1, kScratchGemm = 2, kScratchCtc = 3, kScratchMisc = 4, kScratchSplitK = 5, kScratchSplit16 = 6, kNumScratch = 7 };
// Scratch of the calling thread's current stream: the side stream has its own bank, so ops running there never
// share a partial-sum buffer with ops on the main stream.
void *scratch(int slot, size_t bytes);

// While alive, every launch of this thread goes to the library's side stream, ordered after everything issued on the
// main stream so far (event wait, no host sync).  join_side_stream() makes the main stream wait for the side work
// issued since the last join; it is a no-op when there is none.
class SideStreamScope {
 public:
  SideStreamScope();
  ~SideStreamScope();
  SideStreamScope(const SideStreamScope &) = delete;
  SideStreamScope &operator=(const SideStreamScope &) = delete;
 private:
  void *saved_;
  bool active_;
};
// Times the region between construction and destruction with HIP events on the calling thread's stream when region profiling
// is on (aslp_region_profile(1)); free otherwise.  name must be a string literal.  Read with aslp_region_get(name, &ms).
class RegionScope {
 public:
  explicit RegionScope(const char *name);
  ~RegionScope();
  RegionScope(const RegionScope &) = delete;
  RegionScope &operator=(const RegionScope &) = delete;
 private:
  const char *name_;
  void *e0_;
};
void join_side_stream();
// A marker for the side-stream work the calling thread has issued so far, for waiters that may sit on another host thread (whose own
// join_side_stream() knows nothing of this thread's side stream).  side_stream_mark(ev) records into *ev (created on first use; the
// caller owns it and gives it back with side_stream_mark_free).  Returns 1 = recorded, 0 = there is no pending side work, -1 = the marker
// could not be created or recorded (error set): the caller must then join_side_stream() itself -- "failed" is not "nothing pending".
// side_stream_mark_wait(ev, host): the calling thread's current stream waits for the marker -- or the host does.
int side_stream_mark(void **ev);
void side_stream_mark_wait(void *ev, bool host);
void side_stream_mark_free(void *ev);
// Several processes share this GPU (ASLP_DEVICE_SHARED=1 / aslp_device_shared(1); rnn_persistent.hip): kernels whose workgroups wait for
// ALL workgroups of their launch (resident at once on a device of their own) must not be used -- two such launches half resident beside
// each other never finish.
bool device_shared();
bool on_side_stream();  // is the calling thread inside a SideStreamScope?
// The same hazard between host threads of ONE process: every thread that launches such kernels -- the cooperative BatchNormalization /
// planes launches of nn_fused.hip AND the persistent LSTM / GRU recurrences of rnn_persistent.hip -- registers itself once; with more than
// one registered thread the launches that could stand down to a multi-launch path do (coop_grid_wide_ok), the persistent recurrences
// are serialised against each other by their launch chain.
void register_grid_wide_thread();
int grid_wide_threads();
}  // namespace aslp
