// coop.h -- what the kernels whose workgroups talk to each other INSIDE a launch share (nn_fused.hip: BatchNormalization panels, plane
// conversion; gemm_split16.hip: the in-diff product that also takes the BatchNormalization's backward pass).
// Everything here needs the talking workgroups resident at once; the launchers check that (coop_grid_wide_ok: a device of the process' own,
// one launching host thread, the main stream, no more workgroups than the chip places at once).  Words cross between workgroups as relaxed
// agent-scope atomics -- no fences: a device-scope release is an L2 write-back (DESIGN 7 "Round 5").
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace aslp {

constexpr unsigned long long kCoopNothing = 0xFFFFFFFFFFFFFFFFull;   // "nothing here" in an inbox word (a NaN pattern no sum takes)
constexpr int kCoopSpinLimit = 1 << 22;  // polls before a reader gives up (seconds): the error word is raised, the output is garbage

#if defined(__HIPCC__)
// one inbox word: poll until it holds something, hand the slot back ("nothing here"); *ok goes false on a time-out
__device__ __forceinline__ unsigned long long coop_take(unsigned long long *slot, bool *ok) {
  unsigned long long v;
  int spins = 0;
  for (;;) {
    v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != kCoopNothing) break;
    if (++spins > kCoopSpinLimit) { *ok = false; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(slot, kCoopNothing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
__device__ __forceinline__ void coop_put(unsigned long long *slot, unsigned long long v) {
  __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The largest of one value per workgroup, in every workgroup (256 threads) of a launch of nwg workgroups; wg: this one's index.  They meet
// in gmax, one 8-byte word each: this launch's token | the value's bits -- nothing to reset, an older launch's word never matches.
// wg_max: the workgroup's value (in every thread); red: >= 4 floats of LDS nobody else is using.
__device__ __forceinline__ float coop_grid_max(float wg_max, unsigned long long *gmax, int wg, int nwg, unsigned token, unsigned *err, float *red) {
  if (threadIdx.x == 0)
    __hip_atomic_store(gmax + wg, ((unsigned long long)token << 32) | (unsigned long long)__float_as_uint(wg_max), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  float gm = 0.f;
  bool ok = true;
  for (int i = threadIdx.x; i < nwg; i += 256) {
    int spins = 0;
    for (;;) {
      const unsigned long long v = __hip_atomic_load(gmax + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(v >> 32) == token) { gm = fmaxf(gm, __uint_as_float((unsigned)v)); break; }
      if (++spins > kCoopSpinLimit) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (!ok) __hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  gm = wave_max(gm);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gm;
  __syncthreads();
  gm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  return gm;
}
#endif

// ---- host side (nn_fused.hip owns the state: per host thread) ----
constexpr int kCoopGmaxWords = 1024;        // workgroups whose maxima may meet in one launch
constexpr int kCoopPanelWords = 256 * 128;  // 8-byte words of each of the two panel inboxes of a fused product (256 workgroups x 128 columns)
struct CoopFusedState {
  unsigned long long *inbox1, *inbox2, *gmax;   // inbox1 / inbox2: kCoopPanelWords words each, all "nothing here" between launches
  unsigned *err;
  unsigned token;                               // a fresh one for this launch
};
// the exchange areas of the calling thread for a launch whose workgroups all wait for each other, with a fresh token; false: not available
// (a shared device, a second launching thread, the side stream: the caller takes its unfused path)
bool coop_fused_state(CoopFusedState *out);
int coop_num_cus();

}  // namespace aslp
