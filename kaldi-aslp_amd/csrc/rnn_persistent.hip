// rnn_persistent.hip -- the LSTM recurrence of a whole sequence as ONE launch per layer and pass (both directions at once).
//
// rnn_fused.hip runs one launch per timestep: at S = 32 streams each of those sits on the launch floor (8 us forward,
// 6 + 4 us backward at C = 512) with half the chip idle.  Here the workgroups stay resident for all T timesteps.
//
// The recurrence decomposes into independent CHAINS: one per (direction, group of 8 streams) -- streams never interact and
// the two directions of a BLSTM never do either; at S = 32, bidirectional, that is 8 chains.  A chain is served by C / 16
// workgroups, each owning 16 cells:
//   forward   keeps its 4 x 16 rows of W_eff (gate x cell, K = C) in REGISTERS as MFMA B fragments for the whole launch; per
//             timestep reads m(t-1) of the chain's 8 streams (the only quantity that crosses workgroups), multiplies
//             (v_mfma_f32_16x16x4_f32, one 16 x 16 tile per gate, K split over the 4 waves, partial tiles summed through LDS in
//             wave order), finishes the gate block for its 8 x 16 (stream, cell) pairs and publishes m(t).  c(t-1) of a pair
//             never leaves its thread's registers.
//   backward  keeps its 16 rows of W_eff^T (K = G*C, split over 8 waves) in registers, per timestep reads dGATES(next) of
//             the chain's 8 streams, multiplies, finishes the gate-block backward for its 8 x 16 pairs and publishes its
//             16 x G gate diffs.  The own-cell terms of the next step (d_c, d_i, d_f) are carried in registers.
// Workgroup b serves chain b & 7.  On this chip consecutive workgroups go round-robin to the 8 XCDs, so a chain's C / 16 = 32
// workgroups (at C = 512) are the 32 CUs of ONE XCD and the hand-off of m(t) stays inside that XCD's L2 (measured: a
// hand-off through the fabric costs ~2 us each way under load, 6.7 us per timestep; inside one L2 a fraction of that).
// Nothing depends on that placement for correctness: at kernel start every workgroup publishes the XCC id it runs on, the
// chain's workgroups read all of them, and only if they agree does the chain use plain (L2-resident) stores; otherwise it
// stores write-through at agent scope (sc1), which is correct for any placement, just slower.
//
// Hand-off between workgroups: the data is its own flag.  The host fills the row blocks 1..T of the activation / diff
// buffer with the byte 0xFF before the launch (aslp_lstm_seq_fill), i.e. every float is the bit pattern 0xFFFFFFFF, a NaN no
// arithmetic here produces.  Producers store 16-byte pieces (plain when the chain shares an L2, else write-through at agent
// scope: buffer_store_dwordx4 sc1, the line leaves the XCD's L2); consumers load the pieces they need with agent-scope loads
// (buffer_load_dwordx4 sc1: never served from the CU's L1) and reload until no lane holds the sentinel.  No flag, no fence,
// no barrier across workgroups; every row block is written exactly once per launch, so there is no reuse hazard.  (MI355X_MICROARCH.md, inter-workgroup visibility: "R2 -- the data is the flag"; sc1 stores and
// sc1 loads on both sides replace the release / acquire pair.)
// Every spin is bounded (wall clock): on a timeout the workgroup raises a device-wide abort word, all workgroups leave, and
// the host reports the failure through aslp_get_last_error -- a wrong result is never silent and the GPU never hangs.
// The grid must be co-resident: aslp_lstm_seq_supported() checks it against the occupancy of the kernel with a margin
// of one workgroup per CU, otherwise the caller keeps the one-launch-per-timestep path.
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <cerrno>
#include <mutex>
#include <thread>
#include <string>
#include <utility>
#include <vector>

#include "aslp_kernels.h"
#include "common.h"
#include "scratch.h"
#include "split16.h"   // SeqFillJob / seq_fill_row: the buffer preparation a conversion launch can take along

namespace aslp {
void register_async_error_word(const volatile unsigned *host_word, const char *what);  // runtime.cpp
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kSentinel = 0xFFFFFFFFu;
constexpr int kAuxSc1 = 16;          // buffer instruction cache policy: sc1 = agent scope
constexpr long kSpinLimitTicks = 200000000L;  // wall_clock64 runs at 100 MHz: 2 s

__device__ __forceinline__ float dsigm(float y, float d) { return d * y * (1.0f - y); }
__device__ __forceinline__ float dtanh(float y, float d) { return d * (1.0f - y * y); }

// Every base pointer handed to a buffer instruction here is wave-uniform by construction (kernel arguments, blockIdx, the loop
// counter), but hipcc cannot always prove it (the direction's pointers are picked from the argument struct with an index that
// went through shared memory) and then wraps EVERY buffer load / store in a waterfall loop over the lanes' descriptor values.
// readfirstlane makes the uniformity explicit: the descriptor lives in SGPRs and the access is one instruction.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  float *u = reinterpret_cast<float *>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(u, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ bool has_sentinel(const u32x4 &v) {
  return v.x == kSentinel || v.y == kSentinel || v.z == kSentinel || v.w == kSentinel;
}
__device__ __forceinline__ float as_f(unsigned u) { return __uint_as_float(u); }
// cross-lane moves on the DPP path of the VALU (no LDS crossbar round trip like ds_bpermute): lane K of the caller's quad, and the
// lane N places up within the caller's row of 16 lanes
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), K * 0x55, 0xF, 0xF, true));   // quad_perm:[K,K,K,K]
}
template <int N>
__device__ __forceinline__ float row_up(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x100 + N, 0xF, 0xF, true));  // row_shl:N -> dst[i] = src[i + N]
}

template <int N>
__device__ __forceinline__ float row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x120 + N, 0xF, 0xF, true));  // row_ror:N -> rotation within the row of 16 lanes
}

// device-side status: abort_flag[0] abort flag (zeroed before every launch), abort_flag[2] running count of hand-off
// re-polls (diagnostics, aslp_lstm_seq_polls); host_err: mapped host word, counts timeouts
struct SeqStatus {
  unsigned *abort_flag;
  unsigned *host_err;
  unsigned long long *timing;  // diagnostics (devtools): NULL, or 8 accumulators of 10 ns ticks written by workgroup 0, wave 0
  unsigned long long *trace;   // diagnostics (devtools): NULL, or [workgroup][2] entry / exit clock of the latest launch
  unsigned epoch;              // launch counter (28 bits, never 0): tags the placement table entries of this launch
  unsigned wave_collect;       // LSTM forward.  bit 0: every wave collects the K slice of m(t-1) it multiplies itself (no workgroup barrier behind the
                               // collection); bit 1: operand reads pinned four fragments ahead of the products
};
__device__ __forceinline__ long tick(const SeqStatus &st) { return st.timing ? (long)wall_clock64() : 0; }
// The phase accumulators live in LDS while the kernel runs (a fire-and-forget ds_add per mark): accumulating in global memory put an L2
// round trip and a wait behind every mark -- 0.1-0.2 us charged to the NEXT phase, five times per timestep, and workgroup 0 (hence the
// whole lock-stepped chain) ran that much slower under the timer.  timing_flush adds them to st.timing once, at the end.
__shared__ unsigned long long g_tacc[8];
__device__ __forceinline__ void timing_begin(const SeqStatus &st) {
  if (st.timing && threadIdx.x < 8) g_tacc[threadIdx.x] = 0ull;   // (chain_role's barrier publishes it)
}
__device__ __forceinline__ void timing_flush(const SeqStatus &st) {   // caller: st.timing != NULL, workgroup 0, thread 0, behind the loop's last barrier
  for (int k = 1; k <= 5; k++) st.timing[k] += g_tacc[k];
}
__device__ __forceinline__ void tock(const SeqStatus &st, int slot, long &t) {
  if (!st.timing) return;
  const long now = (long)wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(&g_tacc[slot], (unsigned long long)(now - t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  t = now;
}

// the same for the first thread of the gate role (thread 256) of the wave-specialised kernel
__device__ __forceinline__ void tock_gate(const SeqStatus &st, int slot, long &t) {
  if (!st.timing) return;
  const long now = (long)wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 256) __hip_atomic_fetch_add(&g_tacc[slot], (unsigned long long)(now - t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  t = now;
}

// Bounded spin bookkeeping shared by the two phases below: false = give up (device-wide abort or 2 s without progress).
__device__ __forceinline__ bool spin_ok(unsigned spins, long &t0, const SeqStatus &st) {
  if ((spins & 31u) != 31u) return true;
  if (__hip_atomic_load(st.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
  const long now = (long)wall_clock64();
  if (t0 == 0) { t0 = now; return true; }
  if (now - t0 <= kSpinLimitTicks) return true;
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_store(st.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(st.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return false;
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
// C/D layout of v_mfma_f32_16x16x4_f32: element e of lane l is row 4 * (l >> 4) + e, column l & 15
constexpr int kTP = 17;  // LDS pitch of a 16 x 16 partial tile
__device__ __forceinline__ void store_tile16(float *tile, const f32x4 &acc, int lane) {
  const int n = lane & 15, r0 = 4 * (lane >> 4);
  tile[(r0 + 0) * kTP + n] = acc.x;
  tile[(r0 + 1) * kTP + n] = acc.y;
  tile[(r0 + 2) * kTP + n] = acc.z;
  tile[(r0 + 3) * kTP + n] = acc.w;
}

// ---- chain geometry shared by both kernels ------------------------------------------------------------------------
constexpr int kFirstK = 256;       // largest K of a first-step product served inside the forward launch (aslp_lstm_seq_dir.w_first)
constexpr int kChainStreams = 8;   // streams per chain (rows 0..7 of the 16-row MFMA tile; rows 8..15 repeat them, outputs unused)
constexpr int kCellsPerWg = 16;
constexpr int kMaxChains = 8;      // = XCDs of the chip: workgroup b serves chain b & 7
constexpr int kMaxWgPerChain = 32; // = CUs of one XCD (C <= 512)

struct ChainRole {
  int dir, s0, c0;   // direction, first stream, first cell
  bool active;       // this workgroup has a chain to serve
  bool local;        // the chain's workgroups share one XCD (one L2): plain stores suffice
};

// Who am I, and does my chain sit on one XCD?  place: [kMaxChains][kMaxWgPerChain] words; an entry counts once it carries
// this launch's epoch (a host-side launch counter, st.epoch) -- nothing to clear between launches.
__device__ __forceinline__ ChainRole chain_role(int S, int ndir, int C, const SeqStatus &st, unsigned *place, int *lds_flag) {
  ChainRole r;
  const int chain = blockIdx.x & (kMaxChains - 1), cb = blockIdx.x >> 3;
  const int nsg = (S + kChainStreams - 1) / kChainStreams, nchains = ndir * nsg, wpc = (C + kCellsPerWg - 1) / kCellsPerWg;
  r.active = chain < nchains;
  r.dir = r.active ? chain % ndir : 0;
  r.s0 = (r.active ? chain / ndir : 0) * kChainStreams;
  r.c0 = cb * kCellsPerWg;
  r.local = false;
  if (!r.active) return r;
  if (threadIdx.x < 64) {  // wave 0
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 15u) | (st.epoch << 4);  // entries of earlier launches carry another epoch = "not yet written"
    unsigned *row = place + chain * kMaxWgPerChain;
    if (threadIdx.x == 0) __hip_atomic_store(row + cb, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int l = threadIdx.x;
    unsigned v = xcc;
    long t0 = 0;
    bool ok = true;
    for (unsigned spins = 0;; spins++) {
      if (l < wpc) v = __hip_atomic_load(row + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!__any(l < wpc && (v >> 4) != st.epoch)) break;
      if (!spin_ok(spins, t0, st)) { ok = false; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    const bool same = __all(l >= wpc || v == xcc);
    if (threadIdx.x == 0) *lds_flag = !ok ? -1 : (same ? 1 : 0);
  }
  __syncthreads();
  const int f = *lds_flag;
  if (f < 0) r.active = false;  // timed out waiting for the chain to show up: abort word is set, leave
  r.local = f == 1;
  return r;
}

// The gate non-linearities on the hardware's exp2 and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each) instead of the correctly rounded
// expf and division of sigmoid_ref / tanh_ref: ~6 instructions on the sequential path of a timestep instead of ~60, results within a
// few ulp (1e-6 relative after T = 60 steps; the parity bar is 1e-4).  Default; ASLP_LSTM_FAST_ACT=0 keeps the exact forms (A/B).
template <bool FAST>
__device__ __forceinline__ float act_sigmoid(float x) {
  if (!FAST) return sigmoid_ref(x);
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * fabsf(x));
  return (x > 0.0f ? 1.0f : e) * __builtin_amdgcn_rcpf(1.0f + e);
}
template <bool FAST>
__device__ __forceinline__ float act_tanh(float x) {
  if (!FAST) return tanh_ref(x);
  const float e2 = __builtin_amdgcn_exp2f(-2.88539008177792681472f * fabsf(x));   // exp(-2 |x|)
  const float q = 2.0f * __builtin_amdgcn_rcpf(1.0f + e2);
  return x > 0.0f ? -1.0f + q : 1.0f - q;
}

// ---- forward -------------------------------------------------------------------------------------------------------
// grid 8 * ceil(C / 16) workgroups of 512 threads.  KW: K values per wave (C <= 8 * KW).
// The product runs on v_mfma_f32_4x4x1_16b_f32 (see the backward kernel for the block layout): one instruction covers the
// chain's 8 streams x 32 of the workgroup's 64 gate columns with no padding rows, the 8 waves split K.  The left operand
// m(t-1) [8 x C] is collected ONCE per workgroup into LDS (every thread two 16-byte pieces, agent-scope loads, repeated until
// no piece reads "not yet published") and read from there by all lanes -- the 4 x 4 blocks would otherwise pull every
// piece eight times through the L2.
template <bool CIFG, int KW, bool FAST>
__global__ void __launch_bounds__(512) lstm_seq_fwd(aslp_lstm_seq a, SeqStatus st, unsigned *place) {
  constexpr int G = CIFG ? 3 : 4, KMAX = 8 * KW, MP = KMAX + 4, RP = 80;  // RP = 16 mod 32: the epilogue's reads hit 32 distinct banks
  __shared__ __attribute__((aligned(16))) float m_lds[kChainStreams][MP];
  __shared__ __attribute__((aligned(16))) float wf_lds[64][kFirstK + 4];   // this workgroup's 64 rows of W_first (first step only, see w_first)
  __shared__ float red[2][8][kChainStreams][RP];
  __shared__ int fail[2][8];
  __shared__ int place_flag;
  __shared__ __attribute__((aligned(16))) float zero_lds[4];
  const long t_entry = (st.timing || st.trace) ? (long)wall_clock64() : 0;
  timing_begin(st);
  const int SE = a.s_count > 0 ? a.s_begin + a.s_count : a.S;   // this launch's streams: [a.s_begin, SE)
  if (threadIdx.x < 4) zero_lds[threadIdx.x] = 0.f;   // (chain_role's barrier publishes it)
  const ChainRole R = chain_role(SE - a.s_begin, a.ndir, a.C, st, place, &place_flag);
  if (!R.active) return;
  const aslp_lstm_seq_dir D = a.dir[R.dir];
  const int C = a.C, S = a.S, T = a.T, ld = a.ld;
  const int GC = G * C, oc = GC, oh = GC + C, om = GC + 2 * C;
  const int c0 = R.c0, s0 = a.s_begin + R.s0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qs = (lane >> 2) & 1, qc = lane >> 3, jl = lane & 3;
  const int kw = ((C + 31) / 32) * 4, kb = wave * kw;  // this wave's K range [kb, kb + kw), a multiple of 4 long
  // B fragments, resident for the launch: instruction h covers tile columns 32 h + 4 qc + jl; column n = gate n >> 4, cell c0 + (n & 15)
  f32x4 bw[2][KW / 4];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int n = 32 * h + 4 * qc + jl, gate = n >> 4, cellb = c0 + (n & 15);
    const bool nvalid = gate < G && cellb < C;
    const float *brow = D.w + (long)(nvalid ? gate * C + cellb : 0) * a.ldw;
#pragma unroll
    for (int i = 0; i < KW / 4; i++) {
      const int k0 = kb + 4 * i;
      bw[h][i] = (nvalid && 4 * i < kw && k0 < C) ? *reinterpret_cast<const f32x4 *>(brow + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // W_first rows of this workgroup's 64 gate columns -> LDS (zero where the column or k does not exist); read at step 0 only
  const bool first_in_kernel = D.w_first != nullptr && D.k_first > 0 && D.k_first <= kFirstK && D.k_first <= KMAX;   // r(0) is staged in m_lds rows of KMAX floats
  if (first_in_kernel) {
    const int kq = (D.k_first + 3) >> 2;   // 16-byte pieces per row
    for (int p = threadIdx.x; p < 64 * kq; p += 512) {
      const int n = p / kq, k0 = 4 * (p % kq), gate = n >> 4, cellb = c0 + (n & 15);
      const bool ok = gate < G && cellb < C;
      *reinterpret_cast<f32x4 *>(&wf_lds[n][k0]) = ok ? *reinterpret_cast<const f32x4 *>(D.w_first + (long)(gate * C + cellb) * D.ldw_first + k0)
                                                      : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // collection role.  Workgroup-wide: pieces tid and tid + 512 of [stream][C / 4], a barrier, then every wave reads its K slice.
  // Wave-local (st.wave_collect): a wave fetches exactly what it multiplies -- the 8 streams' pieces of its own K slice [kb, kb + kw),
  // 2 kw <= 128 pieces, lanes 0..63 take pieces lane and lane + 64 of [stream][kw / 4] -- so no other wave's data is involved, the
  // workgroup barrier behind the collection goes, and a wave waits for the 4 producers of its slice instead of all 32.
  const bool wave_collect = (st.wave_collect & 1u) != 0u;
  const int c4 = C >> 2, npiece = kChainStreams * c4, kq_w = kw >> 2, npw = kChainStreams * kq_w;
  const int p0 = wave_collect ? lane : threadIdx.x, p1 = wave_collect ? lane + 64 : threadIdx.x + 512;
  int st0, kq0, st1, kq1;
  bool h0, h1;
  if (wave_collect) {
    st0 = p0 / kq_w; kq0 = (kb >> 2) + p0 % kq_w; st1 = p1 / kq_w; kq1 = (kb >> 2) + p1 % kq_w;
    h0 = p0 < npw && kq0 < c4; h1 = p1 < npw && kq1 < c4;
    if (!h0) { st0 = 0; kq0 = 0; }
    if (!h1) { st1 = 0; kq1 = 0; }
  } else {
    h0 = p0 < npiece; h1 = p1 < npiece;
    st0 = h0 ? p0 / c4 : 0; kq0 = h0 ? p0 % c4 : 0; st1 = h1 ? p1 / c4 : 0; kq1 = h1 ? p1 % c4 : 0;
  }
  const int off0 = (min(s0 + st0, SE - 1) * ld + om + 4 * kq0) * 4, off1 = (min(s0 + st1, SE - 1) * ld + om + 4 * kq1) * 4;
  // gate-block role: a quad of lanes owns one (stream, cell) pair, lane r of the quad its gate r (g, i, f, o; CIFG: g, f, o, -).
  // The five transcendentals of a pair then take two rounds (gates side by side, then tanh(c) beside the output gate) instead of
  // five in a row on one lane, and all 8 waves share the work.
  const int pair = threadIdx.x >> 2, role = threadIdx.x & 3;
  const int sl = pair >> 4, cc = pair & 15, s = s0 + sl, cell = c0 + cc;
  const bool live = s < SE && cell < C;
  const int cq = live ? cell : 0, sq = live ? s : 0;
  // the peephole weight of this lane's gate (none for g): i <- c(t-1), f <- c(t-1), o <- c(t)
  float pw = 0.f;
  if (!CIFG) pw = role == 1 ? D.peep_i[cq] : role == 2 ? D.peep_f[cq] : role == 3 ? D.peep_o[cq] : 0.f;
  else pw = role == 1 ? D.peep_f[cq] : role == 2 ? D.peep_o[cq] : 0.f;
  const int slen = (D.seq_lengths && live) ? D.seq_lengths[sq] : 0x7fffffff;
  float cprev = 0.f;
  {
    const int tp0 = D.reverse ? T + 1 : 0;
    if (live) cprev = D.y[((long)tp0 * S + sq) * ld + oc + cq];
  }
  unsigned polls = 0u;
  for (int step = 0; step < T; step++) {
    const int t = D.reverse ? T - step : 1 + step, tp = D.reverse ? t + 1 : t - 1;
    const int par = step & 1;
    float *ys = D.y + ((long)t * S + sq) * ld;
    long tm = tick(st);
    // the x-part (+ bias) of this pair's gates: written before the launch, requested before the hand-off wait
    const float xr = (live && role < G) ? ys[role * C + cq] : 0.f;
    f32x4 acc[2][2];
#pragma unroll
    for (int h = 0; h < 2; h++) { acc[h][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[h][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    bool ok = true;
    const bool first_special = step == 0 && first_in_kernel;
    const bool product = !first_special && !(step == 0 && D.skip_first_product);
    if (first_special) {   // r(0) of the chain's streams -> LDS (the history row block: stored before the launch, no hand-off)
      const int kq = D.k_first >> 2;
      for (int p = threadIdx.x; p < kChainStreams * kq; p += 512) {
        const int sp = p / kq, k0 = 4 * (p % kq);
        *reinterpret_cast<f32x4 *>(&m_lds[sp][k0]) =
            *reinterpret_cast<const f32x4 *>(D.y + ((long)tp * S + min(s0 + sp, SE - 1)) * ld + D.col_first + k0);
      }
    }
    if (product) {
      // 1. m(t-1) of the chain's streams -> LDS
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(D.y + (long)tp * S * ld);
      u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
      long t0 = 0;
      for (unsigned spins = 0;; spins++) {
        if (h0) v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off0, 0, kAuxSc1);
        if (h1) v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off1, 0, kAuxSc1);
        if (!__any((h0 && has_sentinel(v0)) || (h1 && has_sentinel(v1)))) break;
        asm volatile("" ::: "memory");
        polls++;
        if (!spin_ok(spins, t0, st)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (h0) *reinterpret_cast<u32x4 *>(&m_lds[st0][4 * kq0]) = v0;
      if (h1) *reinterpret_cast<u32x4 *>(&m_lds[st1][4 * kq1]) = v1;
    }
    tock(st, 1, tm);  // collection
    if (lane == 0) fail[par][wave] = ok ? 0 : 1;
    if (!wave_collect || first_special) __syncthreads();   // (uniform) wave-local collection: this wave reads back only what it stored itself
    else __builtin_amdgcn_wave_barrier();
    tock(st, 3, tm);  // barrier behind the collection
    if (product) {
      // 2. this wave's K slice of the product
      // all operand reads first, unconditionally (slices past C read element 0 of the row: finite, and their B fragments are 0),
      // so that the LDS latency is paid once and not in front of every group of products
      // The operand reads run PD fragments ahead of the products that use them (sched_barrier pins the order: left alone, the
      // scheduler re-uses one register quad for every read and waits for each of the 16 reads two products after issuing it --
      // the LDS latency 16 times per timestep, with all eight waves of the workgroup in step).
      const float *arow = &m_lds[4 * qs + jl][0];
      constexpr int NF = KW / 4, PD = NF < 4 ? NF : 4;
      auto load_a = [&](int i) -> f32x4 {   // fragments past the wave's slice / past C read four zeros kept in LDS (nobody else's data, no select on the loaded value)
        const bool in_slice = 4 * i < kw && kb + 4 * i < C;
        return *reinterpret_cast<const f32x4 *>(in_slice ? arow + kb + 4 * i : zero_lds);
      };
      f32x4 av[NF];
      if ((st.wave_collect & 2u) == 0u) {   // A/B: the scheduler's own order (ASLP_LSTM_READ_AHEAD=0)
#pragma unroll
        for (int i = 0; i < NF; i++) av[i] = load_a(i);
#pragma unroll
        for (int i = 0; i < NF; i++) {
#pragma unroll
          for (int h = 0; h < 2; h++) acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].x, bw[h][i].x, acc[h][0], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < 2; h++) acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].y, bw[h][i].y, acc[h][1], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < 2; h++) acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].z, bw[h][i].z, acc[h][0], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < 2; h++) acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].w, bw[h][i].w, acc[h][1], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int i = 0; i < PD; i++) av[i] = load_a(i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NF; i++) {
        if (i + PD < NF) av[i + PD] = load_a(i + PD);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; h++) acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].x, bw[h][i].x, acc[h][0], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; h++) acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].y, bw[h][i].y, acc[h][1], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; h++) acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].z, bw[h][i].z, acc[h][0], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; h++) acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].w, bw[h][i].w, acc[h][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    if (first_special) {   // r(0) W_first^T, both operands from LDS; this wave's slice of K = k_first
      const int kwf = ((D.k_first + 31) / 32) * 4, kbf = wave * kwf;
      const float *arow = &m_lds[4 * qs + jl][0];
      for (int k0 = kbf; k0 < min(kbf + kwf, D.k_first); k0 += 4) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(arow + k0);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const f32x4 b = *reinterpret_cast<const f32x4 *>(&wf_lds[32 * h + 4 * qc + jl][k0]);
          acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, b.x, acc[h][0], 0, 0, 0);
          acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, b.y, acc[h][1], 0, 0, 0);
          acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, b.z, acc[h][0], 0, 0, 0);
          acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, b.w, acc[h][1], 0, 0, 0);
        }
      }
    }
    // result register r of a lane = stream 4 qs + r of tile column 32 h + 4 qc + jl
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const f32x4 sum = acc[h][0] + acc[h][1];
      float *rp = &red[par][wave][4 * qs][32 * h + 4 * qc + jl];
      rp[0 * RP] = sum.x; rp[1 * RP] = sum.y; rp[2 * RP] = sum.z; rp[3 * RP] = sum.w;
    }
    tock(st, 2, tm);  // product
    __syncthreads();
    tock(st, 4, tm);
    {
      int f = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) f |= fail[par][w];
      if (f) return;  // uniform: every wave reads the same eight words
    }
    float pre = 0.f;
    if (role < G) {
      pre = red[par][0][sl][role * 16 + cc];
#pragma unroll
      for (int w = 1; w < 8; w++) pre += red[par][w][sl][role * 16 + cc];
    }
    const bool masked = t > slen;  // nnet-blstm-projected-streams.h:654-657: rows past the utterance end are zeroed
    // round 1: g = tanh(.), i / f = sigmoid(. + c(t-1) * peephole), each on its own lane
    float gate = 0.f;
    if (role == 0) gate = act_tanh<FAST>(xr + pre);
    else if (role < G - 1) gate = act_sigmoid<FAST>(xr + pre + cprev * pw);
    const float gg = quad_bcast<0>(gate), g1 = quad_bcast<1>(gate), g2 = quad_bcast<2>(gate);
    float cellv;
    if (!CIFG) cellv = gg * g1 + cprev * g2;        // g * i + c(t-1) * f
    else cellv = -gg * g1 + gg + cprev * g1;        // coupled input gate: i = 1 - f
    cellv = fminf(fmaxf(cellv, -50.0f), 50.0f);
    // round 2: h = tanh(c) on lane 0, o = sigmoid(. + c * peephole) on lane G - 1
    float hh = 0.f;
    if (role == 0) hh = act_tanh<FAST>(cellv);
    if (role == G - 1) gate = act_sigmoid<FAST>(xr + pre + cellv * pw);
    const float oo = quad_bcast<G - 1>(gate);
    float mm = hh * oo;   // meaningful on lane 0 of the quad
    if (masked) { gate = 0.f; cellv = 0.f; hh = 0.f; mm = 0.f; }
    // publish m(t) first: it is what the other workgroups wait for.  Four consecutive cells = lane 0 of four consecutive quads.
    {
      const float m1 = row_up<4>(mm), m2 = row_up<8>(mm), m3 = row_up<12>(mm);
      if (live && role == 0 && (cc & 3) == 0) {
        u32x4 pk = {__float_as_uint(mm), __float_as_uint(m1), __float_as_uint(m2), __float_as_uint(m3)};
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(D.y + (long)t * S * ld);
        const int off = (s * ld + om + cell) * 4;
        if (R.local) __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, kAuxSc1);
      }
    }
    if (live) {
      if (role < G) ys[role * C + cell] = gate;
      if (role == 0) ys[oh + cell] = hh;
      if (role == 1) ys[oc + cell] = cellv;
    }
    cprev = cellv;
    tock(st, 5, tm);  // epilogue
  }
  if (st.timing && blockIdx.x == 0 && threadIdx.x == 0) {
    timing_flush(st); st.timing[0] += (unsigned long long)T; st.timing[6] += R.local ? 1ull : 0ull;
    st.timing[7] += (unsigned long long)((long)wall_clock64() - t_entry);   // this workgroup's whole stay, entry to exit
  }
  if (st.trace && threadIdx.x == 0) {   // ring of the 8 latest launches
    unsigned long long *tr = st.trace + (st.epoch & 7u) * 2048u;
    tr[2 * blockIdx.x] = (unsigned long long)t_entry; tr[2 * blockIdx.x + 1] = wall_clock64();
  }
  if (polls && lane == 0) __hip_atomic_fetch_add(st.abort_flag + 2, polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // diagnostics, once per wave
}

// ---- forward, product on fp16 matrix instructions with fp32-equivalent operands ---------------------------------------------------
// Same chains, hand-off, gate block and buffers as lstm_seq_fwd; only the recurrent product m(t-1) W^T differs.  The fp32 instruction
// (v_mfma_f32_4x4x1: 256 FLOP / clk / CU) makes that product the longest phase of a timestep (8 streams x 64 gate columns x K = 512 per
// workgroup = 2048 clk, 0.85 us of ~2.35); v_mfma_f32_16x16x32_f16 runs 16 x that rate.  Each fp32 operand is therefore carried as TWO
// fp16 pieces (x = x_hi + 2^-11 x_lo', 22-23 significant bits; weights after a per-column power-of-two scale, so any magnitude is safe):
// the A tile's rows 0..7 are m_hi of the chain's 8 streams and rows 8..15 m_lo' of the same streams -- the instruction's 16 rows are all
// used -- and two instructions per chunk (B = w_hi, w_lo') form all four partial products, every one exact in the multiplier and
// accumulated in fp32.  Measured against a double product the result is CLOSER than an fp32 fma chain (devtools/micro/f16_split.hip:
// 3-5e-8 of sum |m w| against 0.9-2e-7).  Wave w multiplies its own K slice (as collected) for all gate tiles; the two row halves are
// joined with v_permlane32_swap and the 8 waves' partial sums go through the same LDS reduction as before.
// Step 0 with a W_first operand keeps the fp32 instructions (once per launch, operands from LDS).  A/B switch: ASLP_LSTM_SPLIT_F16.
template <bool CIFG, int NCH, bool FAST>
__global__ void __launch_bounds__(512) lstm_seq_fwd_h(aslp_lstm_seq a, SeqStatus st, unsigned *place) {
  constexpr int KW = 32 * NCH;   // K values per wave: NCH chunks of the instruction's 32
  constexpr int G = CIFG ? 3 : 4, KMAX = 8 * KW, MP = KMAX + 4, RP = 80;  // RP = 16 mod 32: the epilogue's reads hit 32 distinct banks
  constexpr int AP = KW + 8;     // halves per operand row: 16-byte reads of 16 rows x 4 k-groups then spread over the banks
  __shared__ __attribute__((aligned(16))) _Float16 a_h[8][16][AP];   // per wave: rows 0..7 = m_hi of stream r, rows 8..15 = m_lo' of stream r - 8, its K slice
  __shared__ __attribute__((aligned(16))) float m_lds[kChainStreams][MP];
  __shared__ __attribute__((aligned(16))) float wf_lds[64][kFirstK + 4];   // this workgroup's 64 rows of W_first (first step only, see w_first)
  __shared__ float red[2][8][kChainStreams][RP];
  __shared__ int fail[2][8];
  __shared__ int place_flag;
  __shared__ __attribute__((aligned(16))) float zero_lds[4];
  const long t_entry = (st.timing || st.trace) ? (long)wall_clock64() : 0;
  __builtin_amdgcn_s_setprio(3);   // a latency chain: when weight-gradient workgroups share the CU (side stream), this kernel's waves issue first
  timing_begin(st);
  const int SE = a.s_count > 0 ? a.s_begin + a.s_count : a.S;   // this launch's streams: [a.s_begin, SE)
  if (threadIdx.x < 4) zero_lds[threadIdx.x] = 0.f;   // (chain_role's barrier publishes it)
  const ChainRole R = chain_role(SE - a.s_begin, a.ndir, a.C, st, place, &place_flag);
  if (!R.active) return;
  const aslp_lstm_seq_dir D = a.dir[R.dir];
  const int C = a.C, S = a.S, T = a.T, ld = a.ld;
  const int GC = G * C, oc = GC, oh = GC + C, om = GC + 2 * C;
  const int c0 = R.c0, s0 = a.s_begin + R.s0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qs = (lane >> 2) & 1, qc = lane >> 3, jl = lane & 3;
  const int kw = ((C + 31) / 32) * 4, kb = wave * kw;  // this wave's K range [kb, kb + kw), a multiple of 4 long
  // B fragments, resident for the launch: tile = gate, column l & 15 = cell c0 + (l & 15); lane l holds k = kb + 32 j + 8 (l >> 4) + 0..7 of
  // chunk j.  Each weight is carried as two fp16 pieces of w * sc (sc = the column's power of two that puts max |w| in [2^13, 2^14)):
  // w_hi = fp16(w sc), w_lo = fp16(w sc - w_hi) -- 22 significant bits for every weight down to 2^-17 of its column's largest, no
  // overflow whatever the weights' size; both pieces accumulate into ONE result tile (A w_hi + A w_lo = A w).
  const int hr = lane & 15, hg = lane >> 4;
  half8 bh[4][NCH], bl[4][NCH];
  float inv_sc[4];
  {
    float wv[4][NCH][8];
    float *cm = &red[0][0][0][0];   // scratch before the first timestep: [wave][tile][lane]
#pragma unroll
    for (int tile = 0; tile < 4; tile++) {
      const int cellb = c0 + hr;
      const bool nvalid = tile < G && cellb < C;
      const float *brow = D.w + (long)(nvalid ? tile * C + cellb : 0) * a.ldw;
      float lmax = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int kl = 32 * j + 8 * hg, k0 = kb + kl;   // kw and C are multiples of 4: a piece of four is inside or outside as a whole
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const bool in = nvalid && kl + 4 * q < kw && k0 + 4 * q < C;
          const f32x4 w4 = in ? *reinterpret_cast<const f32x4 *>(brow + k0 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
          wv[tile][j][4 * q + 0] = w4.x; wv[tile][j][4 * q + 1] = w4.y; wv[tile][j][4 * q + 2] = w4.z; wv[tile][j][4 * q + 3] = w4.w;
          lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(w4.x), fabsf(w4.y)), fmaxf(fabsf(w4.z), fabsf(w4.w))));
        }
      }
      cm[(wave * 4 + tile) * 64 + lane] = lmax;
    }
    __syncthreads();
#pragma unroll
    for (int tile = 0; tile < 4; tile++) {
      float cmax = 0.f;
      for (int w = 0; w < 8; w++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) cmax = fmaxf(cmax, cm[(w * 4 + tile) * 64 + 16 * g4 + hr]);
      int e = 0;
      (void)frexpf(cmax, &e);   // cmax = f 2^e, f in [0.5, 1)
      const bool scaled = cmax > 0.f && cmax < 3.0e38f;
      const float sc = scaled ? ldexpf(1.f, 14 - e) : 1.f;
      inv_sc[tile] = scaled ? ldexpf(1.f, e - 14) : 1.f;
#pragma unroll
      for (int j = 0; j < NCH; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const float x = wv[tile][j][i] * sc;
          const _Float16 xh = (_Float16)x;
          bh[tile][j][i] = xh;
          bl[tile][j][i] = (_Float16)(x - (float)xh);   // the residual as it is (<= 8 for the column's largest weight; a subnormal one is < 2^-39 of it)
        }
    }
    __syncthreads();   // the scratch is the reduction buffer of the timesteps
  }
  // W_first rows of this workgroup's 64 gate columns -> LDS (zero where the column or k does not exist); read at step 0 only
  const bool first_in_kernel = D.w_first != nullptr && D.k_first > 0 && D.k_first <= kFirstK && D.k_first <= KMAX;   // r(0) is staged in m_lds rows of KMAX floats
  if (first_in_kernel) {
    const int kq = (D.k_first + 3) >> 2;   // 16-byte pieces per row
    for (int p = threadIdx.x; p < 64 * kq; p += 512) {
      const int n = p / kq, k0 = 4 * (p % kq), gate = n >> 4, cellb = c0 + (n & 15);
      const bool ok = gate < G && cellb < C;
      *reinterpret_cast<f32x4 *>(&wf_lds[n][k0]) = ok ? *reinterpret_cast<const f32x4 *>(D.w_first + (long)(gate * C + cellb) * D.ldw_first + k0)
                                                      : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // collection role.  Workgroup-wide: pieces tid and tid + 512 of [stream][C / 4], a barrier, then every wave reads its K slice.
  // Wave-local (st.wave_collect): a wave fetches exactly what it multiplies -- the 8 streams' pieces of its own K slice [kb, kb + kw),
  // 2 kw <= 128 pieces, lanes 0..63 take pieces lane and lane + 64 of [stream][kw / 4] -- so no other wave's data is involved, the
  // workgroup barrier behind the collection goes, and a wave waits for the 4 producers of its slice instead of all 32.
  const bool wave_collect = (st.wave_collect & 1u) != 0u;
  const int c4 = C >> 2, npiece = kChainStreams * c4, kq_w = kw >> 2, npw = kChainStreams * kq_w;
  const int p0 = wave_collect ? lane : threadIdx.x, p1 = wave_collect ? lane + 64 : threadIdx.x + 512;
  int st0, kq0, st1, kq1;
  bool h0, h1;
  if (wave_collect) {
    st0 = p0 / kq_w; kq0 = (kb >> 2) + p0 % kq_w; st1 = p1 / kq_w; kq1 = (kb >> 2) + p1 % kq_w;
    h0 = p0 < npw && kq0 < c4; h1 = p1 < npw && kq1 < c4;
    if (!h0) { st0 = 0; kq0 = 0; }
    if (!h1) { st1 = 0; kq1 = 0; }
  } else {
    h0 = p0 < npiece; h1 = p1 < npiece;
    st0 = h0 ? p0 / c4 : 0; kq0 = h0 ? p0 % c4 : 0; st1 = h1 ? p1 / c4 : 0; kq1 = h1 ? p1 % c4 : 0;
  }
  const int off0 = (min(s0 + st0, SE - 1) * ld + om + 4 * kq0) * 4, off1 = (min(s0 + st1, SE - 1) * ld + om + 4 * kq1) * 4;
  // gate-block role: a quad of lanes owns one (stream, cell) pair, lane r of the quad its gate r (g, i, f, o; CIFG: g, f, o, -).
  // The five transcendentals of a pair then take two rounds (gates side by side, then tanh(c) beside the output gate) instead of
  // five in a row on one lane, and all 8 waves share the work.
  const int pair = threadIdx.x >> 2, role = threadIdx.x & 3;
  const int sl = pair >> 4, cc = pair & 15, s = s0 + sl, cell = c0 + cc;
  const bool live = s < SE && cell < C;
  const int cq = live ? cell : 0, sq = live ? s : 0;
  // the peephole weight of this lane's gate (none for g): i <- c(t-1), f <- c(t-1), o <- c(t)
  float pw = 0.f;
  if (!CIFG) pw = role == 1 ? D.peep_i[cq] : role == 2 ? D.peep_f[cq] : role == 3 ? D.peep_o[cq] : 0.f;
  else pw = role == 1 ? D.peep_f[cq] : role == 2 ? D.peep_o[cq] : 0.f;
  const int slen = (D.seq_lengths && live) ? D.seq_lengths[sq] : 0x7fffffff;
  float cprev = 0.f;
  {
    const int tp0 = D.reverse ? T + 1 : 0;
    if (live) cprev = D.y[((long)tp0 * S + sq) * ld + oc + cq];
  }
  for (int p = lane; p < 16 * AP / 8; p += 64) reinterpret_cast<u32x4 *>(&a_h[wave][0][0])[p] = u32x4{0u, 0u, 0u, 0u};   // own wave's rows; read back by this wave only
  unsigned polls = 0u;
  for (int step = 0; step < T; step++) {
    const int t = D.reverse ? T - step : 1 + step, tp = D.reverse ? t + 1 : t - 1;
    const int par = step & 1;
    float *ys = D.y + ((long)t * S + sq) * ld;
    long tm = tick(st);
    // the x-part (+ bias) of this pair's gates: written before the launch, requested before the hand-off wait
    const float xr = (live && role < G) ? ys[role * C + cq] : 0.f;
    f32x4 acc[2][2];
#pragma unroll
    for (int h = 0; h < 2; h++) { acc[h][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[h][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    bool ok = true;
    const bool first_special = step == 0 && first_in_kernel;
    const bool product = !first_special && !(step == 0 && D.skip_first_product);
    if (first_special) {   // r(0) of the chain's streams -> LDS (the history row block: stored before the launch, no hand-off)
      const int kq = D.k_first >> 2;
      for (int p = threadIdx.x; p < kChainStreams * kq; p += 512) {
        const int sp = p / kq, k0 = 4 * (p % kq);
        *reinterpret_cast<f32x4 *>(&m_lds[sp][k0]) =
            *reinterpret_cast<const f32x4 *>(D.y + ((long)tp * S + min(s0 + sp, SE - 1)) * ld + D.col_first + k0);
      }
    }
    if (product) {
      // 1. m(t-1) of the chain's streams -> LDS
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(D.y + (long)tp * S * ld);
      u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
      long t0 = 0;
      for (unsigned spins = 0;; spins++) {
        if (h0) v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off0, 0, kAuxSc1);
        if (h1) v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off1, 0, kAuxSc1);
        if (!__any((h0 && has_sentinel(v0)) || (h1 && has_sentinel(v1)))) break;
        asm volatile("" ::: "memory");
        polls++;
        if (!spin_ok(spins, t0, st)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (wave_collect) {   // m = m_hi + 2^-11 m_lo' (|m| <= 1: no scale needed), into this wave's operand rows
        auto put = [&](const u32x4 v, int stv, int kqv) {
          const int kl = 4 * kqv - kb;
          _Float16 hi[4], lo[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float x = __uint_as_float(v[i]);
            hi[i] = (_Float16)x;
            lo[i] = (_Float16)((x - (float)hi[i]) * 2048.f);
          }
          *reinterpret_cast<half4 *>(&a_h[wave][stv][kl]) = half4{hi[0], hi[1], hi[2], hi[3]};
          *reinterpret_cast<half4 *>(&a_h[wave][8 + stv][kl]) = half4{lo[0], lo[1], lo[2], lo[3]};
        };
        if (h0) put(v0, st0, kq0);
        if (h1) put(v1, st1, kq1);
      } else {
        if (h0) *reinterpret_cast<u32x4 *>(&m_lds[st0][4 * kq0]) = v0;
        if (h1) *reinterpret_cast<u32x4 *>(&m_lds[st1][4 * kq1]) = v1;
      }
    }
    tock(st, 1, tm);  // collection
    if (lane == 0) fail[par][wave] = ok ? 0 : 1;
    if (!wave_collect || first_special) __syncthreads();   // (uniform) wave-local collection: this wave reads back only what it stored itself
    else __builtin_amdgcn_wave_barrier();
    tock(st, 3, tm);  // barrier behind the collection
    f32x4 hacc[4];
    if (product) {
      // 2. this wave's K slice of the product: NCH chunks x G gate tiles x {w_hi, w_lo'} instructions of 16 x 16 x 32
      if (!wave_collect) {   // (A/B path: the workgroup-wide collection left fp32 rows in m_lds; split this wave's slice from there)
        for (int p = lane; p < kChainStreams * (kw >> 2); p += 64) {
          const int stv = p / (kw >> 2), kl = 4 * (p % (kw >> 2));
          if (kb + kl < C) {
            const f32x4 x4 = *reinterpret_cast<const f32x4 *>(&m_lds[stv][kb + kl]);
            _Float16 hi[4], lo[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { hi[i] = (_Float16)x4[i]; lo[i] = (_Float16)((x4[i] - (float)hi[i]) * 2048.f); }
            *reinterpret_cast<half4 *>(&a_h[wave][stv][kl]) = half4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<half4 *>(&a_h[wave][8 + stv][kl]) = half4{lo[0], lo[1], lo[2], lo[3]};
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      half8 af[NCH];
#pragma unroll
      for (int j = 0; j < NCH; j++) af[j] = *reinterpret_cast<const half8 *>(&a_h[wave][hr][32 * j + 8 * hg]);
#pragma unroll
      for (int tile = 0; tile < 4; tile++) hacc[tile] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NCH; j++) {
#pragma unroll
        for (int tile = 0; tile < G; tile++) hacc[tile] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], bh[tile][j], hacc[tile], 0, 0, 0);
#pragma unroll
        for (int tile = 0; tile < G; tile++) hacc[tile] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], bl[tile][j], hacc[tile], 0, 0, 0);
      }
    }
    if (first_special) {   // r(0) W_first^T, both operands from LDS; this wave's slice of K = k_first
      const int kwf = ((D.k_first + 31) / 32) * 4, kbf = wave * kwf;
      const float *arow = &m_lds[4 * qs + jl][0];
      for (int k0 = kbf; k0 < min(kbf + kwf, D.k_first); k0 += 4) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(arow + k0);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const f32x4 b = *reinterpret_cast<const f32x4 *>(&wf_lds[32 * h + 4 * qc + jl][k0]);
          acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, b.x, acc[h][0], 0, 0, 0);
          acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, b.y, acc[h][1], 0, 0, 0);
          acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.z, b.z, acc[h][0], 0, 0, 0);
          acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.w, b.w, acc[h][1], 0, 0, 0);
        }
      }
    }
    if (product) {
      // rows 0..7 (lanes 0..31) hold m_hi w, rows 8..15 (lanes 32..63) m_lo' w of the same streams: m w = ([m_hi w] + 2^-11 [m_lo' w]) / sc.
      // One v_permlane32_swap joins the halves of TWO result registers at once: of (x, y) it leaves x's two halves in lanes 0..31 of the
      // pair and y's two halves in lanes 32..63, so the sum is register e's total in the lower lanes and register e + 1's in the upper.
      const int hg2 = hg & 1, eodd = lane >> 5;
#pragma unroll
      for (int tile = 0; tile < G; tile++) {
        const float f = lane < 32 ? inv_sc[tile] : inv_sc[tile] * 0x1p-11f;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(hacc[tile][e] * f), __float_as_uint(hacc[tile][e + 1] * f), false, false);
          red[par][wave][4 * hg2 + e + eodd][16 * tile + hr] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);   // stream = result row 4 (l >> 4 & 1) + e (+ 1 in the upper lanes)
        }
      }
    } else {
      // result register r of a lane = stream 4 qs + r of tile column 32 h + 4 qc + jl
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const f32x4 sum = acc[h][0] + acc[h][1];
        float *rp = &red[par][wave][4 * qs][32 * h + 4 * qc + jl];
        rp[0 * RP] = sum.x; rp[1 * RP] = sum.y; rp[2 * RP] = sum.z; rp[3 * RP] = sum.w;
      }
    }
    tock(st, 2, tm);  // product
    __syncthreads();
    tock(st, 4, tm);
    {
      int f = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) f |= fail[par][w];
      if (f) return;  // uniform: every wave reads the same eight words
    }
    float pre = 0.f;
    if (role < G) {
      pre = red[par][0][sl][role * 16 + cc];
#pragma unroll
      for (int w = 1; w < 8; w++) pre += red[par][w][sl][role * 16 + cc];
    }
    const bool masked = t > slen;  // nnet-blstm-projected-streams.h:654-657: rows past the utterance end are zeroed
    // round 1: g = tanh(.), i / f = sigmoid(. + c(t-1) * peephole), each on its own lane
    float gate = 0.f;
    if (role == 0) gate = act_tanh<FAST>(xr + pre);
    else if (role < G - 1) gate = act_sigmoid<FAST>(xr + pre + cprev * pw);
    const float gg = quad_bcast<0>(gate), g1 = quad_bcast<1>(gate), g2 = quad_bcast<2>(gate);
    float cellv;
    if (!CIFG) cellv = gg * g1 + cprev * g2;        // g * i + c(t-1) * f
    else cellv = -gg * g1 + gg + cprev * g1;        // coupled input gate: i = 1 - f
    cellv = fminf(fmaxf(cellv, -50.0f), 50.0f);
    // round 2: h = tanh(c) on lane 0, o = sigmoid(. + c * peephole) on lane G - 1
    float hh = 0.f;
    if (role == 0) hh = act_tanh<FAST>(cellv);
    if (role == G - 1) gate = act_sigmoid<FAST>(xr + pre + cellv * pw);
    const float oo = quad_bcast<G - 1>(gate);
    float mm = hh * oo;   // meaningful on lane 0 of the quad
    if (masked) { gate = 0.f; cellv = 0.f; hh = 0.f; mm = 0.f; }
    // publish m(t) first: it is what the other workgroups wait for.  Four consecutive cells = lane 0 of four consecutive quads.
    {
      const float m1 = row_up<4>(mm), m2 = row_up<8>(mm), m3 = row_up<12>(mm);
      if (live && role == 0 && (cc & 3) == 0) {
        u32x4 pk = {__float_as_uint(mm), __float_as_uint(m1), __float_as_uint(m2), __float_as_uint(m3)};
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(D.y + (long)t * S * ld);
        const int off = (s * ld + om + cell) * 4;
        if (R.local) __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, kAuxSc1);
      }
    }
    if (live) {
      if (role < G) ys[role * C + cell] = gate;
      if (role == 0) ys[oh + cell] = hh;
      if (role == 1) ys[oc + cell] = cellv;
    }
    cprev = cellv;
    tock(st, 5, tm);  // epilogue
  }
  if (st.timing && blockIdx.x == 0 && threadIdx.x == 0) {
    timing_flush(st); st.timing[0] += (unsigned long long)T; st.timing[6] += R.local ? 1ull : 0ull;
    st.timing[7] += (unsigned long long)((long)wall_clock64() - t_entry);   // this workgroup's whole stay, entry to exit
  }
  if (st.trace && threadIdx.x == 0) {   // ring of the 8 latest launches
    unsigned long long *tr = st.trace + (st.epoch & 7u) * 2048u;
    tr[2 * blockIdx.x] = (unsigned long long)t_entry; tr[2 * blockIdx.x + 1] = wall_clock64();
  }
  if (polls && lane == 0) __hip_atomic_fetch_add(st.abort_flag + 2, polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // diagnostics, once per wave
}

// ---- backward ------------------------------------------------------------------------------------------------------
// d_m(t) = dm_ext(t) + dGATES(next) W_eff is a product over K = G*C gate columns.  Partitioned by OUTPUT cell (like the forward
// pass) every workgroup would have to pull all of dGATES(next) of its chain -- 64 KiB per workgroup and timestep through the
// CU's 64 B/clk path, measured 2.5 us of a 7 us timestep.  It is partitioned by INPUT instead: a workgroup multiplies the
// 16 x G gate diffs it produced itself at the previous step (they never leave the CU: LDS) with its 16 x G ROWS of W_eff
// (registers), which yields its share of d_m for ALL cells of the chain; the shares travel -- 512 B to each of the chain's
// workgroups -- and every workgroup adds up the C / 16 shares of its own 16 cells in workgroup order (deterministic).
// Traffic per workgroup and timestep: 16 KiB out, 16 KiB in.
//   inbox[ring slot][chain][consumer block][producer block][tile row group 0..1][column 0..15] of f32x4 (4 tile rows):
//   exactly the accumulator layout of v_mfma_f32_16x16x4_f32, so a producer's lanes store their accumulators as they are.
// The inbox is a ring (kRing slots, slot = timestep % kRing): a consumer writes the sentinel back over what it has read and
// drains those stores (vmcnt(0)) before it goes on -- its next publication, and through it every later write to that
// slot by anybody, is therefore ordered after the reset.
// grid 8 * ceil(C / 16) workgroups of 512 threads.  TPW: N tiles per wave (C <= 128 * TPW).
constexpr int kRing = 4;
template <bool CIFG, int TPW>
__global__ void __launch_bounds__(512) lstm_seq_bwd(aslp_lstm_seq a, SeqStatus st, unsigned *place, float *inbox) {
  constexpr int G = CIFG ? 3 : 4, KS = 4 * G;  // KS: MFMA k-steps over the workgroup's own 16 * G gate columns
  __shared__ __attribute__((aligned(16))) float own_dg[2][kChainStreams][16 * G + 4];   // [parity][stream][gate * 16 + cell]
  __shared__ __attribute__((aligned(16))) float shares[kMaxWgPerChain * 2 * 16 * 4];     // [producer][row group][column][row]
  __shared__ int fail[2][8];
  __shared__ int place_flag;
  const long t_entry = st.trace ? (long)wall_clock64() : 0;
  timing_begin(st);
  const int SE = a.s_count > 0 ? a.s_begin + a.s_count : a.S;
  const ChainRole R = chain_role(SE - a.s_begin, a.ndir, a.C, st, place, &place_flag);
  if (!R.active) return;
  const aslp_lstm_seq_dir D = a.dir[R.dir];
  const int C = a.C, S = a.S, T = a.T, ld = a.ld;
  const int GC = G * C, oc = GC, oh = GC + C, om = GC + 2 * C;
  const int og = 0, oi = C, of = CIFG ? C : 2 * C, oo = CIFG ? 2 * C : 3 * C;
  const int c0 = R.c0, s0 = a.s_begin + R.s0;
  const int chain = blockIdx.x & (kMaxChains - 1), me = blockIdx.x >> 3, wpc = (C + kCellsPerWg - 1) / kCellsPerWg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // The product runs on v_mfma_f32_4x4x1_16b_f32: 16 independent 4 x 4 blocks per instruction, here 2 stream quads x 8 cell
  // quads = the chain's 8 streams x 32 cells with NO padding rows (the 16 x 16 tile carries 8 idle rows: twice the MFMA time).
  // Lane l: block l >> 2 = (stream quad sq = block & 1, cell quad cq = block >> 1); A operand = own gate diff of stream
  // 4 sq + (l & 3), B operand = W_eff[own gate column k][cell], result register r = stream 4 sq + r of that cell.
  // This wave's cells: consumer blocks wave + 8 j, two blocks (32 cells) per instruction h.
  constexpr int NH = (TPW + 1) / 2;
  const int qs = (lane >> 2) & 1, qc = lane >> 3, jl = lane & 3;
  float bw[NH][KS * 4];
#pragma unroll
  for (int h = 0; h < NH; h++) {
    const int cb = wave + 8 * (2 * h + (qc >> 2));
    const int col = cb * 16 + 4 * (qc & 3) + jl;
    const bool colok = (2 * h + (qc >> 2)) < TPW && cb < wpc && col < C;
#pragma unroll
    for (int kk = 0; kk < KS * 4; kk++) {
      const int cellk = c0 + (kk & 15);
      bw[h][kk] = (colok && cellk < C) ? D.w[(long)((kk >> 4) * C + cellk) * a.ldw + col] : 0.f;
    }
  }
  // inbox geometry
  const size_t slot_words = (size_t)kMaxChains * kMaxWgPerChain * kMaxWgPerChain * 128;  // floats per ring slot
  float *chain_box = inbox + (size_t)chain * kMaxWgPerChain * kMaxWgPerChain * 128;
  const int npiece = wpc * 32;  // 16-byte pieces addressed to this workgroup per timestep: [producer][row group][column]
  // epilogue role: threads 0..127 own one (stream, cell) pair each
  const int sl = threadIdx.x >> 4, cc = threadIdx.x & 15, s = s0 + sl, cell = c0 + cc;
  const bool live = threadIdx.x < 128 && s < SE && cell < C;
  const int cq = live ? cell : 0, sq = live ? s : 0;
  const float pf = D.peep_f[cq], po = D.peep_o[cq], pi = CIFG ? 0.f : D.peep_i[cq];
  // own-cell quantities of the step processed just before (BPTT order): all zero ahead of the first step
  float dn_c = 0.f, dn_f = 0.f, dn_i = 0.f;
  float gsum[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  unsigned polls = 0u;
  for (int step = 0; step < T; step++) {
    // BPTT runs against the direction's recursion: reverse = 0 (t = T..1), reverse = 1 (t = 1..T)
    const int t = D.reverse ? 1 + step : T - step;
    const int tn = D.reverse ? t - 1 : t + 1, tp = D.reverse ? t + 1 : t - 1;
    const int par = step & 1;
    const long o_ = ((long)t * S + sq) * ld;
    long tm = tick(st);
    float dm = 0.f, yo = 0.f, yh = 0.f, yg = 0.f, yf = 0.f, yi = 0.f, yn_f = 0.f, cprev = 0.f, ccur = 0.f;
    if (live) {  // everything that does not depend on the other workgroups, requested first
      dm = D.d[o_ + om + cq];
      yo = D.y[o_ + oo + cq]; yh = D.y[o_ + oh + cq]; yg = D.y[o_ + og + cq]; yf = D.y[o_ + of + cq];
      if (!CIFG) yi = D.y[o_ + oi + cq];
      yn_f = D.y[((long)tn * S + sq) * ld + of + cq];
      cprev = D.y[((long)tp * S + sq) * ld + oc + cq];
      if (a.grad_partial) ccur = D.y[o_ + oc + cq];
    }
    bool ok = true;
    if (step > 0) {
      float *box = chain_box + (size_t)(step % kRing) * slot_words;
      // 1. my share of d_m for every cell of the chain: own gate diffs of the previous step (LDS) x my rows of W_eff
      {
        f32x4 av[KS];
        const float *arow = &own_dg[par ^ 1][4 * qs + jl][0];
        constexpr int PD = KS < 4 ? KS : 4;   // operand reads pinned PD fragments ahead of their products (see lstm_seq_fwd)
#pragma unroll
        for (int q = 0; q < PD; q++) av[q] = *reinterpret_cast<const f32x4 *>(arow + 4 * q);
        f32x4 acc[NH][2];  // two accumulators per instruction stream (even / odd k): four independent chains keep the pipe full
#pragma unroll
        for (int h = 0; h < NH; h++) { acc[h][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[h][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < KS; q++) {
          if (q + PD < KS) av[q + PD] = *reinterpret_cast<const f32x4 *>(arow + 4 * (q + PD));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int h = 0; h < NH; h++) acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q].x, bw[h][4 * q + 0], acc[h][0], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < NH; h++) acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q].y, bw[h][4 * q + 1], acc[h][1], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < NH; h++) acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q].z, bw[h][4 * q + 2], acc[h][0], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < NH; h++) acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q].w, bw[h][4 * q + 3], acc[h][1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        // 2. hand the shares out: one 16-byte piece (4 streams of one cell) per lane and instruction
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(box);
#pragma unroll
        for (int h = 0; h < NH; h++) {
          const int tj = 2 * h + (qc >> 2), cb = wave + 8 * tj;
          if (tj < TPW && cb < wpc) {
            const f32x4 sum = acc[h][0] + acc[h][1];
            const int off = (((cb * kMaxWgPerChain + me) * 2 + qs) * 16 + 4 * (qc & 3) + jl) * 16;
            const u32x4 pk = {__float_as_uint(sum.x), __float_as_uint(sum.y), __float_as_uint(sum.z), __float_as_uint(sum.w)};
            if (R.local) __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off, 0, kAuxSc1);
          }
        }
      }
      tock(st, 1, tm);  // product + publication
      // 3. collect what the chain's workgroups sent me: pieces tid and tid + 512 of [producer][row group][column]
      {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(box + (size_t)me * kMaxWgPerChain * 128);
        const int i0 = threadIdx.x, i1 = threadIdx.x + 512;
        const bool h0 = i0 < npiece, h1 = i1 < npiece;
        u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
        long t0 = 0;
        for (unsigned spins = 0;; spins++) {
          if (h0) v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, i0 * 16, 0, kAuxSc1);
          if (h1) v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, i1 * 16, 0, kAuxSc1);
          if (!__any((h0 && has_sentinel(v0)) || (h1 && has_sentinel(v1)))) break;
          asm volatile("" ::: "memory");
          polls++;
          if (!spin_ok(spins, t0, st)) { ok = false; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        // give the slot back (sentinel) and make sure that has happened before anything of mine is published again
        const u32x4 sent = {kSentinel, kSentinel, kSentinel, kSentinel};
        if (R.local) {
          if (h0) __builtin_amdgcn_raw_buffer_store_b128(sent, rs, i0 * 16, 0, 0);
          if (h1) __builtin_amdgcn_raw_buffer_store_b128(sent, rs, i1 * 16, 0, 0);
        } else {
          if (h0) __builtin_amdgcn_raw_buffer_store_b128(sent, rs, i0 * 16, 0, kAuxSc1);
          if (h1) __builtin_amdgcn_raw_buffer_store_b128(sent, rs, i1 * 16, 0, kAuxSc1);
        }
        if (h0) *reinterpret_cast<u32x4 *>(&shares[i0 * 4]) = v0;
        if (h1) *reinterpret_cast<u32x4 *>(&shares[i1 * 4]) = v1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      tock(st, 2, tm);  // collection
    }
    if (lane == 0) fail[par][wave] = ok ? 0 : 1;
    __syncthreads();
    tock(st, 4, tm);
    {
      int f = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) f |= fail[par][w];
      if (f) return;
    }
    if (threadIdx.x < 128) {
      if (step > 0) {  // shares of my 16 cells, added in workgroup order
        float psum = 0.f;
        const int base = ((sl >> 2) * 16 + cc) * 4 + (sl & 3);
        for (int p = 0; p < wpc; p++) psum += shares[p * 128 + base];
        dm += psum;
      }
      const float dh = dtanh(yh, dm * yo);
      const float dov = dsigm(yo, dm * yh);
      float dc = dh + dn_c * yn_f;
      if (!CIFG) dc += dn_i * pi;
      dc += dn_f * pf;
      dc += dov * po;
      float dg, df, di = 0.f;
      if (!CIFG) {
        df = dsigm(yf, dc * cprev);
        di = dsigm(yi, dc * yg);
        dg = dtanh(yg, dc * yi);
      } else {
        df = dsigm(yf, dc * cprev - dc * yg);
        dg = dtanh(yg, dc - dc * yf);
      }
      // the next step's left operand stays here; the buffer copy is for the batched products after the launch
      float *mine = &own_dg[par][sl][cc];
      mine[0] = dg;
      if (!CIFG) { mine[16] = di; mine[32] = df; mine[48] = dov; }
      else { mine[16] = df; mine[32] = dov; }
      if (live) {
        D.d[o_ + og + cell] = dg; D.d[o_ + of + cell] = df; D.d[o_ + oo + cell] = dov;
        if (!CIFG) D.d[o_ + oi + cell] = di;
        D.d[o_ + om + cell] = dm;  // the reference's d_m (lc.h:793) -- kept for InfoGradient-style dumps
        D.d[o_ + oh + cell] = dh;
        D.d[o_ + oc + cell] = dc;
      }
      dn_c = dc; dn_f = df; dn_i = di;
      if (a.grad_partial && live) {   // what the bias and peephole gradients are sums of (lc.h:1005-1058), this pair's share
        gsum[0] += dg; gsum[2] += df; gsum[3] += dov;
        gsum[5] += df * cprev; gsum[6] += dov * ccur;
        if (!CIFG) { gsum[1] += di; gsum[4] += di * cprev; }
      }
    }
    tock(st, 5, tm);
    __syncthreads();  // own_dg[par] complete before anybody multiplies with it; shares[] free for the next collection
  }
  if (a.grad_partial) {   // the chain's 8 streams meet in LDS (stream order), one row of 16 cells per quantity goes out per workgroup
    float *gl = shares;   // [stream 8][quantity 7][cell 16]: free after the loop's last barrier
    if (threadIdx.x < 128) {
#pragma unroll
      for (int k = 0; k < 7; k++) gl[(sl * 7 + k) * 16 + cc] = gsum[k];
    }
    __syncthreads();
    if (threadIdx.x < 7 * 16) {
      const int k = threadIdx.x >> 4, c = threadIdx.x & 15;
      float v = gl[k * 16 + c];
#pragma unroll
      for (int q = 1; q < kChainStreams; q++) v += gl[(q * 7 + k) * 16 + c];
      if (c0 + c < C) a.grad_partial[((long)chain * 7 + k) * a.grad_ld + c0 + c] = v;
    }
  }
  if (st.timing && blockIdx.x == 0 && threadIdx.x == 0) { timing_flush(st); st.timing[0] += (unsigned long long)T; st.timing[6] += R.local ? 1ull : 0ull; }
  if (st.trace && threadIdx.x == 0) {   // ring of the 8 latest launches
    unsigned long long *tr = st.trace + (st.epoch & 7u) * 2048u;
    tr[2 * blockIdx.x] = (unsigned long long)t_entry; tr[2 * blockIdx.x + 1] = wall_clock64();
  }
  if (polls && lane == 0) __hip_atomic_fetch_add(st.abort_flag + 2, polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- backward, product on fp16 matrix instructions with fp32-equivalent operands ------------------------------------------------------
// lstm_seq_bwd with the share product dG(next) W_eff carried as in lstm_seq_fwd_h: the fp32 instruction made "product + publication" the
// longest phase of a backward timestep (1.4 of ~3 us).  The gate diffs a workgroup multiplies are its own (LDS), so their power-of-two
// scale is per stream AND timestep (the largest of the 16 * G values of a stream goes to [2^13, 2^14)): gradients of any magnitude keep 22
// significant bits.  The share layout inside a producer -> consumer block changes with the instruction's result layout (8-byte pieces).
template <bool CIFG, int TPW>
__global__ void __launch_bounds__(512) lstm_seq_bwd_h(aslp_lstm_seq a, SeqStatus st, unsigned *place, float *inbox) {
  constexpr int G = CIFG ? 3 : 4;
  constexpr int AP = 64 + 8;   // halves per operand row: K = the workgroup's own 16 * G gate columns (two chunks of 32; CIFG leaves 16 zeros)
  __shared__ __attribute__((aligned(16))) _Float16 a_h[2][16][AP];        // [parity][rows 0..7 = dG_hi of stream r, 8..15 = dG_lo' of stream r - 8][gate * 16 + cell]
  __shared__ __attribute__((aligned(16))) float row_inv[2][kChainStreams];   // [parity][stream]: 1 / (the power of two its gate diffs were scaled by)
  __shared__ __attribute__((aligned(16))) float dmsum[128];   // the chain's shares of this workgroup's 8 x 16 d_m values, summed (layout of one producer block)
  __shared__ float shares[kChainStreams * 7 * 16];              // scratch of the final bias / peephole reduction
  __shared__ int fail[2][8];
  __shared__ int place_flag;
  const long t_entry = st.trace ? (long)wall_clock64() : 0;
  __builtin_amdgcn_s_setprio(3);   // a latency chain: when weight-gradient workgroups share the CU (side stream), this kernel's waves issue first
  timing_begin(st);
  const int SE = a.s_count > 0 ? a.s_begin + a.s_count : a.S;
  const ChainRole R = chain_role(SE - a.s_begin, a.ndir, a.C, st, place, &place_flag);
  if (!R.active) {
    if (threadIdx.x == 0) {
      if (a.dmax_parts[0]) a.dmax_parts[0][blockIdx.x] = 0.f;
      if (a.dmax_parts[1]) a.dmax_parts[1][blockIdx.x] = 0.f;
    }
    return;
  }
  const aslp_lstm_seq_dir D = a.dir[R.dir];
  const int C = a.C, S = a.S, T = a.T, ld = a.ld;
  const int GC = G * C, oc = GC, oh = GC + C, om = GC + 2 * C;
  const int og = 0, oi = C, of = CIFG ? C : 2 * C, oo = CIFG ? 2 * C : 3 * C;
  const int c0 = R.c0, s0 = a.s_begin + R.s0;
  const int chain = blockIdx.x & (kMaxChains - 1), me = blockIdx.x >> 3, wpc = (C + kCellsPerWg - 1) / kCellsPerWg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // The product runs on v_mfma_f32_16x16x32_f16 with two-piece operands (see lstm_seq_fwd_h): A rows 0..7 = dG_hi of the chain's streams,
  // rows 8..15 = dG_lo' (dG scaled per stream and timestep by a power of two, so gradients of any size keep 22 bits), B = this workgroup's
  // 16 * G rows of W_eff as w_hi / w_lo after a power-of-two scale per column.  This wave's cells: consumer blocks wave + 8 j, one 16-cell
  // tile each; K = 64 = two chunks -> 4 instructions per tile.
  const int hr = lane & 15, hg = lane >> 4, hg2 = hg & 1, eodd = lane >> 5;
  half8 bh[TPW][2], bl[TPW][2];
  float csc[TPW];
#pragma unroll
  for (int tj = 0; tj < TPW; tj++) {
    const int cb = wave + 8 * tj, col = cb * 16 + hr;
    const bool colok = cb < wpc && col < C;
    float wv[2][8];
    float cmax = 0.f;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int kk = 32 * j + 8 * hg + i, cellk = c0 + (kk & 15);
        wv[j][i] = (colok && (kk >> 4) < G && cellk < C) ? D.w[(long)((kk >> 4) * C + cellk) * a.ldw + col] : 0.f;
        cmax = fmaxf(cmax, fabsf(wv[j][i]));
      }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 16));   // the column's other k-groups
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
    int e = 0;
    (void)frexpf(cmax, &e);
    const bool scaled = cmax > 0.f && cmax < 3.0e38f;
    const float sc = scaled ? ldexpf(1.f, 14 - e) : 1.f;
    csc[tj] = scaled ? ldexpf(1.f, e - 14) : 1.f;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float x = wv[j][i] * sc;
        const _Float16 xh = (_Float16)x;
        bh[tj][j][i] = xh;
        bl[tj][j][i] = (_Float16)(x - (float)xh);
      }
  }
  for (int p = threadIdx.x; p < 2 * 16 * AP / 8; p += 512) reinterpret_cast<u32x4 *>(&a_h[0][0][0])[p] = u32x4{0u, 0u, 0u, 0u};   // (the loop's first barrier publishes it)
  // inbox geometry
  const size_t slot_words = (size_t)kMaxChains * kMaxWgPerChain * kMaxWgPerChain * 128;  // floats per ring slot
  float *chain_box = inbox + (size_t)chain * kMaxWgPerChain * kMaxWgPerChain * 128;
  // epilogue role: threads 0..127 own one (stream, cell) pair each
  const int sl = threadIdx.x >> 4, cc = threadIdx.x & 15, s = s0 + sl, cell = c0 + cc;
  const bool live = threadIdx.x < 128 && s < SE && cell < C;
  const int cq = live ? cell : 0, sq = live ? s : 0;
  const float pf = D.peep_f[cq], po = D.peep_o[cq], pi = CIFG ? 0.f : D.peep_i[cq];
  // own-cell quantities of the step processed just before (BPTT order): all zero ahead of the first step
  float dn_c = 0.f, dn_f = 0.f, dn_i = 0.f;
  float gsum[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float dmax = 0.f;   // largest finite gate diff of this thread's stream so far (a.dmax_parts)
  unsigned polls = 0u;
  for (int step = 0; step < T; step++) {
    // BPTT runs against the direction's recursion: reverse = 0 (t = T..1), reverse = 1 (t = 1..T)
    const int t = D.reverse ? 1 + step : T - step;
    const int tn = D.reverse ? t - 1 : t + 1, tp = D.reverse ? t + 1 : t - 1;
    const int par = step & 1;
    const long o_ = ((long)t * S + sq) * ld;
    long tm = tick(st);
    float dm = 0.f, yo = 0.f, yh = 0.f, yg = 0.f, yf = 0.f, yi = 0.f, yn_f = 0.f, cprev = 0.f, ccur = 0.f;
    if (live) {  // everything that does not depend on the other workgroups, requested first
      dm = D.d[o_ + om + cq];
      yo = D.y[o_ + oo + cq]; yh = D.y[o_ + oh + cq]; yg = D.y[o_ + og + cq]; yf = D.y[o_ + of + cq];
      if (!CIFG) yi = D.y[o_ + oi + cq];
      yn_f = D.y[((long)tn * S + sq) * ld + of + cq];
      cprev = D.y[((long)tp * S + sq) * ld + oc + cq];
      if (a.grad_partial) ccur = D.y[o_ + oc + cq];
    }
    bool ok = true;
    if (step > 0) {
      float *box = chain_box + (size_t)(step % kRing) * slot_words;
      // 1. my share of d_m for every cell of the chain: own gate diffs of the previous step (LDS) x my rows of W_eff
      {
        half8 af[2];
#pragma unroll
        for (int j = 0; j < 2; j++) af[j] = *reinterpret_cast<const half8 *>(&a_h[par ^ 1][hr][32 * j + 8 * hg]);
        const f32x4 rinv = *reinterpret_cast<const f32x4 *>(&row_inv[par ^ 1][4 * hg2]);   // streams 4 hg2 + e of this lane's result registers
        f32x4 hacc[TPW];
#pragma unroll
        for (int tj = 0; tj < TPW; tj++) hacc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; j++) {
#pragma unroll
          for (int tj = 0; tj < TPW; tj++) hacc[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], bh[tj][j], hacc[tj], 0, 0, 0);
#pragma unroll
          for (int tj = 0; tj < TPW; tj++) hacc[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], bl[tj][j], hacc[tj], 0, 0, 0);
        }
        // 2. hand the shares out.  Rows 0..7 (lanes 0..31) hold dG_hi W, rows 8..15 dG_lo' W of the same streams; one v_permlane32_swap joins
        // the halves of two result registers (see lstm_seq_fwd_h), leaving registers 0 / 2 in the lower lanes and 1 / 3 in the upper: an
        // 8-byte piece per lane, [row group][odd][column][register pair] inside the 512 bytes a producer sends a consumer.
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(box);
        const float half_f = lane < 32 ? 1.f : 0x1p-11f;
#pragma unroll
        for (int tj = 0; tj < TPW; tj++) {
          const int cb = wave + 8 * tj;
          const float f = csc[tj] * half_f;
          const auto s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(hacc[tj][0] * (f * rinv[0])), __float_as_uint(hacc[tj][1] * (f * rinv[1])), false, false);
          const auto s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(hacc[tj][2] * (f * rinv[2])), __float_as_uint(hacc[tj][3] * (f * rinv[3])), false, false);
          if (cb < wpc) {
            const float v01 = __uint_as_float(s01[0]) + __uint_as_float(s01[1]), v23 = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);
            const int off = ((cb * kMaxWgPerChain + me) * 128 + ((hg2 * 2 + eodd) * 16 + hr) * 2) * 4;
            const u32x2 pk = {__float_as_uint(v01), __float_as_uint(v23)};
            if (R.local) __builtin_amdgcn_raw_buffer_store_b64(pk, rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b64(pk, rs, off, 0, kAuxSc1);
          }
        }
      }
      tock(st, 1, tm);  // product + publication
      // 3. collect what the chain's workgroups sent me AND add it up on the way: a producer block is 32 pieces of 16 bytes; wave w takes
      // pieces 4 w .. 4 w + 3, one per row of 16 lanes, lane j of the row producers j and j + 16.  The row's 16 partial sums meet in lane 0
      // over four DPP row shifts (a fixed tree: deterministic), which writes the piece's total to LDS -- the gate-diff threads then read ONE
      // value each instead of adding 32 (that loop was 0.55 us of the timestep, on two waves while six waited).
      {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(box + (size_t)me * kMaxWgPerChain * 128);
        const int q = 4 * wave + (lane >> 4), pa = lane & 15, pb = pa + 16;
        const int i0 = pa * 32 + q, i1 = pb * 32 + q;
        const bool h0 = pa < wpc, h1 = pb < wpc;
        u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
        long t0 = 0;
        for (unsigned spins = 0;; spins++) {
          if (h0) v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, i0 * 16, 0, kAuxSc1);
          if (h1) v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, i1 * 16, 0, kAuxSc1);
          if (!__any((h0 && has_sentinel(v0)) || (h1 && has_sentinel(v1)))) break;
          asm volatile("" ::: "memory");
          polls++;
          if (!spin_ok(spins, t0, st)) { ok = false; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        // give the slot back (sentinel) and make sure that has happened before anything of mine is published again
        const u32x4 sent = {kSentinel, kSentinel, kSentinel, kSentinel};
        if (R.local) {
          if (h0) __builtin_amdgcn_raw_buffer_store_b128(sent, rs, i0 * 16, 0, 0);
          if (h1) __builtin_amdgcn_raw_buffer_store_b128(sent, rs, i1 * 16, 0, 0);
        } else {
          if (h0) __builtin_amdgcn_raw_buffer_store_b128(sent, rs, i0 * 16, 0, kAuxSc1);
          if (h1) __builtin_amdgcn_raw_buffer_store_b128(sent, rs, i1 * 16, 0, kAuxSc1);
        }
        f32x4 sum;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float x = (h0 ? as_f(v0[c]) : 0.f) + (h1 ? as_f(v1[c]) : 0.f);
          x += row_up<8>(x); x += row_up<4>(x); x += row_up<2>(x); x += row_up<1>(x);   // lane 0 of the row: all 16 lanes' values
          sum[c] = x;
        }
        if ((lane & 15) == 0) *reinterpret_cast<f32x4 *>(&dmsum[4 * q]) = sum;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      tock(st, 2, tm);  // collection
    }
    if (lane == 0) fail[par][wave] = ok ? 0 : 1;
    __syncthreads();
    tock(st, 4, tm);
    {
      int f = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) f |= fail[par][w];
      if (f) return;
    }
    if (threadIdx.x < 128) {
      if (step > 0)   // the chain's shares of this pair, already added up: [row group][odd register][column][register pair] (see the publication)
        dm += dmsum[(((sl >> 2) * 2 + (sl & 1)) * 16 + cc) * 2 + ((sl & 3) >> 1)];
      tock(st, 3, tm);  // (devtools) shares summed
      const float dh = dtanh(yh, dm * yo);
      const float dov = dsigm(yo, dm * yh);
      float dc = dh + dn_c * yn_f;
      if (!CIFG) dc += dn_i * pi;
      dc += dn_f * pf;
      dc += dov * po;
      float dg, df, di = 0.f;
      if (!CIFG) {
        df = dsigm(yf, dc * cprev);
        di = dsigm(yi, dc * yg);
        dg = dtanh(yg, dc * yi);
      } else {
        df = dsigm(yf, dc * cprev - dc * yg);
        dg = dtanh(yg, dc - dc * yf);
      }
      // the next step's left operand stays here; the buffer copy is for the batched products after the launch
      {
        // this stream's gate diffs, scaled by the power of two that puts the largest of its 16 * G in [2^13, 2^14), as two fp16 pieces
        float rmax = fmaxf(fmaxf(fabsf(dg), fabsf(df)), fmaxf(fabsf(di), fabsf(dov)));
        rmax = fmaxf(rmax, row_ror<8>(rmax)); rmax = fmaxf(rmax, row_ror<4>(rmax));
        rmax = fmaxf(rmax, row_ror<2>(rmax)); rmax = fmaxf(rmax, row_ror<1>(rmax));   // the 16 lanes of a stream (tid = stream * 16 + cell) = one DPP row
        int e = 0;
        (void)frexpf(rmax, &e);
        const bool scaled = rmax > 0.f && rmax < 3.0e38f;
        dmax = fmaxf(dmax, scaled ? rmax : 0.f);
        const int up = scaled ? min(14 - e, 120) : 0;
        const float sc = ldexpf(1.f, up), inv = ldexpf(1.f, -up);
        auto put = [&](int k, float x) {
          const float xs = x * sc;
          const _Float16 xh = (_Float16)xs;
          a_h[par][sl][k] = xh;
          a_h[par][8 + sl][k] = (_Float16)((xs - (float)xh) * 2048.f);
        };
        put(cc, dg);
        if (!CIFG) { put(16 + cc, di); put(32 + cc, df); put(48 + cc, dov); }
        else { put(16 + cc, df); put(32 + cc, dov); }
        if (cc == 0) row_inv[par][sl] = inv;
      }
      if (live) {
        D.d[o_ + og + cell] = dg; D.d[o_ + of + cell] = df; D.d[o_ + oo + cell] = dov;
        if (!CIFG) D.d[o_ + oi + cell] = di;
        D.d[o_ + om + cell] = dm;  // the reference's d_m (lc.h:793) -- kept for InfoGradient-style dumps
        D.d[o_ + oh + cell] = dh;
        D.d[o_ + oc + cell] = dc;
      }
      dn_c = dc; dn_f = df; dn_i = di;
      if (a.grad_partial && live) {   // what the bias and peephole gradients are sums of (lc.h:1005-1058), this pair's share
        gsum[0] += dg; gsum[2] += df; gsum[3] += dov;
        gsum[5] += df * cprev; gsum[6] += dov * ccur;
        if (!CIFG) { gsum[1] += di; gsum[4] += di * cprev; }
      }
    }
    tock(st, 5, tm);
    __syncthreads();  // a_h[par] complete before anybody multiplies with it; shares[] free for the next collection
  }
  if (a.grad_partial) {   // the chain's 8 streams meet in LDS (stream order), one row of 16 cells per quantity goes out per workgroup
    float *gl = shares;   // [stream 8][quantity 7][cell 16]: free after the loop's last barrier
    if (threadIdx.x < 128) {
#pragma unroll
      for (int k = 0; k < 7; k++) gl[(sl * 7 + k) * 16 + cc] = gsum[k];
    }
    __syncthreads();
    if (threadIdx.x < 7 * 16) {
      const int k = threadIdx.x >> 4, c = threadIdx.x & 15;
      float v = gl[k * 16 + c];
#pragma unroll
      for (int q = 1; q < kChainStreams; q++) v += gl[(q * 7 + k) * 16 + c];
      if (c0 + c < C) a.grad_partial[((long)chain * 7 + k) * a.grad_ld + c0 + c] = v;
    }
  }
  if (a.dmax_parts[0] != nullptr || a.dmax_parts[1] != nullptr) {   // uniform.  (a dead lane's gate diffs are zeros: the row maximum is the live cells')
    __syncthreads();
    if (threadIdx.x < 128 && cc == 0) shares[sl] = dmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = shares[0];
#pragma unroll
      for (int q = 1; q < kChainStreams; q++) m = fmaxf(m, shares[q]);
      if (a.dmax_parts[0]) a.dmax_parts[0][blockIdx.x] = R.dir == 0 ? m : 0.f;
      if (a.dmax_parts[1]) a.dmax_parts[1][blockIdx.x] = R.dir == 1 ? m : 0.f;
    }
  }
  if (st.timing && blockIdx.x == 0 && threadIdx.x == 0) { timing_flush(st); st.timing[0] += (unsigned long long)T; st.timing[6] += R.local ? 1ull : 0ull; }
  if (st.trace && threadIdx.x == 0) {   // ring of the 8 latest launches
    unsigned long long *tr = st.trace + (st.epoch & 7u) * 2048u;
    tr[2 * blockIdx.x] = (unsigned long long)t_entry; tr[2 * blockIdx.x + 1] = wall_clock64();
  }
  if (polls && lane == 0) __hip_atomic_fetch_add(st.abort_flag + 2, polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ---- GruStreams (nnet-gru-streams.h:238-430): the same persistent scheme, two hand-offs per timestep ------------------------------
// Buffer columns [z | r | m | g | h], H each.  In each direction of time a GRU step is two products that depend on each other:
//   forward   zr(t) += h(t-1) W_zr_h^T -> z, r = sigmoid, g = r .* h(t-1)     then   m(t) += g(t) W_m_g^T -> m = tanh, h(t) = h(t-1) - h(t-1) z + z m
//   backward  d_h(t) += [d_z | d_r](t+1) W_zr_h -> d_h, d_m                    then   d_g(t) = d_m(t) W_m_g -> d_r, d_z
// so every timestep is two rounds of "collect the chain's left operand (8 streams x K, published by the chain's workgroups in the
// previous round) into LDS -- multiply with this workgroup's columns of the recurrent matrix (registers, resident for the launch,
// v_mfma_f32_4x4x1, 8 waves split K) -- finish the 8 x 16 (stream, cell) pairs -- publish 16-byte pieces".  Chains, placement check,
// "the data is its own flag" hand-off and the bounded spins are those of the LSTM kernels above; GruStreams is unidirectional, so
// S = 32 gives 4 chains of 32 workgroups (H = 512).  A lane pair owns a (stream, cell): lane 0 of the pair z (forward) / d_z, d_h,
// d_m, d_g (backward), lane 1 r and g / d_r; h(t-1) and the terms the backward step carries from t+1 never leave the pair's registers.
//
// One MFMA instruction is 16 independent 4 x 4 blocks.  The first forward product has 32 columns per workgroup (z and r of its 16
// cells): blocks = 2 stream quads x 8 column quads ("wide").  The other three products have 16 columns: blocks = 2 stream quads x 4
// column quads x 2 HALVES OF THE WAVE'S K SLICE ("split") -- no idle blocks, half the instructions and half the weight registers;
// the two halves' partial sums are two of the 16 terms the epilogue adds per output.
struct GruGeom {
  int lane, wave, qs, qc, jl, kh, col16, pair, role, sl, cc;
};
__device__ __forceinline__ GruGeom gru_geom() {
  GruGeom g;
  g.lane = threadIdx.x & 63; g.wave = threadIdx.x >> 6;
  g.qs = (g.lane >> 2) & 1; g.qc = g.lane >> 3; g.jl = g.lane & 3;
  g.kh = g.qc >> 2; g.col16 = 4 * (g.qc & 3) + g.jl;   // split products: K half and column of this lane's block
  g.pair = threadIdx.x >> 1; g.role = threadIdx.x & 1;
  g.sl = (g.pair >> 4) & 7; g.cc = g.pair & 15;
  return g;
}
// Collects NP pieces per thread (piece p = threadIdx.x + 512 j of [stream][K / 4], 16 bytes each) of the chain's 8 streams from
// row block `base` columns [col0, col0 + K) into a_lds[stream][MP]; reloads until no piece reads "not yet published".
// Returns false on a timeout / device-wide abort (wave-uniform; the caller posts it in fail[] for the workgroup).
template <int NP>
__device__ __forceinline__ bool gru_collect(const float *base, int ld, int S, int s0, int col0, int K, float *a_lds, int MP, const SeqStatus &st,
                                            unsigned &polls) {
  const int k4 = K >> 2, npiece = kChainStreams * k4;
  int off[NP], dst[NP];
  bool have[NP];
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const int p = threadIdx.x + 512 * j;
    have[j] = p < npiece;
    const int sp = have[j] ? p / k4 : 0, kq = have[j] ? p % k4 : 0;
    off[j] = (min(s0 + sp, S - 1) * ld + col0 + 4 * kq) * 4;
    dst[j] = sp * MP + 4 * kq;
  }
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(base);
  u32x4 v[NP];
#pragma unroll
  for (int j = 0; j < NP; j++) v[j] = u32x4{0u, 0u, 0u, 0u};
  long t0 = 0;
  bool ok = true;
  for (unsigned spins = 0;; spins++) {
    bool missing = false;
#pragma unroll
    for (int j = 0; j < NP; j++) {
      if (have[j]) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], 0, kAuxSc1);
    }
#pragma unroll
    for (int j = 0; j < NP; j++) missing = missing || (have[j] && has_sentinel(v[j]));
    if (!__any(missing)) break;
    asm volatile("" ::: "memory");
    polls++;
    if (!spin_ok(spins, t0, st)) { ok = false; break; }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int j = 0; j < NP; j++)
    if (have[j]) *reinterpret_cast<u32x4 *>(a_lds + dst[j]) = v[j];
  return ok;
}
// This lane's K slice [k0, k0 + klen) of (8 streams x K) x (K x columns): NB fragments of 4 k values.  Result register r of a lane =
// stream 4 qs + r of the lane's column.  Slices past K read element 0 of the row (finite; their B fragments are 0).
template <int NB>
__device__ __forceinline__ f32x4 gru_product(const float *a_lds, int MP, const f32x4 (&bw)[NB], int k0, int klen, int K, const GruGeom &g) {
  const float *arow = a_lds + (4 * g.qs + g.jl) * MP;
  // operand reads pinned PD fragments ahead of their products (see lstm_seq_fwd: left alone, the scheduler waits for each read right
  // behind the previous fragment's four products)
  constexpr int PD = NB < 6 ? NB : 6;
  auto load_a = [&](int i) -> f32x4 { return *reinterpret_cast<const f32x4 *>(arow + ((4 * i < klen && k0 + 4 * i < K) ? k0 + 4 * i : 0)); };
  f32x4 av[NB];
#pragma unroll
  for (int i = 0; i < PD; i++) av[i] = load_a(i);
  f32x4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NB; i++) {   // four independent accumulators: the instruction's latency is hidden without padding
    if (i + PD < NB) av[i + PD] = load_a(i + PD);
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].x, bw[i].x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].y, bw[i].y, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].z, bw[i].z, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].w, bw[i].w, acc[3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// loads NB fragments of row `row` (K-contiguous, `ldw` floats apart) starting at k0; zero where !valid or beyond klen / K
template <int NB>
__device__ __forceinline__ void gru_load_b(f32x4 (&bw)[NB], const float *w, int ldw, int row, bool valid, int k0, int klen, int K) {
  const float *b = w + (long)(valid ? row : 0) * ldw;
#pragma unroll
  for (int i = 0; i < NB; i++)
    bw[i] = (valid && 4 * i < klen && k0 + 4 * i < K) ? *reinterpret_cast<const f32x4 *>(b + k0 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
}
// partial sums of a round meet in LDS: wide products [wave 8][stream 8][kGruRPW], split products [wave * 2 + K half 16][stream 8][kGruRPS]
constexpr int kGruRPW = 48, kGruRPS = 24, kGruRed = 8 * kChainStreams * kGruRPW;   // = 16 * 8 * 24 floats
__device__ __forceinline__ void gru_store_wide(float *red, const f32x4 &sum, const GruGeom &g) {
  float *rp = red + (g.wave * kChainStreams + 4 * g.qs) * kGruRPW + 4 * g.qc + g.jl;
  rp[0 * kGruRPW] = sum.x; rp[1 * kGruRPW] = sum.y; rp[2 * kGruRPW] = sum.z; rp[3 * kGruRPW] = sum.w;
}
__device__ __forceinline__ void gru_store_split(float *red, const f32x4 &sum, const GruGeom &g) {
  float *rp = red + ((2 * g.wave + g.kh) * kChainStreams + 4 * g.qs) * kGruRPS + g.col16;
  rp[0 * kGruRPS] = sum.x; rp[1 * kGruRPS] = sum.y; rp[2 * kGruRPS] = sum.z; rp[3 * kGruRPS] = sum.w;
}
__device__ __forceinline__ float gru_sum_wide(const float *red, int sl, int col) {
  float v = red[sl * kGruRPW + col];
#pragma unroll
  for (int w = 1; w < 8; w++) v += red[(w * kChainStreams + sl) * kGruRPW + col];
  return v;
}
__device__ __forceinline__ float gru_sum_split(const float *red, int sl, int col) {
  float v = red[sl * kGruRPS + col];
#pragma unroll
  for (int w = 1; w < 16; w++) v += red[(w * kChainStreams + sl) * kGruRPS + col];
  return v;
}
// 16-byte piece = this lane's value and those of the lanes 2, 4, 6 places up in its row of 16 (the same role of the next three cells)
__device__ __forceinline__ u32x4 gru_piece(float v) {
  const float v1 = row_up<2>(v), v2 = row_up<4>(v), v3 = row_up<6>(v);
  return u32x4{__float_as_uint(v), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)};
}
__device__ __forceinline__ void gru_publish(const u32x4 &pk, float *rowblock, int off_floats, bool local) {
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(rowblock);
  if (local) __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off_floats * 4, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(pk, rs, off_floats * 4, 0, kAuxSc1);
}
// K values per wave for a reduction of length K: a multiple of 8 (so that the split products' halves are whole fragments), 8 waves cover K
__device__ __forceinline__ int gru_kw(int K) { return ((K + 63) / 64) * 8; }
__device__ __forceinline__ bool gru_failed(const int (*fail)[8], int round) {
  int f = 0;
#pragma unroll
  for (int w = 0; w < 8; w++) f |= fail[round][w];
  return f != 0;  // uniform: every wave reads the same eight words
}

// grid 8 * ceil(H / 16) workgroups of 512 threads (chains >= S / 8 leave at once).  KW: K values per wave, H <= 8 * KW.
template <int KW>
__global__ void __launch_bounds__(512) gru_seq_fwd(aslp_gru_seq a, SeqStatus st, unsigned *place) {
  constexpr int NBW = KW / 4, NBS = KW / 8, MP = 8 * KW + 4;
  __shared__ __attribute__((aligned(16))) float a_lds[kChainStreams * MP];
  __shared__ float red[2][kGruRed];
  __shared__ int fail[2][8];
  __shared__ int place_flag;
  const long t_entry = st.trace ? (long)wall_clock64() : 0;
  const int SE = a.s_count > 0 ? a.s_begin + a.s_count : a.S;
  const ChainRole R = chain_role(SE - a.s_begin, 1, a.H, st, place, &place_flag);
  if (!R.active) return;
  const int H = a.H, S = a.S, T = a.T, ld = a.ld;
  const int omm = 2 * H, og = 3 * H, oh = 4 * H;
  const int c0 = R.c0, s0 = a.s_begin + R.s0;
  const GruGeom g = gru_geom();
  const int kw = gru_kw(H), kb = g.wave * kw, kbs = kb + g.kh * (kw / 2);
  // B fragments, resident for the launch.  Round 1 (wide): column n = 4 qc + jl is gate n >> 4 (z, r) of cell c0 + (n & 15);
  // round 2 (split): column col16 is m of cell c0 + col16, this lane's half of the wave's K slice
  f32x4 bw1[NBW], bw2[NBS];
  {
    const int n = 4 * g.qc + g.jl;
    gru_load_b<NBW>(bw1, a.w_zr, a.ldw_zr, (n >> 4) * H + c0 + (n & 15), c0 + (n & 15) < H, kb, kw, H);
    gru_load_b<NBS>(bw2, a.w_m, a.ldw_m, c0 + g.col16, c0 + g.col16 < H, kbs, kw / 2, H);
  }
  const int s = s0 + g.sl, cell = c0 + g.cc;
  const bool live = threadIdx.x < 256 && s < SE && cell < H;
  const int cq = live ? cell : 0, sq = live ? s : 0;
  float hp = live ? a.y[(long)sq * ld + oh + cq] : 0.f;   // h(0): the carried history in row block 0
  unsigned polls = 0u;
  for (int step = 0; step < T; step++) {
    const int t = 1 + step;
    float *ys = a.y + ((long)t * S + sq) * ld;
    // the x-parts (+ bias) of this pair: written before the launch, requested before the hand-off wait
    const float xg = live ? ys[g.role * H + cq] : 0.f;
    const float xm = (live && g.role == 0) ? ys[omm + cq] : 0.f;
    // ---- round 1: h(t-1) -> z, r, g ------------------------------------------------------------------------------
    bool ok = gru_collect<2>(a.y + (long)(t - 1) * S * ld, ld, SE, s0, oh, H, a_lds, MP, st, polls);
    if (g.lane == 0) fail[0][g.wave] = ok ? 0 : 1;
    __syncthreads();
    gru_store_wide(red[0], gru_product<NBW>(a_lds, MP, bw1, kb, kw, H, g), g);
    __syncthreads();
    if (gru_failed(fail, 0)) return;
    float gate = 0.f;
    if (threadIdx.x < 256) gate = sigmoid_ref(xg + gru_sum_wide(red[0], g.sl, g.role * 16 + g.cc));
    const float zz = gate;                       // meaningful on lane 0 of the pair
    const float gg = gate * hp;                  // g = r .* h(t-1), meaningful on lane 1
    {
      const u32x4 pk = gru_piece(gg);
      if (live && g.role == 1 && (g.cc & 3) == 0) gru_publish(pk, a.y + (long)t * S * ld, s * ld + og + cell, R.local);
      if (live) ys[g.role * H + cell] = gate;
    }
    // ---- round 2: g(t) -> m, h -----------------------------------------------------------------------------------
    ok = gru_collect<2>(a.y + (long)t * S * ld, ld, SE, s0, og, H, a_lds, MP, st, polls);
    if (g.lane == 0) fail[1][g.wave] = ok ? 0 : 1;
    __syncthreads();
    gru_store_split(red[1], gru_product<NBS>(a_lds, MP, bw2, kbs, kw / 2, H, g), g);
    __syncthreads();
    if (gru_failed(fail, 1)) return;
    float mm = 0.f, hh = 0.f;
    if (threadIdx.x < 256 && g.role == 0) {
      mm = tanh_ref(xm + gru_sum_split(red[1], g.sl, g.cc));
      hh = hp - hp * zz + zz * mm;
    }
    {
      const u32x4 pk = gru_piece(hh);
      if (live && g.role == 0 && (g.cc & 3) == 0) gru_publish(pk, a.y + (long)t * S * ld, s * ld + oh + cell, R.local);
      if (live && g.role == 0) ys[omm + cell] = mm;
    }
    // h(t) to both lanes of the pair
    const float from_left = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(hh), 0x111, 0xF, 0xF, true));  // row_shr:1
    hp = g.role == 0 ? hh : from_left;
  }
  if (st.trace && threadIdx.x == 0) {
    unsigned long long *tr = st.trace + (st.epoch & 7u) * 2048u;
    tr[2 * blockIdx.x] = (unsigned long long)t_entry; tr[2 * blockIdx.x + 1] = wall_clock64();
  }
  if (polls && g.lane == 0) __hip_atomic_fetch_add(st.abort_flag + 2, polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// backward: a.w_zr = W_zr_h^T [H x 2H], a.w_m = W_m_g^T [H x H] (K-contiguous rows of this workgroup's 16 cells); both products split.
// KW1 / KW2: K values per wave of the two products (2H <= 8 KW1, H <= 8 KW2).
template <int KW1, int KW2>
__global__ void __launch_bounds__(512) gru_seq_bwd(aslp_gru_seq a, SeqStatus st, unsigned *place) {
  constexpr int NB1 = KW1 / 8, NB2 = KW2 / 8, MP = 8 * KW1 + 4;
  __shared__ __attribute__((aligned(16))) float a_lds[kChainStreams * MP];
  __shared__ float red[2][kGruRed];
  __shared__ int fail[2][8];
  __shared__ int place_flag;
  const long t_entry = st.trace ? (long)wall_clock64() : 0;
  const int SE = a.s_count > 0 ? a.s_begin + a.s_count : a.S;
  const ChainRole R = chain_role(SE - a.s_begin, 1, a.H, st, place, &place_flag);
  if (!R.active) return;
  const int H = a.H, S = a.S, T = a.T, ld = a.ld;
  const int orr = H, omm = 2 * H, og = 3 * H, oh = 4 * H;
  const int c0 = R.c0, s0 = a.s_begin + R.s0;
  const GruGeom g = gru_geom();
  const int kw1 = gru_kw(2 * H), k1 = g.wave * kw1 + g.kh * (kw1 / 2), kw2 = gru_kw(H), k2 = g.wave * kw2 + g.kh * (kw2 / 2);
  f32x4 bwa[NB1], bwb[NB2];
  gru_load_b<NB1>(bwa, a.w_zr, a.ldw_zr, c0 + g.col16, c0 + g.col16 < H, k1, kw1 / 2, 2 * H);
  gru_load_b<NB2>(bwb, a.w_m, a.ldw_m, c0 + g.col16, c0 + g.col16 < H, k2, kw2 / 2, H);
  const int s = s0 + g.sl, cell = c0 + g.cc;
  const bool live = threadIdx.x < 256 && s < SE && cell < H;
  const int cq = live ? cell : 0, sq = live ? s : 0;
  float dhn = 0.f, zn = 0.f, dgn = 0.f, rn = 0.f;   // d_h, z, d_g, r of this pair at t + 1 (row block T + 1 is zero)
  unsigned polls = 0u;
  for (int step = 0; step < T; step++) {
    const int t = T - step;
    const long o = ((long)t * S + sq) * ld;
    const float yz = live ? a.y[o + cq] : 0.f, yr = live ? a.y[o + orr + cq] : 0.f, ym = live ? a.y[o + omm + cq] : 0.f;
    const float hprev = live ? a.y[o - (long)S * ld + oh + cq] : 0.f;   // h(t-1)
    const float dh_ext = live ? a.d[o + oh + cq] : 0.f;                  // the loss's share, stored before the launch
    // ---- round 1: [d_z | d_r](t+1) W_zr_h -> d_h, d_m ----------------------------------------------------------------
    bool ok = true;
    if (step > 0) ok = gru_collect<4>(a.d + (long)(t + 1) * S * ld, ld, SE, s0, 0, 2 * H, a_lds, MP, st, polls);
    if (g.lane == 0) fail[0][g.wave] = ok ? 0 : 1;
    __syncthreads();
    if (step > 0) gru_store_split(red[0], gru_product<NB1>(a_lds, MP, bwa, k1, kw1 / 2, 2 * H, g), g);
    __syncthreads();
    if (gru_failed(fail, 0)) return;
    float dh = 0.f, dm = 0.f;
    if (threadIdx.x < 256) {   // both lanes of the pair form d_h (lane 1 needs it for nothing; the arithmetic is uniform)
      const float prod = step > 0 ? gru_sum_split(red[0], g.sl, g.cc) : 0.f;
      dh = dh_ext + prod + dhn - dhn * zn + dgn * rn;
      dm = dtanh(ym, dh * yz);
    }
    {
      const u32x4 pk = gru_piece(dm);
      if (live && g.role == 0 && (g.cc & 3) == 0) gru_publish(pk, a.d + (long)t * S * ld, s * ld + omm + cell, R.local);
      if (live && g.role == 0) a.d[o + oh + cell] = dh;
    }
    // ---- round 2: d_m(t) W_m_g -> d_g, d_r, d_z ----------------------------------------------------------------------
    ok = gru_collect<2>(a.d + (long)t * S * ld, ld, SE, s0, omm, H, a_lds, MP, st, polls);
    if (g.lane == 0) fail[1][g.wave] = ok ? 0 : 1;
    __syncthreads();
    gru_store_split(red[1], gru_product<NB2>(a_lds, MP, bwb, k2, kw2 / 2, H, g), g);
    __syncthreads();
    if (gru_failed(fail, 1)) return;
    float dg = 0.f, dzr = 0.f;
    if (threadIdx.x < 256) {
      dg = gru_sum_split(red[1], g.sl, g.cc);
      dzr = g.role == 0 ? dsigm(yz, dh * ym - dh * hprev) : dsigm(yr, dg * hprev);
    }
    {
      const u32x4 pk = gru_piece(dzr);   // lane 0 of the pairs: four d_z; lane 1: four d_r
      if (live && (g.cc & 3) == 0) gru_publish(pk, a.d + (long)t * S * ld, s * ld + g.role * H + cell, R.local);
      if (live && g.role == 0) a.d[o + og + cell] = dg;
    }
    dhn = dh; zn = yz; dgn = dg; rn = yr;
  }
  if (st.trace && threadIdx.x == 0) {
    unsigned long long *tr = st.trace + (st.epoch & 7u) * 2048u;
    tr[2 * blockIdx.x] = (unsigned long long)t_entry; tr[2 * blockIdx.x + 1] = wall_clock64();
  }
  if (polls && g.lane == 0) __hip_atomic_fetch_add(st.abort_flag + 2, polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Finishes the bias / peephole gradients of one direction from the per-chain sums lstm_seq_bwd left (grad_partial): thread = (quantity k,
// cell c); chains of the direction added in order; then momentum, clip and the folded SGD step as rnn_vec_grads_kernel does them.
struct SeqVecGradArgs {
  const float *partial; int ld, ndir, dir, nsg, C, cifg;
  float *corr[7], *param[7];   // per quantity: where its [C] slice lives (bias rows are slices of bias_corr / bias)
  float mmt, clip, neg_lr;
};
struct SeqVecGradPair { SeqVecGradArgs d[2]; };   // blockIdx.y picks the direction
__global__ void __launch_bounds__(256) lstm_seq_vec_grads_kernel(SeqVecGradPair gp) {
  const SeqVecGradArgs &g = gp.d[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 7 * g.C) return;
  const int k = idx / g.C, c = idx - k * g.C;
  if (g.corr[k] == nullptr) return;
  float v = 0.f;
  for (int sg = 0; sg < g.nsg; sg++) v += g.partial[((long)(sg * g.ndir + g.dir) * 7 + k) * g.ld + c];
  if (g.mmt != 0.0f) v += g.mmt * g.corr[k][c];
  if (g.clip > 0.0f) v = v < -g.clip ? -g.clip : (v > g.clip ? g.clip : v);
  g.corr[k][c] = v;
  if (g.neg_lr != 0.0f) g.param[k][c] += g.neg_lr * v;
}

// Prepares an activation buffer for the forward kernel: boundary row blocks 0 and T + 1 := 0 (all columns), columns
// [col0, col0 + ncols) of row blocks 1..T := "not yet published".  One launch instead of three memsets over the whole buffer.
// blockIdx.y picks the buffer (both directions of a layer in one launch); `init` (S rows of `ld_init` floats, may be NULL) is
// copied into row block 0 of buffer 0 instead of zeros -- the history the forward direction of a stream-carrying layer starts from.
__global__ void __launch_bounds__(256) lstm_seq_fill_kernel(float *buf0, float *buf1, int ld, int T, int S, int col0, int ncols,
                                                            const float *init, int ld_init, int init_cols) {
  const SeqFillJob f = {buf0, buf1, ld, T, S, col0, ncols, init, ld_init, init_cols};
  seq_fill_row(f, (int)blockIdx.x, (int)blockIdx.y);   // (split16.h: the same rows a conversion launch fills when it takes the job along)
}

// ---- host side ---------------------------------------------------------------------------------------------------------
struct SeqRuntime {
  unsigned *abort_flag = nullptr;  // device: [0] abort, [2] poll diagnostics, [16 ...] the placement table of chain_role
  unsigned *place = nullptr;
  float *inbox = nullptr;  // backward: kRing slots of [chain][consumer][producer][128 floats]
  unsigned long long *timing = nullptr;  // device, 8 words; handed to the kernels only while aslp_lstm_seq_timing(1) is in effect
  int timing_mode = 0;  // 0 off, 1 forward kernel, 2 backward kernel
  unsigned epoch = 0, err_seen = 0;
  bool ring_ready = false;
  hipEvent_t last_done = nullptr;   // (made when a launch first comes from another stream than its predecessor)
  hipStream_t last_stream = nullptr;   // the stream the latest persistent launch went to
  std::thread::id last_thread;         // ... and the host thread that issued it (its stream is only touched again by that thread)
  bool last_done_recorded = false;     // last_done already stands behind the latest launch (recorded by its own launcher)
  bool launched = false;
  unsigned *host_err = nullptr;    // mapped host memory (device-visible)
  unsigned *host_err_dev = nullptr;
  int num_cu = 0;
  bool ok = false;
  std::mutex launch_mu;   // one persistent launch (LSTM or GRU) is issued at a time: epoch, event chaining, abort word, placement table, ring
};
// Several PROCESSES on one GPU (ranks of parallel/comm.cpp's ShmComm, or two training jobs given the same device): each one's persistent
// grid is sized to be resident at once on an otherwise free device, and two of them half resident beside each other wait for workgroups
// that cannot be scheduled until the 2 s spin limit ends both.  With ASLP_DEVICE_SHARED=1 (or aslp_device_shared(1)) a persistent launch
// therefore holds a per-device file lock (flock on /dev/shm/aslp_seq_gate.<uid>.<pci bus id>) from before the launch until the kernel
// has completed -- one host-side stream synchronise per launch, the price of sharing.  Off by default: one process per GPU needs none of it.
struct DeviceGate {
  int fd = -1;
  bool on = false;
  std::once_flag once;
  void open_once() {
    std::call_once(once, [this] {
      int dev = 0;
      char bus[64] = "dev";
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetPCIBusId(bus, sizeof(bus), dev);
      for (char *c = bus; *c; c++) if (*c == ':' || *c == '/' || *c == '.') *c = '_';
      const std::string path = "/dev/shm/aslp_seq_gate." + std::to_string((long)getuid()) + "." + bus;
      fd = ::open(path.c_str(), O_CREAT | O_RDWR | O_CLOEXEC, 0600);
    });
  }
};
DeviceGate &device_gate() {
  static DeviceGate g;
  static std::once_flag env_once;
  std::call_once(env_once, [] { const char *e = getenv("ASLP_DEVICE_SHARED"); g.on = e != nullptr && e[0] == '1'; });
  return g;
}
// RAII around one persistent launch: lock -> (launch) -> wait for the kernel -> unlock.  A no-op unless the device is declared shared.
struct SharedDeviceLaunch {
  bool held = false;
  SharedDeviceLaunch() {
    DeviceGate &g = device_gate();
    if (!g.on) return;
    g.open_once();
    if (g.fd < 0) return;
    while (flock(g.fd, LOCK_EX) != 0 && errno == EINTR) {}
    held = true;
  }
  ~SharedDeviceLaunch() {
    if (!held) return;
    (void)hipStreamSynchronize(cur_stream());
    (void)flock(device_gate().fd, LOCK_UN);
  }
};

SeqRuntime &seq_runtime() {
  static SeqRuntime rt;
  static std::once_flag once;
  std::call_once(once, [] {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
    rt.num_cu = prop.multiProcessorCount;
    const size_t bytes = 64 + sizeof(unsigned) * 2 * kMaxChains * kMaxWgPerChain;
    if (hipMalloc(&rt.abort_flag, bytes) != hipSuccess || hipMemset(rt.abort_flag, 0, bytes) != hipSuccess) return;
    rt.place = rt.abort_flag + 16;
    // 8 timing words, a ring of 8 launches x 2048 residency words, 1024 placement-census words
    if (hipMalloc(&rt.timing, 64 + 8 * 18 * 1024) != hipSuccess || hipMemset(rt.timing, 0, 64 + 8 * 18 * 1024) != hipSuccess) return;   // + 1024 phase-log words
    if (hipMalloc(&rt.inbox, sizeof(float) * (size_t)kRing * kMaxChains * kMaxWgPerChain * kMaxWgPerChain * 128) != hipSuccess) return;
    if (hipHostMalloc(&rt.host_err, 64, hipHostMallocMapped) != hipSuccess) return;
    *rt.host_err = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&rt.host_err_dev), rt.host_err, 0) != hipSuccess) return;
    register_async_error_word(rt.host_err, "persistent LSTM recurrence: a workgroup timed out waiting for another workgroup's hand-off "
                                           "(results of that call are invalid; if several processes share this GPU set ASLP_DEVICE_SHARED=1)");
    rt.ok = true;
  });
  return rt;
}

// One persistent launch at a time: a launch that goes to another stream than its predecessor waits for it.  The event is recorded only then,
// on the predecessor's stream (it then stands behind that launch and whatever followed it there) -- an event record behind EVERY persistent
// launch cost the common single-stream case ~6 us of idle stream per launch (8 launches per cfg3 step).  Caller holds rt.launch_mu.
static void chain_behind_last_launch(SeqRuntime &rt) {
  if (!rt.launched || rt.last_stream == cur_stream()) return;
  if (rt.last_done_recorded && rt.last_done) {   // several launching threads: the predecessor left its own marker (note_launch)
    if (hipStreamWaitEvent(cur_stream(), rt.last_done, 0) == hipSuccess) return;
  } else if (rt.last_thread == std::this_thread::get_id()) {   // another stream of THIS thread (main / side): alive by construction
    if (!rt.last_done && hipEventCreateWithFlags(&rt.last_done, hipEventDisableTiming) != hipSuccess) rt.last_done = nullptr;
    if (rt.last_done && hipEventRecord(rt.last_done, rt.last_stream) == hipSuccess && hipStreamWaitEvent(cur_stream(), rt.last_done, 0) == hipSuccess) return;
  }
  // the predecessor came from another host thread without a marker (that thread, and a stream it owned, may be gone), or an event call
  // failed: wait for the device instead of touching a stream that is not ours
  (void)hipGetLastError();
  (void)hipDeviceSynchronize();
}
// behind a persistent launch (caller holds rt.launch_mu).  With one launching thread nothing is recorded (the ~6 us above); as soon as a
// second thread launches grid-wide kernels every launch leaves its marker on its OWN stream, so that a successor never has to record
// into a stream of another thread.
static void note_launch(SeqRuntime &rt) {
  rt.last_stream = cur_stream();
  rt.last_thread = std::this_thread::get_id();
  rt.launched = true;
  rt.last_done_recorded = false;
  if (grid_wide_threads() > 1) {
    if (!rt.last_done && hipEventCreateWithFlags(&rt.last_done, hipEventDisableTiming) != hipSuccess) rt.last_done = nullptr;
    if (rt.last_done && hipEventRecord(rt.last_done, cur_stream()) == hipSuccess) rt.last_done_recorded = true;
  }
}

typedef void (*SeqKernel)(aslp_lstm_seq, SeqStatus, unsigned *);
typedef void (*SeqKernelB)(aslp_lstm_seq, SeqStatus, unsigned *, float *);
bool fast_act() {   // A/B switch: ASLP_LSTM_FAST_ACT=0 keeps the correctly rounded expf / division of the reference's CPU code
  static const bool off = getenv("ASLP_LSTM_FAST_ACT") != nullptr && getenv("ASLP_LSTM_FAST_ACT")[0] == '0';
  return !off;
}
SeqKernel pick_fwd(bool cifg, int C) {
  if (fast_act()) {
    if (C <= 128) return cifg ? lstm_seq_fwd<true, 16, true> : lstm_seq_fwd<false, 16, true>;
    if (C <= 512) return cifg ? lstm_seq_fwd<true, 64, true> : lstm_seq_fwd<false, 64, true>;
    return nullptr;
  }
  if (C <= 128) return cifg ? lstm_seq_fwd<true, 16, false> : lstm_seq_fwd<false, 16, false>;
  if (C <= 512) return cifg ? lstm_seq_fwd<true, 64, false> : lstm_seq_fwd<false, 64, false>;
  return nullptr;
}
// product on fp16 matrix instructions with two-piece fp32-equivalent operands (lstm_seq_fwd_h): A/B switch ASLP_LSTM_SPLIT_F16
int g_lstm_split_override = -1;   // aslp_lstm_split16(): -1 = the environment decides
bool split_f16_on() {   // default on; ASLP_LSTM_SPLIT_F16=0 puts the recurrent products back on the fp32 instruction (lstm_seq_fwd / lstm_seq_bwd)
  static const bool off = getenv("ASLP_LSTM_SPLIT_F16") != nullptr && getenv("ASLP_LSTM_SPLIT_F16")[0] == '0';
  return g_lstm_split_override >= 0 ? g_lstm_split_override != 0 : !off;
}
SeqKernel pick_fwd_h(bool cifg, int C) {
  if (fast_act()) {
    if (C <= 256) return cifg ? lstm_seq_fwd_h<true, 1, true> : lstm_seq_fwd_h<false, 1, true>;
    if (C <= 512) return cifg ? lstm_seq_fwd_h<true, 2, true> : lstm_seq_fwd_h<false, 2, true>;
    return nullptr;
  }
  if (C <= 256) return cifg ? lstm_seq_fwd_h<true, 1, false> : lstm_seq_fwd_h<false, 1, false>;
  if (C <= 512) return cifg ? lstm_seq_fwd_h<true, 2, false> : lstm_seq_fwd_h<false, 2, false>;
  return nullptr;
}
SeqKernelB pick_bwd_h(bool cifg, int C) {
  if (C <= 128) return cifg ? lstm_seq_bwd_h<true, 1> : lstm_seq_bwd_h<false, 1>;
  if (C <= 512) return cifg ? lstm_seq_bwd_h<true, 4> : lstm_seq_bwd_h<false, 4>;
  return nullptr;
}
SeqKernelB pick_bwd(bool cifg, int C) {
  if (C <= 128) return cifg ? lstm_seq_bwd<true, 1> : lstm_seq_bwd<false, 1>;
  if (C <= 512) return cifg ? lstm_seq_bwd<true, 4> : lstm_seq_bwd<false, 4>;
  return nullptr;
}

typedef void (*GruKernel)(aslp_gru_seq, SeqStatus, unsigned *);
GruKernel pick_gru(bool backward, int H) {
  if (H <= 128) return backward ? gru_seq_bwd<32, 16> : gru_seq_fwd<16>;
  if (H <= 512) return backward ? gru_seq_bwd<128, 64> : gru_seq_fwd<64>;
  return nullptr;
}
bool gru_args_ok(const aslp_gru_seq *a, bool backward) {
  return a && a->y && a->w_zr && a->w_m && (!backward || a->d) && a->T > 0 && a->S > 0 && a->H > 0 && (a->H & 3) == 0 && (a->ld & 3) == 0 &&
         (a->ldw_zr & 3) == 0 && (a->ldw_m & 3) == 0 && a->ld >= 5 * a->H && aligned16(a->y) && aligned16(a->w_zr) && aligned16(a->w_m) &&
         (!backward || aligned16(a->d));
}

bool seq_args_ok(const aslp_lstm_seq *a) {
  return a && a->ndir >= 1 && a->ndir <= 2 && a->T > 0 && a->S > 0 && a->C > 0 && (a->C & 3) == 0 && (a->ld & 3) == 0 && (a->ldw & 3) == 0;
}

// the whole grid has to be resident at once.  Where the runtime reports room for two or more workgroups per CU one of them
// is left as slack (MI355X_MICROARCH.md: the API can be one block per CU high); a kernel that fits exactly once per CU
// cannot be over-reported -- it would not launch at all -- so one workgroup per CU is accepted as is.
// occupancy of a kernel at `threads` per workgroup, asked once per kernel (the engine probes *_supported on every Propagate / Backpropagate)
int cached_occupancy(const void *k, int threads) {
  static std::mutex mu;
  static std::vector<std::pair<const void *, int>> cache;
  std::lock_guard<std::mutex> lock(mu);
  for (auto &e : cache) if (e.first == k) return e.second;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, threads, 0) != hipSuccess || occ < 1) occ = -1;
  cache.emplace_back(k, occ);
  return occ;
}
bool grid_fits(const void *k, int threads, long blocks) {
  SeqRuntime &rt = seq_runtime();
  if (!rt.ok || !k) return false;
  const int occ = cached_occupancy(k, threads);
  if (occ < 1) return false;
  return blocks <= (long)rt.num_cu * (occ >= 2 ? occ - 1 : 1);
}
// Streams per chain a launch with these arguments uses (half chains of 4 streams, two workgroups per CU, were measured level with the chains
// of 8 and are gone: DESIGN 4a)
int chain_streams_for(const aslp_lstm_seq *, bool) { return kChainStreams; }

}  // namespace
bool device_shared() { return device_gate().on; }   // (scratch.h)
}  // namespace aslp

using namespace aslp;

extern "C" {

int aslp_lstm_seq_supported(const aslp_lstm_seq *a, int backward) {
  static const bool disabled = getenv("ASLP_LSTM_PERSISTENT") != nullptr && getenv("ASLP_LSTM_PERSISTENT")[0] == '0';
  if (disabled || !seq_args_ok(a)) return 0;
  if (a->s_begin < 0 || a->s_count < 0 || a->s_begin + a->s_count > a->S) return 0;
  const int ns = a->s_count > 0 ? a->s_count : a->S;   // streams of this launch
  const int nsg = (ns + kChainStreams - 1) / kChainStreams, wpc = (a->C + kCellsPerWg - 1) / kCellsPerWg;
  if (a->ndir * nsg > kMaxChains || wpc > kMaxWgPerChain) return 0;   // <= 32 streams per launch (bidirectional) / 64, C <= 512
  const void *k = backward ? reinterpret_cast<const void *>(split_f16_on() ? pick_bwd_h(a->cifg != 0, a->C) : pick_bwd(a->cifg != 0, a->C))
                           : reinterpret_cast<const void *>(split_f16_on() ? pick_fwd_h(a->cifg != 0, a->C) : pick_fwd(a->cifg != 0, a->C));
  return grid_fits(k, 512, (long)kMaxChains * wpc) ? 1 : 0;
}

void aslp_device_shared(int on) { device_gate().on = on != 0; }
void aslp_lstm_split16(int on) { g_lstm_split_override = on < 0 ? -1 : (on != 0); }

int aslp_lstm_seq_first_product_supported(int k_first) { return k_first > 0 && k_first <= kFirstK && (k_first & 3) == 0; }
// ... for a layer of C cells: the kernels stage r(0) in LDS rows as long as the K range of their instantiation (128 floats for C <= 128, else 512)
int aslp_lstm_seq_first_product_supported_for(int k_first, int C) {
  return aslp_lstm_seq_first_product_supported(k_first) && k_first <= (C <= 128 ? 128 : 512);
}
int aslp_lstm_seq_chain_streams(const aslp_lstm_seq *a, int backward) { return seq_args_ok(a) ? chain_streams_for(a, backward != 0) : kChainStreams; }

void aslp_lstm_seq_fill(float *buf, int ld, int T, int S, int col0, int ncols) {
  if (!buf || T <= 0 || S <= 0 || ld <= 0) return;
  if ((ld & 3) || (col0 & 3) || (ncols & 3) || col0 < 0 || col0 + ncols > ld || !aligned16(buf)) { set_error("aslp_lstm_seq_fill: 16-byte alignment"); return; }
  hipLaunchKernelGGL(lstm_seq_fill_kernel, dim3((T + 2) * S), dim3(256), 0, cur_stream(), buf, buf, ld, T, S, col0, ncols, nullptr, 0, 0);
  check_launch("aslp_lstm_seq_fill");
}

void aslp_lstm_seq_fill_pair(float *buf0, float *buf1, int ld, int T, int S, int col0, int ncols, const float *init0, int ld_init, int init_cols) {
  if (!buf0 || !buf1 || T <= 0 || S <= 0 || ld <= 0) return;
  if ((ld & 3) || (col0 & 3) || (ncols & 3) || col0 < 0 || col0 + ncols > ld || !aligned16(buf0) || !aligned16(buf1) ||
      (init0 && ((ld_init & 3) || (init_cols & 3) || init_cols > ld || init_cols > ld_init || !aligned16(init0)))) {
    set_error("aslp_lstm_seq_fill_pair: 16-byte alignment");
    return;
  }
  hipLaunchKernelGGL(lstm_seq_fill_kernel, dim3((T + 2) * S, 2), dim3(256), 0, cur_stream(), buf0, buf1, ld, T, S, col0, ncols, init0, ld_init, init_cols);
  check_launch("aslp_lstm_seq_fill_pair");
}

thread_local int t_last_dmax = 0;   // aslp_lstm_seq_last_dmax()
static void launch_seq(const aslp_lstm_seq *a, bool backward, const char *who) {
  if (backward) t_last_dmax = 0;
  if (!seq_args_ok(a) || !aslp_lstm_seq_supported(a, backward ? 1 : 0)) {
    set_error(std::string(who) + ": arguments outside what the persistent kernel supports (check aslp_lstm_seq_supported first)");
    return;
  }
  for (int d = 0; d < a->ndir; d++) {   // the kernels move 16-byte pieces: the real pointers of this launch (seq_args_ok saw the strides)
    const aslp_lstm_seq_dir &q = a->dir[d];
    if (!q.y || !q.w || !aligned16(q.y) || !aligned16(q.w) || (backward && (!q.d || !aligned16(q.d))) ||
        (q.w_first && (!aligned16(q.w_first) || (q.ldw_first & 3) || (q.col_first & 3) || (q.k_first & 3)))) {
      set_error(std::string(who) + ": buffers of direction " + std::to_string(d) + " missing or not 16-byte aligned");
      return;
    }
  }
  SeqRuntime &rt = seq_runtime();
  // One persistent launch at a time per process: the kernels share the placement table, the abort word and the share ring, and two
  // grids that both need every CU must not be half resident beside each other.  Launches from different host threads / streams
  // are therefore chained by an event (no host wait); the common single-stream case costs one event record per launch.
  register_grid_wide_thread();   // (scratch.h: another thread's grid-wide BatchNormalization / planes launches now stand down)
  std::lock_guard<std::mutex> launch_lock(rt.launch_mu);
  SharedDeviceLaunch shared_device;   // ASLP_DEVICE_SHARED=1 only: cross-process lock held until this kernel has completed
  chain_behind_last_launch(rt);
  // Device-side state is self-cleaning: the placement table is epoch-tagged and every share a backward launch publishes is
  // consumed and reset inside that launch.  Only after a launch that gave up (the mapped error word moved) are the abort
  // word and the share ring put back by hand.
  const size_t slot_bytes = sizeof(float) * (size_t)kMaxChains * kMaxWgPerChain * kMaxWgPerChain * 128;
  if (*rt.host_err != rt.err_seen || !rt.ring_ready) {
    rt.err_seen = *rt.host_err;
    ASLP_CHECK_HIP(hipMemsetAsync(rt.abort_flag, 0, 4, cur_stream()));
    ASLP_CHECK_HIP(hipMemsetAsync(rt.inbox, 0xFF, slot_bytes * kRing, cur_stream()));
    rt.ring_ready = true;
  }
  rt.epoch = (rt.epoch + 1u) & 0x0FFFFFFFu;
  if (rt.epoch == 0u) rt.epoch = 1u;
  static const unsigned wave_collect = ((getenv("ASLP_LSTM_WAVE_COLLECT") != nullptr && getenv("ASLP_LSTM_WAVE_COLLECT")[0] == '0') ? 0u : 1u) |   // A/B switches
                                       ((getenv("ASLP_LSTM_READ_AHEAD") != nullptr && getenv("ASLP_LSTM_READ_AHEAD")[0] == '0') ? 0u : 2u);
  SeqStatus st = {rt.abort_flag, rt.host_err_dev, ((rt.timing_mode == 1 && !backward) || (rt.timing_mode == 2 && backward)) ? rt.timing : nullptr,
                  ((rt.timing_mode == 3 && !backward) || (rt.timing_mode == 4 && backward)) ? rt.timing + 8 : nullptr, rt.epoch, wave_collect};
  const int wpc = (a->C + kCellsPerWg - 1) / kCellsPerWg;
  if (!backward) hipLaunchKernelGGL(split_f16_on() ? pick_fwd_h(a->cifg != 0, a->C) : pick_fwd(a->cifg != 0, a->C), dim3(kMaxChains * wpc), dim3(512), 0, cur_stream(), *a, st, rt.place);
  else {
    hipLaunchKernelGGL(split_f16_on() ? pick_bwd_h(a->cifg != 0, a->C) : pick_bwd(a->cifg != 0, a->C), dim3(kMaxChains * wpc), dim3(512), 0, cur_stream(), *a, st, rt.place, rt.inbox);
    // (only lstm_seq_bwd_h forms the per-workgroup maxima of the gate diffs, and only a single launch per pass leaves a complete set)
    if (split_f16_on() && a->s_count == 0 && (a->dmax_parts[0] || a->dmax_parts[1])) t_last_dmax = kMaxChains * wpc;
  }
  note_launch(rt);
  check_launch(who);
}

int aslp_gru_seq_supported(const aslp_gru_seq *a, int backward) {
  static const bool disabled = (getenv("ASLP_LSTM_PERSISTENT") != nullptr && getenv("ASLP_LSTM_PERSISTENT")[0] == '0') ||
                               (getenv("ASLP_GRU_PERSISTENT") != nullptr && getenv("ASLP_GRU_PERSISTENT")[0] == '0');
  if (disabled || !a || a->T <= 0 || a->S <= 0 || a->H <= 0 || (a->H & 3) || (a->ld & 3)) return 0;
  if (a->s_begin < 0 || a->s_count < 0 || a->s_begin + a->s_count > a->S) return 0;
  const int ns = a->s_count > 0 ? a->s_count : a->S;
  const int nsg = (ns + kChainStreams - 1) / kChainStreams, wpc = (a->H + kCellsPerWg - 1) / kCellsPerWg;
  if (nsg > kMaxChains || wpc > kMaxWgPerChain) return 0;   // <= 64 streams per launch, H <= 512
  return grid_fits(reinterpret_cast<const void *>(pick_gru(backward != 0, a->H)), 512, (long)kMaxChains * wpc) ? 1 : 0;
}

static void launch_gru(const aslp_gru_seq *a, bool backward, const char *who) {
  if (!gru_args_ok(a, backward) || !aslp_gru_seq_supported(a, backward ? 1 : 0)) {
    set_error(std::string(who) + ": arguments outside what the persistent kernel supports (check aslp_gru_seq_supported first)");
    return;
  }
  SeqRuntime &rt = seq_runtime();
  register_grid_wide_thread();
  std::lock_guard<std::mutex> launch_lock(rt.launch_mu);   // as launch_seq (the same lock: LSTM and GRU launches share the runtime state)
  SharedDeviceLaunch shared_device;
  chain_behind_last_launch(rt);
  if (*rt.host_err != rt.err_seen) {
    rt.err_seen = *rt.host_err;
    rt.ring_ready = false;   // the LSTM backward's share ring may be half consumed: launch_seq puts it back
    ASLP_CHECK_HIP(hipMemsetAsync(rt.abort_flag, 0, 4, cur_stream()));
  }
  rt.epoch = (rt.epoch + 1u) & 0x0FFFFFFFu;
  if (rt.epoch == 0u) rt.epoch = 1u;
  SeqStatus st = {rt.abort_flag, rt.host_err_dev, nullptr, ((rt.timing_mode == 3 && !backward) || (rt.timing_mode == 4 && backward)) ? rt.timing + 8 : nullptr,
                  rt.epoch, 0u};
  const int wpc = (a->H + kCellsPerWg - 1) / kCellsPerWg;
  hipLaunchKernelGGL(pick_gru(backward, a->H), dim3(kMaxChains * wpc), dim3(512), 0, cur_stream(), *a, st, rt.place);
  note_launch(rt);
  check_launch(who);
}
void aslp_gru_seq_forward(const aslp_gru_seq *a) { launch_gru(a, false, "aslp_gru_seq_forward"); }
void aslp_gru_seq_backward(const aslp_gru_seq *a) { launch_gru(a, true, "aslp_gru_seq_backward"); }

// devtools: phase timing of the forward (enable = 1) or backward (2) kernel, workgroup 0, wave 0; 0 switches it off; out (8 words, may be NULL)
// receives {timesteps, ticks waiting for the readiness sample, full load, MFMA + LDS stores, barrier, epilogue, launches that
// ran L2-local, -} in 10 ns ticks and clears them.  Synchronises.
void aslp_lstm_seq_timing(int enable, unsigned long long *out) {
  SeqRuntime &rt = seq_runtime();
  if (!rt.ok) return;
  (void)hipStreamSynchronize(cur_stream());
  if (out) {
    (void)hipMemcpy(out, rt.timing, 64, hipMemcpyDeviceToHost);
    (void)hipMemset(rt.timing, 0, 64);
  }
  rt.timing_mode = enable;
}
// devtools: with aslp_lstm_seq_timing(3 | 4, NULL) in effect every workgroup of a forward | backward launch records the clock
// (10 ns ticks) at entry and exit; this returns the latest launch's pairs for workgroups 0 .. n-1 (n <= 1024).  Synchronises.
void aslp_lstm_seq_residency(unsigned long long *out, int n, int launches_back) {
  SeqRuntime &rt = seq_runtime();
  if (!rt.ok || !out || n <= 0 || n > 1024 || launches_back < 0 || launches_back > 7) return;
  (void)hipStreamSynchronize(cur_stream());
  (void)hipMemcpy(out, rt.timing + 8 + ((rt.epoch - (unsigned)launches_back) & 7u) * 2048u, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
}
unsigned aslp_lstm_seq_polls(int reset) {
  SeqRuntime &rt = seq_runtime();
  unsigned v = 0;
  if (!rt.ok) return 0;
  (void)hipStreamSynchronize(cur_stream());
  (void)hipMemcpy(&v, rt.abort_flag + 2, 4, hipMemcpyDeviceToHost);
  if (reset) (void)hipMemset(rt.abort_flag + 2, 0, 4);
  return v;
}
static bool fill_vec_grad_args(SeqVecGradArgs &g, const aslp_lstm_seq *a, int dir, float *bias_corr, float *bias, float *peep_i_corr, float *peep_i,
                               float *peep_f_corr, float *peep_f, float *peep_o_corr, float *peep_o, float mmt, float clip, float neg_lr) {
  if (!seq_args_ok(a) || !a->grad_partial || a->grad_ld < a->C || dir < 0 || dir >= a->ndir || !bias_corr || !bias || !peep_f_corr || !peep_f ||
      !peep_o_corr || !peep_o || (!a->cifg && (!peep_i_corr || !peep_i)))
    return false;
  g.partial = a->grad_partial; g.ld = a->grad_ld; g.ndir = a->ndir; g.dir = dir;
  const int cs = chain_streams_for(a, true);
  g.nsg = ((a->s_count > 0 ? a->s_count : a->S) + cs - 1) / cs;
  g.C = a->C; g.cifg = a->cifg; g.mmt = mmt; g.clip = clip; g.neg_lr = neg_lr;
  const int C = a->C;
  // gate order of the buffer: g, i, f, o (cifg: g, f, o)
  const int off_g = 0, off_i = C, off_f = a->cifg ? C : 2 * C, off_o = a->cifg ? 2 * C : 3 * C;
  g.corr[0] = bias_corr + off_g; g.param[0] = bias + off_g;
  g.corr[1] = a->cifg ? nullptr : bias_corr + off_i; g.param[1] = a->cifg ? nullptr : bias + off_i;
  g.corr[2] = bias_corr + off_f; g.param[2] = bias + off_f;
  g.corr[3] = bias_corr + off_o; g.param[3] = bias + off_o;
  g.corr[4] = a->cifg ? nullptr : peep_i_corr; g.param[4] = a->cifg ? nullptr : peep_i;
  g.corr[5] = peep_f_corr; g.param[5] = peep_f;
  g.corr[6] = peep_o_corr; g.param[6] = peep_o;
  return true;
}
void aslp_lstm_seq_vec_grads(const aslp_lstm_seq *a, int dir, float *bias_corr, float *bias, float *peep_i_corr, float *peep_i, float *peep_f_corr,
                             float *peep_f, float *peep_o_corr, float *peep_o, float mmt, float clip, float neg_lr) {
  SeqVecGradPair gp;
  if (!fill_vec_grad_args(gp.d[0], a, dir, bias_corr, bias, peep_i_corr, peep_i, peep_f_corr, peep_f, peep_o_corr, peep_o, mmt, clip, neg_lr)) {
    set_error("aslp_lstm_seq_vec_grads: bad arguments");
    return;
  }
  gp.d[1] = gp.d[0];
  hipLaunchKernelGGL(lstm_seq_vec_grads_kernel, dim3((7 * a->C + 255) / 256, 1), dim3(256), 0, cur_stream(), gp);
  check_launch("aslp_lstm_seq_vec_grads");
}
void aslp_lstm_seq_vec_grads2(const aslp_lstm_seq *a, float *const *vec8_dir0, float *const *vec8_dir1, float mmt, float clip, float neg_lr) {
  SeqVecGradPair gp;
  bool ok = a && a->ndir == 2 && vec8_dir0 && vec8_dir1;
  for (int d = 0; d < 2 && ok; d++) {
    float *const *v = d == 0 ? vec8_dir0 : vec8_dir1;
    ok = fill_vec_grad_args(gp.d[d], a, d, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], mmt, clip, neg_lr);
  }
  if (!ok) { set_error("aslp_lstm_seq_vec_grads2: bad arguments"); return; }
  hipLaunchKernelGGL(lstm_seq_vec_grads_kernel, dim3((7 * a->C + 255) / 256, 2), dim3(256), 0, cur_stream(), gp);
  check_launch("aslp_lstm_seq_vec_grads2");
}
void aslp_lstm_seq_forward(const aslp_lstm_seq *a) { launch_seq(a, false, "aslp_lstm_seq_forward"); }
void aslp_lstm_seq_backward(const aslp_lstm_seq *a) { launch_seq(a, true, "aslp_lstm_seq_backward"); }
int aslp_lstm_seq_last_dmax(void) { return t_last_dmax; }

}  // extern "C"
