// cu-matrix.cpp -- see cu-matrix.h.  Every method maps to one kernel of csrc/ (no CPU branch).
#include "cu-matrix.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>

#include "common.h"
#include "split16.h"

namespace aslp {


static const aslp_dim3 kD3 = {1, 1, 1};

// ---- caching device allocator ----------------------------------------------------------------
namespace {
std::mutex g_alloc_mu;
std::multimap<size_t, void *> g_free;   // size -> block
std::map<void *, size_t> g_live;        // block -> size
size_t RoundSize(size_t b) { return (b + 255) & ~(size_t)255; }
void CheckHip(hipError_t e, const char *what) {
  if (e != hipSuccess) ASLP_ERR << what << ": " << hipGetErrorString(e);
}
void CheckKernels() {
  char buf[512];
  if (aslp_get_last_error(buf, sizeof(buf))) ASLP_ERR << buf;
}
}  // namespace

// New blocks of up to 32 MB are carved out of 256 MB slabs instead of being hipMalloc'ed one by one: the first training step of a net asks
// for a hundred buffers, and a hundred hipMalloc calls stood as ~15 ms of idle GPU at the head of every tool run (on a 288 GB device
// the slack of a slab is nothing).  Freed pieces go to the size-keyed cache like every other block.  ASLP_ALLOC_SLAB=0: one hipMalloc
// per new block, as before.
namespace {
constexpr size_t kSlabBytes = 256u << 20, kSlabMaxRequest = 32u << 20;
struct Slab { char *base; size_t used; };
std::vector<Slab> g_slabs;
bool InSlab(const void *p) {
  for (const Slab &s : g_slabs)
    if (static_cast<const char *>(p) >= s.base && static_cast<const char *>(p) < s.base + kSlabBytes) return true;
  return false;
}
void *SlabCarve(size_t sz) {   // caller holds g_alloc_mu; NULL: not served (too large, switched off, no memory for a slab)
  static const bool on = !(getenv("ASLP_ALLOC_SLAB") != nullptr && getenv("ASLP_ALLOC_SLAB")[0] == '0');
  if (!on || sz > kSlabMaxRequest) return nullptr;
  if (g_slabs.empty() || g_slabs.back().used + sz > kSlabBytes) {
    void *base = nullptr;
    if (hipMalloc(&base, kSlabBytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    g_slabs.push_back(Slab{static_cast<char *>(base), 0});
  }
  Slab &s = g_slabs.back();
  void *p = s.base + s.used;
  s.used += sz;
  return p;
}
}  // namespace

void *DeviceAlloc(size_t bytes) {
  if (bytes == 0) return nullptr;
  size_t sz = RoundSize(bytes);
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  auto it = g_free.find(sz);
  void *p = nullptr;
  if (it != g_free.end()) {
    p = it->second;
    g_free.erase(it);
  } else {
    p = SlabCarve(sz);
    if (!p) {
      hipError_t e = hipMalloc(&p, sz);
      if (e != hipSuccess) {
        // release the cache (the blocks that are allocations of their own: a piece of a slab cannot go back alone) and retry once
        for (auto it2 = g_free.begin(); it2 != g_free.end();) {
          if (InSlab(it2->second)) { ++it2; continue; }
          (void)hipFree(it2->second);
          it2 = g_free.erase(it2);
        }
        CheckHip(hipMalloc(&p, sz), "hipMalloc");
      }
    }
  }
  g_live[p] = sz;
  // debugging aid: ASLP_ALLOC_POISON=1 hands every block out filled with 0xFF bytes (a NaN in every float), so that a kernel which reads
  // memory nobody wrote shows in a fresh process and not only when the allocator happens to recycle a dirty block
  static const bool poison = getenv("ASLP_ALLOC_POISON") != nullptr && getenv("ASLP_ALLOC_POISON")[0] == '1';
  if (poison) CheckHip(hipMemsetAsync(p, 0xFF, sz, cur_stream()), "hipMemset (poison)");
  return p;
}
void DeviceFree(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  auto it = g_live.find(p);
  if (it == g_live.end()) return;
  // Kernels are stream-ordered: a block handed out again is only touched by later launches.
  g_free.insert({it->second, p});
  g_live.erase(it);
}
void DeviceToHost(void *dst, const void *src, size_t bytes) {
  if (!bytes) return;
  CheckHip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cur_stream()), "hipMemcpy D2H");
  CheckHip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
}
// Host -> device without draining the stream: the bytes are copied into a ring of pinned staging slots and sent with
// hipMemcpyAsync; `src` is free again on return, and the only wait is for the copy that used the same slot a whole ring
// ago (long finished).  A stream synchronise per upload -- labels, frame weights, every utterance of a cache fill -- would
// empty the launch queue each time and leave the GPU idle while the host refills it (measured on the tool path: 546 k
// -> see DESIGN.md frames/s end to end).
// Two rings: a minibatch's labels and frame weights are a few KB each and go out every step, so a ring of eight slots let the host run at most
// FOUR steps (3 ms) ahead of the GPU -- every longer piece of host work (a randomizer refill: 3.6 ms; the reader's hand-over) then reached the
// device as idle time, 140 ms per million frames of the cfg2 tool run.  Uploads of up to 64 KB have 512 slots of their own (the host may
// now lead by a quarter of a thousand steps; the launch queue is what bounds it), the large ones keep the 4 MB slots.
namespace {
constexpr size_t kStageBytes = 4u << 20, kSmallBytes = 64u << 10;
constexpr int kStageSlots = 8, kSmallSlots = 512;
struct StageSlot { void *pinned = nullptr; hipEvent_t ev = nullptr; bool pending = false; };
StageSlot g_stage[kStageSlots], g_small[kSmallSlots];
int g_stage_next = 0, g_small_next = 0;
char *g_small_block = nullptr;   // one page-locked block carved into the small slots
std::mutex g_stage_mu;
}  // namespace
// next free staging slot (caller holds g_stage_mu)
static StageSlot &TakeStageSlot(bool small = false) {
  StageSlot &slot = small ? g_small[g_small_next] : g_stage[g_stage_next];
  if (small) g_small_next = (g_small_next + 1) % kSmallSlots;
  else g_stage_next = (g_stage_next + 1) % kStageSlots;
  if (!slot.pinned) {
    if (small) {
      if (!g_small_block) CheckHip(hipHostMalloc(reinterpret_cast<void **>(&g_small_block), kSmallBytes * kSmallSlots, hipHostMallocDefault), "hipHostMalloc(staging)");
      slot.pinned = g_small_block + kSmallBytes * (size_t)(&slot - g_small);
    } else {
      CheckHip(hipHostMalloc(&slot.pinned, kStageBytes, hipHostMallocDefault), "hipHostMalloc(staging)");
    }
    CheckHip(hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming), "hipEventCreate(staging)");
  }
  if (slot.pending) { CheckHip(hipEventSynchronize(slot.ev), "hipEventSynchronize(staging)"); slot.pending = false; }
  return slot;
}
void HostToDevice(void *dst, const void *src, size_t bytes) {
  if (!bytes) return;
  std::lock_guard<std::mutex> lk(g_stage_mu);
  const char *s = static_cast<const char *>(src);
  char *d = static_cast<char *>(dst);
  while (bytes > 0) {
    StageSlot &slot = TakeStageSlot(bytes <= kSmallBytes);
    const size_t n = bytes < kStageBytes ? bytes : kStageBytes;
    std::memcpy(slot.pinned, s, n);
    CheckHip(hipMemcpyAsync(d, slot.pinned, n, hipMemcpyHostToDevice, cur_stream()), "hipMemcpy H2D");
    CheckHip(hipEventRecord(slot.ev, cur_stream()), "hipEventRecord(staging)");
    slot.pending = true;
    s += n; d += n; bytes -= n;
  }
}
// rows of `cols` floats at host pitch `ld` -> device rows at pitch `stride`, through the staging ring
static void HostToDevice2D(float *dst, int stride, const float *src, int ld, int rows, int cols) {
  if (rows <= 0 || cols <= 0) return;
  std::lock_guard<std::mutex> lk(g_stage_mu);
  const size_t row_bytes = sizeof(float) * (size_t)cols;
  int rows_per_slot = (int)(kStageBytes / row_bytes);
  if (rows_per_slot < 1) ASLP_ERR << "matrix row of " << cols << " floats exceeds the staging slot";
  for (int r0 = 0; r0 < rows; r0 += rows_per_slot) {
    const int nr = rows - r0 < rows_per_slot ? rows - r0 : rows_per_slot;
    StageSlot &slot = TakeStageSlot();
    if (ld == cols) std::memcpy(slot.pinned, src + (size_t)r0 * ld, row_bytes * nr);
    else for (int r = 0; r < nr; r++) std::memcpy(static_cast<char *>(slot.pinned) + row_bytes * r, src + (size_t)(r0 + r) * ld, row_bytes);
    if (stride == cols)
      CheckHip(hipMemcpyAsync(dst + (size_t)r0 * stride, slot.pinned, row_bytes * nr, hipMemcpyHostToDevice, cur_stream()), "hipMemcpy H2D");
    else
      CheckHip(hipMemcpy2DAsync(dst + (size_t)r0 * stride, sizeof(float) * stride, slot.pinned, row_bytes, row_bytes, nr, hipMemcpyHostToDevice,
                                cur_stream()), "hipMemcpy2D H2D");
    CheckHip(hipEventRecord(slot.ev, cur_stream()), "hipEventRecord(staging)");
    slot.pending = true;
  }
}
void DeviceToDevice(void *dst, const void *src, size_t bytes) {
  if (!bytes) return;
  CheckHip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, cur_stream()), "hipMemcpy D2D");
}
void DeviceMemset(void *dst, int v, size_t bytes) {
  if (!bytes) return;
  CheckHip(hipMemsetAsync(dst, v, bytes, cur_stream()), "hipMemset");
}
// Page-locked blocks are cached for the life of the process: hipHostMalloc / hipHostFree cost a fraction of a millisecond
// per call plus the (un)pinning of every page, and a cache fill of the frame tools cycles through a hundred blocks.  Sizes
// are rounded to 1 MiB so the handful of distinct utterance lengths share blocks.  PinnedPoolRelease() gives them back.
namespace {
std::mutex g_pin_mu;
std::multimap<size_t, void *> g_pin_free;
std::map<void *, size_t> g_pin_live;
}  // namespace
void *PinnedAlloc(size_t bytes) {
  const size_t sz = (std::max<size_t>(bytes, 1) + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    auto it = g_pin_free.find(sz);
    if (it != g_pin_free.end()) {
      void *p = it->second;
      g_pin_free.erase(it);
      g_pin_live[p] = sz;
      return p;
    }
  }
  void *p = nullptr;
  CheckHip(hipHostMalloc(&p, sz, hipHostMallocDefault), "hipHostMalloc");
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pin_live[p] = sz;
  return p;
}
void PinnedFree(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pin_live.find(p);
  if (it == g_pin_live.end()) return;
  g_pin_free.insert({it->second, p});
  g_pin_live.erase(it);
}
void PinnedPoolRelease() {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  for (auto &kv : g_pin_free) (void)hipHostFree(kv.second);
  g_pin_free.clear();
}
StreamMarker::StreamMarker() : ev_(nullptr), recorded_(false) {
  hipEvent_t e;
  CheckHip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
  ev_ = e;
}
StreamMarker::~StreamMarker() { if (ev_) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_)); }
void StreamMarker::Record() { CheckHip(hipEventRecord(static_cast<hipEvent_t>(ev_), cur_stream()), "hipEventRecord"); recorded_ = true; }
bool StreamMarker::Done() const { return !recorded_ || hipEventQuery(static_cast<hipEvent_t>(ev_)) == hipSuccess; }
void StreamMarker::Wait() const { if (recorded_) CheckHip(hipEventSynchronize(static_cast<hipEvent_t>(ev_)), "hipEventSynchronize"); }
CopyLane::CopyLane() : stream_(nullptr), ev_(nullptr) {
  hipStream_t s;
  hipEvent_t e;
  CheckHip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");  // non-blocking: no implicit ordering with the null stream
  CheckHip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
  stream_ = s;
  ev_ = e;
}
CopyLane::~CopyLane() {
  if (stream_) { (void)hipStreamSynchronize(static_cast<hipStream_t>(stream_)); (void)hipStreamDestroy(static_cast<hipStream_t>(stream_)); }
  if (ev_) (void)hipEventDestroy(static_cast<hipEvent_t>(ev_));
}
void CopyLane::Upload(float *dst, int dst_stride, const float *src, int ld, int rows, int cols) {
  if (!rows) return;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  if (dst_stride == cols && ld == cols)
    CheckHip(hipMemcpyAsync(dst, src, sizeof(float) * (size_t)rows * cols, hipMemcpyHostToDevice, s), "hipMemcpy H2D (lane)");
  else
    CheckHip(hipMemcpy2DAsync(dst, sizeof(float) * dst_stride, src, sizeof(float) * ld, sizeof(float) * cols, rows, hipMemcpyHostToDevice, s),
             "hipMemcpy2D H2D (lane)");
}
void CopyLane::Zero(void *dst, size_t bytes) {
  if (bytes) CheckHip(hipMemsetAsync(dst, 0, bytes, static_cast<hipStream_t>(stream_)), "hipMemset (lane)");
}
void CopyLane::Record(StreamMarker *m) {
  CheckHip(hipEventRecord(static_cast<hipEvent_t>(m->ev_), static_cast<hipStream_t>(stream_)), "hipEventRecord (lane)");
  m->recorded_ = true;
}
void CopyLane::LaneWaitsForStream() {
  CheckHip(hipEventRecord(static_cast<hipEvent_t>(ev_), cur_stream()), "hipEventRecord");
  CheckHip(hipStreamWaitEvent(static_cast<hipStream_t>(stream_), static_cast<hipEvent_t>(ev_), 0), "hipStreamWaitEvent");
}
void CopyLane::StreamWaitsForLane() {
  CheckHip(hipEventRecord(static_cast<hipEvent_t>(ev_), static_cast<hipStream_t>(stream_)), "hipEventRecord (lane)");
  CheckHip(hipStreamWaitEvent(cur_stream(), static_cast<hipEvent_t>(ev_), 0), "hipStreamWaitEvent");
}
void CopyLane::Sync() { CheckHip(hipStreamSynchronize(static_cast<hipStream_t>(stream_)), "hipStreamSynchronize (lane)"); }
void StreamSync() {
  CheckHip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
  CheckKernels();
}

void WarmUpDevice() {
  CuMatrix m(2, 2);
  m.Set(1.0f);
  CuVector v(2);
  v.AddRowSumMat(1.0f, m, 0.0f);
  (void)v.Sum();
  StreamSync();
}

// ---- CuVector ---------------------------------------------------------------------------------------
void CuVector::Resize(int dim, MatrixResizeType t) {
  if (dim != dim_) {
    DeviceFree(data_);
    data_ = dim ? static_cast<float *>(DeviceAlloc(sizeof(float) * dim)) : nullptr;
    dim_ = dim;
  }
  if (t == kSetZero) SetZero();
}
void CuVectorBase::SetZero() { DeviceMemset(data_, 0, sizeof(float) * dim_); }
void CuVectorBase::Set(float v) { if (dim_) { cudaF_set_const(kD3, kD3, data_, v, AsRow()); CheckKernels(); } }
void CuVectorBase::Add(float v) { if (dim_) { cudaF_add(kD3, kD3, data_, v, AsRow()); CheckKernels(); } }
void CuVectorBase::Scale(float v) { if (dim_) { cudaF_scale(kD3, kD3, data_, v, AsRow()); CheckKernels(); } }
void CuVectorBase::CopyFromVec(const CuVectorBase &v) {
  ASLP_ASSERT(v.Dim() == dim_);
  DeviceToDevice(data_, v.Data(), sizeof(float) * dim_);
}
void CuVectorBase::CopyFromHost(const float *src, int n) { ASLP_ASSERT(n == dim_); HostToDevice(data_, src, sizeof(float) * n); }
void CuVectorBase::CopyToHost(float *dst) const { DeviceToHost(dst, data_, sizeof(float) * dim_); }
void CuVectorBase::AddVec(float alpha, const CuVectorBase &v, float beta) {
  ASLP_ASSERT(v.Dim() == dim_);
  if (!dim_) return;
  if (beta != 1.0f) cudaF_scale(kD3, kD3, data_, beta, AsRow());
  aslp_vec_axpy(alpha, v.Data(), data_, dim_);
  CheckKernels();
}
void CuVectorBase::AddRowSumMat(float alpha, const CuMatrixBase &M, float beta) {
  ASLP_ASSERT(M.NumCols() == dim_);
  if (!dim_) return;
  aslp_add_row_sum_mat_vec(alpha, M.Data(), M.Dim(), beta, data_);
  CheckKernels();
}
void CuVectorBase::AddColSumMat(float alpha, const CuMatrixBase &M, float beta) {
  ASLP_ASSERT(M.NumRows() == dim_);
  if (!dim_) return;
  aslp_add_col_sum_mat_vec(alpha, M.Data(), M.Dim(), beta, data_);
  CheckKernels();
}
void CuVectorBase::AddDiagMatMat(float alpha, const CuMatrixBase &M, MatrixTransposeType tM, const CuMatrixBase &N,
                                 MatrixTransposeType tN, float beta) {
  // cu-vector.cc:540-590
  int M_col_dim = (tM == kTrans ? M.NumRows() : M.NumCols()), N_row_dim = (tN == kTrans ? N.NumCols() : N.NumRows());
  ASLP_ASSERT(M_col_dim == N_row_dim);
  int M_row_stride = M.Stride(), M_col_stride = 1;
  if (tM == kTrans) std::swap(M_row_stride, M_col_stride);
  int N_row_stride = N.Stride(), N_col_stride = 1;
  if (tN == kTrans) std::swap(N_row_stride, N_col_stride);
  if (!dim_) return;
  cudaF_add_diag_mat_mat(0, 0, alpha, data_, dim_, M.Data(), M_col_dim, M_row_stride, M_col_stride, N.Data(), N_row_stride,
                         N_col_stride, 1, beta);
  CheckKernels();
}
void CuVectorBase::AddVecVec(float alpha, const CuVectorBase &x, const CuVectorBase &y, float beta) {
  ASLP_ASSERT(x.Dim() == dim_ && y.Dim() == dim_);
  cudaF_add_vec_vec(0, 0, alpha, data_, x.Data(), y.Data(), beta, dim_);
  CheckKernels();
}
void CuVectorBase::MulElements(const CuVectorBase &v) {
  ASLP_ASSERT(v.Dim() == dim_);
  if (dim_) { cudaF_mul_elements(kD3, kD3, data_, v.Data(), AsRow(), dim_); CheckKernels(); }
}
void CuVectorBase::ApplyFloor(float v) { if (dim_) { cudaF_apply_floor(kD3, kD3, data_, v, AsRow()); CheckKernels(); } }
void CuVectorBase::ApplyCeiling(float v) { if (dim_) { cudaF_apply_ceiling(kD3, kD3, data_, v, AsRow()); CheckKernels(); } }
void CuVectorBase::ApplyPow(float p) { if (dim_) { cudaF_apply_pow(kD3, kD3, data_, p, AsRow()); CheckKernels(); } }
void CuVectorBase::InvertElements() { if (dim_) { cudaF_invert_elements(kD3, kD3, data_, AsRow()); CheckKernels(); } }
float CuVectorBase::Sum() const {
  if (!dim_) return 0.0f;
  std::vector<float> h(dim_);
  CopyToHost(h.data());
  double s = 0.0;
  for (float x : h) s += x;
  return (float)s;
}
CuSubVector CuVectorBase::Range(int o, int n) {
  ASLP_ASSERT(o >= 0 && n >= 0 && o + n <= dim_);
  return CuSubVector(data_ + o, n);
}

// ---- CuVectorD ---------------------------------------------------------------------------------------
CuVectorD &CuVectorD::operator=(const CuVectorD &o) {
  if (this != &o) {
    Resize(o.dim_, kUndefined);
    DeviceToDevice(data_, o.data_, sizeof(double) * dim_);
  }
  return *this;
}
void CuVectorD::Resize(int dim, MatrixResizeType t) {
  if (dim != dim_) {
    DeviceFree(data_);
    data_ = dim ? static_cast<double *>(DeviceAlloc(sizeof(double) * dim)) : nullptr;
    dim_ = dim;
  }
  if (t == kSetZero) SetZero();
}
void CuVectorD::SetZero() { DeviceMemset(data_, 0, sizeof(double) * dim_); }
void CuVectorD::CopyFromHost(const double *src, int n) { ASLP_ASSERT(n == dim_); HostToDevice(data_, src, sizeof(double) * n); }
void CuVectorD::CopyToHost(double *dst) const { DeviceToHost(dst, data_, sizeof(double) * dim_); }
void CuVectorD::Read(std::istream &is, bool binary) {
  HostVectorD v;
  v.Read(is, binary);
  Resize(v.Dim(), kUndefined);
  CopyFromHost(v.data.data(), v.Dim());
}
void CuVectorD::Write(std::ostream &os, bool binary) const {
  HostVectorD v;
  v.data.resize(dim_);
  CopyToHost(v.data.data());
  v.Write(os, binary);
}

// ---- CuMatrix ------------------------------------------------------------------------------------------
void CuMatrix::Resize(int rows, int cols, MatrixResizeType t) {
  ASLP_ASSERT(rows >= 0 && cols >= 0);
  if (rows * cols == 0) rows = cols = 0;
  if (rows != rows_ || cols != cols_) {
    DeviceFree(data_);
    data_ = nullptr;
    stride_ = PaddedStride(cols);
    if (rows) data_ = static_cast<float *>(DeviceAlloc(sizeof(float) * (size_t)rows * stride_));
    rows_ = rows;
    cols_ = cols;
  }
  if (t == kSetZero) SetZero();
}
CuSubMatrix CuMatrixBase::Range(int r0, int nr, int c0, int nc) const { return CuSubMatrix(*this, r0, nr, c0, nc); }
CuSubMatrix CuMatrixBase::RowRange(int r0, int nr) const { return CuSubMatrix(*this, r0, nr, 0, cols_); }
CuSubMatrix CuMatrixBase::ColRange(int c0, int nc) const { return CuSubMatrix(*this, 0, rows_, c0, nc); }
CuSubVector CuMatrixBase::Row(int r) { ASLP_ASSERT(r >= 0 && r < rows_); return CuSubVector(RowData(r), cols_); }

void CuMatrixBase::SetZero() {
  if (!rows_) return;
  // contiguous (owning) matrices: one memset incl. padding; views: strided kernel
  if (stride_ == PaddedStride(cols_) || rows_ == 1) DeviceMemset(data_, 0, sizeof(float) * ((size_t)(rows_ - 1) * stride_ + cols_));
  else { cudaF_set_const(kD3, kD3, data_, 0.0f, Dim()); CheckKernels(); }
}
#define ASLP_EW(call) do { if (rows_ && cols_) { call; CheckKernels(); } } while (0)
void CuMatrixBase::Set(float v) { ASLP_EW(cudaF_set_const(kD3, kD3, data_, v, Dim())); }
void CuMatrixBase::Add(float v) { ASLP_EW(cudaF_add(kD3, kD3, data_, v, Dim())); }
void CuMatrixBase::Scale(float v) { ASLP_EW(cudaF_scale(kD3, kD3, data_, v, Dim())); }
void CuMatrixBase::ApplyFloor(float v) { ASLP_EW(cudaF_apply_floor(kD3, kD3, data_, v, Dim())); }
void CuMatrixBase::ApplyCeiling(float v) { ASLP_EW(cudaF_apply_ceiling(kD3, kD3, data_, v, Dim())); }
void CuMatrixBase::ApplyPow(float p) { ASLP_EW(cudaF_apply_pow(kD3, kD3, data_, p, Dim())); }
void CuMatrixBase::ApplyLog() { ASLP_EW(cudaF_apply_log(kD3, kD3, data_, Dim())); }
void CuMatrixBase::ApplyExp() { ASLP_EW(cudaF_apply_exp(kD3, kD3, data_, Dim())); }
void CuMatrixBase::ApplyHeaviside() { ASLP_EW(cudaF_apply_heaviside(kD3, kD3, data_, Dim())); }
void CuMatrixBase::InvertElements() { ASLP_EW(cudaF_invert_elements(kD3, kD3, data_, Dim())); }

void CuMatrixBase::CopyFromMat(const CuMatrixBase &src) {
  ASLP_ASSERT(SameDim(*this, src));
  if (!rows_) return;
  if (src.Data() == data_ && src.Stride() == stride_) return;
  aslp_copy_mat(data_, Dim(), src.Data(), src.Stride());
  CheckKernels();
}
void CuMatrixBase::CopyFromHost(const float *src, int ld) {
  if (!rows_) return;
  HostToDevice2D(data_, stride_, src, ld, rows_, cols_);
}
void CuMatrixBase::CopyFromPinnedHost(const float *src, int ld) {
  if (!rows_) return;
  if (stride_ == cols_ && ld == cols_)
    CheckHip(hipMemcpyAsync(data_, src, sizeof(float) * (size_t)rows_ * cols_, hipMemcpyHostToDevice, cur_stream()), "hipMemcpy H2D");
  else
    CheckHip(hipMemcpy2DAsync(data_, sizeof(float) * stride_, src, sizeof(float) * ld, sizeof(float) * cols_, rows_, hipMemcpyHostToDevice,
                              cur_stream()), "hipMemcpy2D H2D");
}
void CuMatrixBase::CopyToHost(float *dst, int ld) const {
  if (!rows_) return;
  CheckHip(hipMemcpy2DAsync(dst, sizeof(float) * ld, data_, sizeof(float) * stride_, sizeof(float) * cols_, rows_,
                            hipMemcpyDeviceToHost, cur_stream()), "hipMemcpy2D D2H");
  CheckHip(hipStreamSynchronize(cur_stream()), "hipStreamSynchronize");
}
void CuMatrixBase::CopyFromMat(const HostMatrix &m) {
  ASLP_ASSERT(m.rows == rows_ && m.cols == cols_);
  CopyFromHost(m.data.data(), m.cols);
}
// Kaldi binary matrix ("FM", rows, cols, packed rows -- what HostMatrix::Write emits) straight from HBM: packed row blocks
// come down into two page-locked buffers in turn, the next block is in flight while the previous one is written to the stream.
// The pageable-memory detour of CopyToMat() (a zero-filled std::vector, a staged pageable copy) cost more than the file write.
void CuMatrixBase::WriteBinary(std::ostream &os) const {
  if (!os.good()) ASLP_ERR << "Failed to write matrix to stream: stream not good";
  WriteToken(os, true, "FM");
  WriteBasicType(os, true, (int32)rows_);
  WriteBasicType(os, true, (int32)cols_);
  if (rows_ == 0 || cols_ == 0) return;
  const size_t row_bytes = sizeof(float) * (size_t)cols_;
  const size_t kBuf = 8u << 20;
  const int rows_per = (int)std::max<size_t>(1, kBuf / row_bytes);
  void *buf[2] = {PinnedAlloc(std::max(kBuf, row_bytes)), PinnedAlloc(std::max(kBuf, row_bytes))};
  StreamMarker done[2];
  auto fetch = [&](int r0, int k) {
    const int nr = std::min(rows_per, rows_ - r0);
    CheckHip(hipMemcpy2DAsync(buf[k], row_bytes, data_ + (size_t)r0 * stride_, sizeof(float) * stride_, row_bytes, nr, hipMemcpyDeviceToHost,
                              cur_stream()), "hipMemcpy2D D2H");
    done[k].Record();
  };
  fetch(0, 0);
  int k = 0;
  for (int r0 = 0; r0 < rows_; r0 += rows_per, k ^= 1) {
    if (r0 + rows_per < rows_) fetch(r0 + rows_per, k ^ 1);
    done[k].Wait();
    os.write(static_cast<const char *>(buf[k]), row_bytes * std::min(rows_per, rows_ - r0));
  }
  PinnedFree(buf[0]);
  PinnedFree(buf[1]);
  if (!os.good()) ASLP_ERR << "Failed to write matrix to stream";
}
void CuMatrixBase::CopyToMat(HostMatrix *m) const {
  m->Resize(rows_, cols_);
  CopyToHost(m->data.data(), cols_);
}
void CuMatrixBase::CopyFromMatTrans(const CuMatrixBase &src) {
  ASLP_ASSERT(src.NumRows() == cols_ && src.NumCols() == rows_);
  aslp_copy_mat_trans(data_, Dim(), src.Data(), src.Stride());
  CheckKernels();
}
void CuMatrixBase::AddMat(float alpha, const CuMatrixBase &A, MatrixTransposeType tA) {
  if (tA == kNoTrans) ASLP_ASSERT(SameDim(*this, A));
  else ASLP_ASSERT(A.NumRows() == cols_ && A.NumCols() == rows_);
  ASLP_EW(cudaF_add_mat(kD3, kD3, alpha, A.Data(), data_, Dim(), A.Stride(), tA == kTrans ? 1 : 0));
}
void CuMatrixBase::AddMatMat(float alpha, const CuMatrixBase &A, MatrixTransposeType tA, const CuMatrixBase &B, MatrixTransposeType tB,
                             float beta, const aslp_gemm_epilogue *ep) {
  AddMatMat(alpha, A, tA, B, tB, beta, ep, nullptr, nullptr);
}
void CuMatrixBase::AddMatMat(float alpha, const CuMatrixBase &A, MatrixTransposeType tA, const CuMatrixBase &B, MatrixTransposeType tB,
                             float beta, const aslp_gemm_epilogue *ep, const PlaneSet *pa, const PlaneSet *pb) {
  // cu-matrix.cc:1027-1046 dimension checks
  int m = (tB == kTrans ? B.NumRows() : B.NumCols());
  int n = (tA == kTrans ? A.NumCols() : A.NumRows());
  int k = (tB == kTrans ? B.NumCols() : B.NumRows());
  int k1 = (tA == kTrans ? A.NumRows() : A.NumCols());
  ASLP_ASSERT(m == NumCols());
  ASLP_ASSERT(n == NumRows());
  ASLP_ASSERT(k == k1);
  if (m == 0) return;
  int rc = aslp_sgemm_planes_ex(tA == kTrans, tB == kTrans, rows_, cols_, k, alpha, A.Data(), A.Stride(), reinterpret_cast<const aslp_planes *>(pa),
                                B.Data(), B.Stride(), reinterpret_cast<const aslp_planes *>(pb), beta, data_, stride_, ep);
  if (rc != 0) ASLP_ERR << "aslp_sgemm argument error " << rc;
  CheckKernels();
}
void AddMatMatPair(CuMatrixBase &C0, CuMatrixBase &C1, float alpha, const CuMatrixBase &A0, const CuMatrixBase &A1, MatrixTransposeType tA,
                   const CuMatrixBase &B0, const CuMatrixBase &B1, MatrixTransposeType tB, float beta, const aslp_gemm_epilogue *ep0,
                   const aslp_gemm_epilogue *ep1) {
  AddMatMatPair(C0, C1, alpha, A0, A1, tA, B0, B1, tB, beta, ep0, ep1, nullptr);
}
void AddMatMatPair(CuMatrixBase &C0, CuMatrixBase &C1, float alpha, const CuMatrixBase &A0, const CuMatrixBase &A1, MatrixTransposeType tA,
                   const CuMatrixBase &B0, const CuMatrixBase &B1, MatrixTransposeType tB, float beta, const aslp_gemm_epilogue *ep0,
                   const aslp_gemm_epilogue *ep1, const S16View *views) {
  const bool same = SameDim(C0, C1) && SameDim(A0, A1) && SameDim(B0, B1) && C0.Stride() == C1.Stride() && A0.Stride() == A1.Stride() &&
                    B0.Stride() == B1.Stride();
  if (!same || C0.NumCols() == 0) {
    C0.AddMatMat(alpha, A0, tA, B0, tB, beta, ep0);
    C1.AddMatMat(alpha, A1, tA, B1, tB, beta, ep1);
    return;
  }
  const int m = (tB == kTrans ? B0.NumRows() : B0.NumCols()), n = (tA == kTrans ? A0.NumCols() : A0.NumRows());
  const int k = (tB == kTrans ? B0.NumCols() : B0.NumRows()), k1 = (tA == kTrans ? A0.NumRows() : A0.NumCols());
  ASLP_ASSERT(m == C0.NumCols() && n == C0.NumRows() && k == k1);
  const int rc = sgemm_pair_views(tA == kTrans, tB == kTrans, C0.NumRows(), C0.NumCols(), k, alpha, A0.Data(), A1.Data(), A0.Stride(), B0.Data(),
                                  B1.Data(), B0.Stride(), beta, C0.Data(), C1.Data(), C0.Stride(), ep0, ep1, views ? views + 0 : nullptr,
                                  views ? views + 1 : nullptr, views ? views + 2 : nullptr, views ? views + 3 : nullptr);
  if (rc != 0) ASLP_ERR << "aslp_sgemm_pair argument error " << rc;
  CheckKernels();
}
void CuMatrixBase::AddVecToRows(float alpha, const CuVectorBase &row, float beta) {
  if (row.Dim() != NumCols()) ASLP_ERR << "Non matching dimensions: Cols:" << NumCols() << " VectorDim:" << row.Dim();
  ASLP_EW(cudaF_add_vec_to_rows(kD3, kD3, alpha, row.Data(), beta, data_, Dim()));
}
void CuMatrixBase::AddVecToCols(float alpha, const CuVectorBase &col, float beta) {
  if (col.Dim() != NumRows()) ASLP_ERR << "Non matching dimensions: Rows:" << NumRows() << " VectorDim:" << col.Dim();
  ASLP_EW(cudaF_add_vec_to_cols(kD3, kD3, alpha, col.Data(), beta, data_, Dim()));
}
void CuMatrixBase::AddMatMatElements(float alpha, const CuMatrixBase &A, const CuMatrixBase &B, float beta) {
  ASLP_ASSERT(SameDim(*this, A) && SameDim(A, B));
  ASLP_EW(cudaF_add_mat_mat_elements(kD3, kD3, data_, A.Data(), B.Data(), Dim(), A.Stride(), B.Stride(), alpha, beta));
}
void CuMatrixBase::AddMatDiagVec(float alpha, const CuMatrixBase &M, MatrixTransposeType tM, const CuVectorBase &v, float beta) {
  if (tM == kNoTrans) ASLP_ASSERT(SameDim(*this, M));
  else ASLP_ASSERT(M.NumRows() == NumCols() && M.NumCols() == NumRows());
  ASLP_ASSERT(v.Dim() == NumCols());
  int rs = M.Stride(), cs = 1;
  if (tM == kTrans) std::swap(rs, cs);
  ASLP_EW(cudaF_add_mat_diag_vec(kD3, kD3, alpha, data_, Dim(), M.Data(), rs, cs, v.Data(), beta));
}
void CuMatrixBase::AddRowSumMat(float alpha, const CuMatrixBase &A, float beta) {
  ASLP_ASSERT(NumRows() != 0 && A.NumRows() % NumRows() == 0);
  ASLP_ASSERT(A.NumCols() == NumCols());
  ASLP_EW(cudaF_add_row_sum_mat(kD3, kD3, data_, A.Data(), Dim(), A.Stride(), A.NumRows() / NumRows(), alpha, beta));
}
void CuMatrixBase::AddConvMatMatElements(float alpha, const CuMatrixBase &A, const CuMatrixBase &B, float beta) {
  if (NumRows() == 0) return;
  int a = A.NumRows(), b = B.NumRows();
  ASLP_ASSERT(a >= b && NumRows() == ((a - b + 1) * b));
  ASLP_ASSERT(NumCols() == A.NumCols() && A.NumCols() == B.NumCols());
  aslp_dim3 bl = {2, (uint32_t)b, 1};
  ASLP_EW(cudaF_add_conv_mat_mat_elements(kD3, bl, data_, A.Data(), B.Data(), Dim(), A.Stride(), B.Stride(), alpha, beta));
}
void CuMatrixBase::MulElements(const CuMatrixBase &A) {
  ASLP_ASSERT(SameDim(*this, A));
  ASLP_EW(cudaF_mul_elements(kD3, kD3, data_, A.Data(), Dim(), A.Stride()));
}
void CuMatrixBase::MulColsVec(const CuVectorBase &scale) {
  ASLP_ASSERT(scale.Dim() == NumCols());
  ASLP_EW(cudaF_mul_cols_vec(kD3, kD3, data_, scale.Data(), Dim()));
}
void CuMatrixBase::MulRowsVec(const CuVectorBase &scale) {
  ASLP_ASSERT(scale.Dim() == NumRows());
  ASLP_EW(cudaF_mul_rows_vec(kD3, kD3, data_, scale.Data(), Dim()));
}
void CuMatrixBase::Sigmoid(const CuMatrixBase &src) { ASLP_ASSERT(SameDim(*this, src)); ASLP_EW(cudaF_sigmoid(kD3, kD3, data_, src.Data(), Dim(), src.Stride())); }
void CuMatrixBase::Tanh(const CuMatrixBase &src) { ASLP_ASSERT(SameDim(*this, src)); ASLP_EW(cudaF_tanh(kD3, kD3, data_, src.Data(), Dim(), src.Stride())); }
void CuMatrixBase::DiffSigmoid(const CuMatrixBase &value, const CuMatrixBase &diff) {
  ASLP_ASSERT(SameDim(*this, value) && SameDim(*this, diff));
  ASLP_EW(cudaF_diff_sigmoid(kD3, kD3, data_, diff.Data(), value.Data(), Dim(), diff.Stride(), value.Stride()));
}
void CuMatrixBase::DiffTanh(const CuMatrixBase &value, const CuMatrixBase &diff) {
  ASLP_ASSERT(SameDim(*this, value) && SameDim(*this, diff));
  ASLP_EW(cudaF_diff_tanh(kD3, kD3, data_, diff.Data(), value.Data(), Dim(), diff.Stride(), value.Stride()));
}
void CuMatrixBase::ApplySoftMaxPerRow(const CuMatrixBase &src) {
  ASLP_ASSERT(SameDim(*this, src));
  ASLP_EW(cudaF_softmax_reduce(0, 0, data_, src.Data(), Dim(), src.Stride()));
}
void CuMatrixBase::ApplyLogSoftMaxPerRow(const CuMatrixBase &src) {
  ASLP_ASSERT(SameDim(*this, src));
  ASLP_EW(cudaF_log_softmax_reduce(0, 0, data_, src.Data(), Dim(), src.Stride()));
}
void CuMatrixBase::FindRowMaxId(CuArray<int32> *id) const {
  id->Resize(rows_);
  if (!rows_) return;
  aslp_find_row_max_id(data_, Dim(), id->Data());
  CheckKernels();
}
void CuMatrixBase::CopyRows(const CuMatrixBase &src, const CuArray<int32> &indices) {
  ASLP_ASSERT(indices.Dim() == NumRows() && NumCols() == src.NumCols());
  ASLP_EW(cudaF_copy_rows(kD3, kD3, data_, src.Data(), indices.Data(), Dim(), src.Stride()));
}
void CuMatrixBase::AddRows(float alpha, const CuMatrixBase &src, const CuArray<int32> &indices) {
  ASLP_ASSERT(indices.Dim() == NumRows() && NumCols() == src.NumCols());
  ASLP_EW(cudaF_add_rows(kD3, kD3, alpha, data_, src.Data(), indices.Data(), Dim(), src.Stride()));
}
void CuMatrixBase::CopyCols(const CuMatrixBase &src, const CuArray<int32> &indices) {
  ASLP_ASSERT(indices.Dim() == NumCols() && NumRows() == src.NumRows());
  ASLP_EW(cudaF_copy_cols(kD3, kD3, data_, src.Data(), indices.Data(), Dim(), src.Stride()));
}
double CuMatrixBase::Sum() const {
  if (!rows_) return 0.0;
  double *d = static_cast<double *>(DeviceAlloc(sizeof(double)));
  aslp_matrix_sum(data_, Dim(), d);
  double h = 0.0;
  DeviceToHost(&h, d, sizeof(double));
  DeviceFree(d);
  CheckKernels();
  return h;
}

namespace {
void DownloadRange(const CuMatrixBase &m, float *lo, float *hi) {
  HostMatrix h;
  m.CopyToMat(&h);
  *lo = *hi = h.data.empty() ? 0.0f : h.data[0];
  for (float v : h.data) { *lo = v < *lo ? v : *lo; *hi = v > *hi ? v : *hi; }
}
}  // namespace
void CuMatrixBase::SetRandn() {
  if (!rows_) return;
  HostMatrix h(rows_, cols_);
  for (float &v : h.data) v = RandGauss();
  CopyFromMat(h);
}
void CuVectorBase::SetRandn() {
  if (!dim_) return;
  std::vector<float> h(dim_);
  for (float &v : h) v = RandGauss();
  CopyFromHost(h.data(), dim_);
}
float CuMatrixBase::Min() const { float lo, hi; DownloadRange(*this, &lo, &hi); return lo; }
float CuMatrixBase::Max() const { float lo, hi; DownloadRange(*this, &lo, &hi); return hi; }

namespace cu {
void Splice(const CuMatrixBase &src, const CuArray<int32> &frame_offsets, CuMatrixBase *tgt) {
  ASLP_ASSERT(src.NumCols() * frame_offsets.Dim() == tgt->NumCols());
  ASLP_ASSERT(src.NumRows() == tgt->NumRows());
  cudaF_splice(kD3, kD3, tgt->Data(), src.Data(), frame_offsets.Data(), tgt->Dim(), src.Dim());
  CheckKernels();
}
void Copy(const CuMatrixBase &src, const CuArray<int32> &copy_from_indices, CuMatrixBase *tgt) {
  ASLP_ASSERT(copy_from_indices.Dim() == tgt->NumCols());
  ASLP_ASSERT(src.NumRows() == tgt->NumRows());
  cudaF_copy(kD3, kD3, tgt->Data(), src.Data(), copy_from_indices.Data(), tgt->Dim(), src.Dim());
  CheckKernels();
}
void Randomize(const CuMatrixBase &src, const CuArray<int32> &copy_from_idx, CuMatrixBase *tgt) {
  ASLP_ASSERT(src.NumCols() == tgt->NumCols());
  ASLP_ASSERT(src.NumRows() == tgt->NumRows());
  ASLP_ASSERT(copy_from_idx.Dim() <= tgt->NumRows());
  MatrixDim dimsrc = src.Dim(), dimtgt = tgt->Dim();
  dimsrc.rows = copy_from_idx.Dim();
  dimtgt.rows = copy_from_idx.Dim();
  cudaF_randomize(kD3, kD3, tgt->Data(), src.Data(), copy_from_idx.Data(), dimtgt, dimsrc);
  CheckKernels();
}
void RegularizeL1(CuMatrixBase *weight, CuMatrixBase *grad, float l1, float lr) {
  ASLP_ASSERT(SameDim(*weight, *grad));
  cudaF_regularize_l1(kD3, kD3, weight->Data(), grad->Data(), l1, lr, weight->Dim(), grad->Stride());
  CheckKernels();
}
}  // namespace cu

// nnet-utils.h:61-124
static std::string MomentStatisticsHost(const std::vector<float> &v) {
  double n = (double)v.size();
  if (v.empty()) return " ( empty ) ";
  double sum = 0, mn = v[0], mx = v[0];
  for (float x : v) { sum += x; if (x < mn) mn = x; if (x > mx) mx = x; }
  float mean = sum / n;
  double m2 = 0, m3 = 0, m4 = 0;
  for (float x : v) { double d = x - mean; m2 += d * d; m3 += d * d * d; m4 += d * d * d * d; }
  float variance = m2 / n;
  float skewness = m3 / pow(variance, 3.0 / 2.0) / n;
  float kurtosis = m4 / (variance * variance) / n - 3.0;
  std::ostringstream ostr;
  ostr << " ( min " << (float)mn << ", max " << (float)mx << ", mean " << mean << ", variance " << variance << ", stddev "
       << sqrt(variance) << ", skewness " << skewness << ", kurtosis " << kurtosis << " ) ";
  return ostr.str();
}
std::string MomentStatistics(const CuMatrixBase &m) {
  HostMatrix h;
  m.CopyToMat(&h);
  return MomentStatisticsHost(h.data);
}
std::string MomentStatistics(const CuVectorBase &v) {
  std::vector<float> h(v.Dim());
  v.CopyToHost(h.data());
  return MomentStatisticsHost(h);
}

}  // namespace aslp
