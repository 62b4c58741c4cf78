// split16.h -- fp32 matrices carried as two fp16 planes behind a power-of-two scale (the operand format of gemm_split16.hip).
//
// An fp32 value x of a matrix with scale s = 2^up is held as  x s = hi + 2^-11 lo',  hi = fp16(x s),  lo' = fp16((x s - hi) 2^11):
// 22 significant bits for every element whose |x s| lies in fp16's normal range.  The scale comes from a BOUND of the matrix: any
// finite b >= max |x| (its bits live in a device word, the "slot"); s puts b into [2^13, 2^14), so every |x s| < 2^14 (fp16 overflows
// at 2^16) and elements down to 2^-28 b keep their 22 bits (an element below that adds < 2^-39 b to a sum either way).  The bound need not
// be tight: a producer that knows one before it has written its last element (a sigmoid's 1, |y - t| <= 1, |W| + lr |dW|) writes
// the planes in the same pass as the fp32 values; everything else leaves per-workgroup maxima and is converted by one pass.
//
// Planes have the LAYOUT OF THE MATRIX (row r, column c at [r * ld + c]), ld = cols rounded up to 64, rows rounded up to 64, zeros in
// the padding: the product kernels read an operand whose reduction index is contiguous with ds_read_b128 and one whose reduction
// index is the row index with the transposing LDS read (ds_read_b64_tr_b16), so ONE pair of planes serves every product that reads
// the matrix (x: forward and weight gradient; dy: in-diff and weight gradient; W: forward and in-diff).
#pragma once
#include "aslp_kernels.h"
#include "common.h"

namespace aslp {

typedef _Float16 h16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int kS16Pad = 64;         // both extents of the planes are multiples of this
constexpr int kS16MaxParts = 4096;  // maxima (one per workgroup) a producer may leave for one matrix
constexpr int kS16ConvParts = 256;  // what the library's own maximum pass leaves

// halves per plane row for a matrix of `cols` columns: cols rounded up to 64
int s16_plane_ld(int cols);

// what a kernel needs of a pair of planes (by value in kernel arguments)
struct S16View {
  h16 *hi, *lo;
  int ld;            // halves per plane row
  int rows, cols;    // of the fp32 matrix
  unsigned *slot;    // bits of the bound the planes are (to be) scaled by
};

// exponent `up` of the scale 2^up for a bound (0, inf and NaN: no scaling)
__host__ __device__ inline int s16_exponent(unsigned bound_bits) {
  union { unsigned u; float f; } v;
  v.u = bound_bits;
  const float b = v.f;
  if (!(b > 0.f && b < 3.0e38f)) return 0;
  int e = 0;
  (void)frexpf(b, &e);   // b = f 2^e, f in [0.5, 1)
  int up = 14 - e;
  return up > 120 ? 120 : (up < -120 ? -120 : up);
}

// where a producer kernel leaves planes / maxima (include/aslp_kernels.h aslp_planes_out, as the kernels take it)
struct S16Out {
  h16 *hi, *lo;
  int ld;
  const unsigned *slot;
  float *parts;
};

#if defined(__HIPCC__)
__device__ __forceinline__ void s16_split(float x, float s, h16 *hi, h16 *lo) {
  const float y = x * s;
  const h16 h = (h16)y;
  *hi = h;
  *lo = (h16)((y - (float)h) * 2048.f);
}
__device__ __forceinline__ void s16_split4(const float4 v, float s, half4 *hi, half4 *lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    h16 h, l;
    s16_split(x[i], s, &h, &l);
    (*hi)[i] = h;
    (*lo)[i] = l;
  }
}
// largest finite |x| of four values (non-finite elements do not set the scale: they stay inf / NaN in the planes and make their rows
// and columns of a product non-finite, like they do in the fp32 kernels, while every other element keeps its precision)
__device__ __forceinline__ float s16_absmax4(float m, const float4 v) {
  const float a[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
  for (int i = 0; i < 4; i++) m = fmaxf(m, a[i] < 3.0e38f ? a[i] : 0.f);
  return m;
}
#endif

// (Optional copy and) planes of up to kS16MaxJobs matrices in ONE launch, each scaled by its own maximum, which the launch's workgroups find
// among themselves (nn_fused.hip copy_planes_coop): instead of a maximum launch and a conversion launch.  parts / nparts: a bound of the
// matrix known on the device (as ConvertSpec).  false = not served (too large for one resident grid, unaligned, a shared device): nothing
// was launched.
struct CoopConvJob { const float *src; int ld_src; float *dst; int ld_dst; S16View pl; const float *parts; int nparts; };
constexpr int kS16MaxJobs = 8;     // matrices per maximum / conversion launch (blockIdx.y)
// A recurrent layer prepares its two activation buffers for the persistent kernel (rnn_persistent.hip: boundary row blocks 0 and T + 1 := 0, or
// the carried history in block 0 of buffer 0; columns [col0, col0 + ncols) of row blocks 1..T := "not yet published", every float the pattern
// 0xFFFFFFFF) right in front of a conversion launch of the same step: the launch takes the fill along (one launch less per layer and pass).
// buf0 == NULL: nothing to fill.  The conversions never read or write these buffers.
struct SeqFillJob { float *buf0, *buf1; int ld, T, S, col0, ncols; const float *init; int ld_init, init_cols; };
bool coop_convert_launch(const CoopConvJob *jobs, int n, const SeqFillJob *fill = nullptr);
#ifdef __HIPCC__
// one row (0 .. (T + 2) S - 1) of buffer `which`, by all threads of the calling workgroup
__device__ __forceinline__ void seq_fill_row(const SeqFillJob &f, int row, int which) {
  typedef unsigned fill_u32x4 __attribute__((ext_vector_type(4)));
  float *buf = which == 0 ? f.buf0 : f.buf1;
  const bool boundary = row < f.S || row >= (f.T + 1) * f.S;
  fill_u32x4 *p = reinterpret_cast<fill_u32x4 *>(buf + (long)row * f.ld + (boundary ? 0 : f.col0));
  const int nt = (int)blockDim.x;
  if (f.init != nullptr && which == 0 && row < f.S) {
    const fill_u32x4 *q = reinterpret_cast<const fill_u32x4 *>(f.init + (long)row * f.ld_init);
    const fill_u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < (f.ld >> 2); i += nt) p[i] = i < (f.init_cols >> 2) ? q[i] : z;
    return;
  }
  const int n4 = (boundary ? f.ld : f.ncols) >> 2;
  const unsigned w = boundary ? 0u : 0xFFFFFFFFu;
  const fill_u32x4 v = {w, w, w, w};
  for (int i = threadIdx.x; i < n4; i += nt) p[i] = v;
}
#endif

// ---- host side: device planes of one fp32 matrix, reused from step to step --------------------------------------------------------
// valid_for(...): the planes hold the values of that matrix as of the tag's epoch (the executor bumps the epochs: nnet-nnet.cpp).
class PlaneSet {
 public:
  PlaneSet() = default;
  ~PlaneSet();
  PlaneSet(const PlaneSet &) = delete;
  PlaneSet &operator=(const PlaneSet &) = delete;
  // planes for a [rows x cols] matrix (zeroed when the buffers are new or the shape changed: producers only write the matrix region)
  bool Reserve(int rows, int cols);
  bool ReserveParts();   // the maxima array and the slot only (a set that only carries a producer's maxima to their reader)
  S16View View() const { return S16View{hi_, lo_, ld_, rows_, cols_, slot_}; }
  float *Parts() const { return parts_; }        // kS16MaxParts per-workgroup maxima (producers) -> Convert / bound kernels
  unsigned *Slot() const { return slot_; }
  int Rows() const { return rows_; }
  int Cols() const { return cols_; }
  // max pass + conversion of src (two launches on the current stream)
  bool ConvertFrom(const float *src, int rows, int cols, int stride);
  // several matrices with ONE maximum launch and ONE conversion launch (at most 8)
  // (parts / nparts: device maxima that bound |src| -- left by the kernel that wrote src, or OneBound() for values known to lie in
  // [-1, 1] -- spare that matrix the maximum pass; when every matrix of the call has them there is no maximum launch)
  struct ConvertSpec { PlaneSet *planes; const float *src; int rows, cols, stride; const float *parts = nullptr; int nparts = 0; };
  // fill / fill_done: a recurrent layer's buffer preparation to take along (SeqFillJob above); *fill_done says whether this call did it
  static bool ConvertMany(const ConvertSpec *specs, int n, const SeqFillJob *fill = nullptr, bool *fill_done = nullptr);
  static const float *OneBound();   // a device float holding 1.0
  // a [rows x cols] window of the planes at (row0, col0), for a product that reads that block of the matrix.  As a reduction extent
  // the window's must be a multiple of 64 (or end at the matrix' edge: behind it the planes hold zeros, inside they hold the neighbours)
  S16View Window(int row0, int rows, int col0, int cols) const {
    return S16View{hi_ + (long)row0 * ld_ + col0, lo_ + (long)row0 * ld_ + col0, ld_, rows, cols, slot_};
  }
  // conversion with the bound already in Parts() (nparts per-workgroup maxima left by the kernel that wrote src)
  bool ConvertWithParts(const float *src, int rows, int cols, int stride, int nparts);
  // the planes were / will be written by a producer under a bound known on the host (e.g. 1 for sigmoid outputs)
  bool SetBound(float bound);
  void ForgetHostBound() { host_bound_ = -1.f; }   // a kernel has written the slot
  // validity tag
  void Tag(const void *src, int stride, long epoch) { src_ = src; stride_ = stride; epoch_ = epoch; }
  bool ValidFor(const void *src, int rows, int cols, int stride, long epoch) const {
    return hi_ && src_ == src && rows_ == rows && cols_ == cols && stride_ == stride && epoch_ == epoch;
  }
  void Invalidate() { src_ = nullptr; epoch_ = -1; }
  // the kernel that wrote `src` left nparts per-workgroup maxima in Parts(): the next conversion of that matrix skips its maximum pass
  void TagParts(const void *src, int nparts, long epoch) { parts_src_ = src; nparts_ = nparts; parts_epoch_ = epoch; }
  int PartsFor(const void *src, long epoch) const { return (epoch != 0 && parts_src_ == src && parts_epoch_ == epoch) ? nparts_ : 0; }

 private:
  h16 *hi_ = nullptr, *lo_ = nullptr;
  unsigned *slot_ = nullptr;
  float *parts_ = nullptr;
  float host_bound_ = -1.f;
  size_t cap_ = 0;   // halves per plane allocated
  int rows_ = 0, cols_ = 0, ld_ = 0, rows_p_ = 0;
  const void *src_ = nullptr;
  int stride_ = 0;
  long epoch_ = -1;
  const void *parts_src_ = nullptr;
  int nparts_ = 0;
  long parts_epoch_ = -1;
};

bool gemm_split16_enabled();
// would aslp_sgemm_ex carry an M x N x K product on the fp16 instruction (switch on, shape served)?
bool gemm_split16_serves(int M, int N, int K);
int gemm_split16_last_parts();
int gemm_split16_last_tile();   // 311 / 308 / 328 (gemm_split16.hip)
void gemm_split16_reset_last_parts();
int gemm_split16_max_parts(int M, int N);

// ---- when may planes be reused?  Only inside the graph executor (nnet-nnet.cpp), which knows when its buffers are written: it draws a
// fresh epoch for every forward pass and for every backward pass and publishes both to the calling thread while a pass runs; planes are
// tagged with the epoch they were made under.  Outside a pass the epochs are 0 = "nothing may be reused": a component called on its own
// converts its operands in every call.
long s16_new_epoch();                        // process-wide unique, > 0
struct S16Epochs { long fwd = 0, bwd = 0; };
S16Epochs &s16_epochs();                     // the calling thread's
struct S16EpochScope {                       // publishes (fwd, bwd) for the duration of a pass
  S16EpochScope(long fwd, long bwd) : saved_(s16_epochs()) { s16_epochs().fwd = fwd; s16_epochs().bwd = bwd; }
  ~S16EpochScope() { s16_epochs() = saved_; }
  S16EpochScope(const S16EpochScope &) = delete;
  S16EpochScope &operator=(const S16EpochScope &) = delete;
 private:
  S16Epochs saved_;
};

// ---- the loss writes the out-diff of the network's last layer product: Nnet::LossDiff() says (to the calling thread) where that
// layer wants the diff's planes and under which epoch the backward pass (BackpropagateFromLossDiff) will look for them; a loss whose
// kernel knows a bound of |diff| before it runs (Xent on posteriors: |y - t| w <= max w) writes them, tags them and clears this.
struct S16DiffTarget { PlaneSet *planes = nullptr; long epoch = 0; const void *diff = nullptr; };
S16DiffTarget &s16_loss_diff_target();

// aslp_sgemm_pair_ex with prepared planes (or windows of planes) of all four operands -- all four or none; K a multiple of 64
int sgemm_pair_views(int transA, int transB, int M, int N, int K, float alpha, const float *A0, const float *A1, int lda, const float *B0,
                     const float *B1, int ldb, float beta, float *C0, float *C1, int ldc, const aslp_gemm_epilogue *ep0,
                     const aslp_gemm_epilogue *ep1, const S16View *va0, const S16View *va1, const S16View *vb0, const S16View *vb1);

// ---- parameters written behind the components' backs (model averaging through the GetGpuParams pointers): whoever writes them calls
// aslp_params_changed(), which moves this epoch; planes of weights kept from step to step are tagged with it
long s16_param_epoch();
bool s16_keep_weight_planes();   // A/B switch ASLP_KEEP_WEIGHT_PLANES=0 / aslp_keep_weight_planes(0): weights' planes made anew in every step

}  // namespace aslp
