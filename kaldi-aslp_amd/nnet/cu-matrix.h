// cu-matrix.h -- device matrix / vector / array substrate of the host engine.
//
// Mirrors the method vocabulary of the reference's CuMatrixBase / CuVectorBase / CuArray
// (src/aslp-cudamatrix/cu-matrix.h, cu-vector.h, cu-array.h) for the subset the hot path
// uses, on top of the gfx950 kernels of csrc/.  There is no CPU branch: every method launches
// a HIP kernel on the library stream.  Memory comes from a caching device allocator (the
// reference has one for the same reason, cu-allocator.h:67-70); rows are padded to a multiple
// of 16 floats so every row starts on a 64-byte boundary and float4 lanes never straddle rows.
#pragma once
#include <vector>

#include "aslp_kernels.h"
#include "base.h"
#include "host-matrix.h"

namespace aslp {

enum MatrixResizeType { kSetZero, kUndefined, kCopyData };
enum MatrixTransposeType { kNoTrans = 0, kTrans = 1 };

void *DeviceAlloc(size_t bytes);
void DeviceFree(void *p);
void DeviceToHost(void *dst, const void *src, size_t bytes);
void HostToDevice(void *dst, const void *src, size_t bytes);
void DeviceToDevice(void *dst, const void *src, size_t bytes);
void DeviceMemset(void *dst, int v, size_t bytes);
void StreamSync();
// Page-locked host memory + a completion marker, for uploads that must not pass through a staging copy: the caller fills
// a pinned block (possibly on another thread), CopyFromPinnedHost() sends it with one async copy, and the block may be
// reused once a StreamMarker recorded after the copy reports Done().
void *PinnedAlloc(size_t bytes);
void PinnedFree(void *p);      // back to the process-wide cache of page-locked blocks
void PinnedPoolRelease();      // hipHostFree of everything cached
class StreamMarker {
 public:
  StreamMarker();
  ~StreamMarker();
  void Record();        // on the calling thread's current stream
  bool Done() const;    // everything issued before Record() has finished
  void Wait() const;
 private:
  friend class CopyLane;
  void *ev_;
  bool recorded_;
  StreamMarker(const StreamMarker &) = delete;
  StreamMarker &operator=(const StreamMarker &) = delete;
};
// A second HIP stream for uploads that overlap the training stream (the calling thread's current stream): the frame tools
// fill the NEXT randomizer cache on it while the minibatches of the current one train.  Ordering between the two streams is
// explicit: LaneWaitsForStream() before the lane overwrites something the training stream may still read,
// StreamWaitsForLane() before the training stream reads what the lane wrote.  Both are device-side waits (no host stall).
class CopyLane {
 public:
  CopyLane();
  ~CopyLane();
  void Upload(float *dst, int dst_stride, const float *pinned_src, int ld, int rows, int cols);  // async H2D on the lane
  void Zero(void *dst, size_t bytes);                                                            // async on the lane
  void Record(StreamMarker *m);   // marker on the lane
  void LaneWaitsForStream();
  void StreamWaitsForLane();
  void Sync();                    // host waits for the lane (buffer growth, teardown)
 private:
  void *stream_, *ev_;
  CopyLane(const CopyLane &) = delete;
  CopyLane &operator=(const CopyLane &) = delete;
};
// Brings the HIP runtime fully up (context, code object, first launch).  The runtime draws from libc rand() while it
// initialises, so anything that seeds rand() for reproducible parameters (aslp-nnet-init) calls this BEFORE srand().
void WarmUpDevice();
inline int PaddedStride(int cols) { return (cols + 15) & ~15; }

template <typename T>
class CuArray {
 public:
  CuArray() : data_(nullptr), dim_(0) {}
  explicit CuArray(const std::vector<T> &v) : data_(nullptr), dim_(0) { CopyFromVec(v); }
  CuArray(const CuArray &o) : data_(nullptr), dim_(0) { *this = o; }
  CuArray &operator=(const CuArray &o) {
    if (this != &o) {
      Resize(o.dim_);
      if (dim_) DeviceToDevice(data_, o.data_, sizeof(T) * dim_);
    }
    return *this;
  }
  CuArray &operator=(const std::vector<T> &v) { CopyFromVec(v); return *this; }
  ~CuArray() { DeviceFree(data_); }
  void Resize(int n) {
    if (n == dim_) return;
    DeviceFree(data_);
    data_ = n ? static_cast<T *>(DeviceAlloc(sizeof(T) * n)) : nullptr;
    dim_ = n;
  }
  void CopyFromVec(const std::vector<T> &v) {
    Resize((int)v.size());
    if (dim_) HostToDevice(data_, v.data(), sizeof(T) * dim_);
  }
  void CopyToVec(std::vector<T> *v) const {
    v->resize(dim_);
    if (dim_) DeviceToHost(v->data(), data_, sizeof(T) * dim_);
  }
  int Dim() const { return dim_; }
  T *Data() { return data_; }
  const T *Data() const { return data_; }

 private:
  T *data_;
  int dim_;
};

class CuMatrixBase;
class CuSubVector;
class PlaneSet;

// a kernel-level error recorded since the last check (aslp_get_last_error: launch failures, unsupported arguments, a scratch block that
// could not be had) becomes the engine's exception here
inline void CheckKernelError() {
  char buf[512];
  if (aslp_get_last_error(buf, sizeof(buf))) ASLP_ERR << buf;
}

class CuVectorBase {
 public:
  int Dim() const { return dim_; }
  float *Data() { return data_; }
  const float *Data() const { return data_; }
  void SetZero();
  void Set(float v);
  void Add(float v);
  void Scale(float v);
  void CopyFromVec(const CuVectorBase &v);
  void CopyFromHost(const float *src, int n);
  void CopyToHost(float *dst) const;
  void CopyToVec(HostVector *v) const { v->data.resize(dim_); CopyToHost(v->data.data()); }
  void AddVec(float alpha, const CuVectorBase &v, float beta = 1.0f);               // this = alpha*v + beta*this
  void SetRandn();   // standard normals from the engine's seeded host generator (base.h RandGauss), one upload (cu-vector.h SetRandn)
  void AddRowSumMat(float alpha, const CuMatrixBase &M, float beta = 1.0f);         // column sums
  void AddColSumMat(float alpha, const CuMatrixBase &M, float beta = 1.0f);         // row sums
  void AddDiagMatMat(float alpha, const CuMatrixBase &M, MatrixTransposeType tM, const CuMatrixBase &N,
                     MatrixTransposeType tN, float beta = 1.0f);
  void AddVecVec(float alpha, const CuVectorBase &x, const CuVectorBase &y, float beta);
  void MulElements(const CuVectorBase &v);
  void ApplyFloor(float v);
  void ApplyCeiling(float v);
  void ApplyPow(float p);
  void InvertElements();
  float Sum() const;  // blocking
  CuSubVector Range(int o, int n);

 protected:
  CuVectorBase() : data_(nullptr), dim_(0) {}
  MatrixDim AsRow() const { MatrixDim d = {1, dim_, dim_}; return d; }
  float *data_;
  int dim_;
};

class CuVector : public CuVectorBase {
 public:
  CuVector() {}
  explicit CuVector(int dim, MatrixResizeType t = kSetZero) { Resize(dim, t); }
  CuVector(const CuVectorBase &o) { *this = o; }
  CuVector(const CuVector &o) : CuVectorBase() { *this = static_cast<const CuVectorBase &>(o); }
  CuVector &operator=(const CuVectorBase &o) {
    Resize(o.Dim(), kUndefined);
    CopyFromVec(o);
    return *this;
  }
  CuVector &operator=(const CuVector &o) { return *this = static_cast<const CuVectorBase &>(o); }
  CuVector &operator=(const HostVector &v) {
    Resize(v.Dim(), kUndefined);
    CopyFromHost(v.data.data(), v.Dim());
    return *this;
  }
  ~CuVector() { DeviceFree(data_); }
  void Resize(int dim, MatrixResizeType t = kSetZero);
  void Read(std::istream &is, bool binary) { HostVector v; v.Read(is, binary); *this = v; }
  void Write(std::ostream &os, bool binary) const { HostVector v; CopyToVec(&v); v.Write(os, binary); }
};

class CuSubVector : public CuVectorBase {
 public:
  CuSubVector(float *data, int dim) { data_ = data; dim_ = dim; }
};

// double vector: only what BatchNormalization's running statistics need
class CuVectorD {
 public:
  CuVectorD() : data_(nullptr), dim_(0) {}
  CuVectorD(const CuVectorD &o) : data_(nullptr), dim_(0) { *this = o; }
  CuVectorD &operator=(const CuVectorD &o);
  ~CuVectorD() { DeviceFree(data_); }
  void Resize(int dim, MatrixResizeType t = kSetZero);
  void SetZero();
  int Dim() const { return dim_; }
  double *Data() { return data_; }
  const double *Data() const { return data_; }
  void CopyFromHost(const double *src, int n);
  void CopyToHost(double *dst) const;
  void Read(std::istream &is, bool binary);
  void Write(std::ostream &os, bool binary) const;

 private:
  double *data_;
  int dim_;
};

class CuSubMatrix;

class CuMatrixBase {
 public:
  int NumRows() const { return rows_; }
  int NumCols() const { return cols_; }
  int Stride() const { return stride_; }
  float *Data() { return data_; }
  const float *Data() const { return data_; }
  float *RowData(int r) { return data_ + (size_t)r * stride_; }
  const float *RowData(int r) const { return data_ + (size_t)r * stride_; }
  MatrixDim Dim() const { MatrixDim d = {rows_, cols_, stride_}; return d; }

  CuSubMatrix Range(int r0, int nr, int c0, int nc) const;
  CuSubMatrix RowRange(int r0, int nr) const;
  CuSubMatrix ColRange(int c0, int nc) const;
  CuSubVector Row(int r);

  void SetZero();
  void Set(float v);
  void Add(float v);
  void Scale(float v);
  void ApplyFloor(float v);
  void ApplyCeiling(float v);
  void ApplyPow(float p);
  void ApplyLog();
  void ApplyExp();
  void ApplyHeaviside();
  void InvertElements();
  void CopyFromMat(const CuMatrixBase &src);
  void CopyFromMatTrans(const CuMatrixBase &src);   // *this = src^T (CopyFromMat(src, kTrans))
  void CopyFromHost(const float *src, int ld);
  void CopyFromPinnedHost(const float *src, int ld);  // async, no staging copy: `src` must stay valid until the stream passes
  void CopyToHost(float *dst, int ld) const;
  void CopyFromMat(const HostMatrix &m);
  void CopyToMat(HostMatrix *m) const;
  void WriteBinary(std::ostream &os) const;  // the bytes HostMatrix::Write(os, true) would write
  void AddMat(float alpha, const CuMatrixBase &A, MatrixTransposeType tA = kNoTrans);  // this += alpha*A
  void SetRandn();   // standard normals, row by row, from the engine's seeded host generator (cu-matrix.h SetRandn draws on the device)
  void AddMatMat(float alpha, const CuMatrixBase &A, MatrixTransposeType tA, const CuMatrixBase &B, MatrixTransposeType tB,
                 float beta, const aslp_gemm_epilogue *ep = nullptr);
  // the same with the prepared fp16 planes of A and / or B (csrc/split16.h; NULL: none)
  void AddMatMat(float alpha, const CuMatrixBase &A, MatrixTransposeType tA, const CuMatrixBase &B, MatrixTransposeType tB,
                 float beta, const aslp_gemm_epilogue *ep, const PlaneSet *pa, const PlaneSet *pb);
  void AddVecToRows(float alpha, const CuVectorBase &row, float beta = 1.0f);
  void AddVecToCols(float alpha, const CuVectorBase &col, float beta = 1.0f);
  void AddMatMatElements(float alpha, const CuMatrixBase &A, const CuMatrixBase &B, float beta);
  void AddMatDiagVec(float alpha, const CuMatrixBase &M, MatrixTransposeType tM, const CuVectorBase &v, float beta = 1.0f);
  void AddRowSumMat(float alpha, const CuMatrixBase &A, float beta);           // ASLP group-row sum (cu-matrix.cc:3010)
  void AddConvMatMatElements(float alpha, const CuMatrixBase &A, const CuMatrixBase &B, float beta);  // cu-matrix.cc:3037
  void MulElements(const CuMatrixBase &A);
  void MulColsVec(const CuVectorBase &scale);
  void MulRowsVec(const CuVectorBase &scale);
  void Sigmoid(const CuMatrixBase &src);
  void Tanh(const CuMatrixBase &src);
  void DiffSigmoid(const CuMatrixBase &value, const CuMatrixBase &diff);
  void DiffTanh(const CuMatrixBase &value, const CuMatrixBase &diff);
  void ApplySoftMaxPerRow(const CuMatrixBase &src);
  void ApplyLogSoftMaxPerRow(const CuMatrixBase &src);
  void FindRowMaxId(CuArray<int32> *id) const;
  void CopyRows(const CuMatrixBase &src, const CuArray<int32> &indices);
  void AddRows(float alpha, const CuMatrixBase &src, const CuArray<int32> &indices);
  void CopyCols(const CuMatrixBase &src, const CuArray<int32> &indices);
  double Sum() const;  // blocking (CuMatrixBase::Sum)
  float Min() const;   // blocking, one download (cu-matrix.h Min / Max: the forward tools' range checks on an utterance's output)
  float Max() const;

 protected:
  CuMatrixBase() : data_(nullptr), rows_(0), cols_(0), stride_(0) {}
  CuMatrixBase(float *d, int r, int c, int s) : data_(d), rows_(r), cols_(c), stride_(s) {}
  float *data_;
  int rows_, cols_, stride_;
};

class CuMatrix : public CuMatrixBase {
 public:
  CuMatrix() {}
  CuMatrix(int rows, int cols, MatrixResizeType t = kSetZero) { Resize(rows, cols, t); }
  CuMatrix(const CuMatrixBase &o) { *this = o; }
  CuMatrix(const CuMatrix &o) : CuMatrixBase() { *this = static_cast<const CuMatrixBase &>(o); }
  explicit CuMatrix(const HostMatrix &m) { *this = m; }
  CuMatrix &operator=(const CuMatrixBase &o) {
    Resize(o.NumRows(), o.NumCols(), kUndefined);
    CopyFromMat(o);
    return *this;
  }
  CuMatrix &operator=(const CuMatrix &o) { return *this = static_cast<const CuMatrixBase &>(o); }
  CuMatrix &operator=(const HostMatrix &m) {
    Resize(m.rows, m.cols, kUndefined);
    CopyFromMat(m);
    return *this;
  }
  ~CuMatrix() { DeviceFree(data_); }
  void Resize(int rows, int cols, MatrixResizeType t = kSetZero);
  void Swap(CuMatrix *o) {
    std::swap(data_, o->data_); std::swap(rows_, o->rows_); std::swap(cols_, o->cols_); std::swap(stride_, o->stride_);
  }
  void Read(std::istream &is, bool binary) { HostMatrix m; m.Read(is, binary); *this = m; }
  void Write(std::ostream &os, bool binary) const {
    if (binary) { WriteBinary(os); return; }
    HostMatrix m; CopyToMat(&m); m.Write(os, binary);
  }
};

class CuSubMatrix : public CuMatrixBase {
 public:
  CuSubMatrix(const CuMatrixBase &m, int r0, int nr, int c0, int nc)
      : CuMatrixBase(const_cast<float *>(m.Data()) + (size_t)r0 * m.Stride() + c0, nr, nc, m.Stride()) {
    ASLP_ASSERT(r0 >= 0 && nr >= 0 && r0 + nr <= m.NumRows() && c0 >= 0 && nc >= 0 && c0 + nc <= m.NumCols());
  }
  CuSubMatrix(float *data, int rows, int cols, int stride) : CuMatrixBase(data, rows, cols, stride) {}
};

// C0 = alpha op(A0) op(B0) + beta C0 (+ ep0) and C1 = alpha op(A1) op(B1) + beta C1 (+ ep1): one launch when the two products agree
// in shape and strides (aslp_sgemm_pair_ex), two AddMatMat calls otherwise -- same results either way.  The forward and backward
// direction of a bidirectional recurrent layer issue every batched product as such a pair.
void AddMatMatPair(CuMatrixBase &C0, CuMatrixBase &C1, float alpha, const CuMatrixBase &A0, const CuMatrixBase &A1, MatrixTransposeType tA,
                   const CuMatrixBase &B0, const CuMatrixBase &B1, MatrixTransposeType tB, float beta, const aslp_gemm_epilogue *ep0 = nullptr,
                   const aslp_gemm_epilogue *ep1 = nullptr);
// ... with prepared fp16 planes (windows) of the four operands, in the order A0, A1, B0, B1 (csrc/split16.h); NULL: none
struct S16View;
void AddMatMatPair(CuMatrixBase &C0, CuMatrixBase &C1, float alpha, const CuMatrixBase &A0, const CuMatrixBase &A1, MatrixTransposeType tA,
                   const CuMatrixBase &B0, const CuMatrixBase &B1, MatrixTransposeType tB, float beta, const aslp_gemm_epilogue *ep0,
                   const aslp_gemm_epilogue *ep1, const S16View *views);

inline bool SameDim(const CuMatrixBase &a, const CuMatrixBase &b) { return a.NumRows() == b.NumRows() && a.NumCols() == b.NumCols(); }

namespace cu {
// cu-math.h
void Splice(const CuMatrixBase &src, const CuArray<int32> &frame_offsets, CuMatrixBase *tgt);
void Copy(const CuMatrixBase &src, const CuArray<int32> &copy_from_indices, CuMatrixBase *tgt);
void Randomize(const CuMatrixBase &src, const CuArray<int32> &copy_from_idx, CuMatrixBase *tgt);
void RegularizeL1(CuMatrixBase *weight, CuMatrixBase *grad, float l1, float lr);
}  // namespace cu

std::string MomentStatistics(const CuMatrixBase &m);
std::string MomentStatistics(const CuVectorBase &v);

}  // namespace aslp
