/* aslp_compat_kaldi_matrix.h -- kaldi::Matrix / Vector / SubVector / SubMatrix for callers written against the reference (part of the B4
 * source-level drop-in, included by aslp_compat_kaldi.h).
 *
 * The reference's tools hold utterances and frame weights in kaldi::Matrix<BaseFloat> / Vector<BaseFloat> (matrix/kaldi-matrix.h,
 * matrix/kaldi-vector.h -- the un-vendored Kaldi matrix library) between the table readers and the device.  The engine's host types are
 * aslp::HostMatrix (row-major, unpadded) and std::vector<float>; the classes here ARE those types (public bases), with the members the
 * reference's aslp-nnetbin mains call on them, so a Matrix goes wherever the engine takes a HostMatrix (CuMatrix::operator=, CopyToMat,
 * the table writers) and a Vector wherever it takes a std::vector<float> (LossItf::Eval's frame weights, VectorRandomizer::AddData).
 * Host-side glue only: nothing here runs on the training path. */
#ifndef ASLP_COMPAT_KALDI_MATRIX_H_
#define ASLP_COMPAT_KALDI_MATRIX_H_

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "cu-matrix.h"
#include "host-matrix.h"

namespace kaldi {

/* a run of floats somebody else owns: a row of a Matrix, a range of a Vector (matrix/kaldi-vector.h SubVector) */
template <typename Real>
class SubVector {
 public:
  SubVector(float *data, int dim) : data_(data), dim_(dim) {}
  template <class V> SubVector(V &v, int origin, int length) : data_(v.Data() + origin), dim_(length) {}
  int Dim() const { return dim_; }
  float *Data() { return data_; }
  const float *Data() const { return data_; }
  float &operator()(int i) { return data_[i]; }
  float operator()(int i) const { return data_[i]; }
  template <class V> void CopyFromVec(const V &v) { ASLP_ASSERT(v.Dim() == dim_); std::memcpy(data_, v.Data(), sizeof(float) * dim_); }
  void Set(float v) { std::fill(data_, data_ + dim_, v); }
  void Scale(float a) { for (int i = 0; i < dim_; i++) data_[i] *= a; }
  float Sum() const { double s = 0.0; for (int i = 0; i < dim_; i++) s += data_[i]; return (float)s; }
 private:
  float *data_;
  int dim_;
};

template <typename Real>
class Vector : public std::vector<float> {
 public:
  Vector() {}
  explicit Vector(int dim, ::aslp::MatrixResizeType = ::aslp::kSetZero) : std::vector<float>(dim, 0.0f) {}
  Vector(const std::vector<float> &v) : std::vector<float>(v) {}
  Vector(const ::aslp::HostVector &v) : std::vector<float>(v.data) {}
  Vector(const ::aslp::CuVectorBase &v) : std::vector<float>(v.Dim()) { if (v.Dim()) v.CopyToHost(data()); }
  template <class R> Vector(const SubVector<R> &v) : std::vector<float>(v.Data(), v.Data() + v.Dim()) {}
  int Dim() const { return (int)size(); }
  float *Data() { return data(); }
  const float *Data() const { return data(); }
  float &operator()(int i) { return (*this)[i]; }
  float operator()(int i) const { return (*this)[i]; }
  void Resize(int dim, ::aslp::MatrixResizeType t = ::aslp::kSetZero) {
    if (t == ::aslp::kCopyData) resize(dim, 0.0f);
    else assign(dim, 0.0f);
  }
  void Set(float v) { std::fill(begin(), end(), v); }
  void SetZero() { Set(0.0f); }
  void Scale(float a) { for (float &x : *this) x *= a; }
  void Add(float a) { for (float &x : *this) x += a; }
  float Sum() const { double s = 0.0; for (float x : *this) s += x; return (float)s; }
  float Min() const { return empty() ? 0.0f : *std::min_element(begin(), end()); }
  float Max() const { return empty() ? 0.0f : *std::max_element(begin(), end()); }
  template <class V> void CopyFromVec(const V &v) { ASLP_ASSERT(v.Dim() == Dim()); if (Dim()) std::memcpy(data(), v.Data(), sizeof(float) * size()); }
  SubVector<Real> Range(int origin, int length) { return SubVector<Real>(data() + origin, length); }
  void Read(std::istream &is, bool binary) { ::aslp::HostVector v; v.Read(is, binary); static_cast<std::vector<float> &>(*this) = v.data; }
  void Write(std::ostream &os, bool binary) const { ::aslp::HostVector v; v.data = *this; v.Write(os, binary); }
};
template <typename Real> using VectorBase = Vector<Real>;

template <typename Real> class SubMatrix;

template <typename Real>
class Matrix : public ::aslp::HostMatrix {
 public:
  Matrix() {}
  Matrix(int r, int c, ::aslp::MatrixResizeType = ::aslp::kSetZero) : ::aslp::HostMatrix(r, c) {}
  Matrix(const ::aslp::HostMatrix &m) : ::aslp::HostMatrix(m) {}
  explicit Matrix(const ::aslp::CuMatrixBase &m) { m.CopyToMat(this); }
  Matrix(const SubMatrix<Real> &m);
  Matrix &operator=(const ::aslp::HostMatrix &m) { static_cast<::aslp::HostMatrix &>(*this) = m; return *this; }
  Matrix &operator=(const SubMatrix<Real> &m) { Matrix t(m); Swap(&t); return *this; }
  int NumRows() const { return rows; }
  int NumCols() const { return cols; }
  int Stride() const { return cols; }
  float *Data() { return data.data(); }
  const float *Data() const { return data.data(); }
  float *RowData(int r) { return data.data() + (size_t)r * cols; }
  const float *RowData(int r) const { return data.data() + (size_t)r * cols; }
  void Swap(Matrix *o) { std::swap(rows, o->rows); std::swap(cols, o->cols); data.swap(o->data); }
  void Resize(int r, int c, ::aslp::MatrixResizeType t = ::aslp::kSetZero) {
    if (t != ::aslp::kCopyData) { ::aslp::HostMatrix::Resize(r, c); return; }
    Matrix n(r, c);   /* matrix/kaldi-matrix.cc Resize(kCopyData): the overlapping block stays, the rest is zero */
    const int rr = std::min(r, rows), cc = std::min(c, cols);
    for (int i = 0; i < rr; i++) std::memcpy(n.RowData(i), RowData(i), sizeof(float) * cc);
    Swap(&n);
  }
  void SetZero() { std::fill(data.begin(), data.end(), 0.0f); }
  void Set(float v) { std::fill(data.begin(), data.end(), v); }
  void Scale(float a) { for (float &x : data) x *= a; }
  float Sum() const { double s = 0.0; for (float x : data) s += x; return (float)s; }
  float Min() const { return data.empty() ? 0.0f : *std::min_element(data.begin(), data.end()); }
  float Max() const { return data.empty() ? 0.0f : *std::max_element(data.begin(), data.end()); }
  SubVector<Real> Row(int r) { return SubVector<Real>(RowData(r), cols); }
  const SubVector<Real> Row(int r) const { return SubVector<Real>(const_cast<float *>(RowData(r)), cols); }
  template <class V> void CopyRowFromVec(const V &v, int r) { ASLP_ASSERT(v.Dim() == cols); std::memcpy(RowData(r), v.Data(), sizeof(float) * cols); }
  template <class M> void CopyFromMat(const M &m) {
    ASLP_ASSERT(m.NumRows() == rows && m.NumCols() == cols);
    for (int i = 0; i < rows; i++) std::memcpy(RowData(i), m.RowData(i), sizeof(float) * cols);
  }
  SubMatrix<Real> Range(int r0, int nr, int c0, int nc);
  SubMatrix<Real> RowRange(int r0, int nr);
  SubMatrix<Real> ColRange(int c0, int nc);
};
template <typename Real> using MatrixBase = Matrix<Real>;

/* a block of a Matrix (rows r0.., columns c0..), row stride the parent's */
template <typename Real>
class SubMatrix {
 public:
  SubMatrix(float *data, int rows, int cols, int stride) : data_(data), rows_(rows), cols_(cols), stride_(stride) {}
  SubMatrix(Matrix<Real> &m, int r0, int nr, int c0, int nc) : data_(m.RowData(r0) + c0), rows_(nr), cols_(nc), stride_(m.NumCols()) {
    ASLP_ASSERT(r0 >= 0 && nr >= 0 && r0 + nr <= m.NumRows() && c0 >= 0 && nc >= 0 && c0 + nc <= m.NumCols());
  }
  int NumRows() const { return rows_; }
  int NumCols() const { return cols_; }
  int Stride() const { return stride_; }
  float *RowData(int r) { return data_ + (size_t)r * stride_; }
  const float *RowData(int r) const { return data_ + (size_t)r * stride_; }
  float &operator()(int r, int c) { return RowData(r)[c]; }
  float operator()(int r, int c) const { return RowData(r)[c]; }
  SubVector<Real> Row(int r) { return SubVector<Real>(RowData(r), cols_); }
  template <class M> void CopyFromMat(const M &m) {
    ASLP_ASSERT(m.NumRows() == rows_ && m.NumCols() == cols_);
    for (int i = 0; i < rows_; i++) std::memcpy(RowData(i), m.RowData(i), sizeof(float) * cols_);
  }
  float Sum() const { double s = 0.0; for (int i = 0; i < rows_; i++) for (int j = 0; j < cols_; j++) s += RowData(i)[j]; return (float)s; }
 private:
  float *data_;
  int rows_, cols_, stride_;
};
template <typename Real> Matrix<Real>::Matrix(const SubMatrix<Real> &m) : ::aslp::HostMatrix(m.NumRows(), m.NumCols()) {
  for (int i = 0; i < rows; i++) std::memcpy(RowData(i), m.RowData(i), sizeof(float) * cols);
}
template <typename Real> SubMatrix<Real> Matrix<Real>::Range(int r0, int nr, int c0, int nc) { return SubMatrix<Real>(*this, r0, nr, c0, nc); }
template <typename Real> SubMatrix<Real> Matrix<Real>::RowRange(int r0, int nr) { return SubMatrix<Real>(*this, r0, nr, 0, cols); }
template <typename Real> SubMatrix<Real> Matrix<Real>::ColRange(int c0, int nc) { return SubMatrix<Real>(*this, 0, rows, c0, nc); }

}  // namespace kaldi

#endif  /* ASLP_COMPAT_KALDI_MATRIX_H_ */
