/* aslp_compat_kaldi_matrix.h -- kaldi::Matrix / Vector / SubVector / SubMatrix (+ MatrixBase / VectorBase) for callers written against the
 * reference (part of the B4 source-level drop-in, included by aslp_compat_kaldi.h).
 *
 * The reference's tools and unit tests hold utterances and frame weights in kaldi::Matrix<BaseFloat> / Vector<BaseFloat> (matrix/kaldi-matrix.h,
 * matrix/kaldi-vector.h -- the un-vendored Kaldi matrix library) between the table readers and the device.  The engine's host types are
 * aslp::HostMatrix (row-major, unpadded) and std::vector<float>; Matrix and Vector here ARE those types (public bases), so a Matrix goes
 * wherever the engine takes a HostMatrix (CuMatrix::operator=, CopyToMat, the table writers) and a Vector wherever it takes a
 * std::vector<float> (LossItf::Eval's frame weights, VectorRandomizer::AddData, SequenceDataReader::ReadData's mask).  What Kaldi's callers
 * do with them -- element access, rows and ranges as views, sums, copies -- is in MatrixBase / VectorBase, interfaces over "where the floats
 * are", which the owning classes and the views both implement.  Host-side glue only: nothing here runs on the training path. */
#ifndef ASLP_COMPAT_KALDI_MATRIX_H_
#define ASLP_COMPAT_KALDI_MATRIX_H_

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "cu-matrix.h"
#include "host-matrix.h"

namespace kaldi {

typedef int MatrixIndexT;   /* matrix/matrix-common.h */
template <typename Real> class SubVector;
template <typename Real> class SubMatrix;

/* ---- vectors ------------------------------------------------------------------------------------------------------------------------- */
template <typename Real>
class VectorBase {
 public:
  virtual ~VectorBase() {}
  virtual int Dim() const = 0;
  virtual const float *Data() const = 0;
  float *Data() { return const_cast<float *>(static_cast<const VectorBase *>(this)->Data()); }
  float &operator()(int i) { return Data()[i]; }
  float operator()(int i) const { return Data()[i]; }
  void Set(float v) { std::fill(Data(), Data() + Dim(), v); }
  void SetZero() { Set(0.0f); }
  void Scale(float a) { float *p = Data(); for (int i = 0, n = Dim(); i < n; i++) p[i] *= a; }
  void Add(float a) { float *p = Data(); for (int i = 0, n = Dim(); i < n; i++) p[i] += a; }
  float Sum() const { double s = 0.0; const float *p = Data(); for (int i = 0, n = Dim(); i < n; i++) s += p[i]; return (float)s; }
  float Min() const { return Dim() ? *std::min_element(Data(), Data() + Dim()) : 0.0f; }
  float Max() const { return Dim() ? *std::max_element(Data(), Data() + Dim()) : 0.0f; }
  void CopyFromVec(const VectorBase &v) { ASLP_ASSERT(v.Dim() == Dim()); if (Dim()) std::memmove(Data(), v.Data(), sizeof(float) * Dim()); }
  SubVector<Real> Range(int origin, int length) const;
};

/* a run of floats somebody else owns: a row of a Matrix, a range of a Vector */
template <typename Real>
class SubVector : public VectorBase<Real> {
 public:
  SubVector(const float *data, int dim) : data_(data), dim_(dim) {}
  SubVector(const VectorBase<Real> &v, int origin, int length) : data_(v.Data() + origin), dim_(length) { ASLP_ASSERT(origin >= 0 && length >= 0 && origin + length <= v.Dim()); }
  int Dim() const override { return dim_; }
  const float *Data() const override { return data_; }
  using VectorBase<Real>::Data;
 private:
  const float *data_;
  int dim_;
};
template <typename Real> SubVector<Real> VectorBase<Real>::Range(int origin, int length) const { return SubVector<Real>(*this, origin, length); }

template <typename Real>
class Vector : public std::vector<float>, public VectorBase<Real> {
 public:
  Vector() {}
  explicit Vector(int dim, ::aslp::MatrixResizeType = ::aslp::kSetZero) : std::vector<float>(dim, 0.0f) {}
  Vector(const Vector &v) : std::vector<float>(v), VectorBase<Real>() {}
  Vector(const std::vector<float> &v) : std::vector<float>(v) {}
  Vector(const ::aslp::HostVector &v) : std::vector<float>(v.data) {}
  Vector(const ::aslp::CuVectorBase &v) : std::vector<float>(v.Dim()) { if (v.Dim()) v.CopyToHost(data()); }
  Vector(const VectorBase<Real> &v) : std::vector<float>(v.Data(), v.Data() + v.Dim()) {}
  Vector &operator=(const Vector &v) { static_cast<std::vector<float> &>(*this) = v; return *this; }
  Vector &operator=(const std::vector<float> &v) { static_cast<std::vector<float> &>(*this) = v; return *this; }
  Vector &operator=(const ::aslp::HostVector &v) { static_cast<std::vector<float> &>(*this) = v.data; return *this; }
  int Dim() const override { return (int)size(); }
  const float *Data() const override { return data(); }
  using VectorBase<Real>::Data;
  void Resize(int dim, ::aslp::MatrixResizeType t = ::aslp::kSetZero) {
    if (t == ::aslp::kCopyData) resize(dim, 0.0f);
    else assign(dim, 0.0f);
  }
  void Read(std::istream &is, bool binary) { ::aslp::HostVector v; v.Read(is, binary); *this = v; }
  void Write(std::ostream &os, bool binary) const { ::aslp::HostVector v; v.data = *this; v.Write(os, binary); }
};

/* ---- matrices ------------------------------------------------------------------------------------------------------------------------ */
template <typename Real>
class MatrixBase {
 public:
  virtual ~MatrixBase() {}
  virtual int NumRows() const = 0;
  virtual int NumCols() const = 0;
  virtual int Stride() const = 0;
  virtual const float *Data() const = 0;
  float *Data() { return const_cast<float *>(static_cast<const MatrixBase *>(this)->Data()); }
  float *RowData(int r) { return Data() + (size_t)r * Stride(); }
  const float *RowData(int r) const { return Data() + (size_t)r * Stride(); }
  float &operator()(int r, int c) { return RowData(r)[c]; }
  float operator()(int r, int c) const { return RowData(r)[c]; }
  SubVector<Real> Row(int r) const { return SubVector<Real>(RowData(r), NumCols()); }
  void SetZero() { Set(0.0f); }
  void Set(float v) { for (int i = 0; i < NumRows(); i++) std::fill(RowData(i), RowData(i) + NumCols(), v); }
  void Scale(float a) { for (int i = 0; i < NumRows(); i++) for (int j = 0; j < NumCols(); j++) RowData(i)[j] *= a; }
  float Sum() const { double s = 0.0; for (int i = 0; i < NumRows(); i++) for (int j = 0; j < NumCols(); j++) s += RowData(i)[j]; return (float)s; }
  float Min() const { float m = NumRows() && NumCols() ? RowData(0)[0] : 0.0f; for (int i = 0; i < NumRows(); i++) for (int j = 0; j < NumCols(); j++) m = std::min(m, RowData(i)[j]); return m; }
  float Max() const { float m = NumRows() && NumCols() ? RowData(0)[0] : 0.0f; for (int i = 0; i < NumRows(); i++) for (int j = 0; j < NumCols(); j++) m = std::max(m, RowData(i)[j]); return m; }
  /* sigma_max / sigma_min (matrix/kaldi-matrix.h Cond, there through an SVD): from the eigenvalues of the smaller Gram matrix, cyclic Jacobi in double */
  float Cond() const {
    const int R = NumRows(), C = NumCols(), n = std::min(R, C);
    if (n == 0) return 0.0f;
    std::vector<double> g((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++)
      for (int j = i; j < n; j++) {
        double s = 0.0;
        if (C <= R) for (int k = 0; k < R; k++) s += (double)RowData(k)[i] * RowData(k)[j];
        else for (int k = 0; k < C; k++) s += (double)RowData(i)[k] * RowData(j)[k];
        g[(size_t)i * n + j] = g[(size_t)j * n + i] = s;
      }
    for (int sweep = 0; sweep < 60; sweep++) {
      double off = 0.0;
      for (int p = 0; p < n; p++) for (int q = p + 1; q < n; q++) off += g[(size_t)p * n + q] * g[(size_t)p * n + q];
      if (off < 1e-30) break;
      for (int p = 0; p < n; p++)
        for (int q = p + 1; q < n; q++) {
          const double apq = g[(size_t)p * n + q];
          if (apq == 0.0) continue;
          const double theta = (g[(size_t)q * n + q] - g[(size_t)p * n + p]) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0)), c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
          for (int k = 0; k < n; k++) {
            const double gkp = g[(size_t)k * n + p], gkq = g[(size_t)k * n + q];
            g[(size_t)k * n + p] = c * gkp - sn * gkq; g[(size_t)k * n + q] = sn * gkp + c * gkq;
          }
          for (int k = 0; k < n; k++) {
            const double gpk = g[(size_t)p * n + k], gqk = g[(size_t)q * n + k];
            g[(size_t)p * n + k] = c * gpk - sn * gqk; g[(size_t)q * n + k] = sn * gpk + c * gqk;
          }
        }
    }
    double lo = g[0], hi = g[0];
    for (int i = 0; i < n; i++) { lo = std::min(lo, g[(size_t)i * n + i]); hi = std::max(hi, g[(size_t)i * n + i]); }
    return lo > 0.0 ? (float)std::sqrt(hi / lo) : INFINITY;
  }
  void CopyRowFromVec(const VectorBase<Real> &v, int r) { ASLP_ASSERT(v.Dim() == NumCols()); std::memmove(RowData(r), v.Data(), sizeof(float) * NumCols()); }
  void CopyFromMat(const MatrixBase &m) {
    ASLP_ASSERT(m.NumRows() == NumRows() && m.NumCols() == NumCols());
    for (int i = 0; i < NumRows(); i++) std::memmove(RowData(i), m.RowData(i), sizeof(float) * NumCols());
  }
  SubMatrix<Real> Range(int r0, int nr, int c0, int nc) const;
  SubMatrix<Real> RowRange(int r0, int nr) const;
  SubMatrix<Real> ColRange(int c0, int nc) const;
};

/* a block of a matrix (rows r0.., columns c0..), row stride the parent's */
template <typename Real>
class SubMatrix : public MatrixBase<Real> {
 public:
  SubMatrix(const MatrixBase<Real> &m, int r0, int nr, int c0, int nc) : data_(m.RowData(r0) + c0), rows_(nr), cols_(nc), stride_(m.Stride()) {
    ASLP_ASSERT(r0 >= 0 && nr >= 0 && r0 + nr <= m.NumRows() && c0 >= 0 && nc >= 0 && c0 + nc <= m.NumCols());
  }
  int NumRows() const override { return rows_; }
  int NumCols() const override { return cols_; }
  int Stride() const override { return stride_; }
  const float *Data() const override { return data_; }
  using MatrixBase<Real>::Data;
 private:
  const float *data_;
  int rows_, cols_, stride_;
};
template <typename Real> SubMatrix<Real> MatrixBase<Real>::Range(int r0, int nr, int c0, int nc) const { return SubMatrix<Real>(*this, r0, nr, c0, nc); }
template <typename Real> SubMatrix<Real> MatrixBase<Real>::RowRange(int r0, int nr) const { return SubMatrix<Real>(*this, r0, nr, 0, NumCols()); }
template <typename Real> SubMatrix<Real> MatrixBase<Real>::ColRange(int c0, int nc) const { return SubMatrix<Real>(*this, 0, NumRows(), c0, nc); }

template <typename Real>
class Matrix : public ::aslp::HostMatrix, public MatrixBase<Real> {
 public:
  Matrix() {}
  Matrix(int r, int c, ::aslp::MatrixResizeType = ::aslp::kSetZero) : ::aslp::HostMatrix(r, c) {}
  Matrix(const Matrix &m) : ::aslp::HostMatrix(m), MatrixBase<Real>() {}
  Matrix(const ::aslp::HostMatrix &m) : ::aslp::HostMatrix(m) {}
  explicit Matrix(const ::aslp::CuMatrixBase &m) { m.CopyToMat(this); }
  Matrix(const MatrixBase<Real> &m) : ::aslp::HostMatrix(m.NumRows(), m.NumCols()) { MatrixBase<Real>::CopyFromMat(m); }
  Matrix &operator=(const Matrix &m) { static_cast<::aslp::HostMatrix &>(*this) = m; return *this; }
  Matrix &operator=(const ::aslp::HostMatrix &m) { static_cast<::aslp::HostMatrix &>(*this) = m; return *this; }
  Matrix &operator=(const MatrixBase<Real> &m) { Matrix t(m); Swap(&t); return *this; }
  int NumRows() const override { return rows; }
  int NumCols() const override { return cols; }
  int Stride() const override { return cols; }
  const float *Data() const override { return data.data(); }
  using MatrixBase<Real>::Data;
  using MatrixBase<Real>::operator();
  void Swap(Matrix *o) { std::swap(rows, o->rows); std::swap(cols, o->cols); data.swap(o->data); }
  void Resize(int r, int c, ::aslp::MatrixResizeType t = ::aslp::kSetZero) {
    if (t != ::aslp::kCopyData) { ::aslp::HostMatrix::Resize(r, c); return; }
    Matrix n(r, c);   /* matrix/kaldi-matrix.cc Resize(kCopyData): the overlapping block stays, the rest is zero */
    const int rr = std::min(r, rows), cc = std::min(c, cols);
    for (int i = 0; i < rr; i++) std::memcpy(n.RowData(i), this->RowData(i), sizeof(float) * cc);
    Swap(&n);
  }
  void Read(std::istream &is, bool binary) { ::aslp::HostMatrix::Read(is, binary); }
  void Write(std::ostream &os, bool binary) const { ::aslp::HostMatrix::Write(os, binary); }
};

/* matrix/kaldi-matrix.h AssertEqual: ||A - B|| <= tol ||A|| (Frobenius) */
template <typename Real>
void AssertEqual(const MatrixBase<Real> &A, const MatrixBase<Real> &B, float tol = 0.01) {
  ASLP_ASSERT(A.NumRows() == B.NumRows() && A.NumCols() == B.NumCols());
  double d = 0.0, a = 0.0;
  for (int i = 0; i < A.NumRows(); i++)
    for (int j = 0; j < A.NumCols(); j++) { const double x = A(i, j), y = B(i, j); d += (x - y) * (x - y); a += x * x; }
  ASLP_ASSERT(std::sqrt(d) <= tol * std::sqrt(a));
}

}  // namespace kaldi

#endif  /* ASLP_COMPAT_KALDI_MATRIX_H_ */
