// host-matrix.h -- host-side matrix / vector used for I/O and parameter exchange (replaces kaldi::Matrix / Vector there).
// Host-only: no device types, so archive / table tools can use it without the HIP library.
#pragma once
#include <vector>

#include "base.h"

namespace aslp {

struct HostMatrix {
  int rows = 0, cols = 0;
  std::vector<float> data;
  HostMatrix() {}
  HostMatrix(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0.0f) {}
  float &operator()(int r, int c) { return data[(size_t)r * cols + c]; }
  float operator()(int r, int c) const { return data[(size_t)r * cols + c]; }
  void Resize(int r, int c) { rows = r; cols = c; data.assign((size_t)r * c, 0.0f); }
  void Read(std::istream &is, bool binary);
  void Write(std::ostream &os, bool binary) const;
};
// A reader whose matrices go straight on to the device (FrameDataReader's parsing thread) wants the payload of a binary float matrix in
// page-locked memory, not in `data`: with a sink installed for the calling thread, HostMatrix::Read hands (rows, cols) to the sink, reads the
// payload into the pointer it returns and leaves `data` EMPTY (rows / cols are set).  Other encodings (double, compressed, text) ignore the
// sink and fill `data` as always.  One pass over the bytes instead of three (zero-fill, read, copy).
struct HostMatrixSink {
  float *(*take)(void *ctx, int rows, int cols) = nullptr;
  void *ctx = nullptr;
};
HostMatrixSink &host_matrix_sink();   // the calling thread's

struct HostVector {
  std::vector<float> data;
  HostVector() {}
  explicit HostVector(int n) : data(n, 0.0f) {}
  int Dim() const { return (int)data.size(); }
  void Read(std::istream &is, bool binary);
  void Write(std::ostream &os, bool binary) const;
};
struct HostVectorD {
  std::vector<double> data;
  int Dim() const { return (int)data.size(); }
  void Read(std::istream &is, bool binary);
  void Write(std::ostream &os, bool binary) const;
};


}  // namespace aslp
