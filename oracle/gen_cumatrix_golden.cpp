// gen_cumatrix_golden.cpp -- TEST INFRASTRUCTURE.  Runs the REFERENCE's own CuMatrix CPU branch (src/aslp-cudamatrix with
// HAVE_CUDA undefined, on top of src/matrix) on seeded inputs and writes inputs + outputs to tests/golden/cumatrix_ops.bin.
// Built by `make -C oracle ref` from the reference sources where they lie; only BLAS-free operations are called (the image has
// no BLAS and none is faked): activations and their derivatives, element-wise ops, row arg-max, DiffXent, the index ops
// Splice / Randomize / Copy (CopyRows is not here: its CPU branch goes through cblas_scopy).  Record format: char name[32]; int32 rows, cols, kind (0 = float32, 1 = int32); data.
#include <cstdio>
#include <cstring>
#include <vector>

#include "aslp-cudamatrix/cu-array.h"
#include "aslp-cudamatrix/cu-math.h"
#include "aslp-cudamatrix/cu-matrix.h"
#include "aslp-cudamatrix/cu-vector.h"

using namespace kaldi;

static FILE *g_out;
static void Put(const char *name, int rows, int cols, int kind, const void *data) {
  char nm[32];
  std::memset(nm, 0, sizeof(nm));
  std::strncpy(nm, name, 31);
  std::fwrite(nm, 1, 32, g_out);
  int32 hdr[3] = {rows, cols, kind};
  std::fwrite(hdr, sizeof(int32), 3, g_out);
  std::fwrite(data, 4, (size_t)rows * cols, g_out);
}
static void PutMat(const char *name, const CuMatrixBase<float> &m) {
  Matrix<float> h(m.NumRows(), m.NumCols());
  m.CopyToMat(&h);
  std::vector<float> flat((size_t)h.NumRows() * h.NumCols());
  for (int r = 0; r < h.NumRows(); r++) std::memcpy(&flat[(size_t)r * h.NumCols()], h.RowData(r), sizeof(float) * h.NumCols());
  Put(name, h.NumRows(), h.NumCols(), 0, flat.data());
}
static void PutInts(const char *name, const std::vector<int32> &v) { Put(name, 1, (int)v.size(), 1, v.data()); }

// own generator: the fixture must not depend on the C library's rand()
static unsigned long long g_state = 88172645463325252ull;
static float Uniform() {
  g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
  return (float)((g_state >> 40) * (1.0 / 16777216.0));
}
static void Fill(CuMatrix<float> *m, int rows, int cols, float lo, float hi) {
  Matrix<float> h(rows, cols);
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) h(r, c) = lo + (hi - lo) * Uniform();
  m->Resize(rows, cols);
  m->CopyFromMat(h);
}

int main(int argc, char **argv) {
  if (argc != 2) { std::fprintf(stderr, "usage: %s <out.bin>\n", argv[0]); return 1; }
  g_out = std::fopen(argv[1], "wb");
  if (!g_out) return 1;
  const int R = 37, C = 53;
  CuMatrix<float> x, y, d, t;
  Fill(&x, R, C, -12.0f, 12.0f);
  // a few extremes for the piecewise sigmoid / tanh of kaldi-vector.cc:885-936
  {
    Matrix<float> h(R, C);
    x.CopyToMat(&h);
    const float ext[8] = {0.0f, -0.0f, 1e-8f, -1e-8f, 40.0f, -40.0f, 88.0f, -88.0f};
    for (int i = 0; i < 8; i++) h(0, i) = ext[i];
    x.CopyFromMat(h);
  }
  PutMat("x", x);
  y.Resize(R, C); y.Sigmoid(x); PutMat("sigmoid", y);
  Fill(&d, R, C, -2.0f, 2.0f); PutMat("d", d);
  t.Resize(R, C); t.DiffSigmoid(y, d); PutMat("diff_sigmoid", t);
  y.Tanh(x); PutMat("tanh", y);
  t.DiffTanh(y, d); PutMat("diff_tanh", t);
  t.CopyFromMat(x); t.ApplyFloor(0.0); PutMat("relu", t);
  t.CopyFromMat(x); t.ApplyHeaviside(); PutMat("heaviside", t);
  t.CopyFromMat(x); t.ApplyFloor(-1.5); t.ApplyCeiling(2.5); PutMat("floor_ceil", t);
  t.CopyFromMat(x); t.MulElements(d); PutMat("mul_elements", t);
  t.CopyFromMat(x); t.ApplyPow(2.0); PutMat("pow2", t);
  {
    CuMatrix<float> p;
    Fill(&p, R, C, 0.01f, 9.0f); PutMat("pos", p);
    t.CopyFromMat(p); t.ApplyLog(); PutMat("log", t);
    t.CopyFromMat(d); t.ApplyExp(); PutMat("exp", t);
    t.CopyFromMat(p); t.InvertElements(); PutMat("invert", t);
  }
  {  // row arg-max (first maximum wins) incl. ties
    CuMatrix<float> m;
    Fill(&m, R, C, 0.0f, 1.0f);
    Matrix<float> h(R, C); m.CopyToMat(&h);
    h(3, 5) = 2.0f; h(3, 20) = 2.0f;  // tie
    h(4, C - 1) = 3.0f;
    m.CopyFromMat(h);
    PutMat("argmax_in", m);
    CuArray<int32> id;
    m.FindRowMaxId(&id);
    std::vector<int32> hv; id.CopyToVec(&hv);
    PutInts("argmax", hv);
  }
  {  // DiffXent: log_post[r] = log(y[r][tgt]); y[r][tgt] -= 1
    CuMatrix<float> p;
    Fill(&p, R, C, 0.001f, 1.0f); PutMat("xent_in", p);
    std::vector<int32> tgt(R);
    for (int r = 0; r < R; r++) tgt[r] = (int)(Uniform() * C) % C;
    PutInts("xent_tgt", tgt);
    CuArray<int32> ctgt(tgt);
    CuVector<float> logpost(R);
    p.DiffXent(ctgt, &logpost);
    PutMat("xent_diff", p);
    CuMatrix<float> lp(1, R); lp.Row(0).CopyFromVec(logpost); PutMat("xent_logpost", lp);
  }
  {  // Splice (nnet-various.h Splice component), Copy (column gather), Randomize (row gather by mask)
    const int T = 29, D = 11;
    CuMatrix<float> f;
    Fill(&f, T, D, -3.0f, 3.0f); PutMat("splice_in", f);
    std::vector<int32> off = {-5, -2, -1, 0, 1, 3, 7};
    PutInts("splice_off", off);
    CuArray<int32> coff(off);
    CuMatrix<float> o(T, D * (int)off.size());
    cu::Splice(f, coff, &o); PutMat("splice_out", o);
    std::vector<int32> cols = {10, 0, 3, 3, 7, 1};
    PutInts("copy_cols", cols);
    CuArray<int32> ccols(cols);
    CuMatrix<float> oc(T, (int)cols.size());
    cu::Copy(f, ccols, &oc); PutMat("copy_out", oc);
    std::vector<int32> mask(T);
    for (int i = 0; i < T; i++) mask[i] = (i * 12 + 5) % T;  // a permutation (gcd(12, 29) = 1)
    PutInts("rand_mask", mask);
    CuArray<int32> cmask(mask);
    CuMatrix<float> orr(T, D);
    cu::Randomize(f, cmask, &orr); PutMat("randomize_out", orr);
  }
  std::fclose(g_out);
  return 0;
}
