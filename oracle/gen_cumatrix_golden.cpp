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
  // ---- round 2: the BLAS-free arithmetic under FSMN, the LSTM peepholes, the affine L1 step and the bias / scale broadcasts.
  // Appended AFTER the records above so those stay byte-identical (the generator stream just continues).
  {  // AddConvMatMatElements (cu-matrix.cc:3037-3073 -> kaldi-matrix.cc:483-501): CompactFsmn's product; and its building block
    const int a_rows = 19, b_rows = 7, W = 23;
    CuMatrix<float> A, B, dst;
    Fill(&A, a_rows, W, -2.0f, 2.0f); PutMat("conv_A", A);
    Fill(&B, b_rows, W, -1.0f, 1.0f); PutMat("conv_B", B);
    Fill(&dst, (a_rows - b_rows + 1) * b_rows, W, -1.0f, 1.0f); PutMat("conv_dst_in", dst);
    dst.AddConvMatMatElements(0.7f, A, B, 0.3f); PutMat("conv_dst_out", dst);
    CuMatrix<float> z((a_rows - b_rows + 1) * b_rows, W);
    z.AddConvMatMatElements(1.0f, A, B, 0.0f); PutMat("conv_dst_beta0", z);
    CuMatrix<float> e, f, g;
    Fill(&e, R, C, -2.0f, 2.0f); PutMat("mme_A", e);
    Fill(&f, R, C, -2.0f, 2.0f); PutMat("mme_B", f);
    Fill(&g, R, C, -2.0f, 2.0f); PutMat("mme_dst_in", g);
    g.AddMatMatElements(-1.25f, e, f, 0.5f); PutMat("mme_dst_out", g);
  }
  {  // AddMatDiagVec with beta = 1 (the peephole term: YGIFO += YC * diag(peephole), lc.h:585-590), both orientations of M
    CuMatrix<float> m, m2, dst;
    CuVector<float> v(C);
    { Vector<float> hv(C); for (int i = 0; i < C; i++) hv(i) = -1.0f + 2.0f * Uniform(); v.CopyFromVec(hv);
      CuMatrix<float> vm(1, C); vm.Row(0).CopyFromVec(v); PutMat("mdv_vec", vm); }
    Fill(&m, R, C, -2.0f, 2.0f); PutMat("mdv_M", m);
    Fill(&dst, R, C, -2.0f, 2.0f); PutMat("mdv_dst_in", dst);
    dst.AddMatDiagVec(0.75f, m, kNoTrans, v, 1.0f); PutMat("mdv_dst_out", dst);
    Fill(&m2, C, R, -2.0f, 2.0f); PutMat("mdv_Mt", m2);
    Fill(&dst, R, C, -2.0f, 2.0f); PutMat("mdv_dst_in_t", dst);
    dst.AddMatDiagVec(-0.5f, m2, kTrans, v, 1.0f); PutMat("mdv_dst_out_t", dst);
  }
  {  // bias / scale broadcasts on <= 64 rows / columns (beyond that the CPU branch goes through BLAS): AddVecToRows, AddVecToCols
     // with beta = 1, MulColsVec, MulRowsVec
    const int r2 = 41, c2 = 59;
    CuMatrix<float> m;
    CuVector<float> row(c2), col(r2);
    { Vector<float> h(c2); for (int i = 0; i < c2; i++) h(i) = -3.0f + 6.0f * Uniform(); row.CopyFromVec(h);
      CuMatrix<float> t2(1, c2); t2.Row(0).CopyFromVec(row); PutMat("bc_row", t2); }
    { Vector<float> h(r2); for (int i = 0; i < r2; i++) h(i) = -3.0f + 6.0f * Uniform(); col.CopyFromVec(h);
      CuMatrix<float> t2(1, r2); t2.Row(0).CopyFromVec(col); PutMat("bc_col", t2); }
    Fill(&m, r2, c2, -2.0f, 2.0f); PutMat("bc_in", m);
    CuMatrix<float> t2(m);
    t2.AddVecToRows(0.5f, row, 1.0f); PutMat("add_vec_to_rows", t2);
    t2.CopyFromMat(m); t2.AddVecToCols(-1.5f, col, 1.0f); PutMat("add_vec_to_cols", t2);
    t2.CopyFromMat(m); t2.MulColsVec(row); PutMat("mul_cols_vec", t2);
    t2.CopyFromMat(m); t2.MulRowsVec(col); PutMat("mul_rows_vec", t2);
  }
  {  // cu::RegularizeL1 (cu-math.cc:37-75): zero weights are skipped, a sign change clamps weight AND gradient to zero
    CuMatrix<float> w, gr;
    Fill(&w, R, C, -0.01f, 0.01f);
    Fill(&gr, R, C, -1.0f, 1.0f);
    Matrix<float> h(R, C); w.CopyToMat(&h);
    for (int i = 0; i < C; i += 5) h(1, i) = 0.0f;        // exact zeros
    for (int i = 0; i < C; i += 3) h(2, i) *= 1e-3f;      // near zero: the step overshoots
    w.CopyFromMat(h);
    PutMat("l1_w_in", w); PutMat("l1_g_in", gr);
    cu::RegularizeL1(&w, &gr, 0.002f, 0.01f);
    PutMat("l1_w_out", w); PutMat("l1_g_out", gr);
  }
  {  // CopyCols / AddCols (kaldi-matrix.cc:2561-2610): -1 = zero / skip
    const int T = 29, D = 11;
    CuMatrix<float> f;
    Fill(&f, T, D, -3.0f, 3.0f); PutMat("cols_in", f);
    std::vector<int32> idx = {10, -1, 3, 3, 0, 7, -1, 1, 5};
    PutInts("cols_idx", idx);
    CuArray<int32> cidx(idx);
    CuMatrix<float> o;
    Fill(&o, T, (int)idx.size(), -1.0f, 1.0f); PutMat("cols_dst_in", o);
    CuMatrix<float> o2(o);
    o2.CopyCols(f, cidx); PutMat("copy_cols_out", o2);
    o.AddCols(f, cidx); PutMat("add_cols_out", o);
  }
  std::fclose(g_out);
  return 0;
}
