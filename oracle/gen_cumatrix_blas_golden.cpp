// gen_cumatrix_blas_golden.cpp -- TEST INFRASTRUCTURE.  Runs the REFERENCE's own CuMatrix / CuVector CPU branch (src/aslp-cudamatrix
// with HAVE_CUDA undefined, on top of src/matrix) INCLUDING its BLAS-backed operations and writes inputs + outputs to
// tests/golden/cumatrix_blas_ops.bin (same record format as gen_cumatrix_golden.cpp).
//
// The BLAS is a real one: the OpenBLAS build that ships inside this image's scipy wheel (scipy.libs/libscipy_openblas-*.so, LP64,
// CBLAS + LAPACK).  That build exports every routine under a vendor prefix (scipy_cblas_sgemm, scipy_sgesvd_, ...); the Makefile
// derives one -Dname=scipy_name per exported routine from `nm -D` of the library, so that the reference's sources (compiled where they
// lie, with the CLAPACK headers of the reference tree) bind to it.  Nothing is written in place of a library or a header.
//
// Part 1 -- operations: AddMatMat in the four operand layouts (cu-matrix.cc:1027-1061 -> cblas_sgemm), ApplySoftMaxPerRow
// (cu-matrix.cc:1351-1371 -> kaldi-vector.cc:852-859), CuVector::AddRowSumMat / AddColSumMat in float and double, the bias broadcasts
// beyond 64 columns / rows (the cblas_sger branch of kaldi-matrix.cc:2749-2799), CuMatrix::AddRowSumMat (cu-matrix.cc:3010-3034).
// Part 2 -- the op sequences of the components, issued by this driver against the reference's library in the order the component
// headers issue them (those headers themselves need OpenFst's fst/fst-decl.h through nnet-utils.h and cannot be compiled here):
// AffineTransform (nnet-affine-transform.h:186-245), BatchNormalization (nnet-batch-normalization.h:177-284), LstmProjectedStreams
// (nnet-lstm-projected-streams.h:313-617), GruStreams (nnet-gru-streams.h:238-450), the two directions of BLstmProjectedStreamsLC
// (nnet-blstm-projected-streams-lc.h:503-1040), LstmCifgProjectedStreams (nnet-lstm-couple-if-projected-streams.h), Lstm / BLstm (nnet-recurrent-component.cc),
// RowConvolution (nnet-row-convolution.cc:90-169), CompactFsmn (nnet-cfsmn-component.h:169-264), Xent::Eval
// (nnet-loss.cc:63-156).  The sequences are this file's reading of those lines; the arithmetic of every step is
// the reference's own.
#include <cstdio>
#include <cstring>
#include <vector>

#include "aslp-cudamatrix/cu-array.h"
#include "aslp-cudamatrix/cu-math.h"
#include "aslp-cudamatrix/cu-matrix.h"
#include "aslp-cudamatrix/cu-vector.h"

using namespace kaldi;
typedef CuMatrix<float> Mat;
typedef CuSubMatrix<float> Sub;
typedef CuVector<float> Vec;

static FILE *g_out;
static void Put(const char *name, int rows, int cols, int kind, const void *data, int elem = 4) {
  char nm[32];
  std::memset(nm, 0, sizeof(nm));
  std::strncpy(nm, name, 31);
  std::fwrite(nm, 1, 32, g_out);
  int32 hdr[3] = {rows, cols, kind};
  std::fwrite(hdr, sizeof(int32), 3, g_out);
  std::fwrite(data, elem, (size_t)rows * cols, g_out);
}
// digest mode (the full-width fixture: tensors of megabytes): a matrix goes out as every g_stride-th element of its row-major image plus a
// float64 record `name#` = {sum, sum of squares, element count}; inputs are not written at all (g_skip_inputs) -- the reader regenerates
// them from the generator state recorded in front (`<tag>_rng`), which is the whole reason the fixture stays small
static int g_stride = 0;
static bool g_skip = false;   // set around the PutMat calls of tensors the reader regenerates
static void PutMat(const char *name, const CuMatrixBase<float> &m) {
  if (g_skip) return;
  Matrix<float> h(m.NumRows(), m.NumCols());
  m.CopyToMat(&h);
  std::vector<float> flat((size_t)h.NumRows() * h.NumCols());
  for (int r = 0; r < h.NumRows(); r++) std::memcpy(&flat[(size_t)r * h.NumCols()], h.RowData(r), sizeof(float) * h.NumCols());
  if (g_stride <= 0) { Put(name, h.NumRows(), h.NumCols(), 0, flat.data()); return; }
  std::vector<float> pick;
  double acc[3] = {0.0, 0.0, (double)flat.size()};
  for (size_t i = 0; i < flat.size(); i++) {
    acc[0] += flat[i];
    acc[1] += (double)flat[i] * flat[i];
    if (i % (size_t)g_stride == 0) pick.push_back(flat[i]);
  }
  Put(name, 1, (int)pick.size(), 0, pick.data());
  char nm[32];
  std::snprintf(nm, 32, "%s#", name);
  Put(nm, 1, 3, 2, acc, 8);
}
static void PutVec(const char *name, const CuVectorBase<float> &v) {
  if (g_skip) return;
  Vector<float> h(v.Dim());
  v.CopyToVec(&h);
  Put(name, 1, v.Dim(), 0, h.Data());
}
static void PutVecD(const char *name, const CuVectorBase<double> &v) {  // kind 2 = float64
  Vector<double> h(v.Dim());
  v.CopyToVec(&h);
  Put(name, 1, v.Dim(), 2, h.Data(), 8);
}

static unsigned long long g_state = 0x9E3779B97F4A7C15ull;
static float Uniform() {
  g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
  return (float)((g_state >> 40) * (1.0 / 16777216.0));
}
static void Fill(Mat *m, int rows, int cols, float lo, float hi) {
  Matrix<float> h(rows, cols);
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) h(r, c) = lo + (hi - lo) * Uniform();
  m->Resize(rows, cols);
  m->CopyFromMat(h);
}
static void FillVec(Vec *v, int n, float lo, float hi) {
  Vector<float> h(n);
  for (int i = 0; i < n; i++) h(i) = lo + (hi - lo) * Uniform();
  v->Resize(n);
  v->CopyFromVec(h);
}

static void Operations() {
  {  // AddMatMat: C = alpha op(A) op(B) + beta C, all four layouts, two shapes (ragged; long reduction)
    const int shapes[2][3] = {{67, 45, 83}, {33, 20, 1000}};
    const char *lay[4] = {"nn", "nt", "tn", "tt"};
    for (int s = 0; s < 2; s++) {
      const int M = shapes[s][0], N = shapes[s][1], K = shapes[s][2];
      for (int l = 0; l < 4; l++) {
        const bool ta = l >= 2, tb = l & 1;
        Mat A, B, Cm;
        Fill(&A, ta ? K : M, ta ? M : K, -2.0f, 2.0f);
        Fill(&B, tb ? N : K, tb ? K : N, -2.0f, 2.0f);
        Fill(&Cm, M, N, -1.0f, 1.0f);
        char nm[32];
        std::snprintf(nm, 32, "gemm%d_%s_A", s, lay[l]); PutMat(nm, A);
        std::snprintf(nm, 32, "gemm%d_%s_B", s, lay[l]); PutMat(nm, B);
        std::snprintf(nm, 32, "gemm%d_%s_Cin", s, lay[l]); PutMat(nm, Cm);
        Cm.AddMatMat(0.7f, A, ta ? kTrans : kNoTrans, B, tb ? kTrans : kNoTrans, 0.3f);
        std::snprintf(nm, 32, "gemm%d_%s_Cout", s, lay[l]); PutMat(nm, Cm);
      }
    }
  }
  {  // softmax over rows, 300 classes, logits in [-9, 9] and one row with a dominating class
    Mat x, y;
    Fill(&x, 37, 300, -9.0f, 9.0f);
    Matrix<float> h(37, 300); x.CopyToMat(&h);
    h(5, 17) = 60.0f;
    for (int c = 0; c < 300; c++) h(6, c) = 0.25f;   // uniform row
    x.CopyFromMat(h);
    PutMat("softmax_in", x);
    y.Resize(37, 300);
    y.ApplySoftMaxPerRow(x);
    PutMat("softmax_out", y);
  }
  {  // column sums / row sums into vectors, float and double (the BatchNormalization statistics take the double ones)
    Mat m;
    Fill(&m, 200, 150, -3.0f, 5.0f); PutMat("sum_in", m);
    Vec v; FillVec(&v, 150, -1.0f, 1.0f); PutVec("colsum_v_in", v);
    v.AddRowSumMat(0.43f, m, 1.4f); PutVec("colsum_v_out", v);
    Vec w; FillVec(&w, 200, -1.0f, 1.0f); PutVec("rowsum_v_in", w);
    w.AddColSumMat(-0.6f, m, 0.5f); PutVec("rowsum_v_out", w);
    CuVector<double> vd(150);
    vd.AddRowSumMat(1.0, CuMatrix<double>(m), 1.0);
    Mat sq(200, 150);
    sq.AddMatMatElements(1.0, m, m, 0.0);
    CuVector<double> vq(150);
    vq.AddRowSumMat(1.0, CuMatrix<double>(sq), 1.0);
    PutVecD("colsum_d", vd); PutVecD("colsumsq_d", vq);
  }
  {  // bias broadcasts beyond 64 columns / rows
    Mat m, t;
    Fill(&m, 90, 130, -2.0f, 2.0f); PutMat("bc_in", m);
    Vec row, col;
    FillVec(&row, 130, -3.0f, 3.0f); PutVec("bc_row", row);
    FillVec(&col, 90, -3.0f, 3.0f); PutVec("bc_col", col);
    t = m; t.AddVecToRows(0.5f, row, 1.0f); PutMat("add_vec_to_rows", t);
    t = m; t.AddVecToRows(1.0f, row, 0.0f); PutMat("add_vec_to_rows_beta0", t);
    t = m; t.AddVecToCols(-1.5f, col, 1.0f); PutMat("add_vec_to_cols", t);
  }
  {  // CuMatrix::AddRowSumMat (ASLP): dst row r = beta dst_r + alpha * sum of the r-th group of src rows
    Mat src, dst;
    Fill(&src, 21 * 7, 6, -2.0f, 2.0f); PutMat("grp_src", src);
    Fill(&dst, 21, 6, -1.0f, 1.0f); PutMat("grp_dst_in", dst);
    dst.AddRowSumMat(0.8f, src, 0.25f); PutMat("grp_dst_out", dst);
  }
}

// AffineTransform: nnet-affine-transform.h:186-245 (Propagate, Backpropagate, Update with momentum, l2, max-norm), two minibatches
static void Affine() {
  const int rows = 70, din = 45, dout = 83;
  const float lr = 0.02f, lr_bias = 0.02f * 0.5f, mmt = 0.9f, l2 = 1e-3f, max_norm = 0.9f;
  Mat W, Wc(dout, din), in, out(rows, dout), od, id(rows, din);
  Vec b, bc(dout);
  Fill(&W, dout, din, -0.3f, 0.3f); FillVec(&b, dout, -1.0f, 1.0f);
  PutMat("aff_W0", W); PutVec("aff_b0", b);
  for (int step = 0; step < 2; step++) {
    char nm[32];
    Fill(&in, rows, din, -2.0f, 2.0f); Fill(&od, rows, dout, -1.0f, 1.0f);
    std::snprintf(nm, 32, "aff_in%d", step); PutMat(nm, in);
    std::snprintf(nm, 32, "aff_od%d", step); PutMat(nm, od);
    out.SetZero();                                            // Component::Propagate clears the output first
    out.AddVecToRows(1.0, b, 0.0);
    out.AddMatMat(1.0, in, kNoTrans, W, kTrans, 1.0);
    std::snprintf(nm, 32, "aff_out%d", step); PutMat(nm, out);
    id.AddMatMat(1.0, od, kNoTrans, W, kNoTrans, 0.0);
    std::snprintf(nm, 32, "aff_id%d", step); PutMat(nm, id);
    Wc.AddMatMat(1.0, od, kTrans, in, kNoTrans, mmt);
    bc.AddRowSumMat(1.0, od, mmt);
    W.AddMat(-lr * l2 * rows, W);
    W.AddMat(-lr, Wc);
    b.AddVec(-lr_bias, bc);
    {
      Mat sq(W);
      sq.MulElements(W);
      Vec nrm(dout);
      nrm.AddColSumMat(1.0, sq, 0.0);
      nrm.ApplyPow(0.5);
      Vec scl(nrm);
      scl.Scale(1.0 / max_norm);
      scl.ApplyFloor(1.0);
      scl.InvertElements();
      W.MulRowsVec(scl);
    }
    std::snprintf(nm, 32, "aff_W%d", step + 1); PutMat(nm, W);
    std::snprintf(nm, 32, "aff_b%d", step + 1); PutVec(nm, b);
    std::snprintf(nm, 32, "aff_Wc%d", step + 1); PutMat(nm, Wc);
    std::snprintf(nm, 32, "aff_bc%d", step + 1); PutVec(nm, bc);
  }
}

// BatchNormalization: nnet-batch-normalization.h:177-284, two minibatches (the running sums carry over, momentum on the second)
static void BatchNorm() {
  const int rows = 96, dim = 70;
  const float var_floor = 1e-7f, mmt = 0.9f, lr = 0.05f;   // var_floor_: nnet-batch-normalization.h (member initialiser)
  Vec scale, shift, mean(dim), var(dim), dmean(dim), dvar(dim), dshift(dim), dscale(dim);
  CuVector<double> acc_m(dim), acc_v(dim);
  FillVec(&scale, dim, 0.5f, 1.5f); FillVec(&shift, dim, -1.0f, 1.0f);
  PutVec("bn_scale0", scale); PutVec("bn_shift0", shift);
  Mat in, od, out(rows, dim), xs(rows, dim), e(rows, dim), id(rows, dim);
  for (int step = 0; step < 2; step++) {
    char nm[32];
    Fill(&in, rows, dim, -1.5f, 2.5f); Fill(&od, rows, dim, -1.0f, 1.0f);
    std::snprintf(nm, 32, "bn_in%d", step); PutMat(nm, in);
    std::snprintf(nm, 32, "bn_od%d", step); PutMat(nm, od);
    // forward :193-220
    mean.AddRowSumMat(1.0 / rows, in, 0.0);
    xs.CopyFromMat(in);
    xs.AddVecToRows(-1.0, mean, 1.0);
    out.AddMatMatElements(1.0, xs, xs, 0.0);
    var.AddRowSumMat(1.0 / rows, out, 0.0);
    var.Add(var_floor);
    var.ApplyPow(0.5);
    var.InvertElements();
    xs.MulColsVec(var);
    out.CopyFromMat(xs);
    out.MulColsVec(scale);
    out.AddVecToRows(1.0, shift, 1.0);
    acc_m.AddRowSumMat(1.0, CuMatrix<double>(in), 1.0);
    e.AddMatMatElements(1.0, in, in, 0.0);
    acc_v.AddRowSumMat(1.0, CuMatrix<double>(e), 1.0);
    std::snprintf(nm, 32, "bn_out%d", step); PutMat(nm, out);
    std::snprintf(nm, 32, "bn_xhat%d", step); PutMat(nm, xs);
    std::snprintf(nm, 32, "bn_mean%d", step); PutVec(nm, mean);
    std::snprintf(nm, 32, "bn_invstd%d", step); PutVec(nm, var);
    std::snprintf(nm, 32, "bn_accm%d", step); PutVecD(nm, acc_m);
    std::snprintf(nm, 32, "bn_accv%d", step); PutVecD(nm, acc_v);
    // backward :233-274
    const float m_step = step == 0 ? 0.0f : mmt;
    e.AddMatMatElements(1.0, xs, od, 0.0);
    dscale.AddRowSumMat(1.0, e, m_step);
    dshift.AddRowSumMat(1.0, od, m_step);
    xs.CopyFromMat(od);
    xs.MulColsVec(scale);
    dvar.CopyFromVec(var);
    dvar.ApplyPow(3);
    dvar.Scale(-0.5);
    e.CopyFromMat(in);
    e.AddVecToRows(-1.0, mean, 1.0);
    e.MulElements(xs);
    e.MulColsVec(dvar);
    dvar.AddRowSumMat(1.0, e, 0.0);
    e.CopyFromMat(xs);
    e.MulColsVec(var);
    e.Scale(-1.0);
    dmean.AddRowSumMat(1.0, e, 0.0);
    e.CopyFromMat(in);
    e.AddVecToRows(-1.0, mean);
    e.Scale(2.0 / rows);
    e.MulColsVec(dvar);
    dmean.AddRowSumMat(-1.0, e, 1.0);
    id.CopyFromMat(xs);
    id.MulColsVec(var);
    id.AddMat(1.0, e);
    id.AddVecToRows(1.0 / rows, dmean, 1.0);
    std::snprintf(nm, 32, "bn_id%d", step); PutMat(nm, id);
    std::snprintf(nm, 32, "bn_dscale%d", step); PutVec(nm, dscale);
    std::snprintf(nm, 32, "bn_dshift%d", step); PutVec(nm, dshift);
    // update :280-284
    scale.AddVec(-lr, dscale, 1.0);
    shift.AddVec(-lr, dshift, 1.0);
    std::snprintf(nm, 32, "bn_scale%d", step + 1); PutVec(nm, scale);
    std::snprintf(nm, 32, "bn_shift%d", step + 1); PutVec(nm, shift);
  }
}

// The projected-LSTM gate block: LstmProjectedStreams (nnet-lstm-projected-streams.h:313-617) = forward in time from a zero state;
// BLstmProjectedStreamsLC (nnet-blstm-projected-streams-lc.h:503-1040) = one direction forward in time from the state carried out of the
// previous chunk + one direction backward in time from zero.  Buffer columns g|i|f|o|c|h|m|r; row block t holds the S streams of
// frame t; block 0 / T+1 are the boundaries (carried state or zero).  `tag` prefixes the record names.
static void LstmProjected(const char *tag, bool reverse, bool carried, bool cifg = false, int R = 5, const int *lens = nullptr,
                          int T = 5, int S = 3, int D = 6, int C = 8, float wr = 0.4f, bool inputs_regenerated = false) {
  // lens (backward-in-time direction of the whole-utterance BLSTMs only): after a frame is computed, the buffer rows of the streams that
  // have already ended are cleared (nnet-blstm-projected-streams.h:654-657, nnet-recurrent-component.cc:1077-1080)
  // cifg: LstmCifgProjectedStreams (nnet-lstm-couple-if-projected-streams.h): no input gate, i = 1 - f, columns g|f|o|c|h|m|r.
  // R = 0: Lstm of nnet-recurrent-component.cc:235-420: no projection, the recurrence runs on m, columns g|i|f|o|c|h|m.
  // (T, S, D, C, wr, inputs_regenerated: the full-width digest records of main()'s second mode; the defaults are the small fixture's)
  const int NG = cifg ? 3 : 4, rec_w = R > 0 ? R : C, W = (NG + 3) * C + R;
  Mat Wx, Wr, Wrm, in, od;
  Vec bias, pi, pf, po;
  Fill(&Wx, NG * C, D, -wr, wr); Fill(&Wr, NG * C, rec_w, -wr, wr);
  if (R > 0) Fill(&Wrm, R, C, -wr, wr);
  FillVec(&bias, NG * C, -0.3f, 0.3f);
  if (!cifg) FillVec(&pi, C, -0.3f, 0.3f);
  FillVec(&pf, C, -0.3f, 0.3f); FillVec(&po, C, -0.3f, 0.3f);
  Fill(&in, T * S, D, -1.5f, 1.5f); Fill(&od, T * S, rec_w, -1.0f, 1.0f);
  char nm[32];
#define NAME(x) (std::snprintf(nm, 32, "%s_%s", tag, x), nm)
  g_skip = inputs_regenerated;
  PutMat(NAME("Wx"), Wx); PutMat(NAME("Wr"), Wr);
  if (R > 0) PutMat(NAME("Wrm"), Wrm);
  PutVec(NAME("bias"), bias);
  if (!cifg) PutVec(NAME("pi"), pi);
  PutVec(NAME("pf"), pf); PutVec(NAME("po"), po); PutMat(NAME("in"), in); PutMat(NAME("od"), od);
  Mat Y((T + 2) * S, W), Dd((T + 2) * S, W);
  if (carried) {   // f_propagate_buf_.RowRange(0, S).CopyFromMat(f_prev_nnet_state_): a whole row block of an earlier chunk
    Mat st;
    Fill(&st, S, W, -0.8f, 0.8f);
    PutMat(NAME("state"), st);
    Sub(Y, 0, S, 0, W).CopyFromMat(st);
  }
  g_skip = false;
  const int G = 0, I = cifg ? -1 : 1, F = cifg ? 1 : 2, O = NG - 1, Cc = NG, H = NG + 1, Mm = NG + 2;
  struct View {
    Mat &b; int C, R, S, NG;
    Sub gate(int k, int t) { return Sub(b, t * S, S, k * C, C); }
    Sub rec(int t) { return R > 0 ? Sub(b, t * S, S, (NG + 3) * C, R) : gate(NG + 2, t); }   // what the next frame recurs on: r, or m
    Sub gates(int t0, int n) { return Sub(b, t0 * S, n * S, 0, NG * C); }
    Sub cols(int k, int t0, int n) { return Sub(b, t0 * S, n * S, k * C, C); }
    Sub recs(int t0, int n) { return R > 0 ? Sub(b, t0 * S, n * S, (NG + 3) * C, R) : cols(NG + 2, t0, n); }
  } y = {Y, C, R, S, NG}, d = {Dd, C, R, S, NG};
  const int step = reverse ? -1 : 1;      // the frame a step depends on is t - step, the one that depends on it t + step
  y.gates(1, T).AddMatMat(1.0, in, kNoTrans, Wx, kTrans, 0.0);
  y.gates(1, T).AddVecToRows(1.0, bias);
  for (int n = 0, t = reverse ? T : 1; n < T; n++, t += step) {
    const int p = t - step;
    y.gates(t, 1).AddMatMat(1.0, y.rec(p), kNoTrans, Wr, kTrans, 1.0);
    if (!cifg) y.gate(I, t).AddMatDiagVec(1.0, y.gate(Cc, p), kNoTrans, pi, 1.0);
    y.gate(F, t).AddMatDiagVec(1.0, y.gate(Cc, p), kNoTrans, pf, 1.0);
    if (!cifg) y.gate(I, t).Sigmoid(y.gate(I, t));
    y.gate(F, t).Sigmoid(y.gate(F, t));
    y.gate(G, t).Tanh(y.gate(G, t));
    if (cifg) {   // c = g (1 - f) + c_prev f, spelled g - g f
      y.gate(Cc, t).AddMatMatElements(-1.0, y.gate(G, t), y.gate(F, t), 0.0);
      y.gate(Cc, t).AddMat(1.0, y.gate(G, t));
    } else {
      y.gate(Cc, t).AddMatMatElements(1.0, y.gate(G, t), y.gate(I, t), 0.0);
    }
    y.gate(Cc, t).AddMatMatElements(1.0, y.gate(Cc, p), y.gate(F, t), 1.0);
    y.gate(Cc, t).ApplyFloor(-50);
    y.gate(Cc, t).ApplyCeiling(50);
    y.gate(H, t).Tanh(y.gate(Cc, t));
    y.gate(O, t).AddMatDiagVec(1.0, y.gate(Cc, t), kNoTrans, po, 1.0);
    y.gate(O, t).Sigmoid(y.gate(O, t));
    y.gate(Mm, t).AddMatMatElements(1.0, y.gate(H, t), y.gate(O, t), 0.0);
    if (R > 0) y.rec(t).AddMatMat(1.0, y.gate(Mm, t), kNoTrans, Wrm, kTrans, 0.0);
    if (lens)
      for (int sidx = 0; sidx < S; sidx++)
        if (t > lens[sidx]) Sub(Y, t * S + sidx, 1, 0, W).SetZero();
  }
  if (lens) { std::vector<int32> lv(lens, lens + S); Put(NAME("lens"), 1, S, 1, lv.data()); }
  PutMat(NAME("fwd_buf"), Y);
  d.recs(1, T).CopyFromMat(od);
  for (int n = 0, t = reverse ? 1 : T; n < T; n++, t -= step) {
    const int p = t - step, q = t + step;
    d.rec(t).AddMatMat(1.0, d.gates(q, 1), kNoTrans, Wr, kNoTrans, 1.0);
    if (R > 0) d.gate(Mm, t).AddMatMat(1.0, d.rec(t), kNoTrans, Wrm, kNoTrans, 0.0);
    d.gate(H, t).AddMatMatElements(1.0, d.gate(Mm, t), y.gate(O, t), 0.0);
    d.gate(H, t).DiffTanh(y.gate(H, t), d.gate(H, t));
    d.gate(O, t).AddMatMatElements(1.0, d.gate(Mm, t), y.gate(H, t), 0.0);
    d.gate(O, t).DiffSigmoid(y.gate(O, t), d.gate(O, t));
    d.gate(Cc, t).AddMat(1.0, d.gate(H, t));
    d.gate(Cc, t).AddMatMatElements(1.0, d.gate(Cc, q), y.gate(F, q), 1.0);
    if (!cifg) d.gate(Cc, t).AddMatDiagVec(1.0, d.gate(I, q), kNoTrans, pi, 1.0);
    d.gate(Cc, t).AddMatDiagVec(1.0, d.gate(F, q), kNoTrans, pf, 1.0);
    d.gate(Cc, t).AddMatDiagVec(1.0, d.gate(O, t), kNoTrans, po, 1.0);
    d.gate(F, t).AddMatMatElements(1.0, d.gate(Cc, t), y.gate(Cc, p), 0.0);
    if (cifg) d.gate(F, t).AddMatMatElements(-1.0, d.gate(Cc, t), y.gate(G, t), 1.0);
    d.gate(F, t).DiffSigmoid(y.gate(F, t), d.gate(F, t));
    if (cifg) {
      d.gate(G, t).AddMatMatElements(-1.0, d.gate(Cc, t), y.gate(F, t), 0.0);
      d.gate(G, t).AddMat(1.0, d.gate(Cc, t));
    } else {
      d.gate(I, t).AddMatMatElements(1.0, d.gate(Cc, t), y.gate(G, t), 0.0);
      d.gate(I, t).DiffSigmoid(y.gate(I, t), d.gate(I, t));
      d.gate(G, t).AddMatMatElements(1.0, d.gate(Cc, t), y.gate(I, t), 0.0);
    }
    d.gate(G, t).DiffTanh(y.gate(G, t), d.gate(G, t));
  }
  PutMat(NAME("bwd_buf"), Dd);
  Mat id(T * S, D);
  id.AddMatMat(1.0, d.gates(1, T), kNoTrans, Wx, kNoTrans, 0.0);
  PutMat(NAME("in_diff"), id);
  // gradients (momentum 0, no clipping: clipping is ApplyFloor / ApplyCeiling, pinned in cumatrix_ops.bin); the frames a step
  // depended on are the row blocks 0..T-1 (forward in time) or 2..T+1 (backward in time)
  const int pb = reverse ? 2 : 0;
  Mat gWx(NG * C, D), gWr(NG * C, rec_w), gWrm(R > 0 ? R : 1, C);
  Vec gb(NG * C), gpi(C), gpf(C), gpo(C);
  gWx.AddMatMat(1.0, d.gates(1, T), kTrans, in, kNoTrans, 0.0);
  gWr.AddMatMat(1.0, d.gates(1, T), kTrans, y.recs(pb, T), kNoTrans, 0.0);
  gb.AddRowSumMat(1.0, d.gates(1, T), 0.0);
  if (!cifg) gpi.AddDiagMatMat(1.0, d.cols(I, 1, T), kTrans, y.cols(Cc, pb, T), kNoTrans, 0.0);
  gpf.AddDiagMatMat(1.0, d.cols(F, 1, T), kTrans, y.cols(Cc, pb, T), kNoTrans, 0.0);
  gpo.AddDiagMatMat(1.0, d.cols(O, 1, T), kTrans, y.cols(Cc, 1, T), kNoTrans, 0.0);
  if (R > 0) gWrm.AddMatMat(1.0, d.recs(1, T), kTrans, y.cols(Mm, 1, T), kNoTrans, 0.0);
  PutMat(NAME("gWx"), gWx); PutMat(NAME("gWr"), gWr);
  if (R > 0) PutMat(NAME("gWrm"), gWrm);
  PutVec(NAME("gb"), gb);
  if (!cifg) PutVec(NAME("gpi"), gpi);
  PutVec(NAME("gpf"), gpf); PutVec(NAME("gpo"), gpo);
#undef NAME
}

// Two TRAINING steps of the projected LSTM with momentum, element-wise gradient clipping and the component's Update, issued as the reference
// issues them: the corr buffers accumulate with AddMatMat / AddRowSumMat / AddDiagMatMat at beta = momentum (lc.h:976-998 =
// nnet-lstm-projected-streams.h:560-585), are clipped by ApplyFloor / ApplyCeiling (lc.h:1000-1016), then W.AddMat(-lr, corr)
// (lc.h:1085-1098).  Sized so that the engine's persistent recurrence serves it (C = 64, S = 8 streams): a golden from the reference's
// library reaches those kernels.  State zero at the start of every step (the streams are reset).
static void LstmProjectedTrain(const char *tag, int T, int S, int D, int C, int R, float mmt, float clip, float lr, float wr = 0.2f, float odr = 1.0f,
                               bool inputs_regenerated = false) {
  // wr: range of the weight matrices (0.02 at full width: cfg3's <ParamScale>; at 0.2 a 512-cell layer saturates and any two fp32 summation
  // orders part ways); odr: range of the out-diff; inputs_regenerated: initial parameters, inputs and out-diffs are not written (digest mode)
  const int NG = 4, W = (NG + 3) * C + R;
  Mat Wx, Wr, Wrm;
  Vec bias, pi, pf, po;
  Fill(&Wx, NG * C, D, -wr, wr); Fill(&Wr, NG * C, R, -wr, wr); Fill(&Wrm, R, C, -wr, wr);
  FillVec(&bias, NG * C, -0.3f, 0.3f); FillVec(&pi, C, -0.3f, 0.3f); FillVec(&pf, C, -0.3f, 0.3f); FillVec(&po, C, -0.3f, 0.3f);
  Mat cWx(NG * C, D), cWr(NG * C, R), cWrm(R, C);
  Vec cb(NG * C), cpi(C), cpf(C), cpo(C);
  char nm[32];
#define NAME2(x, k) (std::snprintf(nm, 32, "%s_%s%d", tag, x, k), nm)
  auto put_params = [&](int k) {
    PutMat(NAME2("Wx", k), Wx); PutMat(NAME2("Wr", k), Wr); PutMat(NAME2("Wrm", k), Wrm); PutVec(NAME2("bias", k), bias);
    PutVec(NAME2("pi", k), pi); PutVec(NAME2("pf", k), pf); PutVec(NAME2("po", k), po);
  };
  g_skip = inputs_regenerated;
  put_params(0);
  g_skip = false;
  const int G = 0, I = 1, F = 2, O = 3, Cc = 4, H = 5, Mm = 6;
  for (int step = 0; step < 2; step++) {
    Mat in, od;
    Fill(&in, T * S, D, -1.5f, 1.5f); Fill(&od, T * S, R, -odr, odr);
    g_skip = inputs_regenerated;
    PutMat(NAME2("in", step), in); PutMat(NAME2("od", step), od);
    g_skip = false;
    Mat Y((T + 2) * S, W), Dd((T + 2) * S, W);
    struct View {
      Mat &b; int C, R, S;
      Sub gate(int k, int t) { return Sub(b, t * S, S, k * C, C); }
      Sub rec(int t) { return Sub(b, t * S, S, 7 * C, R); }
      Sub gates(int t0, int n) { return Sub(b, t0 * S, n * S, 0, 4 * C); }
      Sub cols(int k, int t0, int n) { return Sub(b, t0 * S, n * S, k * C, C); }
      Sub recs(int t0, int n) { return Sub(b, t0 * S, n * S, 7 * C, R); }
    } y = {Y, C, R, S}, d = {Dd, C, R, S};
    y.gates(1, T).AddMatMat(1.0, in, kNoTrans, Wx, kTrans, 0.0);
    y.gates(1, T).AddVecToRows(1.0, bias);
    for (int t = 1; t <= T; t++) {
      const int p = t - 1;
      y.gates(t, 1).AddMatMat(1.0, y.rec(p), kNoTrans, Wr, kTrans, 1.0);
      y.gate(I, t).AddMatDiagVec(1.0, y.gate(Cc, p), kNoTrans, pi, 1.0);
      y.gate(F, t).AddMatDiagVec(1.0, y.gate(Cc, p), kNoTrans, pf, 1.0);
      y.gate(I, t).Sigmoid(y.gate(I, t));
      y.gate(F, t).Sigmoid(y.gate(F, t));
      y.gate(G, t).Tanh(y.gate(G, t));
      y.gate(Cc, t).AddMatMatElements(1.0, y.gate(G, t), y.gate(I, t), 0.0);
      y.gate(Cc, t).AddMatMatElements(1.0, y.gate(Cc, p), y.gate(F, t), 1.0);
      y.gate(Cc, t).ApplyFloor(-50);
      y.gate(Cc, t).ApplyCeiling(50);
      y.gate(H, t).Tanh(y.gate(Cc, t));
      y.gate(O, t).AddMatDiagVec(1.0, y.gate(Cc, t), kNoTrans, po, 1.0);
      y.gate(O, t).Sigmoid(y.gate(O, t));
      y.gate(Mm, t).AddMatMatElements(1.0, y.gate(H, t), y.gate(O, t), 0.0);
      y.rec(t).AddMatMat(1.0, y.gate(Mm, t), kNoTrans, Wrm, kTrans, 0.0);
    }
    PutMat(NAME2("out", step), y.recs(1, T));
    d.recs(1, T).CopyFromMat(od);
    for (int t = T; t >= 1; t--) {
      const int p = t - 1, q = t + 1;
      d.rec(t).AddMatMat(1.0, d.gates(q, 1), kNoTrans, Wr, kNoTrans, 1.0);
      d.gate(Mm, t).AddMatMat(1.0, d.rec(t), kNoTrans, Wrm, kNoTrans, 0.0);
      d.gate(H, t).AddMatMatElements(1.0, d.gate(Mm, t), y.gate(O, t), 0.0);
      d.gate(H, t).DiffTanh(y.gate(H, t), d.gate(H, t));
      d.gate(O, t).AddMatMatElements(1.0, d.gate(Mm, t), y.gate(H, t), 0.0);
      d.gate(O, t).DiffSigmoid(y.gate(O, t), d.gate(O, t));
      d.gate(Cc, t).AddMat(1.0, d.gate(H, t));
      d.gate(Cc, t).AddMatMatElements(1.0, d.gate(Cc, q), y.gate(F, q), 1.0);
      d.gate(Cc, t).AddMatDiagVec(1.0, d.gate(I, q), kNoTrans, pi, 1.0);
      d.gate(Cc, t).AddMatDiagVec(1.0, d.gate(F, q), kNoTrans, pf, 1.0);
      d.gate(Cc, t).AddMatDiagVec(1.0, d.gate(O, t), kNoTrans, po, 1.0);
      d.gate(F, t).AddMatMatElements(1.0, d.gate(Cc, t), y.gate(Cc, p), 0.0);
      d.gate(F, t).DiffSigmoid(y.gate(F, t), d.gate(F, t));
      d.gate(I, t).AddMatMatElements(1.0, d.gate(Cc, t), y.gate(G, t), 0.0);
      d.gate(I, t).DiffSigmoid(y.gate(I, t), d.gate(I, t));
      d.gate(G, t).AddMatMatElements(1.0, d.gate(Cc, t), y.gate(I, t), 0.0);
      d.gate(G, t).DiffTanh(y.gate(G, t), d.gate(G, t));
    }
    Mat id(T * S, D);
    id.AddMatMat(1.0, d.gates(1, T), kNoTrans, Wx, kNoTrans, 0.0);
    PutMat(NAME2("in_diff", step), id);
    // corr = gradient + momentum * corr
    cWx.AddMatMat(1.0, d.gates(1, T), kTrans, in, kNoTrans, mmt);
    cWr.AddMatMat(1.0, d.gates(1, T), kTrans, y.recs(0, T), kNoTrans, mmt);
    cb.AddRowSumMat(1.0, d.gates(1, T), mmt);
    cpi.AddDiagMatMat(1.0, d.cols(I, 1, T), kTrans, y.cols(Cc, 0, T), kNoTrans, mmt);
    cpf.AddDiagMatMat(1.0, d.cols(F, 1, T), kTrans, y.cols(Cc, 0, T), kNoTrans, mmt);
    cpo.AddDiagMatMat(1.0, d.cols(O, 1, T), kTrans, y.cols(Cc, 1, T), kNoTrans, mmt);
    cWrm.AddMatMat(1.0, d.recs(1, T), kTrans, y.cols(Mm, 1, T), kNoTrans, mmt);
    // element-wise clipping
    cWx.ApplyFloor(-clip); cWx.ApplyCeiling(clip); cWr.ApplyFloor(-clip); cWr.ApplyCeiling(clip);
    cb.ApplyFloor(-clip); cb.ApplyCeiling(clip); cWrm.ApplyFloor(-clip); cWrm.ApplyCeiling(clip);
    cpi.ApplyFloor(-clip); cpi.ApplyCeiling(clip); cpf.ApplyFloor(-clip); cpf.ApplyCeiling(clip); cpo.ApplyFloor(-clip); cpo.ApplyCeiling(clip);
    PutMat(NAME2("cWx", step), cWx); PutMat(NAME2("cWrm", step), cWrm);
    // Update
    Wx.AddMat(-lr, cWx); Wr.AddMat(-lr, cWr); bias.AddVec(-lr, cb, 1.0);
    pi.AddVec(-lr, cpi, 1.0); pf.AddVec(-lr, cpf, 1.0); po.AddVec(-lr, cpo, 1.0);
    Wrm.AddMat(-lr, cWrm);
    put_params(step + 1);
  }
#undef NAME2
}

// GruStreams: nnet-gru-streams.h:238-450.  Buffer columns z|r|m|g|h; row blocks as above.
static void Gru(const char *tag = "gru", int T = 5, int S = 3, int D = 6, int H = 7, float wr = 0.4f, bool inputs_regenerated = false) {
  char nm[32];
#define NAME(x) (std::snprintf(nm, 32, "%s_%s", tag, x), nm)
  Mat Wx, Wh, Wg, in, od;
  Vec bias;
  Fill(&Wx, 3 * H, D, -wr, wr); Fill(&Wh, 2 * H, H, -wr, wr); Fill(&Wg, H, H, -wr, wr); FillVec(&bias, 3 * H, -0.3f, 0.3f);
  Fill(&in, T * S, D, -1.5f, 1.5f); Fill(&od, T * S, H, -1.0f, 1.0f);
  g_skip = inputs_regenerated;
  PutMat(NAME("Wx"), Wx); PutMat(NAME("Wh"), Wh); PutMat(NAME("Wg"), Wg); PutVec(NAME("bias"), bias); PutMat(NAME("in"), in); PutMat(NAME("od"), od);
  g_skip = false;
  Mat Y((T + 2) * S, 5 * H), Dd((T + 2) * S, 5 * H);
  enum { Z, Rr, Mm, G, Hh };
  struct View {
    Mat &b; int H, S;
    Sub col(int k, int t, int n = 1, int w = 1) { return Sub(b, t * S, n * S, k * H, w * H); }
  } y = {Y, H, S}, d = {Dd, H, S};
  y.col(Z, 1, T, 3).AddMatMat(1.0, in, kNoTrans, Wx, kTrans, 0.0);
  y.col(Z, 1, T, 3).AddVecToRows(1.0, bias);
  for (int t = 1; t <= T; t++) {
    y.col(Z, t, 1, 2).AddMatMat(1.0, y.col(Hh, t - 1), kNoTrans, Wh, kTrans, 1.0);
    y.col(Z, t, 1, 2).Sigmoid(y.col(Z, t, 1, 2));
    y.col(G, t).AddMatMatElements(1.0, y.col(Rr, t), y.col(Hh, t - 1), 0.0);
    y.col(Mm, t).AddMatMat(1.0, y.col(G, t), kNoTrans, Wg, kTrans, 1.0);
    y.col(Mm, t).Tanh(y.col(Mm, t));
    y.col(Hh, t).AddMat(1.0, y.col(Hh, t - 1));
    y.col(Hh, t).AddMatMatElements(-1.0, y.col(Hh, t - 1), y.col(Z, t), 1.0);
    y.col(Hh, t).AddMatMatElements(1.0, y.col(Z, t), y.col(Mm, t), 1.0);
  }
  PutMat(NAME("fwd_buf"), Y);
  d.col(Hh, 1, T).CopyFromMat(od);
  for (int t = T; t >= 1; t--) {
    d.col(Hh, t).AddMatMat(1.0, d.col(Z, t + 1, 1, 2), kNoTrans, Wh, kNoTrans, 1.0);
    d.col(Hh, t).AddMat(1.0, d.col(Hh, t + 1));
    d.col(Hh, t).AddMatMatElements(-1.0, d.col(Hh, t + 1), y.col(Z, t + 1), 1.0);
    d.col(Hh, t).AddMatMatElements(1.0, d.col(G, t + 1), y.col(Rr, t + 1), 1.0);
    d.col(Mm, t).AddMatMatElements(1.0, d.col(Hh, t), y.col(Z, t), 0.0);
    d.col(Mm, t).DiffTanh(y.col(Mm, t), d.col(Mm, t));
    d.col(G, t).AddMatMat(1.0, d.col(Mm, t), kNoTrans, Wg, kNoTrans, 0.0);
    d.col(Rr, t).AddMatMatElements(1.0, d.col(G, t), y.col(Hh, t - 1), 0.0);
    d.col(Rr, t).DiffSigmoid(y.col(Rr, t), d.col(Rr, t));
    d.col(Z, t).AddMatMatElements(1.0, d.col(Hh, t), y.col(Mm, t), 0.0);
    d.col(Z, t).AddMatMatElements(-1.0, d.col(Hh, t), y.col(Hh, t - 1), 1.0);
    d.col(Z, t).DiffSigmoid(y.col(Z, t), d.col(Z, t));
  }
  PutMat(NAME("bwd_buf"), Dd);
  Mat id(T * S, D), gWx(3 * H, D), gWh(2 * H, H), gWg(H, H);
  Vec gb(3 * H);
  id.AddMatMat(1.0, d.col(Z, 1, T, 3), kNoTrans, Wx, kNoTrans, 0.0);
  gWx.AddMatMat(1.0, d.col(Z, 1, T, 3), kTrans, in, kNoTrans, 0.0);
  gb.AddRowSumMat(1.0, d.col(Z, 1, T, 3), 0.0);
  gWh.AddMatMat(1.0, d.col(Z, 1, T, 2), kTrans, y.col(Hh, 0, T), kNoTrans, 0.0);
  gWg.AddMatMat(1.0, d.col(Mm, 1, T), kTrans, y.col(G, 1, T), kNoTrans, 0.0);
  PutMat(NAME("in_diff"), id); PutMat(NAME("gWx"), gWx); PutMat(NAME("gWh"), gWh); PutMat(NAME("gWg"), gWg); PutVec(NAME("gb"), gb);
#undef NAME
}

// RowConvolution: nnet-row-convolution.cc:90-169 (a D x D product per frame whose diagonal is the output), ragged lengths
// (tag, sizes, lengths: the full-size digest records of main()'s third mode; the defaults are the small fixture's)
static void RowConv(const char *tag = "rc", int T = 6, int S = 2, int D = 7, int K = 3, const int *len_in = nullptr, bool inputs_regenerated = false) {
  const int Ts = T + K;
  const int len_small[2] = {6, 4};
  const int *len = len_in ? len_in : len_small;
  char nm[32];
#define NAME(x) (std::snprintf(nm, 32, "%s_%s", tag, x), nm)
  Mat w, in, od, out(T * S, D), in_buf(Ts * S, D), conv(D, D), idb(Ts * S, D), cd(D, K + 1), wdiff(D, K + 1), idf(T * S, D);
  Fill(&w, D, K + 1, -0.8f, 0.8f); Fill(&in, T * S, D, -1.5f, 1.5f); Fill(&od, T * S, D, -1.0f, 1.0f);
  g_skip = inputs_regenerated;
  PutMat(NAME("w"), w); PutMat(NAME("in"), in); PutMat(NAME("od"), od);
  g_skip = false;
  std::vector<int32> lens(len, len + S);
  Put(NAME("lens"), 1, S, 1, lens.data());
  for (int s = 0; s < S; s++) {
    for (int t = 0; t < len[s] + K; t++) in_buf.Row(s * Ts + t).CopyFromVec(in.Row((t < len[s] ? t : len[s] - 1) * S + s));
    for (int t = 0; t < len[s]; t++) {
      conv.AddMatMat(1.0, w, kNoTrans, in_buf.RowRange(s * Ts + t, K + 1), kNoTrans, 0.0);
      out.Row(t * S + s).CopyDiagFromMat(conv);
    }
  }
  PutMat(NAME("out"), out);
  for (int s = 0; s < S; s++)
    for (int t = 0; t < len[s]; t++) {
      Sub yh(in_buf.RowRange(s * Ts + t, K + 1)), yd(idb.RowRange(s * Ts + t, K + 1));
      cd.SetZero(); cd.AddMat(1.0, w); cd.MulRowsVec(od.Row(t * S + s)); yd.AddMat(1.0, cd, kTrans);
      cd.SetZero(); cd.AddMat(1.0, yh, kTrans); cd.MulRowsVec(od.Row(t * S + s)); wdiff.AddMat(1.0, cd);
    }
  for (int s = 0; s < S; s++)
    for (int t = 0; t < len[s]; t++) idf.Row(t * S + s).CopyFromVec(idb.Row(s * Ts + t));
  PutMat(NAME("in_diff"), idf); PutMat(NAME("w_diff"), wdiff);
#undef NAME
}

// CompactFsmn: nnet-cfsmn-component.h:169-264 (past 3, future 2 taps)
static void Fsmn(const char *tag = "fsmn", int T = 11, int D = 9, int P = 3, int F = 2, bool inputs_regenerated = false) {
  const int C = P + F + 1;
  char nm[32];
#define NAME(x) (std::snprintf(nm, 32, "%s_%s", tag, x), nm)
  Mat coef, in, od, pad(T + C - 1, D), tmp(T * C, D), out(T, D), corr(C, D), rev(C, D), idf(T, D);
  Fill(&coef, C, D, -0.5f, 0.5f); Fill(&in, T, D, -1.5f, 1.5f); Fill(&od, T, D, -1.0f, 1.0f);
  g_skip = inputs_regenerated;
  PutMat(NAME("coef"), coef); PutMat(NAME("in"), in); PutMat(NAME("od"), od);
  g_skip = false;
  pad.RowRange(P, T).CopyFromMat(in);
  tmp.AddConvMatMatElements(1.0, pad, coef, 0.0);
  out.CopyFromMat(in);
  out.AddRowSumMat(1.0, tmp, 1.0);
  PutMat(NAME("out"), out);
  for (int i = 0; i < C; i++) tmp.RowRange(i * T, T).AddMatMatElements(1.0, pad.RowRange(i, T), od, 0.0);
  corr.AddRowSumMat(1.0, tmp, 0.0);
  pad.SetZero();
  pad.RowRange(F, T).CopyFromMat(od);
  for (int i = 0, j = C - 1; i < C; i++, j--) rev.Row(j).CopyFromVec(coef.Row(i));
  tmp.AddConvMatMatElements(1.0, pad, rev, 0.0);
  idf.CopyFromMat(od);
  idf.AddRowSumMat(1.0, tmp, 1.0);
  PutMat(NAME("in_diff"), idf); PutMat(NAME("corr"), corr);
#undef NAME
}

// Xent::Eval: nnet-loss.cc:63-156 -- frames whose target row sums to zero are masked through the frame weights; diff = (y - t) w;
// frame accuracy; cross entropy, entropy and likelihood sums.  stats = {frames, correct, loss, entropy, likelihood} (double).
static void XentChain() {
  const int R = 40, Cn = 50;
  Mat logits, y, tgt;
  Fill(&logits, R, Cn, -3.0f, 3.0f);
  y.Resize(R, Cn);
  y.ApplySoftMaxPerRow(logits);
  Matrix<float> ht(R, Cn);
  for (int r = 0; r < R; r++) ht(r, (int)(Uniform() * Cn) % Cn) = 1.0f;
  for (int c = 0; c < Cn; c++) ht(1, c) = 0.0f;                       // a frame without a target: masked
  for (int c = 0; c < Cn; c++) ht(2, c) = 0.0f;
  ht(2, 0) = 0.25f; ht(2, 1) = 0.75f;                                 // a soft posterior
  tgt.Resize(R, Cn); tgt.CopyFromMat(ht);
  Vector<float> fw(R);
  for (int r = 0; r < R; r++) fw(r) = Uniform();
  fw(0) = 0.0f;
  PutMat("xe_y", y); PutMat("xe_tgt", tgt);
  { Vec t(R); t.CopyFromVec(fw); PutVec("xe_fw", t); }
  Vec w(R), tsum(R);
  w.CopyFromVec(fw);
  tsum.AddColSumMat(1.0, tgt, 0.0);
  w.MulElements(tsum);
  double st[5];
  st[0] = w.Sum();
  Mat diff(y);
  diff.AddMat(-1.0, tgt);
  diff.MulRowsVec(w);
  PutMat("xe_diff", diff);
  CuArray<int32> io, it;
  y.FindRowMaxId(&io);
  tgt.FindRowMaxId(&it);
  std::vector<int32> ho, htg;
  io.CopyToVec(&ho); it.CopyToVec(&htg);
  Vector<float> hw(R); w.CopyToVec(&hw);
  st[1] = 0.0;
  for (int r = 0; r < R; r++) if (ho[r] == htg[r]) st[1] += hw(r);     // CountCorrectFramesWeighted (nnet-loss.cc:40-60): weights of the hits
  Mat a(y);
  a.Add(1e-20); a.ApplyLog(); a.MulElements(tgt); a.MulRowsVec(w);
  st[2] = -a.Sum();
  a.CopyFromMat(tgt);
  a.Add(1e-20); a.ApplyLog(); a.MulElements(tgt); a.MulRowsVec(w);
  st[3] = -a.Sum();
  a.CopyFromMat(y);
  a.MulElements(tgt); a.MulRowsVec(w);
  st[4] = a.Sum();
  Put("xe_stats", 1, 5, 2, st, 8);
}

int main(int argc, char **argv) {
  if (argc == 3 && !std::strcmp(argv[2], "lstm_fullwidth")) {
    // tests/golden/lstm_fullwidth.bin: two training steps of ONE projected-LSTM layer at BASELINE cfg3's widths (C 512, R 256, input 512, S = 32
    // streams, T = 60 frames = chunk 40 + right context 20), momentum 0.9, element-wise clipping 5, learn rate 0.002, on the reference's library
    // -- as a digest (every 61st element + sums): 6.7 MB of parameters would not be a "small fixture"
    g_out = std::fopen(argv[1], "wb");
    if (!g_out) return 1;
    const int32 rng[2] = {(int32)(g_state & 0xFFFFFFFFull), (int32)(g_state >> 32)};
    Put("lcfull_rng", 1, 2, 1, rng);
    g_stride = 61;
    LstmProjectedTrain("lcfull", 60, 32, 512, 512, 256, 0.9f, 5.0f, 0.002f, 0.02f, 1.0f, true);
    // the two directions of BLstmProjectedStreamsLC at the same widths, as the small `lcf` / `lcb` records pin them: forward in time from a
    // CARRIED state (a whole row block of an earlier chunk), backward in time from zero -- every gate of every frame (forward and backward
    // buffers), input diff, all gradients; stride 257 (the buffers are 7.6 M floats each).  Appended: the records above stay as they are.
    g_stride = 257;
    const int32 rng2[2] = {(int32)(g_state & 0xFFFFFFFFull), (int32)(g_state >> 32)};
    Put("lcdir_rng", 1, 2, 1, rng2);
    LstmProjected("lcff", false, true, false, 256, nullptr, 60, 32, 512, 512, 0.02f, true);
    LstmProjected("lcfb", true, false, false, 256, nullptr, 60, 32, 512, 512, 0.02f, true);
    std::fclose(g_out);
    return 0;
  }
  if (argc == 3 && !std::strcmp(argv[2], "temporal_fullsize")) {
    // tests/golden/temporal_fullsize.bin: RowConvolution (512 wide, FutureContext 20, T = 800, S = 32 ragged streams) and CompactFsmn (512 wide,
    // 30 + 30 taps, T = 800) at the sizes BASELINE cfg5 swaps them in at, on the reference's library (the D x D product per frame of
    // nnet-row-convolution.cc:128-133 included), as a digest with stride 257; weights, inputs and out-diffs replayed from `tmp_rng`
    g_out = std::fopen(argv[1], "wb");
    if (!g_out) return 1;
    const int32 rng[2] = {(int32)(g_state & 0xFFFFFFFFull), (int32)(g_state >> 32)};
    Put("tmp_rng", 1, 2, 1, rng);
    g_stride = 257;
    std::vector<int> lens(32);
    for (int i = 0; i < 32; i++) lens[i] = i == 0 ? 800 : 400 + (int)(Uniform() * 401.0f) % 401;   // 400 .. 800 frames, the first one full
    RowConv("rcfull", 800, 32, 512, 20, lens.data(), true);
    Fsmn("fsmnfull", 800, 512, 30, 30, true);
    // GruStreams 512 -> 512 (cfg5's swap), S = 32 streams, T = 60 frames: every gate of every frame, input diff, the four gradients (appended)
    Gru("grufull", 60, 32, 512, 512, 0.03f, true);
    std::fclose(g_out);
    return 0;
  }
  if (argc != 2) { std::fprintf(stderr, "usage: %s <out.bin> [lstm_fullwidth | temporal_fullsize]\n", argv[0]); return 1; }
  g_out = std::fopen(argv[1], "wb");
  if (!g_out) return 1;
  Operations();
  Affine();
  BatchNorm();
  LstmProjected("lstm", false, false);
  Gru();
  LstmProjected("lcf", false, true);    // appended after the records above: those stay byte-identical
  LstmProjected("lcb", true, false);
  LstmProjected("cifg", false, false, true);
  LstmProjected("lstmnp", false, false, false, 0);
  LstmProjected("blstmnp", true, false, false, 0);   // the backward-in-time direction of BLstm (nnet-recurrent-component.cc:912-1450)
  RowConv();
  Fsmn();
  XentChain();
  { const int lens[3] = {5, 3, 4}; LstmProjected("bmask", true, false, false, 5, lens); }   // appended last: earlier records unchanged
  LstmProjectedTrain("lstm2", 6, 8, 48, 64, 32, 0.9f, 0.5f, 0.01f);   // appended behind everything else (round 4)
  std::fclose(g_out);
  return 0;
}
