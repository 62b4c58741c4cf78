// gen_component_golden.cpp -- TEST INFRASTRUCTURE.  Runs the op sequences of the reference's front-end components on the REFERENCE's
// own CuMatrix / CuVector CPU branch (src/aslp-cudamatrix with HAVE_CUDA undefined on src/matrix, linked against the OpenBLAS of the
// image's scipy wheel exactly as gen_cumatrix_blas_golden.cpp is: see oracle/Makefile) and writes inputs + outputs to
// tests/golden/component_ops.bin (same record format).
//
// The component headers themselves (aslp-nnet/nnet-*.h) need OpenFst's fst/fst-decl.h through nnet-utils.h and cannot be compiled
// here; the sequences below are this file's reading of those lines, the arithmetic of every step is the reference's own:
//   LinearTransform         nnet-linear-transform.h:127-160      (two minibatches: momentum, l2, l1, learn-rate coefficient)
//   ConvolutionalComponent  nnet-convolutional-component.h:268-470 (CopyCols by the column map, one product per patch, AddCols over the
//                                                                 reversed map, gradient summed over patches, max-norm; two minibatches)
//   MaxPoolingComponent     nnet-max-pooling-component.h:101-162 (overlapping pools, ties)
//   LengthNormComponent     nnet-various.h:338-358               (<= 64 columns: double row sums; wider: sgemv)
//   Pnorm / Maxout          nnet-activation.h:341-373            (GroupPnorm p = 2, 1, 3; GroupMax; their derivatives x MulRowsGroupMat)
// and the bare operations cudaF_max / equal_element_mask / group_* / mul_rows_group_mat stand for.
#include <algorithm>
#include <cmath>
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
static void PutMat(const char *name, const CuMatrixBase<float> &m) {
  Matrix<float> h(m.NumRows(), m.NumCols());
  m.CopyToMat(&h);
  std::vector<float> flat((size_t)h.NumRows() * h.NumCols());
  for (int r = 0; r < h.NumRows(); r++) std::memcpy(&flat[(size_t)r * h.NumCols()], h.RowData(r), sizeof(float) * h.NumCols());
  Put(name, h.NumRows(), h.NumCols(), 0, flat.data());
}
static void PutVec(const char *name, const CuVectorBase<float> &v) {
  Vector<float> h(v.Dim());
  v.CopyToVec(&h);
  Put(name, 1, v.Dim(), 0, h.Data());
}
static void PutScalars(const char *name, const std::vector<float> &v) { Put(name, 1, (int)v.size(), 0, v.data()); }

static unsigned long long g_state = 0xD1B54A32D192ED03ull;
static float Uniform() {
  g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
  return (float)((g_state >> 40) * (1.0 / 16777216.0));
}
static void Fill(Mat *m, int rows, int cols, float lo, float hi, float quantum = 0.0f) {
  Matrix<float> h(rows, cols);
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      float v = lo + (hi - lo) * Uniform();
      if (quantum > 0.0f) v = quantum * std::floor(v / quantum + 0.5f);   // coarse values: ties between pool members / group members
      h(r, c) = v;
    }
  m->Resize(rows, cols);
  m->CopyFromMat(h);
}
static void FillVec(Vec *v, int n, float lo, float hi) {
  Vector<float> h(n);
  for (int i = 0; i < n; i++) h(i) = lo + (hi - lo) * Uniform();
  v->Resize(n);
  v->CopyFromVec(h);
}

static void Linear() {
  const int N = 37, D = 23, O = 19;
  const float lr = 0.01f, mmt = 0.9f, l2 = 1e-3f, l1 = 1e-4f, coef = 0.7f;
  Mat W, corr(O, D);
  Fill(&W, O, D, -0.5f, 0.5f);
  PutMat("lin_W0", W);
  PutScalars("lin_opts", {lr, mmt, l2, l1, coef});
  for (int step = 0; step < 2; step++) {
    char nm[32];
    Mat in, od, out(N, O), id(N, D);
    Fill(&in, N, D, -1.0f, 1.0f);
    Fill(&od, N, O, -0.3f, 0.3f);
    std::snprintf(nm, 32, "lin_in%d", step); PutMat(nm, in);
    std::snprintf(nm, 32, "lin_od%d", step); PutMat(nm, od);
    out.AddMatMat(1.0, in, kNoTrans, W, kTrans, 0.0);                    // :129
    id.AddMatMat(1.0, od, kNoTrans, W, kNoTrans, 0.0);                   // :135
    corr.AddMatMat(1.0, od, kTrans, in, kNoTrans, mmt);                  // :149
    W.AddMat(-lr * l2 * N, W);                                           // :152
    cu::RegularizeL1(&W, &corr, lr * l1 * N, lr);                        // :156
    W.AddMat(-lr * coef, corr);                                          // :159
    std::snprintf(nm, 32, "lin_out%d", step); PutMat(nm, out);
    std::snprintf(nm, 32, "lin_id%d", step); PutMat(nm, id);
    std::snprintf(nm, 32, "lin_corr%d", step); PutMat(nm, corr);
    std::snprintf(nm, 32, "lin_W%d", step + 1); PutMat(nm, W);
  }
}

static void Conv() {
  // 3 spliced frames of 10 bands; patches of 4 bands every 2: 4 patches; 5 filters of 3 x 4 -> output 20
  const int N = 13, num_splice = 3, patch_stride = 10, patch_dim = 4, patch_step = 2, F = 5;
  const int in_dim = num_splice * patch_stride, P = 1 + (patch_stride - patch_dim) / patch_step, K = num_splice * patch_dim;
  const float lr = 0.05f, coef = 0.8f, bcoef = 1.2f, max_norm = 1.5f;
  Mat filters;
  Vec bias;
  Fill(&filters, F, K, -1.0f, 1.0f);
  FillVec(&bias, F, -0.5f, 0.5f);
  PutMat("conv_filters0", filters); PutVec("conv_bias0", bias);
  PutScalars("conv_geom", {(float)in_dim, (float)F, (float)patch_dim, (float)patch_step, (float)patch_stride, lr, coef, bcoef, max_norm});
  std::vector<int32> column_map(K * P);   // :318-325
  for (int p = 0, index = 0; p < P; p++)
    for (int s = 0; s < num_splice; s++)
      for (int d = 0; d < patch_dim; d++, index++) column_map[index] = p * patch_step + s * patch_stride + d;
  for (int step = 0; step < 2; step++) {
    char nm[32];
    Mat in, od, out(N, F * P), patches(N, K * P), patch_diffs(N, K * P), id(N, in_dim);
    Fill(&in, N, in_dim, -1.0f, 1.0f);
    Fill(&od, N, F * P, -0.3f, 0.3f);
    std::snprintf(nm, 32, "conv_in%d", step); PutMat(nm, in);
    std::snprintf(nm, 32, "conv_od%d", step); PutMat(nm, od);
    // PropagateFnc :268-340
    CuArray<int32> cu_map(column_map);
    patches.CopyCols(in, cu_map);
    for (int p = 0; p < P; p++) {
      Sub tgt(out.ColRange(p * F, F)), patch(patches.ColRange(p * K, K));
      tgt.AddVecToRows(1.0, bias, 0.0);
      tgt.AddMatMat(1.0, patch, kNoTrans, filters, kTrans, 1.0);
    }
    // BackpropagateFnc :390-422 (in_diff zeroed by Component::Backpropagate)
    for (int p = 0; p < P; p++) {
      Sub pd(patch_diffs.ColRange(p * K, K)), odp(od.ColRange(p * F, F));
      pd.AddMatMat(1.0, odp, kNoTrans, filters, kNoTrans, 0.0);
    }
    std::vector<std::vector<int32> > rev(in_dim);
    for (int j = 0; j < K * P; j++) rev[column_map[j]].push_back(j);   // ReverseIndexes :346-361
    size_t L = 0;
    for (int i = 0; i < in_dim; i++) L = std::max(L, rev[i].size());
    for (size_t k = 0; k < L; k++) {                                   // RearrangeIndexes :375-388 + the AddCols loop :417-420
      std::vector<int32> pass(in_dim, -1);
      for (int i = 0; i < in_dim; i++) if (k < rev[i].size()) pass[i] = rev[i][k];
      CuArray<int32> cu_cols(pass);
      id.AddCols(patch_diffs, cu_cols);
    }
    // Update :425-470
    Mat fgrad(F, K);
    Vec bgrad(F);
    for (int p = 0; p < P; p++) {
      Sub dp(od.ColRange(p * F, F)), patch(patches.ColRange(p * K, K));
      fgrad.AddMatMat(1.0, dp, kTrans, patch, kNoTrans, 1.0);
      bgrad.AddRowSumMat(1.0, dp, 1.0);
    }
    filters.AddMat(-lr * coef, fgrad);
    bias.AddVec(-lr * bcoef, bgrad);
    {
      Mat lin_sqr(filters);
      lin_sqr.MulElements(filters);
      Vec l2(F);
      l2.AddColSumMat(1.0, lin_sqr, 0.0);
      l2.ApplyPow(0.5);
      Vec scl(l2);
      scl.Scale(1.0 / max_norm);
      scl.ApplyFloor(1.0);
      scl.InvertElements();
      filters.MulRowsVec(scl);
    }
    std::snprintf(nm, 32, "conv_out%d", step); PutMat(nm, out);
    std::snprintf(nm, 32, "conv_id%d", step); PutMat(nm, id);
    std::snprintf(nm, 32, "conv_fgrad%d", step); PutMat(nm, fgrad);
    std::snprintf(nm, 32, "conv_bgrad%d", step); PutVec(nm, bgrad);
    std::snprintf(nm, 32, "conv_filters%d", step + 1); PutMat(nm, filters);
    std::snprintf(nm, 32, "conv_bias%d", step + 1); PutVec(nm, bias);
  }
}

static void MaxPool() {
  // 7 patches of 6 values; pools of 3 patches every 2 patches: 3 overlapping pools
  const int N = 11, pool_stride = 6, num_patches = 7, pool_size = 3, pool_step = 2;
  const int in_dim = num_patches * pool_stride, num_pools = 1 + (num_patches - pool_size) / pool_step;
  Mat in, od, out(N, num_pools * pool_stride), id(N, in_dim);
  Fill(&in, N, in_dim, -2.0f, 2.0f, 0.5f);
  Fill(&od, N, num_pools * pool_stride, -1.0f, 1.0f);
  PutMat("pool_in", in); PutMat("pool_od", od);
  PutScalars("pool_geom", {(float)in_dim, (float)pool_size, (float)pool_step, (float)pool_stride});
  for (int q = 0; q < num_pools; q++) {   // :107-115
    Sub pool(out.ColRange(q * pool_stride, pool_stride));
    pool.Set(-1e20);
    for (int r = 0; r < pool_size; r++) pool.Max(in.ColRange((r + q * pool_step) * pool_stride, pool_stride));
  }
  PutMat("pool_out", out);
  std::vector<int32> summands(num_patches, 0);   // :118-162
  id.SetZero();
  for (int q = 0; q < num_pools; q++)
    for (int r = 0; r < pool_size; r++) {
      const int p = r + q * pool_step;
      Sub in_p(in.ColRange(p * pool_stride, pool_stride)), out_q(out.ColRange(q * pool_stride, pool_stride));
      Sub tgt(id.ColRange(p * pool_stride, pool_stride));
      Mat src(od.ColRange(q * pool_stride, pool_stride));
      Mat mask;
      in_p.EqualElementMask(out_q, &mask);
      src.MulElements(mask);
      tgt.AddMat(1.0, src);
      summands[p] += 1;
    }
  for (int p = 0; p < num_patches; p++) {
    Sub tgt(id.ColRange(p * pool_stride, pool_stride));
    tgt.Scale(1.0 / summands[p]);
  }
  PutMat("pool_id", id);
  {   // the bare operations: Max, EqualElementMask
    Mat a, b, mask;
    Fill(&a, 9, 21, -1.0f, 1.0f, 0.25f);
    Fill(&b, 9, 21, -1.0f, 1.0f, 0.25f);
    PutMat("op_max_a", a); PutMat("op_max_b", b);
    a.EqualElementMask(b, &mask);
    PutMat("op_eqmask", mask);
    a.Max(b);
    PutMat("op_max_out", a);
  }
}

static void LengthNorm() {
  const int widths[2] = {40, 100};
  for (int w = 0; w < 2; w++) {
    const int N = 9, D = widths[w];
    char nm[32];
    Mat in, od, out(N, D), id(N, D), aux;
    Fill(&in, N, D, -3.0f, 3.0f);
    Fill(&od, N, D, -1.0f, 1.0f);
    std::snprintf(nm, 32, "ln%d_in", w); PutMat(nm, in);
    std::snprintf(nm, 32, "ln%d_od", w); PutMat(nm, od);
    Vec scales(N);
    aux = in;                                   // :344-351
    aux.MulElements(aux);
    scales.AddColSumMat(1.0, aux, 0.0);
    scales.ApplyPow(0.5);
    scales.InvertElements();
    out.CopyFromMat(in);
    out.MulRowsVec(scales);
    id.CopyFromMat(od);                         // :356-357
    id.MulRowsVec(scales);
    std::snprintf(nm, 32, "ln%d_out", w); PutMat(nm, out);
    std::snprintf(nm, 32, "ln%d_scales", w); PutVec(nm, scales);
    std::snprintf(nm, 32, "ln%d_id", w); PutMat(nm, id);
  }
}

static void Groups() {
  const int N = 6, G = 5, O = 8;
  Mat in, od;
  Fill(&in, N, G * O, -2.0f, 2.0f, 0.25f);
  {   // zeros inside groups, and one group that is all zeros (output 0: derivative defined as 0)
    Matrix<float> h(N, G * O);
    in.CopyToMat(&h);
    for (int k = 0; k < G; k++) h(2, 3 * G + k) = 0.0f;
    h(0, 1) = 0.0f; h(4, 17) = 0.0f;
    in.CopyFromMat(h);
  }
  Fill(&od, N, O, -1.0f, 1.0f);
  PutMat("grp_in", in); PutMat("grp_od", od);
  const float powers[3] = {2.0f, 1.0f, 3.0f};
  for (int i = 0; i < 3; i++) {
    char nm[32];
    Mat out(N, O), id(N, G * O);
    out.GroupPnorm(in, powers[i]);              // nnet-activation.h:342
    id.GroupPnormDeriv(in, out, powers[i]);     // :347
    std::snprintf(nm, 32, "pnorm%d_deriv", i); PutMat(nm, id);
    id.MulRowsGroupMat(od);                     // :348
    std::snprintf(nm, 32, "pnorm%d_out", i); PutMat(nm, out);
    std::snprintf(nm, 32, "pnorm%d_id", i); PutMat(nm, id);
  }
  {
    Mat out(N, O), id(N, G * O);
    out.GroupMax(in);                           // :366
    id.GroupMaxDeriv(in, out);                  // :371
    PutMat("gmax_deriv", id);
    id.MulRowsGroupMat(od);                     // :372
    PutMat("gmax_out", out); PutMat("gmax_id", id);
  }
}

int main(int argc, char **argv) {
  if (argc != 2) { std::fprintf(stderr, "usage: %s <out.bin>\n", argv[0]); return 1; }
  g_out = std::fopen(argv[1], "wb");
  if (!g_out) return 1;
  Linear();
  Conv();
  MaxPool();
  LengthNorm();
  Groups();
  std::fclose(g_out);
  return 0;
}
