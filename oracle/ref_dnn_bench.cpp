// ref_dnn_bench.cpp -- TEST / MEASUREMENT INFRASTRUCTURE (bench.py's cpu_baseline leg, kind "reference").  One training step of
// BASELINE cfg2 (5 x 2048 sigmoid DNN + BatchNormalization, 440 -> 3000, minibatch 1024) executed on the REFERENCE's own CuMatrix /
// CuVector CPU branch (src/aslp-cudamatrix with HAVE_CUDA undefined on src/matrix, compiled where they lie) over a real BLAS, the
// OpenBLAS inside the image's scipy wheel -- the arrangement of oracle/gen_cumatrix_blas_golden.cpp (see oracle/Makefile), and
// the one BASELINE.md's CPU numbers were taken with.  The component headers cannot be compiled here (OpenFst), so this driver
// issues the operations of AffineTransform (nnet-affine-transform.h:186-245), BatchNormalization (nnet-batch-normalization.h:
// 177-284), Sigmoid (nnet-activation.h:153-165), Softmax, Xent::Eval (nnet-loss.cc:63-156) in the order of Nnet::Propagate /
// Backpropagate (nnet-nnet.cc:70-154: Update right after each component's Backpropagate; the unused in-diff of the first layer
// is computed like the reference computes it); every operation is the reference's own code.  The same sequences are what
// tests/golden/cumatrix_blas_ops.bin pins the oracle and the HIP engine against.
// usage: ref_dnn_bench <seconds> [max_steps [cfg1]]  ->  one JSON line on stdout ("cfg1": the net without BatchNormalization at minibatch 256)
//        ref_dnn_bench golden <out.bin>        ->  tests/golden/dnn_cfg2_fullsize.bin: two training steps of the same net at lr 0.008 on fresh
//                                                  minibatches, as a digest (every 257th element of a tensor + its sums; weights and data
//                                                  are replayed by the reader from the recorded generator state, oracle_lib.GoldenRng)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "aslp-cudamatrix/cu-array.h"
#include "aslp-cudamatrix/cu-math.h"
#include "aslp-cudamatrix/cu-matrix.h"
#include "aslp-cudamatrix/cu-vector.h"

extern "C" int scipy_openblas_get_num_threads(void);

using namespace kaldi;
typedef CuMatrix<float> Mat;
typedef CuVector<float> Vec;

static unsigned long long g_state = 777;
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

// ---- golden mode: records as in oracle/gen_cumatrix_blas_golden.cpp ({name[32], rows, cols, kind, data}; kind 0 float32, 1 int32, 2 float64)
static FILE *g_out = nullptr;
static const int kDigestStride = 257;
static void Put(const char *name, int rows, int cols, int kind, const void *data, int elem = 4) {
  char nm[32];
  std::memset(nm, 0, sizeof(nm));
  std::strncpy(nm, name, 31);
  std::fwrite(nm, 1, 32, g_out);
  int32 hdr[3] = {rows, cols, kind};
  std::fwrite(hdr, sizeof(int32), 3, g_out);
  std::fwrite(data, elem, (size_t)rows * cols, g_out);
}
static void PutDigest(const char *name, const CuMatrixBase<float> &m) {   // every kDigestStride-th element + `name#` = {sum, sum of squares, count}
  Matrix<float> h(m.NumRows(), m.NumCols());
  m.CopyToMat(&h);
  std::vector<float> pick;
  double acc[3] = {0.0, 0.0, (double)h.NumRows() * h.NumCols()};
  size_t i = 0;
  for (int r = 0; r < h.NumRows(); r++)
    for (int c = 0; c < h.NumCols(); c++, i++) {
      const float v = h(r, c);
      acc[0] += v;
      acc[1] += (double)v * v;
      if (i % (size_t)kDigestStride == 0) pick.push_back(v);
    }
  Put(name, 1, (int)pick.size(), 0, pick.data());
  char nm[32];
  std::snprintf(nm, 32, "%s#", name);
  Put(nm, 1, 3, 2, acc, 8);
}
static void PutVec(const char *name, const CuVectorBase<float> &v) {
  Vector<float> h(v.Dim());
  v.CopyToVec(&h);
  Put(name, 1, v.Dim(), 0, h.Data());
}

struct Affine {
  Mat W, Wc;
  Vec b, bc;
  void Init(int out, int in) {
    Fill(&W, out, in, -0.07f, 0.07f);
    Wc.Resize(out, in);
    b.Resize(out); bc.Resize(out);
  }
  void Propagate(const Mat &in, Mat *out) {
    out->Resize(in.NumRows(), W.NumRows());
    out->AddVecToRows(1.0, b, 0.0);
    out->AddMatMat(1.0, in, kNoTrans, W, kTrans, 1.0);
  }
  void Backpropagate(const Mat &od, Mat *id) {
    id->Resize(od.NumRows(), W.NumCols(), kUndefined);
    id->AddMatMat(1.0, od, kNoTrans, W, kNoTrans, 0.0);
  }
  void Update(const Mat &in, const Mat &od, float lr, float mmt) {
    Wc.AddMatMat(1.0, od, kTrans, in, kNoTrans, mmt);
    bc.AddRowSumMat(1.0, od, mmt);
    W.AddMat(-lr, Wc);
    b.AddVec(-lr, bc);
  }
};

struct BatchNorm {
  Vec scale, shift, mean, var, dmean, dvar, dshift, dscale;
  CuVector<double> acc_m, acc_v;
  Mat xs, e;
  void Init(int dim) {
    scale.Resize(dim); scale.Set(1.0); shift.Resize(dim);
    mean.Resize(dim); var.Resize(dim); dmean.Resize(dim); dvar.Resize(dim); dshift.Resize(dim); dscale.Resize(dim);
    acc_m.Resize(dim); acc_v.Resize(dim);
  }
  void Propagate(const Mat &in, Mat *out) {
    const int rows = in.NumRows();
    out->Resize(rows, in.NumCols());
    xs.Resize(rows, in.NumCols(), kUndefined); e.Resize(rows, in.NumCols(), kUndefined);
    mean.AddRowSumMat(1.0 / rows, in, 0.0);
    xs.CopyFromMat(in);
    xs.AddVecToRows(-1.0, mean, 1.0);
    out->AddMatMatElements(1.0, xs, xs, 0.0);
    var.AddRowSumMat(1.0 / rows, *out, 0.0);
    var.Add(1e-7);
    var.ApplyPow(0.5);
    var.InvertElements();
    xs.MulColsVec(var);
    out->CopyFromMat(xs);
    out->MulColsVec(scale);
    out->AddVecToRows(1.0, shift, 1.0);
    acc_m.AddRowSumMat(1.0, CuMatrix<double>(in), 1.0);
    e.AddMatMatElements(1.0, in, in, 0.0);
    acc_v.AddRowSumMat(1.0, CuMatrix<double>(e), 1.0);
  }
  void Backpropagate(const Mat &in, const Mat &od, Mat *id, float mmt) {
    const int rows = in.NumRows();
    e.AddMatMatElements(1.0, xs, od, 0.0);
    dscale.AddRowSumMat(1.0, e, mmt);
    dshift.AddRowSumMat(1.0, od, mmt);
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
    id->Resize(rows, in.NumCols(), kUndefined);
    id->CopyFromMat(xs);
    id->MulColsVec(var);
    id->AddMat(1.0, e);
    id->AddVecToRows(1.0 / rows, dmean, 1.0);
  }
  void Update(float lr) {
    scale.AddVec(-lr, dscale, 1.0);
    shift.AddVec(-lr, dshift, 1.0);
  }
};

int main(int argc, char **argv) {
  const bool golden = argc == 3 && !std::strcmp(argv[1], "golden");
  const double budget = (!golden && argc > 1) ? atof(argv[1]) : 10.0;
  const int max_steps = (!golden && argc > 2) ? atoi(argv[2]) : 20;
  // "cfg1" as a third argument: BASELINE cfg1 / cfg4's per-GPU leg -- the same net WITHOUT BatchNormalization at minibatch 256, learn rate
  // 0.008 (run_dnn.sh:81-83), the configuration the >= 30x target is stated on (aslp-nnetbin/aslp-nnet-train-frame.cc:109-131)
  const bool cfg1 = !golden && argc > 3 && !std::strcmp(argv[3], "cfg1");
  const bool with_bn = !cfg1;
  const int IN = 440, HID = 2048, NH = 5, OUT = 3000, MB = cfg1 ? 256 : 1024;
  const float lr = (golden || cfg1) ? 0.008f : 1e-5f, mmt = 0.0f;
  const unsigned long long state0 = g_state;
  std::vector<Affine> aff(NH + 1);
  std::vector<BatchNorm> bn(NH);
  for (int l = 0; l <= NH; l++) aff[l].Init(l == NH ? OUT : HID, l == 0 ? IN : HID);
  for (int l = 0; l < NH; l++) bn[l].Init(HID);
  Mat x, tgt;
  auto new_batch = [&]() {   // x uniform in [-1.7, 1.7), one label per frame
    Fill(&x, MB, IN, -1.7f, 1.7f);
    Matrix<float> ht(MB, OUT);
    for (int r = 0; r < MB; r++) ht(r, (int)(Uniform() * OUT) % OUT) = 1.0f;
    tgt.Resize(MB, OUT); tgt.CopyFromMat(ht);
  };
  new_batch();
  Vector<float> fw_host(MB);
  fw_host.Set(1.0);
  std::vector<Mat> a(NH + 1), z(NH), y(NH), da(NH + 1), dz(NH), dy(NH);   // affine out, BN out, sigmoid out and their diffs
  Mat post, diff, aux, in_diff0;
  Vec w(MB), tsum(MB);
  CuArray<int32> io, it;
  double loss = 0.0;
  auto step = [&]() {
    const Mat *cur = &x;
    for (int l = 0; l < NH; l++) {
      aff[l].Propagate(*cur, &a[l]);
      if (with_bn) bn[l].Propagate(a[l], &z[l]);
      y[l].Resize(MB, HID, kUndefined);
      y[l].Sigmoid(with_bn ? z[l] : a[l]);
      cur = &y[l];
    }
    aff[NH].Propagate(*cur, &a[NH]);
    post.Resize(MB, OUT, kUndefined);
    post.ApplySoftMaxPerRow(a[NH]);
    // Xent::Eval
    w.CopyFromVec(fw_host);
    tsum.AddColSumMat(1.0, tgt, 0.0);
    w.MulElements(tsum);
    diff = post;
    diff.AddMat(-1.0, tgt);
    diff.MulRowsVec(w);
    post.FindRowMaxId(&io);
    tgt.FindRowMaxId(&it);
    aux = post; aux.Add(1e-20); aux.ApplyLog(); aux.MulElements(tgt); aux.MulRowsVec(w);
    loss = -aux.Sum();
    aux = tgt; aux.Add(1e-20); aux.ApplyLog(); aux.MulElements(tgt); aux.MulRowsVec(w);
    loss += aux.Sum();
    aux = post; aux.MulElements(tgt); aux.MulRowsVec(w);
    (void)aux.Sum();
    // Backpropagate: Softmax passes the diff through (nnet-activation.h:51-55), then top affine, then [Sigmoid, BN, Affine] x 5
    const Mat *d = &diff;
    aff[NH].Backpropagate(*d, &da[NH]);
    aff[NH].Update(y[NH - 1], *d, lr, mmt);
    d = &da[NH];
    for (int l = NH - 1; l >= 0; l--) {
      dy[l].Resize(MB, HID, kUndefined);
      dy[l].DiffSigmoid(y[l], *d);
      if (with_bn) {
        bn[l].Backpropagate(a[l], dy[l], &dz[l], mmt);
        bn[l].Update(lr);
      }
      const Mat &dpre = with_bn ? dz[l] : dy[l];
      aff[l].Backpropagate(dpre, &da[l]);
      aff[l].Update(l == 0 ? x : y[l - 1], dpre, lr, mmt);
      d = &da[l];
    }
  };
  if (golden) {
    g_out = std::fopen(argv[2], "wb");
    if (!g_out) return 1;
    const int32 rng[2] = {(int32)(state0 & 0xFFFFFFFFull), (int32)(state0 >> 32)};
    Put("cfg2_rng", 1, 2, 1, rng);
    const int32 stride = kDigestStride;
    Put("cfg2_stride", 1, 1, 1, &stride);
    char nm[32];
    for (int s = 0; s < 2; s++) {
      if (s > 0) new_batch();   // (the first batch was drawn behind the weights, above)
      step();
      std::snprintf(nm, 32, "cfg2_post%d", s); PutDigest(nm, post);
      std::snprintf(nm, 32, "cfg2_loss%d", s); Put(nm, 1, 1, 2, &loss, 8);
      for (int l = 0; l <= NH; l++) {
        std::snprintf(nm, 32, "cfg2_W%d_%d", l, s + 1); PutDigest(nm, aff[l].W);
        std::snprintf(nm, 32, "cfg2_b%d_%d", l, s + 1); PutVec(nm, aff[l].b);
        if (l < NH) {
          std::snprintf(nm, 32, "cfg2_sc%d_%d", l, s + 1); PutVec(nm, bn[l].scale);
          std::snprintf(nm, 32, "cfg2_sh%d_%d", l, s + 1); PutVec(nm, bn[l].shift);
        }
      }
    }
    std::fclose(g_out);
    return 0;
  }
  step();  // warm-up (page-in, OpenBLAS thread start)
  const auto t0 = std::chrono::steady_clock::now();
  int steps = 0;
  double el = 0.0;
  do {
    step();
    steps++;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } while (el < budget && steps < max_steps);
  std::printf("{\"frames_per_sec\": %.3f, \"steps\": %d, \"seconds\": %.3f, \"minibatch\": %d, \"threads\": %d, \"xent_per_frame\": %.5f, \"batch_norm\": %d}\n",
              steps * MB / el, steps, el, MB, scipy_openblas_get_num_threads(), loss / MB, with_bn ? 1 : 0);
  return 0;
}
