// gen_ctc_golden.cpp -- fixture generator (development container only).
//
// Runs the REFERENCE's own Warp-CTC CPU implementation (compiled from the sources where they
// lie under /root/reference/src/warp-ctc) on the reference's own test inputs (genActs /
// genLabels of tests/test.h, the cases of tests/test_cpu.cpp) plus a few ragged cases, and
// dumps inputs and outputs as small binary fixtures into tests/golden/.  No reference source
// is copied: the two headers are #included from the read-only mount at build time.
//
//   g++ -std=c++11 -O2 -fopenmp -I/root/reference/src/warp-ctc/include -I/root/reference/src/warp-ctc/tests \
//       oracle/gen_ctc_golden.cpp /root/reference/src/warp-ctc/src/ctc_entrypoint.cpp -o oracle/_ref/gen_ctc_golden
//   oracle/_ref/gen_ctc_golden tests/golden
//
// File format ctc_<name>.bin (little endian): int32 A, mb, maxT, n_labels; int32 input_lengths[mb];
// int32 label_lengths[mb]; int32 flat_labels[n_labels]; float acts[maxT*mb*A]; float costs[mb];
// float grads[maxT*mb*A].
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <string>

#include "test.h"  // reference: genActs, genLabels (tests/test.h:28-57)

static void run_and_dump(const std::string &dir, const std::string &name, int A, std::vector<float> acts,
                         const std::vector<std::vector<int>> &labels, const std::vector<int> &sizes) {
  int mb = labels.size();
  std::vector<int> flat, lens;
  for (auto &l : labels) {
    flat.insert(flat.end(), l.begin(), l.end());
    lens.push_back(l.size());
  }
  int maxT = 0;
  for (int s : sizes) maxT = std::max(maxT, s);
  std::vector<float> costs(mb), grads(acts.size(), 0.0f);
  ctcComputeInfo info;
  info.loc = CTC_CPU;
  info.num_threads = 1;
  size_t bytes;
  throw_on_error(get_workspace_size(lens.data(), sizes.data(), A, mb, info, &bytes), "get_workspace_size");
  void *ws = malloc(bytes);
  throw_on_error(compute_ctc_loss(acts.data(), grads.data(), flat.data(), lens.data(), sizes.data(), A, mb, costs.data(), ws, info),
                 "compute_ctc_loss");
  free(ws);
  std::string path = dir + "/ctc_" + name + ".bin";
  FILE *f = fopen(path.c_str(), "wb");
  int hdr[4] = {A, mb, maxT, (int)flat.size()};
  fwrite(hdr, sizeof(int), 4, f);
  fwrite(sizes.data(), sizeof(int), mb, f);
  fwrite(lens.data(), sizeof(int), mb, f);
  fwrite(flat.data(), sizeof(int), flat.size(), f);
  fwrite(acts.data(), sizeof(float), acts.size(), f);
  fwrite(costs.data(), sizeof(float), mb, f);
  fwrite(grads.data(), sizeof(float), grads.size(), f);
  fclose(f);
  printf("%s: A=%d mb=%d maxT=%d cost[0]=%g\n", path.c_str(), A, mb, maxT, costs[0]);
}

int main(int argc, char **argv) {
  std::string dir = argc > 1 ? argv[1] : "tests/golden";
  // small_test (test_cpu.cpp:12-67)
  run_and_dump(dir, "small", 5, {0.1f, 0.6f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.6f, 0.1f, 0.1f}, {{1, 2}}, {2});
  // inf_test (test_cpu.cpp:69-122)
  {
    const int A = 15, T = 50, L = 10;
    std::vector<int> labels = genLabels(A, L);
    labels[0] = 2;
    std::vector<float> acts = genActs(A * T);
    for (int i = 0; i < T; ++i) acts[A * i + 2] = -1e30;
    run_and_dump(dir, "inf", A, acts, {labels}, {T});
  }
  // grad_check problems (test_cpu.cpp:213-216)
  {
    const int A = 20, T = 50, L = 15;
    run_and_dump(dir, "grad_a20_t50_l15", A, genActs(A * T), {genLabels(A, L)}, {T});
  }
  {
    const int A = 5, T = 10, L = 5, mb = 65;
    std::vector<std::vector<int>> labels;
    std::vector<int> sizes;
    for (int i = 0; i < mb; ++i) {
      labels.push_back(genLabels(A, L));
      sizes.push_back(T);
    }
    run_and_dump(dir, "grad_a5_t10_l5_mb65", A, genActs(A * T * mb), labels, sizes);
  }
  // ragged minibatch: different T and L per utterance, an empty label sequence, an infeasible one
  // (L + repeats > T -> cost 0, gradient untouched, cpu_ctc.h:196-198), long repeats
  {
    const int A = 12, mb = 6, maxT = 40;
    std::vector<int> sizes = {40, 33, 7, 40, 12, 25};
    std::vector<std::vector<int>> labels = {genLabels(A, 12), genLabels(A, 5), {3, 3, 3, 3, 3}, {}, {1, 2, 3, 4, 5, 6}, {7, 7, 7, 8, 8, 9, 9, 9, 9}};
    std::vector<float> acts = genActs(A * maxT * mb);
    for (auto &a : acts) a = 6.0f * a - 3.0f;
    run_and_dump(dir, "ragged", A, acts, labels, sizes);
  }
  // a longer utterance with the BASELINE alphabet (A = 128)
  {
    const int A = 128, T = 200, L = 50, mb = 2;
    std::vector<std::vector<int>> labels = {genLabels(A, L), genLabels(A, L - 7)};
    std::vector<float> acts = genActs(A * T * mb);
    run_and_dump(dir, "a128_t200", A, acts, labels, {T, T - 31});
  }
  return 0;
}
