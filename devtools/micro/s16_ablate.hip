// Ablation timings of the split-fp16 product kernel (csrc/gemm_split16.hip): the same launch with the MFMAs, the LDS-DMA or the LDS reads
// taken out (wrong results on purpose), to see which of the three the kernel waits for.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I kaldi-aslp_amd/csrc devtools/micro/s16_ablate.hip kaldi-aslp_amd/csrc/runtime.cpp -o devtools/micro/s16_ablate
#include "../../kaldi-aslp_amd/csrc/gemm_split16.hip"
#include <cstdio>
#include <vector>
using namespace aslp;

template <int BM, int BN, int WGM, int WGN, int NS, bool A_KC, bool B_KC, int ABL>
float time_one(GemmArgs g, const S16Operands &ops, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) launch_s16<BM, BN, WGM, WGN, NS, A_KC, B_KC, ABL>(g, ops);
  (void)hipEventRecord(e0, cur_stream());
  for (int i = 0; i < reps; i++) launch_s16<BM, BN, WGM, WGN, NS, A_KC, B_KC, ABL>(g, ops);
  (void)hipEventRecord(e1, cur_stream());
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

template <int BM, int BN, int WGM, int WGN, int NS, bool A_KC, bool B_KC>
void run(const char *name, int M, int N, int K, h16 *pool, unsigned *slots, float *C) {
  GemmArgs g = {};
  g.M = M; g.N = N; g.K = K; g.C = C; g.ldc = N; g.alpha = 1.f; g.beta = 0.f; g.wide_epilogue = 1;
  g.ep = aslp_gemm_epilogue();
  auto pad = [](int x) { return (x + 63) / 64 * 64; };
  const int ar = A_KC ? M : K, ac = A_KC ? K : M, br = B_KC ? N : K, bc = B_KC ? K : N;
  S16Operands ops;
  size_t pa = (size_t)pad(ar) * pad(ac), pb = (size_t)pad(br) * pad(bc);
  ops.a = S16View{pool, pool + pa, pad(ac), ar, ac, slots};
  ops.b = S16View{pool + 2 * pa, pool + 2 * pa + pb, pad(bc), br, bc, slots + 1};
  ops.a1 = ops.a; ops.b1 = ops.b;
  ops.kp = pad(K);
  const int reps = 50;
  const float full = time_one<BM, BN, WGM, WGN, NS, A_KC, B_KC, 0>(g, ops, reps);
  const float nomma = time_one<BM, BN, WGM, WGN, NS, A_KC, B_KC, 1>(g, ops, reps);
  const float nodma = time_one<BM, BN, WGM, WGN, NS, A_KC, B_KC, 2>(g, ops, reps);
  const float mmaonly = time_one<BM, BN, WGM, WGN, NS, A_KC, B_KC, 6>(g, ops, reps);
  const float dmaonly = time_one<BM, BN, WGM, WGN, NS, A_KC, B_KC, 5>(g, ops, reps);
  const float rdonly = time_one<BM, BN, WGM, WGN, NS, A_KC, B_KC, 3>(g, ops, reps);
  const double bytes = ((double)(M + BM - 1) / BM) * ((double)(N + BN - 1) / BN) * (BM + BN) * (double)ops.kp * 4.0;
  printf("%-28s %5d x %5d x %5d: full %6.1f us | no MFMA %6.1f | no DMA %6.1f | MFMA only %6.1f | DMA only %6.1f (%.1f TB/s L2->LDS) | LDS reads only %6.1f\n", name, M, N, K,
         full, nomma, nodma, mmaonly, dmaonly, bytes / dmaonly / 1e6, rdonly);
}

int main() {
  const size_t halves = (size_t)64 << 20;
  h16 *pool; unsigned *slots; float *C;
  (void)hipMalloc(&pool, halves * 2); (void)hipMalloc(&slots, 64); (void)hipMalloc(&C, (size_t)4096 * 4096 * 4);
  std::vector<h16> h(1 << 20);
  for (size_t i = 0; i < h.size(); i++) h[i] = (h16)(float)((int)(i * 2654435761u >> 20) % 2048 - 1024);
  for (size_t off = 0; off < halves; off += h.size()) (void)hipMemcpy(pool + off, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  float one = 1.f;
  (void)hipMemcpy(slots, &one, 4, hipMemcpyHostToDevice); (void)hipMemcpy(slots + 1, &one, 4, hipMemcpyHostToDevice);
  run<64, 128, 2, 4, 3, true, true>("64x128 8w NS3 NT", 1024, 2048, 2048, pool, slots, C);
  run<64, 128, 2, 4, 3, true, false>("64x128 8w NS3 NN", 1024, 2048, 2048, pool, slots, C);
  run<64, 128, 2, 4, 3, false, false>("64x128 8w NS3 TN", 2048, 2048, 1024, pool, slots, C);
  run<128, 128, 2, 4, 2, true, true>("128x128 8w NS2 NT", 4096, 4096, 4096, pool, slots, C);
  run<128, 128, 4, 2, 2, true, true>("128x128 8w(4x2) NS2 NT", 4096, 4096, 4096, pool, slots, C);
  run<64, 128, 2, 2, 3, true, true>("64x128 4w NS3 NT", 1024, 2048, 2048, pool, slots, C);
  run<64, 128, 2, 2, 3, true, false>("64x128 4w NS3 NN", 1024, 2048, 2048, pool, slots, C);
  run<64, 128, 2, 2, 3, false, false>("64x128 4w NS3 TN", 2048, 2048, 1024, pool, slots, C);
  run<128, 128, 2, 2, 2, true, true>("128x128 4w NS2 NT", 1024, 2048, 2048, pool, slots, C);
  run<128, 128, 2, 2, 2, true, true>("128x128 4w NS2 NT", 4096, 4096, 4096, pool, slots, C);
  run<64, 128, 2, 2, 3, true, true>("64x128 4w NS3 NT", 4096, 4096, 4096, pool, slots, C);
  return 0;
}
