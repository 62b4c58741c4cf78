// Producer / consumer split-fp16 product kernels (csrc/gemm_split16.hip gemm_s16_pc) against the one-role kernels they are meant to replace:
// bit-for-bit comparison of the results on random planes, time per launch (interleaved rounds, HIP events), and the ablations (no MFMA /
// no DMA / no LDS reads) of the new kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I kaldi-aslp_amd/csrc devtools/micro/s16_pc.hip kaldi-aslp_amd/csrc/runtime.cpp -o devtools/micro/s16_pc
#include "../../kaldi-aslp_amd/csrc/gemm_split16.hip"
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
using namespace aslp;

struct Variant { std::string name; std::function<void(GemmArgs &, const S16Operands &)> fn; bool check; };

static float time_us(const std::function<void()> &f, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, cur_stream());
  for (int i = 0; i < reps; i++) f();
  (void)hipEventRecord(e1, cur_stream());
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms * 1000.f / reps;
}

static h16 *g_pool; static unsigned *g_slots; static float *g_C, *g_Cref;

static void run(const char *title, bool a_kc, bool b_kc, int M, int N, int K, std::vector<Variant> vars) {
  GemmArgs g = {};
  g.M = M; g.N = N; g.K = K; g.C = g_C; g.ldc = N; g.alpha = 1.f; g.beta = 0.f; g.wide_epilogue = 1;
  g.ep = aslp_gemm_epilogue();
  auto pad = [](int x) { return (x + 63) / 64 * 64; };
  const int ar = a_kc ? M : K, ac = a_kc ? K : M, br = b_kc ? N : K, bc = b_kc ? K : N;
  S16Operands ops;
  size_t pa = (size_t)pad(ar) * pad(ac), pb = (size_t)pad(br) * pad(bc);
  ops.a = S16View{g_pool, g_pool + pa, pad(ac), ar, ac, g_slots};
  ops.b = S16View{g_pool + 2 * pa, g_pool + 2 * pa + pb, pad(bc), br, bc, g_slots + 1};
  ops.a1 = ops.a; ops.b1 = ops.b;
  ops.kp = pad(K);
  const size_t nC = (size_t)M * N;
  std::vector<float> ref(nC), out(nC);
  printf("== %s  %d x %d x %d  (2 M N K = %.2f GF)\n", title, M, N, K, 2.0 * M * N * K / 1e9);
  std::vector<double> sum(vars.size(), 0.0), best(vars.size(), 1e30);
  for (size_t v = 0; v < vars.size(); v++) {   // correctness against variant 0, and warm-up
    (void)hipMemsetAsync(g_C, 0xFF, nC * 4, cur_stream());
    printf("   [%s]\n", vars[v].name.c_str());
    GemmArgs gg = g;
    vars[v].fn(gg, ops);
    (void)hipStreamSynchronize(cur_stream());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("   %-40s LAUNCH ERROR %s\n", vars[v].name.c_str(), hipGetErrorString(e)); vars[v].check = false; continue; }
    if (v == 0) { (void)hipMemcpy(ref.data(), g_C, nC * 4, hipMemcpyDeviceToHost); continue; }
    if (!vars[v].check) continue;
    (void)hipMemcpy(out.data(), g_C, nC * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; double maxd = 0;
    for (size_t i = 0; i < nC; i++) if (std::memcmp(&out[i], &ref[i], 4)) { bad++; double d = std::fabs((double)out[i] - ref[i]); if (d > maxd || d != d) maxd = d; }
    printf("   %-40s vs %-24s: %zu of %zu elements differ (max |d| %.3g)\n", vars[v].name.c_str(), vars[0].name.c_str(), bad, nC, maxd);
  }
  const int rounds = 5, reps = M * (double)N * K > 3e10 ? 10 : 40;
  for (int r = 0; r < rounds; r++)
    for (size_t v = 0; v < vars.size(); v++) {
      GemmArgs gg = g;
      const float us = time_us([&] { GemmArgs g2 = gg; vars[v].fn(g2, ops); }, reps);
      sum[v] += us; if (us < best[v]) best[v] = us;
    }
  for (size_t v = 0; v < vars.size(); v++)
    printf("   %-40s avg %8.2f us  best %8.2f us  %7.1f TF-eq (%.3f of 839)\n", vars[v].name.c_str(), sum[v] / rounds, best[v], 2.0 * M * N * K / (sum[v] / rounds) / 1e6,
           2.0 * M * N * K / (sum[v] / rounds) / 1e6 / 838.7);
}

#define V(name, ...) Variant{name, [](GemmArgs &g, const S16Operands &o) { __VA_ARGS__(g, o); }, true}
#define VA(name, ...) Variant{name, [](GemmArgs &g, const S16Operands &o) { __VA_ARGS__(g, o); }, false}

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t halves = (size_t)96 << 20;
  (void)hipMalloc(&g_pool, halves * 2); (void)hipMalloc(&g_slots, 64); (void)hipMalloc(&g_C, (size_t)4096 * 4096 * 4);
  {  // random planes: hi like values scaled into [2^-1, 2^13], lo' any 11-bit remainder; a different sequence per 2 MB so that rows differ
    std::vector<h16> h((size_t)16 << 20);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); i++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const float u = (float)((s >> 40) & 0xFFFFFF) / 16777216.0f - 0.5f;
      h[i] = (h16)(u * 4096.0f);
    }
    for (size_t off = 0; off < halves; off += h.size()) (void)hipMemcpy(g_pool + off, h.data(), std::min(h.size(), halves - off) * 2, hipMemcpyHostToDevice);
  }
  float one = 8192.f;
  (void)hipMemcpy(g_slots, &one, 4, hipMemcpyHostToDevice); (void)hipMemcpy(g_slots + 1, &one, 4, hipMemcpyHostToDevice);
  const int sel = argc > 1 ? atoi(argv[1]) : 0;   // 0 = everything, else one group
  if (sel == 0 || sel == 1)
    run("NT (cfg2 forward)", true, true, 1024, 2048, 2048,
        {V("glds 64x128 4w NS3 (308)", launch_s16<64, 128, 2, 2, 3, true, true>),
         V("pc 64x128 KT64 NS3", launch_s16_pc<64, 128, 64, 3, true, true, false>),
         VA("pc 64x128 HOT (every request an L2 hit)", launch_s16_pc<64, 128, 64, 3, true, true, false, 0, 4>),
         VA("pc 64x128 HOT DMA only", launch_s16_pc<64, 128, 64, 3, true, true, false, 5, 4>),
         VA("pc 64x128 no MFMA", launch_s16_pc<64, 128, 64, 3, true, true, false, 1>),
         VA("pc 64x128 no LDS reads (DMA + MFMA)", launch_s16_pc<64, 128, 64, 3, true, true, false, 4>),
         VA("pc 64x128 no DMA", launch_s16_pc<64, 128, 64, 3, true, true, false, 2>),
         VA("pc 64x128 MFMA only", launch_s16_pc<64, 128, 64, 3, true, true, false, 6>),
         VA("pc 64x128 DMA only", launch_s16_pc<64, 128, 64, 3, true, true, false, 5>),
         VA("pc 64x128 reads only", launch_s16_pc<64, 128, 64, 3, true, true, false, 3>)});
  if (sel == 0 || sel == 3)
    run("TN (cfg2 weight gradient)", false, false, 2048, 2048, 1024,
        {V("ks128 128x128 ring4 (328)", launch_s16_ks128<false>),
         V("pc 128x128 KT32 NS4", launch_s16_pc<128, 128, 32, 4, false, false, false>),
         VA("pc 128x128 NS4 HOT", launch_s16_pc<128, 128, 32, 4, false, false, false, 0, 4>),
         VA("pc 128x128 NS4 no MFMA", launch_s16_pc<128, 128, 32, 4, false, false, false, 1>),
         VA("pc 128x128 NS4 no DMA", launch_s16_pc<128, 128, 32, 4, false, false, false, 2>),
         VA("pc 128x128 NS4 DMA only", launch_s16_pc<128, 128, 32, 4, false, false, false, 5>)});
  if (sel == 0 || sel == 4)
    run("NT 4096^3", true, true, 4096, 4096, 4096,
        {V("glds 128x128 4w NS2 (311)", launch_s16<128, 128, 2, 2, 2, true, true>),
         V("pc 128x128 KT64 NS2", launch_s16_pc<128, 128, 64, 2, true, true, false>),
         VA("pc 128x128 NS2 HOT", launch_s16_pc<128, 128, 64, 2, true, true, false, 0, 4>),
         VA("pc 128x128 no MFMA", launch_s16_pc<128, 128, 64, 2, true, true, false, 1>),
         VA("pc 128x128 no DMA", launch_s16_pc<128, 128, 64, 2, true, true, false, 2>)});
  return 0;
}
