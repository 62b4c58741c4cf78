// How fast can ONE CU bring L2-resident data into LDS?  Every workgroup (one per CU, 256 threads) streams 48 KB "tiles" out of a 2 MB window that its
// whole XCD shares (so the lines hit the L2 after the first touch), three tiles in flight, through
//   mode 0: LDS-DMA (global_load_lds_dwordx4), what the product kernels do
//   mode 1: global_load_dwordx4 into registers + ds_write_b128 (as written here the staging array lives in scratch memory: not a fair figure, not reported)
//   mode 2: two thirds by LDS-DMA, one third through registers (the two return paths side by side)
//   mode 3: LDS-DMA, tiles of 24 KB, six in flight (same bytes in flight, smaller requests)
// Build: hipcc -O3 --offload-arch=gfx950 devtools/micro/lds_feed.hip -o devtools/micro/lds_feed ; run: devtools/micro/lds_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// tile = UNITS KB; each wave issues UNITS / 4 one-KB requests per tile; DEPTH tiles in flight; REG of every 3 units go through registers
template <int UNITS, int DEPTH, int REGMODE>
__global__ void __launch_bounds__(256) feed(const char *src, int tiles, unsigned window, float *sink) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int G = UNITS / 4;
  const unsigned start = (blockIdx.x >> 3) * 4096u;   // workgroups of one XCD (blockIdx & 7) walk the same window, a little apart
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto request = [&](int t, float4 (&stage)[G]) {
    const unsigned off = (start + (unsigned)t * (UNITS * 1024u)) & (window - 1u);
#pragma unroll
    for (int u = 0; u < G; u++) {
      const unsigned unit = wave + u * 4;
      const char *p = src + ((off + unit * 1024u + lane * 16u) & (window - 1u));
      const bool via_reg = REGMODE == 1 || (REGMODE == 2 && (u % 3) == 2);
      if (via_reg) stage[u] = *reinterpret_cast<const float4 *>(p);
      else glds16(p, __builtin_amdgcn_readfirstlane(lds_base + (t % DEPTH) * (UNITS * 1024) + unit * 1024));
    }
  };
  auto commit = [&](int t, const float4 (&stage)[G]) {   // register-staged units: into LDS
#pragma unroll
    for (int u = 0; u < G; u++) {
      const bool via_reg = REGMODE == 1 || (REGMODE == 2 && (u % 3) == 2);
      if (via_reg) *reinterpret_cast<float4 *>(lds + (t % DEPTH) * (UNITS * 1024) + (wave + u * 4) * 1024 + lane * 16) = stage[u];
    }
  };
  float4 st[DEPTH][G];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; d++) request(d, st[d]);
  for (int t0 = 0; t0 < tiles; t0 += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; i++) {
      const int t = t0 + i;
      request(t + DEPTH - 1, st[(i + DEPTH - 1) % DEPTH]);
      wait_vmcnt<(DEPTH - 1) * G>();
      commit(t, st[i]);
      __builtin_amdgcn_s_barrier();
      // consume a little so that nothing is optimised away (one 16-byte read per thread)
      const float4 v = *reinterpret_cast<const float4 *>(lds + (t % DEPTH) * (UNITS * 1024) + ((threadIdx.x * 16) % (UNITS * 1024)));
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      __builtin_amdgcn_s_barrier();
    }
  }
  wait_vmcnt<0>();
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[blockIdx.x] = acc.x;
}

template <int UNITS, int DEPTH, int REGMODE>
static void run(const char *name, const char *src, float *sink, int tiles, unsigned window) {
  auto k = feed<UNITS, DEPTH, REGMODE>;
  const int ldsb = UNITS * 1024 * DEPTH;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(256), dim3(256), ldsb, 0, src, tiles, window, sink);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(256), dim3(256), ldsb, 0, src, tiles, window, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps, bytes = (double)tiles * UNITS * 1024.0 * 256.0;
  printf("%-44s %8.1f us   %6.2f TB/s into LDS   %5.1f B/clk/CU at 2.4 GHz\n", name, us, bytes / us / 1e6, bytes / 256.0 / (us * 2400.0));
}

int main() {
  const unsigned window = 2u << 20;
  char *src; float *sink;
  hipMalloc(&src, window + 4096); hipMalloc(&sink, 4096);
  hipMemset(src, 1, window + 4096);
  const int tiles = 768;   // x 48 KB = 36 MB per workgroup: the launch edge does not matter
  run<48, 3, 0>("LDS-DMA, 48 KB tiles, 3 in flight", src, sink, tiles, window);
  run<48, 3, 1>("registers + ds_write_b128, 48 KB x 3", src, sink, tiles, window);
  run<48, 3, 2>("2/3 LDS-DMA + 1/3 registers, 48 KB x 3", src, sink, tiles, window);
  run<24, 6, 0>("LDS-DMA, 24 KB tiles, 6 in flight", src, sink, 2 * tiles, window);
  run<24, 3, 0>("LDS-DMA, 24 KB tiles, 3 in flight", src, sink, 2 * tiles, window);
  run<48, 2, 0>("LDS-DMA, 48 KB tiles, 2 in flight", src, sink, tiles, window);
  return 0;
}
