// The product kernels' operand traffic without the products: 256 workgroups, XCD x = blockIdx % 8 owns an 8 x 4 block of 64 x 128 tiles, workgroup
// (r, c) of it streams its A panel (16 KB per K tile, shared with the 4 workgroups of its row) and its B panel (32 KB per K tile, shared with the 8 of
// its column) into LDS by LDS-DMA, 32 K tiles, three stages -- the lines are NOT in the XCD's L2 when the first workgroup asks (8 MB per XCD, fresh
// after every launch).  Variants: who issues the requests, and whether somebody touches the lines of tile t + P ahead of time.
//   0: all four waves request (12 KB each per tile)              1: waves 0, 1 request, waves 2, 3 idle
//   2: waves 0, 1 request, waves 2, 3 touch tile t + P (one lane per 128-byte line, 4-byte LDS-DMA loads into a dump area)
//   3: all four waves request AND touch (touches in front of the tile's requests)
// Build: hipcc -O3 --offload-arch=gfx950 devtools/micro/lds_feed_cold.hip -o devtools/micro/lds_feed_cold
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory", "m0");
}
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_byte_addr) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int kTile = 48 * 1024, kStages = 3, kTiles = 32;
constexpr size_t kXcdBytes = 8u << 20;   // A: 8 panels x 512 KB, B: 4 panels x 1 MB

template <int MODE, int P>
__global__ void __launch_bounds__(256) feed(const char *src, float *sink) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, r = j >> 2, c = j & 3;
  const char *pa = src + (size_t)xcd * kXcdBytes + (size_t)r * (512u << 10);               // + t * 16 KB
  const char *pb = src + (size_t)xcd * kXcdBytes + (4u << 20) + (size_t)c * (1u << 20);    // + t * 32 KB
  constexpr bool two = MODE == 1 || MODE == 2;
  constexpr int NREQ = two ? 2 : 4, G = 48 / NREQ;
  const bool requester = !two || wave < 2;
  auto unit_addr = [&](int t, int unit) {   // unit: 1 KB; 0..15 of A, 16..47 of B
    return unit < 16 ? pa + (size_t)t * (16u << 10) + unit * 1024 + lane * 16 : pb + (size_t)t * (32u << 10) + (unit - 16) * 1024 + lane * 16;
  };
  auto request = [&](int t) {
    if (t >= kTiles) t = kTiles - 1;
#pragma unroll
    for (int u = 0; u < G; u++) {
      const int unit = wave + u * NREQ;
      glds16(unit_addr(t, unit), __builtin_amdgcn_readfirstlane(lds_base + (t % kStages) * kTile + unit * 1024));
    }
  };
  // touches of tile t: 384 lines; `nt` touching waves, each 384 / 64 / nt instructions, one line per lane
  auto touch = [&](int t, int tw, int nt) {
    if (t >= kTiles) t = kTiles - 1;
    for (int k = tw; k < 6; k += nt) {
      const int line = k * 64 + lane;   // 0..127: A, 128..383: B
      const char *p = line < 128 ? pa + (size_t)t * (16u << 10) + line * 128 : pb + (size_t)t * (32u << 10) + (line - 128) * 128;
      glds4(p, __builtin_amdgcn_readfirstlane(lds_base + kStages * kTile));
    }
  };
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (requester) { request(0); request(1); }
  for (int t = 0; t < kTiles; t++) {
    if (MODE == 3) touch(t + P, wave, 4);
    if (requester) {
      request(t + 2);
      if (MODE == 3) { if (wave < 2) wait_vmcnt<2 * G + 2 + 2>(); else wait_vmcnt<2 * G + 1 + 1>(); }   // (6 touches over 4 waves: waves 0, 1 issue two per tile)
      else wait_vmcnt<2 * G>();
    } else if (MODE == 2) {
      touch(t + P, wave - 2, 2);
    }
    __builtin_amdgcn_s_barrier();
    const float4 v = *reinterpret_cast<const float4 *>(lds + (t % kStages) * kTile + threadIdx.x * 16);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    __builtin_amdgcn_s_barrier();
  }
  wait_vmcnt<0>();
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[blockIdx.x] = acc.x;
}

template <int MODE, int P>
static void run(const char *name, const char *src, float *sink) {
  auto k = feed<MODE, P>;
  const int ldsb = kStages * kTile + 256;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(256), dim3(256), ldsb, 0, src, sink);
  hipEventRecord(a);
  const int reps = 50;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(256), dim3(256), ldsb, 0, src, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps, bytes = (double)kTiles * kTile * 256.0;
  printf("%-64s %7.1f us per launch   %5.1f B/clk/CU   (%.1f MB into LDS, 64 MB distinct)\n", name, us, bytes / 256.0 / (us * 2400.0), bytes / 1e6);
}

int main() {
  char *src; float *sink;
  hipMalloc(&src, 8 * kXcdBytes + 65536); hipMalloc(&sink, 4096);
  hipMemset(src, 1, 8 * kXcdBytes + 65536);
  run<0, 0>("all four waves request", src, sink);
  run<1, 0>("waves 0, 1 request", src, sink);
  run<2, 4>("waves 0, 1 request, waves 2, 3 touch tile t + 4", src, sink);
  run<2, 6>("waves 0, 1 request, waves 2, 3 touch tile t + 6", src, sink);
  run<2, 8>("waves 0, 1 request, waves 2, 3 touch tile t + 8", src, sink);
  run<2, 12>("waves 0, 1 request, waves 2, 3 touch tile t + 12", src, sink);
  run<3, 6>("all four request and touch tile t + 6 (in front of the requests)", src, sink);
  return 0;
}
