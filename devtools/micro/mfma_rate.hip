// Sustained issue rate of fp32 MFMAs on gfx950: dependent chain vs independent accumulators, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(512) k32(float *out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(512) k16(float *out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(512) k4(float *out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; i++) for (int e = 0; e < 4; e++) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
void run(const char *name, K kern, int threads, int nacc, double flop_per_mfma) {
  float *out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 10, 1.0f, 2.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, 2.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfmas = 256.0 * (threads / 64) * iters * 8 * nacc;
  printf("%-34s waves/SIMD %d  %8.1f TFLOP/s   (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", name, threads / 256, mfmas * flop_per_mfma / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
  hipFree(out);
}
int main() {
  run("32x32x2  1 acc (dependent chain)", k32<1>, 256, 1, 4096); run("32x32x2  1 acc (dependent chain)", k32<1>, 512, 1, 4096);
  run("32x32x2  2 acc", k32<2>, 256, 2, 4096); run("32x32x2  2 acc", k32<2>, 512, 2, 4096);
  run("32x32x2  4 acc", k32<4>, 256, 4, 4096); run("32x32x2  4 acc", k32<4>, 512, 4, 4096);
  run("16x16x4  1 acc (dependent chain)", k16<1>, 256, 1, 2048); run("16x16x4  1 acc (dependent chain)", k16<1>, 512, 1, 2048);
  run("16x16x4  2 acc", k16<2>, 256, 2, 2048); run("16x16x4  4 acc", k16<4>, 256, 4, 2048);
  run("16x16x4  8 acc", k16<8>, 256, 8, 2048); run("16x16x4  8 acc", k16<8>, 512, 8, 2048);
  run("4x4x1_16B 1 acc (dependent chain)", k4<1>, 256, 1, 512); run("4x4x1_16B 2 acc", k4<2>, 256, 2, 512);
  run("4x4x1_16B 4 acc", k4<4>, 256, 4, 512); run("4x4x1_16B 8 acc", k4<8>, 256, 8, 512); run("4x4x1_16B 8 acc", k4<8>, 512, 8, 512);
  return 0;
}
