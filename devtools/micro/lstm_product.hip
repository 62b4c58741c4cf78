// How fast does ONE wave per SIMD issue the half-chain LSTM product (128 x v_mfma_f32_4x4x1 per wave and timestep, A fragments from LDS
// with a pinned read-ahead, B fragments resident in registers), alone on its SIMD and beside a second workgroup's wave?
// build: hipcc --offload-arch=gfx950 -O3 lstm_product.hip -o lstm_product
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>   // MODE 0: A from LDS (read-ahead 4), 1: A from registers (no LDS traffic), 2: A from LDS, reads all issued first
__global__ void __launch_bounds__(256, 2) prod(const float *w, float *out, int iters) {
  constexpr int KW = 128, NF = KW / 4, MP = 4 * KW + 4;
  __shared__ __attribute__((aligned(16))) float m_lds[4][MP];
  __shared__ float red[4][4][80];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, jl = lane & 3;
  for (int i = threadIdx.x; i < 4 * MP; i += 256) (&m_lds[0][0])[i] = 0.001f * (i % 97);
  f32x4 bw[NF];
#pragma unroll
  for (int i = 0; i < NF; i++) bw[i] = *reinterpret_cast<const f32x4 *>(w + (size_t)(lane * 512 + wave * KW + 4 * i));
  __syncthreads();
  const float *arow = &m_lds[jl][wave * KW];
  float keep = 0.f;
  for (int it = 0; it < iters; it++) {
    f32x4 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(arow);
#pragma unroll
      for (int i = 0; i < NF; i++) {
        constexpr int dummy = 0;
        acc[(4 * i + 0) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, bw[i].x, acc[(4 * i + 0) % NACC], 0, 0, 0);
        acc[(4 * i + 1) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, bw[i].y, acc[(4 * i + 1) % NACC], 0, 0, 0);
        acc[(4 * i + 2) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, bw[i].z, acc[(4 * i + 2) % NACC], 0, 0, 0);
        acc[(4 * i + 3) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, bw[i].w, acc[(4 * i + 3) % NACC], 0, 0, 0);
      }
    } else {
      constexpr int PD = MODE == 2 ? 8 : 4;
      f32x4 av[NF];
#pragma unroll
      for (int i = 0; i < PD; i++) av[i] = *reinterpret_cast<const f32x4 *>(arow + 4 * i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NF; i++) {
        if (i + PD < NF) av[i + PD] = *reinterpret_cast<const f32x4 *>(arow + 4 * (i + PD));
        __builtin_amdgcn_sched_barrier(0);
        acc[(4 * i + 0) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].x, bw[i].x, acc[(4 * i + 0) % NACC], 0, 0, 0);
        acc[(4 * i + 1) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].y, bw[i].y, acc[(4 * i + 1) % NACC], 0, 0, 0);
        acc[(4 * i + 2) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].z, bw[i].z, acc[(4 * i + 2) % NACC], 0, 0, 0);
        acc[(4 * i + 3) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[i].w, bw[i].w, acc[(4 * i + 3) % NACC], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    f32x4 sum = acc[0];
#pragma unroll
    for (int q = 1; q < NACC; q++) sum += acc[q];
    float *rp = &red[wave][0][lane];
    rp[0] = sum.x; rp[80] = sum.y; rp[160] = sum.z; rp[240] = sum.w;
    __syncthreads();
    keep += red[(wave + 1) & 3][lane & 3][lane];
    m_lds[jl][wave * KW + (it & 63)] = keep * 1e-9f;   // the next product depends on this one
    __builtin_amdgcn_wave_barrier();
  }
  out[blockIdx.x * 256 + threadIdx.x] = keep;
}

template <class K>
void run(const char *name, K kern, int grid, const float *w, float *out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, w, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, w, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s WG/CU %d: %.3f us per product (+ reduce + barrier), %.1f cycles per MFMA per wave at 2.4 GHz\n", name, grid / 256, ms * 1e3 / iters,
         ms * 1e-3 / iters * 2.4e9 / 128.0);
}
int main() {
  float *w, *out;
  hipMalloc(&w, 64 * 512 * 4 * sizeof(float)); hipMemset(w, 0, 64 * 512 * 4 * sizeof(float));
  hipMalloc(&out, 512 * 256 * sizeof(float));
  for (int rep = 0; rep < 2; rep++) {
    run("A from LDS, read-ahead 4, 4 acc", prod<4, 0>, 256, w, out);  run("A from LDS, read-ahead 4, 4 acc", prod<4, 0>, 512, w, out);
    run("A from LDS, read-ahead 4, 8 acc", prod<8, 0>, 256, w, out);  run("A from LDS, read-ahead 4, 8 acc", prod<8, 0>, 512, w, out);
    run("A from LDS, read-ahead 8, 8 acc", prod<8, 2>, 256, w, out);  run("A from LDS, read-ahead 8, 8 acc", prod<8, 2>, 512, w, out);
    run("A in registers, 4 acc", prod<4, 1>, 256, w, out);            run("A in registers, 4 acc", prod<4, 1>, 512, w, out);
    run("A in registers, 8 acc", prod<8, 1>, 256, w, out);            run("A in registers, 8 acc", prod<8, 1>, 512, w, out);
    run("A in registers, 16 acc", prod<16, 1>, 256, w, out);          run("A in registers, 16 acc", prod<16, 1>, 512, w, out);
  }
  return 0;
}
