// Cost of one grid-wide step hand-off on MI355X: NB co-resident workgroups, each iteration every workgroup
// (a) publishes 1 KiB with write-through (sc1) stores, (b) bumps an agent-scope counter, (c) spins (relaxed, bounded)
// until all NB arrived, (d) reads `payload_kb` KiB of the others' data with sc1 loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) step_kernel(unsigned *counter, float *buf, int nb, int iters, int payload_floats, unsigned *fail, float *sink) {
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    // (a) 256 threads x 4 B = 1 KiB per workgroup, write-through
    __hip_atomic_store(buf + (size_t)(it & 1) * nb * 256 + blockIdx.x * 256 + tid, (float)(it + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(it + 1) * nb;
      unsigned spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 4000000u) { *fail = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (*fail) return;
    // (d) read the others' data
    const float *src = buf + (size_t)(it & 1) * nb * 256;
    for (int i = tid; i < payload_floats; i += 256) acc += __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  sink[blockIdx.x * 256 + tid] = acc;
}

int main(int argc, char **argv) {
  const int iters = 2000;
  for (int nb : {64, 128, 256}) {
    for (int payload_kb : {0, 16, 64}) {
      unsigned *counter, *fail;
      float *buf, *sink;
      hipMalloc(&counter, 4); hipMalloc(&fail, 4); hipMalloc(&buf, sizeof(float) * 2 * 256 * 256); hipMalloc(&sink, sizeof(float) * 256 * 256);
      hipMemset(counter, 0, 4); hipMemset(fail, 0, 4);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(step_kernel, dim3(nb), dim3(256), 0, 0, counter, buf, nb, iters, payload_kb * 256, fail, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned f = 0; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
      printf("workgroups %3d  payload %2d KiB: %.2f us per step%s\n", nb, payload_kb, ms * 1e3 / iters, f ? "  (SPIN TIMEOUT)" : "");
      hipFree(counter); hipFree(fail); hipFree(buf); hipFree(sink);
    }
  }
  return 0;
}
