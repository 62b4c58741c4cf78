// Discovers the operand / result layout of v_mfma_f32_4x4x1_16b_f32 empirically.
// build: hipcc --offload-arch=gfx950 -O2 mfma4x4_layout.hip -o mfma4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *d) {
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[threadIdx.x * 4 + r] = acc[r];
}
int main() {
  float ha[64], hb[64], hd[256];
  for (int l = 0; l < 64; l++) { ha[l] = 1 + l; hb[l] = 1000 + 7 * l; }
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  // hypothesis: block = l / 4; A_b[i] from lane 4b + i; B_b[j] from lane 4b + j; D lane l reg r = A_b[r] * B_b[l % 4]
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 4; r++) {
      const int blk = l / 4;
      if (hd[l * 4 + r] != ha[4 * blk + r] * hb[4 * blk + (l % 4)]) ok1 = 0;
      if (hd[l * 4 + r] != ha[4 * blk + (l % 4)] * hb[4 * blk + r]) ok2 = 0;
    }
  printf("D[lane][reg r] = A_blk[r] * B_blk[lane%%4]: %s\n", ok1 ? "YES" : "no");
  printf("D[lane][reg r] = A_blk[lane%%4] * B_blk[r]: %s\n", ok2 ? "YES" : "no");
  for (int l = 0; l < 8; l++) printf("lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
  return 0;
}
