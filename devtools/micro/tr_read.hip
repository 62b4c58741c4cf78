#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short *out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
  __syncthreads();
  int l = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = l * 8;
  else { // [k][m] image, pitch 128 B: lane p of 16-group: row p/4, col chunk
    int p = l & 15, g = (l >> 4) & 1, kh = l >> 5;
    addr = (8 * kh + (p >> 2)) * 128 + 2 * (16 * g + 4 * (p & 3));
  }
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)((__attribute__((address_space(3))) char *)lds + addr));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}
int main() {
  unsigned short *d; hipMalloc(&d, 512);
  for (int mode = 0; mode < 2; mode++) {
    k<<<1, 64>>>(d, mode);
    unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; l++) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  }
}
