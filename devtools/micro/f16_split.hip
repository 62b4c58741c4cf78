// Micro test: an fp32 product carried on v_mfma_f32_16x16x32_f16 with each operand split into two fp16 pieces.
//   a = a_hi + 2^-11 a_lo'   (a_hi = fp16(a), a_lo' = fp16((a - a_hi) 2^11));  w sc = w_hi + w_lo (sc = per-column power of two, w_lo = fp16(w sc - w_hi))
// A tile rows 0..7 = a_hi of stream r, rows 8..15 = a_lo' of stream r - 8; two instructions (B = w_hi, then B = w_lo) accumulate into ONE
// tile: rows 0..7 = a_hi w, rows 8..15 = a_lo' w.  Result = (R[s] + 2^-11 R[s + 8]) / sc, the halves joined by v_permlane32_swap (two
// result registers per swap, as kaldi-aslp_amd/csrc/rnn_persistent.hip lstm_seq_fwd_h does).
// Checks the operand / result layouts and the accuracy against a double product.  hipcc --offload-arch=gfx950 -O3 f16_split.hip -o f16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 512, S = 8, N = 16;

__global__ void k_split(const float *m, const float *w, float *out, float *out32) {   // m [S][K], w [N][K] (row n = column n of the product), out [S][N]
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  // column scale: max |w[n][:]| -> 2^13 .. 2^14
  float cmax = 0.f;
  for (int k = 0; k < K; k++) cmax = fmaxf(cmax, fabsf(w[r * K + k]));
  int e = 0;
  (void)frexpf(cmax, &e);                       // cmax = f 2^e, f in [0.5, 1)
  const float sc = cmax > 0.f ? ldexpf(1.f, 14 - e) : 1.f, inv_sc = cmax > 0.f ? ldexpf(1.f, e - 14) : 1.f;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 32) {
    half8 a, bh, bl;
    for (int j = 0; j < 8; j++) {
      const int k = k0 + 8 * g + j;
      const float av = m[(r & 7) * K + k];
      const _Float16 ah = (_Float16)av;
      const _Float16 al = (_Float16)((av - (float)ah) * 2048.f);
      a[j] = r < 8 ? ah : al;
      const float wv = w[r * K + k] * sc;
      const _Float16 wh = (_Float16)wv;
      bh[j] = wh;
      bl[j] = (_Float16)(wv - (float)wh);
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl, acc0, 0, 0, 0);
  }
  const float f = l < 32 ? inv_sc : inv_sc * 0x1p-11f;
  (void)acc1;
  for (int e4 = 0; e4 < 4; e4 += 2) {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[e4] * f), __float_as_uint(acc0[e4 + 1] * f), false, false);
    out[(4 * (g & 1) + e4 + (l >> 5)) * N + r] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);   // register e4's total in lanes 0..31, e4 + 1's in lanes 32..63
  }
  // plain fp32 for comparison (one lane per output)
  for (int o = l; o < S * N; o += 64) {
    float s32 = 0.f;
    for (int k = 0; k < K; k++) s32 = fmaf(m[(o / N) * K + k], w[(o % N) * K + k], s32);
    out32[o] = s32;
  }
}

int main() {
  std::vector<float> m(S * K), w(N * K);
  srand(5);
  double worst = 0, worst32 = 0;
  for (int trial = 0; trial < 4; trial++) {
    const float wscale = trial == 0 ? 0.05f : trial == 1 ? 3.0f : trial == 2 ? 1e-6f : 2e4f;   // trained-size, large, tiny, near the fp16 limit without the scale
    for (auto &v : m) v = 2.f * rand() / RAND_MAX - 1.f;
    for (auto &v : w) v = wscale * (2.f * rand() / RAND_MAX - 1.f) * (rand() % 7 == 0 ? 1e-3f : 1.f);
    float *dm, *dw, *dout, *d32;
    hipMalloc(&dm, m.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&dout, S * N * 4); hipMalloc(&d32, S * N * 4);
    hipMemcpy(dm, m.data(), m.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dm, dw, dout, d32);
    std::vector<float> out(S * N), o32(S * N);
    hipMemcpy(out.data(), dout, S * N * 4, hipMemcpyDeviceToHost); hipMemcpy(o32.data(), d32, S * N * 4, hipMemcpyDeviceToHost);
    double emax = 0, e32 = 0, ref_abs = 0;
    for (int s = 0; s < S; s++)
      for (int n = 0; n < N; n++) {
        double ref = 0, mag = 0;
        for (int k = 0; k < K; k++) { ref += (double)m[s * K + k] * w[n * K + k]; mag += fabs((double)m[s * K + k] * w[n * K + k]); }
        emax = fmax(emax, fabs(out[s * N + n] - ref) / mag);   // relative to sum |a w|: the scale rounding errors live on
        e32 = fmax(e32, fabs(o32[s * N + n] - ref) / mag);
        ref_abs = fmax(ref_abs, fabs(ref));
      }
    printf("weights ~%g: split-fp16 max err / sum|aw| = %.3g   plain fp32 fma chain = %.3g   (max |result| %.3g)\n", wscale, emax, e32, ref_abs);
    worst = fmax(worst, emax); worst32 = fmax(worst32, e32);
    hipFree(dm); hipFree(dw); hipFree(dout); hipFree(d32);
  }
  printf("worst: split %.3g, fp32 %.3g\n", worst, worst32);
  return worst < 1e-6 ? 0 : 1;
}
