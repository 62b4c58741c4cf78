// nn_fused.hip -- fused hot-path ops: BatchNormalization forward/backward and Xent::Eval.
//
// Both are HBM-bound.  The reference issues ~25 elementwise /
// gemv launches per BatchNormalization step and ~15 + 5 blocking reductions per Xent::Eval;
// here BN forward is ONE column-statistics pass + 1 write pass (optionally producing the following Sigmoid's
// output as well), BN backward 1 statistics pass + 1 write pass (optionally absorbing that Sigmoid's backward), and Xent one pass with the row held in registers.
#include <atomic>

#include "aslp_kernels.h"
#include "colreduce.h"
#include "common.h"
#include "split16.h"

namespace aslp {
unsigned *new_async_error_word(const char *what);  // runtime.cpp
namespace {

// Row r of a matrix as a float4 at column c, zero for rows at / past `end`.  The read itself is unconditional (clamped to the last valid
// row) and the zero a select on the VALUE: `r < end ? *p : zero` compiles to a pointer select against a zero kept in scratch memory and a
// flat load -- every kernel that spelled it that way carried a private segment.
__device__ __forceinline__ float4 load4_rows(const float *base, long ld, int r, int end, int c) {
  const int rc = r < end ? r : (end > 0 ? end - 1 : 0);
  float4 v = *reinterpret_cast<const float4 *>(base + (long)rc * ld + c);
  if (r >= end) v = make_float4(0.f, 0.f, 0.f, 0.f);
  return v;
}

// ---- BatchNormalization ---------------------------------------------------------------
// the ONE statistics pass: sum x and sum fl(x*x) in double for the running statistics
// (nnet-batch-normalization.h:216-220: the square is formed in float first), plus sum x*x with the
// product exact in double, from which the batch variance follows without a second pass over `in`:
//   var = E[x^2] - mean^2  evaluated in double (53-bit sums of 24-bit data: no cancellation problem),
// where the reference makes a second fp32 pass sum (x - mean)^2 (:193-204); the two agree to fp32
// rounding of the reference's own result, far inside the 1e-4 parity tolerance.
struct BnSum1F {
  static constexpr bool kVec = true;
  const float *in; int ld;
  template <int VW>
  __device__ void operator()(int r, int c, double (&acc)[3][VW]) const {
    float x[VW];
    loadv<VW>(in + (long)r * ld + c, x);
#pragma unroll
    for (int i = 0; i < VW; i++) {
      acc[0][i] += (double)x[i];
      acc[1][i] += (double)(x[i] * x[i]);
      acc[2][i] += (double)x[i] * (double)x[i];
    }
  }
};
struct BnSum1G {
  float inv_rows, floor_; float *mean, *inv_std; double *acc_means, *acc_vars;
  __device__ void operator()(int c, const double (&s)[3]) const {
    // the reference's mean is an fp32 gemv result scaled by 1/B; a double sum rounded once to
    // fp32 differs from it by < 1 ulp-of-sum.
    mean[c] = (float)s[0] * inv_rows;
    const double m = s[0] * (double)inv_rows;
    double var = s[2] * (double)inv_rows - m * m;
    var = var > 0.0 ? var : 0.0;
    inv_std[c] = 1.0f / sqrtf((float)var + floor_);
    if (acc_means) acc_means[c] += s[0];
    if (acc_vars) acc_vars[c] += s[1];
  }
};
// write pass: xhat = (x - mean) * inv_std ; out = xhat * gamma + beta ; optionally act = sigmoid(out) for a Sigmoid
// component fused behind the normalisation (then `out` itself may be NULL: nobody else reads it)
template <bool VEC>
__global__ void __launch_bounds__(kBlock) bn_normalize_kernel(const float *in, int ldi, float *out, int ldo, float *xhat, int ldx,
                                                              const float *mean, const float *inv_std, const float *scale,
                                                              const float *shift, int rows, int cols, float *act, int lda) {
  constexpr int W = VEC ? 4 : 1;
  int cw = cols / W;
  long n = (long)rows * cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / cw), c = (int)(i - (long)r * cw) * W;
    if (VEC) {
      float4 x = *reinterpret_cast<const float4 *>(in + (long)r * ldi + c);
      float4 m = *reinterpret_cast<const float4 *>(mean + c), s = *reinterpret_cast<const float4 *>(inv_std + c);
      float4 g = *reinterpret_cast<const float4 *>(scale + c), b = *reinterpret_cast<const float4 *>(shift + c);
      float4 h, o;
      h.x = (x.x - m.x) * s.x; h.y = (x.y - m.y) * s.y; h.z = (x.z - m.z) * s.z; h.w = (x.w - m.w) * s.w;
      o.x = h.x * g.x + b.x; o.y = h.y * g.y + b.y; o.z = h.z * g.z + b.z; o.w = h.w * g.w + b.w;
      if (xhat) *reinterpret_cast<float4 *>(xhat + (long)r * ldx + c) = h;
      if (out) *reinterpret_cast<float4 *>(out + (long)r * ldo + c) = o;
      if (act) {
        float4 y;
        y.x = sigmoid_ref(o.x); y.y = sigmoid_ref(o.y); y.z = sigmoid_ref(o.z); y.w = sigmoid_ref(o.w);
        *reinterpret_cast<float4 *>(act + (long)r * lda + c) = y;
      }
    } else {
      float h = (in[(long)r * ldi + c] - mean[c]) * inv_std[c];
      if (xhat) xhat[(long)r * ldx + c] = h;
      const float o = h * scale[c] + shift[c];
      if (out) out[(long)r * ldo + c] = o;
      if (act) act[(long)r * lda + c] = sigmoid_ref(o);
    }
  }
}

// backward statistics: S1 = sum dy, S2 = sum xhat*dy
// `y` != NULL: a Sigmoid is fused behind the normalisation and `dy` is the diff w.r.t. ITS output;
// the diff w.r.t. the BN output is dy * y * (1 - y), formed on the fly (nnet-activation.h:170-173)
struct BnBwdF {
  static constexpr bool kVec = true;
  const float *dy; int ldd; const float *xhat; int ldx; const float *y; int ldy;
  template <int VW>
  __device__ void operator()(int r, int c, float (&acc)[2][VW]) const {
    float d[VW], h[VW];
    loadv<VW>(dy + (long)r * ldd + c, d);
    loadv<VW>(xhat + (long)r * ldx + c, h);
    if (y) {
      float yy[VW];
      loadv<VW>(y + (long)r * ldy + c, yy);
#pragma unroll
      for (int i = 0; i < VW; i++) d[i] = d[i] * yy[i] * (1.0f - yy[i]);
    }
#pragma unroll
    for (int i = 0; i < VW; i++) {
      acc[0][i] += d[i];
      acc[1][i] += h[i] * d[i];
    }
  }
};
// `step`: the component's SGD step (nnet-batch-normalization.h:280-284) is taken right here instead of in a launch of
// its own; the write pass that follows still needs the scale the forward pass used, so it is kept in `scale_used`.
struct BnBwdG {
  float mmt; float *dscale, *dshift; float *s1, *s2;
  float *scale, *shift, *scale_used; float neg_lr; bool step;
  __device__ void operator()(int c, const float (&s)[2]) const {
    const float dsh = s[0] + mmt * dshift[c], dsc = s[1] + mmt * dscale[c];
    dshift[c] = dsh;
    dscale[c] = dsc;
    s1[c] = s[0];
    s2[c] = s[1];
    const float g = scale[c];
    scale_used[c] = g;
    if (step) {
      scale[c] = g + neg_lr * dsc;
      shift[c] += neg_lr * dsh;
    }
  }
};
// in_diff: with D = dy*gamma, the reference's 4 steps (:238-276) reduce to
//   dvar  = -0.5 * inv^3 * sum (x-mean) * D      = -0.5 * inv^2 * gamma * S2
//   dmean = -inv * gamma * S1  ( - (2/B) dvar * sum(x-mean), which is 0 up to rounding )
//   in_diff = D*inv + (x-mean) * (2/B) * dvar + dmean/B,   (x-mean) = xhat/inv
template <bool VEC>
__global__ void __launch_bounds__(kBlock) bn_backward_kernel(const float *dy, int ldd, float *xhat, int ldx, const float *scale,
                                                             const float *inv_std, const float *s1, const float *s2, float *in_diff,
                                                             int ldi, int rows, int cols, const float *y, int ldy) {
  constexpr int W = VEC ? 4 : 1;
  int cw = cols / W;
  long n = (long)rows * cw;
  const float invB = 1.0f / (float)rows;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / cw), c = (int)(i - (long)r * cw) * W;
    float dyv[W], hv[W], gv[W], iv[W], s1v[W], s2v[W], Dv[W], ov[W];
    if (VEC) {
      *reinterpret_cast<float4 *>(dyv) = *reinterpret_cast<const float4 *>(dy + (long)r * ldd + c);
      *reinterpret_cast<float4 *>(hv) = *reinterpret_cast<const float4 *>(xhat + (long)r * ldx + c);
      *reinterpret_cast<float4 *>(gv) = *reinterpret_cast<const float4 *>(scale + c);
      *reinterpret_cast<float4 *>(iv) = *reinterpret_cast<const float4 *>(inv_std + c);
      *reinterpret_cast<float4 *>(s1v) = *reinterpret_cast<const float4 *>(s1 + c);
      *reinterpret_cast<float4 *>(s2v) = *reinterpret_cast<const float4 *>(s2 + c);
    } else {
      dyv[0] = dy[(long)r * ldd + c]; hv[0] = xhat[(long)r * ldx + c]; gv[0] = scale[c]; iv[0] = inv_std[c];
      s1v[0] = s1[c]; s2v[0] = s2[c];
    }
    if (y) {
      float yv[W];
      if (VEC) *reinterpret_cast<float4 *>(yv) = *reinterpret_cast<const float4 *>(y + (long)r * ldy + c);
      else yv[0] = y[(long)r * ldy + c];
#pragma unroll
      for (int k = 0; k < W; k++) dyv[k] = dyv[k] * yv[k] * (1.0f - yv[k]);
    }
#pragma unroll
    for (int k = 0; k < W; k++) {
      float g = gv[k], inv = iv[k];
      float D = dyv[k] * g;
      float dvar = -0.5f * inv * inv * g * s2v[k];         // = sum (x-mean) D * (-0.5 inv^3)
      float dmean = -inv * g * s1v[k];
      float xm = hv[k] / inv;
      Dv[k] = D;  // XsharpO_ <- dy*gamma, like the reference (:241-242)
      ov[k] = D * inv + xm * (2.0f * invB) * dvar + invB * dmean;
    }
    if (VEC) {
      *reinterpret_cast<float4 *>(xhat + (long)r * ldx + c) = *reinterpret_cast<float4 *>(Dv);
      *reinterpret_cast<float4 *>(in_diff + (long)r * ldi + c) = *reinterpret_cast<float4 *>(ov);
    } else {
      xhat[(long)r * ldx + c] = Dv[0];
      in_diff[(long)r * ldi + c] = ov[0];
    }
  }
}

// ---- BatchNormalization, panel-resident variants -----------------------------------------------------------------
// For minibatch-sized inputs (rows <= SLOTS * row lanes) a workgroup owns a panel of 4*CG columns over ALL rows and
// keeps it in registers: statistics, their finalize and the write pass are one launch that reads its operands once, instead
// of statistics stage 1 + stage 2 + write pass (3 launches, operands read twice).  These passes are latency-bound at
// [1024 x 2048] (8 MB, largely still in L2 / MALL behind the GEMM that produced it), so launches are what counts; each thread
// holds SLOTS rows of one 4-column group.
// Thread layout: column group fastest (CG adjacent threads read 16*CG contiguous bytes of a row), kPanelThreads / CG row
// lanes; a column's partial sums meet by wave shuffles, then across the waves in LDS, always in the same order.
constexpr int kPanelThreads = 256;
constexpr int kPanelWaves = kPanelThreads / kWave;

template <typename T, int N, int CG>
__device__ __forceinline__ void panel_reduce(T (&acc)[N][4], T *red /* [4 waves][CG][N*4] */) {
#pragma unroll
  for (int off = CG; off < kWave; off <<= 1)
#pragma unroll
    for (int a = 0; a < N; a++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[a][i] += __shfl_xor(acc[a][i], off);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  if (lane < CG)
#pragma unroll
    for (int a = 0; a < N; a++)
#pragma unroll
      for (int i = 0; i < 4; i++) red[(wave * CG + lane) * (N * 4) + a * 4 + i] = acc[a][i];
  __syncthreads();
}
// after panel_reduce: total of accumulator a for panel column pc (0 .. 4*CG-1)
template <typename T, int N, int CG>
__device__ __forceinline__ T panel_total(const T *red, int a, int pc) {
  const int cg = pc / 4, i = pc % 4;
  T s = red[(0 * CG + cg) * (N * 4) + a * 4 + i];
#pragma unroll
  for (int w = 1; w < kPanelWaves; w++) s += red[(w * CG + cg) * (N * 4) + a * 4 + i];
  return s;
}

template <int CG, int SLOTS>
__global__ void __launch_bounds__(kPanelThreads) bn_forward_panel(const float *__restrict__ in, int ldi, float *__restrict__ out, int ldo,
                                                           float *__restrict__ xhat, int ldx, const float *__restrict__ scale,
                                                           const float *__restrict__ shift, float *__restrict__ mean,
                                                           float *__restrict__ inv_std, double *__restrict__ acc_means,
                                                           double *__restrict__ acc_vars, float inv_rows, float floor_, int rows,
                                                           float *__restrict__ act, int lda) {
  constexpr int L = kPanelThreads / CG;
  __shared__ double red[kPanelWaves * CG * 12];
  __shared__ float stat[2][4 * CG];
  const int cg = threadIdx.x % CG, lane = threadIdx.x / CG;
  const int c = ((int)blockIdx.x * CG + cg) * 4;
  float4 x[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    const int r = lane + k * L;
    x[k] = load4_rows(in, ldi, r, rows, c);
  }
  const float4 g = *reinterpret_cast<const float4 *>(scale + c), b = *reinterpret_cast<const float4 *>(shift + c);   // (in the rows' round trip)
  double acc[3][4] = {};
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    const float v[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc[0][i] += (double)v[i];
      acc[1][i] += (double)(v[i] * v[i]);
      acc[2][i] += (double)v[i] * (double)v[i];
    }
  }
  panel_reduce<double, 3, CG>(acc, red);
  if (threadIdx.x < 4 * CG) {  // the finalize of BnSum1G for one column
    const int pc = threadIdx.x, col = (int)blockIdx.x * CG * 4 + pc;
    const double s0 = panel_total<double, 3, CG>(red, 0, pc), s1 = panel_total<double, 3, CG>(red, 1, pc), s2 = panel_total<double, 3, CG>(red, 2, pc);
    const float mu = (float)s0 * inv_rows;
    const double m = s0 * (double)inv_rows;
    double var = s2 * (double)inv_rows - m * m;
    var = var > 0.0 ? var : 0.0;
    const float is = 1.0f / sqrtf((float)var + floor_);
    mean[col] = mu;
    inv_std[col] = is;
    if (acc_means) acc_means[col] += s0;
    if (acc_vars) acc_vars[col] += s1;
    stat[0][pc] = mu;
    stat[1][pc] = is;
  }
  __syncthreads();
  const float4 m = *reinterpret_cast<const float4 *>(&stat[0][cg * 4]), is = *reinterpret_cast<const float4 *>(&stat[1][cg * 4]);
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    const int r = lane + k * L;
    if (r >= rows) break;
    float4 h, o;
    h.x = (x[k].x - m.x) * is.x; h.y = (x[k].y - m.y) * is.y; h.z = (x[k].z - m.z) * is.z; h.w = (x[k].w - m.w) * is.w;
    o.x = h.x * g.x + b.x; o.y = h.y * g.y + b.y; o.z = h.z * g.z + b.z; o.w = h.w * g.w + b.w;
    if (xhat) *reinterpret_cast<float4 *>(xhat + (long)r * ldx + c) = h;
    if (out) *reinterpret_cast<float4 *>(out + (long)r * ldo + c) = o;
    if (act) {
      float4 y;
      y.x = sigmoid_ref(o.x); y.y = sigmoid_ref(o.y); y.z = sigmoid_ref(o.z); y.w = sigmoid_ref(o.w);
      *reinterpret_cast<float4 *>(act + (long)r * lda + c) = y;
    }
  }
}

// RECOMPUTE: no normalised copy of the input was kept by the forward pass; x_hat = (x - mean) * inv_std is formed again from
// the layer input with the forward kernel's own two operations (same bits), and the in-place D = dy * scale the reference
// leaves in that buffer is not written at all -- 8.4 MB less traffic each way at [1024 x 2048].
template <int CG, int SLOTS, bool HAS_Y, bool RECOMPUTE>
__global__ void __launch_bounds__(kPanelThreads) bn_backward_panel(const float *__restrict__ dy, int ldd, float *__restrict__ xhat, int ldx,
                                                            float *__restrict__ scale, float *__restrict__ shift,
                                                            const float *__restrict__ inv_std, float *__restrict__ dscale,
                                                            float *__restrict__ dshift, float mmt, float neg_lr, bool step,
                                                            float *__restrict__ in_diff, int ldi, int rows, const float *__restrict__ y, int ldy,
                                                            const float *__restrict__ xin, int ldxin, const float *__restrict__ mean) {
  constexpr int L = kPanelThreads / CG;
  __shared__ float red[kPanelWaves * CG * 8];
  __shared__ float stat[3][4 * CG];  // S1, S2, the scale the forward pass used
  const int cg = threadIdx.x % CG, lane = threadIdx.x / CG;
  const int c = ((int)blockIdx.x * CG + cg) * 4;
  float4 d[SLOTS], h[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    const int r = lane + k * L;
    d[k] = load4_rows(dy, ldd, r, rows, c);
    if (RECOMPUTE) h[k] = load4_rows(xin, ldxin, r, rows, c);
    else h[k] = load4_rows(xhat, ldx, r, rows, c);
  }
  if (RECOMPUTE) {
    const float4 m = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(inv_std + c);
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      const int r = lane + k * L;
      if (r < rows) { h[k].x = (h[k].x - m.x) * is.x; h[k].y = (h[k].y - m.y) * is.y; h[k].z = (h[k].z - m.z) * is.z; h[k].w = (h[k].w - m.w) * is.w; }
    }
  }
  if (HAS_Y) {
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      const int r = lane + k * L;
      const float4 yy = load4_rows(y, ldy, r, rows, c);
      {
        // products rounded on their own, never contracted into the sums below: the values a separate Sigmoid backward would
        // have stored, so the folded and the unfolded executor agree bit for bit (HIP's __fmul_rn is a plain multiply)
#pragma clang fp contract(off)
        d[k].x = d[k].x * yy.x * (1.0f - yy.x); d[k].y = d[k].y * yy.y * (1.0f - yy.y);
        d[k].z = d[k].z * yy.z * (1.0f - yy.z); d[k].w = d[k].w * yy.w * (1.0f - yy.w);
      }
    }
  }
  float acc[2][4] = {};
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    acc[0][0] += d[k].x; acc[0][1] += d[k].y; acc[0][2] += d[k].z; acc[0][3] += d[k].w;
    acc[1][0] += h[k].x * d[k].x; acc[1][1] += h[k].y * d[k].y; acc[1][2] += h[k].z * d[k].z; acc[1][3] += h[k].w * d[k].w;
  }
  panel_reduce<float, 2, CG>(acc, red);
  if (threadIdx.x < 4 * CG) {  // BnBwdG for one column
    const int pc = threadIdx.x, col = (int)blockIdx.x * CG * 4 + pc;
    const float s1 = panel_total<float, 2, CG>(red, 0, pc), s2 = panel_total<float, 2, CG>(red, 1, pc);
    const float dsh = s1 + mmt * dshift[col], dsc = s2 + mmt * dscale[col];
    dshift[col] = dsh;
    dscale[col] = dsc;
    const float g = scale[col];
    if (step) {
      scale[col] = g + neg_lr * dsc;
      shift[col] += neg_lr * dsh;
    }
    stat[0][pc] = s1;
    stat[1][pc] = s2;
    stat[2][pc] = g;
  }
  __syncthreads();
  if (in_diff == nullptr) return;
  const float invB = 1.0f / (float)rows;
  const float4 iv4 = *reinterpret_cast<const float4 *>(inv_std + c);
  const float iv[4] = {iv4.x, iv4.y, iv4.z, iv4.w};
  float gv[4], ca[4], cb[4];  // per column: in_diff = D*inv + xm * ca + cb
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float g = stat[2][cg * 4 + i], inv = iv[i];
    const float dvar = -0.5f * inv * inv * g * stat[1][cg * 4 + i];
    const float dmean = -inv * g * stat[0][cg * 4 + i];
    gv[i] = g;
    ca[i] = (2.0f * invB) * dvar;
    cb[i] = invB * dmean;
  }
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    const int r = lane + k * L;
    if (r >= rows) break;
    const float dv[4] = {d[k].x, d[k].y, d[k].z, d[k].w}, hv[4] = {h[k].x, h[k].y, h[k].z, h[k].w};
    float Dv[4], ov[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float D = dv[i] * gv[i];
      const float xm = hv[i] / iv[i];
      Dv[i] = D;
      ov[i] = D * iv[i] + xm * ca[i] + cb[i];
    }
    if (!RECOMPUTE) *reinterpret_cast<float4 *>(xhat + (long)r * ldx + c) = make_float4(Dv[0], Dv[1], Dv[2], Dv[3]);
    *reinterpret_cast<float4 *>(in_diff + (long)r * ldi + c) = make_float4(ov[0], ov[1], ov[2], ov[3]);
  }
}

// ---- cooperative panels ---------------------------------------------------------------------------------------------
// The panel kernels above give a workgroup 16 columns over ALL rows: 128 workgroups for [1024 x 2048], half the chip idle and
// 64-byte row segments (measured 1.5-2.2 TB/s).  Here a column panel is 32 columns (one 128-byte line per row) and is shared
// by Q workgroups, each holding rows/Q rows in registers: 64 x 4 = 256 workgroups.  The Q partial statistics of a panel meet
// through a tiny inbox in global memory: every workgroup stores its partial into the inbox of each of the Q readers
// (agent-scope stores), every reader polls its own inbox until all Q partials are there, adds them in workgroup order
// (so all Q readers -- and every run -- get the same bits), and puts the "nothing here" pattern back before it leaves.
// The launch is ordered behind the previous one on its stream, so a reader-owned reset needs no further protocol.
// Workgroups p, p + P, p + 2P, ... share a panel: with P a multiple of 8 they sit on one XCD (speed only).
constexpr int kCoopCG = 8, kCoopCols = 32, kCoopLanes = kPanelThreads / kCoopCG;
constexpr unsigned long long kNothing = 0xFFFFFFFFFFFFFFFFull;
constexpr int kCoopSpinLimit = 1 << 22;  // polls before a reader gives up (seconds): the error word is raised, the output is garbage

template <int NW>  // 8-byte words per column
__device__ __forceinline__ bool coop_exchange(unsigned long long *inbox, int P, int Q, int p, int q, int pc, const unsigned long long (&mine)[NW],
                                              unsigned long long (*got)[NW] /* [Q][NW] */, unsigned *err) {
  // writer: my partial for column pc into every reader's inbox
  for (int qr = 0; qr < Q; qr++) {
    unsigned long long *dst = inbox + ((((size_t)p * Q + qr) * Q + q) * kCoopCols + pc) * NW;
#pragma unroll
    for (int w = 0; w < NW; w++) __hip_atomic_store(dst + w, mine[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // reader: all Q partials of column pc
  unsigned long long *src = inbox + (((size_t)p * Q + q) * Q * kCoopCols + pc) * NW;
  bool ok = true;
  for (int qw = 0; qw < Q; qw++) {
    unsigned long long *w0 = src + (size_t)qw * kCoopCols * NW;
    int spins = 0;
    for (;;) {
      bool all = true;
#pragma unroll
      for (int w = 0; w < NW; w++) { got[qw][w] = __hip_atomic_load(w0 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); all &= got[qw][w] != kNothing; }
      if (all) break;
      if (++spins > kCoopSpinLimit) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int w = 0; w < NW; w++) __hip_atomic_store(w0 + w, kNothing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // hand the slot back
  }
  if (!ok && pc == 0) __hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return ok;
}

template <int SLOTS>
__global__ void __launch_bounds__(kPanelThreads) bn_forward_coop(const float *__restrict__ in, int ldi, float *__restrict__ out, int ldo,
                                                          float *__restrict__ xhat, int ldx, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, float *__restrict__ mean,
                                                          float *__restrict__ inv_std, double *__restrict__ acc_means,
                                                          double *__restrict__ acc_vars, float inv_rows, float floor_, int rows,
                                                          float *__restrict__ act, int lda, int Q, unsigned long long *inbox, unsigned *err) {
  constexpr int CG = kCoopCG, L = kCoopLanes;
  __shared__ double red[kPanelWaves * CG * 12];
  __shared__ float stat[2][4 * CG];
  const int P = gridDim.x / Q, p = blockIdx.x % P, q = blockIdx.x / P;
  const int rp = (rows + Q - 1) / Q, r0 = q * rp, r1 = min(rows, r0 + rp);
  const int cg = threadIdx.x % CG, lane = threadIdx.x / CG;
  const int c = (p * CG + cg) * 4;
  float4 x[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    const int r = r0 + lane + k * L;
    x[k] = load4_rows(in, ldi, r, r1, c);
  }
  const float4 g = *reinterpret_cast<const float4 *>(scale + c), b = *reinterpret_cast<const float4 *>(shift + c);   // (in the rows' round trip)
  double acc[3][4] = {};
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    const float v[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc[0][i] += (double)v[i];
      acc[1][i] += (double)(v[i] * v[i]);
      acc[2][i] += (double)v[i] * (double)v[i];
    }
  }
  panel_reduce<double, 3, CG>(acc, red);
  if (threadIdx.x < 4 * CG) {
    const int pc = threadIdx.x, col = p * CG * 4 + pc;
    unsigned long long mine[3], got[8][3];
#pragma unroll
    for (int a = 0; a < 3; a++) mine[a] = (unsigned long long)__double_as_longlong(panel_total<double, 3, CG>(red, a, pc));
    coop_exchange<3>(inbox, P, Q, p, q, pc, mine, got, err);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int qw = 0; qw < Q; qw++) {  // workgroup order: the same bits in all Q readers
      s0 += __longlong_as_double((long long)got[qw][0]);
      s1 += __longlong_as_double((long long)got[qw][1]);
      s2 += __longlong_as_double((long long)got[qw][2]);
    }
    const float mu = (float)s0 * inv_rows;
    const double m = s0 * (double)inv_rows;
    double var = s2 * (double)inv_rows - m * m;
    var = var > 0.0 ? var : 0.0;
    const float is = 1.0f / sqrtf((float)var + floor_);
    if (q == 0) {
      mean[col] = mu;
      inv_std[col] = is;
      if (acc_means) acc_means[col] += s0;
      if (acc_vars) acc_vars[col] += s1;
    }
    stat[0][pc] = mu;
    stat[1][pc] = is;
  }
  __syncthreads();
  const float4 m = *reinterpret_cast<const float4 *>(&stat[0][cg * 4]), is = *reinterpret_cast<const float4 *>(&stat[1][cg * 4]);
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    const int r = r0 + lane + k * L;
    if (r >= r1) break;
    float4 h, o;
    h.x = (x[k].x - m.x) * is.x; h.y = (x[k].y - m.y) * is.y; h.z = (x[k].z - m.z) * is.z; h.w = (x[k].w - m.w) * is.w;
    o.x = h.x * g.x + b.x; o.y = h.y * g.y + b.y; o.z = h.z * g.z + b.z; o.w = h.w * g.w + b.w;
    if (xhat) *reinterpret_cast<float4 *>(xhat + (long)r * ldx + c) = h;
    if (out) *reinterpret_cast<float4 *>(out + (long)r * ldo + c) = o;
    if (act) {
      float4 y;
      y.x = sigmoid_ref(o.x); y.y = sigmoid_ref(o.y); y.z = sigmoid_ref(o.z); y.w = sigmoid_ref(o.w);
      *reinterpret_cast<float4 *>(act + (long)r * lda + c) = y;
    }
  }
}

// BatchNormalization forward whose column statistics come from the producer of `in` (the forward GEMM's epilogue,
// aslp_gemm_epilogue.colstats: `groups` partial sums per column and statistic).  No statistics pass over the matrix, no exchange
// between workgroups: a workgroup adds the partials of its 32 columns (group order, double), then streams its rows once.
// Grid: (cols / 32) x Q row chunks like bn_forward_coop; every row chunk derives the same statistics, chunk 0 publishes them.
constexpr int kStatSlots = 4;
__global__ void __launch_bounds__(kPanelThreads) bn_forward_stats_kernel(const float *__restrict__ in, int ldi, float *__restrict__ out, int ldo,
                                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                                        float *__restrict__ mean, float *__restrict__ inv_std,
                                                                        double *__restrict__ acc_means, double *__restrict__ acc_vars, float inv_rows,
                                                                        float floor_, int rows, float *__restrict__ act, int lda, int Q,
                                                                        const double *__restrict__ part, int groups, int ldp, S16Out po) {
  constexpr int CG = kCoopCG, L = kCoopLanes, COLS = kCoopCols, SL = kPanelThreads / COLS;  // SL partial-sum slices per column
  __shared__ double red[3][SL][COLS];
  __shared__ float stat[2][COLS];
  const int P = gridDim.x / Q, p = blockIdx.x % P, q = blockIdx.x / P;
  const int rp = (rows + Q - 1) / Q, r0 = q * rp, r1 = min(rows, r0 + rp);
  const int cg = threadIdx.x % CG, lane = threadIdx.x / CG;
  const int c = (p * CG + cg) * 4;
  // the rows first (<= kStatSlots per thread, the host sizes Q for that): their latency overlaps the statistics below
  float4 x[kStatSlots];
#pragma unroll
  for (int k = 0; k < kStatSlots; k++) {
    const int r = r0 + lane + k * L;
    x[k] = load4_rows(in, ldi, r, r1, c);
  }
  // everything else this workgroup reads goes out behind them in the same round: scale and shift, the planes' bound, and the statistics'
  // partial sums four groups at a time (a loop that adds one group per iteration is one round trip to L2 per iteration)
  const float4 g = *reinterpret_cast<const float4 *>(scale + c), b = *reinterpret_cast<const float4 *>(shift + c);
  const unsigned bound_bits = po.hi ? *po.slot : 0u;
  {
    const int pc = threadIdx.x % COLS, sl = threadIdx.x / COLS, col = p * COLS + pc;
    const long plane = (long)groups * ldp;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int g0 = sl; g0 < groups; g0 += 4 * SL) {
      double v[4][3];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int gi = g0 + u * SL;
        const double *pp = part + (long)(gi < groups ? gi : g0) * ldp + col;
        v[u][0] = pp[0]; v[u][1] = pp[plane]; v[u][2] = pp[2 * plane];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (g0 + u * SL < groups) { s0 += v[u][0]; s1 += v[u][1]; s2 += v[u][2]; }   // (group order, as before)
    }
    red[0][sl][pc] = s0; red[1][sl][pc] = s1; red[2][sl][pc] = s2;
  }
  __syncthreads();
  if (threadIdx.x < COLS) {
    const int pc = threadIdx.x, col = p * COLS + pc;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int sl = 0; sl < SL; sl++) { s0 += red[0][sl][pc]; s1 += red[1][sl][pc]; s2 += red[2][sl][pc]; }
    const float mu = (float)s0 * inv_rows;   // as bn_forward_coop
    const double m = s0 * (double)inv_rows;
    double var = s2 * (double)inv_rows - m * m;
    var = var > 0.0 ? var : 0.0;
    const float is = 1.0f / sqrtf((float)var + floor_);
    if (q == 0) {
      mean[col] = mu;
      inv_std[col] = is;
      if (acc_means) acc_means[col] += s0;
      if (acc_vars) acc_vars[col] += s1;
    }
    stat[0][pc] = mu;
    stat[1][pc] = is;
  }
  __syncthreads();
  const float4 m = *reinterpret_cast<const float4 *>(&stat[0][cg * 4]), is = *reinterpret_cast<const float4 *>(&stat[1][cg * 4]);
  const float pscale = po.hi ? ldexpf(1.f, s16_exponent(bound_bits)) : 0.f;
#pragma unroll
  for (int k = 0; k < kStatSlots; k++) {
    const int r = r0 + lane + k * L;
    if (r >= r1) break;
    float4 h, o;
    h.x = (x[k].x - m.x) * is.x; h.y = (x[k].y - m.y) * is.y; h.z = (x[k].z - m.z) * is.z; h.w = (x[k].w - m.w) * is.w;
    o.x = h.x * g.x + b.x; o.y = h.y * g.y + b.y; o.z = h.z * g.z + b.z; o.w = h.w * g.w + b.w;
    if (out) *reinterpret_cast<float4 *>(out + (long)r * ldo + c) = o;
    if (act) {
      float4 y;
      y.x = sigmoid_ref(o.x); y.y = sigmoid_ref(o.y); y.z = sigmoid_ref(o.z); y.w = sigmoid_ref(o.w);
      *reinterpret_cast<float4 *>(act + (long)r * lda + c) = y;
      if (po.hi) {   // the planes of the activations for the products that read them (bound: a sigmoid's 1)
        half4 hi, lo;
        s16_split4(y, pscale, &hi, &lo);
        *reinterpret_cast<half4 *>(po.hi + (long)r * po.ld + c) = hi;
        *reinterpret_cast<half4 *>(po.lo + (long)r * po.ld + c) = lo;
      }
    }
  }
}

// The largest of one value per workgroup, in every workgroup of a launch whose workgroups are all resident at once (the host checks: they
// wait for each other).  They meet in gmax, one 8-byte word each: this launch's token | the value's bits -- nothing to reset, an older
// launch's word never matches.  wg_max: the workgroup's value (in every thread); red: >= kPanelWaves floats of LDS nobody else is using.
__device__ __forceinline__ float coop_grid_max(float wg_max, unsigned long long *gmax, unsigned token, unsigned *err, float *red) {
  if (threadIdx.x == 0)
    __hip_atomic_store(gmax + blockIdx.x, ((unsigned long long)token << 32) | (unsigned long long)__float_as_uint(wg_max), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  float gm = 0.f;
  bool ok = true;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += kPanelThreads) {
    int spins = 0;
    for (;;) {
      const unsigned long long v = __hip_atomic_load(gmax + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(v >> 32) == token) { gm = fmaxf(gm, __uint_as_float((unsigned)v)); break; }
      if (++spins > kCoopSpinLimit) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (!ok) __hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  gm = wave_max(gm);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gm;
  __syncthreads();
  gm = red[0];
  for (int w = 1; w < kPanelWaves; w++) gm = fmaxf(gm, red[w]);
  return gm;
}

// Matrices (up to kS16MaxJobs, blockIdx.y) copied (optional) and their fp16 planes made in ONE launch (coop_convert_launch: the network
// input -- copy + maximum pass + conversion pass were three launches in front of the first layer product -- and PlaneSet::ConvertFrom /
// ConvertMany): a thread holds U 16-byte pieces of its matrix, the workgroups of a matrix find its maximum among themselves as above.  A
// matrix whose bound is given (parts) takes it from there.
struct CoopConvJobs { CoopConvJob j[kS16MaxJobs]; };
template <int U>
__global__ void __launch_bounds__(kPanelThreads) copy_planes_coop(CoopConvJobs jobs, unsigned long long *gmax, unsigned token, unsigned *err, SeqFillJob fill) {
  __shared__ float red[kPanelWaves];
  if (fill.buf0 != nullptr) {   // (uniform) a recurrent layer's buffer preparation rides along: stores only, spread over the launch's workgroups
    const int per = (fill.T + 2) * fill.S, nb = (int)(gridDim.x * gridDim.y);
    for (int r = (int)(blockIdx.y * gridDim.x + blockIdx.x); r < 2 * per; r += nb) seq_fill_row(fill, r % per, r / per);
  }
  const CoopConvJob job = jobs.j[blockIdx.y];
  const int rows = job.pl.rows, cols = job.pl.cols, c4 = cols >> 2, units = rows * c4;
  float4 v[U];
  int rr[U], cc[U];
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < U; k++) {
    const int u = ((int)blockIdx.x * U + k) * kPanelThreads + (int)threadIdx.x;
    const int r = u / c4;
    rr[k] = u < units ? r : -1;
    cc[k] = 4 * (u - r * c4);
    v[k] = u < units ? *reinterpret_cast<const float4 *>(job.src + (long)r * job.ld_src + cc[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (job.parts != nullptr) {   // (uniform) the bound is known: every workgroup reads it
    for (int i0 = threadIdx.x; i0 < job.nparts; i0 += 4 * kPanelThreads) {
      float p[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { const int i = i0 + q * kPanelThreads; p[q] = i < job.nparts ? job.parts[i] : 0.f; }
      m = fmaxf(fmaxf(m, fmaxf(p[0], p[1])), fmaxf(p[2], p[3]));
    }
  }
#pragma unroll
  for (int k = 0; k < U; k++) {
    if (rr[k] >= 0 && job.dst != nullptr) *reinterpret_cast<float4 *>(job.dst + (long)rr[k] * job.ld_dst + cc[k]) = v[k];
    if (job.parts == nullptr) m = s16_absmax4(m, v[k]);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = red[0];
  for (int w = 1; w < kPanelWaves; w++) m = fmaxf(m, red[w]);
  unsigned mbits = __float_as_uint(m);
  if (job.parts == nullptr) mbits = __float_as_uint(coop_grid_max(m, gmax + (size_t)blockIdx.y * gridDim.x, token, err, red));
  if (blockIdx.x == 0 && threadIdx.x == 0) *job.pl.slot = mbits;
  const float ps = ldexpf(1.f, s16_exponent(mbits));
#pragma unroll
  for (int k = 0; k < U; k++) {
    if (rr[k] < 0) continue;
    half4 hi, lo;
    s16_split4(v[k], ps, &hi, &lo);
    *reinterpret_cast<half4 *>(job.pl.hi + (long)rr[k] * job.pl.ld + cc[k]) = hi;
    *reinterpret_cast<half4 *>(job.pl.lo + (long)rr[k] * job.pl.ld + cc[k]) = lo;
  }
}

// what bn_backward_coop needs to bound its in-diff before it has written it (see there)
template <int SLOTS, bool HAS_Y, bool RECOMPUTE>
__global__ void __launch_bounds__(kPanelThreads) bn_backward_coop(const float *__restrict__ dy, int ldd, float *__restrict__ xhat, int ldx,
                                                           float *__restrict__ scale, float *__restrict__ shift,
                                                           const float *__restrict__ inv_std, float *__restrict__ dscale,
                                                           float *__restrict__ dshift, float mmt, float neg_lr, bool step,
                                                           float *__restrict__ in_diff, int ldi, int rows, const float *__restrict__ y, int ldy,
                                                           const float *__restrict__ xin, int ldxin, const float *__restrict__ mean, int Q,
                                                           unsigned long long *inbox, unsigned *err, float *__restrict__ max_parts, S16Out po,
                                                           unsigned long long *gmax, unsigned token) {
  constexpr int CG = kCoopCG, L = kCoopLanes;
  __shared__ float red[kPanelWaves * CG * 8];
  __shared__ float stat[3][4 * CG];  // S1, S2, the scale the forward pass used
  const int P = gridDim.x / Q, p = blockIdx.x % P, q = blockIdx.x / P;
  const int rp = (rows + Q - 1) / Q, r0 = q * rp, r1 = min(rows, r0 + rp);
  const int cg = threadIdx.x % CG, lane = threadIdx.x / CG;
  const int c = (p * CG + cg) * 4;
  float4 d[SLOTS], h[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    const int r = r0 + lane + k * L;
    d[k] = load4_rows(dy, ldd, r, r1, c);
    if (RECOMPUTE) h[k] = load4_rows(xin, ldxin, r, r1, c);
    else h[k] = load4_rows(xhat, ldx, r, r1, c);
  }
  // the per-column values go out in the rows' round trip (a load issued where its value is first used costs a trip to L2 of its own there).
  // The scale the forward pass used: workgroup q == 0 rewrites it (the folded SGD step) only after it has seen every partial of the panel,
  // and a workgroup publishes its partial only after this load has returned (the wait in front of the exchange).
  const float4 iv4 = *reinterpret_cast<const float4 *>(inv_std + c);
  const bool col_thread = threadIdx.x < 4 * CG;
  const int my_col = p * CG * 4 + (int)threadIdx.x;
  const float g_fwd = col_thread ? scale[my_col] : 0.f;
  const float dsh_old = col_thread && q == 0 ? dshift[my_col] : 0.f, dsc_old = col_thread && q == 0 ? dscale[my_col] : 0.f;
  const float sh_old = col_thread && q == 0 && step ? shift[my_col] : 0.f;
  if (RECOMPUTE) {
    const float4 m = *reinterpret_cast<const float4 *>(mean + c), is = iv4;
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      const int r = r0 + lane + k * L;
      if (r < r1) { h[k].x = (h[k].x - m.x) * is.x; h[k].y = (h[k].y - m.y) * is.y; h[k].z = (h[k].z - m.z) * is.z; h[k].w = (h[k].w - m.w) * is.w; }
    }
  }
  if (HAS_Y) {
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      const int r = r0 + lane + k * L;
      const float4 yy = load4_rows(y, ldy, r, r1, c);
      {
#pragma clang fp contract(off)
        d[k].x = d[k].x * yy.x * (1.0f - yy.x); d[k].y = d[k].y * yy.y * (1.0f - yy.y);
        d[k].z = d[k].z * yy.z * (1.0f - yy.z); d[k].w = d[k].w * yy.w * (1.0f - yy.w);
      }
    }
  }
  float acc[2][4] = {};
#pragma unroll
  for (int k = 0; k < SLOTS; k++) {
    acc[0][0] += d[k].x; acc[0][1] += d[k].y; acc[0][2] += d[k].z; acc[0][3] += d[k].w;
    acc[1][0] += h[k].x * d[k].x; acc[1][1] += h[k].y * d[k].y; acc[1][2] += h[k].z * d[k].z; acc[1][3] += h[k].w * d[k].w;
  }
  panel_reduce<float, 2, CG>(acc, red);
  if (threadIdx.x < 4 * CG) {
    const int pc = threadIdx.x, col = p * CG * 4 + pc;
    unsigned long long mine[1], got[8][1];
    const float p1 = panel_total<float, 2, CG>(red, 0, pc), p2 = panel_total<float, 2, CG>(red, 1, pc);
    mine[0] = ((unsigned long long)__float_as_uint(p2) << 32) | (unsigned long long)__float_as_uint(p1);   // one granule: {S2, S1}
    const float g = g_fwd;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (g_fwd has landed before the partial is published; see the top)
    coop_exchange<1>(inbox, P, Q, p, q, pc, mine, got, err);
    float s1 = 0.f, s2 = 0.f;
    for (int qw = 0; qw < Q; qw++) { s1 += __uint_as_float((unsigned)got[qw][0]); s2 += __uint_as_float((unsigned)(got[qw][0] >> 32)); }
    stat[0][pc] = s1;
    stat[1][pc] = s2;
    stat[2][pc] = g;
    if (q == 0) {
      const float dsh = s1 + mmt * dsh_old, dsc = s2 + mmt * dsc_old;
      dshift[col] = dsh;
      dscale[col] = dsc;
      if (step) {
        scale[col] = g + neg_lr * dsc;
        shift[col] = sh_old + neg_lr * dsh;
      }
    }
  }
  __syncthreads();
  const float invB = 1.0f / (float)rows;
  if (in_diff != nullptr) {
    const float iv[4] = {iv4.x, iv4.y, iv4.z, iv4.w};
    float gv[4], ca[4], cb[4];  // per column: in_diff = D*inv + xm * ca + cb
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float g = stat[2][cg * 4 + i], inv = iv[i];
      const float dvar = -0.5f * inv * inv * g * stat[1][cg * 4 + i];
      const float dmean = -inv * g * stat[0][cg * 4 + i];
      gv[i] = g;
      ca[i] = (2.0f * invB) * dvar;
      cb[i] = invB * dmean;
    }
    float omax = 0.f;
#pragma unroll
    for (int k = 0; k < SLOTS; k++) {
      const int r = r0 + lane + k * L;
      if (r >= r1) break;
      const float dv[4] = {d[k].x, d[k].y, d[k].z, d[k].w}, hv[4] = {h[k].x, h[k].y, h[k].z, h[k].w};
      float Dv[4], ov[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float D = dv[i] * gv[i];
        const float xm = hv[i] / iv[i];
        Dv[i] = D;
        ov[i] = D * iv[i] + xm * ca[i] + cb[i];
      }
      if (!RECOMPUTE) *reinterpret_cast<float4 *>(xhat + (long)r * ldx + c) = make_float4(Dv[0], Dv[1], Dv[2], Dv[3]);
      *reinterpret_cast<float4 *>(in_diff + (long)r * ldi + c) = make_float4(ov[0], ov[1], ov[2], ov[3]);
      omax = s16_absmax4(omax, make_float4(ov[0], ov[1], ov[2], ov[3]));
      d[k] = make_float4(ov[0], ov[1], ov[2], ov[3]);   // (kept for the planes below)
    }
    if (max_parts != nullptr || po.hi != nullptr) {   // this workgroup's largest |in_diff|
      omax = wave_max(omax);
      __syncthreads();            // (red[] is free: every thread is past the statistics)
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = omax;
      __syncthreads();
      float m = red[0];
      for (int w = 1; w < kPanelWaves; w++) m = fmaxf(m, red[w]);
      // ... for the conversion of in_diff, which takes its scale from these (split16.h) ...
      if (max_parts != nullptr && threadIdx.x == 0) max_parts[blockIdx.x] = m;
      // ... or the planes from this launch: the workgroups' maxima meet (coop_grid_max) and every workgroup scales its rows by the matrix
      // maximum, exactly as the conversion pass would have.
      if (po.hi != nullptr) {
        const float gm = coop_grid_max(m, gmax, token, err, red);
        const unsigned mbits = __float_as_uint(gm);
        if (blockIdx.x == 0 && threadIdx.x == 0) *const_cast<unsigned *>(po.slot) = mbits;   // for the products (launched behind this kernel)
        const float ps = ldexpf(1.f, s16_exponent(mbits));
#pragma unroll
        for (int k = 0; k < SLOTS; k++) {
          const int r = r0 + lane + k * L;
          if (r >= r1) break;
          half4 hi, lo;
          s16_split4(d[k], ps, &hi, &lo);
          *reinterpret_cast<half4 *>(po.hi + (long)r * po.ld + c) = hi;
          *reinterpret_cast<half4 *>(po.lo + (long)r * po.ld + c) = lo;
        }
      }
    }
  }
}
constexpr int kCoopGmaxWords = 1024;   // workgroups whose maxima may meet in one launch
// Host threads of this process that launch cooperative kernels.  With more than one, the launches that wait for ALL their workgroups are not
// used: two of them (or one beside another thread's persistent recurrence) half resident on one chip would wait for each other's
// unplaced workgroups -- the same reason as device_shared() (scratch.h), between threads instead of processes.
// (the count lives in runtime.cpp -- register_grid_wide_thread -- because the persistent recurrences' launchers register in it too: a thread
//  that only ever runs recurrences never comes through coop_state())
struct CoopState {
  unsigned long long *inbox = nullptr, *gmax = nullptr;
  unsigned *err = nullptr, token = 0;
  bool tried = false;
};
// (per host thread: launches of different threads run side by side on their own streams and must not meet in one exchange area)
CoopState &coop_state() {
  static thread_local CoopState st;
  if (!st.tried) {
    st.tried = true;
    register_grid_wide_thread();
    const size_t words = (size_t)256 * 8 * 8 * kCoopCols * 3;  // up to 256 panels x Q <= 8 readers x 8 writers
    unsigned long long *p = nullptr;
    if (hipMalloc(&p, words * 8) == hipSuccess && hipMemset(p, 0xFF, words * 8) == hipSuccess) {
      st.err = new_async_error_word("cooperative kernel (BatchNormalization / input planes): a workgroup timed out waiting for the other workgroups of its "
                                    "launch (results of that call are invalid; if several processes share this GPU set ASLP_DEVICE_SHARED=1)");
      if (st.err) st.inbox = p;
      unsigned long long *gm = nullptr;
      if (st.inbox && hipMalloc(&gm, kCoopGmaxWords * 8) == hipSuccess && hipMemset(gm, 0, kCoopGmaxWords * 8) == hipSuccess) st.gmax = gm;   // (token 0 is never used)
    }
  }
  return st;
}
// may a launch wait for every one of its workgroups?  (a device of this process' own, one launching host thread, the main stream)
inline bool coop_grid_wide_ok() { return !device_shared() && !on_side_stream() && grid_wide_threads() <= 1; }
inline int coop_cu_count() {
  static int num_cu = [] { hipDeviceProp_t pr; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 0; }();
  return num_cu;
}
// rows x cols served by the cooperative kernels?  returns Q (row parts per panel) and the slots per thread, or Q = 0
struct CoopShape { int q, slots; };
inline CoopShape bn_coop_shape(int rows, int cols) {
  static const int forced = [] { const char *e = getenv("ASLP_BN_COOP"); return e ? atoi(e) : -1; }();
  CoopShape none = {0, 0};
  if (forced == 0 || cols % kCoopCols != 0) return none;
  const int num_cu = coop_cu_count();
  const int P = cols / kCoopCols;
  if (P > 256 || num_cu <= 0) return none;
  int Q = num_cu / P;              // every workgroup resident at once, one per CU (they wait for each other)
  Q = Q > 4 ? 4 : Q;
  if (Q < 2) return none;
  const int rp = (rows + Q - 1) / Q;
  if (rp < 64) return none;        // small batches: the single-workgroup panels are already launch-latency bound
  for (int slots : {4, 8, 16})
    if (rp <= slots * kCoopLanes) return CoopShape{Q, slots};
  return none;
}

// 0: the three-launch path; else column groups per workgroup and row slots per thread (4, 8 or 16).  Measured on the cfg2 step
// (1024 x 2048): 4 groups x 256 threads 729 k frames/s, 2 groups (32-byte row segments) 698 k, 1024-thread workgroups with 4 / 8
// groups 680 k / 713 k, the three-launch path 717-721 k.
struct PanelShape { int cg, slots; };
inline PanelShape bn_panel_shape(int rows, int cols) {
  static const int forced = [] { const char *e = getenv("ASLP_BN_PANEL"); return e ? atoi(e) : -1; }();
  PanelShape none = {0, 0};
  if (forced == 0) return none;
  const int cg = 4;
  if (cols % (4 * cg) != 0) return none;
  const int lanes = kPanelThreads / cg;
  for (int slots : {4, 8, 16})
    if (rows <= slots * lanes) return PanelShape{cg, slots};
  return none;
}

// ---- Xent ------------------------------------------------------------------------------
constexpr int kXentPerThread = 32;  // cols <= 8192 cached in registers

// stats[0..4] += {frames, correct, -xent, -entropy, likelihood} summed over the rows in a fixed order (thread t: rows t, t + 256, ...; the
// 64 lanes of a wave; the four waves), by one workgroup of 256 threads.  coherent: the rows were written by other workgroups of the SAME launch
// (device-scope loads; unused by the shipped callers).  Four rows' loads are in flight at a time.
template <bool RAW = false>   // RAW: the five sums as they are into stats[0..4] (xent_apply_kernel adds them to the accumulators later)
__device__ __forceinline__ void xent_sum_rows(const double *rowstats, int rows, double *stats, bool coherent) {
  __shared__ double sh_sum[4][5];
  double a[5] = {0, 0, 0, 0, 0};
  for (int r0 = threadIdx.x; r0 < rows; r0 += 4 * 256) {
    double v[4][5];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int r = r0 + u * 256;
        const double *p = rowstats + (long)(r < rows ? r : r0) * 5 + k;
        const double x = coherent ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
        v[u][k] = r < rows ? x : 0.0;
      }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (r0 + u * 256 < rows) {   // (adding the 0.0 of a missing row would turn a -0.0 sum into +0.0)
#pragma unroll
        for (int k = 0; k < 5; k++) a[k] += v[u][k];
      }
  }
#pragma unroll
  for (int k = 0; k < 5; k++) a[k] = wave_sum_d(a[k]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 5; k++) sh_sum[threadIdx.x >> 6][k] = a[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    double s[5];
    for (int k = 0; k < 5; k++) s[k] = sh_sum[0][k] + sh_sum[1][k] + sh_sum[2][k] + sh_sum[3][k];
    if (RAW) {
      for (int k = 0; k < 5; k++) stats[k] = s[k];
      return;
    }
    stats[0] += s[0];
    stats[1] += s[1];
    stats[2] += -s[2];
    stats[3] += -s[3];
    stats[4] += s[4];
  }
}

// One block (256 threads) per row.  rowstats[r][0..4] = {w, correct*w, w*sum t log(y+1e-20),
// w*sum t log(t+1e-20), w*sum t y} as double; summed in fixed order by xent_finalize.
// SOFTMAX: `y` holds the activations in front of the network's final Softmax and the row softmax is formed here,
// with exactly the arithmetic of softmax_rows_kernel<256> (same lane -> column mapping, same reduction order), so
// folding the Softmax component into the loss changes no bit; `y_out` (nullable) receives the posteriors.
template <bool DENSE, bool SOFTMAX, int PER, bool VEC = false>
__global__ void __launch_bounds__(256) xent_rows_kernel(const float *y, int ldy, const float *t, int ldt, const int32_t *labels,
                                                        const float *fw, float *diff, int ldd, int rows, int cols, double *rowstats,
                                                        float *y_out, int ldyo, S16Out po) {
  // VEC: slot k of a thread is column 4 (tid + 256 (k / 4)) + k % 4, every array touched 16 bytes at a time (softmax_rows_kernel's second
  // map, chosen by the same rule: cols % 4 == 0 and 16-byte aligned rows); else column tid + 256 k
  static_assert(!VEC || PER % 4 == 0, "whole groups of four slots");
  __shared__ float shf[4][4];
  __shared__ int shi[4][2];
  __shared__ float shs[4];
  const float pscale = po.hi ? ldexpf(1.f, s16_exponent(*po.slot)) : 0.f;   // planes of diff for the product that reads it (split16.h)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  auto col_of = [tid](int k) { return VEC ? 4 * (tid + (k >> 2) * 256) + (k & 3) : tid + k * 256; };
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float *yr = y + (long)r * ldy;
    const float *tr = DENSE ? t + (long)r * ldt : nullptr;
    const int label = DENSE ? -1 : labels[r];
    const float fw_r = fw[r];   // (here, in the row's round trip: read where it is used it is a trip of its own behind the last reduction)
    float yv[PER], tv[PER];
    float tsum = 0.0f, ybest = -1e21f, tbest = -1e21f;
    int yi = -1, ti = -1;
    // the row (activations or posteriors), as many loads in flight as the thread has slots
    if (VEC) {
#pragma unroll
      for (int k4 = 0; k4 < PER / 4; k4++) {
        const int c = 4 * (tid + k4 * 256);
        if (c < cols) {
          const float4 v4 = *reinterpret_cast<const float4 *>(yr + c);
          yv[4 * k4] = v4.x; yv[4 * k4 + 1] = v4.y; yv[4 * k4 + 2] = v4.z; yv[4 * k4 + 3] = v4.w;
          if (DENSE) {
            const float4 t4 = *reinterpret_cast<const float4 *>(tr + c);
            tv[4 * k4] = t4.x; tv[4 * k4 + 1] = t4.y; tv[4 * k4 + 2] = t4.z; tv[4 * k4 + 3] = t4.w;
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < PER; k++) {
        const int c = tid + k * 256;
        if (c < cols) {
          yv[k] = yr[c];
          if (DENSE) tv[k] = tr[c];
        }
      }
    }
    if (SOFTMAX) {
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < PER; k++)
        if (col_of(k) < cols) m = fmaxf(m, yv[k]);
      m = wave_max(m);
      if (lane == 0) shs[w] = m;
      __syncthreads();
      m = fmaxf(fmaxf(fmaxf(shs[0], shs[1]), shs[2]), shs[3]);
      __syncthreads();
      float sum = 0.0f;
#pragma unroll
      for (int k = 0; k < PER; k++) {
        if (col_of(k) < cols) {
          yv[k] = expf(yv[k] - m);
          sum += yv[k];
        }
      }
      sum = wave_sum(sum);
      if (lane == 0) shs[w] = sum;
      __syncthreads();
      sum = shs[0];
      sum += shs[1]; sum += shs[2]; sum += shs[3];
      __syncthreads();
      const float inv = 1.0f / sum;
#pragma unroll
      for (int k = 0; k < PER; k++)
        if (col_of(k) < cols) yv[k] *= inv;
      if (y_out) {
        if (VEC) {
#pragma unroll
          for (int k4 = 0; k4 < PER / 4; k4++) {
            const int c = 4 * (tid + k4 * 256);
            if (c < cols) *reinterpret_cast<float4 *>(y_out + (long)r * ldyo + c) = make_float4(yv[4 * k4], yv[4 * k4 + 1], yv[4 * k4 + 2], yv[4 * k4 + 3]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < PER; k++) {
            const int c = tid + k * 256;
            if (c < cols) y_out[(long)r * ldyo + c] = yv[k];
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int c = col_of(k);
      if (c < cols) {
        if (!DENSE) tv[k] = (c == label ? 1.0f : 0.0f);
        tsum += tv[k];
        if (ybest < yv[k]) { ybest = yv[k]; yi = c; }
        if (tbest < tv[k]) { tbest = tv[k]; ti = c; }
      }
    }
    // diff (and the three sums' terms) from the posteriors in the registers.  wr = frame weight x sum(t): frames with sum(t) == 0 are switched
    // off (nnet-loss.cc:80-85).  With a label target sum(t) is 1 or 0 by the label alone -- known here, so the diff leaves BEFORE the block
    // reductions below (which then only feed the row's statistics) instead of behind them.
    double xe = 0.0, en = 0.0, lk = 0.0;
    auto emit = [&](const float wr) {
#pragma unroll
    for (int k = 0; k < PER; k++) {
      if (col_of(k) < cols) {
        const float yy = yv[k], tt = tv[k];
        yv[k] = (yy - tt) * wr;   // the diff, kept for the stores below
        if (tt != 0.0f) {  // t == 0 terms are exactly 0 (t*log(...) with finite log)
          xe += (double)(logf(yy + 1e-20f) * tt * wr);
          en += (double)(logf(tt + 1e-20f) * tt * wr);
          lk += (double)(yy * tt * wr);
        }
      }
    }
    if (VEC) {
#pragma unroll
      for (int k4 = 0; k4 < PER / 4; k4++) {
        const int c = 4 * (tid + k4 * 256);
        if (c < cols) {
          const float4 d4 = make_float4(yv[4 * k4], yv[4 * k4 + 1], yv[4 * k4 + 2], yv[4 * k4 + 3]);
          *reinterpret_cast<float4 *>(diff + (long)r * ldd + c) = d4;
          if (po.hi) {
            half4 hi, lo;
            s16_split4(d4, pscale, &hi, &lo);
            *reinterpret_cast<half4 *>(po.hi + (long)r * po.ld + c) = hi;
            *reinterpret_cast<half4 *>(po.lo + (long)r * po.ld + c) = lo;
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < PER; k++) {
        const int c = tid + k * 256;
        if (c < cols) {
          diff[(long)r * ldd + c] = yv[k];
          if (po.hi) {
            h16 h, l;
            s16_split(yv[k], pscale, &h, &l);
            po.hi[(long)r * po.ld + c] = h;
            po.lo[(long)r * po.ld + c] = l;
          }
        }
      }
    }
    };
    const float tsum_by_label = (label >= 0 && label < cols) ? 1.0f : 0.0f;
    if (!DENSE) emit(fw_r * tsum_by_label);
    // block reductions: argmax(y); with dense targets also tsum and argmax(t) -- a label target's are known: sum 1 or 0, arg-max the label
    // (column 0 for a row of zeros: the lowest index among equals)
    if (DENSE) tsum = wave_sum(tsum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(ybest, o, 64); int oi = __shfl_xor(yi, o, 64);
      if (ov > ybest || (ov == ybest && oi >= 0 && (yi < 0 || oi < yi))) { ybest = ov; yi = oi; }
      if (DENSE) {
        ov = __shfl_xor(tbest, o, 64); oi = __shfl_xor(ti, o, 64);
        if (ov > tbest || (ov == tbest && oi >= 0 && (ti < 0 || oi < ti))) { tbest = ov; ti = oi; }
      }
    }
    if (lane == 0) { shf[w][0] = tsum; shf[w][1] = ybest; shf[w][2] = tbest; shi[w][0] = yi; shi[w][1] = ti; }
    __syncthreads();
    tsum = shf[0][0] + shf[1][0] + shf[2][0] + shf[3][0];
    ybest = shf[0][1]; yi = shi[0][0]; tbest = shf[0][2]; ti = shi[0][1];
#pragma unroll
    for (int j = 1; j < 4; j++) {
      if (shf[j][1] > ybest || (shf[j][1] == ybest && shi[j][0] >= 0 && (yi < 0 || shi[j][0] < yi))) { ybest = shf[j][1]; yi = shi[j][0]; }
      if (DENSE && (shf[j][2] > tbest || (shf[j][2] == tbest && shi[j][1] >= 0 && (ti < 0 || shi[j][1] < ti)))) { tbest = shf[j][2]; ti = shi[j][1]; }
    }
    if (!DENSE) { tsum = tsum_by_label; ti = (label >= 0 && label < cols) ? label : 0; }
    __syncthreads();
    const float wr = fw_r * tsum;
    if (DENSE) emit(wr);
    double *rs = rowstats + (long)r * 5;
    if (DENSE) {
      xe = wave_sum_d(xe); en = wave_sum_d(en); lk = wave_sum_d(lk);
      __shared__ double shd[4][3];
      if (lane == 0) { shd[w][0] = xe; shd[w][1] = en; shd[w][2] = lk; }
      __syncthreads();
      if (tid == 0) {
        rs[0] = (double)wr;
        rs[1] = (double)wr * (yi == ti ? 1.0 : 0.0);
        rs[2] = shd[0][0] + shd[1][0] + shd[2][0] + shd[3][0];
        rs[3] = shd[0][1] + shd[1][1] + shd[2][1] + shd[3][1];
        rs[4] = shd[0][2] + shd[1][2] + shd[2][2] + shd[3][2];
      }
    } else {
      // a label target has ONE non-zero term, in the thread that holds the label's column: that thread's three sums ARE the row's (every
      // other thread's are +0.0, and x + 0.0 = x), so they go out as they are -- no reduction over the workgroup
      const bool has_label = label >= 0 && label < cols;
      const bool mine = has_label && (VEC ? ((label >> 2) & 255) == tid : (label & 255) == tid);
      if (mine) { rs[2] = xe; rs[3] = en; rs[4] = lk; }
      if (tid == 0) {
        rs[0] = (double)wr;
        rs[1] = (double)wr * (yi == ti ? 1.0 : 0.0);
        if (!has_label) { rs[2] = 0.0; rs[3] = 0.0; rs[4] = 0.0; }
      }
    }
    __syncthreads();
  }
}

// The same for rows of any width (more than 256 * kXentPerThread classes): nothing is cached in registers, the row is streamed twice
// (targets: sum and both arg-maxes; then diff and the three sums).  Every thread visits the columns tid, tid + 256, ... in the same
// order and the reductions are those of the kernel above, so the result does not depend on which of the two kernels served a row.
template <bool DENSE>
__global__ void __launch_bounds__(256) xent_rows_wide_kernel(const float *y, int ldy, const float *t, int ldt, const int32_t *labels, const float *fw,
                                                             float *diff, int ldd, int rows, int cols, double *rowstats) {
  __shared__ float shf[4][4];
  __shared__ int shi[4][2];
  __shared__ double shd[4][3];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float *yr = y + (long)r * ldy;
    const float *tr = DENSE ? t + (long)r * ldt : nullptr;
    const int label = DENSE ? -1 : labels[r];
    float tsum = 0.0f, ybest = -1e21f, tbest = -1e21f;
    int yi = -1, ti = -1;
    for (int c = tid; c < cols; c += 256) {
      const float yy = yr[c], tt = DENSE ? tr[c] : (c == label ? 1.0f : 0.0f);
      tsum += tt;
      if (ybest < yy) { ybest = yy; yi = c; }
      if (tbest < tt) { tbest = tt; ti = c; }
    }
    tsum = wave_sum(tsum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(ybest, o, 64); int oi = __shfl_xor(yi, o, 64);
      if (ov > ybest || (ov == ybest && oi >= 0 && (yi < 0 || oi < yi))) { ybest = ov; yi = oi; }
      ov = __shfl_xor(tbest, o, 64); oi = __shfl_xor(ti, o, 64);
      if (ov > tbest || (ov == tbest && oi >= 0 && (ti < 0 || oi < ti))) { tbest = ov; ti = oi; }
    }
    if (lane == 0) { shf[w][0] = tsum; shf[w][1] = ybest; shf[w][2] = tbest; shi[w][0] = yi; shi[w][1] = ti; }
    __syncthreads();
    tsum = shf[0][0] + shf[1][0] + shf[2][0] + shf[3][0];
    ybest = shf[0][1]; yi = shi[0][0]; tbest = shf[0][2]; ti = shi[0][1];
#pragma unroll
    for (int j = 1; j < 4; j++) {
      if (shf[j][1] > ybest || (shf[j][1] == ybest && shi[j][0] >= 0 && (yi < 0 || shi[j][0] < yi))) { ybest = shf[j][1]; yi = shi[j][0]; }
      if (shf[j][2] > tbest || (shf[j][2] == tbest && shi[j][1] >= 0 && (ti < 0 || shi[j][1] < ti))) { tbest = shf[j][2]; ti = shi[j][1]; }
    }
    __syncthreads();
    const float wr = fw[r] * tsum;
    double xe = 0.0, en = 0.0, lk = 0.0;
    for (int c = tid; c < cols; c += 256) {
      const float yy = yr[c], tt = DENSE ? tr[c] : (c == label ? 1.0f : 0.0f);
      diff[(long)r * ldd + c] = (yy - tt) * wr;
      if (tt != 0.0f) {
        xe += (double)(logf(yy + 1e-20f) * tt * wr);
        en += (double)(logf(tt + 1e-20f) * tt * wr);
        lk += (double)(yy * tt * wr);
      }
    }
    xe = wave_sum_d(xe); en = wave_sum_d(en); lk = wave_sum_d(lk);
    if (lane == 0) { shd[w][0] = xe; shd[w][1] = en; shd[w][2] = lk; }
    __syncthreads();
    if (tid == 0) {
      double *rs = rowstats + (long)r * 5;
      rs[0] = (double)wr;
      rs[1] = (double)wr * (yi == ti ? 1.0 : 0.0);
      rs[2] = shd[0][0] + shd[1][0] + shd[2][0] + shd[3][0];
      rs[3] = shd[0][1] + shd[1][1] + shd[2][1] + shd[3][1];
      rs[4] = shd[0][2] + shd[1][2] + shd[2][2] + shd[3][2];
    }
    __syncthreads();
  }
}

// stats[0..4] += {frames, correct, xent, entropy, likelihood}; fixed-order sum over rows
__global__ void __launch_bounds__(256) xent_finalize_kernel(const double *rowstats, int rows, double *stats) { xent_sum_rows(rowstats, rows, stats, false); }
// The same for the per-row statistics of several batches at once (aslp_xent_sum_rowstats): workgroup b sums batch b exactly as above, then
// one thread adds the batches' sums to the accumulators in batch order -- the bits of one xent_finalize_kernel launch per batch.
__global__ void __launch_bounds__(256) xent_batch_sums_kernel(const double *rowstats, int rows, double *sums) {
  xent_sum_rows<true>(rowstats + (size_t)blockIdx.x * rows * 5, rows, sums + (size_t)blockIdx.x * 5, false);
}
__global__ void xent_apply_kernel(const double *sums, int batches, double *stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a[5] = {stats[0], stats[1], stats[2], stats[3], stats[4]};
  for (int b = 0; b < batches; b++) {
    const double *s = sums + (size_t)b * 5;
    a[0] += s[0]; a[1] += s[1]; a[2] += -s[2]; a[3] += -s[3]; a[4] += s[4];
  }
  for (int k = 0; k < 5; k++) stats[k] = a[k];
}

}  // namespace

static bool g_coop_convert_on = true;   // aslp_coop_convert (tests: the two-launch conversion as the reference)
// split16.h: (optional copy and) planes of n matrices in one launch; false = not served, nothing was launched
bool coop_convert_launch(const CoopConvJob *jobs, int n, const SeqFillJob *fill) {
  static const bool off = [] { const char *e = getenv("ASLP_COPY_PLANES"); return e != nullptr && e[0] == '0'; }();   // A/B switch
  // Not beside the main stream: a launch there may share the chip with a persistent recurrence whose workgroups need every CU (each
  // would wait for workgroups the other keeps from being placed), and two launches of this thread would meet in one exchange area.
  if (off || !g_coop_convert_on || n <= 0 || n > kS16MaxJobs) return false;
  long max_units = 0;
  CoopConvJobs js;
  for (int i = 0; i < n; i++) {
    const CoopConvJob &c = jobs[i];
    if (!c.src || !c.pl.hi || !c.pl.lo || !c.pl.slot || c.pl.rows <= 0 || c.pl.cols <= 0 || (c.pl.cols & 3) || (c.ld_src & 3) || !aligned16(c.src) ||
        (c.dst && ((c.ld_dst & 3) || !aligned16(c.dst))) || c.pl.ld < c.pl.cols || (c.pl.ld & 3) || (c.parts && c.nparts <= 0))
      return false;
    max_units = std::max(max_units, (long)c.pl.rows * (c.pl.cols >> 2));
    js.j[i] = c;
  }
  CoopState &st = coop_state();
  if (!st.gmax || !coop_grid_wide_ok()) return false;
  // every workgroup of the launch resident at once (those of a matrix wait for each other): at most two per CU (four fit by registers and
  // wave slots; beside a layer product's workgroup, one -- the second then waits for that workgroup to end, which it does)
  const int cus = std::min(2 * coop_cu_count(), kCoopGmaxWords);
  int U = 0, gx = 0;
  for (int u : {4, 8, 16}) {
    const long g = (max_units + (long)u * kPanelThreads - 1) / ((long)u * kPanelThreads);
    if (U == 0 && g * n <= cus) { U = u; gx = (int)g; }
  }
  if (U == 0) return false;
  if (++st.token == 0) st.token = 1;
  SeqFillJob fj = {nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, 0, 0};
  if (fill) fj = *fill;
#define ASLP_COPY_PLANES(UU) hipLaunchKernelGGL((copy_planes_coop<UU>), dim3(gx, n), dim3(kPanelThreads), 0, cur_stream(), js, st.gmax, st.token, st.err, fj)
  if (U == 4) ASLP_COPY_PLANES(4); else if (U == 8) ASLP_COPY_PLANES(8); else ASLP_COPY_PLANES(16);
#undef ASLP_COPY_PLANES
  check_launch("coop_convert");
  return true;
}
}  // namespace aslp

using namespace aslp;

extern "C" {

void aslp_bn_forward_act(const float *in, MatrixDim d, float *out, int out_stride, float *xhat, int xhat_stride, const float *scale,
                         const float *shift, float *mean, float *inv_std, double *acc_means, double *acc_vars, float var_floor, float *act_out,
                         int act_stride) {
  if (d.rows <= 0 || d.cols <= 0) return;
  if (!out && !act_out) { set_error("aslp_bn_forward: no output"); return; }
  const float invB = 1.0f / (float)d.rows;
  const bool in_vec = aligned16(in) && d.stride % 4 == 0;
  bool vec = d.cols % 4 == 0 && d.stride % 4 == 0 && (!out || (out_stride % 4 == 0 && aligned16(out))) &&
             (!xhat || (xhat_stride % 4 == 0 && aligned16(xhat))) && (!act_out || (act_stride % 4 == 0 && aligned16(act_out))) &&
             aligned16(in) && aligned16(mean) && aligned16(inv_std) && aligned16(scale) && aligned16(shift);
  const PanelShape ps = vec ? bn_panel_shape(d.rows, d.cols) : PanelShape{0, 0};
  const CoopShape cs = (vec && ps.cg) ? bn_coop_shape(d.rows, d.cols) : CoopShape{0, 0};
  if (cs.q && coop_state().inbox) {
    CoopState &st = coop_state();
    const dim3 grid((d.cols / kCoopCols) * cs.q), block(kPanelThreads);
#define ASLP_BN_FWD_COOP(SLOTS)                                                                                                             \
    case SLOTS:                                                                                                                             \
      hipLaunchKernelGGL((bn_forward_coop<SLOTS>), grid, block, 0, cur_stream(), in, d.stride, out, out_stride, xhat, xhat_stride, scale, shift, \
                         mean, inv_std, acc_means, acc_vars, invB, var_floor, d.rows, act_out, act_stride, cs.q, st.inbox, st.err);           \
      break;
    switch (cs.slots) { ASLP_BN_FWD_COOP(4) ASLP_BN_FWD_COOP(8) ASLP_BN_FWD_COOP(16) }
#undef ASLP_BN_FWD_COOP
    check_launch("bn_forward_coop");
    return;
  }
  if (ps.cg) {
    const dim3 grid(d.cols / (4 * ps.cg)), block(kPanelThreads);
#define ASLP_BN_FWD_PANEL(CG, SLOTS)                                                                                                       \
    case CG * 100 + SLOTS:                                                                                                                \
      hipLaunchKernelGGL((bn_forward_panel<CG, SLOTS>), grid, block, 0, cur_stream(), in, d.stride, out, out_stride, xhat, xhat_stride, scale, \
                         shift, mean, inv_std, acc_means, acc_vars, invB, var_floor, d.rows, act_out, act_stride);                          \
      break;
    switch (ps.cg * 100 + ps.slots) {
      ASLP_BN_FWD_PANEL(4, 4) ASLP_BN_FWD_PANEL(4, 8) ASLP_BN_FWD_PANEL(4, 16)
    }
#undef ASLP_BN_FWD_PANEL
    check_launch("bn_forward_panel");
    return;
  }
  colreduce<3, double>("bn_forward.stats", d.rows, d.cols, BnSum1F{in, d.stride}, BnSum1G{invB, var_floor, mean, inv_std, acc_means, acc_vars},
                       in_vec);
  long n = (long)d.rows * (vec ? d.cols / 4 : d.cols);
  if (vec) hipLaunchKernelGGL((bn_normalize_kernel<true>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), in, d.stride, out, out_stride, xhat, xhat_stride, mean, inv_std, scale, shift, d.rows, d.cols, act_out, act_stride);
  else hipLaunchKernelGGL((bn_normalize_kernel<false>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), in, d.stride, out, out_stride, xhat, xhat_stride, mean, inv_std, scale, shift, d.rows, d.cols, act_out, act_stride);
  check_launch("bn_forward");
}
int aslp_bn_forward_stats(const float *in, MatrixDim d, float *out, int out_stride, const float *scale, const float *shift, float *mean,
                          float *inv_std, double *acc_means, double *acc_vars, float var_floor, float *act_out, int act_stride,
                          const double *colstats, int groups, int stats_ld) {
  return aslp_bn_forward_stats_p(in, d, out, out_stride, scale, shift, mean, inv_std, acc_means, acc_vars, var_floor, act_out, act_stride, colstats,
                                 groups, stats_ld, nullptr);
}
int aslp_bn_forward_stats_p(const float *in, MatrixDim d, float *out, int out_stride, const float *scale, const float *shift, float *mean,
                            float *inv_std, double *acc_means, double *acc_vars, float var_floor, float *act_out, int act_stride,
                            const double *colstats, int groups, int stats_ld, const aslp_planes_out *act_planes) {
  if (d.rows <= 0 || d.cols <= 0 || !colstats) return 0;
  if (!out && !act_out) { set_error("aslp_bn_forward_stats: no output"); return 0; }
  if (groups != (d.rows + 31) / 32 || stats_ld < d.cols) { set_error("aslp_bn_forward_stats: statistics layout does not match the matrix"); return 0; }
  static const bool off = [] { const char *e = getenv("ASLP_BN_FROM_STATS"); return e && e[0] == '0'; }();   // A/B switch
  const bool vec = d.cols % 4 == 0 && d.stride % 4 == 0 && (!out || (out_stride % 4 == 0 && aligned16(out))) &&
                   (!act_out || (act_stride % 4 == 0 && aligned16(act_out))) && aligned16(in) && aligned16(mean) && aligned16(inv_std) &&
                   aligned16(scale) && aligned16(shift);
  // the shapes the backward pass serves without a stored normalised copy (it recomputes xhat from `in` and the batch mean)
  if (off || !vec || d.cols % kCoopCols != 0 || !bn_panel_shape(d.rows, d.cols).cg) return 0;
  const int P = d.cols / kCoopCols;
  int Q = 1;
  while (Q < 8 && P * Q * 2 <= 1024 && (d.rows + 2 * Q - 1) / (2 * Q) >= 64) Q *= 2;   // ~4 workgroups per CU at most, >= 64 rows each
  while ((d.rows + Q - 1) / Q > kStatSlots * kCoopLanes) Q *= 2;                       // at most kStatSlots rows per thread
  S16Out po = {nullptr, nullptr, 0, nullptr, nullptr};
  if (act_planes && act_planes->hi && act_out && act_planes->slot && act_planes->ld >= d.cols)
    po = S16Out{static_cast<h16 *>(act_planes->hi), static_cast<h16 *>(act_planes->lo), act_planes->ld, act_planes->slot, nullptr};
  hipLaunchKernelGGL(bn_forward_stats_kernel, dim3(P * Q), dim3(kPanelThreads), 0, cur_stream(), in, d.stride, out, out_stride, scale, shift, mean,
                     inv_std, acc_means, acc_vars, 1.0f / (float)d.rows, var_floor, d.rows, act_out, act_stride, Q, colstats, groups, stats_ld, po);
  check_launch("bn_forward_stats");
  return 1;
}

void aslp_bn_forward(const float *in, MatrixDim d, float *out, int out_stride, float *xhat, int xhat_stride, const float *scale,
                     const float *shift, float *mean, float *inv_std, double *acc_means, double *acc_vars, float var_floor) {
  aslp_bn_forward_act(in, d, out, out_stride, xhat, xhat_stride, scale, shift, mean, inv_std, acc_means, acc_vars, var_floor, nullptr, 0);
}

void aslp_bn_apply(const float *in, MatrixDim d, float *out, int out_stride, const float *mean, const float *inv_std, const float *scale, const float *shift) {
  if (d.rows <= 0 || d.cols <= 0) return;
  long n = (long)d.rows * d.cols;
  hipLaunchKernelGGL((bn_normalize_kernel<false>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), in, d.stride, out, out_stride, (float *)nullptr, 0, mean, inv_std, scale, shift, d.rows, d.cols, (float *)nullptr, 0);
  check_launch("bn_apply");
}

static void bn_backward_impl(MatrixDim d, const float *out_diff, int od_stride, float *xhat, int xhat_stride, float *scale, float *shift,
                             const float *inv_std, float *dscale, float *dshift, float momentum, float *in_diff, int id_stride,
                             const float *act_y, int act_stride, bool step, float learn_rate, const float *in, const float *mean,
                             aslp_planes_out *diff_out = nullptr) {
  if (diff_out) { diff_out->nparts = 0; diff_out->planes_written = 0; }
  if (d.rows <= 0 || d.cols <= 0) return;
  const bool y_ok = !act_y || (aligned16(act_y) && act_stride % 4 == 0);
  const bool recompute = xhat == nullptr;
  if (recompute && (!in || !mean)) { set_error("aslp_bn_backward: without a normalised copy (xhat == NULL) the layer input and the batch mean are needed"); return; }
  const bool panel_ok = od_stride % 4 == 0 && (!in_diff || (id_stride % 4 == 0 && aligned16(in_diff))) && aligned16(out_diff) && aligned16(inv_std) && y_ok &&
                        (recompute ? (d.stride % 4 == 0 && aligned16(in) && aligned16(mean)) : (xhat_stride % 4 == 0 && aligned16(xhat)));
  const PanelShape ps = panel_ok ? bn_panel_shape(d.rows, d.cols) : PanelShape{0, 0};
  if (!ps.cg && recompute) { set_error("aslp_bn_backward: xhat == NULL is only served by the single-launch panel path (check aslp_bn_panel_supported)"); return; }
  const CoopShape cs = ps.cg ? bn_coop_shape(d.rows, d.cols) : CoopShape{0, 0};
  if (cs.q && coop_state().inbox) {
    CoopState &st = coop_state();
    const dim3 grid((d.cols / kCoopCols) * cs.q), block(kPanelThreads);
    // in_diff's planes from this launch when all its workgroups are resident at once (they wait for each other's maxima: one per CU is
    // what the shape was chosen for), else the maxima for the conversion pass
    S16Out po = {nullptr, nullptr, 0, nullptr, nullptr};
    unsigned token = 0;
    static const bool planes_off = [] { const char *e = getenv("ASLP_BN_DIFF_PLANES"); return e != nullptr && e[0] == '0'; }();   // A/B switch
    if (diff_out && diff_out->hi && diff_out->lo && diff_out->slot && in_diff && st.gmax && !planes_off && coop_grid_wide_ok() && (int)grid.x <= kCoopGmaxWords &&
        (int)grid.x <= coop_cu_count() && diff_out->ld >= d.cols && diff_out->ld % 4 == 0) {
      po = S16Out{static_cast<h16 *>(diff_out->hi), static_cast<h16 *>(diff_out->lo), diff_out->ld, diff_out->slot, nullptr};
      if (++st.token == 0) st.token = 1;
      token = st.token;
      diff_out->planes_written = 1;
    }
    float *max_parts = (!po.hi && diff_out && diff_out->parts && in_diff && (int)grid.x <= kS16MaxParts) ? diff_out->parts : nullptr;
    if (max_parts) diff_out->nparts = (int)grid.x;
#define ASLP_BN_BWD_CL(SLOTS, Y, RC)                                                                                                           \
    hipLaunchKernelGGL((bn_backward_coop<SLOTS, Y, RC>), grid, block, 0, cur_stream(), out_diff, od_stride, xhat, xhat_stride, scale, shift, inv_std, \
                       dscale, dshift, momentum, -learn_rate, step, in_diff, id_stride, d.rows, act_y, act_stride, in, d.stride, mean, cs.q, st.inbox, st.err, \
                       max_parts, po, st.gmax, token)
#define ASLP_BN_BWD_COOP(SLOTS)                                                                      \
    case SLOTS:                                                                                      \
      if (act_y) { if (recompute) ASLP_BN_BWD_CL(SLOTS, true, true); else ASLP_BN_BWD_CL(SLOTS, true, false); }     \
      else { if (recompute) ASLP_BN_BWD_CL(SLOTS, false, true); else ASLP_BN_BWD_CL(SLOTS, false, false); }         \
      break;
    switch (cs.slots) { ASLP_BN_BWD_COOP(4) ASLP_BN_BWD_COOP(8) ASLP_BN_BWD_COOP(16) }
#undef ASLP_BN_BWD_COOP
#undef ASLP_BN_BWD_CL
    check_launch("bn_backward_coop");
    return;
  }
  if (ps.cg) {
    const dim3 grid(d.cols / (4 * ps.cg)), block(kPanelThreads);
#define ASLP_BN_BWD_LAUNCH(CG, SLOTS, Y, RC)                                                                                                      \
    hipLaunchKernelGGL((bn_backward_panel<CG, SLOTS, Y, RC>), grid, block, 0, cur_stream(), out_diff, od_stride, xhat, xhat_stride, scale, shift,    \
                       inv_std, dscale, dshift, momentum, -learn_rate, step, in_diff, id_stride, d.rows, act_y, act_stride, in, d.stride, mean)
#define ASLP_BN_BWD_PANEL(CG, SLOTS)                                                             \
    case CG * 100 + SLOTS:                                                                      \
      if (act_y) { if (recompute) ASLP_BN_BWD_LAUNCH(CG, SLOTS, true, true); else ASLP_BN_BWD_LAUNCH(CG, SLOTS, true, false); }   \
      else { if (recompute) ASLP_BN_BWD_LAUNCH(CG, SLOTS, false, true); else ASLP_BN_BWD_LAUNCH(CG, SLOTS, false, false); }       \
      break;
    switch (ps.cg * 100 + ps.slots) {
      ASLP_BN_BWD_PANEL(4, 4) ASLP_BN_BWD_PANEL(4, 8) ASLP_BN_BWD_PANEL(4, 16)
    }
#undef ASLP_BN_BWD_PANEL
#undef ASLP_BN_BWD_LAUNCH
    check_launch("bn_backward_panel");
    return;
  }
  float *s12 = static_cast<float *>(scratch(kScratchReduce2, sizeof(float) * 3 * (size_t)d.cols));
  if (!s12) return;
  float *scale_used = s12 + 2 * (size_t)d.cols;
  colreduce<2, float>("bn_backward.stats", d.rows, d.cols, BnBwdF{out_diff, od_stride, xhat, xhat_stride, act_y, act_stride},
                      BnBwdG{momentum, dscale, dshift, s12, s12 + d.cols, scale, shift, scale_used, -learn_rate, step},
                      aligned16(out_diff) && od_stride % 4 == 0 && aligned16(xhat) && xhat_stride % 4 == 0 && y_ok);
  if (!in_diff) return;
  long n = (long)d.rows * d.cols;
  bool vec = d.cols % 4 == 0 && od_stride % 4 == 0 && xhat_stride % 4 == 0 && id_stride % 4 == 0 && aligned16(out_diff) &&
             aligned16(xhat) && aligned16(in_diff) && aligned16(inv_std) && aligned16(s12) && y_ok;
  if (vec) hipLaunchKernelGGL((bn_backward_kernel<true>), dim3(grid_for(n / 4)), dim3(kBlock), 0, cur_stream(), out_diff, od_stride, xhat, xhat_stride, scale_used, inv_std, s12, s12 + d.cols, in_diff, id_stride, d.rows, d.cols, act_y, act_stride);
  else hipLaunchKernelGGL((bn_backward_kernel<false>), dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), out_diff, od_stride, xhat, xhat_stride, scale_used, inv_std, s12, s12 + d.cols, in_diff, id_stride, d.rows, d.cols, act_y, act_stride);
  check_launch("bn_backward");
}
void aslp_bn_backward_act(const float *in, MatrixDim d, const float *out_diff, int od_stride, float *xhat, int xhat_stride, const float *scale,
                          const float *mean, const float *inv_std, float *dscale, float *dshift, float momentum, float *in_diff, int id_stride,
                          const float *act_y, int act_stride) {
  // with xhat: (x - mean) is recovered as xhat / inv_std; without (panel path only): x_hat is formed again from `in` and `mean`
  bn_backward_impl(d, out_diff, od_stride, xhat, xhat_stride, const_cast<float *>(scale), nullptr, inv_std, dscale, dshift, momentum, in_diff,
                   id_stride, act_y, act_stride, false, 0.0f, in, mean);
}
// backward + the component's own Update (scale -= lr*dscale, shift -= lr*dshift) in the statistics finalize
void aslp_bn_backward_step(MatrixDim d, const float *out_diff, int od_stride, float *xhat, int xhat_stride, float *scale, float *shift,
                           const float *inv_std, float *dscale, float *dshift, float momentum, float learn_rate, float *in_diff, int id_stride,
                           const float *act_y, int act_stride, const float *in, const float *mean) {
  bn_backward_impl(d, out_diff, od_stride, xhat, xhat_stride, scale, shift, inv_std, dscale, dshift, momentum, in_diff, id_stride, act_y,
                   act_stride, true, learn_rate, in, mean);
}
void aslp_bn_backward_step_p(MatrixDim d, const float *out_diff, int od_stride, float *xhat, int xhat_stride, float *scale, float *shift,
                             const float *inv_std, float *dscale, float *dshift, float momentum, float learn_rate, float *in_diff, int id_stride,
                             const float *act_y, int act_stride, const float *in, const float *mean, aslp_planes_out *diff_out) {
  bn_backward_impl(d, out_diff, od_stride, xhat, xhat_stride, scale, shift, inv_std, dscale, dshift, momentum, in_diff, id_stride, act_y,
                   act_stride, true, learn_rate, in, mean, diff_out);
}
// 1: a [rows x cols] batch is served by the single-launch panel kernels (given 16-byte aligned operands), which need no
// normalised copy of the input: pass xhat = NULL to the forward and backward entry points and save its 8.4 MB each way
int aslp_bn_panel_supported(int rows, int cols) { return bn_panel_shape(rows, cols).cg != 0 ? 1 : 0; }
void aslp_bn_backward(const float *in, MatrixDim d, const float *out_diff, int od_stride, float *xhat, int xhat_stride, const float *scale,
                      const float *mean, const float *inv_std, float *dscale, float *dshift, float momentum, float *in_diff, int id_stride) {
  aslp_bn_backward_act(in, d, out_diff, od_stride, xhat, xhat_stride, scale, mean, inv_std, dscale, dshift, momentum, in_diff, id_stride, nullptr,
                       0);
}

static bool xent_eval_impl(const float *net_out, MatrixDim d, const float *tgt, int tgt_stride, const int32_cuda *labels,
                           const float *frame_weights, float *diff, int diff_stride, double *stats_dev, bool softmax, float *y_out, int y_stride,
                           const aslp_planes_out *diff_planes = nullptr, double *rowstats_out = nullptr) {
  if (d.rows <= 0 || d.cols <= 0) return false;
  if (!tgt && !labels) { set_error("aslp_xent_eval: need dense targets or labels"); return false; }
  S16Out po = {nullptr, nullptr, 0, nullptr, nullptr};
  bool planes_written = false;
  // rowstats_out: the caller keeps the per-row statistics and adds them to its accumulators later (aslp_xent_sum_rowstats)
  double *rowstats = rowstats_out ? rowstats_out : static_cast<double *>(scratch(kScratchReduce, sizeof(double) * 5 * (size_t)d.rows));
  if (!rowstats) return false;
  int g = d.rows > kMaxGrid * 2 ? kMaxGrid * 2 : d.rows;
  if (d.cols > 256 * kXentPerThread) {   // wider than the register-cached kernels hold: the streaming kernel (the reference has no limit)
    if (softmax) { set_error("aslp_softmax_xent_eval: unsupported number of classes"); return false; }
    if (tgt) hipLaunchKernelGGL((xent_rows_wide_kernel<true>), dim3(g), dim3(256), 0, cur_stream(), net_out, d.stride, tgt, tgt_stride, labels, frame_weights,
                                diff, diff_stride, d.rows, d.cols, rowstats);
    else hipLaunchKernelGGL((xent_rows_wide_kernel<false>), dim3(g), dim3(256), 0, cur_stream(), net_out, d.stride, tgt, tgt_stride, labels, frame_weights,
                            diff, diff_stride, d.rows, d.cols, rowstats);
    if (!rowstats_out) hipLaunchKernelGGL(xent_finalize_kernel, dim3(1), dim3(256), 0, cur_stream(), rowstats, d.rows, stats_dev);
    check_launch("xent_eval");
    return false;
  }
  // elements cached per thread: the smallest of 4 / 8 / 16 / 32 that covers the row (the loops are fully unrolled: a row of
  // 3000 classes runs 16 slots per thread instead of 32 predicated ones)
  const int per = (d.cols + 255) / 256;
  if (diff_planes && diff_planes->hi && diff_planes->slot && !tgt && diff_planes->ld >= d.cols) {   // the row kernel, label targets
    po = S16Out{static_cast<h16 *>(diff_planes->hi), static_cast<h16 *>(diff_planes->lo), diff_planes->ld, diff_planes->slot, nullptr};
    planes_written = true;
  }
  // 16-byte accesses under softmax_rows_kernel's rule (same map: folding the Softmax into the loss changes no bit), every array's rows aligned
  const bool vec = softmax_rows_vec_ok(d.cols) && (d.stride & 3) == 0 && aligned16(net_out) && (diff_stride & 3) == 0 && aligned16(diff) &&
                   (!tgt || ((tgt_stride & 3) == 0 && aligned16(tgt))) && (!y_out || ((y_stride & 3) == 0 && aligned16(y_out))) &&
                   (!po.hi || ((po.ld & 3) == 0 && aligned16(po.hi) && aligned16(po.lo)));
#define XENT_LAUNCH_P(DENSE, SM, P)                                                                                                       \
  do {                                                                                                                                    \
    if (vec) hipLaunchKernelGGL((xent_rows_kernel<DENSE, SM, P, true>), dim3(g), dim3(256), 0, cur_stream(), net_out, d.stride, tgt, tgt_stride, labels, \
                                frame_weights, diff, diff_stride, d.rows, d.cols, rowstats, y_out, y_stride, po);                       \
    else hipLaunchKernelGGL((xent_rows_kernel<DENSE, SM, P>), dim3(g), dim3(256), 0, cur_stream(), net_out, d.stride, tgt, tgt_stride, labels, \
                            frame_weights, diff, diff_stride, d.rows, d.cols, rowstats, y_out, y_stride, po);                           \
  } while (0)
#define XENT_LAUNCH(DENSE, SM)                                        \
  do {                                                                \
    if (per <= 4) XENT_LAUNCH_P(DENSE, SM, 4);                        \
    else if (per <= 8) XENT_LAUNCH_P(DENSE, SM, 8);                   \
    else if (per <= 16) XENT_LAUNCH_P(DENSE, SM, 16);                 \
    else XENT_LAUNCH_P(DENSE, SM, 32);                                \
  } while (0)
  if (tgt) { if (softmax) XENT_LAUNCH(true, true); else XENT_LAUNCH(true, false); }
  else { if (softmax) XENT_LAUNCH(false, true); else XENT_LAUNCH(false, false); }
#undef XENT_LAUNCH
#undef XENT_LAUNCH_P
  // (folding this sum into the rows' launch behind a ticket was measured: 54 us instead of 14 -- the device-scope release in front of the
  //  ticket makes every one of the 1024 workgroups write the L2 back, 12 MB of diff included)
  if (!rowstats_out) hipLaunchKernelGGL(xent_finalize_kernel, dim3(1), dim3(256), 0, cur_stream(), rowstats, d.rows, stats_dev);
  check_launch("xent_eval");
  return planes_written;
}

void aslp_xent_eval(const float *net_out, MatrixDim d, const float *tgt, int tgt_stride, const int32_cuda *labels, const float *frame_weights,
                    float *diff, int diff_stride, double *stats_dev) {
  xent_eval_impl(net_out, d, tgt, tgt_stride, labels, frame_weights, diff, diff_stride, stats_dev, false, nullptr, 0);
}
int aslp_xent_eval_p(const float *net_out, MatrixDim d, const int32_cuda *labels, const float *frame_weights, float *diff, int diff_stride,
                     double *stats_dev, int softmax, const aslp_planes_out *diff_planes) {
  if (softmax && !aslp_softmax_xent_supported(d.cols)) { set_error("aslp_softmax_xent_eval: unsupported number of classes"); return 0; }
  return xent_eval_impl(net_out, d, nullptr, 0, labels, frame_weights, diff, diff_stride, stats_dev, softmax != 0, nullptr, 0, diff_planes) ? 1 : 0;
}
// dst (nullable) <- src and src's planes in one launch; 1 = done (planes_written set), 0 = not served: the caller copies and converts
int aslp_copy_mat_planes(float *dst, MatrixDim d, const float *src, int src_stride, aslp_planes_out *out) {
  if (out) { out->nparts = 0; out->planes_written = 0; }
  if (!out || !out->hi || !out->lo || !out->slot || d.rows <= 0 || d.cols <= 0 || out->ld < d.cols) return 0;
  CoopConvJob job = {src, src_stride, dst, d.stride, S16View{static_cast<h16 *>(out->hi), static_cast<h16 *>(out->lo), out->ld, d.rows, d.cols,
                                                              const_cast<unsigned *>(out->slot)}, nullptr, 0};
  if (!coop_convert_launch(&job, 1)) return 0;
  out->planes_written = 1;
  return 1;
}
void aslp_coop_convert(int on) { g_coop_convert_on = on != 0; }
int aslp_xent_eval_rows(const float *net_out, MatrixDim d, const int32_cuda *labels, const float *frame_weights, float *diff, int diff_stride,
                        double *rowstats_out, int softmax, const aslp_planes_out *diff_planes) {
  if (!rowstats_out) { set_error("aslp_xent_eval_rows: no room for the per-row statistics"); return 0; }
  if (softmax && !aslp_softmax_xent_supported(d.cols)) { set_error("aslp_softmax_xent_eval: unsupported number of classes"); return 0; }
  return xent_eval_impl(net_out, d, nullptr, 0, labels, frame_weights, diff, diff_stride, nullptr, softmax != 0, nullptr, 0, diff_planes, rowstats_out) ? 1 : 0;
}
void aslp_xent_sum_rowstats(const double *rowstats, int rows, int batches, double *stats_dev) {
  if (rows <= 0 || batches <= 0) return;
  double *sums = static_cast<double *>(scratch(kScratchReduce, sizeof(double) * 5 * (size_t)batches));
  if (!sums) return;
  hipLaunchKernelGGL(xent_batch_sums_kernel, dim3(batches), dim3(256), 0, cur_stream(), rowstats, rows, sums);
  hipLaunchKernelGGL(xent_apply_kernel, dim3(1), dim3(64), 0, cur_stream(), sums, batches, stats_dev);
  check_launch("xent_sum_rowstats");
}
// Softmax + Xent::Eval in one pass over the activations in front of the Softmax; only rows of 513..8192 classes (the
// range where cudaF_softmax_reduce uses the same 256-lane row layout, so the posteriors are bit-identical)
int aslp_softmax_xent_supported(int cols) { return cols > 512 && cols <= 256 * kXentPerThread; }
void aslp_softmax_xent_eval(const float *acts, MatrixDim d, const float *tgt, int tgt_stride, const int32_cuda *labels,
                            const float *frame_weights, float *diff, int diff_stride, double *stats_dev, float *post_out, int post_stride) {
  if (!aslp_softmax_xent_supported(d.cols)) { set_error("aslp_softmax_xent_eval: unsupported number of classes"); return; }
  xent_eval_impl(acts, d, tgt, tgt_stride, labels, frame_weights, diff, diff_stride, stats_dev, true, post_out, post_stride);
}

}  // extern "C"
