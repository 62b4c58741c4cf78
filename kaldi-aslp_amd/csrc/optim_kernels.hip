// optim_kernels.hip -- the element-wise solvers of the SOD model sync (aslp-parallel/optimizer.h:40-171) for gfx950.
//
// The reference runs each solver as 2-9 CuVector launches per tensor (AddVecVec, ApplyFloor, ApplyPow, InvertElements,
// MulElements, AddVec ...), each a full pass over HBM.  Here a sync step is two launches per tensor: the model delta before
// the all-reduce and, after it, ONE pass that advances the solver state, steps the parameter and records it as the new
// reference point -- 16-32 bytes per element instead of 60-100.  Arithmetic follows the reference operation by operation
// (same fp32 order, sqrtf for ApplyPow(0.5), the 1e-8 floors) so results agree to rounding.
#include "aslp_kernels.h"
#include "common.h"

namespace aslp {
namespace {

__global__ void __launch_bounds__(kBlock) vec_diff_kernel(float *__restrict__ out, const float *__restrict__ a, const float *__restrict__ b, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = a[i] - b[i];
}

__device__ __forceinline__ float inv_sqrt_floor(float v) { return 1.0f / sqrtf(v < 1e-8f ? 1e-8f : v); }

template <int SOLVER>
__global__ void __launch_bounds__(kBlock) sod_solve_kernel(aslp_sod_solver a, const float *__restrict__ grad, float *__restrict__ param,
                                                           float *__restrict__ prev, float *__restrict__ s1, float *__restrict__ s2, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float g = grad[i];
    float p = param[i];
    if (SOLVER == ASLP_SOD_SGD) {                       // optimizer.h:47-49
      p += -a.lr * g;
    } else if (SOLVER == ASLP_SOD_MOMENTUM) {           // :61-64
      const float m = a.lr * g + a.momentum * s1[i];
      s1[i] = m;
      p += -1.0f * m;
    } else if (SOLVER == ASLP_SOD_ADAGRAD || SOLVER == ASLP_SOD_RMSPROP) {  // :77-86, :99-108
      const float G = SOLVER == ASLP_SOD_ADAGRAD ? 1.0f * g * g + 1.0f * s1[i] : 0.1f * g * g + 0.9f * s1[i];
      s1[i] = G;
      p += -a.lr * (inv_sqrt_floor(G) * g);
    } else if (SOLVER == ASLP_SOD_ADADELTA) {           // :123-139
      const float G = (1.0f - a.gamma) * g * g + a.gamma * s1[i];
      s1[i] = G;
      const float D = s2[i];
      const float step = inv_sqrt_floor(G) * sqrtf(D < 1e-8f ? 1e-8f : D) * g;
      p += -1.0f * step;
      s2[i] = (1.0f - a.gamma) * step * step + a.gamma * D;
    } else {                                            // Adam, :156-170; bias corrections arrive precomputed on the host
      const float m = (1.0f - a.beta1) * g + a.beta1 * s1[i];
      const float v = (1.0f - a.beta2) * g * g + a.beta2 * s2[i];
      s1[i] = m;
      s2[i] = v;
      p += -a.lr * a.corr1 * (inv_sqrt_floor(v * a.corr2) * m);
    }
    param[i] = p;
    prev[i] = p;
  }
}

}  // namespace
}  // namespace aslp

using namespace aslp;

extern "C" {

void aslp_vec_diff(float *out, const float *a, const float *b, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(vec_diff_kernel, dim3(grid_for(n)), dim3(kBlock), 0, cur_stream(), out, a, b, n);
  check_launch("vec_diff");
}

void aslp_sod_solve(const aslp_sod_solver *a, const float *grad, float *param, float *prev, float *state1, float *state2, int n) {
  if (n <= 0) return;
  const dim3 g(grid_for(n)), b(kBlock);
  switch (a->solver) {
#define ASLP_SOD_CASE(S) case S: hipLaunchKernelGGL((sod_solve_kernel<S>), g, b, 0, cur_stream(), *a, grad, param, prev, state1, state2, n); break;
    ASLP_SOD_CASE(ASLP_SOD_SGD)
    ASLP_SOD_CASE(ASLP_SOD_MOMENTUM)
    ASLP_SOD_CASE(ASLP_SOD_ADAGRAD)
    ASLP_SOD_CASE(ASLP_SOD_RMSPROP)
    ASLP_SOD_CASE(ASLP_SOD_ADADELTA)
    ASLP_SOD_CASE(ASLP_SOD_ADAM)
#undef ASLP_SOD_CASE
    default: set_error("aslp_sod_solve: unknown solver"); return;
  }
  check_launch("sod_solve");
}

}  // extern "C"
