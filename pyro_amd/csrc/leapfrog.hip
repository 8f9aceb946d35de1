// leapfrog.hip -- the momentum / position updates of the velocity-Verlet integrator
// (pyro/ops/integrator.py:45-65) for C chains of dimension D stored row-major [C,D].
// Pure streaming updates: read z, r, grad (+ inverse mass), write z, r. HBM/L2-bound.
#include "common.h"

namespace pa {

template <typename T, bool DRIFT>
__global__ __launch_bounds__(256) void leapfrog_kernel(T* __restrict__ z, T* __restrict__ r,
                                                       const T* __restrict__ grad,
                                                       const T* __restrict__ inv_mass,
                                                       int64_t im_stride_row,
                                                       const T* __restrict__ step,
                                                       int64_t step_stride, int64_t C, int64_t D) {
  const int64_t n = C * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i / D, d = i - c * D;
    const T eps = step[c * step_stride];
    // r(n+1/2) = r + 0.5 * eps * (-grad)          integrator.py:54-56 (and :62-63)
    const T rn = r[i] + T(0.5) * eps * (-grad[i]);
    r[i] = rn;
    if (DRIFT) {
      // z(n+1) = z + eps * dK/dr,  dK/dr = M^-1 r   integrator.py:58-60, adaptation.py:328-347
      const T v = inv_mass[c * im_stride_row + d] * rn;
      z[i] = z[i] + eps * v;
    }
  }
}

template <bool DRIFT>
static int launch(int dtype, void* z, void* r, const void* grad, const void* inv_mass,
                  int64_t im_stride_row, const void* step, int64_t step_stride, int64_t C,
                  int64_t D, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "leapfrog: bad dtype %d", dtype);
  PA_REQUIRE(C >= 0 && D >= 0, "leapfrog: negative shape");
  PA_REQUIRE(step_stride == 0 || step_stride == 1, "leapfrog: step_stride must be 0 or 1");
  const int64_t n = C * D;
  if (n == 0) return PA_OK;
  PA_REQUIRE(r && grad && step, "leapfrog: NULL operand");
  PA_REQUIRE(!DRIFT || (z && inv_mass), "leapfrog: NULL z / inv_mass");
  int64_t grid = (n + 255) / 256;
  const int64_t cap = (int64_t)cu_count() * 8;
  if (grid > cap) grid = cap;
  if (dtype == PA_F32)
    hipLaunchKernelGGL((leapfrog_kernel<float, DRIFT>), dim3((unsigned)grid), dim3(256), 0,
                       as_stream(stream), (float*)z, (float*)r, (const float*)grad,
                       (const float*)inv_mass, im_stride_row, (const float*)step, step_stride, C,
                       D);
  else
    hipLaunchKernelGGL((leapfrog_kernel<double, DRIFT>), dim3((unsigned)grid), dim3(256), 0,
                       as_stream(stream), (double*)z, (double*)r, (const double*)grad,
                       (const double*)inv_mass, im_stride_row, (const double*)step, step_stride, C,
                       D);
  return check_launch("leapfrog_kernel");
}

}  // namespace pa

extern "C" {

int pa_leapfrog_kick_drift(int dtype, void* z, void* r, const void* grad, const void* inv_mass,
                           int64_t im_stride_row, const void* step, int64_t step_stride, int64_t C,
                           int64_t D, pa_stream_t stream) {
  return pa::launch<true>(dtype, z, r, grad, inv_mass, im_stride_row, step, step_stride, C, D,
                          stream);
}

int pa_leapfrog_kick(int dtype, void* r, const void* grad, const void* step, int64_t step_stride,
                     int64_t C, int64_t D, pa_stream_t stream) {
  return pa::launch<false>(dtype, nullptr, r, grad, nullptr, 0, step, step_stride, C, D, stream);
}

}  // extern "C"
