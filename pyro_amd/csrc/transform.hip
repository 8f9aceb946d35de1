// transform.hip -- a latent with support (lower, inf) under a mean-field Normal guide, in one launch each
// way.
//
// Reference: AutoNormal.forward (pyro/infer/autoguide/guides.py:494-519) maps the unconstrained draw u
// through biject_to(site.support) -- for constraints.positive / greater_than / greater_than_eq that is
// ExpTransform (composed with an AffineTransform(lower, 1)) -- and scores the site with a Delta whose
// log-density is transform.inv.log_abs_det_jacobian(value, u) summed over the site's event dims:
//     value = lower + exp(u),        log_density = - sum_event u
// As torch operators that is exp, mul, add, exp (again, inside the composed Jacobian), full_like, add,
// sum, neg and eight autograd duals, each a 5-us node of a captured step.  Here: one kernel forward, one
// backward ( d u = d value * exp(u) - d log_density ).
#include "common.h"

namespace pa {

// one wave per row of C = prod(event dims) elements; rows = everything to the left
template <typename T>
__global__ __launch_bounds__(256) void exp_site_fwd_kernel(const T* __restrict__ u, int64_t R, int64_t C,
                                                           T lower, T* __restrict__ value,
                                                           T* __restrict__ ld) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < R; r += nwaves) {
    T acc = T(0);
    for (int64_t c = lane; c < C; c += 64) {
      const T x = u[r * C + c];
      T e;
      if constexpr (sizeof(T) == 4) e = expf(x);
      else e = exp(x);
      value[r * C + c] = lower + e;
      acc += x;
    }
    acc = wave_sum(acc);
    if (lane == 0) ld[r] = -acc;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void exp_site_bwd_kernel(const T* __restrict__ value, const T* __restrict__ g_value,
                                                           const T* __restrict__ g_ld, int64_t R, int64_t C,
                                                           T lower, T* __restrict__ g_u) {
  const int64_t n = R * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T g = T(0);
    if (g_value != nullptr) g = g_value[i] * (value[i] - lower);
    if (g_ld != nullptr) g -= g_ld[i / C];
    g_u[i] = g;
  }
}

}  // namespace pa

extern "C" {

int pa_exp_site_fwd(int dtype, const void* u, int64_t rows, int64_t cols, double lower, void* value,
                    void* log_density, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "exp_site_fwd: bad dtype %d", dtype);
  PA_REQUIRE(rows >= 0 && cols >= 1, "exp_site_fwd: bad shape %lld x %lld", (long long)rows, (long long)cols);
  if (rows == 0) return PA_OK;
  PA_REQUIRE(u && value && log_density, "exp_site_fwd: NULL pointer");
  int64_t grid = (rows + 3) / 4;
  const int64_t cap = (int64_t)pa::cu_count() * 8;
  if (grid > cap) grid = cap;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL(pa::exp_site_fwd_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, (const float*)u,
                       rows, cols, (float)lower, (float*)value, (float*)log_density);
  else
    hipLaunchKernelGGL(pa::exp_site_fwd_kernel<double>, dim3((unsigned)grid), dim3(256), 0, s,
                       (const double*)u, rows, cols, lower, (double*)value, (double*)log_density);
  return pa::check_launch("exp_site_fwd_kernel");
}

int pa_exp_site_bwd(int dtype, const void* value, const void* g_value, const void* g_log_density, int64_t rows,
                    int64_t cols, double lower, void* g_u, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "exp_site_bwd: bad dtype %d", dtype);
  PA_REQUIRE(rows >= 0 && cols >= 1, "exp_site_bwd: bad shape %lld x %lld", (long long)rows, (long long)cols);
  if (rows == 0) return PA_OK;
  PA_REQUIRE(value && g_u, "exp_site_bwd: NULL pointer");
  int64_t grid = (rows * cols + 255) / 256;
  const int64_t cap = (int64_t)pa::cu_count() * 8;
  if (grid > cap) grid = cap;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL(pa::exp_site_bwd_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, (const float*)value,
                       (const float*)g_value, (const float*)g_log_density, rows, cols, (float)lower, (float*)g_u);
  else
    hipLaunchKernelGGL(pa::exp_site_bwd_kernel<double>, dim3((unsigned)grid), dim3(256), 0, s,
                       (const double*)value, (const double*)g_value, (const double*)g_log_density, rows, cols,
                       lower, (double*)g_u);
  return pa::check_launch("exp_site_bwd_kernel");
}

}  // extern "C"
