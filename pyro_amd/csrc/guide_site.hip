// guide_site.hip -- two per-site pieces of a mean-field Normal guide that are too large for the
// many-small-sites launch (multisite.hip) and too regular to deserve operator-by-operator autograd:
//
// (1) a latent with support (lower, inf), in one launch each way;
// (2) the score  sum log q(z)  of a site whose draw z = loc + scale * eps came from the guide itself,
//     with its TOTAL derivative in closed form (pa_meanfield_score, below).
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

// the value alone (a parameter with a positive / greater-than constraint: no Jacobian term is asked for)
template <typename T>
__global__ __launch_bounds__(256) void exp_lower_kernel(const T* __restrict__ u, int64_t n, T lower,
                                                        T* __restrict__ value) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const T x = u[i];
    T e;
    if constexpr (sizeof(T) == 4) e = expf(x);
    else e = exp(x);
    value[i] = lower + e;
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

// (2) Reference: Trace_ELBO scores the guide site  z ~ Normal(loc, scale)  with Normal.log_prob and lets
// autograd differentiate it (pyro/infer/trace_elbo.py:142-160; torch/distributions/normal.py log_prob):
//     log q = -log scale - (z - loc)^2 / (2 scale^2) - log sqrt(2 pi)
// through z = loc + scale * eps AND through loc, scale directly.  With eps held fixed (the reparameterised
// draw) the sum over the three paths is  d/d loc = 0,  d/d scale = -1/scale  per element: the eps terms
// cancel exactly.  So the site needs ONE pass over z (for the value) and no pass at all for the gradient:
//     partial[b] = coef * sum_{elements of block b} ( -e^2/2 ),  e = (z - loc) / scale
//                + coef * ( -P * sum_{columns of block b} log scale_c )      [+ the constant in block 0]
//     gscale[c]  = -coef * P / scale_c
// Per-thread sums in double, a fixed-order LDS tree per block: deterministic.
template <typename T>
__global__ __launch_bounds__(256) void meanfield_score_kernel(const T* __restrict__ z, const T* __restrict__ loc,
                                                              const T* __restrict__ scale, int64_t P, int64_t n,
                                                              double coef, T* __restrict__ partial,
                                                              T* __restrict__ gscale) {
  __shared__ double sm[256];
  const int64_t total = P * n, nb = gridDim.x;
  const int64_t per = ((total + nb - 1) / nb + 255) / 256 * 256;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < total ? lo + per : total;
  double acc = 0.0;
  int64_t c = (lo + threadIdx.x) % n;
  const int64_t step = 256 % n;                     // the column advances by this (mod n) per iteration
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double e = ((double)z[i] - (double)loc[c]) / (double)scale[c];
    acc -= 0.5 * e * e;
    c += step;
    if (c >= n) c -= n;
  }
  const int64_t cper = (n + nb - 1) / nb;
  const int64_t c0 = (int64_t)blockIdx.x * cper, c1 = c0 + cper < n ? c0 + cper : n;
  for (int64_t c = c0 + threadIdx.x; c < c1; c += 256) {
    const double s = (double)scale[c];
    acc -= (double)P * log(s);
    gscale[c] = (T)(-coef * (double)P / s);
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double t = sm[0];
    if (blockIdx.x == 0) t -= 0.91893853320467274178 * (double)total;      // log sqrt(2 pi) per element
    partial[blockIdx.x] = (T)(coef * t);
  }
}

}  // namespace pa

extern "C" {

int64_t pa_meanfield_score_blocks(int64_t P, int64_t n) {
  if (P <= 0 || n <= 0) return 0;
  int64_t nb = (P * n + 4095) / 4096;
  return nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
}

int pa_meanfield_score(int dtype, const void* z, const void* loc, const void* scale, int64_t P, int64_t n,
                       double coef, void* partial, void* gscale, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "meanfield_score: bad dtype %d", dtype);
  PA_REQUIRE(P >= 1 && n >= 1 && P < (int64_t(1) << 40) / n, "meanfield_score: bad shape %lld x %lld",
             (long long)P, (long long)n);
  PA_REQUIRE(z && loc && scale && partial && gscale, "meanfield_score: NULL pointer");
  const int64_t nb = pa_meanfield_score_blocks(P, n);
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL(pa::meanfield_score_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float*)z,
                       (const float*)loc, (const float*)scale, P, n, coef, (float*)partial, (float*)gscale);
  else
    hipLaunchKernelGGL(pa::meanfield_score_kernel<double>, dim3((unsigned)nb), dim3(256), 0, s, (const double*)z,
                       (const double*)loc, (const double*)scale, P, n, coef, (double*)partial, (double*)gscale);
  return pa::check_launch("meanfield_score_kernel");
}

int pa_exp_site_fwd(int dtype, const void* u, int64_t rows, int64_t cols, double lower, void* value,
                    void* log_density, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "exp_site_fwd: bad dtype %d", dtype);
  PA_REQUIRE(rows >= 0 && cols >= 1, "exp_site_fwd: bad shape %lld x %lld", (long long)rows, (long long)cols);
  if (rows == 0) return PA_OK;
  PA_REQUIRE(u && value, "exp_site_fwd: NULL pointer");
  const int64_t cap = (int64_t)pa::cu_count() * 8;
  hipStream_t s = pa::as_stream(stream);
  if (log_density == nullptr) {                  // the value only, element by element
    const int64_t n = rows * cols;
    int64_t g1 = (n + 255) / 256;
    if (g1 > cap) g1 = cap;
    if (dtype == PA_F32)
      hipLaunchKernelGGL(pa::exp_lower_kernel<float>, dim3((unsigned)g1), dim3(256), 0, s, (const float*)u, n,
                         (float)lower, (float*)value);
    else
      hipLaunchKernelGGL(pa::exp_lower_kernel<double>, dim3((unsigned)g1), dim3(256), 0, s, (const double*)u, n,
                         lower, (double*)value);
    return pa::check_launch("exp_lower_kernel");
  }
  int64_t grid = (rows + 3) / 4;
  if (grid > cap) grid = cap;
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
