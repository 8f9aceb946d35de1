// gamma.hip -- reparameterised standard-Gamma draws on the keyed Philox stream, with the implicit
// reparameterisation gradient d sample / d concentration.
//
// Reference path replaced: torch.distributions.Gamma.rsample (gamma.py:80-88) = torch._standard_gamma
// (ATen's Marsaglia-Tsang rejection sampler on the cuRAND/hipRAND Philox state: a
// distribution_elementwise_grid_stride_kernel whose draws depend on launch geometry) and, in the
// backward, torch._standard_gamma_grad; Beta and Dirichlet draws are built from it
// (pyro/distributions/torch.py wraps torch's classes; examples/lda.py:107-109 draws its guide's
// Gamma / Dirichlet sites this way on every step).
//
// Here every element owns a Philox stream keyed by (seed, block offset + element index): attempt k
// of the rejection loop reads block (offset + i, GAMMA_TAG | k) -- a pure function of the element,
// not of the launch, so the draw is reproducible under any geometry, shardable across ranks and
// replay-safe inside a hipGraph (the base offset comes from device memory).  Marsaglia & Tsang
// (2000): d = a - 1/3, c = 1/sqrt(9 d); x ~ N(0,1), v = (1 + c x)^3, accept if
// log u < x^2/2 + d - d v + d log v; a < 1 is boosted through Gamma(a + 1) u'^(1/a).  The loop has a
// fixed budget of 24 attempts (acceptance >= 0.95 per attempt: the budget fails with probability
// < 1e-31; the last candidate is then taken).
//
// The gradient is the implicit one (Figurnov et al. 2018): d x / d a = -(dP/da)(a, x) / p(x; a) with P
// the regularised lower incomplete gamma function.  dP/da comes from differentiating the two classic
// evaluations of P term by term (dual numbers): the power series for x < a + 1 and the modified-Lentz
// continued fraction of Q = 1 - P otherwise; the prefactor x^a e^-x / Gamma(a) divides out against
// the density, leaving  d x / d a = -x [S (ln x - psi(a)) + dS/da]  resp.  +x [h (ln x - psi(a)) + dh/da].
// Evaluated in fp64 whatever the tensor dtype (these sites are small: concentrations of a guide).
#include "common.h"
#include "dist_fam.h"

namespace pa {

constexpr uint64_t GAMMA_TAG = 0x47414d4d00000000ull;     // "GAMM": the high counter word's key
constexpr int GAMMA_ATTEMPTS = 24;

// standard Gamma(a) draw of the element whose Philox block is `block`; fp64.  Attempt k reads the
// words (x, y | z | w) of block (block, GAMMA_TAG | k): a 53-bit and a 32-bit uniform for the normal
// (Box-Muller, cosine branch) and a 32-bit uniform for the acceptance test; the boost of a < 1 reads
// (x, y) of block (block, GAMMA_TAG | 255).
__device__ __forceinline__ double gamma_draw(double a, uint64_t seed, uint64_t block) {
  const double a1 = a < 1.0 ? a + 1.0 : a;
  const double d = a1 - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  double cand = d;
  for (int k = 0; k < GAMMA_ATTEMPTS; ++k) {
    const u32x4 r = philox4x32_10(seed, block, GAMMA_TAG | (uint64_t)k);
    const double u1 = u32x2_to_unit_f64(r.x, r.y);
    const double u2 = ((double)r.z + 0.5) * (1.0 / 4294967296.0);
    const double ua = ((double)r.w + 0.5) * (1.0 / 4294967296.0);
    const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    const double t = 1.0 + c * x;
    if (t <= 0.0) continue;
    const double v = t * t * t;
    cand = d * v;
    if (log(ua) < 0.5 * x * x + d - d * v + d * log(v)) break;
  }
  if (a < 1.0) {
    const u32x4 r = philox4x32_10(seed, block, GAMMA_TAG | 255ull);
    cand *= pow(u32x2_to_unit_f64(r.x, r.y), 1.0 / a);
  }
  return cand;
}

// d x / d a at (a, x), see the header
// `tol`: where the series / continued fraction stop.  1e-17 / 1e-16 for fp64 tensors; a float32 tensor rounds
// the result to 6e-8 anyway and stops at 1e-10 (both converge at least geometrically: a third fewer of the
// serial fp64 iterations that bound this kernel -- one lane's longest chain is the launch's duration).
__device__ __forceinline__ double gamma_implicit_grad(double a, double x, double tol_series = 1e-17,
                                                      double tol_cf = 1e-16) {
  if (!(x > 0.0) || !(a > 0.0)) return 0.0;
  // Both evaluations below need ~ c sqrt(a) terms near x ~ a (the terms only start to fall once
  // n > x - a and then fall like exp(-n^2 / 2a)): the budget grows with sqrt(a) -- a fixed 500 was
  // silently truncated from a ~ 2e4 upwards.  From a = 1e4 on, a draw within 8 standard deviations of
  // the mean takes the Cornish-Fisher form of the quantile instead: with the standard-normal
  // quantile z held fixed (that is what "the same draw at a different concentration" means),
  //   x(a, z) = a + sqrt(a) z + (z^2-1)/3 + (z^3-7z)/(36 sqrt(a)) - (3z^4+7z^2-16)/(810 a)
  //               + (9z^5+256z^3-433z)/(38880 a^1.5) + O(a^-2)
  // (cumulants a (n-1)!), z from x by four Newton steps, and d x / d a its partial derivative in a:
  // within 4e-11 of the series at a = 1e4 and better above (the first dropped term is O(a^-3) in the
  // derivative), a dozen flops instead of 2 000 - 160 000 serial iterations.  Tails beyond 8 sigma
  // keep the series / continued fraction.
  if (a >= 1e4 && fabs(x - a) <= 8.0 * sqrt(a)) {
    const double s = sqrt(a);
    double z = (x - a) / s;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const double z2 = z * z;
      const double f = a + s * z + (z2 - 1.0) / 3.0 + (z2 * z - 7.0 * z) / (36.0 * s) -
                       (3.0 * z2 * z2 + 7.0 * z2 - 16.0) / (810.0 * a) +
                       (9.0 * z2 * z2 * z + 256.0 * z2 * z - 433.0 * z) / (38880.0 * a * s) - x;
      const double fp = s + 2.0 * z / 3.0 + (3.0 * z2 - 7.0) / (36.0 * s) -
                        (12.0 * z2 * z + 14.0 * z) / (810.0 * a) +
                        (45.0 * z2 * z2 + 768.0 * z2 - 433.0) / (38880.0 * a * s);
      z -= f / fp;
    }
    const double z2 = z * z;
    return 1.0 + z / (2.0 * s) - (z2 * z - 7.0 * z) / (72.0 * a * s) +
           (3.0 * z2 * z2 + 7.0 * z2 - 16.0) / (810.0 * a * a) -
           1.5 * (9.0 * z2 * z2 * z + 256.0 * z2 * z - 433.0 * z) / (38880.0 * a * a * s);
  }
  if (a > 1e8) return 1.0 + (x - a) / (2.0 * a);   // (a tail draw out there: the normal limit)
  const int budget = 500 + (int)(16.0 * sqrt(a));
  const double lx_psi = log(x) - t_digamma<double>(a);
  if (x < a + 1.0) {
    // S = sum_n x^n / (a (a+1) ... (a+n)) and its derivative in a
    double t = 1.0 / a, dt = -1.0 / (a * a), S = t, dS = dt;
    for (int n = 1; n < budget; ++n) {
      const double den = a + n, f = x / den;
      dt = dt * f - t * f / den;
      t *= f;
      S += t;
      dS += dt;
      if (fabs(t) < tol_series * fabs(S) && fabs(dt) < tol_series * fabs(dS)) break;
    }
    return -x * (S * lx_psi + dS);
  }
  // modified Lentz on Q = pref * h, h = 1 / (x+1-a - 1(1-a)/(x+3-a - 2(2-a)/(x+5-a - ...))), with
  // every quantity carried together with its derivative in a
  const double tiny = 1e-300;
  double b = x + 1.0 - a, db = -1.0;
  double c = 1.0 / tiny, dc = 0.0;
  double d = 1.0 / b, dd = -db / (b * b);
  double h = d, dh = dd;
  for (int i = 1; i < budget; ++i) {
    const double an = -(double)i * ((double)i - a), dan = (double)i;
    b += 2.0;                                     // (db stays -1)
    double dn = an * d + b, ddn = dan * d + an * dd + db;
    if (fabs(dn) < tiny) dn = tiny;
    double cn = b + an / c, dcn = db + dan / c - an * dc / (c * c);
    if (fabs(cn) < tiny) cn = tiny;
    d = 1.0 / dn;
    dd = -ddn / (dn * dn);
    c = cn;
    dc = dcn;
    const double del = d * c, ddel = dd * c + d * dc;
    dh = dh * del + h * ddel;
    h *= del;
    if (fabs(del - 1.0) < tol_cf && fabs(ddel) < tol_cf) break;
  }
  return x * (h * lx_psi + dh);
}

template <typename T>
__global__ __launch_bounds__(256) void gamma_rsample_kernel(T* __restrict__ out, T* __restrict__ dalpha,
                                                            ViewT<T> alpha, int64_t rows, int64_t cols,
                                                            uint64_t seed, uint64_t offset,
                                                            const uint64_t* __restrict__ offset_dev) {
  const uint64_t off = offset + (offset_dev ? *offset_dev : 0);
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double a = (double)alpha.at(i / cols, i % cols);
    double x = gamma_draw(a, seed, off + (uint64_t)i);
    // the smallest positive value of the dtype instead of 0 (torch clamps the same way)
    const double lo = sizeof(T) == 4 ? 1.1754943508222875e-38 : 2.2250738585072014e-308;
    x = x > lo ? x : lo;
    out[i] = (T)x;
    if (dalpha != nullptr)
      dalpha[i] = (T)gamma_implicit_grad(a, (double)(T)x, sizeof(T) == 4 ? 1e-10 : 1e-17,
                                         sizeof(T) == 4 ? 1e-10 : 1e-16);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gamma_grad_kernel(T* __restrict__ dalpha, ViewT<T> alpha,
                                                         ViewT<T> value, int64_t rows, int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    dalpha[i] = (T)gamma_implicit_grad((double)alpha.at(i / cols, i % cols),
                                       (double)value.at(i / cols, i % cols), sizeof(T) == 4 ? 1e-10 : 1e-17,
                                       sizeof(T) == 4 ? 1e-10 : 1e-16);
}

}  // namespace pa

extern "C" {

int pa_gamma_rsample(int dtype, void* out, void* d_alpha, pa_view2d alpha, int64_t rows, int64_t cols,
                     uint64_t seed, uint64_t offset, const uint64_t* offset_dev, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "gamma_rsample: bad dtype %d", dtype);
  PA_REQUIRE(rows >= 0 && cols >= 0, "gamma_rsample: bad shape");
  if (rows * cols == 0) return PA_OK;
  PA_REQUIRE(out && alpha.ptr, "gamma_rsample: NULL pointer");
  int64_t grid = (rows * cols + 255) / 256;
  const int64_t cap = (int64_t)pa::cu_count() * 8;
  if (grid > cap) grid = cap;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::gamma_rsample_kernel<float>), dim3((unsigned)grid), dim3(256), 0, s,
                       (float*)out, (float*)d_alpha, pa::as_view<float>(alpha), rows, cols, seed, offset,
                       offset_dev);
  else
    hipLaunchKernelGGL((pa::gamma_rsample_kernel<double>), dim3((unsigned)grid), dim3(256), 0, s,
                       (double*)out, (double*)d_alpha, pa::as_view<double>(alpha), rows, cols, seed,
                       offset, offset_dev);
  return pa::check_launch("gamma_rsample_kernel");
}

int pa_gamma_implicit_grad(int dtype, void* d_alpha, pa_view2d alpha, pa_view2d value, int64_t rows,
                           int64_t cols, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "gamma_implicit_grad: bad dtype %d", dtype);
  PA_REQUIRE(rows >= 0 && cols >= 0, "gamma_implicit_grad: bad shape");
  if (rows * cols == 0) return PA_OK;
  PA_REQUIRE(d_alpha && alpha.ptr && value.ptr, "gamma_implicit_grad: NULL pointer");
  int64_t grid = (rows * cols + 255) / 256;
  const int64_t cap = (int64_t)pa::cu_count() * 8;
  if (grid > cap) grid = cap;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::gamma_grad_kernel<float>), dim3((unsigned)grid), dim3(256), 0, s,
                       (float*)d_alpha, pa::as_view<float>(alpha), pa::as_view<float>(value), rows, cols);
  else
    hipLaunchKernelGGL((pa::gamma_grad_kernel<double>), dim3((unsigned)grid), dim3(256), 0, s,
                       (double*)d_alpha, pa::as_view<double>(alpha), pa::as_view<double>(value), rows,
                       cols);
  return pa::check_launch("gamma_grad_kernel");
}

}  // extern "C"
