// mvn_guide.hip -- the full-covariance Normal guide (AutoMultivariateNormal) in two launches.
//
// Reference path replaced: AutoMultivariateNormal.get_posterior builds
// MultivariateNormal(loc, scale_tril = scale[..., None] * scale_tril) (pyro/infer/autoguide/
// guides.py:855-965; scale = softplus(unconstrained), scale_tril = unit lower Cholesky =
// tril(unconstrained, -1) + I); its rsample is loc + scale_tril @ eps and its log_prob solves the
// triangular system back for eps (torch/distributions/multivariate_normal.py: _batch_mahalanobis)
// -- a few dozen small launches per step (triangular masks, trsm, reductions, their autograd
// duals).  The draw KNOWS eps, so here
//     z_p = loc + S * (L eps_p),   log q(z_p) = -|eps_p|^2 / 2 - sum_i log S_i - n/2 log(2 pi)
// come out of ONE kernel (one workgroup per particle), and ONE backward kernel turns (d z, d log q)
// into the gradients of the three unconstrained parameter tensors.
#include "common.h"
#include "dist_fam.h"

namespace pa {

template <typename T> __device__ __forceinline__ T mvn_softplus(T x) {
  // torch.nn.functional.softplus (threshold 20), the transform of the softplus_positive constraint
  return x > T(20) ? x : t_log1p(t_exp(x));
}

// grid = P workgroups
template <typename T>
__global__ __launch_bounds__(256) void mvn_tril_sample_kernel(
    const T* __restrict__ loc, const T* __restrict__ rho, const T* __restrict__ A,
    T* __restrict__ eps, T* __restrict__ z, T* __restrict__ logq, int n, uint64_t seed,
    uint64_t offset, const uint64_t* __restrict__ offset_dev, int eps_given) {
  extern __shared__ unsigned char smem_raw[];
  T* es = reinterpret_cast<T*>(smem_raw);                 // [n] this particle's eps
  __shared__ double red[16];
  const int64_t p = blockIdx.x;
  const uint64_t off = offset + (offset_dev ? *offset_dev : 0);
  double part = 0.0;                                      // -eps^2/2 - log S over this thread's j
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const int64_t i = p * n + j;
    T e;
    if (eps_given) {
      e = eps[i];
    } else {
      if constexpr (sizeof(T) == 4) e = philox_normal_f32(seed, off, (uint64_t)i);
      else e = philox_normal_f64(seed, off, (uint64_t)i);
      eps[i] = e;
    }
    es[j] = e;
    part -= 0.5 * (double)e * (double)e + (double)t_log(mvn_softplus<T>(rho[j]));
  }
  __syncthreads();
  if (n <= 128) {
    // small latent space: one thread per row, the row's loads independent of each other (the
    // factor stays in L2 / the vector cache: P workgroups read the same n*n values)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const T* row = A + (int64_t)i * n;
      T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
      int j = 0;
      for (; j + 3 < i; j += 4) {
        a0 += row[j] * es[j];
        a1 += row[j + 1] * es[j + 1];
        a2 += row[j + 2] * es[j + 2];
        a3 += row[j + 3] * es[j + 3];
      }
      for (; j < i; ++j) a0 += row[j] * es[j];
      z[p * n + i] = loc[i] + mvn_softplus<T>(rho[i]) * (((a0 + a1) + (a2 + a3)) + es[i]);
    }
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = wave; i < n; i += nw) {                    // wave-uniform rows, coalesced
      const T* row = A + (int64_t)i * n;
      T acc = T(0);
      for (int j = lane; j < i; j += 64) acc += row[j] * es[j];   // strictly lower part
      acc = wave_sum(acc);
      if (lane == 0) z[p * n + i] = loc[i] + mvn_softplus<T>(rho[i]) * (acc + es[i]);
    }
  }
  const double tot = block_sum_f64(part, red);
  if (threadIdx.x == 0) logq[p] = (T)(tot - 0.5 * (double)n * 1.8378770664093453);  // log(2 pi)
}

// grid = n workgroups (row i of the factor)
template <typename T>
__global__ __launch_bounds__(256) void mvn_tril_sample_bwd_kernel(
    const T* __restrict__ loc, const T* __restrict__ rho, const T* __restrict__ eps,
    const T* __restrict__ z, const T* __restrict__ d_z, const T* __restrict__ d_logq, int n,
    int P, T* __restrict__ d_loc, T* __restrict__ d_rho, T* __restrict__ d_A, int accumulate) {
  __shared__ double red[3][4];
  __shared__ T part[4][64];
  const int i = blockIdx.x;
  const T rho_i = rho[i], loc_i = loc[i];
  const T S = mvn_softplus<T>(rho_i);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // d A[i, j] = S_i * sum_p d_z[p, i] * eps[p, j] for j < i, 0 elsewhere (unit diagonal, zero
  // upper).  Wave w takes the particles p = w (mod 4); 64 columns per pass.
  if (d_A != nullptr)
    for (int jb = 0; jb < n; jb += 64) {
      const int j = jb + lane;
      T a0 = T(0), a1 = T(0);
      if (j < i && d_z != nullptr) {
        int p = wave;
        for (; p + 4 < P; p += 8) {
          a0 += d_z[(int64_t)p * n + i] * eps[(int64_t)p * n + j];
          a1 += d_z[(int64_t)(p + 4) * n + i] * eps[(int64_t)(p + 4) * n + j];
        }
        for (; p < P; p += 4) a0 += d_z[(int64_t)p * n + i] * eps[(int64_t)p * n + j];
      }
      part[wave][lane] = a0 + a1;
      __syncthreads();
      if (wave == 0 && j < n) {
        const int64_t o = (int64_t)i * n + j;
        const T acc = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        d_A[o] = (accumulate ? d_A[o] : T(0)) + S * acc;
      }
      __syncthreads();
    }
  // d loc_i = sum_p d_z[p, i];  d S_i = sum_p d_z[p, i] * u[p, i] - (sum_p d_logq[p]) / S_i,
  // u = L eps = (z - loc) / S
  double sl = 0.0, ss = 0.0, sg = 0.0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const T g = d_z != nullptr ? d_z[(int64_t)p * n + i] : T(0);
    sl += (double)g;
    ss += (double)g * (double)((z[(int64_t)p * n + i] - loc_i) / S);
    sg += d_logq != nullptr ? (double)d_logq[p] : 0.0;
  }
  sl = wave_sum(sl);
  ss = wave_sum(ss);
  sg = wave_sum(sg);
  if (lane == 0) { red[0][wave] = sl; red[1][wave] = ss; red[2][wave] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sl = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    sg = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    const double dS = ss - sg / (double)S;
    const double x = (double)rho_i;
    const double sig = x > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-x));   // d softplus / d x
    if (d_loc != nullptr) d_loc[i] = (accumulate ? d_loc[i] : T(0)) + (T)sl;
    if (d_rho != nullptr) d_rho[i] = (accumulate ? d_rho[i] : T(0)) + (T)(dS * sig);
  }
}

}  // namespace pa

extern "C" {

int pa_mvn_tril_sample(int dtype, const void* loc, const void* rho, const void* A, int64_t n,
                       int64_t P, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                       int eps_given, void* eps, void* z, void* logq, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "mvn_tril_sample: bad dtype %d", dtype);
  PA_REQUIRE(n >= 1 && n <= 4096, "mvn_tril_sample: n=%lld outside [1, 4096]", (long long)n);
  PA_REQUIRE(P >= 0 && P < (int64_t(1) << 31), "mvn_tril_sample: bad particle count");
  if (P == 0) return PA_OK;
  PA_REQUIRE(loc && rho && A && eps && z && logq, "mvn_tril_sample: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::mvn_tril_sample_kernel<float>), dim3((unsigned)P), dim3(256),
                       (size_t)n * 4, s, (const float*)loc, (const float*)rho, (const float*)A,
                       (float*)eps, (float*)z, (float*)logq, (int)n, seed, offset, offset_dev,
                       eps_given);
  else
    hipLaunchKernelGGL((pa::mvn_tril_sample_kernel<double>), dim3((unsigned)P), dim3(256),
                       (size_t)n * 8, s, (const double*)loc, (const double*)rho, (const double*)A,
                       (double*)eps, (double*)z, (double*)logq, (int)n, seed, offset, offset_dev,
                       eps_given);
  return pa::check_launch("mvn_tril_sample_kernel");
}

int pa_mvn_tril_sample_bwd(int dtype, const void* loc, const void* rho, const void* eps,
                           const void* z, const void* d_z, const void* d_logq, int64_t n,
                           int64_t P, void* d_loc, void* d_rho, void* d_A, int accumulate,
                           pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "mvn_tril_sample_bwd: bad dtype %d", dtype);
  PA_REQUIRE(n >= 1 && n <= 4096, "mvn_tril_sample_bwd: n=%lld outside [1, 4096]", (long long)n);
  PA_REQUIRE(P >= 0 && P < (int64_t(1) << 31), "mvn_tril_sample_bwd: bad particle count");
  PA_REQUIRE(loc && rho && eps && z, "mvn_tril_sample_bwd: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::mvn_tril_sample_bwd_kernel<float>), dim3((unsigned)n), dim3(256), 0, s,
                       (const float*)loc, (const float*)rho, (const float*)eps, (const float*)z,
                       (const float*)d_z, (const float*)d_logq, (int)n, (int)P, (float*)d_loc,
                       (float*)d_rho, (float*)d_A, accumulate);
  else
    hipLaunchKernelGGL((pa::mvn_tril_sample_bwd_kernel<double>), dim3((unsigned)n), dim3(256), 0,
                       s, (const double*)loc, (const double*)rho, (const double*)eps,
                       (const double*)z, (const double*)d_z, (const double*)d_logq, (int)n, (int)P,
                       (double*)d_loc, (double*)d_rho, (double*)d_A, accumulate);
  return pa::check_launch("mvn_tril_sample_bwd_kernel");
}

}  // extern "C"
