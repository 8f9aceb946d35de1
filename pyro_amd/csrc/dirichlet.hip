// dirichlet.hip -- Dirichlet log-density rows and their gradients (SURVEY 8a rows a3 / a4 for the
// simplex-valued sites of examples/lda.py:45-60: topic_words, doc_topics):
//   log p(x | c) = sum_k xlogy(c_k - 1, x_k) + lgamma(sum_k c_k) - sum_k lgamma(c_k)
// (torch/distributions/dirichlet.py log_prob).  The reference evaluates it as five ATen kernels and
// their autograd duals per site; here one launch forward, one backward.  Operands are 2-D strided
// views [rows, K] (row stride 0 = a concentration vector shared by all rows: never materialised).
// Mapping: K <= 32 (topics) -> one thread per row; larger K (a vocabulary) -> one wave per row with
// DPP reductions, or one 1024-thread workgroup per row when the rows are few (round 6).  lgamma / digamma as in dist_fam.h.
#include "common.h"
#include "dist_fam.h"

namespace pa {

template <typename T>
__global__ __launch_bounds__(256) void dirichlet_lp_thread_kernel(T* __restrict__ out, ViewT<T> x,
                                                                  ViewT<T> c, int64_t rows, int K) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  T acc = T(0), csum = T(0);
  for (int k = 0; k < K; ++k) {
    const T ck = c.at(r, k);
    acc += t_xlogy(ck - T(1), x.at(r, k)) - t_lgamma(ck);
    csum += ck;
  }
  out[r] = acc + t_lgamma(csum);
}

template <typename T>
__global__ __launch_bounds__(256) void dirichlet_lp_wave_kernel(T* __restrict__ out, ViewT<T> x,
                                                                ViewT<T> c, int64_t rows, int K) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  double acc = 0.0, csum = 0.0;
  for (int k = lane; k < K; k += 64) {
    const T ck = c.at(r, k);
    acc += (double)(t_xlogy(ck - T(1), x.at(r, k)) - t_lgamma(ck));
    csum += (double)ck;
  }
  acc = wave_sum(acc);
  csum = wave_sum(csum);
  if (lane == 0) out[r] = (T)(acc + (double)t_lgamma((T)csum));
}

// d_x[r,k] = g[r] (c_k - 1) / x_k ;  d_c[r,k] = g[r] (log x_k + psi(sum c) - psi(c_k))
template <typename T>
__global__ __launch_bounds__(256) void dirichlet_grad_thread_kernel(
    const T* __restrict__ g, ViewT<T> x, ViewT<T> c, int64_t rows, int K, T* __restrict__ dx,
    T* __restrict__ dc) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  T csum = T(0);
  for (int k = 0; k < K; ++k) csum += c.at(r, k);
  const T psum = dc ? t_digamma(csum) : T(0);
  const T gr = g[r];
  for (int k = 0; k < K; ++k) {
    const T ck = c.at(r, k), xk = x.at(r, k);
    if (dx) dx[r * K + k] = gr * (ck - T(1)) / xk;
    if (dc) dc[r * K + k] = gr * (t_log(xk) + psum - t_digamma(ck));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dirichlet_grad_wave_kernel(
    const T* __restrict__ g, ViewT<T> x, ViewT<T> c, int64_t rows, int K, T* __restrict__ dx,
    T* __restrict__ dc) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  double csum = 0.0;
  for (int k = lane; k < K; k += 64) csum += (double)c.at(r, k);
  csum = wave_sum(csum);
  const T psum = dc ? t_digamma((T)csum) : T(0);
  const T gr = g[r];
  for (int k = lane; k < K; k += 64) {
    const T ck = c.at(r, k), xk = x.at(r, k);
    if (dx) dx[r * K + k] = gr * (ck - T(1)) / xk;
    if (dc) dc[r * K + k] = gr * (t_log(xk) + psum - t_digamma(ck));
  }
}

// Few rows of a long simplex (examples/lda.py's topic_words: 8 rows of 1024): one WORKGROUP of 1024 threads per row
// -- a wave per row leaves the chip to 8 waves whose lanes each walk 16 lgamma / digamma evaluations one after the
// other (10-15 us per launch for 8192 elements).  The sums: lanes -> waves (wave_sum) -> the 16 waves in index
// order through LDS (bit-reproducible).
template <typename T>
__device__ __forceinline__ double dirichlet_block_sum(double v, double* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                               // (sm may still be read from a previous sum)
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
  return t;
}

template <typename T>
__global__ __launch_bounds__(1024) void dirichlet_lp_block_kernel(T* __restrict__ out, ViewT<T> x, ViewT<T> c,
                                                                  int64_t rows, int K) {
  __shared__ double sm[16];
  const int64_t r = blockIdx.x;
  double acc = 0.0, csum = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const T ck = c.at(r, k);
    acc += (double)(t_xlogy(ck - T(1), x.at(r, k)) - t_lgamma(ck));
    csum += (double)ck;
  }
  acc = dirichlet_block_sum<T>(acc, sm);
  csum = dirichlet_block_sum<T>(csum, sm);
  if (threadIdx.x == 0) out[r] = (T)(acc + (double)t_lgamma((T)csum));
}

template <typename T>
__global__ __launch_bounds__(1024) void dirichlet_grad_block_kernel(const T* __restrict__ g, ViewT<T> x, ViewT<T> c,
                                                                    int64_t rows, int K, T* __restrict__ dx,
                                                                    T* __restrict__ dc) {
  __shared__ double sm[16];
  const int64_t r = blockIdx.x;
  double csum = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) csum += (double)c.at(r, k);
  csum = dirichlet_block_sum<T>(csum, sm);
  const T psum = dc ? t_digamma((T)csum) : T(0);
  const T gr = g[r];
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const T ck = c.at(r, k), xk = x.at(r, k);
    if (dx) dx[r * K + k] = gr * (ck - T(1)) / xk;
    if (dc) dc[r * K + k] = gr * (t_log(xk) + psum - t_digamma(ck));
  }
}

// a workgroup per row when a wave per row would leave most of the chip idle
static bool dirichlet_block_rows(int64_t rows, int64_t K) { return K >= 256 && rows * 4 <= (int64_t)cu_count() * 4; }

}  // namespace pa

extern "C" {

int pa_dirichlet_log_prob(int dtype, void* out, pa_view2d value, pa_view2d concentration,
                          int64_t rows, int64_t K, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "dirichlet_log_prob: bad dtype %d", dtype);
  PA_REQUIRE(rows >= 0 && K >= 1 && K < (1 << 30), "dirichlet_log_prob: bad shape");
  if (rows == 0) return PA_OK;
  PA_REQUIRE(out && value.ptr && concentration.ptr, "dirichlet_log_prob: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  const bool per_thread = K <= 32;
  const unsigned grid = (unsigned)(per_thread ? (rows + 255) / 256 : (rows + 3) / 4);
  if (pa::dirichlet_block_rows(rows, K)) {
    if (dtype == PA_F32)
      hipLaunchKernelGGL((pa::dirichlet_lp_block_kernel<float>), dim3((unsigned)rows), dim3(1024), 0, s, (float*)out,
                         pa::as_view<float>(value), pa::as_view<float>(concentration), rows, (int)K);
    else
      hipLaunchKernelGGL((pa::dirichlet_lp_block_kernel<double>), dim3((unsigned)rows), dim3(1024), 0, s, (double*)out,
                         pa::as_view<double>(value), pa::as_view<double>(concentration), rows, (int)K);
    return pa::check_launch("dirichlet_lp_block_kernel");
  }
  if (dtype == PA_F32) {
    auto x = pa::as_view<float>(value), c = pa::as_view<float>(concentration);
    if (per_thread)
      hipLaunchKernelGGL((pa::dirichlet_lp_thread_kernel<float>), dim3(grid), dim3(256), 0, s,
                         (float*)out, x, c, rows, (int)K);
    else
      hipLaunchKernelGGL((pa::dirichlet_lp_wave_kernel<float>), dim3(grid), dim3(256), 0, s,
                         (float*)out, x, c, rows, (int)K);
  } else {
    auto x = pa::as_view<double>(value), c = pa::as_view<double>(concentration);
    if (per_thread)
      hipLaunchKernelGGL((pa::dirichlet_lp_thread_kernel<double>), dim3(grid), dim3(256), 0, s,
                         (double*)out, x, c, rows, (int)K);
    else
      hipLaunchKernelGGL((pa::dirichlet_lp_wave_kernel<double>), dim3(grid), dim3(256), 0, s,
                         (double*)out, x, c, rows, (int)K);
  }
  return pa::check_launch("dirichlet_lp_kernel");
}

int pa_dirichlet_log_prob_grad(int dtype, const void* g, pa_view2d value, pa_view2d concentration,
                               int64_t rows, int64_t K, void* d_value, void* d_concentration,
                               pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "dirichlet_log_prob_grad: bad dtype %d", dtype);
  PA_REQUIRE(rows >= 0 && K >= 1 && K < (1 << 30), "dirichlet_log_prob_grad: bad shape");
  if (rows == 0 || (!d_value && !d_concentration)) return PA_OK;
  PA_REQUIRE(g && value.ptr && concentration.ptr, "dirichlet_log_prob_grad: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  const bool per_thread = K <= 32;
  const unsigned grid = (unsigned)(per_thread ? (rows + 255) / 256 : (rows + 3) / 4);
  if (pa::dirichlet_block_rows(rows, K)) {
    if (dtype == PA_F32)
      hipLaunchKernelGGL((pa::dirichlet_grad_block_kernel<float>), dim3((unsigned)rows), dim3(1024), 0, s,
                         (const float*)g, pa::as_view<float>(value), pa::as_view<float>(concentration), rows, (int)K,
                         (float*)d_value, (float*)d_concentration);
    else
      hipLaunchKernelGGL((pa::dirichlet_grad_block_kernel<double>), dim3((unsigned)rows), dim3(1024), 0, s,
                         (const double*)g, pa::as_view<double>(value), pa::as_view<double>(concentration), rows,
                         (int)K, (double*)d_value, (double*)d_concentration);
    return pa::check_launch("dirichlet_grad_block_kernel");
  }
  if (dtype == PA_F32) {
    auto x = pa::as_view<float>(value), c = pa::as_view<float>(concentration);
    if (per_thread)
      hipLaunchKernelGGL((pa::dirichlet_grad_thread_kernel<float>), dim3(grid), dim3(256), 0, s,
                         (const float*)g, x, c, rows, (int)K, (float*)d_value,
                         (float*)d_concentration);
    else
      hipLaunchKernelGGL((pa::dirichlet_grad_wave_kernel<float>), dim3(grid), dim3(256), 0, s,
                         (const float*)g, x, c, rows, (int)K, (float*)d_value,
                         (float*)d_concentration);
  } else {
    auto x = pa::as_view<double>(value), c = pa::as_view<double>(concentration);
    if (per_thread)
      hipLaunchKernelGGL((pa::dirichlet_grad_thread_kernel<double>), dim3(grid), dim3(256), 0, s,
                         (const double*)g, x, c, rows, (int)K, (double*)d_value,
                         (double*)d_concentration);
    else
      hipLaunchKernelGGL((pa::dirichlet_grad_wave_kernel<double>), dim3(grid), dim3(256), 0, s,
                         (const double*)g, x, c, rows, (int)K, (double*)d_value,
                         (double*)d_concentration);
  }
  return pa::check_launch("dirichlet_grad_kernel");
}

}  // extern "C"
