// logsumexp.hip -- one elimination step of the plated sum-product in log space
// (pyro/ops/contract.py:79-160 _contract_component -> pyro/ops/einsum/torch_log.py:14-55:
//  sum the aligned log-factors that mention a variable, max-shift, exp, sum over the variable, log):
//     out[kept] = logsumexp_r ( sum_k term_k[frame] )
// The reference (and the generic route here) does it as broadcasting adds that materialise the
// full frame, a max pass, an exp pass, a sum pass, a log -- and the autograd duals of each.  Here the
// aligned sum is never written: every term is read through its own strides (stride 0 = the term
// does not depend on that dim), forward is ONE pass with an online max / rescaled sum, and the
// backward writes the one tensor every term's gradient is a reduction of:
//     G[frame] = g_out[kept] * exp( sum_k term_k[frame] - out[kept] )        (the posterior weights)
// HBM-bound: forward reads each term once (a term constant along the reduced dim K times from
// cache), backward reads them once more and writes the frame.
#include "common.h"

namespace pa {

struct LseTermDev {
  const void* p;
  int64_t s[PA_LSE_MAX_DIMS];
};
struct LseArgs {
  int nterms, ndim, rdim, pad;
  int64_t sizes[PA_LSE_MAX_DIMS];
  LseTermDev t[PA_LSE_MAX_TERMS];
};

template <typename T> __device__ __forceinline__ T lse_exp(T x);
template <> __device__ __forceinline__ float lse_exp(float x) { return expf(x); }
template <> __device__ __forceinline__ double lse_exp(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T lse_log(T x);
template <> __device__ __forceinline__ float lse_log(float x) { return logf(x); }
template <> __device__ __forceinline__ double lse_log(double x) { return log(x); }

// thread = one kept element; the reduced dim is walked with the terms' own strides
template <typename T>
__global__ __launch_bounds__(256) void lse_fwd_kernel(const LseArgs a, T* __restrict__ out,
                                                      int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  int64_t base[PA_LSE_MAX_TERMS] = {0, 0, 0, 0}, rs[PA_LSE_MAX_TERMS] = {0, 0, 0, 0};
  int64_t rem = i;
  for (int d = a.ndim - 1; d >= 0; --d) {
    if (d == a.rdim) continue;
    const int64_t c = rem % a.sizes[d];
    rem /= a.sizes[d];
#pragma unroll
    for (int k = 0; k < PA_LSE_MAX_TERMS; ++k)
      if (k < a.nterms) base[k] += c * a.t[k].s[d];
  }
#pragma unroll
  for (int k = 0; k < PA_LSE_MAX_TERMS; ++k)
    if (k < a.nterms) rs[k] = a.t[k].s[a.rdim];
  const int64_t K = a.sizes[a.rdim];
  const T ninf = -__builtin_huge_val();
  // torch.logsumexp semantics at the edges: a NaN term makes the result NaN (a comparison with NaN
  // is false: without the flag the term would silently drop out and an enumerated model with a NaN
  // log-factor would report a finite loss and a zero gradient); a +inf term makes it +inf (the
  // rescaled sum would form exp(inf - inf))
  const T pinf = __builtin_huge_val();
  T m = ninf, s = T(0);
  bool has_nan = false, has_pinf = false;
  for (int64_t r = 0; r < K; ++r) {
    T v = T(0);
#pragma unroll
    for (int k = 0; k < PA_LSE_MAX_TERMS; ++k)
      if (k < a.nterms) v += ((const T*)a.t[k].p)[base[k] + r * rs[k]];
    if (v != v) {
      has_nan = true;
    } else if (v == pinf) {
      has_pinf = true;
    } else if (v > m) {                // (never taken for v = -inf)
      s = s * lse_exp(m - v) + T(1);   // m = -inf: exp(-inf) = 0
      m = v;
    } else if (v > ninf) {
      s += lse_exp(v - m);
    }
  }
  const T fin = s > T(0) ? m + lse_log(s) : ninf;
  out[i] = has_nan ? (T)__builtin_nan("") : (has_pinf ? pinf : fin);
}

// thread = one frame element
template <typename T>
__global__ __launch_bounds__(256) void lse_bwd_kernel(const LseArgs a, T* __restrict__ G,
                                                      const T* __restrict__ g_out,
                                                      const T* __restrict__ out, int64_t total) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= total) return;
  int64_t rem = j, kept = 0, kmul = 1;
  T v = T(0);
  int64_t off[PA_LSE_MAX_TERMS] = {0, 0, 0, 0};
  for (int d = a.ndim - 1; d >= 0; --d) {
    const int64_t c = rem % a.sizes[d];
    rem /= a.sizes[d];
#pragma unroll
    for (int k = 0; k < PA_LSE_MAX_TERMS; ++k)
      if (k < a.nterms) off[k] += c * a.t[k].s[d];
    if (d != a.rdim) {
      kept += c * kmul;
      kmul *= a.sizes[d];
    }
  }
#pragma unroll
  for (int k = 0; k < PA_LSE_MAX_TERMS; ++k)
    if (k < a.nterms) v += ((const T*)a.t[k].p)[off[k]];
  const T o = out[kept];
  const T ninf = -__builtin_huge_val();
  // NaN in -> NaN out (as autograd through torch.logsumexp); out = +inf: exp(v - inf) = 0 for the
  // finite terms and NaN for the +inf ones, torch's values
  if (v != v || o != o) G[j] = (T)__builtin_nan("");
  else G[j] = (v > ninf && o > ninf) ? g_out[kept] * lse_exp(v - o) : T(0);
}

static int lse_fill(LseArgs* a, int nterms, const pa_lse_term* terms, int ndim, const int64_t* sizes,
                    int rdim, int64_t* total, int64_t* kept, const char* who) {
  PA_REQUIRE(nterms >= 1 && nterms <= PA_LSE_MAX_TERMS, "%s: 1..%d terms, got %d", who,
             PA_LSE_MAX_TERMS, nterms);
  PA_REQUIRE(ndim >= 1 && ndim <= PA_LSE_MAX_DIMS, "%s: 1..%d dims, got %d", who, PA_LSE_MAX_DIMS,
             ndim);
  PA_REQUIRE(rdim >= 0 && rdim < ndim, "%s: reduced dim %d out of range", who, rdim);
  PA_REQUIRE(terms && sizes, "%s: NULL pointer", who);
  a->nterms = nterms; a->ndim = ndim; a->rdim = rdim; a->pad = 0;
  *total = 1; *kept = 1;
  for (int d = 0; d < PA_LSE_MAX_DIMS; ++d) {
    a->sizes[d] = d < ndim ? sizes[d] : 1;
    if (d < ndim) {
      PA_REQUIRE(sizes[d] >= 0, "%s: negative size", who);
      *total *= sizes[d];
      if (d != rdim) *kept *= sizes[d];
    }
  }
  for (int k = 0; k < PA_LSE_MAX_TERMS; ++k) {
    a->t[k].p = k < nterms ? terms[k].ptr : nullptr;
    for (int d = 0; d < PA_LSE_MAX_DIMS; ++d)
      a->t[k].s[d] = (k < nterms && d < ndim) ? terms[k].strides[d] : 0;
    PA_REQUIRE(k >= nterms || *total == 0 || terms[k].ptr, "%s: NULL term %d", who, k);
  }
  return PA_OK;
}

}  // namespace pa

extern "C" {

int pa_logsumexp_terms(int dtype, void* out, int nterms, const pa_lse_term* terms, int ndim,
                       const int64_t* sizes, int rdim, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "logsumexp_terms: bad dtype %d", dtype);
  pa::LseArgs a;
  int64_t total, kept;
  int rc = pa::lse_fill(&a, nterms, terms, ndim, sizes, rdim, &total, &kept, "logsumexp_terms");
  if (rc != PA_OK) return rc;
  if (kept == 0) return PA_OK;
  PA_REQUIRE(out, "logsumexp_terms: NULL out");
  hipStream_t s = pa::as_stream(stream);
  const unsigned grid = (unsigned)((kept + 255) / 256);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::lse_fwd_kernel<float>), dim3(grid), dim3(256), 0, s, a, (float*)out, kept);
  else
    hipLaunchKernelGGL((pa::lse_fwd_kernel<double>), dim3(grid), dim3(256), 0, s, a, (double*)out,
                       kept);
  return pa::check_launch("lse_fwd_kernel");
}

int pa_logsumexp_terms_grad(int dtype, void* G, const void* g_out, const void* out, int nterms,
                            const pa_lse_term* terms, int ndim, const int64_t* sizes, int rdim,
                            pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "logsumexp_terms_grad: bad dtype %d", dtype);
  pa::LseArgs a;
  int64_t total, kept;
  int rc = pa::lse_fill(&a, nterms, terms, ndim, sizes, rdim, &total, &kept, "logsumexp_terms_grad");
  if (rc != PA_OK) return rc;
  if (total == 0) return PA_OK;
  PA_REQUIRE(G && g_out && out, "logsumexp_terms_grad: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::lse_bwd_kernel<float>), dim3(grid), dim3(256), 0, s, a, (float*)G,
                       (const float*)g_out, (const float*)out, total);
  else
    hipLaunchKernelGGL((pa::lse_bwd_kernel<double>), dim3(grid), dim3(256), 0, s, a, (double*)G,
                       (const double*)g_out, (const double*)out, total);
  return pa::check_launch("lse_bwd_kernel");
}

}  // extern "C"
