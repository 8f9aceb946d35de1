// logchain.hip -- a chain of enumerated discrete variables summed out in ONE launch.
//
// Reference path replaced: for a model written with pyro.markov (examples/hmm.py) TraceEnum_ELBO
// contracts the T enumerated states pairwise in log space (pyro/ops/contract.py:79-202 ->
// pyro/ops/einsum/torch_log.py:14-55: max-shift, exp, einsum, log per pair) and differentiates
// through every one of those steps: several thousand small launches per ELBO gradient for a
// sequence of a hundred steps.  The sum over a chain
//     Z[b] = sum_{v_0..v_{T-1}} prod_t exp(u[b,t,v_t]) prod_t exp(P[b,t,v_t,v_{t+1}])
// is the forward algorithm, and its gradient the forward-backward posteriors:
//     d log Z / d u[b,t,j]   = gamma_t(j)      d log Z / d P[b,t,i,j] = xi_t(i,j).
// One wave per batch element b (sequence): lane j owns state j (K <= 64), the alpha recursion runs
// over t with the transition matrix of the step staged in LDS by coalesced loads; the same launch
// then runs the beta recursion and writes gamma and xi.  Sequential in t by nature; parallel over
// the batch (plates) -- HBM traffic is the potentials read twice and the posteriors written once.
#include <limits>

#include "common.h"
#include "dist_fam.h"

namespace pa {

constexpr int LC_MAXK = 64;

template <typename T>
__device__ __forceinline__ T lc_neg_inf() { return -std::numeric_limits<T>::infinity(); }

// online log-sum-exp accumulation of value v into (m, s)
template <typename T>
__device__ __forceinline__ void lse_push(T v, T& m, T& s) {
  if (v == lc_neg_inf<T>()) return;
  if (v > m) {
    s = s * t_exp(m - v) + T(1);
    m = v;
  } else {
    s += t_exp(v - m);
  }
}
template <typename T>
__device__ __forceinline__ T lse_value(T m, T s) {
  return m == lc_neg_inf<T>() ? m : m + t_log(s);
}

// U[B, T, K]; P[b, t] = P + b * spb + t * spt, row-major [K, K] (i = state at t, j = state at t+1);
// alpha: workspace [B, T, K]; outputs logZ[B], gamma[B, T, K], xi[B, T-1, K, K].
template <typename T>
__global__ __launch_bounds__(64) void logchain_kernel(const T* __restrict__ U,
                                                      const T* __restrict__ P, int64_t spb,
                                                      int64_t spt, int Tn, int K,
                                                      T* __restrict__ alpha,
                                                      T* __restrict__ logZ,
                                                      T* __restrict__ gamma,
                                                      T* __restrict__ xi) {
  __shared__ T sP[LC_MAXK * LC_MAXK];
  __shared__ T sv[LC_MAXK];
  const int64_t b = blockIdx.x;
  const int j = threadIdx.x;
  const bool act = j < K;
  const T NEG = lc_neg_inf<T>();
  const T* Ub = U + b * (int64_t)Tn * K;
  const T* Pb = P + b * spb;
  T* ab = alpha + b * (int64_t)Tn * K;
  T* gb = gamma + b * (int64_t)Tn * K;
  T* xb = xi + b * (int64_t)(Tn - 1) * K * K;

  // ---- forward ------------------------------------------------------------------------------
  T a = act ? Ub[j] : NEG;
  if (act) ab[j] = a;
  for (int t = 1; t < Tn; ++t) {
    __syncthreads();
    if (act) sv[j] = a;
    const T* Pt = Pb + (int64_t)(t - 1) * spt;
    for (int e = j; e < K * K; e += 64) sP[e] = Pt[e];
    __syncthreads();
    T m = NEG, s = T(0);
    if (act)
      for (int i = 0; i < K; ++i) lse_push(sv[i] + sP[i * K + j], m, s);
    a = act ? Ub[(int64_t)t * K + j] + lse_value(m, s) : NEG;
    if (act) ab[(int64_t)t * K + j] = a;
  }
  // log Z = LSE_j alpha_{T-1}[j] over the wave
  T mz = a;
  for (int o = 32; o > 0; o >>= 1) {
    const T other = __shfl_xor(mz, o, 64);
    mz = other > mz ? other : mz;
  }
  T sz = (act && a != NEG) ? t_exp(a - mz) : T(0);
  for (int o = 32; o > 0; o >>= 1) sz += __shfl_xor(sz, o, 64);
  const T lz = mz == NEG ? NEG : mz + t_log(sz);
  if (j == 0) logZ[b] = lz;

  // ---- backward: beta recursion, posteriors ---------------------------------------------------
  T beta = T(0);
  if (act) gb[(int64_t)(Tn - 1) * K + j] = (lz == NEG) ? T(0) : t_exp(a + beta - lz);
  for (int t = Tn - 2; t >= 0; --t) {
    // w[j] = u[t+1][j] + beta_{t+1}[j]; alpha_t in LDS for the xi pass
    const T w = act ? Ub[(int64_t)(t + 1) * K + j] + beta : NEG;
    const T at = act ? ab[(int64_t)t * K + j] : NEG;
    __syncthreads();
    if (act) sv[j] = at;
    const T* Pt = Pb + (int64_t)t * spt;
    for (int e = j; e < K * K; e += 64) sP[e] = Pt[e];
    __syncthreads();
    // xi_t[i, j] (lane j, loop i): coalesced writes along j
    if (act) {
      T* xt = xb + (int64_t)t * K * K;
      for (int i = 0; i < K; ++i) {
        const T v = sv[i] + sP[i * K + j] + w - lz;
        xt[i * K + j] = (lz == NEG || v == NEG) ? T(0) : t_exp(v);
      }
    }
    __syncthreads();
    if (act) sv[j] = w;                       // now lane i needs w[j] for every j
    __syncthreads();
    T m = NEG, s = T(0);
    if (act)
      for (int jj = 0; jj < K; ++jj) lse_push(sP[j * K + jj] + sv[jj], m, s);   // lane = i
    beta = act ? lse_value(m, s) : T(0);
    if (act) gb[(int64_t)t * K + j] = (lz == NEG) ? T(0) : t_exp(at + beta - lz);
  }
}

}  // namespace pa

extern "C" {

size_t pa_logchain_workspace(int dtype, int64_t B, int64_t T, int64_t K) {
  if (B <= 0 || T <= 0 || K <= 0) return 0;
  return (size_t)(B * T * K) * (dtype == PA_F32 ? 4 : 8);
}

int pa_logchain_fwd_bwd(int dtype, const void* unary, const void* pairwise,
                        int64_t pair_stride_batch, int64_t pair_stride_step, int64_t B,
                        int64_t T, int64_t K, void* log_z, void* grad_unary, void* grad_pairwise,
                        void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "logchain: bad dtype %d", dtype);
  PA_REQUIRE(B >= 0 && T >= 1 && K >= 1, "logchain: bad shape B=%lld T=%lld K=%lld", (long long)B,
             (long long)T, (long long)K);
  if (K > pa::LC_MAXK)
    return pa::fail(PA_ERR_UNSUPPORTED, "logchain: K=%lld > %d states", (long long)K, pa::LC_MAXK);
  PA_REQUIRE(B < (int64_t(1) << 31) && T < (int64_t(1) << 31), "logchain: shape too large");
  if (B == 0) return PA_OK;
  PA_REQUIRE(unary && log_z && grad_unary, "logchain: NULL pointer");
  PA_REQUIRE(T == 1 || (pairwise && grad_pairwise), "logchain: NULL pairwise pointer");
  PA_REQUIRE(pair_stride_batch >= 0 && pair_stride_step >= 0, "logchain: negative stride");
  PA_REQUIRE(workspace && workspace_bytes >= pa_logchain_workspace(dtype, B, T, K),
             "logchain: workspace too small");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::logchain_kernel<float>), dim3((unsigned)B), dim3(64), 0, s,
                       (const float*)unary, (const float*)pairwise, pair_stride_batch,
                       pair_stride_step, (int)T, (int)K, (float*)workspace, (float*)log_z,
                       (float*)grad_unary, (float*)grad_pairwise);
  else
    hipLaunchKernelGGL((pa::logchain_kernel<double>), dim3((unsigned)B), dim3(64), 0, s,
                       (const double*)unary, (const double*)pairwise, pair_stride_batch,
                       pair_stride_step, (int)T, (int)K, (double*)workspace, (double*)log_z,
                       (double*)grad_unary, (double*)grad_pairwise);
  return pa::check_launch("logchain_kernel");
}

}  // extern "C"
