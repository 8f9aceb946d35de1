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

// ---- K <= 32: every lane works ------------------------------------------------------------------
// The kernel above gives a lane to a state: with the 16 states of examples/hmm.py three quarters of the wave
// idle, every step stages its matrix through LDS behind two barriers and folds 16 terms into a running
// log-sum-exp one exp after the other -- 3.6 us per step and direction, 940 us for 229 x 129 x 16 on the
// MI355X.  Here the wave is a KP x G grid (KP = 16 or 32 states padded, G = 64 / KP chunks): in the forward
// pass lane (j, g) holds the NI = KP / G terms i = g NI .. of alpha_t[i] + P[i, j] (alpha by ds_bpermute from
// the lane that owns state i), reduces max and sum-of-exp in registers and across the G chunks with
// log2(G) butterfly steps -- no LDS, no barrier, NI independent exps; in the backward pass the roles of
// the two indices swap (lane (i, g), terms j = g NI ..), so that the log-sum-exp over j is again
// in-lane + butterfly and a lane writes NI consecutive elements of xi.  The matrices and potentials of the
// next D steps are already on their way while a step computes (one wave per CU has nothing else to hide the
// latency of its loads behind).
template <typename T, int KP, int D>
__global__ __launch_bounds__(64) void logchain_lanes_kernel(const T* __restrict__ U,
                                                            const T* __restrict__ P, int64_t spb,
                                                            int64_t spt, int Tn, int K,
                                                            T* __restrict__ alpha,
                                                            T* __restrict__ logZ,
                                                            T* __restrict__ gamma,
                                                            T* __restrict__ xi) {
  constexpr int G = 64 / KP, NI = KP / G;
  const int64_t b = blockIdx.x;
  const int l = threadIdx.x, x = l % KP, g = l / KP;
  const bool xok = x < K, owner = xok && g == 0;
  const int xc = xok ? x : 0;
  const T NEG = lc_neg_inf<T>();
  const T* Ub = U + b * (int64_t)Tn * K;
  const T* Pb = P + b * spb;
  T* ab = alpha + b * (int64_t)Tn * K;
  T* gb = gamma + b * (int64_t)Tn * K;
  T* xb = xi + b * (int64_t)(Tn - 1) * K * K;
  bool cok[NI];                         // is the q-th index of this lane's chunk a state?
  int cc[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    cok[q] = g * NI + q < K;
    cc[q] = cok[q] ? g * NI + q : 0;
  }
  // log-sum-exp of the NI terms of every lane of a column, over the G chunks: the same value in all of them
  auto lse_chunks = [&](const T (&v)[NI]) {
    T m = v[0];
#pragma unroll
    for (int q = 1; q < NI; ++q) m = v[q] > m ? v[q] : m;
#pragma unroll
    for (int o = KP; o < 64; o <<= 1) {
      const T other = __shfl_xor(m, o, 64);
      m = other > m ? other : m;
    }
    T s = T(0);
#pragma unroll
    for (int q = 0; q < NI; ++q) s += v[q] == NEG ? T(0) : t_exp(v[q] - m);
#pragma unroll
    for (int o = KP; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
    return m == NEG ? NEG : m + t_log(s);
  };

  // ---- forward: lane (j = x, chunk g of i) ------------------------------------------------------
  T pre[D][NI], ub[D];
  auto load_fwd = [&](int t, T (&dst)[NI], T& u) {          // matrix t-1, potentials t (unconditional loads)
    const T* Pt = Pb + (int64_t)(t - 1) * spt;
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const T p = Pt[cc[q] * K + xc];
      dst[q] = (cok[q] && xok) ? p : NEG;
    }
    u = Ub[(int64_t)t * K + xc];
  };
  T a = xok ? Ub[x] : NEG;
  if (owner) ab[x] = a;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (1 + d < Tn) load_fwd(1 + d, pre[d], ub[d]);
  for (int t0 = 1; t0 < Tn; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 + d;
      if (t < Tn) {                                            // (wave-uniform)
        T v[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) v[q] = __shfl(a, cc[q], 64) + pre[d][q];
        const T u = ub[d];
        if (t + D < Tn) load_fwd(t + D, pre[d], ub[d]);
        const T r = lse_chunks(v);
        a = xok ? u + r : NEG;
        if (owner) ab[(int64_t)t * K + x] = a;
      }
    }
  }
  // log Z = LSE_j alpha_{T-1}[j]
  const T az = owner ? a : NEG;
  T mz = az;
  for (int o = 32; o > 0; o >>= 1) {
    const T other = __shfl_xor(mz, o, 64);
    mz = other > mz ? other : mz;
  }
  T sz = az != NEG ? t_exp(az - mz) : T(0);
  for (int o = 32; o > 0; o >>= 1) sz += __shfl_xor(sz, o, 64);
  const T lz = mz == NEG ? NEG : mz + t_log(sz);
  if (l == 0) logZ[b] = lz;
  // alpha was written by the owners and is read back by every chunk of the column
  __threadfence_block();
  __syncthreads();

  // ---- backward: lane (i = x, chunk g of j) -----------------------------------------------------
  T abuf[D];
  auto load_bwd = [&](int t, T (&dst)[NI], T& u, T& at) {   // matrix t, potentials t+1, alpha t
    const T* Pt = Pb + (int64_t)t * spt;
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const T p = Pt[xc * K + cc[q]];
      dst[q] = (cok[q] && xok) ? p : NEG;
    }
    u = Ub[(int64_t)(t + 1) * K + xc];
    at = ab[(int64_t)t * K + xc];
  };
  T beta = T(0);
  if (owner) gb[(int64_t)(Tn - 1) * K + x] = (lz == NEG) ? T(0) : t_exp(a + beta - lz);
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (Tn - 2 - d >= 0) load_bwd(Tn - 2 - d, pre[d], ub[d], abuf[d]);
  for (int t0 = Tn - 2; t0 >= 0; t0 -= D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 - d;
      if (t >= 0) {                                            // (wave-uniform)
        const T wv = xok ? ub[d] + beta : NEG;                 // w[x] = u_{t+1}[x] + beta_{t+1}[x]
        const T at = xok ? abuf[d] : NEG;
        T v[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) v[q] = pre[d][q] + __shfl(wv, cc[q], 64);     // P[x, j] + w[j]
        if (t - D >= 0) load_bwd(t - D, pre[d], ub[d], abuf[d]);
        if (xok) {
          T* xt = xb + (int64_t)t * K * K + (int64_t)x * K + g * NI;
#pragma unroll
          for (int q = 0; q < NI; ++q) {
            const T e = at + v[q] - lz;
            if (cok[q]) xt[q] = (lz == NEG || e == NEG || v[q] == NEG || at == NEG) ? T(0) : t_exp(e);
          }
        }
        beta = lse_chunks(v);
        if (!xok) beta = T(0);
        if (owner) gb[(int64_t)t * K + x] = (lz == NEG) ? T(0) : t_exp(at + beta - lz);
      }
    }
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
#define PA_LC_LANES(T_, KP_, D_)                                                                          \
  hipLaunchKernelGGL((pa::logchain_lanes_kernel<T_, KP_, D_>), dim3((unsigned)B), dim3(64), 0, s,       \
                     (const T_*)unary, (const T_*)pairwise, pair_stride_batch, pair_stride_step, (int)T, \
                     (int)K, (T_*)workspace, (T_*)log_z, (T_*)grad_unary, (T_*)grad_pairwise)
  if (K <= 16 && T > 1) {
    if (dtype == PA_F32) PA_LC_LANES(float, 16, 8); else PA_LC_LANES(double, 16, 8);
    return pa::check_launch("logchain_lanes_kernel");
  }
  if (K <= 32 && T > 1) {
    if (dtype == PA_F32) PA_LC_LANES(float, 32, 4); else PA_LC_LANES(double, 32, 2);
    return pa::check_launch("logchain_lanes_kernel");
  }
#undef PA_LC_LANES
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
