// lda.hip -- fused enumerated Categorical-Categorical mixture factor of examples/lda.py
// (model lines :53-71) under TraceEnum_ELBO: forward value and gradient in one pass.
//
// Reference path replaced (per step, B documents, Wd words/doc, T topics, V vocabulary):
//   word_topics enumerated to arange(T) on a fresh dim     pyro/poutine/enum_messenger.py:114-231
//   topic_words[word_topics] gather + Categorical.log_prob  -> [T,Wd,B] log-factor
//   log-space sum-product over the enum dim (max-shift, exp, einsum, log)
//                                                           pyro/ops/einsum/torch_log.py:14-55
//   plate products over words and documents (= sums)        pyro/ops/contract.py:79-160
//   and the autograd duals of all of it.
// i.e.  F = sum_{d,w} logsumexp_t( log_theta[d,t] + log_phi[t, words[w,d]] ).
// Here the int64 word ids are read once (coalesced along the document axis, the layout of
// examples/lda.py data: [Wd, B]), log_phi lives in LDS as [V][T] so one word needs T adjacent
// words of LDS, each thread owns a document (its T log_theta and g_theta values stay in
// registers), and d F / d log_phi is accumulated in an LDS histogram per workgroup
// (ds_add_f32 / ds_add_f64), written out as per-workgroup partial tables and summed by a
// finalize kernel.  HBM-bound on the 8 B/word ids.
#include "common.h"

namespace pa {

constexpr int LDA_THREADS = 256;

template <typename T> __device__ __forceinline__ T lexp(T x);
template <> __device__ __forceinline__ float lexp(float x) { return expf(x); }
template <> __device__ __forceinline__ double lexp(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T llog(T x);
template <> __device__ __forceinline__ float llog(float x) { return logf(x); }
template <> __device__ __forceinline__ double llog(double x) { return log(x); }

template <typename T, int TMAX>
__global__ __launch_bounds__(LDA_THREADS) void lda_factor_kernel(
    const int64_t* __restrict__ words, const T* __restrict__ log_theta,
    const T* __restrict__ log_phi, int64_t Wd, int64_t B, int Tn, int V, T* __restrict__ out_doc,
    T* __restrict__ g_theta, T* __restrict__ part_hist, int* __restrict__ bad_index) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lda_smem[];
  T* phi_s = reinterpret_cast<T*>(lda_smem);  // [V][TMAX]
  T* hist_s = phi_s + (size_t)V * TMAX;       // [V][TMAX]
  for (int i = threadIdx.x; i < V * TMAX; i += LDA_THREADS) {
    const int vv = i / TMAX, tt = i % TMAX;
    phi_s[i] = tt < Tn ? log_phi[(int64_t)tt * V + vv] : T(0);
    hist_s[i] = T(0);
  }
  __syncthreads();

  for (int64_t d = (int64_t)blockIdx.x * LDA_THREADS + threadIdx.x; d < B;
       d += (int64_t)gridDim.x * LDA_THREADS) {
    T th[TMAX], gth[TMAX];
#pragma unroll
    for (int tt = 0; tt < TMAX; ++tt) {
      th[tt] = tt < Tn ? log_theta[d * Tn + tt] : T(0);
      gth[tt] = T(0);
    }
    T acc = T(0);
    for (int64_t wi = 0; wi < Wd; ++wi) {
      int64_t v = words[wi * B + d];
      if (v < 0 || v >= V) {  // Categorical support violation: flag, keep memory safe
        *bad_index = 1;
        v = 0;
      }
      const T* ph = phi_s + (size_t)v * TMAX;
      T a[TMAX];
      T mx = th[0] + ph[0];
      a[0] = mx;
#pragma unroll
      for (int tt = 1; tt < TMAX; ++tt) {
        a[tt] = th[tt] + ph[tt];
        if (tt < Tn && a[tt] > mx) mx = a[tt];
      }
      // max-shifted logsumexp, torch_log.py:25-45 (an all -inf column yields -inf, no NaN)
      const bool finite = mx > -__builtin_huge_val();
      const T shift = finite ? mx : T(0);
      T s = T(0);
#pragma unroll
      for (int tt = 0; tt < TMAX; ++tt) {
        a[tt] = tt < Tn ? lexp(a[tt] - shift) : T(0);
        s += a[tt];
      }
      acc += llog(s) + shift;
      const T inv = s > T(0) ? T(1) / s : T(0);
      T* hs = hist_s + (size_t)v * TMAX;
#pragma unroll
      for (int tt = 0; tt < TMAX; ++tt) {
        if (tt < Tn) {
          const T post = a[tt] * inv;  // posterior responsibility of topic tt for this word
          gth[tt] += post;
          atomicAdd(hs + tt, post);
        }
      }
    }
    out_doc[d] = acc;
#pragma unroll
    for (int tt = 0; tt < TMAX; ++tt)
      if (tt < Tn) g_theta[d * Tn + tt] = gth[tt];
  }
  __syncthreads();
  T* ph_out = part_hist + (size_t)blockIdx.x * Tn * V;  // [T][V]
  for (int i = threadIdx.x; i < Tn * V; i += LDA_THREADS) {
    const int tt = i / V, vv = i % V;
    ph_out[i] = hist_s[vv * TMAX + tt];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void lda_finalize_kernel(const T* __restrict__ part_hist,
                                                           int nblocks, int64_t n,
                                                           T* __restrict__ g_phi) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double t = 0.0;
  for (int b = 0; b < nblocks; ++b) t += (double)part_hist[(int64_t)b * n + i];
  g_phi[i] = (T)t;
}

static int lda_tmax(int64_t T) { return T <= 8 ? 8 : (T <= 16 ? 16 : (T <= 32 ? 32 : 64)); }

static int lda_nblocks(int64_t B) {
  int64_t want = (B + LDA_THREADS - 1) / LDA_THREADS;
  int64_t cap = (int64_t)cu_count() * 2;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

template <typename T, int TMAX>
static int lda_launch(const int64_t* words, const T* log_theta, const T* log_phi, int64_t Wd,
                      int64_t B, int Tn, int V, T* out_doc, T* g_theta, T* g_phi, void* ws,
                      hipStream_t s) {
  const int nb = lda_nblocks(B);
  const size_t lds = 2 * (size_t)V * TMAX * sizeof(T);
  auto k = lda_factor_kernel<T, TMAX>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess)
      return fail(PA_ERR_LAUNCH, "lda_factor: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  int* bad = (int*)ws;
  T* part = (T*)((char*)ws + 256);
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), s);
  if (e != hipSuccess) return fail(PA_ERR_LAUNCH, "lda_factor: memset: %s", hipGetErrorString(e));
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_LDA, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  hipLaunchKernelGGL(k, dim3(nb), dim3(LDA_THREADS), lds, s, words, log_theta, log_phi, Wd, B, Tn,
                     V, out_doc, g_theta, part, bad);
  if (br) (void)hipEventRecord(ev1, s);
  int rc = check_launch("lda_factor_kernel");
  if (rc != PA_OK) return rc;
  const int64_t n = (int64_t)Tn * V;
  hipLaunchKernelGGL((lda_finalize_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     part, nb, n, g_phi);
  return check_launch("lda_finalize_kernel");
}

}  // namespace pa

extern "C" {

size_t pa_lda_factor_workspace(int dtype, int64_t B, int64_t T, int64_t V) {
  if (B < 0 || T < 1 || V < 1) return 0;
  return 256 + (size_t)pa::lda_nblocks(B) * (size_t)T * (size_t)V * (dtype == PA_F32 ? 4 : 8);
}

int pa_lda_factor_fwd_bwd(int dtype, const int64_t* words, const void* log_theta,
                          const void* log_phi, int64_t Wd, int64_t B, int64_t T, int64_t V,
                          void* out_doc, void* g_theta, void* g_phi, void* workspace,
                          size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "lda_factor: bad dtype %d", dtype);
  PA_REQUIRE(Wd >= 0 && B >= 0 && T >= 1 && V >= 1, "lda_factor: bad shape");
  if (T > 64) return pa::fail(PA_ERR_UNSUPPORTED, "lda_factor: T=%lld > 64", (long long)T);
  const int tmax = pa::lda_tmax(T);
  const size_t esz = dtype == PA_F32 ? 4 : 8;
  if (2 * (size_t)V * tmax * esz > 152 * 1024)
    return pa::fail(PA_ERR_UNSUPPORTED, "lda_factor: V*T table (%lld x %d) does not fit in LDS",
                    (long long)V, tmax);
  PA_REQUIRE(log_phi && g_phi && workspace, "lda_factor: NULL pointer");
  PA_REQUIRE(B == 0 || (log_theta && out_doc && g_theta), "lda_factor: NULL pointer");
  PA_REQUIRE(B == 0 || Wd == 0 || words, "lda_factor: NULL words");
  PA_REQUIRE(workspace_bytes >= pa_lda_factor_workspace(dtype, B, T, V),
             "lda_factor: workspace too small");
  hipStream_t s = pa::as_stream(stream);
#define PA_LDA_CASE(TM)                                                                           \
  if (tmax == TM) {                                                                               \
    if (dtype == PA_F32)                                                                          \
      return pa::lda_launch<float, TM>(words, (const float*)log_theta, (const float*)log_phi, Wd, \
                                       B, (int)T, (int)V, (float*)out_doc, (float*)g_theta,       \
                                       (float*)g_phi, workspace, s);                              \
    return pa::lda_launch<double, TM>(words, (const double*)log_theta, (const double*)log_phi, Wd, \
                                      B, (int)T, (int)V, (double*)out_doc, (double*)g_theta,      \
                                      (double*)g_phi, workspace, s);                              \
  }
  PA_LDA_CASE(8)
  PA_LDA_CASE(16)
  PA_LDA_CASE(32)
  PA_LDA_CASE(64)
#undef PA_LDA_CASE
  return pa::fail(PA_ERR_UNSUPPORTED, "lda_factor: unreachable");
}

}  // extern "C"
