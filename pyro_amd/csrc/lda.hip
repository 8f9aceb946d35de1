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
//
// gfx950 mapping (HBM-bound on the 8-byte word ids: 8.5 B per (word, document)):
//   * a workgroup of NW waves walks tiles of 64 documents; lane = document (the int64 ids are
//     read coalesced along the document axis, the layout of examples/lda.py data: [Wd, B]),
//     wave j takes the words j*WPW .. of every document of the tile, so that a 100 k-document
//     corpus is 1563 tiles x NW wave-tasks instead of 1563 serial threads-over-64-words;
//   * the ids of the NEXT round / tile are loaded into registers before the current ones are
//     processed (the LDS atomics below order memory: the compiler would not hoist them);
//   * log_phi lives in LDS as [V][TMAX] (one word's T values = consecutive 16-byte reads),
//     d F / d log_phi is accumulated in an LDS histogram laid out [T][V] -- for a fixed topic the
//     bank is v mod 32, i.e. as random as the words (the [V][T] layout of round 1 put a whole wave on
//     4 banks: 16-way conflicts on every ds_add_f32);
//   * per document the NW partial values (out_doc, g_theta[T]) meet in LDS and are summed in a
//     FIXED order; exp2 / log2 on the hardware units for f32;
//   * per-workgroup partial histograms are summed by a finalize kernel (fp64, fixed order).
#include "common.h"

namespace pa {

template <typename T> __device__ __forceinline__ T lexp(T x);
template <> __device__ __forceinline__ float lexp(float x) {
  return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}
template <> __device__ __forceinline__ double lexp(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T llog(T x);
template <> __device__ __forceinline__ float llog(float x) {
  return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
}
template <> __device__ __forceinline__ double llog(double x) { return log(x); }

constexpr int LDA_WR = 4;     // words per round and wave (ids held in registers one round ahead)

template <typename T, int TMAX, int NW>
__global__ __launch_bounds__(64 * NW) void lda_factor_kernel(
    const int64_t* __restrict__ words, const T* __restrict__ log_theta,
    const T* __restrict__ log_phi, int64_t Wd, int64_t B, int Tn, int V, int64_t ntiles,
    T* __restrict__ out_doc, T* __restrict__ g_theta, T* __restrict__ part_hist,
    int* __restrict__ bad_index) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lda_smem[];
  T* phi_s = reinterpret_cast<T*>(lda_smem);          // [V][TMAX]
  T* hist_s = phi_s + (size_t)V * TMAX;               // [TMAX][V]
  T* red_s = hist_s + (size_t)V * TMAX;               // [NW][TMAX + 1][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < V * TMAX; i += 64 * NW) {
    const int vv = i / TMAX, tt = i % TMAX;
    // padded topics never win the max and add exp(-inf) = 0
    phi_s[i] = tt < Tn ? log_phi[(int64_t)tt * V + vv] : -__builtin_huge_val();
    hist_s[i] = T(0);
  }
  __syncthreads();

  const int64_t wpw = (Wd + NW - 1) / NW;             // words per wave
  const int64_t w0 = (int64_t)wave * wpw, w1 = (w0 + wpw < Wd) ? w0 + wpw : Wd;
  const int64_t rounds = w1 > w0 ? (w1 - w0 + LDA_WR - 1) / LDA_WR : 0;

  auto load_ids = [&](int64_t tile, int64_t r, int64_t (&v)[LDA_WR]) {
    const int64_t d = tile * 64 + lane;
#pragma unroll
    for (int j = 0; j < LDA_WR; ++j) {
      const int64_t wi = w0 + r * LDA_WR + j;
      v[j] = (wi < w1 && d < B) ? words[wi * B + d] : 0;
    }
  };

  int64_t cur[LDA_WR], nxt[LDA_WR];
  int64_t tile = blockIdx.x;
  if (tile < ntiles && rounds > 0) load_ids(tile, 0, cur);
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t d = tile * 64 + lane;
    const bool dv = d < B;
    T th[TMAX], gth[TMAX];
#pragma unroll
    for (int tt = 0; tt < TMAX; ++tt) {
      th[tt] = (dv && tt < Tn) ? log_theta[d * Tn + tt] : T(0);
      gth[tt] = T(0);
    }
    T acc = T(0);
    for (int64_t r = 0; r < rounds; ++r) {
      // the ids of the next round (or of the next tile's first round) travel while this one runs
      if (r + 1 < rounds) load_ids(tile, r + 1, nxt);
      else if (tile + gridDim.x < ntiles) load_ids(tile + gridDim.x, 0, nxt);
#pragma unroll
      for (int j = 0; j < LDA_WR; ++j) {
        const int64_t wi = w0 + r * LDA_WR + j;
        if (wi >= w1) break;                                    // wave-uniform
        int64_t v = cur[j];
        if (v < 0 || v >= V) {  // Categorical support violation: flag, keep memory safe
          if (dv) *bad_index = 1;
          v = 0;
        }
        const T* ph = phi_s + (size_t)v * TMAX;
        T a[TMAX];
        T mx = -__builtin_huge_val();
#pragma unroll
        for (int tt = 0; tt < TMAX; ++tt) {
          a[tt] = th[tt] + ph[tt];
          mx = a[tt] > mx ? a[tt] : mx;
        }
        // max-shifted logsumexp, torch_log.py:25-45 (an all -inf column yields -inf, no NaN)
        const T shift = mx > -__builtin_huge_val() ? mx : T(0);
        T s = T(0);
#pragma unroll
        for (int tt = 0; tt < TMAX; ++tt) {
          a[tt] = lexp(a[tt] - shift);
          s += a[tt];
        }
        const T inv = s > T(0) ? T(1) / s : T(0);
        if (dv) {
          acc += llog(s) + shift;
#pragma unroll
          for (int tt = 0; tt < TMAX; ++tt) {
            if (tt < Tn) {
              const T post = a[tt] * inv;  // posterior responsibility of topic tt for this word
              gth[tt] += post;
              atomicAdd(hist_s + (size_t)tt * V + v, post);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < LDA_WR; ++j) cur[j] = nxt[j];
    }
    // ---- the NW partial values of every document meet in LDS, summed in wave order -------------
    T* mine = red_s + (size_t)wave * (TMAX + 1) * 64;
    mine[lane] = acc;
#pragma unroll
    for (int tt = 0; tt < TMAX; ++tt) mine[(tt + 1) * 64 + lane] = gth[tt];
    __syncthreads();
    const int nd = (int)((B - tile * 64 < 64) ? (B - tile * 64) : 64);       // documents in the tile
    for (int i = threadIdx.x; i < nd * (Tn + 1); i += 64 * NW) {
      // i < nd: out_doc of document i; else g_theta in its memory order (document-major)
      const int k = i < nd ? 0 : 1 + (i - nd) % Tn;
      const int l = i < nd ? i : (i - nd) / Tn;
      T t = T(0);
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) t += red_s[((size_t)wv * (TMAX + 1) + k) * 64 + l];
      if (i < nd) out_doc[tile * 64 + i] = t;
      else g_theta[tile * 64 * Tn + (i - nd)] = t;
    }
    __syncthreads();
  }
  T* ph_out = part_hist + (size_t)blockIdx.x * Tn * V;  // [T][V]
  for (int i = threadIdx.x; i < Tn * V; i += 64 * NW) ph_out[i] = hist_s[i];
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

// waves per workgroup: as many as the LDS left by the two V x T tables admits (the per-document
// reduction area is NW x (T+1) x 64 values), at most 16
static int lda_waves(int64_t V, int tmax, size_t esz) {
  const size_t budget = 160 * 1024;
  const size_t tables = 2 * (size_t)V * tmax * esz;
  for (int nw = 16; nw >= 1; nw >>= 1)
    if (tables + (size_t)nw * (tmax + 1) * 64 * esz <= budget) return nw;
  return 0;
}

static int lda_nblocks(int64_t B) {
  int64_t want = (B + 63) / 64;
  int64_t cap = (int64_t)cu_count();          // one workgroup (up to 16 waves) per CU
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

template <typename T, int TMAX, int NW>
static int lda_launch_nw(const int64_t* words, const T* log_theta, const T* log_phi, int64_t Wd,
                         int64_t B, int Tn, int V, T* out_doc, T* g_theta, T* g_phi, void* ws,
                         hipStream_t s) {
  const int nb = lda_nblocks(B);
  const size_t lds = (2 * (size_t)V * TMAX + (size_t)NW * (TMAX + 1) * 64) * sizeof(T);
  auto k = lda_factor_kernel<T, TMAX, NW>;
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
  hipLaunchKernelGGL(k, dim3(nb), dim3(64 * NW), lds, s, words, log_theta, log_phi, Wd, B, Tn, V,
                     (B + 63) / 64, out_doc, g_theta, part, bad);
  if (br) (void)hipEventRecord(ev1, s);
  int rc = check_launch("lda_factor_kernel");
  if (rc != PA_OK) return rc;
  const int64_t n = (int64_t)Tn * V;
  hipLaunchKernelGGL((lda_finalize_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     part, nb, n, g_phi);
  return check_launch("lda_finalize_kernel");
}

template <typename T, int TMAX>
static int lda_launch(const int64_t* words, const T* log_theta, const T* log_phi, int64_t Wd,
                      int64_t B, int Tn, int V, T* out_doc, T* g_theta, T* g_phi, void* ws,
                      hipStream_t s) {
  switch (lda_waves(V, TMAX, sizeof(T))) {
#define PA_LDA_NW(N_)                                                                           \
  case N_:                                                                                      \
    return lda_launch_nw<T, TMAX, N_>(words, log_theta, log_phi, Wd, B, Tn, V, out_doc, g_theta, \
                                      g_phi, ws, s);
    PA_LDA_NW(16)
    PA_LDA_NW(8)
    PA_LDA_NW(4)
    PA_LDA_NW(2)
    PA_LDA_NW(1)
#undef PA_LDA_NW
  }
  return fail(PA_ERR_UNSUPPORTED, "lda_factor: V*T table does not fit in LDS");
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
  if (pa::lda_waves(V, tmax, esz) == 0)
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
