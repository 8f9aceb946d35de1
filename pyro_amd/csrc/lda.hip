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

// HIST = false: the document-major half of the INDEXED path (no histogram, no atomics): out_doc and
// g_theta only; d F / d log_phi comes from lda_vocab_kernel below.
template <typename T, int TMAX, int NW, bool HIST>
__global__ __launch_bounds__(64 * NW) void lda_factor_kernel(
    const int64_t* __restrict__ words, const T* __restrict__ log_theta,
    const T* __restrict__ log_phi, int64_t Wd, int64_t B, int Tn, int V, int64_t ntiles,
    T* __restrict__ out_doc, T* __restrict__ g_theta, T* __restrict__ part_hist,
    int* __restrict__ bad_index) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lda_smem[];
  T* phi_s = reinterpret_cast<T*>(lda_smem);          // [V][TMAX]
  T* hist_s = phi_s + (size_t)V * TMAX;               // [TMAX][V]   (HIST only)
  T* red_s = hist_s + (HIST ? (size_t)V * TMAX : 0);  // [NW][TMAX + 1][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // log_phi is [T][V] in memory: read it in that order (coalesced), transpose on the LDS side
  for (int i = threadIdx.x; i < V * TMAX; i += 64 * NW) {
    const int tt = i / V, vv = i - tt * V;
    // padded topics never win the max and add exp(-inf) = 0
    phi_s[(size_t)vv * TMAX + tt] = tt < Tn ? log_phi[i] : -__builtin_huge_val();
    if (HIST) hist_s[i] = T(0);
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
              if (HIST) atomicAdd(hist_s + (size_t)tt * V + v, post);
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
  if (HIST) {
    T* ph_out = part_hist + (size_t)blockIdx.x * Tn * V;  // [T][V]
    for (int i = threadIdx.x; i < Tn * V; i += 64 * NW) ph_out[i] = hist_s[i];
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
  auto k = lda_factor_kernel<T, TMAX, NW, true>;
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


// =================================================================================================
// INDEXED path: the corpus never changes between ELBO-gradient steps, so the scatter that made the
// kernel above LDS-atomic-bound (d F / d log_phi[t, v] = sum over the (word, document) pairs that
// hold word v: 8 ds_add_f32 per pair, ~4 cycles per lane each -- 335 us for 6.4 M pairs where the
// 51 MB of ids stream in 7 us) is turned into a GATHER through an inverted index built once:
//   docs[off[v] .. off[v+1])  = the documents d of all pairs (w, d) with words[w, d] == v, in
//                               ascending pair order w * B + d (a stable counting sort);
//   tasks: every word's list cut into segments of <= LDA_SEG pairs (one workgroup each), so that a
//                               frequent word does not serialise on one workgroup.
// Per step:  lda_factor_kernel<HIST = false>  (document-major: out_doc, g_theta; no atomics)
//            lda_vocab_kernel                 (word-major: lane = pair of the word's list, gathers
//                                              the document's log_theta row, posterior, register
//                                              accumulation, fixed-order reduction)
//            lda_vocab_finalize_kernel        (sums a word's segments: a wave per word, fixed tree)
// Everything is summed in a fixed order: g_phi is bitwise reproducible (the atomic path is not).
//
// Index image (int32 words; header first):
//   hdr[0]=magic hdr[1]=Wd hdr[2]=B hdr[3]=V hdr[4]=ntasks hdr[5]=task capacity hdr[6]=n pairs
//   off[V+1] | first_task[V+1] | task_v[cap] | task_start[cap] | task_len[cap] | docs[n]
// =================================================================================================
constexpr int LDA_SEG = 2048;         // pairs per word-major task
constexpr int LDA_IDX_MAGIC = 0x4c444131;
constexpr int LDA_IDX_CHUNKS = 1024;  // chunks of the pair range in the counting sort
constexpr int LDA_HDR = 8;

struct LdaIndexLayout {
  int64_t n, cap;
  size_t off, first_task, task_v, task_start, task_len, docs, total;   // in int32 units
};
static bool lda_index_layout(int64_t Wd, int64_t B, int64_t V, LdaIndexLayout* L) {
  if (Wd < 0 || B < 0 || V < 1 || V > (1 << 20)) return false;
  if (Wd > 0 && B > ((int64_t)1 << 31) / (Wd > 0 ? Wd : 1) - 1) return false;   // int32 pair ranks
  L->n = Wd * B;
  L->cap = L->n / LDA_SEG + V + 1;
  size_t p = LDA_HDR;
  L->off = p; p += (size_t)V + 1;
  L->first_task = p; p += (size_t)V + 1;
  L->task_v = p; p += (size_t)L->cap;
  L->task_start = p; p += (size_t)L->cap;
  L->task_len = p; p += (size_t)L->cap;
  L->docs = p; p += (size_t)L->n;
  L->total = p;
  return true;
}
static int64_t lda_chunk_len(int64_t n) {
  int64_t ch = (n + LDA_IDX_CHUNKS - 1) / LDA_IDX_CHUNKS;
  return ((ch + 63) / 64) * 64;
}

// one wave per chunk of the pair range: per-chunk word counts (integer LDS atomics: exact)
__global__ __launch_bounds__(64) void lda_index_count_kernel(const int64_t* __restrict__ words,
                                                             int64_t n, int64_t chunk, int V,
                                                             int* __restrict__ counts) {
  extern __shared__ int lda_cnt[];
  for (int v = threadIdx.x; v < V; v += 64) lda_cnt[v] = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 64) {
    int64_t v = words[i];
    v = (v < 0 || v >= V) ? 0 : v;       // support violation: flagged by the step kernel
    atomicAdd(&lda_cnt[(int)v], 1);
  }
  __syncthreads();
  for (int v = threadIdx.x; v < V; v += 64) counts[(int64_t)blockIdx.x * V + v] = lda_cnt[v];
}

// counts[c][v] -> exclusive prefix over chunks (in place); off[], first_task[], the task table
__global__ __launch_bounds__(1024) void lda_index_scan_kernel(int* __restrict__ counts, int V,
                                                              int* __restrict__ img,
                                                              LdaIndexLayout L, int64_t Wd,
                                                              int64_t B) {
  int* off = img + L.off;
  int* first_task = img + L.first_task;
  for (int v = threadIdx.x; v < V; v += 1024) {
    int run = 0;
    for (int c = 0; c < LDA_IDX_CHUNKS; ++c) {
      const int k = counts[(int64_t)c * V + v];
      counts[(int64_t)c * V + v] = run;
      run += k;
    }
    off[v + 1] = run;            // the word's total for now
  }
  __syncthreads();
  if (threadIdx.x == 0) {        // V is a vocabulary (thousands): a serial scan, once per corpus
    int run = 0, trun = 0;
    off[0] = 0;
    first_task[0] = 0;
    for (int v = 0; v < V; ++v) {
      const int k = off[v + 1];
      run += k;
      off[v + 1] = run;
      trun += (k + LDA_SEG - 1) / LDA_SEG;
      first_task[v + 1] = trun;
    }
    img[0] = LDA_IDX_MAGIC; img[1] = (int)Wd; img[2] = (int)B; img[3] = V;
    img[4] = trun; img[5] = (int)L.cap; img[6] = (int)L.n; img[7] = 0;
  }
  __syncthreads();
  for (int v = threadIdx.x; v < V; v += 1024) {
    const int a = off[v], b = off[v + 1];
    int t = first_task[v];
    for (int st = a; st < b; st += LDA_SEG, ++t) {
      img[L.task_v + t] = v;
      img[L.task_start + t] = st;
      img[L.task_len + t] = (b - st < LDA_SEG) ? b - st : LDA_SEG;
    }
  }
}

// one wave per chunk again: every pair goes to its word's list at (pairs of the word in earlier
// chunks) + (pairs of the word earlier in this chunk) -- ascending pair order, no run-to-run change
__global__ __launch_bounds__(64) void lda_index_fill_kernel(const int64_t* __restrict__ words,
                                                            int64_t n, int64_t chunk, int V,
                                                            int64_t B, const int* __restrict__ counts,
                                                            int* __restrict__ img, LdaIndexLayout L) {
  extern __shared__ int lda_cur[];
  const int* off = img + L.off;
  int* docs = img + L.docs;
  for (int v = threadIdx.x; v < V; v += 64)
    lda_cur[v] = off[v] + counts[(int64_t)blockIdx.x * V + v];
  __syncthreads();
  const int lane = threadIdx.x;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
  for (int64_t ib = i0; ib < i1; ib += 64) {
    const int64_t i = ib + lane;
    const bool ok = i < i1;
    int64_t v64 = ok ? words[i] : 0;
    const int v = (int)((v64 < 0 || v64 >= V) ? 0 : v64);
    int rank = 0, total = 0;
    uint64_t todo = __ballot(ok);
    while (todo) {                                   // one round per distinct word of the 64
      const int leader = __ffsll((unsigned long long)todo) - 1;
      const int vl = __shfl(v, leader);
      const uint64_t m = __ballot(ok && v == vl) & todo;
      if (ok && v == vl) {
        rank = __popcll(m & lt);
        total = __popcll(m);
      }
      todo &= ~m;
    }
    int base = 0;
    if (ok) base = lda_cur[v];
    if (ok) docs[base + rank] = (int)(i % B);
    __builtin_amdgcn_s_waitcnt(0);                   // every lane has read its cursor
    if (ok && rank == total - 1) lda_cur[v] = base + total;
    __builtin_amdgcn_s_waitcnt(0);
  }
}

// word-major half of the step: workgroup = one task (a segment of one word's document list)
template <typename T, int TMAX>
__global__ __launch_bounds__(256) void lda_vocab_kernel(const int* __restrict__ img,
                                                        LdaIndexLayout L,
                                                        const T* __restrict__ log_theta,
                                                        const T* __restrict__ log_phi, int Tn,
                                                        int V, T* __restrict__ part) {
  __shared__ double red[4 * TMAX];
  const int task = blockIdx.x;
  if (task >= img[4]) return;
  const int v = img[L.task_v + task], start = img[L.task_start + task];
  const int len = img[L.task_len + task];
  const int* docs = img + L.docs + start;
  T ph[TMAX], acc[TMAX];
#pragma unroll
  for (int tt = 0; tt < TMAX; ++tt) {
    ph[tt] = tt < Tn ? log_phi[(int64_t)tt * V + v] : -__builtin_huge_val();
    acc[tt] = T(0);
  }
  constexpr int U = 4;       // gathers in flight per thread
  for (int ib = threadIdx.x; ib < len; ib += 256 * U) {
    int d[U];
    T th[U][TMAX];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = ib + u * 256;
      d[u] = i < len ? docs[i] : -1;
    }
    if (Tn == TMAX) {
      // full rows (the usual T = 8 / 16 / 32 / 64): 16-byte loads -- the gather is bound by the
      // number of scattered load instructions (every lane its own cache line), not by bytes
      constexpr int PER = 16 / sizeof(T), NV = TMAX / PER;
      struct alignas(16) Vec { T x[PER]; };
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const Vec* row = reinterpret_cast<const Vec*>(log_theta + (int64_t)(d[u] < 0 ? 0 : d[u]) * TMAX);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const Vec q = row[j];
#pragma unroll
          for (int e = 0; e < PER; ++e) th[u][j * PER + e] = q.x[e];
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const T* row = log_theta + (int64_t)(d[u] < 0 ? 0 : d[u]) * Tn;
#pragma unroll
        for (int tt = 0; tt < TMAX; ++tt) th[u][tt] = tt < Tn ? row[tt] : T(0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      T a[TMAX];
      T mx = -__builtin_huge_val();
#pragma unroll
      for (int tt = 0; tt < TMAX; ++tt) {
        a[tt] = th[u][tt] + ph[tt];
        mx = a[tt] > mx ? a[tt] : mx;
      }
      const T shift = mx > -__builtin_huge_val() ? mx : T(0);
      T s = T(0);
#pragma unroll
      for (int tt = 0; tt < TMAX; ++tt) {
        a[tt] = lexp(a[tt] - shift);
        s += a[tt];
      }
      const T inv = (d[u] >= 0 && s > T(0)) ? T(1) / s : T(0);
#pragma unroll
      for (int tt = 0; tt < TMAX; ++tt) acc[tt] += a[tt] * inv;
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int tt = 0; tt < TMAX; ++tt) {
    const double w = wave_sum((double)acc[tt]);
    if (lane == 0) red[wave * TMAX + tt] = w;
  }
  __syncthreads();
  if ((int)threadIdx.x < Tn) {
    const int tt = threadIdx.x;
    part[(int64_t)task * Tn + tt] =
        (T)(red[tt] + red[TMAX + tt] + red[2 * TMAX + tt] + red[3 * TMAX + tt]);
  }
}

// One WAVE per word: lane = (task slot, topic) with TP = Tn rounded up to a power of two topics and 64 / TP slots; a
// slot adds every (64 / TP)-th segment of the word (the wave reads 64 consecutive values of `part` per step, four
// steps in flight), the slots are added in a fixed tree.  (Until round 6 a thread per (topic, word) walked the
// word's segments one by one: the most frequent word's ~200 dependent loads set the launch's 39 us.)
template <typename T>
__global__ __launch_bounds__(256) void lda_vocab_finalize_kernel(const int* __restrict__ img,
                                                                 LdaIndexLayout L,
                                                                 const T* __restrict__ part,
                                                                 int Tn, int TP, int V,
                                                                 T* __restrict__ g_phi) {
  const int v = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (v >= V) return;
  const int lane = threadIdx.x & 63;
  const int tt = lane & (TP - 1), slot = lane / TP, nslots = 64 / TP;
  const int t0 = img[L.first_task + v], t1 = img[L.first_task + v + 1];
  const bool tok = tt < Tn;
  double acc = 0.0;
  int t = t0 + slot;
  for (; t + 3 * nslots < t1; t += 4 * nslots) {
    T q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = tok ? part[(int64_t)(t + u * nslots) * Tn + tt] : T(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += (double)q[u];
  }
  for (; t < t1; t += nslots) acc += tok ? (double)part[(int64_t)t * Tn + tt] : 0.0;
  for (int o = TP; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
  if (slot == 0 && tok) g_phi[(int64_t)tt * V + v] = (T)acc;
}

// waves per workgroup of the document-major half (its LDS: one V x T table + the reduction area)
static int lda_doc_waves(int64_t V, int tmax, size_t esz) {
  const size_t budget = 64 * 1024;      // <= 64 KB: two workgroups per CU
  const size_t table = (size_t)V * tmax * esz;
  for (int nw = 16; nw >= 1; nw >>= 1)
    if (table + (size_t)nw * (tmax + 1) * 64 * esz <= budget) return nw;
  for (int nw = 16; nw >= 1; nw >>= 1)
    if (table + (size_t)nw * (tmax + 1) * 64 * esz <= 160 * 1024) return nw;
  return 0;
}

template <typename T, int TMAX, int NW>
static int lda_indexed_launch_nw(const int64_t* words, const int* img, const LdaIndexLayout& L,
                                 const T* log_theta, const T* log_phi, int64_t Wd, int64_t B,
                                 int Tn, int V, T* out_doc, T* g_theta, T* g_phi, void* ws,
                                 hipStream_t s) {
  const int64_t tiles = (B + 63) / 64;
  int64_t nb = tiles < 2 * (int64_t)cu_count() ? tiles : 2 * (int64_t)cu_count();
  if (nb < 1) nb = 1;
  const size_t lds = ((size_t)V * TMAX + (size_t)NW * (TMAX + 1) * 64) * sizeof(T);
  auto k = lda_factor_kernel<T, TMAX, NW, false>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess)
      return fail(PA_ERR_LAUNCH, "lda_factor_indexed: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  int* bad = (int*)ws;
  T* part = (T*)((char*)ws + 256);
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), s);
  if (e != hipSuccess) return fail(PA_ERR_LAUNCH, "lda_factor_indexed: memset: %s", hipGetErrorString(e));
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_LDA, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(64 * NW), lds, s, words, log_theta, log_phi, Wd, B,
                     Tn, V, tiles, out_doc, g_theta, (T*)nullptr, bad);
  int rc = check_launch("lda_factor_kernel<indexed>");
  if (rc != PA_OK) return rc;
  hipLaunchKernelGGL((lda_vocab_kernel<T, TMAX>), dim3((unsigned)L.cap), dim3(256), 0, s, img, L,
                     log_theta, log_phi, Tn, V, part);
  if (br) (void)hipEventRecord(ev1, s);
  rc = check_launch("lda_vocab_kernel");
  if (rc != PA_OK) return rc;
  int TP = 1;
  while (TP < Tn) TP <<= 1;
  hipLaunchKernelGGL((lda_vocab_finalize_kernel<T>), dim3((unsigned)((V + 3) / 4)), dim3(256),
                     0, s, img, L, part, Tn, TP, V, g_phi);
  return check_launch("lda_vocab_finalize_kernel");
}

template <typename T, int TMAX>
static int lda_indexed_launch(const int64_t* words, const int* img, const LdaIndexLayout& L,
                              const T* log_theta, const T* log_phi, int64_t Wd, int64_t B, int Tn,
                              int V, T* out_doc, T* g_theta, T* g_phi, void* ws, hipStream_t s) {
  switch (lda_doc_waves(V, TMAX, sizeof(T))) {
#define PA_LDA_NW(N_)                                                                         \
  case N_:                                                                                    \
    return lda_indexed_launch_nw<T, TMAX, N_>(words, img, L, log_theta, log_phi, Wd, B, Tn, V, \
                                              out_doc, g_theta, g_phi, ws, s);
    PA_LDA_NW(16)
    PA_LDA_NW(8)
    PA_LDA_NW(4)
    PA_LDA_NW(2)
    PA_LDA_NW(1)
#undef PA_LDA_NW
  }
  return fail(PA_ERR_UNSUPPORTED, "lda_factor_indexed: V*T table does not fit in LDS");
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

size_t pa_lda_index_bytes(int64_t Wd, int64_t B, int64_t V) {
  pa::LdaIndexLayout L;
  if (!pa::lda_index_layout(Wd, B, V, &L)) return 0;
  if ((size_t)V * sizeof(int) > 60 * 1024) return 0;      // per-chunk LDS counters of the build
  return L.total * sizeof(int);
}

size_t pa_lda_index_workspace(int64_t Wd, int64_t B, int64_t V) {
  if (pa_lda_index_bytes(Wd, B, V) == 0) return 0;
  return (size_t)pa::LDA_IDX_CHUNKS * (size_t)V * sizeof(int);
}

int pa_lda_build_index(const int64_t* words, int64_t Wd, int64_t B, int64_t V, void* index,
                       size_t index_bytes, void* workspace, size_t workspace_bytes,
                       pa_stream_t stream) {
  pa::LdaIndexLayout L;
  const size_t need = pa_lda_index_bytes(Wd, B, V);
  if (need == 0 || !pa::lda_index_layout(Wd, B, V, &L))
    return pa::fail(PA_ERR_UNSUPPORTED, "lda_build_index: no index for Wd=%lld B=%lld V=%lld",
                    (long long)Wd, (long long)B, (long long)V);
  PA_REQUIRE(index && index_bytes >= need, "lda_build_index: index buffer too small");
  PA_REQUIRE(workspace && workspace_bytes >= pa_lda_index_workspace(Wd, B, V),
             "lda_build_index: workspace too small");
  PA_REQUIRE(L.n == 0 || words, "lda_build_index: NULL words");
  hipStream_t s = pa::as_stream(stream);
  const int64_t chunk = pa::lda_chunk_len(L.n);
  int* counts = (int*)workspace;
  const size_t lds = (size_t)V * sizeof(int);
  hipLaunchKernelGGL(pa::lda_index_count_kernel, dim3(pa::LDA_IDX_CHUNKS), dim3(64), lds, s, words,
                     L.n, chunk, (int)V, counts);
  int rc = pa::check_launch("lda_index_count_kernel");
  if (rc != PA_OK) return rc;
  hipLaunchKernelGGL(pa::lda_index_scan_kernel, dim3(1), dim3(1024), 0, s, counts, (int)V,
                     (int*)index, L, Wd, B);
  rc = pa::check_launch("lda_index_scan_kernel");
  if (rc != PA_OK) return rc;
  hipLaunchKernelGGL(pa::lda_index_fill_kernel, dim3(pa::LDA_IDX_CHUNKS), dim3(64), lds, s, words,
                     L.n, chunk, (int)V, B, counts, (int*)index, L);
  return pa::check_launch("lda_index_fill_kernel");
}

size_t pa_lda_factor_indexed_workspace(int dtype, int64_t Wd, int64_t B, int64_t T, int64_t V) {
  pa::LdaIndexLayout L;
  if (T < 1 || !pa::lda_index_layout(Wd, B, V, &L)) return 0;
  return 256 + (size_t)L.cap * (size_t)T * (dtype == PA_F32 ? 4 : 8);
}

int pa_lda_factor_indexed_fwd_bwd(int dtype, const int64_t* words, const void* index,
                                  size_t index_bytes, const void* log_theta, const void* log_phi,
                                  int64_t Wd, int64_t B, int64_t T, int64_t V, void* out_doc,
                                  void* g_theta, void* g_phi, void* workspace,
                                  size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "lda_factor_indexed: bad dtype %d", dtype);
  PA_REQUIRE(Wd >= 0 && B >= 0 && T >= 1 && V >= 1, "lda_factor_indexed: bad shape");
  if (T > 64) return pa::fail(PA_ERR_UNSUPPORTED, "lda_factor_indexed: T=%lld > 64", (long long)T);
  pa::LdaIndexLayout L;
  const size_t need = pa_lda_index_bytes(Wd, B, V);
  if (need == 0 || !pa::lda_index_layout(Wd, B, V, &L))
    return pa::fail(PA_ERR_UNSUPPORTED, "lda_factor_indexed: no index for this shape");
  PA_REQUIRE(index && index_bytes >= need, "lda_factor_indexed: index buffer too small");
  const int tmax = pa::lda_tmax(T);
  const size_t esz = dtype == PA_F32 ? 4 : 8;
  if (pa::lda_doc_waves(V, tmax, esz) == 0)
    return pa::fail(PA_ERR_UNSUPPORTED, "lda_factor_indexed: V*T table (%lld x %d) does not fit in LDS",
                    (long long)V, tmax);
  PA_REQUIRE(log_phi && g_phi && workspace, "lda_factor_indexed: NULL pointer");
  PA_REQUIRE(B == 0 || (log_theta && out_doc && g_theta), "lda_factor_indexed: NULL pointer");
  PA_REQUIRE(B == 0 || Wd == 0 || words, "lda_factor_indexed: NULL words");
  PA_REQUIRE(workspace_bytes >= pa_lda_factor_indexed_workspace(dtype, Wd, B, T, V),
             "lda_factor_indexed: workspace too small");
  hipStream_t s = pa::as_stream(stream);
  const int* img = (const int*)index;
#define PA_LDA_CASE(TM)                                                                          \
  if (tmax == TM) {                                                                              \
    if (dtype == PA_F32)                                                                         \
      return pa::lda_indexed_launch<float, TM>(words, img, L, (const float*)log_theta,           \
                                               (const float*)log_phi, Wd, B, (int)T, (int)V,     \
                                               (float*)out_doc, (float*)g_theta, (float*)g_phi,  \
                                               workspace, s);                                    \
    return pa::lda_indexed_launch<double, TM>(words, img, L, (const double*)log_theta,           \
                                              (const double*)log_phi, Wd, B, (int)T, (int)V,     \
                                              (double*)out_doc, (double*)g_theta,                \
                                              (double*)g_phi, workspace, s);                     \
  }
  PA_LDA_CASE(8)
  PA_LDA_CASE(16)
  PA_LDA_CASE(32)
  PA_LDA_CASE(64)
#undef PA_LDA_CASE
  return pa::fail(PA_ERR_UNSUPPORTED, "lda_factor_indexed: unreachable");
}

}  // extern "C"
