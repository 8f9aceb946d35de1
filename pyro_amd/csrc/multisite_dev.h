// multisite.hip -- the many-small-sites side of an ELBO step in a handful of launches.
//
// Reference path replaced: for every global latent the reference launches log_prob,
// scale_and_mask, .sum() (pyro/poutine/trace_struct.py:248-288), adds the per-site sums on the
// host side of autograd (pyro/infer/trace_elbo.py:82-112) and runs the autograd duals of each of
// them; AutoNormal draws every site with its own softplus + rsample chain
// (pyro/infer/autoguide/guides.py:415-603).  For tensors of a few thousand elements all of that
// is launch latency.  Here:
//   * multi_sum_kernel     : ONE workgroup walks a table of entries and produces the signed,
//                            scaled, masked total (fp64 accumulation, fixed order);
//   * multi_grad_kernel    : one workgroup per entry writes every requested operand gradient
//                            already reduced to the operand's own broadcast shape;
//   * meanfield_sample_*   : all mean-field Normal sites of a guide in one launch each way.
// The entry tables travel in the kernel arguments (no device-side descriptor buffers).
#pragma once
#include "common.h"
#include "dist_fam.h"

namespace pa {

struct EntryDev {
  int dist, need;
  int64_t rows, cols;
  const void *v, *a, *b;
  const uint8_t* m;
  int64_t vsr, vsc, asr, asc, bsr, bsc, msr, msc;
  double coef;
  void *dv, *da, *db;
  int chain_next;
  const void* xg;
  double xcoef;
};
struct MultiArgs {
  int n;
  EntryDev e[PA_MULTI_MAX_ENTRIES];
};

// The entry tables are kernel arguments passed BY VALUE and indexed with a run-time (uniform)
// index.  Indexing the parameter object itself makes the compiler spill the whole table to scratch
// memory in every thread, and reading it through a generic pointer makes every field access a
// separate (re-issued) vector load.  Instead ONE table element is copied out of the kernarg segment
// through a constant-address-space pointer: scalar loads into SGPRs, done once.  The table must be
// the FIRST kernel parameter (offset 0 of the kernarg segment).
template <typename S>
__device__ __forceinline__ S kernarg_load(uint32_t byte_offset) {
  static_assert(sizeof(S) % 4 == 0, "kernarg element must be a multiple of 4 bytes");
  typedef __attribute__((address_space(4))) const uint32_t* cptr;
  typedef __attribute__((address_space(4))) const char* cbytes;
  cptr p = (cptr)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset);
  union { S s; uint32_t w[sizeof(S) / 4]; } u;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(S) / 4); ++i) u.w[i] = p[i];
  return u.s;
}

// developer probe (chain.hip, PA_CHAIN_LATENCY_PROBE): wall-clock stamps from inside the bodies
#ifdef PA_CHAIN_LATENCY_PROBE
static __device__ uint64_t* pa_dbg_stamps = nullptr;
static __device__ int pa_dbg_next = 0;
#define PA_DBG_STAMP()                                                                  \
  do {                                                                                  \
    if (pa_dbg_stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && pa_dbg_next < 30) \
      pa_dbg_stamps[32 + pa_dbg_next++] = wall_clock64();                               \
  } while (0)
#else
#define PA_DBG_STAMP() do { } while (0)
#endif

constexpr int MULTI_THREADS = 1024;
constexpr int UN = 8;   // independent iterations per batch: their loads are in flight together
template <> struct NParams<PA_SITE_IDENTITY> { static constexpr int n = 1; };

// These kernels touch a few thousand elements: what they cost is dependent memory round trips
// (~1 us each), not bandwidth.  Every loop therefore runs in batches of UN branch-free iterations
// (out-of-range iterations read element 0 and are discarded), so that the loads of a batch are
// issued back to back, and entries / operands are processed side by side wherever possible.
// Entries are small (rows*cols <= PA_MULTI_MAX_ELEMS): 32-bit index arithmetic throughout.
template <typename T>
struct Elem {
  T v, a, b;
  bool keep;
};
template <int DIST, typename T>
__device__ __forceinline__ Elem<T> load_elem(const EntryDev& e, uint32_t r, uint32_t c, bool ok) {
  r = ok ? r : 0u;
  c = ok ? c : 0u;
  Elem<T> x;
  x.v = ((const T*)e.v)[r * (int32_t)e.vsr + c * (int32_t)e.vsc];
  x.a = T(0);
  x.b = T(0);
  if constexpr (DIST < PA_DIST_COUNT) {
    x.a = ((const T*)e.a)[r * (int32_t)e.asr + c * (int32_t)e.asc];
    if (NParams<DIST>::n > 1) x.b = ((const T*)e.b)[r * (int32_t)e.bsr + c * (int32_t)e.bsc];
  }
  x.keep = ok && (e.m == nullptr || e.m[r * (int32_t)e.msr + c * (int32_t)e.msc] != 0);
  return x;
}
template <int DIST, typename T>
__device__ __forceinline__ T elem_lp(const Elem<T>& x) {
  if constexpr (DIST == PA_SITE_IDENTITY) return x.v;
  else if constexpr (DIST == PA_SITE_NONE) return T(0);
  else return Fam<DIST, T>::lp(x.v, x.a, x.b);
}
template <int DIST, typename T>
__device__ __forceinline__ void elem_grad(const Elem<T>& x, T& gv, T& ga, T& gb) {
  if constexpr (DIST == PA_SITE_IDENTITY) { gv = T(1); ga = T(0); gb = T(0); }
  else if constexpr (DIST == PA_SITE_NONE) { gv = T(0); ga = T(0); gb = T(0); }
  else Fam<DIST, T>::grad(x.v, x.a, x.b, gv, ga, gb);
}

#define PA_DISPATCH_ENTRY(DIST_ID, CALL)                                                         \
  switch (DIST_ID) {                                                                             \
    case PA_DIST_NORMAL: { constexpr int D_ = PA_DIST_NORMAL; CALL; } break;                     \
    case PA_DIST_BERNOULLI_LOGITS: { constexpr int D_ = PA_DIST_BERNOULLI_LOGITS; CALL; } break; \
    case PA_DIST_HALF_CAUCHY: { constexpr int D_ = PA_DIST_HALF_CAUCHY; CALL; } break;           \
    case PA_DIST_LOG_NORMAL: { constexpr int D_ = PA_DIST_LOG_NORMAL; CALL; } break;             \
    case PA_DIST_EXPONENTIAL: { constexpr int D_ = PA_DIST_EXPONENTIAL; CALL; } break;           \
    case PA_DIST_HALF_NORMAL: { constexpr int D_ = PA_DIST_HALF_NORMAL; CALL; } break;           \
    case PA_DIST_GAMMA: { constexpr int D_ = PA_DIST_GAMMA; CALL; } break;                       \
    case PA_DIST_BETA: { constexpr int D_ = PA_DIST_BETA; CALL; } break;                         \
    case PA_DIST_POISSON: { constexpr int D_ = PA_DIST_POISSON; CALL; } break;                   \
    case PA_DIST_BINOMIAL_LOGITS: { constexpr int D_ = PA_DIST_BINOMIAL_LOGITS; CALL; } break;   \
    case PA_DIST_KL_NORMAL_LOC: { constexpr int D_ = PA_DIST_KL_NORMAL_LOC; CALL; } break;       \
    case PA_DIST_KL_NORMAL_SCALE: { constexpr int D_ = PA_DIST_KL_NORMAL_SCALE; CALL; } break;   \
    case PA_SITE_IDENTITY: { constexpr int D_ = PA_SITE_IDENTITY; CALL; } break;                 \
    default: { constexpr int D_ = PA_SITE_NONE; CALL; } break;                                   \
  }

// sum of the masked log-densities of entry e over the `nth` threads (tid = 0..nth-1) of the waves
// that share it.  Thread -> (column c0, row group g): no per-element index division, rows in
// batches of UNS branch-free iterations (these loops are as much VALU-issue-bound -- logf, a
// division per element -- as latency-bound, so the batches are kept small: an out-of-range
// iteration still costs its arithmetic).
constexpr int UNS = 4;
template <int DIST, typename T>
__device__ __forceinline__ double entry_sum(const EntryDev& e, uint32_t tid, uint32_t nth) {
  const uint32_t R = (uint32_t)e.rows, C = (uint32_t)e.cols;
  T acc = T(0);
  if constexpr (DIST == PA_SITE_NONE) return 0.0;
  const uint32_t tk = C < nth ? C : nth, ng = nth / tk;
  const uint32_t c0 = tid % tk, g = tid / tk;
  if (g >= ng) return 0.0;
  for (uint32_t c = c0; c < C; c += tk)
    for (uint32_t rb = g; rb < R; rb += UNS * ng) {
      Elem<T> x[UNS];
#pragma unroll
      for (int u = 0; u < UNS; ++u) x[u] = load_elem<DIST, T>(e, rb + u * ng, c, rb + u * ng < R);
#pragma unroll
      for (int u = 0; u < UNS; ++u) acc += x[u].keep ? elem_lp<DIST, T>(x[u]) : T(0);
    }
  return (double)acc;   // few terms per lane; fp64 across lanes
}

// ONE workgroup of 16 waves: the entries are spread over the waves (several waves per entry when
// there are fewer than 16), so that all of them are read concurrently.
// multi_sum_partial: this wave's share of sum_e coef_e * sum(entry e) over the entries selected by
// (skip_mask, only_mask): bit e of skip_mask set = leave entry e out; only_mask != 0 = only those.
// The chained tail sums the entries that do not depend on the finalize phase while that phase is
// still running and the others afterwards -- the same additions in the same order as one call as
// long as every wave owns at most one entry (n <= NTHREADS / 64; the caller guarantees it).
template <typename T, int NTHREADS>
__device__ __forceinline__ double multi_sum_partial(uint32_t kbase, uint32_t skip_mask,
                                                    uint32_t only_mask) {
  const int n_entries = kernarg_load<int>(kbase + offsetof(MultiArgs, n));
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int NW = NTHREADS / 64;
  const int wpe = n_entries >= NW ? 1 : NW / (n_entries > 0 ? n_entries : 1);   // waves per entry
  const int epr = NW / wpe;                                                     // entries per round
  double acc = 0.0;
  for (int k0 = 0; k0 < n_entries; k0 += epr) {
    const int k = k0 + wave / wpe;
    if (k < n_entries && wave / wpe < epr) {
      if ((skip_mask >> k & 1u) || (only_mask != 0u && !(only_mask >> k & 1u))) continue;
      const EntryDev e = kernarg_load<EntryDev>(kbase + offsetof(MultiArgs, e) + k * sizeof(EntryDev));
      const uint32_t tid = (uint32_t)((wave % wpe) * 64 + lane), nth = (uint32_t)(wpe * 64);
      double s = 0.0;
      PA_DISPATCH_ENTRY(e.dist, s = (entry_sum<D_, T>(e, tid, nth)));
      acc += e.coef * s;
    }
  }
  return acc;
}
template <typename T>
__device__ __forceinline__ void multi_sum_finish(double acc, T* __restrict__ out, double coef_all,
                                                 int accumulate) {
  __shared__ double smem[16];
  const double t = block_sum_f64(acc, smem);
  if (threadIdx.x == 0) {
    const double base = accumulate ? (double)*out : 0.0;
    *out = (T)(base + coef_all * t);
  }
}
template <typename T, int NTHREADS>
__device__ __forceinline__ void multi_sum_body(uint32_t kbase, T* __restrict__ out, double coef_all,
                                               int accumulate) {
  multi_sum_finish<T>(multi_sum_partial<T, NTHREADS>(kbase, 0u, 0u), out, coef_all, accumulate);
}

template <typename T>
__global__ __launch_bounds__(MULTI_THREADS) void multi_sum_kernel(const MultiArgs args_by_value,
                                                                  T* __restrict__ out,
                                                                  double coef_all, int accumulate) {
  multi_sum_body<T, MULTI_THREADS>(0u, out, coef_all, accumulate);
}

constexpr int GRAD_THREADS = 256;
// thread groups along the summed dimension for tk threads along the kept one.  Capped: the groups'
// partial sums are combined by ONE thread per kept index reading them back from LDS one after the
// other (fixed order), and 256 dependent LDS reads cost more than the whole rest of the kernel.
__device__ __forceinline__ uint32_t row_groups(uint32_t tk) {
  const uint32_t ng = GRAD_THREADS / tk;
  return ng > 8u ? 8u : ng;
}
enum { PAT_SKIP = 0, PAT_FULL = 1, PAT_ROWRED = 2, PAT_SCALAR = 3, PAT_COLRED = 4 };
__device__ __forceinline__ int pattern_of(bool wanted, int64_t sr, int64_t sc, int64_t R, int64_t C) {
  if (!wanted) return PAT_SKIP;
  const bool red_r = (sr == 0 && R > 1), red_c = (sc == 0 && C > 1);
  return red_r ? (red_c ? PAT_SCALAR : PAT_ROWRED) : (red_c ? PAT_COLRED : PAT_FULL);
}

// One pass over entry e that produces, side by side,
//   the value gradient (un-reduced: written, or added when `accumulate`; plus the known extra term),
//   the p0 and p1 gradients for the patterns FULL / ROWRED (summed over rows) / SCALAR.
// Thread t owns column c = t % tk and the rows g, g + ng, ... (g = t / tk): a fixed element ->
// thread map, so that chained passes over the same value buffer need no synchronisation.
template <int DIST, typename T>
__device__ __forceinline__ void combined_pass(const EntryDev& e, double w, int pv, bool accumulate,
                                              const T* xg, T xw, int pa_, int pb_, double* red) {
  // (no contraction of a * b + c in this function's own expressions: the same sums are formed by
  //  site_tail.h in one pass, and the two must round alike whatever the surrounding code)
#pragma clang fp contract(off)
  const uint32_t R = (uint32_t)e.rows, C = (uint32_t)e.cols, t = threadIdx.x;
  const uint32_t tk = C < GRAD_THREADS ? C : GRAD_THREADS, ng = row_groups(tk);
  const uint32_t c0 = t % tk, g = t / tk;
  T* dv = (T*)e.dv;
  T* da = (T*)e.da;
  T* db = (T*)e.db;
  const T wT = (T)w;
  T tot_a = T(0), tot_b = T(0);                 // SCALAR patterns: everything this thread saw
  for (uint32_t cb = 0; cb < C; cb += tk) {
    const uint32_t c = cb + c0;
    const bool okc = g < ng && c < C;
    T col_a = T(0), col_b = T(0);               // ROWRED patterns: this thread's column
    for (uint32_t rb = g; rb < R; rb += UN * ng) {
      Elem<T> x[UN];
      T old[UN], ex[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const uint32_t r = rb + u * ng;
        const bool ok = okc && r < R;
        x[u] = load_elem<DIST, T>(e, r, c, ok);
        const uint32_t o = ok ? r * C + c : 0u;
        old[u] = (pv == PAT_FULL && accumulate) ? dv[o] : T(0);
        ex[u] = (pv == PAT_FULL && xg != nullptr) ? xg[o] : T(0);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const uint32_t r = rb + u * ng;
        const bool ok = okc && r < R;
        T gv, ga, gb;
        elem_grad<DIST, T>(x[u], gv, ga, gb);
        gv = x[u].keep ? gv : T(0);
        ga = x[u].keep ? ga : T(0);
        gb = x[u].keep ? gb : T(0);
        if (ok) {
          const uint32_t o = r * C + c;
          if (pv == PAT_FULL) dv[o] = old[u] + wT * gv + xw * ex[u];
          if (pa_ == PAT_FULL) da[o] = wT * ga;
          if (pb_ == PAT_FULL) db[o] = wT * gb;
        }
        col_a += ga;
        col_b += gb;
      }
    }
    tot_a += col_a;
    tot_b += col_b;
    if (pa_ == PAT_ROWRED || pb_ == PAT_ROWRED) {
      __syncthreads();
      red[t] = (double)col_a;
      red[GRAD_THREADS + t] = (double)col_b;
      __syncthreads();
      if (g == 0 && c < C) {
        double sa = 0.0, sb = 0.0;
        for (uint32_t j = 0; j < ng; ++j) {
          sa += red[j * tk + c0];
          sb += red[GRAD_THREADS + j * tk + c0];
        }
        if (pa_ == PAT_ROWRED) da[c] = (T)(w * sa);
        if (pb_ == PAT_ROWRED) db[c] = (T)(w * sb);
      }
    }
  }
  if (pa_ == PAT_SCALAR || pb_ == PAT_SCALAR) {
    __syncthreads();
    red[t] = (double)tot_a;
    red[GRAD_THREADS + t] = (double)tot_b;
    __syncthreads();
    if (t < 64) {
      double sa = 0.0, sb = 0.0;
      for (uint32_t j = t; j < GRAD_THREADS; j += 64) {
        sa += red[j];
        sb += red[GRAD_THREADS + j];
      }
      sa = wave_sum(sa);
      sb = wave_sum(sb);
      if (t == 0) {
        if (pa_ == PAT_SCALAR) da[0] = (T)(w * sa);
        if (pb_ == PAT_SCALAR) db[0] = (T)(w * sb);
      }
    }
  }
}

// dv += xw * xg (the known extra term of a value gradient), with combined_pass's element -> thread
// map.  Applied AFTER the entry's own pass and every chained pass: the extra term (the fused GLM
// site's gradient) is the only operand of an ELBO assembly that depends on the big kernel before
// it, so everything else can be computed while that kernel's reduction is still running
// (chain.hip: the fused tail does exactly that).
template <typename T>
__device__ __forceinline__ void extras_pass(const EntryDev& e, const T* xg, T xw) {
#pragma clang fp contract(off)
  const uint32_t R = (uint32_t)e.rows, C = (uint32_t)e.cols, t = threadIdx.x;
  const uint32_t tk = C < GRAD_THREADS ? C : GRAD_THREADS, ng = row_groups(tk);
  const uint32_t c0 = t % tk, g = t / tk;
  T* dv = (T*)e.dv;
  for (uint32_t cb = 0; cb < C; cb += tk) {
    const uint32_t c = cb + c0;
    const bool okc = g < ng && c < C;
    for (uint32_t rb = g; rb < R; rb += UN * ng) {
      T old[UN], ex[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const uint32_t r = rb + u * ng;
        const uint32_t o = (okc && r < R) ? r * C + c : 0u;
        old[u] = dv[o];
        ex[u] = xg[o];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const uint32_t r = rb + u * ng;
        if (okc && r < R) dv[r * C + c] = old[u] + xw * ex[u];
      }
    }
  }
}

// generic single-operand pass (any pattern, incl. reductions over columns): the rarely needed
// fallback for operands the combined pass does not cover
template <int DIST, typename T>
__device__ __forceinline__ void operand_pass(const EntryDev& e, int which, int pat, T* out, double w,
                                             double* red) {
#pragma clang fp contract(off)
  const uint32_t R = (uint32_t)e.rows, C = (uint32_t)e.cols, t = threadIdx.x;
  auto at = [&](uint32_t r, uint32_t c) -> T {
    const Elem<T> x = load_elem<DIST, T>(e, r, c, true);
    T gv, ga, gb;
    elem_grad<DIST, T>(x, gv, ga, gb);
    const T gsel = which == 0 ? gv : (which == 1 ? ga : gb);
    return x.keep ? gsel : T(0);
  };
  if (pat == PAT_FULL) {
    for (uint32_t i = t; i < R * C; i += GRAD_THREADS) out[i] = (T)(w * (double)at(i / C, i % C));
  } else if (pat == PAT_ROWRED || pat == PAT_COLRED) {
    const bool rr = pat == PAT_ROWRED;
    const uint32_t K = rr ? C : R, L = rr ? R : C;
    const uint32_t tk = K < GRAD_THREADS ? K : GRAD_THREADS, ng = row_groups(tk);
    const uint32_t k0 = t % tk, g = t / tk;
    for (uint32_t kb = 0; kb < K; kb += tk) {
      const uint32_t k = kb + k0;
      T acc = T(0);
      if (g < ng && k < K)
        for (uint32_t l = g; l < L; l += ng) acc += rr ? at(l, k) : at(k, l);
      __syncthreads();
      red[t] = (double)acc;
      __syncthreads();
      if (g == 0 && k < K) {
        double sacc = 0.0;
        for (uint32_t j = 0; j < ng; ++j) sacc += red[j * tk + k0];
        out[k] = (T)(w * sacc);
      }
    }
  } else if (pat == PAT_SCALAR) {
    T acc = T(0);
    for (uint32_t i = t; i < R * C; i += GRAD_THREADS) acc += at(i / C, i % C);
    __syncthreads();
    const double tot = block_sum_f64_waves((double)acc, red, GRAD_THREADS / 64);
    if (t == 0) out[0] = (T)(w * tot);
  }
  __syncthreads();
}

// everything workgroup `blockIdx.x` owes for entry e: its own value gradient (unless a chain head
// produces it) followed by the chained entries' contributions to the same buffer, and its p0 / p1
// gradients -- the common patterns in one pass each
// mode 0: everything; 1: everything but the extra term; 2: the extra term only
enum { GRAD_ALL = 0, GRAD_NO_EXTRAS = 1, GRAD_EXTRAS_ONLY = 2 };
template <typename T>
__device__ __forceinline__ void multi_grad_body(uint32_t kbase, int entry, const T* __restrict__ g,
                                                double coef_all, int mode = GRAD_ALL) {
  __shared__ double red[2 * GRAD_THREADS];
  PA_DBG_STAMP();
  const EntryDev e = kernarg_load<EntryDev>(kbase + offsetof(MultiArgs, e) + entry * sizeof(EntryDev));
  if (e.rows * e.cols == 0) return;
  const double gw = (g != nullptr ? (double)g[0] : 1.0) * coef_all;
  PA_DBG_STAMP();
  const bool own_value = (e.need & PA_NEED_VALUE) && e.dv && !(e.need & PA_VALUE_BY_CHAIN);
  const bool param_family = e.dist >= 0 && e.dist < PA_DIST_COUNT;
  int pv = pattern_of(own_value, e.vsr, e.vsc, e.rows, e.cols);
  int pa_ = pattern_of(param_family && (e.need & PA_NEED_P0) && e.da, e.asr, e.asc, e.rows, e.cols);
  int pb_ = pattern_of(param_family && (e.need & PA_NEED_P1) && e.db &&
                           dist_nparams(e.dist) > 1,
                       e.bsr, e.bsc, e.rows, e.cols);
  // what the combined pass cannot do goes through the generic per-operand passes first
  const int pv_c = pv == PAT_FULL ? pv : PAT_SKIP;
  const int pa_c = pa_ == PAT_COLRED ? PAT_SKIP : pa_;
  const int pb_c = pb_ == PAT_COLRED ? PAT_SKIP : pb_;
  const double w = gw * e.coef;
  const bool extras = own_value && pv == PAT_FULL && e.xg != nullptr;
  if (mode == GRAD_EXTRAS_ONLY) {
    if (extras) extras_pass<T>(e, (const T*)e.xg, (T)(gw * e.xcoef));
    return;
  }
  if (pv != PAT_SKIP && pv != PAT_FULL) {
    PA_DISPATCH_ENTRY(e.dist, (operand_pass<D_, T>(e, 0, pv, (T*)e.dv, w, red)));
  }
  if (pa_ == PAT_COLRED) { PA_DISPATCH_ENTRY(e.dist, (operand_pass<D_, T>(e, 1, pa_, (T*)e.da, w, red))); }
  if (pb_ == PAT_COLRED) { PA_DISPATCH_ENTRY(e.dist, (operand_pass<D_, T>(e, 2, pb_, (T*)e.db, w, red))); }
  if (pv_c != PAT_SKIP || pa_c != PAT_SKIP || pb_c != PAT_SKIP) {
    PA_DISPATCH_ENTRY(e.dist, (combined_pass<D_, T>(e, w, pv_c, false, (const T*)nullptr, T(0), pa_c, pb_c, red)));
  }
  PA_DBG_STAMP();
  if (own_value && pv == PAT_FULL) {
    for (int k = e.chain_next; k >= 0;) {   // same value tensor, same frame, same element->thread map
      const EntryDev m = kernarg_load<EntryDev>(kbase + offsetof(MultiArgs, e) + k * sizeof(EntryDev));
      EntryDev mm = m;
      mm.dv = e.dv;
      PA_DISPATCH_ENTRY(m.dist, (combined_pass<D_, T>(mm, gw * m.coef, PAT_FULL, true, (const T*)nullptr, T(0),
                                                      PAT_SKIP, PAT_SKIP, red)));
      k = m.chain_next;
      PA_DBG_STAMP();
    }
  }
  if (extras && mode == GRAD_ALL) extras_pass<T>(e, (const T*)e.xg, (T)(gw * e.xcoef));
}

template <typename T>
__global__ __launch_bounds__(GRAD_THREADS) void multi_grad_kernel(const MultiArgs args_by_value,
                                                                  const T* __restrict__ g,
                                                                  double coef_all) {
  multi_grad_body<T>(0u, (int)blockIdx.x, g, coef_all);
}

// Forward AND backward of the table in one launch, for a caller that knows it will differentiate
// the total right away (Trace_ELBO.loss_and_grads: surrogate.backward() follows the forward
// immediately, pyro/infer/trace_elbo.py:153-157): workgroups 0..n-1 write the operand gradients
// for an upstream gradient g (NULL = 1), workgroup n the total.
template <typename T, int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void multi_sum_grad_kernel(
    const MultiArgs args_by_value, T* __restrict__ out, const T* __restrict__ g, double coef_all,
    int accumulate) {
  const int n_entries = kernarg_load<int>(offsetof(MultiArgs, n));
  if ((int)blockIdx.x == n_entries) {
    multi_sum_body<T, NTHREADS>(0u, out, coef_all, accumulate);
  } else {
    // the gradient code is written for GRAD_THREADS threads; surplus waves of a wider launch (the
    // width the total's workgroup wants) leave at once -- they take no part in its barriers
    if (NTHREADS > GRAD_THREADS && threadIdx.x >= GRAD_THREADS) return;
    multi_grad_body<T>(0u, (int)blockIdx.x, g, coef_all);
  }
}

static int to_dev(const pa_site_entry* in, int n, MultiArgs* out, const char* who) {
  PA_REQUIRE(n >= 0 && n <= PA_MULTI_MAX_ENTRIES, "%s: %d entries (max %d per call)", who, n,
             PA_MULTI_MAX_ENTRIES);
  PA_REQUIRE(n == 0 || in != nullptr, "%s: NULL entry table", who);
  out->n = n;
  for (int k = 0; k < n; ++k) {
    const pa_site_entry& s = in[k];
    PA_REQUIRE((s.dist >= 0 && s.dist < PA_DIST_COUNT) || s.dist == PA_SITE_IDENTITY ||
                   s.dist == PA_SITE_NONE,
               "%s: entry %d: unknown distribution id %d", who, k, s.dist);
    PA_REQUIRE(s.rows >= 0 && s.cols >= 0 && s.rows * s.cols <= PA_MULTI_MAX_ELEMS,
               "%s: entry %d: shape [%lld,%lld] out of range", who, k, (long long)s.rows,
               (long long)s.cols);
    PA_REQUIRE(s.rows * s.cols == 0 || s.value.ptr, "%s: entry %d: NULL value", who, k);
    if (s.dist < PA_DIST_COUNT) {
      PA_REQUIRE(s.rows * s.cols == 0 || s.p0.ptr, "%s: entry %d: NULL p0", who, k);
      PA_REQUIRE(s.rows * s.cols == 0 ||
                     dist_nparams(s.dist) < 2 || s.p1.ptr,
                 "%s: entry %d: family needs p1", who, k);
    }
    EntryDev& d = out->e[k];
    d.dist = s.dist; d.need = s.need; d.rows = s.rows; d.cols = s.cols;
    d.v = s.value.ptr; d.a = s.p0.ptr; d.b = s.p1.ptr; d.m = (const uint8_t*)s.mask.ptr;
    d.vsr = s.value.stride_row; d.vsc = s.value.stride_col;
    d.asr = s.p0.stride_row; d.asc = s.p0.stride_col;
    d.bsr = s.p1.stride_row; d.bsc = s.p1.stride_col;
    d.msr = s.mask.stride_row; d.msc = s.mask.stride_col;
    d.coef = s.coef; d.dv = s.d_value; d.da = s.d_p0; d.db = s.d_p1;
    PA_REQUIRE(s.chain_next >= -1 && s.chain_next < n && s.chain_next != k,
               "%s: entry %d: bad chain_next %d", who, k, s.chain_next);
    const bool red_v = (s.value.stride_row == 0 && s.rows > 1) ||
                       (s.value.stride_col == 0 && s.cols > 1);
    PA_REQUIRE(!(red_v && (s.chain_next >= 0 || s.extra_grad || (s.need & PA_VALUE_BY_CHAIN))),
               "%s: entry %d: chained / extra value gradients need an un-reduced value operand", who,
               k);
    PA_REQUIRE(s.chain_next < 0 || (in[s.chain_next].rows == s.rows && in[s.chain_next].cols == s.cols),
               "%s: entry %d: chain members must share the frame", who, k);
    d.chain_next = s.chain_next; d.xg = s.extra_grad; d.xcoef = s.extra_coef;
  }
  return PA_OK;
}

// ---- mean-field Normal guide -----------------------------------------------------------------
// (one fma, spelled out: the GLM kernel that draws its own weights must produce the same bits)
__device__ __forceinline__ float t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
struct MfSiteDev {
  const void *loc, *rho;
  void *z, *scale, *loc_out, *eps;
  int64_t n;
  uint64_t offset;
  int accumulate;
  const void *d_z, *d_scale, *d_loc_out;
  void *d_loc, *d_rho;
};
struct MfArgs {
  int nsites;
  MfSiteDev s[PA_MF_MAX_SITES];
};

// d softplus / d x (1 beyond torch's threshold of 20)
template <typename T> __device__ __forceinline__ double softplus_slope(T rho) {
  if constexpr (sizeof(T) == 8) {
    const double x = (double)rho;
    return x > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-x));
  } else {
    const float x = (float)rho;
    return x > 20.0f ? 1.0 : (double)(1.0f / (1.0f + expf(-x)));
  }
}
template <typename T> __device__ __forceinline__ T softplus_t(T x) {
  // torch.nn.functional.softplus (threshold 20): x for large x, log1p(exp(x)) otherwise
  return x > T(20) ? x : t_log1p(t_exp(x));
}

template <typename T>
__global__ __launch_bounds__(256) void meanfield_sample_kernel(const MfArgs args_by_value,
                                                               int64_t P, uint64_t seed,
                                                               const uint64_t* __restrict__ offset_dev,
                                                               const int64_t* __restrict__ gate) {
  if (gate != nullptr && *gate != 0) return;        // the step gate gave this replay up (pa_gate)
  const MfSiteDev s = kernarg_load<MfSiteDev>(offsetof(MfArgs, s) + blockIdx.y * sizeof(MfSiteDev));
  const uint64_t off = s.offset + (offset_dev ? *offset_dev : 0);
  const T* loc = (const T*)s.loc;
  const T* rho = (const T*)s.rho;
  T* z = (T*)s.z;
  T* eps = (T*)s.eps;
  T* sc = (T*)s.scale;
  T* lo = (T*)s.loc_out;
  // one element per thread: small sites (a step's latents of a few thousand elements) are a latency
  // problem -- one short dependent chain per thread, the redundant Philox blocks cost nothing
  const int64_t total = P * s.n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i % s.n;
    T e;
    if constexpr (sizeof(T) == 4) e = philox_normal_f32(seed, off, (uint64_t)i);
    else e = philox_normal_f64(seed, off, (uint64_t)i);
    const T sp = softplus_t<T>(rho[c]);
    eps[i] = e;
    z[i] = t_fma(sp, e, loc[c]);
    if (i < s.n) {
      sc[c] = sp;
      lo[c] = loc[c];
    }
  }
}

// the same draws for LARGE sites (a plated latent of millions of elements: throughput)
template <typename T>
__global__ __launch_bounds__(256) void meanfield_sample_block_kernel(const MfArgs args_by_value,
                                                               int64_t P, uint64_t seed,
                                                               const uint64_t* __restrict__ offset_dev,
                                                               const int64_t* __restrict__ gate) {
  if (gate != nullptr && *gate != 0) return;        // the step gate gave this replay up (pa_gate)
  const MfSiteDev s = kernarg_load<MfSiteDev>(offsetof(MfArgs, s) + blockIdx.y * sizeof(MfSiteDev));
  const uint64_t off = s.offset + (offset_dev ? *offset_dev : 0);
  const T* loc = (const T*)s.loc;
  const T* rho = (const T*)s.rho;
  T* z = (T*)s.z;
  T* eps = (T*)s.eps;
  T* sc = (T*)s.scale;
  T* lo = (T*)s.loc_out;
  const int64_t total = P * s.n;
  // one Philox block per thread and trip: its 4 (f32) / 2 (f64) normals are the draws of PER
  // consecutive elements (philox_normal_f32 / _f64: element i = lane i % PER of block i / PER) -- the
  // same numbers as one block per element, a quarter / half of the generator work; the column index
  // advances with the element instead of a 64-bit modulo per element
  constexpr int PER = sizeof(T) == 4 ? 4 : 2;
  const int64_t nblk = (total + PER - 1) / PER;
  // A site whose rows are whole blocks (n % PER == 0, 16-byte aligned outputs): thread -> (a block of PER
  // columns, a stride of particles).  softplus(rho) of those columns is evaluated ONCE per thread, not once
  // per element, and each block's draws leave as one 16-byte store per output.  Same blocks, same
  // arithmetic: bitwise the loop below.
  if (s.n % PER == 0 && P >= 2 &&
      ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(eps)) & 15) == 0) {
    typedef T vecT __attribute__((ext_vector_type(PER)));
    const int64_t nq = s.n / PER, nthreads = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t npc = nthreads / nq;                 // particle strides that fit the grid
    const int64_t want = P / 4 > 1 ? P / 4 : 1;  // ... but at least ~4 particles per thread
    if (npc > want) npc = want;
    const int64_t pc = npc > 0 ? tid / nq : 0, pstep = npc > 0 ? npc : 1;
    if (npc > 0 && pc >= npc) return;
    for (int64_t cq = npc > 0 ? tid % nq : tid; cq < nq; cq += npc > 0 ? nq : nthreads) {
      T sp[PER], lc[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        sp[u] = softplus_t<T>(rho[cq * PER + u]);
        lc[u] = loc[cq * PER + u];
      }
      if (pc == 0) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          sc[cq * PER + u] = sp[u];
          lo[cq * PER + u] = lc[u];
        }
      }
      for (int64_t p = pc; p < P; p += pstep) {
        const int64_t q = p * nq + cq;
        const u32x4 blk = philox4x32_10(seed, off + (uint64_t)q, 0);
        vecT e, zz;
        if constexpr (sizeof(T) == 4) {
          float a0, a1, a2, a3;
          box_muller_f32(u32_to_unit_f32(blk.x), u32_to_unit_f32(blk.y), a0, a1);
          box_muller_f32(u32_to_unit_f32(blk.z), u32_to_unit_f32(blk.w), a2, a3);
          e[0] = a0; e[1] = a1; e[2] = a2; e[3] = a3;
        } else {
          double a0, a1;
          box_muller_f64(u32x2_to_unit_f64(blk.x, blk.y), u32x2_to_unit_f64(blk.z, blk.w), a0, a1);
          e[0] = a0; e[1] = a1;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) zz[u] = t_fma(sp[u], (T)e[u], lc[u]);
        *reinterpret_cast<vecT*>(eps + q * PER) = e;
        *reinterpret_cast<vecT*>(z + q * PER) = zz;
      }
    }
    return;
  }
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nblk;
       q += (int64_t)gridDim.x * blockDim.x) {
    const u32x4 blk = philox4x32_10(seed, off + (uint64_t)q, 0);
    T nrm[PER];
    if constexpr (sizeof(T) == 4) {
      float a0, a1, a2, a3;
      box_muller_f32(u32_to_unit_f32(blk.x), u32_to_unit_f32(blk.y), a0, a1);
      box_muller_f32(u32_to_unit_f32(blk.z), u32_to_unit_f32(blk.w), a2, a3);
      nrm[0] = a0; nrm[1] = a1; nrm[2] = a2; nrm[3] = a3;
    } else {
      double a0, a1;
      box_muller_f64(u32x2_to_unit_f64(blk.x, blk.y), u32x2_to_unit_f64(blk.z, blk.w), a0, a1);
      nrm[0] = a0; nrm[1] = a1;
    }
    const int64_t i0 = q * PER;
    int64_t c = i0 % s.n;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int64_t i = i0 + u;
      if (i < total) {
        const T sp = softplus_t<T>(rho[c]);
        eps[i] = nrm[u];
        z[i] = t_fma(sp, nrm[u], loc[c]);
        if (i < s.n) {
          sc[c] = sp;
          lo[c] = loc[c];
        }
      }
      c = c + 1 == s.n ? 0 : c + 1;
    }
  }
}

// grid = (sites, column tiles): a column's sum over the P particles is independent of every other
// column, so a large (plated) site spreads over many workgroups; small sites use tile 0 only
constexpr uint32_t MF_BWD_WIDE_N = 4096;
template <typename T>
__device__ __forceinline__ void meanfield_sample_bwd_body(uint32_t kbase, uint32_t site, uint32_t tile,
                                                          uint32_t ntiles, int64_t P) {
#pragma clang fp contract(off)
  __shared__ double red_l[256], red_s[256];
  const MfSiteDev s = kernarg_load<MfSiteDev>(kbase + offsetof(MfArgs, s) + site * sizeof(MfSiteDev));
  const T* dz = (const T*)s.d_z;
  const T* eps = (const T*)s.eps;
  const T* dsc = (const T*)s.d_scale;
  const T* dlo = (const T*)s.d_loc_out;
  const T* rho = (const T*)s.rho;
  T* dloc = (T*)s.d_loc;
  T* drho = (T*)s.d_rho;
  const uint32_t n = (uint32_t)s.n, t = threadIdx.x, PP = (uint32_t)P;
  // a plated site of thousands of columns: 64-column tiles x 4 row groups instead of 256 x 1 -- four
  // times the tiles to spread over the workgroups and a quarter of the dependent load batches per thread
  // (the pass is a latency chain of P / (UN * ng) round trips, not bandwidth)
  const uint32_t tk = n < 256 ? n : ((n >= MF_BWD_WIDE_N && PP >= 16) ? 64u : 256u);
  if (n == 0 || tile * tk >= n) return;
  const uint32_t ng = row_groups(tk), c0 = t % tk, g = t / tk;
  for (uint32_t cb = tile * tk; cb < n; cb += ntiles * tk) {
    const uint32_t c = cb + c0;
    const bool okc = g < ng && c < n;
    // the per-column inputs of the epilogue are requested up front, next to the first batch
    const uint32_t cc = c < n ? c : 0u;
    const T v_dsc = dsc != nullptr ? dsc[cc] : T(0), v_dlo = dlo != nullptr ? dlo[cc] : T(0);
    const T v_rho = rho[cc];
    const T v_ol = (s.accumulate && dloc) ? dloc[cc] : T(0), v_or = (s.accumulate && drho) ? drho[cc] : T(0);
    T al = T(0), as = T(0);
    if (dz != nullptr)
      for (uint32_t pb = g; pb < PP; pb += UN * ng) {
        T gz[UN], ev[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const uint32_t p = pb + u * ng;
          const uint32_t o = (okc && p < PP) ? p * n + c : 0u;
          gz[u] = dz[o];
          ev[u] = eps[o];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const bool ok = okc && (pb + u * ng) < PP;
          al += ok ? gz[u] : T(0);
          as += ok ? gz[u] * ev[u] : T(0);
        }
      }
    __syncthreads();
    red_l[t] = (double)al;
    red_s[t] = (double)as;
    __syncthreads();
    if (g == 0 && c < n) {
      double sl = 0.0, ss = 0.0;
      for (uint32_t j = 0; j < ng; ++j) {
        sl += red_l[j * tk + c0];
        ss += red_s[j * tk + c0];
      }
      ss += (double)v_dsc;
      sl += (double)v_dlo;
      const double sig = softplus_slope<T>(v_rho);
      if (dloc) dloc[c] = (T)sl + v_ol;
      if (drho) drho[c] = (T)(ss * sig) + v_or;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void meanfield_sample_bwd_kernel(const MfArgs args_by_value,
                                                                   int64_t P) {
  meanfield_sample_bwd_body<T>(0u, blockIdx.x, blockIdx.y, gridDim.y, P);
}

static int mf_to_dev(const pa_mf_site* in, int n, MfArgs* out, const char* who) {
  PA_REQUIRE(n >= 0 && n <= PA_MF_MAX_SITES, "%s: %d sites (max %d per call)", who, n,
             PA_MF_MAX_SITES);
  PA_REQUIRE(n == 0 || in != nullptr, "%s: NULL site table", who);
  out->nsites = n;
  for (int k = 0; k < n; ++k) {
    const pa_mf_site& s = in[k];
    PA_REQUIRE(s.n >= 0 && s.n < (int64_t(1) << 31), "%s: site %d: bad size", who, k);
    MfSiteDev& d = out->s[k];
    d.loc = s.loc; d.rho = s.rho; d.z = s.z; d.scale = s.scale; d.loc_out = s.loc_out;
    d.eps = s.eps; d.n = s.n; d.offset = s.offset; d.accumulate = s.accumulate;
    d.d_z = s.d_z; d.d_scale = s.d_scale;
    d.d_loc_out = s.d_loc_out; d.d_loc = s.d_loc; d.d_rho = s.d_rho;
  }
  return PA_OK;
}

}  // namespace pa
