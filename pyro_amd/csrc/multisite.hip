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
// index.  Indexing the parameter object itself would make the compiler spill the whole table to
// scratch memory in every thread; reading it through the kernarg segment pointer keeps the accesses
// scalar loads from constant memory.  The table must be the FIRST kernel parameter (offset 0).
template <typename A>
__device__ __forceinline__ const A* kernarg_table() {
  return (const A*)__builtin_amdgcn_kernarg_segment_ptr();   // address-space cast (constant -> generic)
}

constexpr int MULTI_THREADS = 1024;
template <> struct NParams<PA_SITE_IDENTITY> { static constexpr int n = 1; };

// entries are small (rows*cols <= PA_MULTI_MAX_ELEMS): 32-bit index arithmetic throughout
template <int DIST, typename T>
__device__ __forceinline__ double entry_sum(const EntryDev& e) {
  const T* v = (const T*)e.v;
  const T* a = (const T*)e.a;
  const T* b = (const T*)e.b;
  const uint32_t C = (uint32_t)e.cols, n = (uint32_t)(e.rows * e.cols);
  const int32_t vsr = (int32_t)e.vsr, vsc = (int32_t)e.vsc, asr = (int32_t)e.asr,
                asc = (int32_t)e.asc, bsr = (int32_t)e.bsr, bsc = (int32_t)e.bsc,
                msr = (int32_t)e.msr, msc = (int32_t)e.msc;
  T acc = T(0);
  for (uint32_t i = threadIdx.x; i < n; i += MULTI_THREADS) {
    const uint32_t r = i / C, c = i - r * C;
    if (e.m != nullptr && e.m[r * msr + c * msc] == 0) continue;
    T x;
    if constexpr (DIST == PA_SITE_IDENTITY) {
      x = v[r * vsr + c * vsc];
    } else {
      const T bb = NParams<DIST>::n > 1 ? b[r * bsr + c * bsc] : T(0);
      x = Fam<DIST, T>::lp(v[r * vsr + c * vsc], a[r * asr + c * asc], bb);
    }
    acc += x;      // at most PA_MULTI_MAX_ELEMS / 1024 = 64 terms per thread; fp64 across threads
  }
  return (double)acc;
}

template <typename T>
__global__ __launch_bounds__(MULTI_THREADS) void multi_sum_kernel(const MultiArgs args_by_value,
                                                                  T* __restrict__ out,
                                                                  double coef_all, int accumulate) {
  __shared__ double smem[16];
  const MultiArgs& args = *kernarg_table<MultiArgs>();
  double acc = 0.0;
  for (int k = 0; k < args.n; ++k) {
    const EntryDev& e = args.e[k];
    double s = 0.0;
    switch (e.dist) {
      case PA_DIST_NORMAL: s = entry_sum<PA_DIST_NORMAL, T>(e); break;
      case PA_DIST_BERNOULLI_LOGITS: s = entry_sum<PA_DIST_BERNOULLI_LOGITS, T>(e); break;
      case PA_DIST_HALF_CAUCHY: s = entry_sum<PA_DIST_HALF_CAUCHY, T>(e); break;
      case PA_DIST_LOG_NORMAL: s = entry_sum<PA_DIST_LOG_NORMAL, T>(e); break;
      case PA_DIST_EXPONENTIAL: s = entry_sum<PA_DIST_EXPONENTIAL, T>(e); break;
      case PA_DIST_HALF_NORMAL: s = entry_sum<PA_DIST_HALF_NORMAL, T>(e); break;
      case PA_SITE_NONE: s = 0.0; break;
      default: s = entry_sum<PA_SITE_IDENTITY, T>(e); break;
    }
    acc += e.coef * s;
  }
  const double t = block_sum_f64(acc, smem);
  if (threadIdx.x == 0) {
    const double base = accumulate ? (double)*out : 0.0;
    *out = (T)(base + coef_all * t);
  }
}

// gradient of one entry w.r.t. operand WHICH (0 value, 1 p0, 2 p1) at element (r, c)
template <int DIST, typename T>
__device__ __forceinline__ T entry_grad_at(const EntryDev& e, int which, uint32_t r, uint32_t c) {
  if (e.m != nullptr && e.m[r * (int32_t)e.msr + c * (int32_t)e.msc] == 0) return T(0);
  if constexpr (DIST == PA_SITE_IDENTITY) {
    return T(1);
  } else {
    const T* v = (const T*)e.v;
    const T* a = (const T*)e.a;
    const T* b = (const T*)e.b;
    const T bb = NParams<DIST>::n > 1 ? b[r * (int32_t)e.bsr + c * (int32_t)e.bsc] : T(0);
    T gv, ga, gb;
    Fam<DIST, T>::grad(v[r * (int32_t)e.vsr + c * (int32_t)e.vsc],
                       a[r * (int32_t)e.asr + c * (int32_t)e.asc], bb, gv, ga, gb);
    return which == 0 ? gv : (which == 1 ? ga : gb);
  }
}

constexpr int GRAD_THREADS = 256;

// out (contiguous [rows or 1, cols or 1]) = w * reduce(grad) for operand `which`.  Reductions over
// a dimension are spread over thread groups and combined through LDS in a fixed order.
template <int DIST, typename T>
__device__ void entry_grad_operand(const EntryDev& e, int which, int64_t sr, int64_t sc, T* out,
                                   double w, double* red /* [GRAD_THREADS] */,
                                   bool accumulate = false) {
  const uint32_t R = (uint32_t)e.rows, C = (uint32_t)e.cols, t = threadIdx.x;
  const bool red_r = (sr == 0 && R > 1), red_c = (sc == 0 && C > 1);
  if (!red_r && !red_c) {
#pragma unroll 4
    for (uint32_t i = t; i < R * C; i += GRAD_THREADS) {
      const uint32_t r = i / C, c = i - r * C;
      const T x = (T)(w * (double)entry_grad_at<DIST, T>(e, which, r, c));
      out[i] = accumulate ? out[i] + x : x;
    }
  } else if (red_r != red_c) {
    // K = kept dimension (its index is the output index), L = summed dimension
    const uint32_t K = red_r ? C : R, L = red_r ? R : C;
    const uint32_t tk = K < GRAD_THREADS ? K : GRAD_THREADS;   // threads along the kept dim
    const uint32_t ng = GRAD_THREADS / tk;                     // groups along the summed dim
    const uint32_t k0 = t % tk, g = t / tk;
    for (uint32_t kb = 0; kb < K; kb += tk) {
      const uint32_t k = kb + k0;
      T acc = T(0);
      if (g < ng && k < K) {
#pragma unroll 4
        for (uint32_t l = g; l < L; l += ng)
          acc += red_r ? entry_grad_at<DIST, T>(e, which, l, k) : entry_grad_at<DIST, T>(e, which, k, l);
      }
      __syncthreads();
      red[t] = (double)acc;
      __syncthreads();
      if (g == 0 && k < K) {
        double s = 0.0;
        for (uint32_t j = 0; j < ng; ++j) s += red[j * tk + k0];
        out[k] = (T)(w * s) + (accumulate ? out[k] : T(0));
      }
    }
    __syncthreads();
  } else {                                // scalar operand
    T acc = T(0);
    for (uint32_t i = t; i < R * C; i += GRAD_THREADS) {
      const uint32_t r = i / C, c = i - r * C;
      acc += entry_grad_at<DIST, T>(e, which, r, c);
    }
    const double tot = block_sum_f64((double)acc, red);
    if (t == 0) out[0] = (T)(w * tot) + (accumulate ? out[0] : T(0));
    __syncthreads();
  }
}

// value gradient of entry e into out (run-time family dispatch: chain members differ in family)
template <typename T>
__device__ void value_grad_pass(const EntryDev& e, T* out, double w, double* smem, bool accumulate) {
  switch (e.dist) {
#define PA_VG(D_) case D_: entry_grad_operand<D_, T>(e, 0, e.vsr, e.vsc, out, w, smem, accumulate); break;
    PA_VG(PA_DIST_NORMAL) PA_VG(PA_DIST_BERNOULLI_LOGITS) PA_VG(PA_DIST_HALF_CAUCHY)
    PA_VG(PA_DIST_LOG_NORMAL) PA_VG(PA_DIST_EXPONENTIAL) PA_VG(PA_DIST_HALF_NORMAL)
    PA_VG(PA_SITE_IDENTITY)
#undef PA_VG
    default:   // PA_SITE_NONE: zero gradient
      if (!accumulate) {
        const bool red_r = (e.vsr == 0 && e.rows > 1), red_c = (e.vsc == 0 && e.cols > 1);
        const uint32_t n = (uint32_t)((red_r ? 1 : e.rows) * (red_c ? 1 : e.cols));
        for (uint32_t i = threadIdx.x; i < n; i += GRAD_THREADS) out[i] = T(0);
      }
      break;
  }
}

template <int DIST, typename T>
__device__ void param_grads(const EntryDev& e, double w, double* smem) {
  if constexpr (DIST < PA_DIST_COUNT) {
    if ((e.need & PA_NEED_P0) && e.da) entry_grad_operand<DIST, T>(e, 1, e.asr, e.asc, (T*)e.da, w, smem);
    if (NParams<DIST>::n > 1 && (e.need & PA_NEED_P1) && e.db)
      entry_grad_operand<DIST, T>(e, 2, e.bsr, e.bsc, (T*)e.db, w, smem);
  }
}

template <typename T>
__global__ __launch_bounds__(GRAD_THREADS) void multi_grad_kernel(const MultiArgs args_by_value,
                                                                  const T* __restrict__ g,
                                                                  double coef_all) {
  __shared__ double smem[GRAD_THREADS];
  const MultiArgs& args = *kernarg_table<MultiArgs>();
  const EntryDev& e = args.e[blockIdx.x];
  const double gw = (double)g[0] * coef_all;
  if ((e.need & PA_NEED_VALUE) && e.dv && !(e.need & PA_VALUE_BY_CHAIN)) {
    T* out = (T*)e.dv;
    value_grad_pass<T>(e, out, gw * e.coef, smem, false);
    for (int k = e.chain_next; k >= 0; k = args.e[k].chain_next) {   // same value tensor, same frame
      __syncthreads();
      value_grad_pass<T>(args.e[k], out, gw * args.e[k].coef, smem, true);
    }
    if (e.xg != nullptr) {
      __syncthreads();
      const T* xg = (const T*)e.xg;
      const T xw = (T)(gw * e.xcoef);
      const uint32_t n = (uint32_t)(e.rows * e.cols);
      for (uint32_t i = threadIdx.x; i < n; i += GRAD_THREADS) out[i] += xw * xg[i];
    }
    __syncthreads();
  }
  const double w = gw * e.coef;
  switch (e.dist) {
    case PA_DIST_NORMAL: param_grads<PA_DIST_NORMAL, T>(e, w, smem); break;
    case PA_DIST_BERNOULLI_LOGITS: param_grads<PA_DIST_BERNOULLI_LOGITS, T>(e, w, smem); break;
    case PA_DIST_HALF_CAUCHY: param_grads<PA_DIST_HALF_CAUCHY, T>(e, w, smem); break;
    case PA_DIST_LOG_NORMAL: param_grads<PA_DIST_LOG_NORMAL, T>(e, w, smem); break;
    case PA_DIST_EXPONENTIAL: param_grads<PA_DIST_EXPONENTIAL, T>(e, w, smem); break;
    case PA_DIST_HALF_NORMAL: param_grads<PA_DIST_HALF_NORMAL, T>(e, w, smem); break;
    default: break;
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
                     !(s.dist == PA_DIST_NORMAL || s.dist == PA_DIST_LOG_NORMAL) || s.p1.ptr,
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

template <typename T> __device__ __forceinline__ T softplus_t(T x) {
  // torch.nn.functional.softplus (threshold 20): x for large x, log1p(exp(x)) otherwise
  return x > T(20) ? x : t_log1p(t_exp(x));
}

template <typename T>
__global__ __launch_bounds__(256) void meanfield_sample_kernel(const MfArgs args_by_value,
                                                               int64_t P, uint64_t seed,
                                                               const uint64_t* __restrict__ offset_dev) {
  const MfSiteDev& s = kernarg_table<MfArgs>()->s[blockIdx.y];
  const uint64_t off = s.offset + (offset_dev ? *offset_dev : 0);
  const T* loc = (const T*)s.loc;
  const T* rho = (const T*)s.rho;
  T* z = (T*)s.z;
  T* eps = (T*)s.eps;
  T* sc = (T*)s.scale;
  T* lo = (T*)s.loc_out;
  const int64_t total = P * s.n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i % s.n;
    T e;
    if constexpr (sizeof(T) == 4) e = philox_normal_f32(seed, off, (uint64_t)i);
    else e = philox_normal_f64(seed, off, (uint64_t)i);
    const T sp = softplus_t<T>(rho[c]);
    eps[i] = e;
    z[i] = loc[c] + sp * e;
    if (i < s.n) {
      sc[c] = sp;
      lo[c] = loc[c];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void meanfield_sample_bwd_kernel(const MfArgs args_by_value,
                                                                   int64_t P) {
  __shared__ double red_l[256], red_s[256];
  const MfSiteDev& s = kernarg_table<MfArgs>()->s[blockIdx.x];
  const T* dz = (const T*)s.d_z;
  const T* eps = (const T*)s.eps;
  const T* dsc = (const T*)s.d_scale;
  const T* dlo = (const T*)s.d_loc_out;
  const T* rho = (const T*)s.rho;
  T* dloc = (T*)s.d_loc;
  T* drho = (T*)s.d_rho;
  const uint32_t n = (uint32_t)s.n, t = threadIdx.x, PP = (uint32_t)P;
  const uint32_t tk = n < 256 ? n : 256, ng = 256 / tk, c0 = t % tk, g = t / tk;
  for (uint32_t cb = 0; cb < n; cb += tk) {
    const uint32_t c = cb + c0;
    T al = T(0), as = T(0);
    if (dz != nullptr && g < ng && c < n) {
#pragma unroll 8
      for (uint32_t p = g; p < PP; p += ng) {
        const T gz = dz[p * n + c];
        al += gz;
        as += gz * eps[p * n + c];
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
      if (dsc != nullptr) ss += (double)dsc[c];
      if (dlo != nullptr) sl += (double)dlo[c];
      const double x = (double)rho[c];
      const double sig = x > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-x));   // d softplus / d x
      if (dloc) dloc[c] = (T)sl + (s.accumulate ? dloc[c] : T(0));
      if (drho) drho[c] = (T)(ss * sig) + (s.accumulate ? drho[c] : T(0));
    }
  }
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

extern "C" {

int pa_multi_log_prob_sum(int dtype, void* out_total, const pa_site_entry* entries, int n,
                          double coef_all, int accumulate, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_multi_log_prob_sum: bad dtype %d", dtype);
  PA_REQUIRE(out_total != nullptr, "pa_multi_log_prob_sum: NULL output");
  pa::MultiArgs args;
  int rc = pa::to_dev(entries, n, &args, "pa_multi_log_prob_sum");
  if (rc != PA_OK) return rc;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::multi_sum_kernel<float>), dim3(1), dim3(pa::MULTI_THREADS), 0, s,
                       args, (float*)out_total, coef_all, accumulate);
  else
    hipLaunchKernelGGL((pa::multi_sum_kernel<double>), dim3(1), dim3(pa::MULTI_THREADS), 0, s,
                       args, (double*)out_total, coef_all, accumulate);
  return pa::check_launch("multi_sum_kernel");
}

int pa_multi_log_prob_grad(int dtype, const void* g, const pa_site_entry* entries, int n,
                           double coef_all, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_multi_log_prob_grad: bad dtype %d", dtype);
  PA_REQUIRE(g != nullptr, "pa_multi_log_prob_grad: NULL upstream gradient");
  pa::MultiArgs args;
  int rc = pa::to_dev(entries, n, &args, "pa_multi_log_prob_grad");
  if (rc != PA_OK) return rc;
  if (n == 0) return PA_OK;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::multi_grad_kernel<float>), dim3((unsigned)n), dim3(pa::GRAD_THREADS), 0, s,
                       args, (const float*)g, coef_all);
  else
    hipLaunchKernelGGL((pa::multi_grad_kernel<double>), dim3((unsigned)n), dim3(pa::GRAD_THREADS), 0, s,
                       args, (const double*)g, coef_all);
  return pa::check_launch("multi_grad_kernel");
}

int pa_meanfield_normal_sample(int dtype, const pa_mf_site* sites, int nsites, int64_t P,
                               uint64_t seed, const uint64_t* offset_dev, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_meanfield_normal_sample: bad dtype %d", dtype);
  PA_REQUIRE(P >= 1, "pa_meanfield_normal_sample: P = %lld", (long long)P);
  pa::MfArgs args;
  int rc = pa::mf_to_dev(sites, nsites, &args, "pa_meanfield_normal_sample");
  if (rc != PA_OK) return rc;
  if (nsites == 0) return PA_OK;
  int64_t maxn = 0;
  for (int k = 0; k < nsites; ++k) {
    PA_REQUIRE(sites[k].n == 0 || (sites[k].loc && sites[k].rho && sites[k].z && sites[k].scale &&
                                   sites[k].loc_out && sites[k].eps),
               "pa_meanfield_normal_sample: site %d: NULL pointer", k);
    if (sites[k].n > maxn) maxn = sites[k].n;
  }
  int64_t gx = (P * maxn + 255) / 256;
  if (gx < 1) gx = 1;
  const int64_t cap = (int64_t)pa::cu_count() * 4;
  if (gx > cap) gx = cap;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::meanfield_sample_kernel<float>), dim3((unsigned)gx, (unsigned)nsites),
                       dim3(256), 0, s, args, P, seed, offset_dev);
  else
    hipLaunchKernelGGL((pa::meanfield_sample_kernel<double>), dim3((unsigned)gx, (unsigned)nsites),
                       dim3(256), 0, s, args, P, seed, offset_dev);
  return pa::check_launch("meanfield_sample_kernel");
}

int pa_meanfield_normal_sample_bwd(int dtype, const pa_mf_site* sites, int nsites, int64_t P,
                                   pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_meanfield_normal_sample_bwd: bad dtype %d",
             dtype);
  PA_REQUIRE(P >= 1, "pa_meanfield_normal_sample_bwd: P = %lld", (long long)P);
  pa::MfArgs args;
  int rc = pa::mf_to_dev(sites, nsites, &args, "pa_meanfield_normal_sample_bwd");
  if (rc != PA_OK) return rc;
  if (nsites == 0) return PA_OK;
  for (int k = 0; k < nsites; ++k)
    PA_REQUIRE(sites[k].n == 0 || (sites[k].rho && sites[k].eps),
               "pa_meanfield_normal_sample_bwd: site %d: NULL pointer", k);
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::meanfield_sample_bwd_kernel<float>), dim3((unsigned)nsites), dim3(256),
                       0, s, args, P);
  else
    hipLaunchKernelGGL((pa::meanfield_sample_bwd_kernel<double>), dim3((unsigned)nsites),
                       dim3(256), 0, s, args, P);
  return pa::check_launch("meanfield_sample_bwd_kernel");
}

}  // extern "C"
