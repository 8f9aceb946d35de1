// dist.hip -- element-wise sample-site kernels: log_prob, fused log_prob+scale_and_mask+plate
// sum, their gradient, and the reparameterised Normal draw.  
//
// All operands are 2-D strided views [rows, cols] (stride 0 = broadcast), so the stride-0
// expanded parameters Pyro hands over (torch_distribution.py:483-488) are never materialised.
// These kernels are HBM-bound: one read of each operand, one write (or none, for the fused
// sum).  cols is the fast (coalesced) axis; every thread handles ITEMS columns 256 apart.
#include "common.h"
#include "dist_fam.h"

namespace pa {

constexpr int DIST_THREADS = 256;
constexpr int DIST_ITEMS = 4;

// ---- kernels ----------------------------------------------------------------------------------
// Element-wise kernels: a workgroup covers RPB = 256 / TC rows x (TC * ITEMS) columns, TC = the
// power of two >= cols / ITEMS (capped at 256): long rows get one row per workgroup and 1024 columns
// per chunk, short rows (a [S*T*K, 88] emission table) share a workgroup instead of leaving most of
// its threads idle behind a one-row-per-workgroup grid.
struct ElemGeom {
  int tc;            // threads along the columns (power of two)
  int64_t bx;        // column chunks per row block
  int64_t grid;
};
static inline ElemGeom elem_geom(int64_t rows, int64_t cols) {
  int64_t want = (cols + DIST_ITEMS - 1) / DIST_ITEMS;
  int tc = 1;
  while (tc < DIST_THREADS && tc < want) tc *= 2;
  ElemGeom g;
  g.tc = tc;
  g.bx = (cols + (int64_t)tc * DIST_ITEMS - 1) / ((int64_t)tc * DIST_ITEMS);
  const int64_t rpb = DIST_THREADS / tc;
  g.grid = ((rows + rpb - 1) / rpb) * g.bx;
  return g;
}

template <int DIST, typename T>
__global__ __launch_bounds__(DIST_THREADS) void log_prob_kernel(T* __restrict__ out, ViewT<T> v,
                                                                ViewT<T> a, ViewT<T> b,
                                                                int64_t rows, int64_t cols,
                                                                int64_t bx, int tc) {
  const int rpb = DIST_THREADS / tc;
  const int64_t row = (blockIdx.x / bx) * rpb + threadIdx.x / tc, chunk = blockIdx.x % bx;
  if (row >= rows) return;
  const int64_t c0 = chunk * ((int64_t)tc * DIST_ITEMS) + threadIdx.x % tc;
#pragma unroll
  for (int k = 0; k < DIST_ITEMS; ++k) {
    const int64_t c = c0 + (int64_t)k * tc;
    if (c < cols) {
      T bb = NParams<DIST>::n > 1 ? b.at(row, c) : T(0);
      out[row * cols + c] = Fam<DIST, T>::lp(v.at(row, c), a.at(row, c), bb);
    }
  }
}

template <int DIST, typename T>
__global__ __launch_bounds__(DIST_THREADS) void log_prob_sum_kernel(
    double* __restrict__ partial, ViewT<T> v, ViewT<T> a, ViewT<T> b, ViewT<uint8_t> m, T scale,
    int64_t rows, int64_t cols, int64_t bx, int64_t iters) {
  __shared__ double smem[16];
  const int64_t row = blockIdx.x / bx, chunk = blockIdx.x % bx;
  T acc = T(0);
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t c0 = (it * bx + chunk) * (DIST_THREADS * DIST_ITEMS) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < DIST_ITEMS; ++k) {
      const int64_t c = c0 + k * DIST_THREADS;
      if (c < cols) {
        T bb = NParams<DIST>::n > 1 ? b.at(row, c) : T(0);
        T lp = Fam<DIST, T>::lp(v.at(row, c), a.at(row, c), bb) * scale;
        // scale_and_mask (distributions/util.py:311-328): where(mask, tensor*scale, 0)
        if (m.p != nullptr && m.at(row, c) == 0) lp = T(0);
        acc += lp;
      }
    }
  }
  double t = block_sum_f64((double)acc, smem);
  if (threadIdx.x == 0) partial[row * bx + chunk] = t;
}

// Small sites (rows*cols <= SMALL_ELEMS: global latents, a few thousand elements) in ONE launch:
// wave w of the single workgroup reduces rows w, w+4, ... (lane-strided columns, butterfly), and
// the four per-wave totals are combined in a fixed order => deterministic, no workspace.
constexpr int64_t SMALL_ELEMS = 32768;
template <int DIST, typename T>
__global__ __launch_bounds__(256) void log_prob_sum_small_kernel(
    T* __restrict__ out_rowsum, T* __restrict__ out_total, ViewT<T> v, ViewT<T> a, ViewT<T> b,
    ViewT<uint8_t> m, T scale, int64_t rows, int64_t cols) {
  __shared__ double wtot[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double tot = 0.0;
  for (int64_t row = wave; row < rows; row += 4) {
    T acc = T(0);
    for (int64_t c = lane; c < cols; c += 64) {
      T bb = NParams<DIST>::n > 1 ? b.at(row, c) : T(0);
      T lp = Fam<DIST, T>::lp(v.at(row, c), a.at(row, c), bb) * scale;
      if (m.p != nullptr && m.at(row, c) == 0) lp = T(0);
      acc += lp;
    }
    const double t = wave_sum((double)acc);
    if (lane == 0) out_rowsum[row] = (T)t;
    tot += t;
  }
  if (lane == 0) wtot[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0 && out_total != nullptr)
    *out_total = (T)(((wtot[0] + wtot[1]) + wtot[2]) + wtot[3]);
}

// out_total = sum_r out_rowsum[r] (large path only), one workgroup, fixed order
template <typename T>
__global__ __launch_bounds__(256) void rowsum_total_kernel(T* __restrict__ out_total,
                                                           const T* __restrict__ rowsum,
                                                           int64_t rows) {
  __shared__ double smem[16];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < rows; i += 256) acc += (double)rowsum[i];
  const double t = block_sum_f64(acc, smem);
  if (threadIdx.x == 0) *out_total = (T)t;
}

template <typename T>
__global__ __launch_bounds__(256) void rowsum_finalize_kernel(T* __restrict__ out,
                                                              const double* __restrict__ partial,
                                                              int64_t rows, int64_t bx) {
  // one wave per row; fixed lane->chunk assignment and butterfly order => deterministic
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  double t = 0.0;
  for (int64_t i = lane; i < bx; i += 64) t += partial[row * bx + i];
  t = wave_sum(t);
  if (lane == 0) out[row] = (T)t;
}

template <int DIST, typename T>
__global__ __launch_bounds__(DIST_THREADS) void log_prob_grad_kernel(
    T* __restrict__ dv, T* __restrict__ da, T* __restrict__ db, ViewT<T> g, ViewT<T> v, ViewT<T> a,
    ViewT<T> b, ViewT<uint8_t> m, T scale, int64_t rows, int64_t cols, int64_t bx, int tc) {
  const int rpb = DIST_THREADS / tc;
  const int64_t row = (blockIdx.x / bx) * rpb + threadIdx.x / tc, chunk = blockIdx.x % bx;
  if (row >= rows) return;
  const int64_t c0 = chunk * ((int64_t)tc * DIST_ITEMS) + threadIdx.x % tc;
#pragma unroll
  for (int k = 0; k < DIST_ITEMS; ++k) {
    const int64_t c = c0 + (int64_t)k * tc;
    if (c < cols) {
      T bb = NParams<DIST>::n > 1 ? b.at(row, c) : T(0);
      T gv, ga, gb;
      Fam<DIST, T>::grad(v.at(row, c), a.at(row, c), bb, gv, ga, gb);
      T w = g.at(row, c) * scale;
      const bool keep = (m.p == nullptr) || (m.at(row, c) != 0);
      const int64_t o = row * cols + c;
      // masked-out elements get an exact 0 gradient (torch.where backward), even if the
      // partial derivative itself is inf/NaN there.
      if (dv) dv[o] = keep ? w * gv : T(0);
      if (da) da[o] = keep ? w * ga : T(0);
      if (db) db[o] = keep ? w * gb : T(0);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(DIST_THREADS) void normal_rsample_kernel(
    T* __restrict__ out, T* __restrict__ eps_out, ViewT<T> loc, ViewT<T> scale, int64_t rows,
    int64_t cols, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev) {
  if (offset_dev) offset += *offset_dev;
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i - r * cols;
    T e;
    if constexpr (sizeof(T) == 4)
      e = philox_normal_f32(seed, offset, (uint64_t)i);
    else
      e = philox_normal_f64(seed, offset, (uint64_t)i);
    if (eps_out) eps_out[i] = e;
    out[i] = loc.at(r, c) + scale.at(r, c) * e;
  }
}

// ---- host dispatch ----------------------------------------------------------------------------
static int check_common(const char* who, int dist, int dtype, int64_t rows, int64_t cols) {
  PA_REQUIRE(dist >= 0 && dist < PA_DIST_COUNT, "%s: unknown distribution id %d", who, dist);
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "%s: bad dtype %d", who, dtype);
  PA_REQUIRE(rows >= 0 && cols >= 0, "%s: negative shape [%lld,%lld]", who, (long long)rows,
             (long long)cols);
  PA_REQUIRE(rows < (int64_t(1) << 31) && cols < (int64_t(1) << 40), "%s: shape too large", who);
  return PA_OK;
}

static inline int64_t chunks_of(int64_t cols) {
  return (cols + DIST_THREADS * DIST_ITEMS - 1) / (DIST_THREADS * DIST_ITEMS);
}

// number of column-chunk blocks per row used by the fused sum (deterministic function of shape)
static inline int64_t sum_bx(int64_t rows, int64_t cols) {
  int64_t bx = chunks_of(cols);
  int64_t cap = (int64_t)2048 / (rows > 0 ? rows : 1);
  if (cap < 1) cap = 1;
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  return bx;
}

template <typename T>
static int log_prob_t(int dist, T* out, pa_view2d value, pa_view2d p0, pa_view2d p1, int64_t rows,
                      int64_t cols, hipStream_t s) {
  const ElemGeom gm = elem_geom(rows, cols);
  PA_REQUIRE(gm.grid < (int64_t(1) << 31), "log_prob: grid too large");
  auto v = as_view<T>(value), a = as_view<T>(p0), b = as_view<T>(p1);
  PA_DISPATCH_DIST(dist, T,
                   hipLaunchKernelGGL((log_prob_kernel<D_, T>), dim3((unsigned)gm.grid),
                                      dim3(DIST_THREADS), 0, s, out, v, a, b, rows, cols, gm.bx,
                                      gm.tc));
  return check_launch("log_prob_kernel");
}

template <typename T>
static int log_prob_sum_t(int dist, T* out, T* out_total, pa_view2d value, pa_view2d p0,
                          pa_view2d p1, pa_view2d mask, double scale, int64_t rows, int64_t cols,
                          double* ws, hipStream_t s) {
  if (rows * cols <= SMALL_ELEMS) {
    auto v = as_view<T>(value), a = as_view<T>(p0), b = as_view<T>(p1);
    auto m = as_view<uint8_t>(mask);
    PA_DISPATCH_DIST(dist, T,
                     hipLaunchKernelGGL((log_prob_sum_small_kernel<D_, T>), dim3(1), dim3(256), 0,
                                        s, out, out_total, v, a, b, m, (T)scale, rows, cols));
    return check_launch("log_prob_sum_small_kernel");
  }
  const int64_t bx = sum_bx(rows, cols);
  const int64_t iters = (chunks_of(cols) + bx - 1) / bx;
  PA_REQUIRE(rows * bx < (int64_t(1) << 31), "log_prob_sum: grid too large");
  auto v = as_view<T>(value), a = as_view<T>(p0), b = as_view<T>(p1);
  auto m = as_view<uint8_t>(mask);
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_SITE_SUM, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  PA_DISPATCH_DIST(dist, T,
                   hipLaunchKernelGGL((log_prob_sum_kernel<D_, T>), dim3((unsigned)(rows * bx)),
                                      dim3(DIST_THREADS), 0, s, ws, v, a, b, m, (T)scale, rows,
                                      cols, bx, iters));
  if (br) (void)hipEventRecord(ev1, s);
  int rc = check_launch("log_prob_sum_kernel");
  if (rc != PA_OK) return rc;
  hipLaunchKernelGGL((rowsum_finalize_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     s, out, ws, rows, bx);
  rc = check_launch("rowsum_finalize_kernel");
  if (rc != PA_OK || out_total == nullptr) return rc;
  hipLaunchKernelGGL((rowsum_total_kernel<T>), dim3(1), dim3(256), 0, s, out_total, out, rows);
  return check_launch("rowsum_total_kernel");
}

template <typename T>
static int log_prob_grad_t(int dist, T* dv, T* da, T* db, pa_view2d g, pa_view2d value,
                           pa_view2d p0, pa_view2d p1, pa_view2d mask, double scale, int64_t rows,
                           int64_t cols, hipStream_t s) {
  const ElemGeom gm = elem_geom(rows, cols);
  PA_REQUIRE(gm.grid < (int64_t(1) << 31), "log_prob_grad: grid too large");
  auto gg = as_view<T>(g), v = as_view<T>(value), a = as_view<T>(p0), b = as_view<T>(p1);
  auto m = as_view<uint8_t>(mask);
  PA_DISPATCH_DIST(dist, T,
                   hipLaunchKernelGGL((log_prob_grad_kernel<D_, T>), dim3((unsigned)gm.grid),
                                      dim3(DIST_THREADS), 0, s, dv, da, db, gg, v, a, b, m,
                                      (T)scale, rows, cols, gm.bx, gm.tc));
  return check_launch("log_prob_grad_kernel");
}

static int nparams(int dist) { return dist_nparams(dist); }

}  // namespace pa

extern "C" {

int pa_dist_log_prob(int dist, int dtype, void* out, pa_view2d value, pa_view2d p0, pa_view2d p1,
                     int64_t rows, int64_t cols, pa_stream_t stream) {
  int rc = pa::check_common("pa_dist_log_prob", dist, dtype, rows, cols);
  if (rc != PA_OK) return rc;
  if (rows == 0 || cols == 0) return PA_OK;
  PA_REQUIRE(out && value.ptr && p0.ptr, "pa_dist_log_prob: NULL operand");
  PA_REQUIRE(pa::nparams(dist) < 2 || p1.ptr, "pa_dist_log_prob: family needs p1");
  if (dtype == PA_F32)
    return pa::log_prob_t<float>(dist, (float*)out, value, p0, p1, rows, cols,
                                 pa::as_stream(stream));
  return pa::log_prob_t<double>(dist, (double*)out, value, p0, p1, rows, cols,
                                pa::as_stream(stream));
}

size_t pa_dist_log_prob_sum_workspace(int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0 || rows * cols <= pa::SMALL_ELEMS) return 0;
  return (size_t)(rows * pa::sum_bx(rows, cols)) * sizeof(double);
}

int pa_dist_log_prob_sum(int dist, int dtype, void* out_rowsum, void* out_total, pa_view2d value,
                         pa_view2d p0, pa_view2d p1, pa_view2d mask, double scale, int64_t rows,
                         int64_t cols, void* workspace, size_t workspace_bytes,
                         pa_stream_t stream) {
  int rc = pa::check_common("pa_dist_log_prob_sum", dist, dtype, rows, cols);
  if (rc != PA_OK) return rc;
  const size_t esz = dtype == PA_F32 ? 4 : 8;
  if (rows == 0 || cols == 0) {
    // empty plate: the sum over nothing is 0 (torch: tensor.sum() of an empty tensor)
    hipError_t e = hipSuccess;
    if (rows > 0) {
      PA_REQUIRE(out_rowsum, "pa_dist_log_prob_sum: NULL output");
      e = hipMemsetAsync(out_rowsum, 0, (size_t)rows * esz, pa::as_stream(stream));
    }
    if (e == hipSuccess && out_total) e = hipMemsetAsync(out_total, 0, esz, pa::as_stream(stream));
    return e == hipSuccess ? PA_OK : pa::fail(PA_ERR_LAUNCH, "memset: %s", hipGetErrorString(e));
  }
  PA_REQUIRE(out_rowsum, "pa_dist_log_prob_sum: NULL output");
  PA_REQUIRE(value.ptr && p0.ptr, "pa_dist_log_prob_sum: NULL operand");
  PA_REQUIRE(pa::nparams(dist) < 2 || p1.ptr, "pa_dist_log_prob_sum: family needs p1");
  const size_t need = pa_dist_log_prob_sum_workspace(rows, cols);
  PA_REQUIRE(need == 0 || (workspace && workspace_bytes >= need),
             "pa_dist_log_prob_sum: workspace too small (%zu < %zu)", workspace_bytes, need);
  if (dtype == PA_F32)
    return pa::log_prob_sum_t<float>(dist, (float*)out_rowsum, (float*)out_total, value, p0, p1,
                                     mask, scale, rows, cols, (double*)workspace,
                                     pa::as_stream(stream));
  return pa::log_prob_sum_t<double>(dist, (double*)out_rowsum, (double*)out_total, value, p0, p1,
                                    mask, scale, rows, cols, (double*)workspace,
                                    pa::as_stream(stream));
}

int pa_dist_log_prob_grad(int dist, int dtype, void* d_value, void* d_p0, void* d_p1, pa_view2d g,
                          pa_view2d value, pa_view2d p0, pa_view2d p1, pa_view2d mask, double scale,
                          int64_t rows, int64_t cols, pa_stream_t stream) {
  int rc = pa::check_common("pa_dist_log_prob_grad", dist, dtype, rows, cols);
  if (rc != PA_OK) return rc;
  if (rows == 0 || cols == 0) return PA_OK;
  PA_REQUIRE(g.ptr && value.ptr && p0.ptr, "pa_dist_log_prob_grad: NULL operand");
  PA_REQUIRE(pa::nparams(dist) < 2 || p1.ptr, "pa_dist_log_prob_grad: family needs p1");
  if (dtype == PA_F32)
    return pa::log_prob_grad_t<float>(dist, (float*)d_value, (float*)d_p0, (float*)d_p1, g, value,
                                      p0, p1, mask, scale, rows, cols, pa::as_stream(stream));
  return pa::log_prob_grad_t<double>(dist, (double*)d_value, (double*)d_p0, (double*)d_p1, g, value,
                                     p0, p1, mask, scale, rows, cols, pa::as_stream(stream));
}

int pa_normal_rsample(int dtype, void* out, void* eps_out, pa_view2d loc, pa_view2d scale,
                      int64_t rows, int64_t cols, uint64_t seed, uint64_t offset,
                      const uint64_t* offset_dev, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_normal_rsample: bad dtype %d", dtype);
  PA_REQUIRE(rows >= 0 && cols >= 0, "pa_normal_rsample: negative shape");
  const int64_t n = rows * cols;
  if (n == 0) return PA_OK;
  PA_REQUIRE(out && loc.ptr && scale.ptr, "pa_normal_rsample: NULL operand");
  int64_t grid = (n + 255) / 256;
  const int64_t cap = (int64_t)pa::cu_count() * 8;
  if (grid > cap) grid = cap;
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::normal_rsample_kernel<float>), dim3((unsigned)grid), dim3(256), 0,
                       pa::as_stream(stream), (float*)out, (float*)eps_out,
                       pa::as_view<float>(loc), pa::as_view<float>(scale), rows, cols, seed, offset,
                       offset_dev);
  else
    hipLaunchKernelGGL((pa::normal_rsample_kernel<double>), dim3((unsigned)grid), dim3(256), 0,
                       pa::as_stream(stream), (double*)out, (double*)eps_out,
                       pa::as_view<double>(loc), pa::as_view<double>(scale), rows, cols, seed,
                       offset, offset_dev);
  return pa::check_launch("normal_rsample_kernel");
}

}  // extern "C"
