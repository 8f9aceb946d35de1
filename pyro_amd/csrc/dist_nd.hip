// dist_nd.hip -- sample-site kernels for operands that broadcast along a MIDDLE dimension.
//
// dist.hip handles operands that are 2-D strided views of the site's frame; a plated latent under
// the particle plate (w[P, G, D] scored against mu[P, 1, D]: BASELINE config 5, the reference's
// `Normal(mu, tau).log_prob(w)` of pyro/poutine/trace_struct.py:248-288 with its expanded
// parameters, torch_distribution.py:483-488) is not one, and used to be scored from materialised
// broadcasts with the operand gradients reduced by generic strided reductions.  Here:
//   * pa_dist_log_prob_sum_nd : sum(scale_and_mask(log_prob)) over a frame of up to 4 dims, every
//     operand with its own strides (0 = broadcast); flat coalesced iteration, the index is taken
//     apart with 32-bit divisions (a few dozen VALU instructions per element: the kernels stay
//     HBM-bound); block partials in fp64, fixed order;
//   * pa_dist_log_prob_grad_nd: the un-reduced operand gradients, contiguous over the frame;
//   * pa_sum_to_nd            : [A, R, B] -> [A, B], the reduction that brings such a gradient
//     back to a broadcast operand's shape (leading, middle or trailing dims), deterministic,
//     split over R so that the whole chip streams the input once.
#include "common.h"
#include "dist_fam.h"

namespace pa {

constexpr int ND_THREADS = 256;
constexpr int ND_ITEMS = 4;
constexpr int ND_MAX_BLOCKS = 2048;

struct NdFrame {
  uint32_t n1, n2, n3;          // sizes of dims 1..3 (dim 0 follows from the flat index)
  uint32_t total;
};
struct NdStrides {
  int32_t s[4];
};
__device__ __forceinline__ void nd_index(const NdFrame& f, uint32_t i, uint32_t idx[4]) {
  const uint32_t t2 = i / f.n3;
  idx[3] = i - t2 * f.n3;
  const uint32_t t1 = t2 / f.n2;
  idx[2] = t2 - t1 * f.n2;
  const uint32_t t0 = t1 / f.n1;
  idx[1] = t1 - t0 * f.n1;
  idx[0] = t0;
}
__device__ __forceinline__ int32_t nd_offset(const NdStrides& st, const uint32_t idx[4]) {
  return (int32_t)idx[0] * st.s[0] + (int32_t)idx[1] * st.s[1] + (int32_t)idx[2] * st.s[2] +
         (int32_t)idx[3] * st.s[3];
}

template <int DIST, typename T>
__global__ __launch_bounds__(ND_THREADS) void log_prob_sum_nd_kernel(
    double* __restrict__ partial, const T* __restrict__ v, const T* __restrict__ a,
    const T* __restrict__ b, const uint8_t* __restrict__ m, NdFrame f, NdStrides sv, NdStrides sa,
    NdStrides sb, NdStrides sm, T scale) {
  __shared__ double smem[16];
  T acc = T(0);
  const uint32_t stride = gridDim.x * ND_THREADS;
  for (uint32_t i0 = blockIdx.x * ND_THREADS + threadIdx.x; i0 < f.total;
       i0 += stride * ND_ITEMS) {
#pragma unroll
    for (int k = 0; k < ND_ITEMS; ++k) {
      const uint32_t i = i0 + k * stride;
      if (i < f.total) {
        uint32_t idx[4];
        nd_index(f, i, idx);
        const T bb = NParams<DIST>::n > 1 ? b[nd_offset(sb, idx)] : T(0);
        T lp = Fam<DIST, T>::lp(v[nd_offset(sv, idx)], a[nd_offset(sa, idx)], bb) * scale;
        // scale_and_mask (distributions/util.py:311-328): where(mask, tensor*scale, 0)
        if (m != nullptr && m[nd_offset(sm, idx)] == 0) lp = T(0);
        acc += lp;
      }
    }
  }
  const double t = block_sum_f64((double)acc, smem);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

template <typename T>
__global__ __launch_bounds__(256) void nd_total_kernel(T* __restrict__ out,
                                                       const double* __restrict__ partial, int n) {
  __shared__ double smem[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  const double t = block_sum_f64(acc, smem);
  if (threadIdx.x == 0) *out = (T)t;
}

template <int DIST, typename T>
__global__ __launch_bounds__(ND_THREADS) void log_prob_grad_nd_kernel(
    T* __restrict__ dv, T* __restrict__ da, T* __restrict__ db, const T* __restrict__ g,
    const T* __restrict__ v, const T* __restrict__ a, const T* __restrict__ b,
    const uint8_t* __restrict__ m, NdFrame f, NdStrides sv, NdStrides sa, NdStrides sb,
    NdStrides sm, T scale) {
  const T w = *g * scale;                     // upstream gradient of the (scalar) site sum
  const uint32_t stride = gridDim.x * ND_THREADS;
  for (uint32_t i0 = blockIdx.x * ND_THREADS + threadIdx.x; i0 < f.total;
       i0 += stride * ND_ITEMS) {
#pragma unroll
    for (int k = 0; k < ND_ITEMS; ++k) {
      const uint32_t i = i0 + k * stride;
      if (i < f.total) {
        uint32_t idx[4];
        nd_index(f, i, idx);
        const T bb = NParams<DIST>::n > 1 ? b[nd_offset(sb, idx)] : T(0);
        T gv, ga, gb;
        Fam<DIST, T>::grad(v[nd_offset(sv, idx)], a[nd_offset(sa, idx)], bb, gv, ga, gb);
        // masked-out elements get an exact 0 gradient (torch.where backward)
        const bool keep = (m == nullptr) || (m[nd_offset(sm, idx)] != 0);
        if (dv) dv[i] = keep ? w * gv : T(0);
        if (da) da[i] = keep ? w * ga : T(0);
        if (db) db[i] = keep ? w * gb : T(0);
      }
    }
  }
}

// in[A, R, B] contiguous -> partial[A, K, B] (K splits of R) or out[A, B] directly when K == 1.
// Thread (tb, tr): TB = min(pow2 >= B, 256) threads along the contiguous B, TR = 256 / TB rows of
// R in flight per workgroup; grid = (B tiles, K, A).
// With a second tensor (in1 / out1, same shape: the two parameter gradients of one site) the grid's y
// extent is 2 K and the upper half works on the second pair; its partials follow the first tensor's.
template <typename T>
__global__ __launch_bounds__(256) void sum_to_nd_kernel(const T* __restrict__ in,
                                                        double* __restrict__ partial,
                                                        T* __restrict__ out, uint32_t A,
                                                        uint32_t R, uint32_t B, uint32_t TB,
                                                        uint32_t K, const T* __restrict__ in1 = nullptr,
                                                        T* __restrict__ out1 = nullptr) {
  __shared__ double sm[256];
  const uint32_t TR = 256 / TB;
  const uint32_t tb = threadIdx.x % TB, tr = threadIdx.x / TB;
  const uint32_t second = blockIdx.y >= K;
  if (second) {
    in = in1;
    out = out1;
    partial += (int64_t)A * K * B;
  }
  const uint32_t bcol = blockIdx.x * TB + tb, k = blockIdx.y - (second ? K : 0), a = blockIdx.z;
  const uint32_t per = (R + K - 1) / K;
  const uint32_t r0 = k * per, r1 = r0 + per < R ? r0 + per : R;
  double acc = 0.0;
  if (bcol < B) {
    const T* base = in + ((int64_t)a * R) * B + bcol;
    uint32_t r = r0 + tr;
    for (; r + 3 * TR < r1; r += 4 * TR) {        // 4 independent loads in flight
      const T x0 = base[(int64_t)r * B], x1 = base[(int64_t)(r + TR) * B];
      const T x2 = base[(int64_t)(r + 2 * TR) * B], x3 = base[(int64_t)(r + 3 * TR) * B];
      acc += ((double)x0 + (double)x1) + ((double)x2 + (double)x3);
    }
    for (; r < r1; r += TR) acc += (double)base[(int64_t)r * B];
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (tr == 0 && bcol < B) {
    double t = 0.0;
    for (uint32_t j = 0; j < TR; ++j) t += sm[j * TB + tb];     // fixed order
    if (K == 1) out[(int64_t)a * B + bcol] = (T)t;
    else partial[((int64_t)a * K + k) * B + bcol] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sum_to_nd_final_kernel(const double* __restrict__ partial,
                                                              T* __restrict__ out, uint32_t A,
                                                              uint32_t B, uint32_t K,
                                                              T* __restrict__ out1 = nullptr) {
  const int64_t n = (int64_t)A * B;
  if (blockIdx.y == 1) {
    partial += n * K;
    out = out1;
  }
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) {
    const int64_t a = o / B, bcol = o - a * B;
    const double* base = partial + a * K * B + bcol;
    double t = 0.0;
    uint32_t k = 0;
    for (; k + 3 < K; k += 4) {                   // loads issued together, summed in a fixed order
      const double x0 = base[(int64_t)k * B], x1 = base[(int64_t)(k + 1) * B];
      const double x2 = base[(int64_t)(k + 2) * B], x3 = base[(int64_t)(k + 3) * B];
      t += (x0 + x1) + (x2 + x3);
    }
    for (; k < K; ++k) t += base[(int64_t)k * B];
    out[o] = (T)t;
  }
}

static int nd_setup(const char* who, int ndim, const int64_t* sizes, NdFrame* f) {
  PA_REQUIRE(ndim >= 1 && ndim <= 4, "%s: ndim=%d outside [1, 4]", who, ndim);
  int64_t n[4] = {1, 1, 1, 1}, total = 1;
  for (int d = 0; d < ndim; ++d) {
    PA_REQUIRE(sizes[d] >= 0, "%s: negative size", who);
    n[4 - ndim + d] = sizes[d];
    total *= sizes[d];
  }
  PA_REQUIRE(total < (int64_t(1) << 31), "%s: frame of %lld elements (max 2^31 - 1)", who,
             (long long)total);
  f->n1 = (uint32_t)(n[1] ? n[1] : 1);
  f->n2 = (uint32_t)(n[2] ? n[2] : 1);
  f->n3 = (uint32_t)(n[3] ? n[3] : 1);
  f->total = (uint32_t)total;
  return PA_OK;
}

static int nd_strides(const char* who, int ndim, const int64_t* sizes, const int64_t* strides,
                      NdStrides* out) {
  for (int d = 0; d < 4; ++d) out->s[d] = 0;
  if (strides == nullptr) return PA_OK;
  int64_t reach = 0;
  for (int d = 0; d < ndim; ++d) {
    PA_REQUIRE(strides[d] >= 0, "%s: negative stride", who);
    out->s[4 - ndim + d] = (int32_t)strides[d];
    if (sizes[d] > 0) reach += (sizes[d] - 1) * strides[d];
  }
  PA_REQUIRE(reach < (int64_t(1) << 31), "%s: operand spans more than 2^31 elements", who);
  return PA_OK;
}

static int nd_blocks(uint32_t total) {
  int64_t want = ((int64_t)total + ND_THREADS * ND_ITEMS - 1) / (ND_THREADS * ND_ITEMS);
  if (want > ND_MAX_BLOCKS) want = ND_MAX_BLOCKS;
  return (int)(want < 1 ? 1 : want);
}

}  // namespace pa

extern "C" {

size_t pa_dist_log_prob_sum_nd_workspace(void) { return pa::ND_MAX_BLOCKS * sizeof(double); }

int pa_dist_log_prob_sum_nd(int dist, int dtype, void* out_total, int ndim, const int64_t* sizes,
                            const void* value, const int64_t* value_strides, const void* p0,
                            const int64_t* p0_strides, const void* p1, const int64_t* p1_strides,
                            const uint8_t* mask, const int64_t* mask_strides, double scale,
                            void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  using pa::fail;
  PA_REQUIRE(dist >= 0 && dist < PA_DIST_COUNT, "log_prob_sum_nd: unknown distribution id %d", dist);
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "log_prob_sum_nd: bad dtype %d", dtype);
  PA_REQUIRE(out_total && sizes, "log_prob_sum_nd: NULL pointer");
  pa::NdFrame f;
  int rc = pa::nd_setup("log_prob_sum_nd", ndim, sizes, &f);
  if (rc != PA_OK) return rc;
  hipStream_t s = pa::as_stream(stream);
  const size_t esz = dtype == PA_F32 ? 4 : 8;
  if (f.total == 0) {   // empty plate: the sum over nothing is 0
    if (hipMemsetAsync(out_total, 0, esz, s) != hipSuccess)
      return pa::fail(PA_ERR_LAUNCH, "log_prob_sum_nd: memset failed");
    return PA_OK;
  }
  const int np = pa::dist_nparams(dist);
  PA_REQUIRE(value && p0 && (np < 2 || p1), "log_prob_sum_nd: NULL operand");
  PA_REQUIRE(workspace && workspace_bytes >= pa_dist_log_prob_sum_nd_workspace(),
             "log_prob_sum_nd: workspace too small");
  pa::NdStrides sv, sa, sb, sm;
  if ((rc = pa::nd_strides("log_prob_sum_nd", ndim, sizes, value_strides, &sv)) != PA_OK) return rc;
  if ((rc = pa::nd_strides("log_prob_sum_nd", ndim, sizes, p0_strides, &sa)) != PA_OK) return rc;
  if ((rc = pa::nd_strides("log_prob_sum_nd", ndim, sizes, p1 ? p1_strides : nullptr, &sb)) != PA_OK) return rc;
  if ((rc = pa::nd_strides("log_prob_sum_nd", ndim, sizes, mask ? mask_strides : nullptr, &sm)) != PA_OK) return rc;
  const int nb = pa::nd_blocks(f.total);
  double* ws = (double*)workspace;
  if (dtype == PA_F32) {
    PA_DISPATCH_DIST(dist, float,
                     hipLaunchKernelGGL((pa::log_prob_sum_nd_kernel<D_, float>), dim3(nb),
                                        dim3(pa::ND_THREADS), 0, s, ws, (const float*)value,
                                        (const float*)p0, (const float*)p1, mask, f, sv, sa, sb,
                                        sm, (float)scale));
    hipLaunchKernelGGL((pa::nd_total_kernel<float>), dim3(1), dim3(256), 0, s, (float*)out_total,
                       ws, nb);
  } else {
    PA_DISPATCH_DIST(dist, double,
                     hipLaunchKernelGGL((pa::log_prob_sum_nd_kernel<D_, double>), dim3(nb),
                                        dim3(pa::ND_THREADS), 0, s, ws, (const double*)value,
                                        (const double*)p0, (const double*)p1, mask, f, sv, sa, sb,
                                        sm, scale));
    hipLaunchKernelGGL((pa::nd_total_kernel<double>), dim3(1), dim3(256), 0, s,
                       (double*)out_total, ws, nb);
  }
  return pa::check_launch("log_prob_sum_nd_kernel");
}

int pa_dist_log_prob_grad_nd(int dist, int dtype, void* d_value, void* d_p0, void* d_p1,
                             const void* g, int ndim, const int64_t* sizes, const void* value,
                             const int64_t* value_strides, const void* p0,
                             const int64_t* p0_strides, const void* p1, const int64_t* p1_strides,
                             const uint8_t* mask, const int64_t* mask_strides, double scale,
                             pa_stream_t stream) {
  using pa::fail;
  PA_REQUIRE(dist >= 0 && dist < PA_DIST_COUNT, "log_prob_grad_nd: unknown distribution id %d", dist);
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "log_prob_grad_nd: bad dtype %d", dtype);
  PA_REQUIRE(g && sizes, "log_prob_grad_nd: NULL pointer");
  pa::NdFrame f;
  int rc = pa::nd_setup("log_prob_grad_nd", ndim, sizes, &f);
  if (rc != PA_OK) return rc;
  if (f.total == 0 || (!d_value && !d_p0 && !d_p1)) return PA_OK;
  const int np = pa::dist_nparams(dist);
  PA_REQUIRE(value && p0 && (np < 2 || p1), "log_prob_grad_nd: NULL operand");
  pa::NdStrides sv, sa, sb, sm;
  if ((rc = pa::nd_strides("log_prob_grad_nd", ndim, sizes, value_strides, &sv)) != PA_OK) return rc;
  if ((rc = pa::nd_strides("log_prob_grad_nd", ndim, sizes, p0_strides, &sa)) != PA_OK) return rc;
  if ((rc = pa::nd_strides("log_prob_grad_nd", ndim, sizes, p1 ? p1_strides : nullptr, &sb)) != PA_OK) return rc;
  if ((rc = pa::nd_strides("log_prob_grad_nd", ndim, sizes, mask ? mask_strides : nullptr, &sm)) != PA_OK) return rc;
  hipStream_t s = pa::as_stream(stream);
  const int nb = pa::nd_blocks(f.total);
  if (dtype == PA_F32) {
    PA_DISPATCH_DIST(dist, float,
                     hipLaunchKernelGGL((pa::log_prob_grad_nd_kernel<D_, float>), dim3(nb),
                                        dim3(pa::ND_THREADS), 0, s, (float*)d_value, (float*)d_p0,
                                        (float*)d_p1, (const float*)g, (const float*)value,
                                        (const float*)p0, (const float*)p1, mask, f, sv, sa, sb,
                                        sm, (float)scale));
  } else {
    PA_DISPATCH_DIST(dist, double,
                     hipLaunchKernelGGL((pa::log_prob_grad_nd_kernel<D_, double>), dim3(nb),
                                        dim3(pa::ND_THREADS), 0, s, (double*)d_value,
                                        (double*)d_p0, (double*)d_p1, (const double*)g,
                                        (const double*)value, (const double*)p0,
                                        (const double*)p1, mask, f, sv, sa, sb, sm, scale));
  }
  return pa::check_launch("log_prob_grad_nd_kernel");
}

static void sum_to_plan(int64_t A, int64_t R, int64_t B, uint32_t* TB, uint32_t* K,
                        uint32_t* btiles) {
  uint32_t tb = 1;
  while (tb < 256 && tb < B) tb *= 2;
  const uint32_t tr = 256 / tb;
  *TB = tb;
  *btiles = (uint32_t)((B + tb - 1) / tb);
  int64_t k = pa::ND_MAX_BLOCKS / (A * (int64_t)*btiles);
  const int64_t kmax = (R + 4 * tr - 1) / (4 * tr);     // at least 4 rows per thread and split
  if (k > kmax) k = kmax;
  if (k > 16) k = 16;                                  // the second pass reads K values per output
  if (k < 1) k = 1;
  *K = (uint32_t)k;
}

size_t pa_sum_to_nd_workspace(int64_t A, int64_t R, int64_t B) {
  if (A <= 0 || R <= 0 || B <= 0) return 0;
  uint32_t TB, K, bt;
  sum_to_plan(A, R, B, &TB, &K, &bt);
  return K > 1 ? (size_t)(A * K * B) * sizeof(double) : 0;
}

static int sum_to_nd_run(const char* who, int dtype, const void* in, void* out, const void* in1, void* out1,
                         int64_t A, int64_t R, int64_t B, void* workspace, size_t workspace_bytes,
                         pa_stream_t stream) {
  const int two = out1 != nullptr;
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "%s: bad dtype %d", who, dtype);
  PA_REQUIRE(A >= 0 && R >= 0 && B >= 0, "%s: negative size", who);
  PA_REQUIRE(A < 65536 && R < (int64_t(1) << 31) && B < (int64_t(1) << 31) &&
                 A * B < (int64_t(1) << 40), "%s: shape too large", who);
  if (A == 0 || B == 0) return PA_OK;
  PA_REQUIRE(out, "%s: NULL output", who);
  hipStream_t s = pa::as_stream(stream);
  const size_t esz = dtype == PA_F32 ? 4 : 8;
  if (R == 0) {
    if (hipMemsetAsync(out, 0, (size_t)(A * B) * esz, s) != hipSuccess ||
        (two && hipMemsetAsync(out1, 0, (size_t)(A * B) * esz, s) != hipSuccess))
      return pa::fail(PA_ERR_LAUNCH, "%s: memset failed", who);
    return PA_OK;
  }
  PA_REQUIRE(in && (!two || in1), "%s: NULL input", who);
  uint32_t TB, K, bt;
  sum_to_plan(A, R, B, &TB, &K, &bt);
  PA_REQUIRE(K == 1 || (workspace && workspace_bytes >= (two ? 2 : 1) * pa_sum_to_nd_workspace(A, R, B)),
             "%s: workspace too small", who);
  dim3 grid(bt, K * (two ? 2 : 1), (unsigned)A);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::sum_to_nd_kernel<float>), grid, dim3(256), 0, s, (const float*)in,
                       (double*)workspace, (float*)out, (uint32_t)A, (uint32_t)R, (uint32_t)B, TB,
                       K, (const float*)in1, (float*)out1);
  else
    hipLaunchKernelGGL((pa::sum_to_nd_kernel<double>), grid, dim3(256), 0, s, (const double*)in,
                       (double*)workspace, (double*)out, (uint32_t)A, (uint32_t)R, (uint32_t)B,
                       TB, K, (const double*)in1, (double*)out1);
  int rc = pa::check_launch("sum_to_nd_kernel");
  if (rc != PA_OK || K == 1) return rc;
  const int64_t n = A * B;
  int64_t fb = (n + 255) / 256;
  if (fb > 1024) fb = 1024;
  dim3 fgrid((unsigned)fb, two ? 2 : 1);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::sum_to_nd_final_kernel<float>), fgrid, dim3(256), 0, s,
                       (const double*)workspace, (float*)out, (uint32_t)A, (uint32_t)B, K, (float*)out1);
  else
    hipLaunchKernelGGL((pa::sum_to_nd_final_kernel<double>), fgrid, dim3(256), 0, s,
                       (const double*)workspace, (double*)out, (uint32_t)A, (uint32_t)B, K, (double*)out1);
  return pa::check_launch("sum_to_nd_final_kernel");
}

int pa_sum_to_nd(int dtype, const void* in, void* out, int64_t A, int64_t R, int64_t B,
                 void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  return sum_to_nd_run("sum_to_nd", dtype, in, out, nullptr, nullptr, A, R, B, workspace, workspace_bytes,
                       stream);
}

int pa_sum_to_nd_pair(int dtype, const void* in0, void* out0, const void* in1, void* out1, int64_t A,
                      int64_t R, int64_t B, void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(out1 != nullptr || A == 0 || B == 0, "sum_to_nd_pair: NULL second output");
  return sum_to_nd_run("sum_to_nd_pair", dtype, in0, out0, in1, out1, A, R, B, workspace, workspace_bytes,
                       stream);
}

}  // extern "C"
