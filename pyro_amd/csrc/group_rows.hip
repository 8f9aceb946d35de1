// group_rows.hip -- the rows of a plate ordered by an UNSORTED int64 group id.
//
// SURVEY 8(d) config 5 writes the hierarchical GLM as the reference would: g = randint(0, G, (N,)),
// logits_n = (w[..., g, :] * X).sum(-1) + b -- an advanced-index gather that materialises [P, N, D]
// (torch: aten index; pyro/poutine/trace_struct.py:264-278 then scores it).  The grouped plane-image
// kernel (glm_planes16.h) streams one group's rows with one group's weights, so the rows have to be
// visited in group order.  This file produces that order ONCE per (X, g): a stable counting sort of
// the row indices by group id --
//     rows[offsets[k] .. offsets[k + 1])  =  the rows n with g[n] == k, ascending in n
// i.e. numpy's argsort(g, kind="stable") and the exclusive cumulative bincount: integer work,
// bit-exact against oracle/glm.py::group_rows.  The image packer then reads X[rows[i]]
// (pa_glm_pack_planes_grouped_rows): no sorted copy of X is ever made.
//
// Three launches (the scheme of lda.hip's word index): per-chunk histograms in LDS (integer atomics:
// exact), one scan over (group, chunk), and a fill in which each wave ranks equal ids among its 64
// rows with ballots -- ascending row order inside every group, no run-to-run change.
#include "common.h"

namespace pa {

constexpr int GR_CHUNKS = 1024;      // chunks of the row range
constexpr int GR_MAX_G = 16384;      // LDS histogram: 64 KiB of int32

static int64_t gr_chunk_len(int64_t n) {
  int64_t ch = (n + GR_CHUNKS - 1) / GR_CHUNKS;
  return ((ch + 63) / 64) * 64;
}

// workspace: counts int32 [GR_CHUNKS][G]
__global__ __launch_bounds__(64) void group_rows_count_kernel(const int64_t* __restrict__ g, int64_t n,
                                                              int64_t chunk, int G,
                                                              int* __restrict__ counts,
                                                              unsigned long long* __restrict__ bad) {
  extern __shared__ int gr_cnt[];
  for (int v = threadIdx.x; v < G; v += 64) gr_cnt[v] = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
  int nbad = 0;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 64) {
    int64_t v = g[i];
    if (v < 0 || v >= G) {      // torch raises an IndexError for such an index: counted, reported
      ++nbad;
      v = 0;
    }
    atomicAdd(&gr_cnt[(int)v], 1);
  }
  __syncthreads();
  for (int v = threadIdx.x; v < G; v += 64) counts[(int64_t)blockIdx.x * G + v] = gr_cnt[v];
  if (nbad) atomicAdd(bad, (unsigned long long)nbad);
}

// counts[c][k] -> rows of group k in earlier chunks (in place); offsets[G + 1]
__global__ __launch_bounds__(1024) void group_rows_scan_kernel(int* __restrict__ counts, int G,
                                                               int64_t* __restrict__ offsets) {
  for (int v = threadIdx.x; v < G; v += 1024) {
    int64_t run = 0;
    for (int c = 0; c < GR_CHUNKS; ++c) {
      const int k = counts[(int64_t)c * G + v];
      counts[(int64_t)c * G + v] = (int)run;       // < 2^31: the launcher bounds n
      run += k;
    }
    offsets[v + 1] = run;                          // the group's total for now
  }
  __syncthreads();
  if (threadIdx.x == 0) {                          // G is a group count (thousands): serial, once
    int64_t run = 0;
    offsets[0] = 0;
    for (int v = 0; v < G; ++v) {
      run += offsets[v + 1];
      offsets[v + 1] = run;
    }
  }
}

__global__ __launch_bounds__(64) void group_rows_fill_kernel(const int64_t* __restrict__ g, int64_t n,
                                                             int64_t chunk, int G,
                                                             const int* __restrict__ counts,
                                                             const int64_t* __restrict__ offsets,
                                                             int64_t* __restrict__ rows) {
  extern __shared__ int gr_cur[];                  // cursor RELATIVE to the group's offset
  for (int v = threadIdx.x; v < G; v += 64) gr_cur[v] = counts[(int64_t)blockIdx.x * G + v];
  __syncthreads();
  const int lane = threadIdx.x;
  const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
  for (int64_t ib = i0; ib < i1; ib += 64) {
    const int64_t i = ib + lane;
    const bool ok = i < i1;
    const int64_t v64 = ok ? g[i] : 0;
    const int v = (int)((v64 < 0 || v64 >= G) ? 0 : v64);
    int rank = 0, total = 0;
    uint64_t todo = __ballot(ok);
    while (todo) {                                 // one round per distinct id among the 64 rows
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
    if (ok) base = gr_cur[v];
    if (ok) rows[offsets[v] + base + rank] = i;
    __builtin_amdgcn_s_waitcnt(0);                 // every lane has read its cursor
    if (ok && rank == total - 1) gr_cur[v] = base + total;
    __builtin_amdgcn_s_waitcnt(0);
  }
}

}  // namespace pa

extern "C" {

size_t pa_group_rows_workspace(int64_t N, int64_t G) {
  if (N < 0 || G < 1 || G > pa::GR_MAX_G || N >= ((int64_t)1 << 31)) return 0;
  return (size_t)pa::GR_CHUNKS * (size_t)G * sizeof(int);
}

int pa_group_rows_build(const int64_t* g, int64_t N, int64_t G, int64_t* offsets, int64_t* rows,
                        int64_t* n_out_of_range, void* workspace, size_t workspace_bytes,
                        pa_stream_t stream) {
  PA_REQUIRE(N >= 0 && G >= 1, "group_rows_build: bad shape N=%lld G=%lld", (long long)N, (long long)G);
  if (G > pa::GR_MAX_G || N >= ((int64_t)1 << 31))
    return pa::fail(PA_ERR_UNSUPPORTED, "group_rows_build: G <= %d and N < 2^31 (got G=%lld N=%lld)",
                    pa::GR_MAX_G, (long long)G, (long long)N);
  PA_REQUIRE(offsets && n_out_of_range && (N == 0 || (g && rows)), "group_rows_build: NULL pointer");
  PA_REQUIRE(workspace && workspace_bytes >= pa_group_rows_workspace(N, G),
             "group_rows_build: workspace too small");
  hipStream_t s = pa::as_stream(stream);
  int* counts = (int*)workspace;
  if (hipMemsetAsync(n_out_of_range, 0, sizeof(int64_t), s) != hipSuccess)
    return pa::fail(PA_ERR_LAUNCH, "group_rows_build: memset failed");
  const int64_t chunk = pa::gr_chunk_len(N);
  const size_t lds = (size_t)G * sizeof(int);
  hipLaunchKernelGGL(pa::group_rows_count_kernel, dim3(pa::GR_CHUNKS), dim3(64), lds, s, g, N, chunk,
                     (int)G, counts, (unsigned long long*)n_out_of_range);
  int rc = pa::check_launch("group_rows_count_kernel");
  if (rc != PA_OK) return rc;
  hipLaunchKernelGGL(pa::group_rows_scan_kernel, dim3(1), dim3(1024), 0, s, counts, (int)G, offsets);
  rc = pa::check_launch("group_rows_scan_kernel");
  if (rc != PA_OK) return rc;
  hipLaunchKernelGGL(pa::group_rows_fill_kernel, dim3(pa::GR_CHUNKS), dim3(64), lds, s, g, N, chunk,
                     (int)G, counts, offsets, rows);
  return pa::check_launch("group_rows_fill_kernel");
}

}  // extern "C"
