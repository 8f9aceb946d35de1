// mixture.hip -- the marginal likelihood of an observed site under an enumerated assignment, in one pass.
//
// Reference path replaced: TraceEnum_ELBO on a plated mixture (pyro/infer/traceenum_elbo.py:112-214): the observed
// site's log_prob against every value of the enumerated variable -- a [K, N] tensor -- is materialised
// (pyro/poutine/trace_struct.py:248-288), added to the assignment's log-probabilities, reduced by logsumexp over K
// (pyro/ops/contract.py:79-160 -> torch_log.einsum) and summed over the plate; autograd keeps the [K, N] frame and
// walks it twice more.  Round 3 fused the adds + logsumexp (logsumexp.hip) but still read a materialised [K, N]
// likelihood (64 MB at N = 1e6, K = 16, written once and read twice).
//
// Here:   S = sum_n log sum_k exp(a_k + log p(x_n | p0_k, p1_k))
// with the responsibilities r_nk never leaving registers, and in the same pass
//         dS/da_k = sum_n r_nk,   dS/dp0_k = sum_n r_nk d log p / d p0,   dS/dp1_k likewise
// -- the whole forward AND backward of the leaf: x is read once (4 MB), nothing of size K N exists.
// Lane layout: KP = K rounded up to a power of two (<= 64) lanes hold one row's K terms; 64 / KP rows per wave
// and iteration; the logsumexp of a row is KP-lane shuffles, the gradient sums live in the lane that owns k.
// Reduction: lanes -> waves -> workgroup partials in double, added in a fixed order by a second small launch
// (bit-reproducible).
#include "common.h"
#include "dist_fam.h"

namespace pa {

constexpr int MIX_THREADS = 256;
constexpr int MIX_MAXK = 64;

template <typename T> __device__ __forceinline__ T mix_max(T a, T b) { return a > b ? a : b; }

// The value of lane (l ^ o) for o < 16 without the LDS crossbar: after quad_perm [1,0,3,2] and [2,3,0,1] every lane
// of a quad has combined the quad's four values; row_half_mirror exchanges the two quads of 8 lanes, row_mirror the
// two halves of 16 -- an all-reduce over 2 / 4 / 8 / 16 lanes in 1..4 DPP moves (ds_bpermute: an LDS instruction
// and its wait each).  64-bit values travel as two halves.
template <int CTRL> __device__ __forceinline__ float mix_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ double mix_dpp(double v) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// all-reduce over the KP lanes (a power of two, aligned) that hold one row's terms
template <int KP, typename T, typename Op> __device__ __forceinline__ T mix_allreduce(T v, Op op) {
  if constexpr (KP >= 2) v = op(v, mix_dpp<0xB1>(v));
  if constexpr (KP >= 4) v = op(v, mix_dpp<0x4E>(v));
  if constexpr (KP >= 8) v = op(v, mix_dpp<0x141>(v));
  if constexpr (KP >= 16) v = op(v, mix_dpp<0x140>(v));
  if constexpr (KP >= 32) v = op(v, __shfl_xor(v, 16, 64));
  if constexpr (KP >= 64) v = op(v, __shfl_xor(v, 32, 64));
  return v;
}

template <int DIST, typename T, int KP>
__global__ __launch_bounds__(MIX_THREADS) void mixture_kernel(const T* __restrict__ x, int64_t N, int K,
                                                              const T* __restrict__ a,
                                                              const T* __restrict__ p0, int64_t s0,
                                                              const T* __restrict__ p1, int64_t s1,
                                                              int64_t a_bs, int64_t p0_bs, int64_t p1_bs,
                                                              double* __restrict__ partial) {
  // blockIdx.y: one of B parameter sets over the SAME data (vectorised chains / particles: a chain's weights and
  // component parameters against the shared observations)
  a += (int64_t)blockIdx.y * a_bs;
  p0 += (int64_t)blockIdx.y * p0_bs;
  if (p1 != nullptr) p1 += (int64_t)blockIdx.y * p1_bs;
  constexpr int RPW = 64 / KP;                     // rows per wave and iteration
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane & (KP - 1), slot = lane / KP;
  const bool kok = k < K;
  const T ninf = -t_inf<T>();
  const T ak = kok ? a[k] : ninf;
  const T p0k = p0[kok ? (int64_t)k * s0 : 0];
  const T p1k = p1 != nullptr ? p1[kok ? (int64_t)k * s1 : 0] : T(0);
  // (a lane sees N / (waves RPW) rows: tens to a few thousand -- sums in T per lane, in double across lanes)
  T acc_s = T(0), acc_a = T(0), acc_0 = T(0), acc_1 = T(0);
  const int64_t step = (int64_t)gridDim.x * (MIX_THREADS / 64) * RPW;
  // (a wave's iteration reads 64 / KP values of x -- a few bytes: four iterations' loads are requested together,
  //  otherwise every iteration waits out a memory round trip: 48 us at N = 1e6, K = 16 with two waves per SIMD)
  constexpr int U = 4;
  for (int64_t base0 = ((int64_t)blockIdx.x * (MIX_THREADS / 64) + wave) * RPW; base0 < N; base0 += U * step) {
    T xs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = base0 + u * step + slot;
      xs[u] = x[row < N ? row : N - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int64_t base = base0 + u * step;
    if (base >= N) break;                            // (wave-uniform)
    const int64_t row = base + slot;
    const bool valid = row < N;
    const T xv = xs[u];
    T t = kok ? ak + Fam<DIST, T>::lp(xv, p0k, p1k) : ninf;
    const T m = mix_allreduce<KP>(t, [](T p, T q) { return mix_max(p, q); });
    // (a row whose every term is -inf: its logsumexp is -inf and it has no responsibilities)
    const bool dead = !(m > ninf);
    const T e = (dead || !kok) ? T(0) : t_exp(t - m);
    const T ssum = mix_allreduce<KP>(e, [](T p, T q) { return p + q; });
    // (ssum >= 1 for a live row: the hardware log / reciprocal of the float path are 1 ulp there)
    const T lse = dead ? ninf : m + pos_log(dead ? T(1) : ssum);
    const T r = (dead || !valid) ? T(0) : e * pos_rcp(dead ? T(1) : ssum);
    T dv, da, db;
    Fam<DIST, T>::grad(xv, p0k, p1k, dv, da, db);
    acc_s += (valid && k == 0) ? lse : T(0);
    acc_a += r;
    acc_0 += r > T(0) ? r * da : T(0);
    acc_1 += r > T(0) ? r * db : T(0);
    }
  }
  // lanes of the same k (the wave's RPW row slots), in double
  double ds = (double)acc_s, dsa = (double)acc_a, ds0 = (double)acc_0, ds1 = (double)acc_1;
#pragma unroll
  for (int o = KP; o < 64; o <<= 1) {
    ds += __shfl_xor(ds, o, 64);
    dsa += __shfl_xor(dsa, o, 64);
    ds0 += __shfl_xor(ds0, o, 64);
    ds1 += __shfl_xor(ds1, o, 64);
  }
  __shared__ double red[MIX_THREADS / 64][4][MIX_MAXK];
  if (lane < KP) {
    red[wave][0][lane] = ds;
    red[wave][1][lane] = dsa;
    red[wave][2][lane] = ds0;
    red[wave][3][lane] = ds1;
  }
  __syncthreads();
  // partial[block][0] = sum of the rows' logsumexp; [1 + q * KP + k], q = 0..2: the three gradient sums
  const int J = 1 + 3 * KP;
  for (int j = threadIdx.x; j < J; j += MIX_THREADS) {
    const int q = j == 0 ? 0 : 1 + (j - 1) / KP, kk = j == 0 ? 0 : (j - 1) % KP;
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < MIX_THREADS / 64; ++w) v += red[w][q][kk];
    partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * J + j] = v;
  }
}

// out[0] = S, out[1 + k] = dS/da_k, out[1 + K + k] = dS/dp0_k, out[1 + 2 K + k] = dS/dp1_k: the workgroups'
// partials added in index order (16 segments per output in parallel, the segments in order)
__global__ __launch_bounds__(1024) void mixture_finalize_kernel(const double* __restrict__ partial, int nblocks,
                                                                int K, int KP, double* __restrict__ out) {
  __shared__ double seg[16][64];
  const int jj = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int J = 1 + 3 * KP;
  const int j = blockIdx.x * 64 + jj;
  partial += (int64_t)blockIdx.y * nblocks * J;
  out += (int64_t)blockIdx.y * (1 + 3 * K);
  double v = 0.0;
  if (j < J) {
    const int per = (nblocks + 15) / 16;
    const int b0 = sg * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    // (eight loads in flight: the loop is a chain of L2 round trips otherwise -- 38 us for 2048 partials)
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      double q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = partial[(int64_t)(b + u) * J + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += q[u];
    }
    for (; b < b1; ++b) v += partial[(int64_t)b * J + j];
  }
  seg[sg][jj] = v;
  __syncthreads();
  if (sg == 0 && j < J) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += seg[q][jj];
    if (j == 0) {
      out[0] = t;
    } else {
      const int q = (j - 1) / KP, kk = (j - 1) % KP;
      if (kk < K) out[1 + q * K + kk] = t;
    }
  }
}

// ---- event-shaped observations: a diagonal Normal over D features (Normal(loc[z], scale).to_event(1)) -------------
//   S[b] = sum_n log sum_k exp(a[b][k] + sum_d log N(x[n][d] | loc[b][k][d], scale[b][k][d]))
// The sum over d sits INSIDE the logsumexp, so the features of a row stay with the lane that holds (row, k): DD = D
// rounded up to 2 / 4 / 8 loc and 1 / scale values and the same number of gradient sums per lane.  Every WAVE writes
// its own partial (no LDS stage: 2 + 2 DD doubles per lane would be 70 KB); the second launch adds them in order.
// Partial / padded output layout per set: [0] = S, [1 + k] = dS/da_k, [1 + KP + k * DD + d] = dS/dloc_kd,
// [1 + KP + KP * DD + k * DD + d] = dS/dscale_kd.
template <typename T, int KP, int DD>
__global__ __launch_bounds__(MIX_THREADS) void mixture_diag_kernel(const T* __restrict__ x, int64_t N, int D, int K,
                                                                   const T* __restrict__ a, int64_t a_bs,
                                                                   const T* __restrict__ loc, int64_t l_sk,
                                                                   int64_t l_sd, int64_t l_bs,
                                                                   const T* __restrict__ scl, int64_t s_sk,
                                                                   int64_t s_sd, int64_t s_bs,
                                                                   double* __restrict__ partial) {
  a += (int64_t)blockIdx.y * a_bs;
  loc += (int64_t)blockIdx.y * l_bs;
  scl += (int64_t)blockIdx.y * s_bs;
  constexpr int RPW = 64 / KP;
  constexpr int JP = 1 + KP + 2 * KP * DD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane & (KP - 1), slot = lane / KP;
  const bool kok = k < K;
  const T ninf = -t_inf<T>();
  T mu[DD], inv[DD];
  T ck = kok ? a[k] : ninf;                         // a_k - sum_d log scale_d - D / 2 log 2 pi
#pragma unroll
  for (int d = 0; d < DD; ++d) {
    const bool dok = d < D;
    mu[d] = loc[(kok ? (int64_t)k * l_sk : 0) + (dok ? (int64_t)d * l_sd : 0)];
    const T sg = scl[(kok ? (int64_t)k * s_sk : 0) + (dok ? (int64_t)d * s_sd : 0)];
    inv[d] = dok ? pos_rcp(sg) : T(0);
    if (dok && kok) ck -= pos_log(sg) + Consts<T>::half_log_2pi;
  }
  T acc_s = T(0), acc_a = T(0), acc_l[DD], acc_c[DD];
#pragma unroll
  for (int d = 0; d < DD; ++d) acc_l[d] = acc_c[d] = T(0);
  const int64_t nwaves = (int64_t)gridDim.x * (MIX_THREADS / 64);
  const int64_t step = nwaves * RPW;
  for (int64_t base = ((int64_t)blockIdx.x * (MIX_THREADS / 64) + wave) * RPW; base < N; base += step) {
    const int64_t row = base + slot;
    const bool valid = row < N;
    const T* xr = x + (valid ? row : N - 1) * D;
    T z[DD];
    T q = T(0);
#pragma unroll
    for (int d = 0; d < DD; ++d) {
      const T xv = xr[d < D ? d : 0];
      z[d] = (xv - mu[d]) * inv[d];                 // (inv = 0 past D)
      q += z[d] * z[d];
    }
    const T t = kok ? ck - T(0.5) * q : ninf;
    const T m = mix_allreduce<KP>(t, [](T p, T r_) { return mix_max(p, r_); });
    const bool dead = !(m > ninf);
    const T e = (dead || !kok) ? T(0) : t_exp(t - m);
    const T ssum = mix_allreduce<KP>(e, [](T p, T r_) { return p + r_; });
    const T lse = dead ? ninf : m + pos_log(dead ? T(1) : ssum);
    const T r = (dead || !valid) ? T(0) : e * pos_rcp(dead ? T(1) : ssum);
    acc_s += (valid && k == 0) ? lse : T(0);
    acc_a += r;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
      const T w = r * inv[d];
      acc_l[d] += w * z[d];                         // r (x - loc) / scale^2
      acc_c[d] += w * (z[d] * z[d] - T(1));         // r ((x - loc)^2 / scale^2 - 1) / scale
    }
  }
  const int64_t gw = (int64_t)blockIdx.x * (MIX_THREADS / 64) + wave;
  double* out = partial + ((int64_t)blockIdx.y * nwaves + gw) * JP;
  auto over_slots = [&](T v) {
    double dv = (double)v;
#pragma unroll
    for (int o = KP; o < 64; o <<= 1) dv += __shfl_xor(dv, o, 64);
    return dv;
  };
  const double ds = over_slots(acc_s), da = over_slots(acc_a);
  if (lane == 0) out[0] = ds;
  if (lane < KP) out[1 + lane] = da;
#pragma unroll
  for (int d = 0; d < DD; ++d) {
    const double dl = over_slots(acc_l[d]), dc = over_slots(acc_c[d]);
    if (lane < KP) {
      out[1 + KP + lane * DD + d] = dl;
      out[1 + KP + KP * DD + lane * DD + d] = dc;
    }
  }
}

// out[b][j] = the waves' partials added in index order (16 segments per output in parallel, eight loads in flight)
__global__ __launch_bounds__(1024) void mixture_sum_partials_kernel(const double* __restrict__ partial, int nparts,
                                                                    int J, double* __restrict__ out) {
  __shared__ double seg[16][64];
  const int jj = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jj;
  partial += (int64_t)blockIdx.y * nparts * J;
  out += (int64_t)blockIdx.y * J;
  double v = 0.0;
  if (j < J) {
    const int per = (nparts + 15) / 16;
    const int b0 = sg * per, b1 = b0 + per < nparts ? b0 + per : nparts;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      double q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = partial[(int64_t)(b + u) * J + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += q[u];
    }
    for (; b < b1; ++b) v += partial[(int64_t)b * J + j];
  }
  seg[sg][jj] = v;
  __syncthreads();
  if (sg == 0 && j < J) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += seg[q][jj];
    out[j] = t;
  }
}

static int mixture_diag_blocks(int64_t N, int KP, int64_t B) {
  const int64_t rows_per_block = (int64_t)(MIX_THREADS / 64) * (64 / KP);
  int64_t g = (N + rows_per_block * 8 - 1) / (rows_per_block * 8);
  int64_t cap = ((int64_t)cu_count() * 2 + B - 1) / B;      // (every wave writes a partial: fewer, longer waves)
  if (cap < 1) cap = 1;
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

template <typename T, int KP>
static int mixture_diag_launch(const T* x, int64_t N, int D, int K, int64_t B, const T* a, int64_t a_bs,
                               const T* loc, int64_t l_sk, int64_t l_sd, int64_t l_bs, const T* scl, int64_t s_sk,
                               int64_t s_sd, int64_t s_bs, double* partial, double* out, hipStream_t s) {
  const int DD = D <= 2 ? 2 : D <= 4 ? 4 : 8;
  const int grid = mixture_diag_blocks(N, KP, B);
  const int nparts = grid * (MIX_THREADS / 64);
  const int JP = 1 + KP + 2 * KP * DD;
#define PA_MIXD_CASE(DD_)                                                                              \
  if (DD == DD_)                                                                                       \
    hipLaunchKernelGGL((mixture_diag_kernel<T, KP, DD_>), dim3((unsigned)grid, (unsigned)B),           \
                       dim3(MIX_THREADS), 0, s, x, N, D, K, a, a_bs, loc, l_sk, l_sd, l_bs, scl, s_sk, \
                       s_sd, s_bs, partial);
  PA_MIXD_CASE(2) PA_MIXD_CASE(4) PA_MIXD_CASE(8)
#undef PA_MIXD_CASE
  hipLaunchKernelGGL(mixture_sum_partials_kernel, dim3((unsigned)((JP + 63) / 64), (unsigned)B), dim3(1024), 0, s,
                     partial, nparts, JP, out);
  return check_launch("mixture_diag_kernel");
}

template <typename T>
static int mixture_diag_t(const T* x, int64_t N, int D, int K, int64_t B, const T* a, int64_t a_bs, const T* loc,
                          int64_t l_sk, int64_t l_sd, int64_t l_bs, const T* scl, int64_t s_sk, int64_t s_sd,
                          int64_t s_bs, double* partial, double* out, hipStream_t s) {
  int KP = 1;
  while (KP < K) KP <<= 1;
#define PA_MIXD_K(KP_)                                                                                 \
  if (KP == KP_)                                                                                       \
    return mixture_diag_launch<T, KP_>(x, N, D, K, B, a, a_bs, loc, l_sk, l_sd, l_bs, scl, s_sk, s_sd, s_bs,     \
                                       partial, out, s);
  PA_MIXD_K(1) PA_MIXD_K(2) PA_MIXD_K(4) PA_MIXD_K(8) PA_MIXD_K(16) PA_MIXD_K(32) PA_MIXD_K(64)
#undef PA_MIXD_K
  return fail(PA_ERR_UNSUPPORTED, "mixture_diag_normal_fwd_bwd: K=%d", K);
}

static int mixture_grid(int64_t N, int KP, int64_t B) {
  const int64_t rows_per_block = (int64_t)(MIX_THREADS / 64) * (64 / KP);
  int64_t g = (N + rows_per_block * 8 - 1) / (rows_per_block * 8);       // >= 8 iterations per wave
  // four waves per SIMD over all parameter sets; 1024 partials for the second launch
  int64_t cap = ((int64_t)cu_count() * 4 + B - 1) / B;
  if (cap < 1) cap = 1;
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

template <int DIST, typename T>
static int mixture_launch(const T* x, int64_t N, int K, int64_t B, const T* a, const T* p0, int64_t s0,
                          const T* p1, int64_t s1, int64_t a_bs, int64_t p0_bs, int64_t p1_bs, double* partial,
                          double* out, hipStream_t s) {
  int KP = 1;
  while (KP < K) KP <<= 1;
  const int grid = mixture_grid(N, KP, B);
#define PA_MIX_CASE(KP_)                                                                               \
  if (KP == KP_)                                                                                       \
    hipLaunchKernelGGL((mixture_kernel<DIST, T, KP_>), dim3((unsigned)grid, (unsigned)B),              \
                       dim3(MIX_THREADS), 0, s, x, N, K, a, p0, s0, p1, s1, a_bs, p0_bs, p1_bs, partial);
  PA_MIX_CASE(1) PA_MIX_CASE(2) PA_MIX_CASE(4) PA_MIX_CASE(8) PA_MIX_CASE(16) PA_MIX_CASE(32) PA_MIX_CASE(64)
#undef PA_MIX_CASE
  const int J = 1 + 3 * KP;
  hipLaunchKernelGGL(mixture_finalize_kernel, dim3((unsigned)((J + 63) / 64), (unsigned)B), dim3(1024), 0, s,
                     partial, grid, K, KP, out);
  return check_launch("mixture_kernel");
}

template <typename T>
static int mixture_t(int dist, const T* x, int64_t N, int K, int64_t B, const T* a, const T* p0, int64_t s0,
                     const T* p1, int64_t s1, int64_t a_bs, int64_t p0_bs, int64_t p1_bs, double* partial,
                     double* out, hipStream_t s) {
#define PA_MIX_FAM(ID)                                                                                 \
  case ID: return mixture_launch<ID, T>(x, N, K, B, a, p0, s0, p1, s1, a_bs, p0_bs, p1_bs, partial, out, s);
  switch (dist) {
    PA_MIX_FAM(PA_DIST_NORMAL)
    PA_MIX_FAM(PA_DIST_LOG_NORMAL)
    PA_MIX_FAM(PA_DIST_EXPONENTIAL)
    PA_MIX_FAM(PA_DIST_BERNOULLI_LOGITS)
    PA_MIX_FAM(PA_DIST_POISSON)
    PA_MIX_FAM(PA_DIST_GAMMA)
    default: return fail(PA_ERR_UNSUPPORTED, "mixture_fwd_bwd: distribution id %d not implemented", dist);
  }
#undef PA_MIX_FAM
}

}  // namespace pa

extern "C" {

size_t pa_mixture_workspace(int K, int64_t B) {
  if (K < 1 || K > pa::MIX_MAXK || B < 1) return 0;
  int KP = 1;
  while (KP < K) KP <<= 1;
  // (at most cu_count * 4 + B workgroups over all parameter sets)
  return ((size_t)pa::cu_count() * 4 + (size_t)B) * (1 + 3 * KP) * sizeof(double);
}

static int mixture_diag_padded(int K, int D, int* kp, int* dd) {
  int KP = 1;
  while (KP < K) KP <<= 1;
  const int DD = D <= 2 ? 2 : D <= 4 ? 4 : 8;
  if (kp) *kp = KP;
  if (dd) *dd = DD;
  return 1 + KP + 2 * KP * DD;
}

int pa_mixture_diag_normal_layout(int K, int D, int* kp_out, int* dd_out) {
  if (K < 1 || K > pa::MIX_MAXK || D < 1 || D > 8) return 0;
  return mixture_diag_padded(K, D, kp_out, dd_out);
}

size_t pa_mixture_diag_normal_workspace(int K, int D, int64_t B) {
  if (K < 1 || K > pa::MIX_MAXK || D < 1 || D > 8 || B < 1) return 0;
  const int JP = mixture_diag_padded(K, D, nullptr, nullptr);
  return ((size_t)pa::cu_count() * 2 + (size_t)B) * (pa::MIX_THREADS / 64) * (size_t)JP * sizeof(double);
}

int pa_mixture_diag_normal_fwd_bwd(int dtype, const void* x, int64_t N, int D, int K, int64_t B, const void* a,
                                   int64_t a_batch_stride, const void* loc, int64_t loc_stride_k,
                                   int64_t loc_stride_d, int64_t loc_batch_stride, const void* scale,
                                   int64_t scale_stride_k, int64_t scale_stride_d, int64_t scale_batch_stride,
                                   void* workspace, size_t workspace_bytes, double* out_padded, pa_stream_t stream) {
  PA_REQUIRE(K >= 1 && K <= pa::MIX_MAXK && D >= 1 && D <= 8, "mixture_diag_normal_fwd_bwd: K=%d (<= %d), D=%d (<= 8)",
             K, pa::MIX_MAXK, D);
  PA_REQUIRE(B >= 1 && B <= 65535, "mixture_diag_normal_fwd_bwd: B=%lld outside [1, 65535]", (long long)B);
  PA_REQUIRE(N >= 1 && x && a && loc && scale && out_padded && workspace, "mixture_diag_normal_fwd_bwd: NULL pointer or N < 1");
  PA_REQUIRE(workspace_bytes >= pa_mixture_diag_normal_workspace(K, D, B), "mixture_diag_normal_fwd_bwd: workspace too small");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    return pa::mixture_diag_t<float>((const float*)x, N, D, K, B, (const float*)a, a_batch_stride, (const float*)loc,
                                     loc_stride_k, loc_stride_d, loc_batch_stride, (const float*)scale, scale_stride_k,
                                     scale_stride_d, scale_batch_stride, (double*)workspace, out_padded, s);
  if (dtype == PA_F64)
    return pa::mixture_diag_t<double>((const double*)x, N, D, K, B, (const double*)a, a_batch_stride,
                                      (const double*)loc, loc_stride_k, loc_stride_d, loc_batch_stride,
                                      (const double*)scale, scale_stride_k, scale_stride_d, scale_batch_stride,
                                      (double*)workspace, out_padded, s);
  return pa::fail(PA_ERR_UNSUPPORTED, "mixture_diag_normal_fwd_bwd: dtype %d", dtype);
}

int pa_mixture_fwd_bwd(int dtype, int dist, const void* x, int64_t N, int K, int64_t B, const void* a,
                       int64_t a_batch_stride, const void* p0, int64_t p0_stride, int64_t p0_batch_stride,
                       const void* p1, int64_t p1_stride, int64_t p1_batch_stride, void* workspace,
                       size_t workspace_bytes, double* out, pa_stream_t stream) {
  PA_REQUIRE(K >= 1 && K <= pa::MIX_MAXK, "mixture_fwd_bwd: K=%d outside [1, %d]", K, pa::MIX_MAXK);
  PA_REQUIRE(B >= 1 && B <= 65535, "mixture_fwd_bwd: B=%lld outside [1, 65535]", (long long)B);
  PA_REQUIRE(N >= 1 && x && a && p0 && out && workspace, "mixture_fwd_bwd: NULL pointer or N < 1");
  PA_REQUIRE(p0_stride >= 0 && p1_stride >= 0 && a_batch_stride >= 0 && p0_batch_stride >= 0 &&
                 p1_batch_stride >= 0, "mixture_fwd_bwd: negative stride");
  PA_REQUIRE(workspace_bytes >= pa_mixture_workspace(K, B), "mixture_fwd_bwd: workspace too small");
  PA_REQUIRE(pa::dist_nparams(dist) == 1 || p1 != nullptr, "mixture_fwd_bwd: the family takes two parameters");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    return pa::mixture_t<float>(dist, (const float*)x, N, K, B, (const float*)a, (const float*)p0, p0_stride,
                                (const float*)p1, p1_stride, a_batch_stride, p0_batch_stride, p1_batch_stride,
                                (double*)workspace, out, s);
  if (dtype == PA_F64)
    return pa::mixture_t<double>(dist, (const double*)x, N, K, B, (const double*)a, (const double*)p0, p0_stride,
                                 (const double*)p1, p1_stride, a_batch_stride, p0_batch_stride, p1_batch_stride,
                                 (double*)workspace, out, s);
  return pa::fail(PA_ERR_UNSUPPORTED, "mixture_fwd_bwd: dtype %d", dtype);
}

}  // extern "C"
