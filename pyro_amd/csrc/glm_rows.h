// glm_rows.h -- the Bernoulli-logits GLM site for FEW particles (P <= 4, incl. the reference's
// default num_particles = 1): the same one-pass forward+backward as glm.hip, on the vector ALU.
//
// With a handful of particles the [P, D] x [D, N] contraction is 2*P*D flops per 4*D-byte row:
// far below the VALU ridge, so the pass is a pure HBM stream and the matrix-core tiling of the
// many-particle kernels (LDS staging, 32-row tiles, operand splitting) only adds latency
// (measured 52 us at N = 1e6, D = 32 for any P <= 32, i.e. 2.5 TB/s).  Here:
//   * LPR = pow2 >= D/4 lanes share a row, each lane holds one float4 of it: a wave-level load
//     instruction reads 64/LPR consecutive rows = one contiguous, fully coalesced span;
//   * the row's dot products are reduced across its LPR lanes with DPP adds (no LDS);
//   * softplus / sigmoid run once per row and particle: lane k of a row evaluates particle
//     k mod PT, the PT gradients are handed back through quad-broadcast DPP;
//   * every lane accumulates gw for ITS four features of all PT particles in registers;
//   * 8 row groups are in flight per wave; block partials are combined in a fixed order.
// Included by glm.hip only.
#pragma once

namespace pa {

constexpr int ROWS_THREADS = 256;
constexpr int ROWS_U = 8;

template <int LPR>
__device__ __forceinline__ float row_lanes_sum(float v) {
  if constexpr (LPR >= 2) v += dpp_get<0xB1>(v);     // lane ^ 1
  if constexpr (LPR >= 4) v += dpp_get<0x4E>(v);     // lane ^ 2
  if constexpr (LPR >= 8) v += dpp_get<0x141>(v);    // row_half_mirror
  if constexpr (LPR >= 16) v += dpp_get<0x140>(v);   // row_mirror
  if constexpr (LPR >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// ll term and d/dlogit of one Bernoulli-logits datum (same arithmetic as glm.hip's exact kernel)
__device__ __forceinline__ void bernoulli_logit_terms(float l, float yv, float& term, float& g) {
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * fabsf(l));
  const float t = 1.0f + e;
  const float lg = 0.69314718055994530942f * __builtin_amdgcn_logf(t);
  const float sp = fmaxf(l, 0.0f) + lg;           // softplus(l)
  const float inv = __builtin_amdgcn_rcpf(t);
  const float sig = l >= 0.0f ? inv : e * inv;    // sigmoid(l)
  term = yv * l - sp;
  g = yv - sig;
}

// part[block][P*D + 2*P]: gw[P, D], ll[P], gb[P] partial sums of the block (unscaled)
template <int PT, int LPR>
__global__ __launch_bounds__(ROWS_THREADS) void glm_rows_kernel(
    const float* __restrict__ X, const float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ b, const uint8_t* __restrict__ mask, int64_t N, int D, int P,
    float* __restrict__ part) {
  constexpr int RPW = 64 / LPR;                     // rows per wave-level load
  constexpr bool SPLIT = (PT > 1) && (LPR >= 4);    // lanes of a row share the particles
  constexpr int NVAL = 4 * PT + (SPLIT ? 2 : 2 * PT);
  __shared__ float sm[NVAL][ROWS_THREADS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane % LPR, r = lane / LPR;
  const int kf = D >> 2;
  const int kc = k < kf ? k : 0;                    // idle lanes re-read column 0 (weights are 0)
  const int mp = SPLIT ? (k & (PT - 1)) : 0;

  float wv[PT][4], bias[PT];
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const bool okp = p < P && k < kf;
    const float4 t = okp ? *reinterpret_cast<const float4*>(w + (int64_t)p * D + 4 * k)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    wv[p][0] = t.x; wv[p][1] = t.y; wv[p][2] = t.z; wv[p][3] = t.w;
    bias[p] = (p < P && b != nullptr) ? b[p] : 0.0f;
  }
  float acc[PT][4];
  float llacc[SPLIT ? 1 : PT], gbacc[SPLIT ? 1 : PT];
#pragma unroll
  for (int p = 0; p < PT; ++p) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[p][c] = 0.0f;
    if (!SPLIT || p == 0) { llacc[SPLIT ? 0 : p] = 0.0f; gbacc[SPLIT ? 0 : p] = 0.0f; }
  }

  const int64_t nchunks = (N + RPW * ROWS_U - 1) / (RPW * ROWS_U);
  for (int64_t chunk = (int64_t)blockIdx.x * (ROWS_THREADS / 64) + wave; chunk < nchunks;
       chunk += (int64_t)gridDim.x * (ROWS_THREADS / 64)) {
    const int64_t row0 = chunk * (RPW * ROWS_U) + r;
    float4 x[ROWS_U];
    float yv[ROWS_U];
    uint8_t mk[ROWS_U];
#pragma unroll
    for (int u = 0; u < ROWS_U; ++u) {
      const int64_t row = row0 + u * RPW;
      const int64_t rc = row < N ? row : 0;         // clamped address, validity applied below
      x[u] = *reinterpret_cast<const float4*>(X + rc * D + 4 * kc);
      yv[u] = y[rc];
      mk[u] = mask != nullptr ? mask[rc] : (uint8_t)1;
    }
#pragma unroll
    for (int u = 0; u < ROWS_U; ++u) {
      const bool keep = (row0 + u * RPW < N) && mk[u] != 0;
      float l[PT];
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        float s = x[u].x * wv[p][0];
        s = fmaf(x[u].y, wv[p][1], s);
        s = fmaf(x[u].z, wv[p][2], s);
        s = fmaf(x[u].w, wv[p][3], s);
        l[p] = row_lanes_sum<LPR>(s) + bias[p];
      }
      float gp[PT];
      if constexpr (SPLIT) {
        float lm = l[0];
#pragma unroll
        for (int p = 1; p < PT; ++p) lm = mp == p ? l[p] : lm;
        float term, g;
        bernoulli_logit_terms(lm, yv[u], term, g);
        // a masked-out (or out-of-range) row contributes exactly 0: scale_and_mask is
        // where(mask, x, 0) (pyro/distributions/util.py:326)
        term = keep ? term : 0.0f;
        g = keep ? g : 0.0f;
        llacc[0] += term;
        gbacc[0] += g;
        gp[0] = dpp_get<0x00>(g);                   // quad lane 0 evaluated particle 0, ...
        if constexpr (PT > 1) gp[1] = dpp_get<0x55>(g);
        if constexpr (PT > 2) gp[2] = dpp_get<0xAA>(g);
        if constexpr (PT > 3) gp[3] = dpp_get<0xFF>(g);
      } else {
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          float term, g;
          bernoulli_logit_terms(l[p], yv[u], term, g);
          term = keep ? term : 0.0f;
          g = keep ? g : 0.0f;
          llacc[p] += term;
          gbacc[p] += g;
          gp[p] = g;
        }
      }
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        acc[p][0] = fmaf(gp[p], x[u].x, acc[p][0]);
        acc[p][1] = fmaf(gp[p], x[u].y, acc[p][1]);
        acc[p][2] = fmaf(gp[p], x[u].z, acc[p][2]);
        acc[p][3] = fmaf(gp[p], x[u].w, acc[p][3]);
      }
    }
  }

  // ---- block record: fixed-order sums over the lanes that hold the same quantity ---------------
#pragma unroll
  for (int p = 0; p < PT; ++p) {
#pragma unroll
    for (int c = 0; c < 4; ++c) sm[4 * p + c][threadIdx.x] = acc[p][c];
    if constexpr (!SPLIT) {
      sm[4 * PT + 2 * p][threadIdx.x] = llacc[p];
      sm[4 * PT + 2 * p + 1][threadIdx.x] = gbacc[p];
    }
  }
  if constexpr (SPLIT) {
    sm[4 * PT][threadIdx.x] = llacc[0];
    sm[4 * PT + 1][threadIdx.x] = gbacc[0];
  }
  __syncthreads();
  const int J = P * D + 2 * P;
  float* rec = part + (int64_t)blockIdx.x * J;
  for (int j = threadIdx.x; j < J; j += ROWS_THREADS) {
    int slot, kk;
    if (j < P * D) {
      const int p = j / D, d = j % D;
      slot = 4 * p + (d & 3);
      kk = d >> 2;
    } else {
      const int q = j - P * D, which = q >= P ? 1 : 0, p = q - which * P;
      // SPLIT: lane k of a row evaluated particle k mod PT, lanes k = p (< PT <= LPR) hold it;
      // otherwise every lane evaluated every particle and lane k = 0 is the one counted
      slot = SPLIT ? 4 * PT + which : 4 * PT + 2 * p + which;
      kk = SPLIT ? p : 0;
    }
    float t = 0.0f;
    for (int i = kk; i < ROWS_THREADS; i += LPR) t += sm[slot][i];
    rec[j] = t;
  }
}

// out[j] = scale * sum_blocks part[block][j] (fp64, fixed order); j: gw[P*D], ll[P], gb[P]
__global__ __launch_bounds__(256) void glm_rows_finalize_kernel(
    const float* __restrict__ part, int nblocks, int D, int P, double scale,
    float* __restrict__ ll, float* __restrict__ gw, float* __restrict__ gb) {
  constexpr int OUT = 8, GROUPS = 32;
  __shared__ double sm[GROUPS][OUT];
  const int jj = threadIdx.x % OUT, s = threadIdx.x / OUT;
  const int J = P * D + 2 * P;
  const int j = blockIdx.x * OUT + jj;
  double acc = 0.0;
  if (j < J) {
    const float* base = part + j;
    float v[8];
    int blk = s;
    for (; blk + 7 * GROUPS < nblocks; blk += 8 * GROUPS) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(blk + u * GROUPS) * J];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; blk < nblocks; blk += GROUPS) acc += (double)base[(int64_t)blk * J];
  }
  sm[s][jj] = acc;
  __syncthreads();
  if (s == 0 && j < J) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) t += sm[g][jj];
    const float v = (float)(t * scale);
    if (j < P * D) gw[j] = v;
    else if (j < P * D + P) ll[j - P * D] = v;
    else gb[j - P * D - P] = v;
  }
}

static bool glm_rows_applicable(const float* X, const float* w, int64_t D, int64_t P) {
  return P <= 4 && D >= 8 && D <= 128 && D % 4 == 0 &&
         ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
}

static int glm_rows_blocks(int64_t N, int64_t D) {
  int lpr = 2;
  while (lpr * 4 < D) lpr *= 2;
  const int64_t rows_per_chunk = (64 / lpr) * ROWS_U;
  const int64_t nchunks = (N + rows_per_chunk - 1) / rows_per_chunk;
  int64_t want = (nchunks + 3) / 4;
  const int64_t cap = (int64_t)cu_count() * 4;      // 16 waves per CU
  if (want > cap) want = cap;
  return (int)(want < 1 ? 1 : want);
}

static size_t glm_rows_workspace_floats(int64_t N, int64_t D, int64_t P) {
  return (size_t)glm_rows_blocks(N, D) * (size_t)(P * D + 2 * P);
}

template <int PT>
static void glm_rows_launch_pt(int lpr, dim3 grid, hipStream_t s, const float* X, const float* y,
                               const float* w, const float* b, const uint8_t* mask, int64_t N,
                               int D, int P, float* part) {
#define PA_ROWS(L)                                                                              \
  hipLaunchKernelGGL((glm_rows_kernel<PT, L>), grid, dim3(ROWS_THREADS), 0, s, X, y, w, b, mask, \
                     N, D, P, part)
  switch (lpr) {
    case 2: PA_ROWS(2); break;
    case 4: PA_ROWS(4); break;
    case 8: PA_ROWS(8); break;
    case 16: PA_ROWS(16); break;
    default: PA_ROWS(32); break;
  }
#undef PA_ROWS
}

static int glm_rows_launch(const float* X, const float* y, const float* w, const float* b,
                           const uint8_t* mask, double scale, int64_t N, int D, int P, float* ll,
                           float* gw, float* gb, float* part, hipStream_t s) {
  int lpr = 2;
  while (lpr * 4 < D) lpr *= 2;
  const int nblocks = glm_rows_blocks(N, D);
  dim3 grid((unsigned)nblocks);
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_GLM, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  if (P == 1) glm_rows_launch_pt<1>(lpr, grid, s, X, y, w, b, mask, N, D, P, part);
  else if (P == 2) glm_rows_launch_pt<2>(lpr, grid, s, X, y, w, b, mask, N, D, P, part);
  else glm_rows_launch_pt<4>(lpr, grid, s, X, y, w, b, mask, N, D, P, part);
  if (br) (void)hipEventRecord(ev1, s);
  int rc = check_launch("glm_rows_kernel");
  if (rc != PA_OK) return rc;
  const int J = P * D + 2 * P;
  hipLaunchKernelGGL(glm_rows_finalize_kernel, dim3((unsigned)((J + 7) / 8)), dim3(256), 0, s, part,
                     nblocks, D, P, scale, ll, gw, gb);
  return check_launch("glm_rows_finalize_kernel");
}

}  // namespace pa
