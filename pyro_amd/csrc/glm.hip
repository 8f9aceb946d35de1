// glm.hip -- fused plated Bernoulli-logits GLM likelihood, forward + gradient in one pass.
//
// Reference path replaced (per ELBO-gradient step, P vectorised particles, plate size N):
//   logits = w @ X^T + b                                  (user model, torch matmul)
//   Bernoulli(logits).log_prob(y) = -BCEWithLogits        (torch: bernoulli.py:121-125)
//   scale_and_mask, .sum()                                (pyro/poutine/trace_struct.py:264-278)
//   backward of all three                                 (pyro/infer/trace_elbo.py:153-157)
// which materialises [P,N] logits / log-probs / grads several times.  Here X and y are read
// ONCE; nothing of size P*N ever reaches HBM.
//
// gfx950 mapping
//   * one wavefront owns a 32-row tile of X staged (padded, stride odd => conflict-free
//     ds_read_b32 in both access directions) in its private slice of LDS;
//   * GEMM1  L[n,p] = sum_d X[n,d] W[p,d]   : v_mfma_f32_32x32x2_f32, A = X tile (LDS),
//     B = W fragments held in VGPRs for the whole kernel;
//   * the C/D register layout of that MFMA (col = lane&31 = p, row = n) is *already* the
//     A-operand layout of GEMM2  gw[p,d] += sum_n G[p,n] X[n,d]  (k-pair {n, n+4} per
//     accumulator register), so G = mask*(y - sigmoid(L)) never leaves registers;
//   * exact f32 MFMA (no xf32/bf16 down-conversion: results are an fmaf chain, see
//     MI355X guide "FP32-input MFMA"), softplus/sigmoid on the VALU overlap the MFMA pipe
//     of the co-resident wave;
//   * deterministic reduction: per-block partials -> fp64 finalize kernel (no float atomics).
#include "common.h"
#include "glm_bf16.h"
#include "glm_planes.h"
#include "glm_planes16.h"
#include "glm_planes16w.h"
#include "glm_planes16d.h"
#include "glm_finalize.h"
#include "chain.h"

namespace pa {

// 0 = automatic: the few-particle vector-ALU kernel (glm_rows.h) for P <= 4, the bf16x3
// split-precision matrix-core kernel (glm_bf16.h) when the layout allows it, else exact f32;
// 1 = always the exact-f32 MFMA kernel below; 2 = as 0 without the few-particle kernel.
// Process-wide; see pa_glm_set_variant.
static int g_glm_variant = 0;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GLM_WAVES = 4;  // waves per workgroup

template <int DT>
struct GlmCfg {
  static constexpr int DP = 32 * DT;      // padded feature count
  static constexpr int S = DP + 3;        // LDS row stride in floats (odd)
  static constexpr int TILE_F = 32 * S;   // floats per wave tile
};


// GROUPED: the hierarchical variant (BASELINE config 5): rows are sorted by group, logits use the
// group's own weights w[p, g, :]; blockIdx.x is a SEGMENT {row_begin, row_end, group} of one
// group's rows (seg[3*blockIdx.x ..]), its four waves take that segment's tiles round-robin, and
// the block's partial record belongs to that group.
template <int DT, int PT, bool VEC4, bool GROUPED>
__global__ __launch_bounds__(64 * GLM_WAVES, 2) void glm_bernoulli_kernel(
    const float* __restrict__ X, const float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ b, const uint8_t* __restrict__ mask, int64_t N, int D, int P,
    int64_t iters, float* __restrict__ part, const int64_t* __restrict__ seg, int G) {
  using C = GlmCfg<DT>;
  constexpr int DP = C::DP, S = C::S, TILE_F = C::TILE_F;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  float* Xs = lds + wave * TILE_F;

  for (int i = threadIdx.x; i < GLM_WAVES * TILE_F; i += 64 * GLM_WAVES) lds[i] = 0.0f;

  int64_t row_begin = 0, row_end = N;
  int group = 0;
  if constexpr (GROUPED) {
    row_begin = seg[3 * (int64_t)blockIdx.x];
    row_end = seg[3 * (int64_t)blockIdx.x + 1];
    group = (int)seg[3 * (int64_t)blockIdx.x + 2];
  }

  // ---- W fragments (B operand of GEMM1) and bias: registers for the whole kernel ----------
  const int pbase = blockIdx.y * 32 * PT;
  float wf[PT][DP / 2];
  float bias[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int p = pbase + pt * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < DP / 2; ++kk) {
      const int d = 2 * kk + h;
      wf[pt][kk] = (p < P && d < D)
                       ? w[(GROUPED ? ((int64_t)p * G + group) : (int64_t)p) * D + d]
                       : 0.0f;
    }
    bias[pt] = (p < P && b != nullptr) ? b[p] : 0.0f;
  }

  f32x16 gwacc[PT][DT];
  float ll_acc[PT], gb_acc[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    ll_acc[pt] = 0.0f;
    gb_acc[pt] = 0.0f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) gwacc[pt][dt][r] = 0.0f;
  }

  // ---- staging registers: the next tile travels global -> VGPR while this one computes ----
  constexpr int NLD = VEC4 ? 4 * DT : 16 * DT;  // loads per lane per tile
  constexpr int EPL = VEC4 ? 4 : 1;             // floats per load
  float stage[NLD * EPL];
  uint8_t stage_m[NLD];        // mask byte of the row each staged load belongs to (1 without a mask)
  float st_y = 0.0f, st_m = 0.0f;
  const int step_e = 64 * EPL;                   // flat-element stride between a lane's loads
  const int q0 = step_e / D, r0 = step_e % D;    // (row, col) increment per load
  const int e0 = lane * EPL;
  const int n_first = e0 / D, d_first = e0 % D;
  const int64_t total_e = row_end * (int64_t)D;
  int nrow[NLD];               // tile-local row of this lane's j-th load
  {
    int n = n_first, d = d_first;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      nrow[j] = n;
      n += q0;
      d += r0;
      if (d >= D) { d -= D; n += 1; }
    }
  }

  // Loads use CLAMPED addresses and are consumed raw; validity is applied in write_stage(), one
  // tile of compute later (a select next to the load makes the compiler wait for it right there
  // and exposes the full HBM latency on every tile).
  auto issue_loads = [&](int64_t tile) {
    const int64_t base = (row_begin + tile * 32) * (int64_t)D;  // flat offset of the tile's first element
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int64_t e = base + e0 + (int64_t)j * step_e;
      const bool ok = ((e0 + j * step_e) < 32 * D) && (e < total_e);
      const int64_t ec = ok ? e : 0;
      if (VEC4) {
        const float4 v = *reinterpret_cast<const float4*>(X + ec);
        stage[4 * j + 0] = v.x; stage[4 * j + 1] = v.y;
        stage[4 * j + 2] = v.z; stage[4 * j + 3] = v.w;
      } else {
        stage[j] = X[ec];
      }
      // a masked row must contribute nothing whatever it holds (where(mask, x, 0),
      // pyro/distributions/util.py:326): its X values are dropped at staging, so that inf-sized
      // garbage cannot turn 0 * l into NaN
      stage_m[j] = (mask != nullptr && ok) ? mask[row_begin + tile * 32 + nrow[j]] : (uint8_t)1;
    }
    const int64_t n = row_begin + tile * 32 + l31;
    const int64_t nc = n < row_end ? n : 0;
    st_y = y[nc];
    st_m = mask == nullptr ? 1.0f : (mask[nc] != 0 ? 1.0f : 0.0f);
  };
  auto write_stage = [&](int64_t tile) {
    const int64_t base = (row_begin + tile * 32) * (int64_t)D;
    int n = n_first, d = d_first;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      if (e0 + j * step_e < 32 * D) {
        const bool ok = (base + e0 + (int64_t)j * step_e < total_e) && stage_m[j] != 0;
#pragma unroll
        for (int k = 0; k < EPL; ++k) Xs[n * S + d + k] = ok ? stage[EPL * j + k] : 0.0f;
      }
      n += q0;
      d += r0;
      if (d >= D) { d -= D; n += 1; }
    }
    if (h == 0) {
      // a masked-out row contributes exactly 0 even if its data are inf/NaN-free garbage:
      // scale_and_mask is where(mask, x, 0) (pyro/distributions/util.py:326)
      const bool okn = row_begin + tile * 32 + l31 < row_end;
      const float m = okn ? st_m : 0.0f;
      Xs[l31 * S + DP] = okn ? m * st_y : 0.0f;
      Xs[l31 * S + DP + 1] = m;
    }
  };

  const int64_t ntiles = (row_end - row_begin + 31) / 32;
  int64_t tile = GROUPED ? (int64_t)wave : (int64_t)blockIdx.x * GLM_WAVES + wave;
  const int64_t tile_stride = GROUPED ? (int64_t)GLM_WAVES : (int64_t)gridDim.x * GLM_WAVES;

  issue_loads(tile);
  __syncthreads();  // LDS zero-fill complete
  write_stage(tile);
  __syncthreads();

  for (int64_t it = 0; it < iters; ++it) {
    const int64_t next = tile + tile_stride;
    if (it + 1 < iters) issue_loads(next);

    if (tile < ntiles) {
      // Per wave and 32-row tile: 16*DT*PT MFMAs (GEMM1) + 16*DT*PT MFMAs (GEMM2), 64 cycles
      // each, own the matrix pipe.  The softplus/sigmoid VALU work is issued BETWEEN MFMAs that
      // do not depend on it (GEMM1 of the other particle tile, GEMM2 of the previous register)
      // so one wave keeps both pipes busy; the co-resident wave fills the remaining bubbles.
      f32x16 acc[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pt][r] = 0.0f;

      auto gemm1_step = [&](int pt, int kk) {
        const float a = Xs[l31 * S + 2 * kk + h];
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wf[pt][kk], acc[pt], 0, 0, 0);
      };
      // element-wise on accumulator register r of particle tile pt (rows nr / nr+4 by lane half):
      //   ll += m*(y*l - softplus(l)),  g = m*(y - sigmoid(l)),  acc[pt][r] := g
      auto elementwise = [&](int pt, int r) {
        const int nr = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float my = Xs[nr * S + DP];
        const float m = Xs[nr * S + DP + 1];
        const float l = acc[pt][r] + bias[pt];
        // e = exp(-|l|) via v_exp_f32 (2^x); t = 1 + e in (1, 2]
        const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * fabsf(l));
        const float t = 1.0f + e;
        // log1p(e) = ln2 * log2(t) via v_log_f32.  Rounding t = 1+e costs <= 6e-8 ABSOLUTE per
        // element (|term| is O(1)), far below the f32 accumulation error of the plate sum.
        const float lg = 0.69314718055994530942f * __builtin_amdgcn_logf(t);
        const float sp = fmaxf(l, 0.0f) + lg;          // softplus(l)
        const float inv = __builtin_amdgcn_rcpf(t);
        const float sig = l >= 0.0f ? inv : e * inv;   // sigmoid(l)
        ll_acc[pt] += my * l - m * sp;
        const float g = my - m * sig;
        gb_acc[pt] += g;
        acc[pt][r] = g;
      };
      auto gemm2_step = [&](int pt, int r) {
        const int nr = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const float bf = Xs[nr * S + dt * 32 + l31];
          gwacc[pt][dt] =
              __builtin_amdgcn_mfma_f32_32x32x2f32(acc[pt][r], bf, gwacc[pt][dt], 0, 0, 0);
        }
      };

#pragma unroll
      for (int kk = 0; kk < DP / 2; ++kk) gemm1_step(0, kk);
      if constexpr (PT == 2) {
        static_assert(PT == 1 || DT == 1, "two particle tiles only with one feature tile");
#pragma unroll
        for (int i = 0; i < 16; ++i) {   // GEMM1(tile 1)  ||  element-wise(tile 0)
          gemm1_step(1, i);
          elementwise(0, i);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // GEMM2(tile 0), GEMM2(tile 1)  ||  element-wise(tile 1)
          gemm2_step(0, r);
          elementwise(1, r);
          gemm2_step(1, r);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // GEMM2 of register r  ||  element-wise of register r+1
          elementwise(0, r);
          gemm2_step(0, r);
        }
      }
    }
    // The LDS slice is private to this wave and a wave's DS operations execute in program order,
    // so re-staging needs no workgroup barrier (waves of a block are free to drift out of phase:
    // one wave's MFMA burst then overlaps the other waves' VALU phase on the shared CU).
    if (it + 1 < iters) write_stage(next);
    tile = next;
  }
  __syncthreads();

  // ---- block reduction over the 4 waves in a fixed order, then one partial record ---------
  constexpr int REC = glm_record_floats<DT, PT>();
  static_assert(PT * DT * 1024 + 2 * PT * 64 <= GLM_WAVES * TILE_F, "LDS too small for epilogue");
  float* red = lds;                       // [PT*DT*16][64] accumulator slots
  float* red2 = lds + PT * DT * 1024;     // [2*PT][64] ll/gb per-lane slots
  for (int wv = 0; wv < GLM_WAVES; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int idx = ((pt * DT + dt) * 16 + r) * 64 + lane;
            red[idx] = (wv == 0 ? 0.0f : red[idx]) + gwacc[pt][dt][r];
          }
        const int i0 = (2 * pt) * 64 + lane, i1 = (2 * pt + 1) * 64 + lane;
        red2[i0] = (wv == 0 ? 0.0f : red2[i0]) + ll_acc[pt];
        red2[i1] = (wv == 0 ? 0.0f : red2[i1]) + gb_acc[pt];
      }
    }
    __syncthreads();
  }
  float* rec = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * REC;
  for (int i = threadIdx.x; i < PT * DT * 1024; i += 64 * GLM_WAVES) rec[i] = red[i];
  // fold the two lane halves (n and n+4 rows) of ll / gb: 32 values per particle tile
  for (int i = threadIdx.x; i < 2 * PT * 32; i += 64 * GLM_WAVES) {
    const int q = i >> 5, j = i & 31;
    rec[PT * DT * 1024 + i] = red2[q * 64 + j] + red2[q * 64 + 32 + j];
  }
}

// Chain rule of the site's two gradient outputs with the upstream gradient g[P] of ll[P]:
//   dw[p, :] = g[p] * gw[p, :],  db[p] = g[p] * gb[p]      (one launch instead of two products)
__global__ __launch_bounds__(256) void glm_chain_kernel(const float* __restrict__ g,
                                                        const float* __restrict__ gw,
                                                        const float* __restrict__ gb, int64_t P,
                                                        int64_t W, float* __restrict__ dw,
                                                        float* __restrict__ db) {
  const int64_t n = P * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + P;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n) {
      if (dw) dw[i] = g[i / W] * gw[i];
    } else if (db) {
      db[i - n] = g[i - n] * gb[i - n];
    }
  }
}

}  // namespace pa
#include "glm_rows.h"
namespace pa {

struct GlmPlan {
  int DT, PT, npass, nblocks, rec;
  int64_t iters;
  size_t lds_bytes;
};

static GlmPlan glm_plan(int64_t N, int64_t D, int64_t P) {
  GlmPlan pl;
  pl.DT = D <= 32 ? 1 : (D <= 64 ? 2 : 4);
  pl.PT = (pl.DT == 1 && P > 32) ? 2 : 1;
  pl.npass = (int)((P + 32 * pl.PT - 1) / (32 * pl.PT));
  const int64_t ntiles = (N + 31) / 32;
  int64_t want = (ntiles + GLM_WAVES - 1) / GLM_WAVES;
  int64_t cap = (int64_t)cu_count() * 2 / pl.npass;  // 2 workgroups (8 waves) per CU in flight
  if (cap < 1) cap = 1;
  pl.nblocks = (int)(want < cap ? (want < 1 ? 1 : want) : cap);
  pl.iters = (ntiles + (int64_t)pl.nblocks * GLM_WAVES - 1) / ((int64_t)pl.nblocks * GLM_WAVES);
  if (pl.iters < 1) pl.iters = 1;
  pl.rec = pl.PT * pl.DT * 1024 + 2 * pl.PT * 32;
  pl.lds_bytes = (size_t)GLM_WAVES * 32 * (32 * pl.DT + 3) * sizeof(float);
  return pl;
}

template <int DT, int PT>
static int glm_launch(const GlmPlan& pl, const float* X, const float* y, const float* w,
                      const float* b, const uint8_t* mask, double scale, int64_t N, int D, int P,
                      float* ll, float* gw, float* gb, float* part, hipStream_t s) {
  const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  dim3 grid((unsigned)pl.nblocks, (unsigned)pl.npass), block(64 * GLM_WAVES);
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_GLM, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  // D > 64 (DT = 4): the split-precision kernel would need 64 gradient-accumulator registers per
  // lane on top of its three operand planes and spills (measured 2.1x SLOWER than the exact-f32
  // kernel at D = 128); that shape stays on the exact kernel.
  if (vec4 && g_glm_variant != 1 && DT < 4) {
    if constexpr (DT < 4) {
      auto k = mask != nullptr ? glm_bernoulli_bf16_kernel<DT, PT, false, true>
                               : glm_bernoulli_bf16_kernel<DT, PT, false, false>;
      constexpr int lds = GlmBfCfg<DT, PT>::LDS_BYTES;
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(k, grid, block, lds, s, X, y, w, b, mask, N, D, P, pl.iters, part,
                         (const int64_t*)nullptr, 1);
    }
  } else if (vec4) {
    auto k = glm_bernoulli_kernel<DT, PT, true, false>;
    if (pl.lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)pl.lds_bytes);
    hipLaunchKernelGGL(k, grid, block, pl.lds_bytes, s, X, y, w, b, mask, N, D, P, pl.iters, part,
                       (const int64_t*)nullptr, 1);
  } else {
    auto k = glm_bernoulli_kernel<DT, PT, false, false>;
    if (pl.lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)pl.lds_bytes);
    hipLaunchKernelGGL(k, grid, block, pl.lds_bytes, s, X, y, w, b, mask, N, D, P, pl.iters, part,
                       (const int64_t*)nullptr, 1);
  }
  if (br) (void)hipEventRecord(ev1, s);
  int rc = check_launch("glm_bernoulli_kernel");
  if (rc != PA_OK) return rc;
  const int64_t J = (int64_t)P * D + 2 * P;
  rc = chain_record_fin((pa_stream_t)s, DT, PT, part, pl.nblocks, pl.npass, D, P, scale, ll, gw, gb,
                        0.0);
  if (rc != 0) return rc < 0 ? rc : PA_OK;       // recorded as a phase of the step's chained tail
  hipLaunchKernelGGL((glm_finalize_kernel<DT, PT>), dim3((unsigned)((J + FIN_OUT - 1) / FIN_OUT)),
                     dim3(FIN_OUT * FIN_GROUPS), 0, s, part, pl.nblocks, pl.npass, D, P, scale, ll,
                     gw, gb, 0.0);
  return check_launch("glm_finalize_kernel");
}


// ---- grouped (hierarchical) variant ------------------------------------------------------------
// gw[p, g, d] = scale * sum over the segments of group g; ll[p], gb[p] = scale * sum over ALL
// segments.  fp64 accumulation in a fixed order.
template <int DT, int PT>
__device__ __forceinline__ void glm_grouped_finalize_gw(
    unsigned bid, const float* __restrict__ part, const int64_t* __restrict__ group_seg_off, int nseg, int D,
    int P, int G, double scale, float* __restrict__ gw) {
  constexpr int REC = glm_record_floats<DT, PT>();
  const int64_t J = (int64_t)P * G * D;
  const int64_t j = (int64_t)bid * 256 + threadIdx.x;
  if (j >= J) return;
  const int d = (int)(j % D);
  const int g = (int)((j / D) % G);
  const int p = (int)(j / ((int64_t)D * G));
  const int pl = p % (32 * PT), pt = pl >> 5, i = pl & 31, dt = d >> 5, c = d & 31;
  const int hh = (i >> 2) & 1, reg = (i & 3) + 4 * (i >> 3);
  const int slot = ((pt * DT + dt) * 16 + reg) * 64 + c + 32 * hh;
  const int pass = p / (32 * PT);
  const float* base = part + (int64_t)pass * nseg * REC + slot;
  double acc = 0.0;
  for (int64_t sgi = group_seg_off[g]; sgi < group_seg_off[g + 1]; ++sgi)
    acc += (double)base[sgi * REC];
  gw[j] = (float)(acc * scale);
}

template <int DT, int PT>
__device__ __forceinline__ void glm_grouped_finalize_scalar(
    unsigned bid, const float* __restrict__ part, int nseg, int P, double scale, float* __restrict__ ll,
    float* __restrict__ gb, double ll_offset) {
  constexpr int REC = glm_record_floats<DT, PT>();
  __shared__ double smem[16];
  const int which = bid >= (unsigned)P ? 1 : 0;       // 0: ll, 1: gb
  const int p = bid - which * P;
  const int pl = p % (32 * PT), pt = pl >> 5, i = pl & 31;
  const int slot = PT * DT * 1024 + (2 * pt + which) * 32 + i;
  const int pass = p / (32 * PT);
  const float* base = part + (int64_t)pass * nseg * REC + slot;
  double acc = 0.0;
  for (int sgi = threadIdx.x; sgi < nseg; sgi += 256) acc += (double)base[(int64_t)sgi * REC];
  const double t = block_sum_f64(acc, smem);
  if (threadIdx.x == 0) (which ? gb : ll)[p] = (float)((t + (which ? 0.0 : ll_offset)) * scale);
}

// one launch: the first `gw_blocks` workgroups write gw, the next 2 P reduce ll and gb
template <int DT, int PT>
__global__ __launch_bounds__(256) void glm_grouped_finalize_kernel(
    const float* __restrict__ part, const int64_t* __restrict__ group_seg_off, int nseg, int D, int P, int G,
    double scale, float* __restrict__ gw, float* __restrict__ ll, float* __restrict__ gb, double ll_offset,
    unsigned gw_blocks) {
  if (blockIdx.x < gw_blocks)
    glm_grouped_finalize_gw<DT, PT>(blockIdx.x, part, group_seg_off, nseg, D, P, G, scale, gw);
  else
    glm_grouped_finalize_scalar<DT, PT>(blockIdx.x - gw_blocks, part, nseg, P, scale, ll, gb, ll_offset);
}

template <int DT, int PT>
static void glm_grouped_finalize_launch(const float* part, const int64_t* group_seg_off, int nseg, int D, int P,
                                        int G, double scale, float* gw, float* ll, float* gb, double ll_offset,
                                        hipStream_t s) {
  const int64_t J = (int64_t)P * G * D;
  const unsigned gw_blocks = (unsigned)((J + 255) / 256);
  hipLaunchKernelGGL((glm_grouped_finalize_kernel<DT, PT>), dim3(gw_blocks + (unsigned)(2 * P)), dim3(256), 0, s,
                     part, group_seg_off, nseg, D, P, G, scale, gw, ll, gb, ll_offset, gw_blocks);
}

template <int DT, int PT>
static int glm_grouped_launch(const float* X, const float* y, const float* w, const float* b,
                              const uint8_t* mask, double scale, int64_t N, int D, int P, int G,
                              const int64_t* seg, int nseg, const int64_t* group_seg_off,
                              int64_t max_seg_rows, float* ll, float* gw, float* gb, float* part,
                              hipStream_t s) {
  const bool vec4 = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  const int npass = (P + 32 * PT - 1) / (32 * PT);
  const int64_t seg_tiles = (max_seg_rows + 31) / 32;
  const int64_t iters = (seg_tiles + GLM_WAVES - 1) / GLM_WAVES;
  const size_t lds_bytes = (size_t)GLM_WAVES * 32 * (32 * DT + 3) * sizeof(float);
  dim3 grid((unsigned)nseg, (unsigned)npass), block(64 * GLM_WAVES);
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_GLM, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  if (vec4 && g_glm_variant != 1 && DT < 4) {
    if constexpr (DT < 4) {
      auto k = mask != nullptr ? glm_bernoulli_bf16_kernel<DT, PT, true, true>
                               : glm_bernoulli_bf16_kernel<DT, PT, true, false>;
      constexpr int lds = GlmBfCfg<DT, PT>::LDS_BYTES;
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(k, grid, block, lds, s, X, y, w, b, mask, N, D, P, iters, part, seg, G);
    }
  } else if (vec4) {
    auto k = glm_bernoulli_kernel<DT, PT, true, true>;
    if (lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
    hipLaunchKernelGGL(k, grid, block, lds_bytes, s, X, y, w, b, mask, N, D, P, iters, part, seg, G);
  } else {
    auto k = glm_bernoulli_kernel<DT, PT, false, true>;
    if (lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
    hipLaunchKernelGGL(k, grid, block, lds_bytes, s, X, y, w, b, mask, N, D, P, iters, part, seg, G);
  }
  if (br) (void)hipEventRecord(ev1, s);
  int rc = check_launch("glm_bernoulli_kernel<grouped>");
  if (rc != PA_OK) return rc;
  glm_grouped_finalize_launch<DT, PT>(part, group_seg_off, nseg, D, P, G, scale, gw, ll, gb, 0.0, s);
  return check_launch("glm_grouped_finalize");
}

// ---- design matrix kept as its bf16 planes (glm_planes.h) -----------------------------------------
// developer knobs (pa_glm_planes_tune): ring depth and workgroups per CU; 0 = defaults
static int g_planes_nb = 3;
static int g_planes_bpc = 0;

static int64_t glm_planes_tiles(int64_t N) {
  const int64_t t = (N + 31) / 32;
  return (t + GLMP_PAD_TILES - 1) / GLMP_PAD_TILES * GLMP_PAD_TILES;
}

struct GlmPlanesPlan {
  int nb, bpc, npass, nblocks;
  int64_t nst;
  int nrt = 2, npt = 2;     // waves per workgroup of the f16 kernel: row tiles x particle tiles (glm_planes16.h)
  int ypass = 0;            // grid.y (0: = npass)
};

static GlmPlanesPlan glm_planes_plan(int64_t N, int64_t P) {
  GlmPlanesPlan pl;
  pl.nb = g_planes_nb;
  const int bpc_max = pl.nb == 3 ? 3 : 2;                         // what the LDS ring admits
  pl.bpc = g_planes_bpc > 0 && g_planes_bpc < bpc_max ? g_planes_bpc : bpc_max;
  pl.npass = (int)((P + 63) / 64);
  pl.nst = ((N + 31) / 32 + 1) / 2;                               // 64-row super-tiles
  int64_t cap = (int64_t)cu_count() * pl.bpc / pl.npass;
  if (cap < 1) cap = 1;
  pl.nblocks = (int)(pl.nst < cap ? (pl.nst < 1 ? 1 : pl.nst) : cap);
  return pl;
}

// 0 (default) = the stand-alone finalize launch (272 workgroups pull the records in parallel: ~6 us,
// or a phase of the chained tail); 1 = inside the kernel: measured SLOWER on the MI355X (the GLM
// kernel 68 -> 105 us at the headline size): the two serial last-arriver sums are made by ONE
// workgroup each, 9 dependent rounds of loads at ~2 us per round (data fresh from other XCDs), where
// the separate launch has the whole chip's memory-level parallelism.  Kept as a measured negative
// result and for small plates (pa_glm_planes_finalize_mode).
static int g_planes_fin_mode = 0;
static unsigned long long* g_planes_stamps = nullptr;      // pa_glm_planes_stamps

template <int NB, int OCC>
static void glm_planes_launch_one(const GlmPlanesPlan& pl, const unsigned char* img, const float* y,
                                  const float* w, const float* b, int64_t N, int D, int P,
                                  float* part, const GlmFinArgs& fin, hipStream_t s) {
  auto k = glm_planes_kernel<2, NB, OCC>;
  constexpr int lds = GlmPlCfg<2, NB>::LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k, dim3((unsigned)pl.nblocks, (unsigned)pl.npass), dim3(256), lds, s, img, y, w,
                     b, N, D, P, pl.nst, part, cu_count(), fin, GlmGroupArgs{nullptr, nullptr, 1});
}

// the hierarchical variant: one workgroup per segment
static void glm_planes_launch_grouped(int nseg, int npass, const unsigned char* img,
                                      const float* y_img, const float* w, const float* b, int64_t N,
                                      int D, int P, int64_t nst_total, float* part,
                                      const GlmGroupArgs& grp, hipStream_t s) {
  auto k = glm_planes_kernel<2, 3, 3, true>;
  constexpr int lds = GlmPlCfg<2, 3>::LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  GlmFinArgs fin;
  fin.counters = nullptr;
  fin.part64 = nullptr;
  fin.ll = fin.gw = fin.gb = nullptr;
  fin.scale = fin.ll_offset = 0.0;
  fin.D = D; fin.P = P;
  fin.tstamps = g_planes_stamps;
  hipLaunchKernelGGL(k, dim3((unsigned)nseg, (unsigned)npass), dim3(256), lds, s, img, y_img, w, b, N,
                     D, P, nst_total, part, cu_count(), fin, grp);
}

// ---- the two-plane f16 image (glm_planes16.h): 8 KiB super-tiles, four workgroups per CU ----------
static int64_t glmh_tile_bytes(int64_t ntiles) { return ntiles * (int64_t)GLMH_TILE; }

static bool g_planes_wide = true;      // pa_glm_planes_tune(11, .): 2 x 2 waves whatever P (measurement knob)
static int g_planes_wide_max = 8;      // pa_glm_planes_tune(12, .): at most 2 x 4 waves (128 particles per pass)
static bool g_planes_wide_force = false;   // pa_glm_planes_tune(13, .): the wide geometries at every N (tests)

static GlmPlanesPlan glmh_plan(int64_t N, int64_t P, bool allow_wide = true) {
  GlmPlanesPlan pl;
  pl.nb = g_planes_nb;
  // measured at the headline size (tools/bench_glm_planes.py, kernel + finalize): 2 workgroups per
  // CU 57.6 us, 3: 59.5, 4: 62.4 -- the loop is issue-bound, more waves only add barrier waits
  pl.bpc = g_planes_bpc > 0 && g_planes_bpc <= 4 ? g_planes_bpc : 2;
  pl.npass = (int)((P + 63) / 64);
  pl.nst = ((N + 31) / 32 + 1) / 2;
  int64_t cap = (int64_t)cu_count() * pl.bpc / pl.npass;
  if (cap < 1) cap = 1;
  pl.nblocks = (int)(pl.nst < cap ? (pl.nst < 1 ? 1 : pl.nst) : cap);
  pl.ypass = pl.npass;
  // Measured (tools/bench_glm_particles.py, kernel + finalize, us): at N = 1e5 the 2 x 2 passes win (P = 256: 31.9
  // against 34.2 / 46.3 for 2 x 4 / 1 x 8: with ~12 tiles per workgroup the wider workgroup's prologue -- W planes
  // of 128 / 256 particles -- and its 8-wave barrier are not amortised); at N = 1e6 the wide geometries are equal
  // or better (P = 256: 174 / 168 / 174, P = 1024: 658 / 620 / 616) and read the image once (1.10 x the
  // algorithmic bytes at P = 256 instead of 2.1 x).  So: wide only from ~48 tiles per workgroup on.
  const bool enough_tiles = (N + 31) / 32 >= (int64_t)48 * cu_count();
  if (allow_wide && g_planes_wide && pl.nb == 3 && g_planes_bpc <= 0 && P > 64 &&
      (enough_tiles || g_planes_wide_force)) {
    // many particles / chains: 128 or 256 of them per pass over the image, eight waves per workgroup, one
    // workgroup per CU (the same two waves per SIMD).  The records keep the 64-particle format: npass groups
    const bool eight = P > 128 && g_planes_wide_max >= 8;
    pl.nrt = eight ? 1 : 2;
    pl.npt = eight ? 8 : 4;
    const int64_t wrows = 32 * pl.npt;
    pl.ypass = (int)((P + wrows - 1) / wrows);
    pl.nst = pl.nrt == 1 ? (N + 31) / 32 : ((N + 31) / 32 + 1) / 2;
    pl.bpc = 1;
    cap = (int64_t)cu_count() / pl.ypass;
    if (cap < 1) cap = 1;
    pl.nblocks = (int)(pl.nst < cap ? (pl.nst < 1 ? 1 : pl.nst) : cap);
  }
  return pl;
}

// the wide geometries: 2 x 4 / 1 x 8 waves (LIN only: a guide draw in the prologue stays with 2 x 2)
template <int NRT, int NPT, bool LIN>
static void glmh_launch_wide(const GlmPlanesPlan& pl, const unsigned char* img, const float* y, const float* w,
                             const float* b, int64_t N, int D, int P, float* part, const uint32_t* trailer,
                             hipStream_t s, const double* moments) {
  auto k = glm_planes_f16_kernel<3, 1, false, false, LIN, false, NRT, NPT>;
  constexpr int lds = GlmHCfg<3, false, NRT, NPT>::LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k, dim3((unsigned)pl.nblocks, (unsigned)pl.ypass), dim3(64 * NRT * NPT), lds, s, img, y, w, b,
                     N, D, P, pl.nst, part, cu_count(), trailer, g_planes_stamps, GlmGroupArgs{nullptr, nullptr, 1},
                     gate_word(), moments, GlmDraw{});
  gate_aware_launch();
}

template <int NB, int OCC, bool PRIV = false, bool LIN = false, bool DRAW = false>
static void glmh_launch_one(const GlmPlanesPlan& pl, const unsigned char* img, const float* y,
                            const float* w, const float* b, int64_t N, int D, int P, float* part,
                            const uint32_t* trailer, hipStream_t s, const double* moments = nullptr,
                            const GlmDraw& draw = GlmDraw{}) {
  auto k = glm_planes_f16_kernel<NB, OCC, false, PRIV, LIN, DRAW>;
  constexpr int lds = GlmHCfg<NB, PRIV>::LDS_BYTES + (DRAW ? 256 : 0);     // + the workgroup's softplus(rho)
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k, dim3((unsigned)pl.nblocks, (unsigned)pl.npass), dim3(256), lds, s, img, y, w,
                     b, N, D, P, pl.nst, part, cu_count(), trailer, g_planes_stamps,
                     GlmGroupArgs{nullptr, nullptr, 1}, gate_word(), moments, draw);
  gate_aware_launch();
}

// 32 < D <= 128 (glm_planes16d.h): kernel + finalize (a phase of the chained tail where the tail knows
// the record shape, its own launch otherwise)
template <int DT, int OCC>
static int glmd_launch(const unsigned char* img, const float* y, const float* w, const float* b, int64_t N,
                       int D, int P, float* part, const uint32_t* trailer, int bpc, int* nblocks_out,
                       int64_t* nst_out, hipStream_t s) {
  auto k = glm_planes_f16d_kernel<DT, 3, OCC>;
  constexpr int lds = GlmDCfg<DT, 3>::LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int npass = (int)((P + 63) / 64);
  const int64_t nst = ((N + 31) / 32 + 1) / 2;
  int64_t cap = (int64_t)cu_count() * bpc / npass;
  if (cap < 1) cap = 1;
  const int nblocks = (int)(nst < cap ? (nst < 1 ? 1 : nst) : cap);
  *nblocks_out = nblocks;
  *nst_out = nst;
  hipLaunchKernelGGL(k, dim3((unsigned)nblocks, (unsigned)npass), dim3(256), lds, s, img, y, w, b, N, D, P,
                     nst, part, trailer, gate_word());
  gate_aware_launch();
  return check_launch("glm_planes_f16d_kernel");
}

static int glmd_run(const void* planes, const float* y, const float* w, const float* b, double scale,
                    int64_t N, int D, int P, float* ll, float* gw, float* gb, float* part,
                    pa_stream_t stream, hipStream_t s) {
  const int DT = D <= 64 ? 2 : 4;
  const unsigned char* img = (const unsigned char*)planes;
  const uint32_t* trailer = (const uint32_t*)(img + glmh_tile_bytes(glm_planes_tiles(N)) * DT);
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_GLM, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  int nblocks = 1;
  int64_t nst = 0;
  // two workgroups per CU with two feature tiles (68 KiB of LDS each), one with four (132 KiB)
  int rc = DT == 2 ? glmd_launch<2, 2>(img, y, w, b, N, D, P, part, trailer, 2, &nblocks, &nst, s)
                   : glmd_launch<4, 1>(img, y, w, b, N, D, P, part, trailer, 1, &nblocks, &nst, s);
  if (br) (void)hipEventRecord(ev1, s);
  if (rc != PA_OK) return rc;
  const int npass = (P + 63) / 64;
  const double ll_offset = (double)(nst * 64 - N) * 0.6931471805599453;
  rc = chain_record_fin(stream, DT, 2, part, nblocks, npass, D, P, scale, ll, gw, gb, ll_offset);
  if (rc != 0) return rc < 0 ? rc : PA_OK;
  const int64_t J = (int64_t)P * D + 2 * P;
  const dim3 fgrid((unsigned)((J + FIN_OUT - 1) / FIN_OUT)), fblock(FIN_OUT * FIN_GROUPS);
  if (DT == 2)
    hipLaunchKernelGGL((glm_finalize_kernel<2, 2>), fgrid, fblock, 0, s, part, nblocks, npass, D, P, scale, ll,
                       gw, gb, ll_offset);
  else
    hipLaunchKernelGGL((glm_finalize_kernel<4, 2>), fgrid, fblock, 0, s, part, nblocks, npass, D, P, scale, ll,
                       gw, gb, ll_offset);
  return check_launch("glm_finalize_kernel");
}

// one wave per 32-row tile and 64 particles (glm_planes16w.h): groups of four tiles (128 rows)
template <int NB>
static void glmw_launch(int bpc, const unsigned char* img, const float* y, const float* w, const float* b,
                        int64_t N, int D, int P, float* part, const uint32_t* trailer, int* nblocks_out,
                        hipStream_t s) {
  auto k = glm_planes_f16w_kernel<NB>;
  constexpr int lds = GlmWCfg<NB>::LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int npass = (int)((P + 63) / 64);
  const int64_t ngrp = (N + 127) / 128;
  int64_t cap = (int64_t)cu_count() * bpc / npass;
  if (cap < 1) cap = 1;
  const int nblocks = (int)(ngrp < cap ? ngrp : cap);
  *nblocks_out = nblocks;
  hipLaunchKernelGGL(k, dim3((unsigned)nblocks, (unsigned)npass), dim3(256), lds, s, img, y, w, b, N, D, P,
                     ngrp, part, trailer, g_planes_stamps, gate_word());
  gate_aware_launch();
}

template <int OCC>
static void glmh_launch_grouped(int nseg, int npass, const unsigned char* img, const float* y_img,
                                const float* w, const float* b, int64_t N, int D, int P,
                                int64_t nst_total, float* part, const uint32_t* trailer,
                                const GlmGroupArgs& grp, hipStream_t s) {
  auto k = glm_planes_f16_kernel<3, OCC, true>;
  constexpr int lds = GlmHCfg<3>::LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k, dim3((unsigned)nseg, (unsigned)npass), dim3(256), lds, s, img, y_img, w, b, N,
                     D, P, nst_total, part, cu_count(), trailer, g_planes_stamps, grp, gate_word(),
                     (const double*)nullptr, GlmDraw{});
  gate_aware_launch();
}

// max |X| -> the image trailer {bits, kx}; the pack kernels derive kx from it
static int glmh_absmax(const float* X, int64_t N, int D, uint32_t* trailer, hipStream_t s) {
  if (hipMemsetAsync(trailer, 0, GLMH_TRAILER, s) != hipSuccess)
    return pa::fail(PA_ERR_LAUNCH, "glm_pack_planes: memset of the image trailer failed");
  if (N == 0) return PA_OK;
  int64_t grid = (N + 8 * 8 - 1) / (8 * 8);               // 8 rows per step and workgroup
  const int64_t cap = (int64_t)cu_count() * 8;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(glm_absmax_kernel, dim3((unsigned)grid), dim3(256), 0, s, X, N, D, trailer);
  return pa::check_launch("glm_absmax_kernel");
}

// ---- in-kernel finalize (glm_planes.h): the arrival counters --------------------------------------
// One zeroed block per device, allocated the first time a plane image is packed (never inside a
// stream capture) and kept: launches leave the counters zero.  One plane-image launch per device at
// a time (stream order), as everywhere in this library.
constexpr int GLMF_MAX_PASSES = 32;
static uint32_t* g_glmf_counters[64] = {nullptr};

static uint32_t* glmf_counters(bool may_allocate) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (g_glmf_counters[dev] == nullptr && may_allocate) {
    void* p = nullptr;
    const size_t bytes = (size_t)GLMF_MAX_PASSES * GLMF_CNT_STRIDE * sizeof(uint32_t);
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, bytes) != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    g_glmf_counters[dev] = (uint32_t*)p;
  }
  return g_glmf_counters[dev];
}

static void glm_tiles_of(int64_t D, int64_t P, int* DT, int* PT) {
  *DT = D <= 32 ? 1 : (D <= 64 ? 2 : 4);
  *PT = (*DT == 1 && P > 32) ? 2 : 1;
}

}  // namespace pa

extern "C" {

int pa_glm_chain(const float* g, const float* gw, const float* gb, int64_t P, int64_t W, float* dw,
                 float* db, pa_stream_t stream) {
  PA_REQUIRE(P >= 0 && W >= 0, "glm_chain: bad shape P=%lld W=%lld", (long long)P, (long long)W);
  if (P == 0) return PA_OK;
  PA_REQUIRE(g && (!dw || gw) && (!db || gb), "glm_chain: NULL input");
  int64_t grid = (P * W + P + 255) / 256;
  const int64_t cap = (int64_t)pa::cu_count() * 8;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(pa::glm_chain_kernel, dim3((unsigned)grid), dim3(256), 0, pa::as_stream(stream),
                     g, gw, gb, P, W, dw, db);
  return pa::check_launch("glm_chain_kernel");
}

int pa_glm_set_variant(int variant) {
  PA_REQUIRE(variant >= 0 && variant <= 2,
             "glm_set_variant: expected 0 (automatic), 1 (exact f32) or 2 (bf16x3 matrix cores)");
  pa::g_glm_variant = variant;
  return PA_OK;
}

size_t pa_glm_bernoulli_workspace(int64_t N, int64_t D, int64_t P) {
  if (N < 0 || D < 1 || D > 128 || P < 1) return 0;
  pa::GlmPlan pl = pa::glm_plan(N, D, P);
  size_t floats = (size_t)pl.nblocks * pl.npass * pl.rec;
  if (P <= 4) {   // the few-particle kernel's records (glm_rows.h) share the workspace
    const size_t rows = pa::glm_rows_workspace_floats(N, D, P);
    if (rows > floats) floats = rows;
  }
  return floats * sizeof(float);
}

int pa_glm_bernoulli_fwd_bwd(const float* X, const float* y, const float* w, const float* b,
                             const uint8_t* mask, double scale, int64_t N, int64_t D, int64_t P,
                             float* ll, float* gw, float* gb, void* workspace,
                             size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(N >= 0 && D >= 1 && P >= 1, "glm_bernoulli: bad shape N=%lld D=%lld P=%lld",
             (long long)N, (long long)D, (long long)P);
  if (D > 128)
    return pa::fail(PA_ERR_UNSUPPORTED, "glm_bernoulli: fused kernel supports D <= 128 (got %lld)",
                    (long long)D);
  PA_REQUIRE(N < (int64_t(1) << 40) && P < (1 << 20), "glm_bernoulli: shape too large");
  PA_REQUIRE(w && ll && gw && gb, "glm_bernoulli: NULL parameter/output pointer");
  PA_REQUIRE(N == 0 || (X && y), "glm_bernoulli: NULL data pointer");
  PA_REQUIRE(workspace && workspace_bytes >= pa_glm_bernoulli_workspace(N, D, P),
             "glm_bernoulli: workspace too small (%zu < %zu)", workspace_bytes,
             pa_glm_bernoulli_workspace(N, D, P));
  pa::GlmPlan pl = pa::glm_plan(N, D, P);
  hipStream_t s = pa::as_stream(stream);
  if (N == 0) {  // empty plate: every sum is 0 (torch: sum over an empty tensor)
    hipError_t e1 = hipMemsetAsync(ll, 0, (size_t)P * 4, s);
    hipError_t e2 = hipMemsetAsync(gw, 0, (size_t)P * D * 4, s);
    hipError_t e3 = hipMemsetAsync(gb, 0, (size_t)P * 4, s);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
      return pa::fail(PA_ERR_LAUNCH, "glm_bernoulli: memset failed");
    return PA_OK;
  }
  float* part = (float*)workspace;
  if (pa::g_glm_variant == 0 && pa::glm_rows_applicable(X, w, D, P))
    return pa::glm_rows_launch(X, y, w, b, mask, scale, N, (int)D, (int)P, ll, gw, gb, part, s);
#define PA_GLM_CASE(DT_, PT_)                                                                    \
  if (pl.DT == DT_ && pl.PT == PT_)                                                              \
    return pa::glm_launch<DT_, PT_>(pl, X, y, w, b, mask, scale, N, (int)D, (int)P, ll, gw, gb,  \
                                    part, s);
  PA_GLM_CASE(1, 1)
  PA_GLM_CASE(1, 2)
  PA_GLM_CASE(2, 1)
  PA_GLM_CASE(4, 1)
#undef PA_GLM_CASE
  return pa::fail(PA_ERR_UNSUPPORTED, "glm_bernoulli: no kernel for DT=%d PT=%d", pl.DT, pl.PT);
}

size_t pa_glm_bernoulli_grouped_workspace(int64_t nseg, int64_t D, int64_t P) {
  if (nseg < 0 || D < 1 || D > 128 || P < 1) return 0;
  int DT, PT;
  pa::glm_tiles_of(D, P, &DT, &PT);
  const int64_t npass = (P + 32 * PT - 1) / (32 * PT);
  const int64_t rec = (int64_t)PT * DT * 1024 + 2 * PT * 32;
  return (size_t)(nseg * npass * rec) * sizeof(float);
}

int pa_glm_bernoulli_grouped_fwd_bwd(const float* X, const float* y, const float* w,
                                     const float* b, const uint8_t* mask, double scale, int64_t N,
                                     int64_t D, int64_t P, int64_t G, const int64_t* seg,
                                     int64_t nseg, const int64_t* group_seg_off,
                                     int64_t max_seg_rows, float* ll, float* gw, float* gb,
                                     void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(N >= 0 && D >= 1 && P >= 1 && G >= 1 && nseg >= 0 && max_seg_rows >= 0,
             "glm_grouped: bad shape N=%lld D=%lld P=%lld G=%lld nseg=%lld", (long long)N,
             (long long)D, (long long)P, (long long)G, (long long)nseg);
  if (D > 128)
    return pa::fail(PA_ERR_UNSUPPORTED, "glm_grouped: fused kernel supports D <= 128 (got %lld)",
                    (long long)D);
  PA_REQUIRE(nseg < (1 << 30) && P < (1 << 20) && G < (1 << 24), "glm_grouped: shape too large");
  PA_REQUIRE(w && ll && gw && gb, "glm_grouped: NULL parameter/output pointer");
  hipStream_t s = pa::as_stream(stream);
  if (N == 0 || nseg == 0) {
    hipError_t e1 = hipMemsetAsync(ll, 0, (size_t)P * 4, s);
    hipError_t e2 = hipMemsetAsync(gw, 0, (size_t)P * G * D * 4, s);
    hipError_t e3 = hipMemsetAsync(gb, 0, (size_t)P * 4, s);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
      return pa::fail(PA_ERR_LAUNCH, "glm_grouped: memset failed");
    return PA_OK;
  }
  PA_REQUIRE(X && y && seg && group_seg_off, "glm_grouped: NULL data pointer");
  PA_REQUIRE(workspace && workspace_bytes >= pa_glm_bernoulli_grouped_workspace(nseg, D, P),
             "glm_grouped: workspace too small");
  int DT, PT;
  pa::glm_tiles_of(D, P, &DT, &PT);
  float* part = (float*)workspace;
#define PA_GLMG_CASE(DT_, PT_)                                                                   \
  if (DT == DT_ && PT == PT_)                                                                    \
    return pa::glm_grouped_launch<DT_, PT_>(X, y, w, b, mask, scale, N, (int)D, (int)P, (int)G,  \
                                            seg, (int)nseg, group_seg_off, max_seg_rows, ll, gw, \
                                            gb, part, s);
  PA_GLMG_CASE(1, 1)
  PA_GLMG_CASE(1, 2)
  PA_GLMG_CASE(2, 1)
  PA_GLMG_CASE(4, 1)
#undef PA_GLMG_CASE
  return pa::fail(PA_ERR_UNSUPPORTED, "glm_grouped: no kernel for DT=%d PT=%d", DT, PT);
}

#define PA_REQUIRE_FORMAT(f, who)                                                   \
  PA_REQUIRE((f) == PA_GLM_PLANES_BF16X3 || (f) == PA_GLM_PLANES_F16X2, who ": unknown image format %d", (f))

size_t pa_glm_grouped_planes_bytes(int format, int64_t nst_total, int64_t D) {
  if (nst_total < 0 || D < 1 || D > 32) return 0;
  // the tile image, then the observations in the image's padded row order (f16: then the trailer)
  if (format == PA_GLM_PLANES_F16X2)
    return (size_t)nst_total * 2 * pa::GLMH_TILE + (size_t)nst_total * 64 * sizeof(float) +
           pa::GLMH_TRAILER;
  if (format != PA_GLM_PLANES_BF16X3) return 0;
  return (size_t)nst_total * 2 * pa::GLMP_TILE + (size_t)nst_total * 64 * sizeof(float);
}

int pa_glm_pack_planes_grouped(int format, const float* X, const float* y, int64_t N, int64_t D,
                               const int64_t* seg, const int64_t* st_off, int64_t nseg,
                               int64_t nst_total, void* planes, size_t planes_bytes,
                               pa_stream_t stream) {
  return pa_glm_pack_planes_grouped_rows(format, X, y, nullptr, N, D, seg, st_off, nseg, nst_total,
                                         planes, planes_bytes, stream);
}

int pa_glm_pack_planes_grouped_rows(int format, const float* X, const float* y, const int64_t* row_of,
                                    int64_t N, int64_t D, const int64_t* seg, const int64_t* st_off,
                                    int64_t nseg, int64_t nst_total, void* planes, size_t planes_bytes,
                                    pa_stream_t stream) {
  PA_REQUIRE(N >= 0 && D >= 1 && D <= 32 && nseg >= 0 && nst_total >= 0,
             "glm_pack_planes_grouped: bad shape N=%lld D=%lld nseg=%lld", (long long)N, (long long)D,
             (long long)nseg);
  PA_REQUIRE_FORMAT(format, "glm_pack_planes_grouped");
  PA_REQUIRE(nseg < (1 << 30), "glm_pack_planes_grouped: too many segments");
  if (nst_total == 0 || nseg == 0) return PA_OK;
  PA_REQUIRE(X && y && seg && st_off && planes, "glm_pack_planes_grouped: NULL pointer");
  PA_REQUIRE(planes_bytes >= pa_glm_grouped_planes_bytes(format, nst_total, D),
             "glm_pack_planes_grouped: image buffer too small");
  PA_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "glm_pack_planes_grouped: unaligned image");
  unsigned char* img = (unsigned char*)planes;
  const int64_t ntiles = nst_total * 2;
  if (format == PA_GLM_PLANES_F16X2) {
    float* y16 = (float*)(img + pa::glmh_tile_bytes(ntiles));
    uint32_t* trailer = (uint32_t*)(y16 + nst_total * 64);
    hipStream_t s = pa::as_stream(stream);
    const int rc = pa::glmh_absmax(X, N, (int)D, trailer, s);
    if (rc != PA_OK) return rc;
    hipLaunchKernelGGL(pa::glm_pack_planes_f16_grouped_kernel,
                       dim3((unsigned)((ntiles * 128 + 255) / 256)), dim3(256), 0, s, X, y, row_of, (int)D,
                       seg, st_off, (int)nseg, ntiles, img, y16, trailer);
    return pa::check_launch("glm_pack_planes_f16_grouped_kernel");
  }
  float* y_img = (float*)(img + (size_t)nst_total * 2 * pa::GLMP_TILE);
  hipLaunchKernelGGL(pa::glm_pack_planes_grouped_kernel, dim3((unsigned)((ntiles * 128 + 255) / 256)),
                     dim3(256), 0, pa::as_stream(stream), X, y, row_of, (int)D, seg, st_off, (int)nseg,
                     ntiles, img, y_img);
  return pa::check_launch("glm_pack_planes_grouped_kernel");
}

size_t pa_glm_bernoulli_grouped_planes_workspace(int64_t nseg, int64_t P) {
  if (nseg < 0 || P < 1) return 0;
  const size_t npass = (size_t)((P + 63) / 64);
  return (size_t)(nseg < 1 ? 1 : nseg) * npass * (2 * 1024 + 2 * 2 * 32) * sizeof(float);
}

int pa_glm_bernoulli_grouped_planes_fwd_bwd(int format, const void* planes, const float* w, const float* b,
                                            double scale, int64_t N, int64_t D, int64_t P, int64_t G,
                                            const int64_t* seg, const int64_t* st_off, int64_t nseg,
                                            const int64_t* group_seg_off, int64_t nst_total,
                                            float* ll, float* gw, float* gb, void* workspace,
                                            size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(N >= 0 && D >= 1 && P >= 1 && G >= 1 && nseg >= 0 && nst_total >= 0,
             "glm_grouped_planes: bad shape N=%lld D=%lld P=%lld G=%lld nseg=%lld", (long long)N,
             (long long)D, (long long)P, (long long)G, (long long)nseg);
  if (D > 32)
    return pa::fail(PA_ERR_UNSUPPORTED, "glm_grouped_planes: the plane image holds D <= 32 (got %lld)",
                    (long long)D);
  PA_REQUIRE_FORMAT(format, "glm_grouped_planes");
  PA_REQUIRE(nseg < (1 << 30) && P < (1 << 20) && G < (1 << 24), "glm_grouped_planes: shape too large");
  PA_REQUIRE(w && ll && gw && gb, "glm_grouped_planes: NULL parameter/output pointer");
  hipStream_t s = pa::as_stream(stream);
  if (N == 0 || nseg == 0 || nst_total == 0) {
    hipError_t e1 = hipMemsetAsync(ll, 0, (size_t)P * 4, s);
    hipError_t e2 = hipMemsetAsync(gw, 0, (size_t)P * G * D * 4, s);
    hipError_t e3 = hipMemsetAsync(gb, 0, (size_t)P * 4, s);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
      return pa::fail(PA_ERR_LAUNCH, "glm_grouped_planes: memset failed");
    return PA_OK;
  }
  PA_REQUIRE(planes && seg && st_off && group_seg_off, "glm_grouped_planes: NULL data pointer");
  PA_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "glm_grouped_planes: unaligned image");
  PA_REQUIRE(workspace && workspace_bytes >= pa_glm_bernoulli_grouped_planes_workspace(nseg, P),
             "glm_grouped_planes: workspace too small");
  const unsigned char* img = (const unsigned char*)planes;
  float* part = (float*)workspace;
  const int npass = (int)((P + 63) / 64);
  hipEvent_t ev0, ev1;
  const bool br = pa::take_bracket(PA_KERNEL_GLM, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  if (format == PA_GLM_PLANES_F16X2) {
    const float* y16 = (const float*)(img + pa::glmh_tile_bytes(nst_total * 2));
    const uint32_t* trailer = (const uint32_t*)(y16 + nst_total * 64);
    const pa::GlmGroupArgs ga{seg, st_off, (int)G};
    // (config 5, 1e7 rows: 0.667 / 0.668 / 0.657 ms per step at 2 / 3 / 4 workgroups per CU)
    const int bpc = pa::g_planes_bpc > 0 ? pa::g_planes_bpc : 4;
    if (bpc >= 4)
      pa::glmh_launch_grouped<4>((int)nseg, npass, img, y16, w, b, N, (int)D, (int)P, nst_total, part,
                                 trailer, ga, s);
    else if (bpc == 3)
      pa::glmh_launch_grouped<3>((int)nseg, npass, img, y16, w, b, N, (int)D, (int)P, nst_total, part,
                                 trailer, ga, s);
    else
      pa::glmh_launch_grouped<2>((int)nseg, npass, img, y16, w, b, N, (int)D, (int)P, nst_total, part,
                                 trailer, ga, s);
  } else {
    const float* y_img = (const float*)(img + (size_t)nst_total * 2 * pa::GLMP_TILE);
    pa::glm_planes_launch_grouped((int)nseg, npass, img, y_img, w, b, N, (int)D, (int)P, nst_total,
                                  part, pa::GlmGroupArgs{seg, st_off, (int)G}, s);
  }
  if (br) (void)hipEventRecord(ev1, s);
  int rc = pa::check_launch("glm_planes_kernel<grouped>");
  if (rc != PA_OK) return rc;
  // the padding rows of every segment's last super-tile added log2(2) = 1 each to the log2(1 + e)
  // sum of every particle: ln2 per row back in
  const double ll_offset = (double)(nst_total * 64 - N) * 0.6931471805599453;
  pa::glm_grouped_finalize_launch<1, 2>(part, group_seg_off, (int)nseg, (int)D, (int)P, (int)G, scale, gw, ll,
                                        gb, ll_offset, s);
  return pa::check_launch("glm_grouped_finalize");
}

// feature tiles of the f16 image: 1 (D <= 32), 2 (<= 64), 4 (<= 128: csrc/glm_planes16d.h)
static int glmd_dt(int64_t D) { return D <= 32 ? 1 : (D <= 64 ? 2 : 4); }

size_t pa_glm_planes_bytes(int format, int64_t N, int64_t D) {
  if (N < 0 || D < 1 || D > 128) return 0;
  if (format == PA_GLM_PLANES_F16X2 && D > 32)
    return (size_t)pa::glmh_tile_bytes(pa::glm_planes_tiles(N)) * glmd_dt(D) + pa::GLMD_TRAILER;
  if (D > 32) return 0;
  if (format == PA_GLM_PLANES_F16X2)
    return (size_t)pa::glmh_tile_bytes(pa::glm_planes_tiles(N)) + pa::GLMH_TRAILER;
  if (format != PA_GLM_PLANES_BF16X3) return 0;
  return (size_t)pa::glm_planes_tiles(N) * pa::GLMP_TILE;
}

int pa_glm_pack_planes(int format, const float* X, int64_t N, int64_t D, void* planes,
                       size_t planes_bytes, pa_stream_t stream) {
  {
    // the arrival counters of the in-kernel finalize: packing never happens inside a capture
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone)
      (void)pa::glmf_counters(true);
  }
  PA_REQUIRE_FORMAT(format, "glm_pack_planes");
  PA_REQUIRE(N >= 0 && D >= 1, "glm_pack_planes: bad shape N=%lld D=%lld", (long long)N, (long long)D);
  if (D > 128 || (D > 32 && format != PA_GLM_PLANES_F16X2))
    return pa::fail(PA_ERR_UNSUPPORTED, "glm_pack_planes: the plane image holds D <= 32 (f16 format: D <= 128), got %lld",
                    (long long)D);
  PA_REQUIRE(N < (int64_t(1) << 40), "glm_pack_planes: shape too large");
  PA_REQUIRE(planes && planes_bytes >= pa_glm_planes_bytes(format, N, D),
             "glm_pack_planes: image too small");
  PA_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "glm_pack_planes: unaligned image");
  PA_REQUIRE(N == 0 || X, "glm_pack_planes: NULL data pointer");
  const int64_t nt = pa::glm_planes_tiles(N);
  if (format == PA_GLM_PLANES_F16X2 && D > 32) {
    // DT feature tiles per row tile, a 1-KiB trailer of column maxima / exponents (glm_planes16d.h)
    hipStream_t s = pa::as_stream(stream);
    const int DT = glmd_dt(D);
    uint32_t* trailer = (uint32_t*)((unsigned char*)planes + pa::glmh_tile_bytes(nt) * DT);
    if (hipMemsetAsync(trailer, 0, pa::GLMD_TRAILER, s) != hipSuccess)
      return pa::fail(PA_ERR_LAUNCH, "glm_pack_planes: memset of the image trailer failed");
    if (N > 0) {
      int64_t grid = (N + 63) / 64;
      const int64_t cap = (int64_t)pa::cu_count() * 8;
      if (grid > cap) grid = cap;
      hipLaunchKernelGGL(pa::glmd_absmax_kernel, dim3((unsigned)grid), dim3(256), 0, s, X, N, (int)D, trailer);
      const int rc = pa::check_launch("glmd_absmax_kernel");
      if (rc != PA_OK) return rc;
    }
    const unsigned nblk = (unsigned)((nt * DT * 128 + 255) / 256 + (nt == 0));
    if (DT == 2)
      hipLaunchKernelGGL(pa::glmd_pack_kernel<2>, dim3(nblk), dim3(256), 0, s, X, N, (int)D, nt,
                         (unsigned char*)planes, trailer);
    else
      hipLaunchKernelGGL(pa::glmd_pack_kernel<4>, dim3(nblk), dim3(256), 0, s, X, N, (int)D, nt,
                         (unsigned char*)planes, trailer);
    return pa::check_launch("glmd_pack_kernel");
  }
  if (format == PA_GLM_PLANES_F16X2) {
    hipStream_t s = pa::as_stream(stream);
    uint32_t* trailer = (uint32_t*)((unsigned char*)planes + pa::glmh_tile_bytes(nt));
    const int rc = pa::glmh_absmax(X, N, (int)D, trailer, s);
    if (rc != PA_OK) return rc;
    // (also with no rows: the kernel's thread 0 writes the exponent into the trailer)
    hipLaunchKernelGGL(pa::glm_pack_planes_f16_kernel, dim3((unsigned)((nt * 128 + 255) / 256 + (nt == 0))),
                       dim3(256), 0, s, X, N, (int)D, nt, (unsigned char*)planes, trailer);
    return pa::check_launch("glm_pack_planes_f16_kernel");
  }
  if (nt == 0) return PA_OK;
  hipLaunchKernelGGL(pa::glm_pack_planes_kernel, dim3((unsigned)((nt * 128 + 255) / 256)), dim3(256),
                     0, pa::as_stream(stream), X, N, (int)D, nt, (unsigned char*)planes);
  return pa::check_launch("glm_pack_planes_kernel");
}

int pa_glm_planes_stamps(void* two_u64) {
  pa::g_planes_stamps = (unsigned long long*)two_u64;
  return PA_OK;
}

int pa_glm_planes_finalize_mode(int in_kernel) {
  PA_REQUIRE(in_kernel == 0 || in_kernel == 1, "glm_planes_finalize_mode: 0 or 1");
  pa::g_planes_fin_mode = in_kernel;
  return PA_OK;
}

int pa_glm_planes_tune(int ring_depth, int blocks_per_cu) {
  // (f16 image: 3 / 4 = ring depth of the 32 x 32-tile kernel; measurement knobs: 5 / 6 the same with
  //  per-wave private rings of depth 3 / 4, 9 / 10 one wave per tile and 64 particles, ring depth 3 / 4)
  //  11 = the default ring with the 2 x 2 wave geometry whatever P: no 128 / 256-particle passes)
  //  12 = at most 128 particles per pass: 2 x 4 waves for every P > 64; 13 = the wide geometries at every N -- by
  //  default only from ~48 32-row tiles per workgroup on, N >= 393 k on 256 CUs)
  PA_REQUIRE(ring_depth == 0 || (ring_depth >= 3 && ring_depth <= 13),
             "glm_planes_tune: ring depth code 3..13 (0 = default)");
  PA_REQUIRE(blocks_per_cu >= 0 && blocks_per_cu <= 4, "glm_planes_tune: 0..4 workgroups per CU");
  pa::g_planes_wide = ring_depth != 11;
  pa::g_planes_wide_max = ring_depth == 12 ? 4 : 8;
  pa::g_planes_wide_force = ring_depth == 12 || ring_depth == 13;
  pa::g_planes_nb = (ring_depth == 0 || ring_depth >= 11) ? 3 : ring_depth;
  pa::g_planes_bpc = blocks_per_cu;
  return PA_OK;
}

size_t pa_glm_bernoulli_planes_workspace(int64_t N, int64_t D, int64_t P) {
  if (N < 0 || D < 1 || D > 128 || P < 1) return 0;
  if (D > 32) {
    // glm_planes16d.h: one record of DT feature tiles x 2 particle tiles per workgroup and pass
    const size_t npass = (size_t)((P + 63) / 64);
    const size_t cap = (size_t)pa::cu_count() * 2;
    const size_t nst = (size_t)(((N + 31) / 32 + 1) / 2);
    const size_t nb = nst < cap ? (nst < 1 ? 1 : nst) : cap;
    return nb * npass * (2 * (size_t)glmd_dt(D) * 1024 + 2 * 2 * 32) * sizeof(float);
  }
  // (the same for both image formats: the record count is capped at four workgroups per CU)
  const pa::GlmPlanesPlan pl = pa::glm_planes_plan(N, P);
  // records of the deepest / widest tuning so that the knob never invalidates a workspace
  const size_t cap = (size_t)pa::cu_count() * 4;
  const size_t nb = (size_t)pl.nst < cap ? (size_t)(pl.nst < 1 ? 1 : pl.nst) : cap;
  const size_t rec = 2 * 1024 + 2 * 2 * 32;
  // the partial records, then the fp64 level-1 partials of the in-kernel finalize
  return nb * pl.npass * rec * sizeof(float) +
         (size_t)pl.npass * pa::GLMF_GROUPS * rec * sizeof(double);
}

size_t pa_glm_label_moments_workspace(int64_t N) {
  if (N < 0) return 0;
  int64_t grid = (N + 63) / 64;
  const int64_t cap = (int64_t)pa::cu_count() * 4;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  return (size_t)grid * 33 * sizeof(double);
}

int pa_glm_label_moments(const float* X, const float* y, int64_t N, int64_t D, double* moments,
                         void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  PA_REQUIRE(N >= 0 && D >= 1 && D <= 32, "glm_label_moments: bad shape N=%lld D=%lld", (long long)N,
             (long long)D);
  PA_REQUIRE(moments != nullptr && (N == 0 || (X && y)), "glm_label_moments: NULL pointer");
  PA_REQUIRE(workspace && workspace_bytes >= pa_glm_label_moments_workspace(N),
             "glm_label_moments: workspace too small");
  const int nblocks = (int)(pa_glm_label_moments_workspace(N) / (33 * sizeof(double)));
  hipStream_t s = pa::as_stream(stream);
  hipLaunchKernelGGL(pa::glm_label_moments_partial_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, X, y,
                     N, (int)D, (double*)workspace);
  int rc = pa::check_launch("glm_label_moments_partial_kernel");
  if (rc != PA_OK) return rc;
  hipLaunchKernelGGL(pa::glm_label_moments_final_kernel, dim3(1), dim3(64), 0, s,
                     (const double*)workspace, nblocks, moments);
  return pa::check_launch("glm_label_moments_final_kernel");
}

int pa_glm_bernoulli_planes_fwd_bwd(int format, const void* planes, const float* y, const float* w,
                                    const float* b, double scale, int64_t N, int64_t D, int64_t P,
                                    float* ll, float* gw, float* gb, void* workspace,
                                    size_t workspace_bytes, const double* moments,
                                    pa_stream_t stream) {
  PA_REQUIRE(N >= 0 && D >= 1 && P >= 1, "glm_planes: bad shape N=%lld D=%lld P=%lld", (long long)N,
             (long long)D, (long long)P);
  if (D > 128 || (D > 32 && format != PA_GLM_PLANES_F16X2))
    return pa::fail(PA_ERR_UNSUPPORTED, "glm_planes: the plane image holds D <= 32 (f16 format: D <= 128), got %lld",
                    (long long)D);
  PA_REQUIRE_FORMAT(format, "glm_planes");
  PA_REQUIRE(N < (int64_t(1) << 40) && P < (1 << 20), "glm_planes: shape too large");
  PA_REQUIRE(w && ll && gw && gb, "glm_planes: NULL parameter/output pointer");
  PA_REQUIRE(N == 0 || (planes && y), "glm_planes: NULL data pointer");
  // a guide draw parked in front of this launch (chain.h): the default f16 kernel draws w and b itself
  pa::GlmDraw draw;
  bool drawn = false;
  if (format == PA_GLM_PLANES_F16X2 && N > 0 && D <= 32) {
    const pa::GlmPlanesPlan p0 = pa::glmh_plan(N, P);
    if (p0.nb == 3 && p0.bpc < 4) drawn = pa::glm_take_pending_draw(stream, w, b, P, D, &draw);
  }
  hipStream_t s = pa::as_stream(stream);
  if (N == 0) {
    hipError_t e1 = hipMemsetAsync(ll, 0, (size_t)P * 4, s);
    hipError_t e2 = hipMemsetAsync(gw, 0, (size_t)P * D * 4, s);
    hipError_t e3 = hipMemsetAsync(gb, 0, (size_t)P * 4, s);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
      return pa::fail(PA_ERR_LAUNCH, "glm_planes: memset failed");
    return PA_OK;
  }
  PA_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "glm_planes: unaligned image");
  PA_REQUIRE(workspace && workspace_bytes >= pa_glm_bernoulli_planes_workspace(N, D, P),
             "glm_planes: workspace too small");
  if (D > 32) return pa::glmd_run(planes, y, w, b, scale, N, (int)D, (int)P, ll, gw, gb, (float*)workspace, stream, s);
  pa::GlmPlanesPlan pl =
      format == PA_GLM_PLANES_F16X2 ? pa::glmh_plan(N, P, !drawn) : pa::glm_planes_plan(N, P);
  // the f16 image: tuning codes 9 / 10 run one wave per 32-row tile and 64 particles (glm_planes16w.h,
  // ring depth 3 / 4) -- measured equal to the default kernel (profiles/r04_glm16_ablation.txt): both
  // are bound by the same element-wise VALU stream
  const bool wide = format == PA_GLM_PLANES_F16X2 && (pl.nb == 9 || pl.nb == 10) && pl.bpc <= 2;
  float* part = (float*)workspace;
  const unsigned char* img = (const unsigned char*)planes;
  hipEvent_t ev0, ev1;
  const bool br = pa::take_bracket(PA_KERNEL_GLM, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  // every padding row of the processed super-tiles added log2(2) = 1 to the log2(1 + e) sum of
  // every particle (glm_planes.h): ln2 per row back in
  const double ll_offset = (double)((wide ? (N + 127) / 128 * 128 : pl.nst * 32 * pl.nrt) - N) * 0.6931471805599453;
  pa::GlmFinArgs fin;
  fin.counters = nullptr;
  if (format == PA_GLM_PLANES_BF16X3 && pa::g_planes_fin_mode == 1 && pl.npass <= pa::GLMF_MAX_PASSES) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    fin.counters = pa::glmf_counters(!capturing);
  }
  const size_t rec_bytes = (size_t)pl.nblocks * pl.npass * (2 * 1024 + 2 * 2 * 32) * sizeof(float);
  fin.part64 = (double*)((char*)workspace + ((rec_bytes + 15) & ~(size_t)15));
  fin.ll = ll; fin.gw = gw; fin.gb = gb;
  fin.scale = scale; fin.ll_offset = ll_offset;
  fin.D = (int)D; fin.P = (int)P;
  fin.tstamps = pa::g_planes_stamps;
  if (format == PA_GLM_PLANES_F16X2) {
    const uint32_t* trailer = (const uint32_t*)(img + pa::glmh_tile_bytes(pa::glm_planes_tiles(N)));
    if (pl.npt == 4 && moments != nullptr) pa::glmh_launch_wide<2, 4, true>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s, moments);
    else if (pl.npt == 4) pa::glmh_launch_wide<2, 4, false>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s, nullptr);
    else if (pl.npt == 8 && moments != nullptr) pa::glmh_launch_wide<1, 8, true>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s, moments);
    else if (pl.npt == 8) pa::glmh_launch_wide<1, 8, false>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s, nullptr);
    else if (wide && pl.nb == 9) pa::glmw_launch<3>(pl.bpc, img, y, w, b, N, (int)D, (int)P, part, trailer, &pl.nblocks, s);
    else if (wide) pa::glmw_launch<4>(pl.bpc, img, y, w, b, N, (int)D, (int)P, part, trailer, &pl.nblocks, s);
    else if (pl.nb == 5) pa::glmh_launch_one<3, 3, true>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s);
    else if (pl.nb == 6) pa::glmh_launch_one<4, 3, true>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s);
    else if (pl.nb == 4) pa::glmh_launch_one<4, 3>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s);
    else if (pl.bpc >= 4) pa::glmh_launch_one<3, 4>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s);
    else if (moments != nullptr && drawn)
      pa::glmh_launch_one<3, 3, false, true, true>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s, moments, draw);
    else if (moments != nullptr)
      pa::glmh_launch_one<3, 3, false, true>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s, moments);
    else if (drawn)
      pa::glmh_launch_one<3, 3, false, false, true>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s, nullptr, draw);
    else pa::glmh_launch_one<3, 3>(pl, img, y, w, b, N, (int)D, (int)P, part, trailer, s);
  } else if (pl.nb == 3) {
    pa::glm_planes_launch_one<3, 3>(pl, img, y, w, b, N, (int)D, (int)P, part, fin, s);
  } else {
    pa::glm_planes_launch_one<4, 2>(pl, img, y, w, b, N, (int)D, (int)P, part, fin, s);
  }
  if (br) (void)hipEventRecord(ev1, s);
  int rc = pa::check_launch("glm_planes_kernel");
  if (rc != PA_OK) return rc;
  if (fin.counters != nullptr) return PA_OK;        // ll / gw / gb were written by the kernel itself
  const int64_t J = (int64_t)P * D + 2 * P;
  rc = pa::chain_record_fin(stream, 1, 2, part, pl.nblocks, pl.npass, (int)D, (int)P, scale, ll, gw,
                            gb, ll_offset);
  if (rc != 0) return rc < 0 ? rc : PA_OK;       // recorded as a phase of the step's chained tail
  hipLaunchKernelGGL((pa::glm_finalize_kernel<1, 2>),
                     dim3((unsigned)((J + pa::FIN_OUT - 1) / pa::FIN_OUT)),
                     dim3(pa::FIN_OUT * pa::FIN_GROUPS), 0, s, part, pl.nblocks, pl.npass, (int)D,
                     (int)P, scale, ll, gw, gb, ll_offset);
  return pa::check_launch("glm_finalize_kernel");
}

}  // extern "C"
