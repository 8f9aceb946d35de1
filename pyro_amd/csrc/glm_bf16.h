// glm_bf16.h -- the fused Bernoulli-logits GLM pass on the bf16 matrix cores with a 3-way
// error-compensated operand split (included by glm.hip; shares its record format / finalize).
//
// Why: exact-f32 MFMA (v_mfma_f32_32x32x2_f32) runs at the f32 VECTOR rate and does not overlap
// VALU work on gfx950 (tools/probes/mfma_valu_overlap.hip), so the exact kernel of glm.hip is
// bound by MFMA + VALU issue.  v_mfma_f32_32x32x16_bf16 is 16x faster and DOES overlap the VALU.
// Every f32 operand is split exactly into three bf16 pieces, x = x1 + x2 + x3 (8+8+8 significand
// bits, round-to-nearest at each step, residuals computed exactly in f32), and a product is
// evaluated as the six piece products of order >= 2^-16:
//     x*w ~= x3*w1 + x2*w2 + x1*w3 + x2*w1 + x1*w2 + x1*w1          (dropped: O(2^-24 |x||w|))
// accumulated in the f32 MFMA accumulator -- f32-roundoff-class error, measured against the f64
// oracle in tests/test_kernels_gpu.py next to the exact-f32 kernel.
//
// Per wave and 32-row tile of X (D <= 32*DT features, 32*PT particles):
//   * the tile arrives as coalesced float4 global loads one tile ahead (registers), is split and
//     written ONCE to the wave's private LDS slice as three row-major bf16 planes [32][DP];
//   * GEMM1  L[n,p] = sum_d X[n,d] W[p,d]: A = X planes (ds_read_b128 rows), B = W planes (staged
//     once per block in LDS).  One extra "aux" MFMA adds the bias (three bf16 pieces against a
//     column of ones) and, for masked / out-of-range rows, a -1e30 logit offset, so that such a
//     row contributes exactly 0 to every output without any mask arithmetic on the VALU;
//   * element-wise on the accumulator registers (VALU, 13 instructions incl. v_exp/v_log/v_rcp):
//         ll += y*l - softplus(l),   g = y - sigmoid(l);
//   * g is split into three bf16 planes IN REGISTERS: the C/D layout of the 32x32 MFMA
//     (lane = p, registers = rows n) is the A-operand layout of GEMM2  gw[p,d] += sum_n g[p,n]
//     X[n,d] with the K index permuted consistently on both operands; B = X planes read
//     column-wise from the same LDS slice (16-bit reads, conflict-free);
//   * nothing of size P*N reaches HBM; X and y are read once.
#pragma once
#include "common.h"

namespace pa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8v __attribute__((ext_vector_type(8)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16* lds_v4i16_ptr;

constexpr int GLMB_WAVES = 4;

template <int DT, int PT>
struct GlmBfCfg {
  static constexpr int DP = 32 * DT;            // padded feature count
  static constexpr int KC = 2 * DT;             // 16-wide K chunks of GEMM1
  static constexpr int RS = DP * 2 + 16;        // plane row stride in bytes (conflict-free b128 rows)
  static constexpr int PLANE = 32 * RS;         // bytes of one 32-row plane
  static constexpr int WROWS = 32 * PT;
  static constexpr int W_BYTES = 3 * WROWS * RS;
  static constexpr int WAUX_BYTES = WROWS * 8;
  static constexpr int WAVE_BYTES = 3 * PLANE + 384;   // + yh[32] f32 + aux[64] u32 (upper half 0)
  static constexpr int LDS_BYTES = W_BYTES + WAUX_BYTES + GLMB_WAVES * WAVE_BYTES;
};

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  f32x2v v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32
}
__device__ __forceinline__ float bf16_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t p) {
  return __builtin_bit_cast(float, p & 0xffff0000u);
}

// (a, b) -> three packed bf16 pairs with a = a1+a2+a3, b = b1+b2+b3 (exact for finite inputs whose
// leading piece does not overflow bf16).  Written on 2-vectors so that the residuals are one
// v_pk_add_f32 each.
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& p1, uint32_t& p2,
                                           uint32_t& p3) {
  const f32x2v v = {a, b};
  p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32
  const f32x2v f1 = {bf16_lo(p1), bf16_hi(p1)};
  const f32x2v r = v - f1;
  p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
  const f32x2v f2 = {bf16_lo(p2), bf16_hi(p2)};
  const f32x2v q = r - f2;
  p3 = __builtin_bit_cast(uint32_t, __builtin_convertvector(q, bf16x2));
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4v v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}

// bf16 pattern of -1e30 (top half of the f32 encoding 0xF149F2CA): the logit offset of rows that
// must not contribute (masked, or beyond the end of the plate)
constexpr uint32_t BF16_NEG_HUGE = 0xF149u;
constexpr uint32_t BF16_ONE = 0x3F80u;

// MASKED: a row mask is given.  A masked row must contribute exactly nothing whatever it holds
// (scale_and_mask is where(mask, x, 0), pyro/distributions/util.py:326): besides the -1e30 logit
// offset (which silences ll and g only while |x . w| stays far below 1e30) its X values are
// replaced by 0 before the split, so that huge finite garbage under the mask cannot reach the
// logits or the gradient products.  Separate instantiation: the unmasked loop carries no selects.
template <int DT, int PT, bool GROUPED, bool MASKED>
__global__ __launch_bounds__(64 * GLMB_WAVES, 2) void glm_bernoulli_bf16_kernel(
    const float* __restrict__ X, const float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ b, const uint8_t* __restrict__ mask, int64_t N, int D, int P,
    int64_t iters, float* __restrict__ part, const int64_t* __restrict__ seg, int G) {
  using C = GlmBfCfg<DT, PT>;
  constexpr int DP = C::DP, KC = C::KC, RS = C::RS, PLANE = C::PLANE, WROWS = C::WROWS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar
  const int l31 = lane & 31, h = lane >> 5;
  unsigned char* Wp = smem;
  uint32_t* waux = reinterpret_cast<uint32_t*>(smem + C::W_BYTES);
  unsigned char* Xp = smem + C::W_BYTES + C::WAUX_BYTES + wave * C::WAVE_BYTES;
  float* yh_s = reinterpret_cast<float*>(Xp + 3 * PLANE);
  uint32_t* aux_s = reinterpret_cast<uint32_t*>(Xp + 3 * PLANE + 128);

  for (int i = threadIdx.x; i < C::LDS_BYTES / 4; i += 64 * GLMB_WAVES)
    reinterpret_cast<uint32_t*>(smem)[i] = 0u;

  int64_t row_begin = 0, row_end = N;
  int group = 0;
  if constexpr (GROUPED) {
    row_begin = seg[3 * (int64_t)blockIdx.x];
    row_end = seg[3 * (int64_t)blockIdx.x + 1];
    group = (int)seg[3 * (int64_t)blockIdx.x + 2];
  }
  const int pbase = blockIdx.y * WROWS;

  // ---- staging: tile k+2 travels global -> VGPR (raw) while tile k computes; tile k+1 is split
  //      into its bf16 planes in registers under the MFMAs of GEMM1 and written to LDS at the end
  //      of tile k (after the last LDS read of tile k's planes) --------------------------------
  constexpr int NLD = 4 * DT;                 // float4 loads per lane per tile
  constexpr bool EARLY_SPLIT = (DT == 1);     // larger D: no registers for the split planes
  float4 stage[NLD];
  uint8_t stage_m[NLD];                       // MASKED: the mask byte of each staged float4's row
  float st_y = 0.0f;
  uint8_t st_m = 0;
  uint32_t xs1[2 * NLD], xs2[2 * NLD], xs3[2 * NLD];
  float xs_yh = -0.5f;
  uint32_t xs_aux = 0u;
  const int D4 = D >> 2;                      // float4 per row
  const int64_t total_e = row_end * (int64_t)D;

  // Loads use CLAMPED addresses and are consumed raw, a whole tile of compute later (a select next
  // to the load would make the compiler wait for the load right there and expose the HBM latency
  // on every tile).  All per-tile address arithmetic is scalar (the tile index is wave-uniform):
  // per load one v_min_u32 of the lane's constant offset against the tile's last valid offset.
  // Rows past the end of the plate / segment (and whole tiles past it) read in-range data of other
  // rows and are switched off through the -1e30 row offset: their X values never matter (a
  // non-finite X entry anywhere makes the result non-finite either way).  No branches in the loop.
  uint32_t lofs[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) lofs[j] = (j * 64 + lane < 8 * D) ? 4u * (uint32_t)(j * 64 + lane) : 0u;
  const int64_t all_e = N * (int64_t)D;
  auto issue_loads = [&](int64_t tile) {
    const int64_t base = (row_begin + tile * 32) * (int64_t)D;
    const bool tv = base < total_e;
    const int64_t sb = tv ? base : 0;
    const int64_t rem = (tv ? total_e : all_e) - sb;                 // >= D >= 4
    const uint32_t lim = (uint32_t)(rem < 32 * (int64_t)D ? rem : 32 * (int64_t)D) - 4u;
    const float* Xb = X + sb;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const uint32_t off = lofs[j] < lim ? lofs[j] : lim;
      stage[j] = *reinterpret_cast<const float4*>(Xb + off);
      if constexpr (MASKED) stage_m[j] = mask[sb / D + off / (uint32_t)D];   // row of this float4
    }
    const int64_t rb = tv ? row_begin + tile * 32 : 0;
    const int64_t rrem = (tv ? row_end : N) - rb;                    // >= 1
    const uint32_t rlim = (uint32_t)(rrem < 32 ? rrem : 32) - 1u;
    const uint32_t ro = (uint32_t)l31 < rlim ? (uint32_t)l31 : rlim;
    st_y = y[rb + ro];
    // always a load (no branch in the loop body): without a mask the byte read is ignored
    st_m = *(mask != nullptr ? mask + rb + ro : reinterpret_cast<const uint8_t*>(y + rb + ro));
  };
  auto split_unit = [&](int j, int64_t tile) {          // stage[j] -> xs*[2j], xs*[2j+1]
#ifdef PA_GLM_PROBE_NOSPLITX
    xs1[2 * j] = cvt_pk_bf16(stage[j].x, stage[j].y); xs1[2 * j + 1] = cvt_pk_bf16(stage[j].z, stage[j].w);
    xs2[2 * j] = xs1[2 * j]; xs2[2 * j + 1] = xs1[2 * j + 1];
    xs3[2 * j] = xs1[2 * j]; xs3[2 * j + 1] = xs1[2 * j + 1];
    return;
#endif
    if constexpr (MASKED) {
      if (stage_m[j] == 0) stage[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    split_pair(stage[j].x, stage[j].y, xs1[2 * j], xs2[2 * j], xs3[2 * j]);
    split_pair(stage[j].z, stage[j].w, xs1[2 * j + 1], xs2[2 * j + 1], xs3[2 * j + 1]);
  };
  auto split_row = [&](int64_t tile) {
    const int64_t rows_left = row_end - (row_begin + tile * 32);     // scalar; <= 0: tile past the end
    const bool okr = (int64_t)l31 < rows_left && (mask == nullptr || st_m != 0);
    // scale_and_mask is where(mask, x, 0) (pyro/distributions/util.py:326): a row that does not
    // count gets y = 0 and the -1e30 logit offset
    xs_yh = (okr ? st_y : 0.0f) - 0.5f;
    xs_aux = (BF16_ONE << 16) | (okr ? 0u : BF16_NEG_HUGE);     // k slots {offset, 1.0}
  };
  // (row, col) of this lane's j-th float4; lanes past the end of a narrow tile (D < DP) write into
  // the 16-byte pad of row 0, which nobody reads
  int wofs[NLD];
  {
    int n = lane / D4, d = (lane % D4) * 4;
    const int qn = 64 / D4, qd = (64 % D4) * 4;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      wofs[j] = (j * 64 + lane < 8 * D) ? n * RS + d * 2 : DP * 2;
      n += qn;
      d += qd;
      if (d >= D) { d -= D; n += 1; }
    }
  }
  auto write_xs = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      unsigned char* q = Xp + wofs[j];
      *reinterpret_cast<uint2*>(q) = make_uint2(xs1[2 * j], xs1[2 * j + 1]);
      *reinterpret_cast<uint2*>(q + PLANE) = make_uint2(xs2[2 * j], xs2[2 * j + 1]);
      *reinterpret_cast<uint2*>(q + 2 * PLANE) = make_uint2(xs3[2 * j], xs3[2 * j + 1]);
    }
    yh_s[l31] = xs_yh;          // both lane halves hold the same row values
    aux_s[l31] = xs_aux;
  };

  int64_t tile = GROUPED ? (int64_t)wave : (int64_t)blockIdx.x * GLMB_WAVES + wave;
  const int64_t tile_stride = GROUPED ? (int64_t)GLMB_WAVES : (int64_t)gridDim.x * GLMB_WAVES;

  issue_loads(tile);
  __syncthreads();  // LDS zero-fill complete

  // ---- W planes and the bias pieces: once per block --------------------------------------
  for (int idx = threadIdx.x; idx < WROWS * (DP / 2); idx += 64 * GLMB_WAVES) {
    const int pl = idx / (DP / 2), d = (idx % (DP / 2)) * 2;
    const int p = pbase + pl;
    const int64_t row = GROUPED ? ((int64_t)p * G + group) : (int64_t)p;
    const float w0 = (p < P && d < D) ? w[row * D + d] : 0.0f;
    const float w1 = (p < P && d + 1 < D) ? w[row * D + d + 1] : 0.0f;
    uint32_t p1, p2, p3;
    split_pair(w0, w1, p1, p2, p3);
    unsigned char* q = Wp + pl * RS + d * 2;
    *reinterpret_cast<uint32_t*>(q) = p1;
    *reinterpret_cast<uint32_t*>(q + WROWS * RS) = p2;
    *reinterpret_cast<uint32_t*>(q + 2 * WROWS * RS) = p3;
  }
  for (int pl = threadIdx.x; pl < WROWS; pl += 64 * GLMB_WAVES) {
    const int p = pbase + pl;
    const float bv = (p < P && b != nullptr) ? b[p] : 0.0f;
    uint32_t p1, p2, p3;
    split_pair(bv, 0.0f, p1, p2, p3);
    waux[2 * pl] = BF16_ONE | (p1 << 16);                    // k slots {0: 1.0, 1: b1}
    waux[2 * pl + 1] = (p2 & 0xffffu) | (p3 << 16);          // k slots {2: b2, 3: b3}
  }
#pragma unroll
  for (int j = 0; j < NLD; ++j) split_unit(j, tile);
  split_row(tile);
  write_xs();
  issue_loads(tile + tile_stride);
  __syncthreads();

  bf16x8 b_aux[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
    b_aux[pt] = as_bf16x8(h == 0 ? waux[2 * (pt * 32 + l31)] : 0u,
                          h == 0 ? waux[2 * (pt * 32 + l31) + 1] : 0u, 0u, 0u);

  f32x16v gwacc[PT][DT];
  f32x2v ll2[PT], gb2[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    ll2[pt] = f32x2v{0.0f, 0.0f};
    gb2[pt] = f32x2v{0.0f, 0.0f};
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) gwacc[pt][dt][r] = 0.0f;
  }

  const unsigned char* a_row = Xp + l31 * RS + 16 * h;            // A operand of GEMM1
  const unsigned char* w_row = Wp + l31 * RS + 16 * h;            // B operand of GEMM1
  // B operand of GEMM2 (transpose read): this lane's row (4h + (q>>2)) and column group of the
  // 16-lane group it belongs to, q = lane & 15
  const uint32_t x_tr_lds = (uint32_t)(uintptr_t)(Xp + (4 * h + ((lane & 15) >> 2)) * RS +
                                                  (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const uint32_t aux_ones = h == 0 ? (BF16_ONE | (BF16_ONE << 16)) : 0u;

  // piece products in increasing order of magnitude: (x3,w1) (x2,w2) (x1,w3) (x2,w1) (x1,w2) (x1,w1)
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0};
  constexpr int TB[6] = {0, 1, 2, 0, 1, 0};
  constexpr int NM1 = 1 + 6 * KC;             // MFMAs of GEMM1 per particle tile

  for (int64_t it = 0; it < iters; ++it) {
    const int64_t nxt = tile + tile_stride;
    f32x16v acc[PT];
    bf16x8 xa[3], wa[3], xb[3];
    uint32_t g1[PT][8], g2[PT][8], g3[PT][8];

    // i-th MFMA of GEMM1 for particle tile pt (0: bias / row-offset, then chunk-major pieces)
    auto gemm1 = [&](int pt, int i) {
      if (i == 0) {
        // aux_s[32..63] stay 0: the upper lane half (k slots 8..15) contributes nothing
        const bf16x8 a_aux = as_bf16x8(aux_s[lane], aux_ones, 0u, 0u);
        const f32x16v zero = {};
        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_aux, b_aux[pt], zero, 0, 0, 0);
        return;
      }
      const int c = (i - 1) / 6, t = (i - 1) % 6;
#ifdef PA_GLM_PROBE_NOGEMM1
      if (t == 0) acc[pt][c] += yh_s[lane & 31];
      return;
#endif
      if (t == 0) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          xa[pl] = *reinterpret_cast<const bf16x8*>(a_row + 32 * c + pl * PLANE);
          wa[pl] = *reinterpret_cast<const bf16x8*>(w_row + pt * 32 * RS + 32 * c + pl * WROWS * RS);
        }
      }
      acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[TA[t]], wa[TB[t]], acc[pt], 0, 0, 0);
    };
    // element-wise on accumulator registers 2i, 2i+1 of particle tile pt (rows n = (r&3) + 8(r>>2)
    // + 4h), two at a time so that the plain f32 arithmetic packs into v_pk_*_f32:
    //   y*l - softplus(l) = (y - 1/2) l - |l|/2 - ln2*log2(1 + e),  e = exp(-|l|)
    //   (a row with the -1e30 offset gives (-1/2)(-1e30) - 1e30/2 - 0 = 0 exactly)
    //   g = y - sigmoid(l) = (y - 1/2) - copysign(1/(1+e) - 1/2, l)
    // then g -> three bf16 pieces, packed as the A operand of GEMM2
    auto elem = [&](int pt, int i) {
      const int r = 2 * i;
      const float* yp = yh_s + 8 * (r >> 2) + 4 * h + (r & 3);
      const f32x2v yh = *reinterpret_cast<const f32x2v*>(yp);
      const f32x2v l = {acc[pt][r], acc[pt][r + 1]};
#ifdef PA_GLM_PROBE_NOELEM   // tools/probes/glm_variants: marginal cost of the element-wise math
      ll2[pt] += l;
      gb2[pt] += yh;
      split_pair(l.x, l.y, g1[pt][i], g2[pt][i], g3[pt][i]);
      return;
#endif
      const f32x2v a = {__builtin_fabsf(l.x), __builtin_fabsf(l.y)};
      const f32x2v na = a * -1.44269504088896340736f;
      const f32x2v e = {__builtin_amdgcn_exp2f(na.x), __builtin_amdgcn_exp2f(na.y)};
      const f32x2v t = e + 1.0f;
      const f32x2v lg = {__builtin_amdgcn_logf(t.x), __builtin_amdgcn_logf(t.y)};
      const f32x2v inv = {__builtin_amdgcn_rcpf(t.x), __builtin_amdgcn_rcpf(t.y)};
      f32x2v u = yh * l;
      u = a * -0.5f + u;
      u = lg * -0.69314718055994530942f + u;
      ll2[pt] += u;
      const f32x2v dd = inv - 0.5f;
      const f32x2v ds = {__builtin_copysignf(dd.x, l.x), __builtin_copysignf(dd.y, l.y)};
      const f32x2v g = yh - ds;
      gb2[pt] += g;
#ifdef PA_GLM_PROBE_NOSPLITG
      g1[pt][i] = cvt_pk_bf16(g.x, g.y);
      g2[pt][i] = g1[pt][i];
      g3[pt][i] = g1[pt][i];
      return;
#endif
      split_pair(g.x, g.y, g1[pt][i], g2[pt][i], g3[pt][i]);
    };
    // B operand of GEMM2 for K half kh, feature tile dt: rows n(8kh+j, h), j = 0..7, of column
    // d = 32dt + l31 from each plane -- a column access into the row-major planes, done by the LDS
    // transpose read ds_read_b64_tr_b16: within a 16-lane group lane q passes the address of 4
    // consecutive bf16 of row (q>>2) (columns 4(q&3)..+3 of the group's 16) and lane i receives
    // column i of that 4x16 block (mapping measured by tools/probes/tr16_probe.hip).  Two reads
    // (rows 16kh+4h+{0..3} and +8) make the 8 K-slots of one operand.
    auto load_xb = [&](int kh, int dt) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const uint32_t a0 = x_tr_lds + (uint32_t)((16 * kh) * RS + dt * 64 + pl * PLANE);
        const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)a0);
        const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(a0 + 8 * RS));
        u16x8v c = {(unsigned short)lo[0], (unsigned short)lo[1], (unsigned short)lo[2],
                    (unsigned short)lo[3], (unsigned short)hi[0], (unsigned short)hi[1],
                    (unsigned short)hi[2], (unsigned short)hi[3]};
        xb[pl] = __builtin_bit_cast(bf16x8, c);
      }
    };
    // i-th MFMA of GEMM2 for (pt, kh): feature tile dt = i / 6, piece product i % 6
    auto gemm2 = [&](int pt, int kh, int i) {
      const int dt = i / 6, t = i % 6;
#ifdef PA_GLM_PROBE_NOGEMM2
      if (t == 0) gwacc[pt][dt][kh] += __builtin_bit_cast(float, g1[pt][4 * kh] ^ g2[pt][4 * kh + 1] ^ g3[pt][4 * kh + 2]);
      return;
#endif
#ifdef PA_GLM_PROBE_NOXB
      if (t == 0 && kh == 0 && pt == 0) load_xb(kh, dt);
#else
      if (t == 0) load_xb(kh, dt);
#endif
      const uint32_t* gp = TA[t] == 0 ? g1[pt] : (TA[t] == 1 ? g2[pt] : g3[pt]);
      const bf16x8 ga = as_bf16x8(gp[4 * kh], gp[4 * kh + 1], gp[4 * kh + 2], gp[4 * kh + 3]);
      gwacc[pt][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, xb[TB[t]], gwacc[pt][dt], 0, 0, 0);
    };
    // NM matrix instructions fm(0..NM-1) with NV units of vector work fv(0..NV-1) spread between
    // them in program order (the two pipes run concurrently; a wave issues in order)
#define PA_INTERLEAVE(NM, NV, FM, FV)                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < ((NM) > 0 ? (NM) : 1); ++i_) {  \
    if (i_ < (NM)) { FM(i_); }                                             \
    const int nm_ = (NM) > 0 ? (NM) : 1;                                   \
    _Pragma("unroll") for (int j_ = i_ * (NV) / nm_; j_ < (i_ + 1) * (NV) / nm_; ++j_) { FV(j_); } \
  }

    // P0: GEMM1(pt 0)  ||  split of the next tile (registers only)
#define FM_(i) gemm1(0, i)
#define FV_(j) { if (EARLY_SPLIT) { if (j < NLD) split_unit(j, nxt); else split_row(nxt); } }
    PA_INTERLEAVE(NM1, NLD + 1, FM_, FV_)
#undef FM_
#undef FV_
    if (EARLY_SPLIT) issue_loads(nxt + tile_stride);
    if constexpr (PT == 2) {
      // P1: GEMM1(pt 1)  ||  element-wise(pt 0, registers 0..7)
#define FM_(i) gemm1(1, i)
#define FV_(j) elem(0, j)
      PA_INTERLEAVE(NM1, 4, FM_, FV_)
#undef FM_
#undef FV_
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) elem(0, j);
    }
    // P2: GEMM2(pt 0, K half 0)  ||  element-wise(pt 0, registers 8..15)
#define FM_(i) gemm2(0, 0, i)
#define FV_(j) elem(0, 4 + j)
    PA_INTERLEAVE(6 * DT, 4, FM_, FV_)
#undef FM_
#undef FV_
    if constexpr (PT == 2) {
      // P3: GEMM2(pt 0, K half 1)  ||  element-wise(pt 1, registers 0..7)
#define FM_(i) gemm2(0, 1, i)
#define FV_(j) elem(1, j)
      PA_INTERLEAVE(6 * DT, 4, FM_, FV_)
#undef FM_
#undef FV_
      // P4: GEMM2(pt 1, K half 0)  ||  element-wise(pt 1, registers 8..15)
#define FM_(i) gemm2(1, 0, i)
#define FV_(j) elem(1, 4 + j)
      PA_INTERLEAVE(6 * DT, 4, FM_, FV_)
#undef FM_
#undef FV_
      // P5: GEMM2(pt 1, K half 1)
#pragma unroll
      for (int i = 0; i < 6 * DT; ++i) gemm2(1, 1, i);
    } else {
#pragma unroll
      for (int i = 0; i < 6 * DT; ++i) gemm2(0, 1, i);
    }
#undef PA_INTERLEAVE
    // every LDS read of this tile's planes has been issued (a wave's DS operations execute in
    // program order, the slice is private to the wave): overwrite them with the next tile
    if (!EARLY_SPLIT) {
#pragma unroll
      for (int j = 0; j < NLD; ++j) split_unit(j, nxt);
      split_row(nxt);
    }
    write_xs();
    if (!EARLY_SPLIT) issue_loads(nxt + tile_stride);
#ifdef PA_GLM_SCHED_PIPELINE
    // ask the machine scheduler for an even MFMA : VALU interleave over the whole (branch-free) body
    if constexpr (DT == 1 && PT == 2) {
#pragma unroll
      for (int i = 0; i < 50; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, PA_GLM_SCHED_PIPELINE, 0);  // VALU
      }
    }
#endif
    tile = nxt;
  }
  float ll_acc[PT], gb_acc[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    ll_acc[pt] = ll2[pt].x + ll2[pt].y;
    gb_acc[pt] = gb2[pt].x + gb2[pt].y;
  }
  __syncthreads();

  // ---- block reduction over the 4 waves in a fixed order, then one partial record (the record
  //      format of glm.hip: raw accumulator tiles + ll + gb) ------------------------------------
  constexpr int REC = PT * DT * 1024 + 2 * PT * 32;
  static_assert((PT * DT * 1024 + 2 * PT * 64) * 4 <= C::LDS_BYTES, "LDS too small for epilogue");
  float* red = reinterpret_cast<float*>(smem);
  float* red2 = red + PT * DT * 1024;
  for (int wv = 0; wv < GLMB_WAVES; ++wv) {
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
  for (int i = threadIdx.x; i < PT * DT * 1024; i += 64 * GLMB_WAVES) rec[i] = red[i];
  for (int i = threadIdx.x; i < 2 * PT * 32; i += 64 * GLMB_WAVES) {
    const int q = i >> 5, j = i & 31;
    rec[PT * DT * 1024 + i] = red2[q * 64 + j] + red2[q * 64 + 32 + j];
  }
}

}  // namespace pa
