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
  static constexpr int WAVE_BYTES = 3 * PLANE + 256;   // + yh[32] f32 + aux[32] u32
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
// leading piece does not overflow bf16)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& p1, uint32_t& p2,
                                           uint32_t& p3) {
  p1 = cvt_pk_bf16(a, b);
  const float ra = a - bf16_lo(p1), rb = b - bf16_hi(p1);
  p2 = cvt_pk_bf16(ra, rb);
  const float sa = ra - bf16_lo(p2), sb = rb - bf16_hi(p2);
  p3 = cvt_pk_bf16(sa, sb);
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4v v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}

// bf16 pattern of -1e30 (top half of the f32 encoding 0xF149F2CA): the logit offset of rows that
// must not contribute (masked, or beyond the end of the plate)
constexpr uint32_t BF16_NEG_HUGE = 0xF149u;
constexpr uint32_t BF16_ONE = 0x3F80u;

template <int DT, int PT, bool GROUPED>
__global__ __launch_bounds__(64 * GLMB_WAVES, 2) void glm_bernoulli_bf16_kernel(
    const float* __restrict__ X, const float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ b, const uint8_t* __restrict__ mask, int64_t N, int D, int P,
    int64_t iters, float* __restrict__ part, const int64_t* __restrict__ seg, int G) {
  using C = GlmBfCfg<DT, PT>;
  constexpr int DP = C::DP, KC = C::KC, RS = C::RS, PLANE = C::PLANE, WROWS = C::WROWS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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

  // ---- staging registers: the next tile travels global -> VGPR while this one computes ------
  constexpr int NLD = 4 * DT;                 // float4 loads per lane per tile
  float4 stage[NLD];
  float st_y = 0.0f;
  uint8_t st_m = 0;
  const int D4 = D >> 2;                      // float4 per row
  const int64_t total_e = row_end * (int64_t)D;

  // The loads are issued with CLAMPED addresses and consumed raw: validity is applied only in
  // write_stage(), one whole tile of compute later.  (Selecting `ok ? v : 0` next to the load
  // makes the compiler wait for the load right there -- s_waitcnt vmcnt(0) in front of the
  // compute block -- which exposes the full HBM latency on every tile.)
  auto issue_loads = [&](int64_t tile) {
    const int64_t base = (row_begin + tile * 32) * (int64_t)D;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = j * 64 + lane;            // float4 index inside the tile
      const int64_t e = base + 4 * (int64_t)f;
      const bool ok = (f < 8 * D) && (e < total_e);
      stage[j] = *reinterpret_cast<const float4*>(X + (ok ? e : 0));
    }
    const int64_t n = row_begin + tile * 32 + l31;
    const int64_t nc = n < row_end ? n : 0;
    st_y = y[nc];
    st_m = mask == nullptr ? (uint8_t)1 : mask[nc];
  };
  // (row, col) of this lane's j-th float4 and the increments between consecutive j
  const int n_first = lane / D4, d_first = (lane % D4) * 4;
  const int qn = 64 / D4, qd = (64 % D4) * 4;
  auto write_stage = [&](int64_t tile) {
    const int64_t base = (row_begin + tile * 32) * (int64_t)D;
    int n = n_first, d = d_first;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int f = j * 64 + lane;
      if (f < 8 * D) {
        const bool ok = base + 4 * (int64_t)f < total_e;
        uint32_t a1, a2, a3, b1, b2, b3;
        split_pair(ok ? stage[j].x : 0.0f, ok ? stage[j].y : 0.0f, a1, a2, a3);
        split_pair(ok ? stage[j].z : 0.0f, ok ? stage[j].w : 0.0f, b1, b2, b3);
        unsigned char* q = Xp + n * RS + d * 2;
        *reinterpret_cast<uint2*>(q) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(q + PLANE) = make_uint2(a2, b2);
        *reinterpret_cast<uint2*>(q + 2 * PLANE) = make_uint2(a3, b3);
      }
      n += qn;
      d += qd;
      if (d >= D) { d -= D; n += 1; }
    }
    if (h == 0) {
      const bool okr = (row_begin + tile * 32 + l31 < row_end) && st_m != 0;
      // scale_and_mask is where(mask, x, 0) (pyro/distributions/util.py:326): a row that does not
      // count gets y = 0 and the -1e30 logit offset
      yh_s[l31] = (okr ? st_y : 0.0f) - 0.5f;
      aux_s[l31] = (BF16_ONE << 16) | (okr ? 0u : BF16_NEG_HUGE);   // {offset, 1.0}
    }
  };

  const int64_t ntiles = (row_end - row_begin + 31) / 32;
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
  write_stage(tile);
  __syncthreads();

  bf16x8 b_aux[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
    b_aux[pt] = as_bf16x8(h == 0 ? waux[2 * (pt * 32 + l31)] : 0u,
                          h == 0 ? waux[2 * (pt * 32 + l31) + 1] : 0u, 0u, 0u);

  f32x16v gwacc[PT][DT];
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

  const unsigned char* a_row = Xp + l31 * RS + 16 * h;            // A operand of GEMM1
  const unsigned char* w_row = Wp + l31 * RS + 16 * h;            // B operand of GEMM1
  const unsigned char* x_col = Xp + (4 * h) * RS + l31 * 2;       // B operand of GEMM2

  for (int64_t it = 0; it < iters; ++it) {
    const int64_t next = tile + tile_stride;
    if (it + 1 < iters) issue_loads(next);

    if (tile < ntiles) {
      // ---- GEMM1 (+ bias / row-offset MFMA) ------------------------------------------------
      f32x16v acc[PT];
      {
        const bf16x8 a_aux = as_bf16x8(h == 0 ? aux_s[l31] : 0u,
                                       h == 0 ? (BF16_ONE | (BF16_ONE << 16)) : 0u, 0u, 0u);
        const f32x16v zero = {};
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_aux, b_aux[pt], zero, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(a_row + 32 * c);
        const bf16x8 x2 = *reinterpret_cast<const bf16x8*>(a_row + 32 * c + PLANE);
        const bf16x8 x3 = *reinterpret_cast<const bf16x8*>(a_row + 32 * c + 2 * PLANE);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const unsigned char* wq = w_row + pt * 32 * RS + 32 * c;
          const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wq);
          const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wq + WROWS * RS);
          const bf16x8 w3 = *reinterpret_cast<const bf16x8*>(wq + 2 * WROWS * RS);
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, w1, acc[pt], 0, 0, 0);
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, w2, acc[pt], 0, 0, 0);
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w3, acc[pt], 0, 0, 0);
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, w1, acc[pt], 0, 0, 0);
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w2, acc[pt], 0, 0, 0);
          acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w1, acc[pt], 0, 0, 0);
        }
      }

      // ---- element-wise on the accumulator registers; g -> three bf16 planes in registers ----
      // register r of a lane (p = l31, h) is row n = (r&3) + 8*(r>>2) + 4*h: yh for r = 4q..4q+3
      // is the float4 at yh_s[8q + 4h]
      float yh[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(yh_s + 8 * q + 4 * h);
        yh[4 * q + 0] = v.x; yh[4 * q + 1] = v.y; yh[4 * q + 2] = v.z; yh[4 * q + 3] = v.w;
      }
      uint32_t g1[PT][8], g2[PT][8], g3[PT][8];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        float gv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float l = acc[pt][r];
          const float a = fabsf(l);
          // e = exp(-|l|) via v_exp_f32 (2^x); t = 1 + e in (1, 2]
          const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * a);
          const float t = 1.0f + e;
          const float lg = __builtin_amdgcn_logf(t);       // log2(1 + e)
          const float inv = __builtin_amdgcn_rcpf(t);      // sigmoid(|l|) in [0.5, 1)
          // y*l - softplus(l) = (y - 1/2) l - |l|/2 - ln2*log2(1 + e); a row with the -1e30
          // offset gives (-1/2)(-1e30) - 1e30/2 - 0 = 0 exactly
          float u = yh[r] * l;
          u = __builtin_fmaf(-0.5f, a, u);
          u = __builtin_fmaf(-0.69314718055994530942f, lg, u);
          ll_acc[pt] += u;
          // sigmoid(l) - 1/2 = copysign(inv - 1/2, l): g = y - sigmoid(l) = yh - that
          const float gg = yh[r] - __builtin_copysignf(inv - 0.5f, l);
          gb_acc[pt] += gg;
          gv[r] = gg;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) split_pair(gv[2 * i], gv[2 * i + 1], g1[pt][i], g2[pt][i], g3[pt][i]);
      }

      // ---- GEMM2: K half kh = accumulator registers 8kh..8kh+7 (rows n(r, h)) -----------------
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          u16x8v c1, c2, c3;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * kh + j;
            const int nrow = (r & 3) + 8 * (r >> 2);           // + 4h is folded into x_col
            const unsigned char* q = x_col + nrow * RS + dt * 64;
            c1[j] = *reinterpret_cast<const unsigned short*>(q);
            c2[j] = *reinterpret_cast<const unsigned short*>(q + PLANE);
            c3[j] = *reinterpret_cast<const unsigned short*>(q + 2 * PLANE);
          }
          const bf16x8 x1 = __builtin_bit_cast(bf16x8, c1);
          const bf16x8 x2 = __builtin_bit_cast(bf16x8, c2);
          const bf16x8 x3 = __builtin_bit_cast(bf16x8, c3);
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) {
            const bf16x8 a1 = as_bf16x8(g1[pt][4 * kh], g1[pt][4 * kh + 1], g1[pt][4 * kh + 2], g1[pt][4 * kh + 3]);
            const bf16x8 a2 = as_bf16x8(g2[pt][4 * kh], g2[pt][4 * kh + 1], g2[pt][4 * kh + 2], g2[pt][4 * kh + 3]);
            const bf16x8 a3 = as_bf16x8(g3[pt][4 * kh], g3[pt][4 * kh + 1], g3[pt][4 * kh + 2], g3[pt][4 * kh + 3]);
            f32x16v t = gwacc[pt][dt];
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, x1, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x2, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x3, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, x1, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x2, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, x1, t, 0, 0, 0);
            gwacc[pt][dt] = t;
          }
        }
      }
    }
    // the LDS slice is private to this wave and a wave's DS operations execute in program order:
    // re-staging needs no workgroup barrier
    if (it + 1 < iters) write_stage(next);
    tile = next;
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
