// glm_planes16d.h -- the f16 plane-image GLM pass for 32 < D <= 128 features (included by glm.hip; the
// image format, scaling rules, partial-record format and finalize of glm_planes16.h with DT = 2 or 4
// FEATURE TILES of 32 columns per 32-row tile).
//
// Until round 4 the plane image stopped at D = 32: D = 33..64 ran the kernel that splits X on the fly
// (187 us at N = 1e6, D = 64) and D = 65..128 the exact-f32 MFMA kernel (489 us at D = 128: 1 TB/s),
// both in two passes over X for 64 particles.  Here:
//   image   per 32-row tile DT sub-tiles of the D <= 32 format (two swizzled f16 planes of 32 columns,
//           4 KiB each), tile T's sub-tile dt at (T * DT + dt) * 4 KiB; one power-of-two exponent per
//           column in a 1-KiB trailer (u32[128] column maxima, i32[128] exponents);
//   kernel  workgroup = 2 row tiles x 2 particle tiles (one pass over the image for 64 particles), a ring
//           of super-tiles (2 x DT x 4 KiB) fed by LDS-DMA, one barrier per tile; the W planes of all
//           feature tiles stay in LDS (registers hold one K chunk at a time); GEMM1 runs 2 DT K chunks
//           x 3 piece products into one accumulator, the element-wise stage and the split are those of
//           glm_planes16.h, GEMM2 runs per feature tile into DT accumulators.
// The stream is the bound from D = 64 upwards (the image is D / 32 times the D = 32 one; the element-wise
// work per row does not grow with D).
#pragma once
#include "glm_planes16.h"

namespace pa {

constexpr int GLMD_TRAILER = 1024;          // u32[128] column max |x| bits, i32[128] exponents
constexpr int GLMD_KX = 128;

// column maxima for D <= 128: 8 rows x 32 columns per step and feature tile
__global__ __launch_bounds__(256) void glmd_absmax_kernel(const float* __restrict__ X, int64_t N, int D,
                                                          uint32_t* __restrict__ out) {
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  for (int d = c; d < D; d += 32) {
    uint32_t m = 0u;
    for (int64_t r = (int64_t)blockIdx.x * 8 + r0; r < N; r += (int64_t)gridDim.x * 8) {
      const uint32_t v = __builtin_bit_cast(uint32_t, X[r * D + d]) & 0x7fffffffu;
      m = v > m ? v : m;
    }
    const uint32_t t = (uint32_t)__shfl_xor((int)m, 32);
    m = t > m ? t : m;
    if ((threadIdx.x & 63) < 32 && m != 0u) atomicMax(out + d, m);
  }
}

// one thread per (tile, feature tile, row, slot)
template <int DT>
__global__ __launch_bounds__(256) void glmd_pack_kernel(const float* __restrict__ X, int64_t N, int D,
                                                        int64_t ntiles, unsigned char* __restrict__ img,
                                                        uint32_t* __restrict__ trailer) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < 128) trailer[GLMD_KX + idx] = (uint32_t)glmh_exponent_of(trailer[idx]);
  if (idx >= ntiles * DT * 128) return;
  const int64_t sub = idx >> 7;                     // (tile, feature tile)
  const int64_t T = sub / DT;
  const int dt = (int)(sub - T * DT);
  const int r = (int)(idx >> 2) & 31, s = (int)idx & 3;
  const int64_t row = T * 32 + r;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int d = 32 * dt + 8 * s + j;
    v[j] = (row < N && d < D) ? ldexpf(X[row * D + d], glmh_exponent_of(trailer[d])) : 0.0f;
  }
  glmh_store_slot(v, img + sub * GLMH_TILE + glmp_slot_ofs(r, s));
}

template <int DT, int NB>
struct GlmDCfg {
  static constexpr int NRT = 2, NPT = 2;
  static constexpr int RT_BYTES = DT * GLMH_TILE;          // one 32-row tile: DT sub-tiles
  static constexpr int ST_BYTES = NRT * RT_BYTES;          // super-tile image (64 rows)
  static constexpr int PW = ST_BYTES / 1024 / 4;           // 1 KiB DMA pieces per wave and super-tile
  static constexpr int NDMA = PW + 1;
  static constexpr int WROWS = 64;
  static constexpr int WPL = WROWS * 64;                   // one W plane of one feature tile
  static constexpr int W_BYTES = DT * 2 * WPL;
  static constexpr int OFS_WAUX = W_BYTES;                 // per particle 16 B: {b1 | b2, b3, descale, -}
  static constexpr int OFS_RING = OFS_WAUX + WROWS * 16;
  static constexpr int OFS_Y = OFS_RING + NB * ST_BYTES;
  static constexpr int LDS_BYTES = OFS_Y + NB * 4 * 256;
};

template <int DT, int NB, int OCC>
__global__ __launch_bounds__(256, OCC) void glm_planes_f16d_kernel(
    const unsigned char* __restrict__ img, const float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ b, int64_t N, int D, int P, int64_t nst, float* __restrict__ part,
    const uint32_t* __restrict__ trailer, const int64_t* __restrict__ gate) {
  if (gate != nullptr && *gate != 0) return;        // the step gate gave this replay up (pa_gate)
  using C = GlmDCfg<DT, NB>;
  constexpr int NRT = C::NRT, NPT = C::NPT, WPL = C::WPL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int rt = wave / NPT, pt = wave % NPT;
  const int pbase = blockIdx.y * C::WROWS;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
  const int64_t grid = gridDim.x, first = blockIdx.x;
  const int64_t my_count = first < nst ? (nst - first + grid - 1) / grid : 0;

  auto issue = [&](int64_t st, int bi) {
    const int64_t stc = st < nst ? st : nst - 1;
    const unsigned char* src = img + stc * C::ST_BYTES + (wave * C::PW) * 1024 + lane * 16;
    const uint32_t dst = lds_base + C::OFS_RING + bi * C::ST_BYTES + (wave * C::PW) * 1024;
#pragma unroll
    for (int k = 0; k < C::PW; ++k) dma16(src + k * 1024, dst + k * 1024);
    int64_t row = (stc * NRT + rt) * 32 + l31;
    row = row < N ? row : N - 1;
    dma4(y + row, lds_base + C::OFS_Y + (bi * 4 + wave) * 256);
  };
#pragma unroll
  for (int k = 0; k < NB - 1; ++k) issue(first + k * grid, k);

  // ---- W planes of all feature tiles and the per-particle constants, once per block: thread (pl, s)
  //      holds 8 features of particle row pl in every feature tile (glm_planes16.h: same rules, the row's
  //      exponent kw[p] from the maximum over ALL its columns) --------------------------------------------
  {
    const int pl = threadIdx.x >> 2, s = threadIdx.x & 3;
    const int p = pbase + pl;
    float v[DT][8];
    float mw = 0.0f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = 32 * dt + 8 * s + j;
        v[dt][j] = (p < P && d < D) ? ldexpf(w[(int64_t)p * D + d] * GLMP_LOG2E, -(int)trailer[GLMD_KX + d]) : 0.0f;
        mw = __builtin_fmaxf(mw, __builtin_fabsf(v[dt][j]));
      }
    mw = __builtin_fmaxf(mw, __shfl_xor(mw, 1));
    mw = __builtin_fmaxf(mw, __shfl_xor(mw, 2));
    const float b2 = (p < P && b != nullptr) ? b[p] * GLMP_LOG2E : 0.0f;
    const uint32_t mwb = __builtin_bit_cast(uint32_t, mw), bb = __builtin_bit_cast(uint32_t, b2) & 0x7fffffffu;
    const int ew = (int)(mwb >> 23), eb = (int)(bb >> 23);
    int kw = (mwb != 0u && ew != 0xff) ? 14 - (ew == 0 ? -127 : ew - 127) : GLMH_KNONE;
    const int kb = (bb != 0u && eb != 0xff) ? 29 - (eb == 0 ? -127 : eb - 127) : GLMH_KNONE;
    kw = kw < kb ? kw : kb;
    if (kw == GLMH_KNONE) kw = 0;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      uint32_t p1[4], p2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        split_pair_f16(ldexpf(v[dt][2 * j], kw), ldexpf(v[dt][2 * j + 1], kw), p1[j], p2[j]);
      unsigned char* q = smem + dt * 2 * WPL + (pl >> 5) * GLMP_PLANE + glmp_slot_ofs(pl & 31, s);
      *reinterpret_cast<uint4*>(q) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
      *reinterpret_cast<uint4*>(q + WPL) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
    }
    if (s == 0) {
      const float bs = ldexpf(b2, kw - 15);
      uint32_t q1, q2, q3, dummy;
      split_pair_f16(bs, 0.0f, q1, q2);
      const float r2 = (bs - f16_lo(q1)) - f16_lo(q2);
      split_pair_f16(r2, 0.0f, q3, dummy);
      uint32_t* wx = reinterpret_cast<uint32_t*>(smem + C::OFS_WAUX) + 4 * pl;
      wx[0] = (q1 & 0xffffu) | (q2 << 16);
      wx[1] = q3 & 0xffffu;
      int kd = -kw;
      kd = kd > 126 ? 126 : (kd < -126 ? -126 : kd);
      wx[2] = __builtin_bit_cast(uint32_t, ldexpf(1.0f, kd));
      wx[3] = 0u;
    }
  }
  __syncthreads();

  const uint32_t* wx_l = reinterpret_cast<const uint32_t*>(smem + C::OFS_WAUX) + 4 * (pt * 32 + l31);
  const f16x8 b_aux = as_f16x8(h == 0 ? wx_l[0] : 0u, h == 0 ? wx_l[1] : 0u, 0u, 0u);
  const float dsc = __builtin_bit_cast(float, wx_l[2]);
  f32x16v gwacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) gwacc[dt][r] = 0.0f;
  float s_yl[2] = {0.0f, 0.0f}, s_abs[2] = {0.0f, 0.0f}, s_g[2] = {0.0f, 0.0f};
  float p_t[2] = {1.0f, 1.0f};
  int e_t[2] = {0, 0};

  const int a_ofs0 = glmp_slot_ofs(l31, h), a_ofs1 = glmp_slot_ofs(l31, 2 + h);
  const unsigned char* w_row = smem + pt * GLMP_PLANE;
  const int q = lane & 15, gi1 = (lane >> 4) & 1;
  const int tr_row = 4 * h + (q >> 2);
  const int tr_slot = 2 * gi1 + ((q & 3) >> 1), tr_in = (q & 1) * 8;
  const int tr_ofs_a = tr_row * 64 + ((tr_slot ^ h) << 4) + tr_in;
  const int tr_ofs_b = (tr_row + 8) * 64 + ((tr_slot ^ ((h + 2) & 3)) << 4) + tr_in;
  constexpr int TA[3] = {1, 0, 0};
  constexpr int TB[3] = {0, 1, 0};

  auto elem2 = [&](float acc0, float acc1, float yh0, float yh1, int chain, float& g0, float& g1) {
    const float l0 = acc0 * dsc, l1 = acc1 * dsc;
    const float t0 = __builtin_amdgcn_exp2f(-__builtin_fabsf(l0)) + 1.0f;
    const float t1 = __builtin_amdgcn_exp2f(-__builtin_fabsf(l1)) + 1.0f;
    const float tt = t0 * t1;
    const float r = __builtin_amdgcn_rcpf(tt);
    const float inv0 = r * t1, inv1 = r * t0;
    s_yl[0] = __builtin_fmaf(yh0, l0, s_yl[0]);
    s_yl[1] = __builtin_fmaf(yh1, l1, s_yl[1]);
    asm("v_add_f32 %0, |%1|, %0" : "+v"(s_abs[0]) : "v"(l0));
    asm("v_add_f32 %0, |%1|, %0" : "+v"(s_abs[1]) : "v"(l1));
    p_t[chain] *= tt;
    g0 = yh0 - __builtin_copysignf(__builtin_fmaf(inv0, GLMH_GSCALE, -0.5f * GLMH_GSCALE), l0);
    g1 = yh1 - __builtin_copysignf(__builtin_fmaf(inv1, GLMH_GSCALE, -0.5f * GLMH_GSCALE), l1);
    s_g[0] += g0;
    s_g[1] += g1;
  };
  auto renorm = [&]() {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      e_t[c2] += __builtin_amdgcn_frexp_expf(p_t[c2]);
      p_t[c2] = __builtin_amdgcn_frexp_mantf(p_t[c2]);
    }
  };

  int64_t st = first;
  int bi = 0;
  for (int64_t it = 0; it < my_count; ++it) {
    if ((it & 7) == 7) renorm();
    wait_vmcnt<(NB - 2) * C::NDMA>();          // this wave's pieces of super-tile `it` have landed ...
    __builtin_amdgcn_s_barrier();              // ... and so have the other waves'; slot bi - 1 is free
    {
      int bf = bi + (NB - 1);
      bf = bf >= NB ? bf - NB : bf;
      issue(st + (NB - 1) * grid, bf);
    }
    const unsigned char* Xc = smem + C::OFS_RING + bi * C::ST_BYTES + rt * C::RT_BYTES;
    float* ysc = reinterpret_cast<float*>(smem + C::OFS_Y + (bi * 4 + wave) * 256);
    const int64_t rows_left = N - (st * NRT + rt) * 32;                        // scalar
    const bool okr = (int64_t)l31 < rows_left;
    // the tile's 32 observations become 2^14 (y - 1/2) in place (0 past the end).  Both particle-tile
    // waves of a row tile hold their own copy of the observations (dma4 per wave)
    if (lane < 32) ysc[lane] = okr ? __builtin_fmaf(ysc[lane], GLMH_GSCALE, -0.5f * GLMH_GSCALE) : 0.0f;

    // -- GEMM1: bias / validity operand, then DT x 2 K chunks x 3 piece products
    f32x16v acc;
    {
      const uint32_t a0 = (h == 0 && okr) ? (F16_2P15 | (F16_2P15 << 16)) : 0u;
      const uint32_t a1 = (h == 0 && okr) ? F16_2P15 : 0u;
      const f32x16v zero = {};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(a0, a1, 0u, 0u), b_aux, zero, 0, 0, 0);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ao = c == 0 ? a_ofs0 : a_ofs1;
        f16x8 xa[2], wa[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          xa[pl] = *reinterpret_cast<const f16x8*>(Xc + dt * GLMH_TILE + pl * GLMP_PLANE + ao);
          wa[pl] = *reinterpret_cast<const f16x8*>(w_row + dt * 2 * WPL + pl * WPL + ao);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[TA[t]], wa[TB[t]], acc, 0, 0, 0);
      }

    // -- element-wise on the 16 accumulator elements, the two-piece split of g per K half
    f16x8 ga[2][2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const float4 y0 = *reinterpret_cast<const float4*>(ysc + 16 * kh + 4 * h);
      const float4 y1 = *reinterpret_cast<const float4*>(ysc + 16 * kh + 8 + 4 * h);
      const float yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
      float g[8];
#pragma unroll
      for (int qp = 0; qp < 4; ++qp)
        elem2(acc[8 * kh + 2 * qp], acc[8 * kh + 2 * qp + 1], yv[2 * qp], yv[2 * qp + 1], qp & 1, g[2 * qp],
              g[2 * qp + 1]);
      uint32_t g1[4], g2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split_pair_f16(g[2 * j], g[2 * j + 1], g1[j], g2[j]);
      ga[kh][0] = as_f16x8(g1[0], g1[1], g1[2], g1[3]);
      ga[kh][1] = as_f16x8(g2[0], g2[1], g2[2], g2[3]);
    }

    // -- GEMM2 per feature tile: the transposed B operand of both K halves, 2 x 3 piece products
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const uint32_t tb = (uint32_t)(uintptr_t)(Xc + dt * GLMH_TILE);
      const uint32_t tr_a = tb + (uint32_t)tr_ofs_a, tr_b = tb + (uint32_t)tr_ofs_b;
      v2u32 xlo[2][2], xhi[2][2];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const uint32_t a = tr_a + (kh ? 1024u : 0u), b2 = tr_b + (kh ? 1024u : 0u);
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(xlo[kh][0]) : "v"(a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(xhi[kh][0]) : "v"(b2));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(xlo[kh][1]) : "v"(a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(xhi[kh][1]) : "v"(b2));
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(xlo[0][0]), "+v"(xhi[0][0]), "+v"(xlo[0][1]), "+v"(xhi[0][1]), "+v"(xlo[1][0]),
                     "+v"(xhi[1][0]), "+v"(xlo[1][1]), "+v"(xhi[1][1])
                   :
                   : "memory");
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        f16x8 xb[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const u32x4v cc = {xlo[kh][pl][0], xlo[kh][pl][1], xhi[kh][pl][0], xhi[kh][pl][1]};
          xb[pl] = __builtin_bit_cast(f16x8, cc);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
          gwacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga[kh][TA[t]], xb[TB[t]], gwacc[dt], 0, 0, 0);
      }
    }
    st += grid;
    bi = bi + 1 == NB ? 0 : bi + 1;
  }
  wait_vmcnt<0>();
  renorm();
  __syncthreads();

  // ---- block reduction over the row tiles in a fixed order, one partial record in the format of
  //      glm.hip (DT feature tiles x 2 particle tiles); the power-of-two scales come out here -------------
  constexpr int REC = NPT * DT * 1024 + 2 * NPT * 32;
  static_assert((NPT * DT * 1024 + 2 * NPT * 64) * 4 <= C::LDS_BYTES - C::OFS_RING, "LDS too small");
  float* red = reinterpret_cast<float*>(smem + C::OFS_RING);
  float* red2 = red + NPT * DT * 1024;
  const float g_dsc = 1.0f / GLMH_GSCALE;
  const float s_lg = (float)(e_t[0] + e_t[1]) + (__builtin_amdgcn_logf(p_t[0]) + __builtin_amdgcn_logf(p_t[1]));
  const float ll_acc = 0.69314718055994530942f *
                       ((s_yl[0] + s_yl[1]) * g_dsc - 0.5f * (s_abs[0] + s_abs[1]) - s_lg);
  const float gb_acc = (s_g[0] + s_g[1]) * g_dsc;
  for (int rr = 0; rr < NRT; ++rr) {
    if (rt == rr) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int kx_l = (int)trailer[GLMD_KX + 32 * dt + l31];      // this lane's gradient column
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int idx = ((pt * DT + dt) * 16 + r) * 64 + lane;
          red[idx] = (rr == 0 ? 0.0f : red[idx]) + ldexpf(gwacc[dt][r], -(14 + kx_l));
        }
      }
      const int i0 = (2 * pt) * 64 + lane, i1 = (2 * pt + 1) * 64 + lane;
      red2[i0] = (rr == 0 ? 0.0f : red2[i0]) + ll_acc;
      red2[i1] = (rr == 0 ? 0.0f : red2[i1]) + gb_acc;
    }
    __syncthreads();
  }
  float* rec = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * REC;
  for (int i = threadIdx.x; i < NPT * DT * 1024; i += 256) rec[i] = red[i];
  for (int i = threadIdx.x; i < 2 * NPT * 32; i += 256) {
    const int qq = i >> 5, j = i & 31;
    rec[NPT * DT * 1024 + i] = red2[qq * 64 + j] + red2[qq * 64 + 32 + j];
  }
}

}  // namespace pa
