// glm_planes16w.h -- the f16 plane-image GLM pass with ONE WAVE PER 32-ROW TILE AND 64 PARTICLES
// (included by glm.hip; image, scaling rules, partial-record format and finalize of glm_planes16.h).
//
// Why (profiles/r04_glm16_ablation.txt).  In glm_planes_f16_kernel a wave owns 32 rows x 32 particles
// and the parts of its tile loop add up almost serially: besides the element-wise work and the MFMAs
// there are 11.5 us of "skeleton" per launch -- one workgroup barrier per tile (the four waves share
// a ring of super-tiles), the counted wait, the transform of the tile's observations and five LDS
// round trips for the operands -- all of it per 32 x 32 tile.  Here a wave owns a 32-row tile for BOTH
// particle tiles of the pass:
//   * the A operand of GEMM1, the transposed B operand of GEMM2 and the observations are read from
//     LDS once per 32 x 64 block of work instead of once per 32 x 32;
//   * every wave streams its own tiles through a PRIVATE ring (4 KiB per tile, no tile is fetched
//     twice: the two particle tiles live in the same wave), so the tile loop has no workgroup barrier
//     at all, only the wave's own counted s_waitcnt;
//   * the two particle tiles give the wave independent work: the MFMAs of one overlap the
//     element-wise stream of the other without a software pipeline across tiles (no second
//     accumulator set).
// Registers: W operands of both particle tiles 32, accumulators 32 + 32, B operands of both K
// halves 16, ~200 in all => two waves per SIMD, which is what the narrow kernel ran at anyway.
#pragma once
#include "glm_planes16.h"

namespace pa {

template <int NB>
struct GlmWCfg {
  static constexpr int NRT = 4;                          // row tiles per workgroup round: one per wave
  static constexpr int PW = GLMH_TILE / 1024;            // 1 KiB DMA pieces per tile
  static constexpr int NDMA = PW + 1;                    // + the tile's 32 observations
  static constexpr int WROWS = 64;
  static constexpr int WPL = WROWS * 64;                 // one W plane
  static constexpr int OFS_WAUX = 2 * WPL;               // per particle 16 B: {b1 | b2, b3, descale, -}
  static constexpr int OFS_RING = OFS_WAUX + WROWS * 16;
  static constexpr int RING_BYTES = 4 * NB * GLMH_TILE;
  static constexpr int OFS_Y = OFS_RING + RING_BYTES;
  static constexpr int LDS_BYTES = OFS_Y + NB * 4 * 256;
};

template <int NB>
__global__ __launch_bounds__(256, 2) void glm_planes_f16w_kernel(
    const unsigned char* __restrict__ img, const float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ b, int64_t N, int D, int P, int64_t ngrp /* 128-row tile groups */,
    float* __restrict__ part, const uint32_t* __restrict__ trailer,
    unsigned long long* __restrict__ tstamps, const int64_t* __restrict__ gate) {
  if (gate != nullptr && *gate != 0) return;        // the step gate gave this replay up (pa_gate)
  using C = GlmWCfg<NB>;
  constexpr int WPL = C::WPL, NPT = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int pbase = blockIdx.y * C::WROWS;

  if (tstamps != nullptr && threadIdx.x == 0)
    __hip_atomic_fetch_min(&tstamps[0], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
  const int64_t grid = gridDim.x, first = blockIdx.x;
  const int64_t my_count = first < ngrp ? (ngrp - first + grid - 1) / grid : 0;
  const uint32_t ring_w = lds_base + C::OFS_RING + wave * NB * GLMH_TILE;
  const uint32_t y_w = lds_base + C::OFS_Y + wave * 256;

  // tile (group gq, this wave) -> ring slot bi
  auto issue = [&](int64_t gq, int bi) {
#ifdef PA_GLMH_ABL_NODMA            // (timing ablations: tools/probes/glm_planes16_probe.hip)
    return;
#endif
    const int64_t gc = gq < ngrp ? gq : ngrp - 1;
    const int64_t T = gc * 4 + wave;
    const unsigned char* src = img + T * GLMH_TILE + lane * 16;
    const uint32_t dst = ring_w + bi * GLMH_TILE;
#pragma unroll
    for (int k = 0; k < C::PW; ++k) dma16(src + k * 1024, dst + k * 1024);
    int64_t row = T * 32 + l31;
    row = row < N ? row : N - 1;
    dma4(y + row, y_w + bi * 1024);
  };
#pragma unroll
  for (int k = 0; k < NB - 1; ++k) issue(first + k * grid, k);

  // ---- W planes and the per-particle constants, once per block (glm_planes16.h: same rules): thread
  //      (pl, s) holds 8 features of particle row pl ------------------------------------------------
  const int kx_l = (int)trailer[GLMH_KX + l31];        // the exponent of this lane's gradient column
  {
    const int pl = threadIdx.x >> 2, s = threadIdx.x & 3;
    const int p = pbase + pl;
    float v[8];
    float mw = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 8 * s + j;
      v[j] = (p < P && d < D) ? ldexpf(w[(int64_t)p * D + d] * GLMP_LOG2E, -(int)trailer[GLMH_KX + d]) : 0.0f;
      mw = __builtin_fmaxf(mw, __builtin_fabsf(v[j]));
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
    for (int j = 0; j < 8; ++j) v[j] = ldexpf(v[j], kw);
    uint32_t p1[4], p2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair_f16(v[2 * j], v[2 * j + 1], p1[j], p2[j]);
    unsigned char* q = smem + (pl >> 5) * GLMP_PLANE + glmp_slot_ofs(pl & 31, s);
    *reinterpret_cast<uint4*>(q) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
    *reinterpret_cast<uint4*>(q + WPL) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
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

  f16x8 b_aux[NPT];
  float dsc[NPT];
  f16x8 wa0[NPT][2], wa1[NPT][2];
  const int a_ofs0 = glmp_slot_ofs(l31, h), a_ofs1 = glmp_slot_ofs(l31, 2 + h);
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
    const uint32_t* wx_l = reinterpret_cast<const uint32_t*>(smem + C::OFS_WAUX) + 4 * (pt * 32 + l31);
    b_aux[pt] = as_f16x8(h == 0 ? wx_l[0] : 0u, h == 0 ? wx_l[1] : 0u, 0u, 0u);
    dsc[pt] = __builtin_bit_cast(float, wx_l[2]);
    const unsigned char* w_row = smem + pt * GLMP_PLANE;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      wa0[pt][pl] = *reinterpret_cast<const f16x8*>(w_row + pl * WPL + a_ofs0);
      wa1[pt][pl] = *reinterpret_cast<const f16x8*>(w_row + pl * WPL + a_ofs1);
    }
  }
  f32x16v gwacc[NPT];
  float s_yl[NPT][2], s_abs[NPT][2], s_g[NPT][2], p_t[NPT][2];
  int e_t[NPT][2];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) gwacc[pt][r] = 0.0f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      s_yl[pt][c] = s_abs[pt][c] = s_g[pt][c] = 0.0f;
      p_t[pt][c] = 1.0f;
      e_t[pt][c] = 0;
    }
  }

  const int q = lane & 15, gi1 = (lane >> 4) & 1;
  const int tr_row = 4 * h + (q >> 2);
  const int tr_slot = 2 * gi1 + ((q & 3) >> 1), tr_in = (q & 1) * 8;
  const int tr_ofs_a = tr_row * 64 + ((tr_slot ^ h) << 4) + tr_in;
  const int tr_ofs_b = (tr_row + 8) * 64 + ((tr_slot ^ ((h + 2) & 3)) << 4) + tr_in;
  constexpr int TA[3] = {1, 0, 0};
  constexpr int TB[3] = {0, 1, 0};

  auto elem1 = [&](float acc, float yh, int pt, int par) -> float {
#ifdef PA_GLMH_ABL_NOELEM
    return acc + yh;
#endif
    const float l2 = acc * dsc[pt];
#ifdef PA_GLMH_ABL_NOTRANS
    const float e = __builtin_fabsf(l2) * -0.001f;
    const float t = e + 1.0f;
    const float inv = t * 0.5f;
#else
    const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(l2));
    const float t = e + 1.0f;
    const float inv = __builtin_amdgcn_rcpf(t);
#endif
    s_yl[pt][par] = __builtin_fmaf(yh, l2, s_yl[pt][par]);
    asm("v_add_f32 %0, |%1|, %0" : "+v"(s_abs[pt][par]) : "v"(l2));
    p_t[pt][par] *= t;
    const float g = yh - __builtin_copysignf(__builtin_fmaf(inv, GLMH_GSCALE, -0.5f * GLMH_GSCALE), l2);
    s_g[pt][par] += g;
    return g;
  };
  auto renorm = [&]() {
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        e_t[pt][c2] += __builtin_amdgcn_frexp_expf(p_t[pt][c2]);
        p_t[pt][c2] = __builtin_amdgcn_frexp_mantf(p_t[pt][c2]);
      }
  };
  auto tr_issue = [&](uint32_t tr_a, uint32_t tr_b, int kh, v2u32 (&xlo)[2], v2u32 (&xhi)[2]) {
#ifdef PA_GLMH_ABL_NOTR
    xlo[0] = xhi[0] = xlo[1] = xhi[1] = v2u32{tr_a, tr_b};
    return;
#endif
    const uint32_t a = tr_a + (kh ? 1024u : 0u), b2 = tr_b + (kh ? 1024u : 0u);
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(xlo[0]) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(xhi[0]) : "v"(b2));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(xlo[1]) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(xhi[1]) : "v"(b2));
  };
  auto tr_take = [&](v2u32 (&xlo)[2], v2u32 (&xhi)[2], f16x8 (&xb)[2]) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const u32x4v cc = {xlo[pl][0], xlo[pl][1], xhi[pl][0], xhi[pl][1]};
      xb[pl] = __builtin_bit_cast(f16x8, cc);
    }
  };
  auto load_y = [&](const float* ys_, int kh, float (&yv)[8]) {
    const float4 y0 = *reinterpret_cast<const float4*>(ys_ + 16 * kh + 4 * h);
    const float4 y1 = *reinterpret_cast<const float4*>(ys_ + 16 * kh + 8 + 4 * h);
    yv[0] = y0.x; yv[1] = y0.y; yv[2] = y0.z; yv[3] = y0.w;
    yv[4] = y1.x; yv[5] = y1.y; yv[6] = y1.z; yv[7] = y1.w;
  };
  // 8 accumulator elements of one K half -> 2^14 g as the two-piece A operand of GEMM2
  auto half_tile = [&](const f32x16v& acc, int kh, const float (&yv)[8], int pt, f16x8 (&ga)[2]) {
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = elem1(acc[8 * kh + j], yv[j], pt, j & 1);
    uint32_t g1[4], g2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair_f16(g[2 * j], g[2 * j + 1], g1[j], g2[j]);
    ga[0] = as_f16x8(g1[0], g1[1], g1[2], g1[3]);
    ga[1] = as_f16x8(g2[0], g2[1], g2[2], g2[3]);
  };

  int64_t gq = first;
  int bi = 0;
  for (int64_t it = 0; it < my_count; ++it) {
    if ((it & 3) == 3) renorm();         // (1 + e <= 2: 4 tiles x 8 factors per chain stay below 2^32)
    {
      int bf = bi + (NB - 1);
      bf = bf >= NB ? bf - NB : bf;
      issue(gq + (NB - 1) * grid, bf);
    }
    wait_vmcnt<(NB - 1) * C::NDMA>();    // this wave's tile `it` has landed (its ring is private)
    const unsigned char* Xc = smem + C::OFS_RING + (wave * NB + bi) * GLMH_TILE;
    float* ysc = reinterpret_cast<float*>(smem + C::OFS_Y + bi * 1024 + wave * 256);
    const int64_t rows_left = N - (gq * 4 + wave) * 32;                       // scalar
    const bool okr = (int64_t)l31 < rows_left;
    // the tile's 32 observations become 2^14 (y - 1/2) in place (0 past the end of the plate)
    if (lane < 32) ysc[lane] = okr ? __builtin_fmaf(ysc[lane], GLMH_GSCALE, -0.5f * GLMH_GSCALE) : 0.0f;
    const uint32_t tr_a = (uint32_t)(uintptr_t)Xc + (uint32_t)tr_ofs_a;
    const uint32_t tr_b = (uint32_t)(uintptr_t)Xc + (uint32_t)tr_ofs_b;

    // -- GEMM1 of both particle tiles: the bias / validity operand, then two K chunks x three pieces
    f32x16v acc[NPT];
    {
      const uint32_t a0 = (h == 0 && okr) ? (F16_2P15 | (F16_2P15 << 16)) : 0u;   // k slots {0, 1}
      const uint32_t a1 = (h == 0 && okr) ? F16_2P15 : 0u;                        // k slot 2
      const f16x8 av = as_f16x8(a0, a1, 0u, 0u);
      const f32x16v zero = {};
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) acc[pt] = GLMH_MFMA1(av, b_aux[pt], zero);
    }
    v2u32 xlo0[2], xhi0[2], xlo1[2], xhi1[2];
    {
      f16x8 xa[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) xa[pl] = *reinterpret_cast<const f16x8*>(Xc + pl * GLMP_PLANE + a_ofs0);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[pt] = GLMH_MFMA1(xa[TA[t]], wa0[pt][TB[t]], acc[pt]);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) xa[pl] = *reinterpret_cast<const f16x8*>(Xc + pl * GLMP_PLANE + a_ofs1);
      tr_issue(tr_a, tr_b, 0, xlo0, xhi0);
      tr_issue(tr_a, tr_b, 1, xlo1, xhi1);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[pt] = GLMH_MFMA1(xa[TA[t]], wa1[pt][TB[t]], acc[pt]);
    }
    float yv0[8], yv1[8];
    load_y(ysc, 0, yv0);
    load_y(ysc, 1, yv1);
    // (the transposed B operands of both K halves: ds_read_tr results, complete after this wait)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xlo0[0]), "+v"(xhi0[0]), "+v"(xlo0[1]), "+v"(xhi0[1]), "+v"(xlo1[0]),
                   "+v"(xhi1[0]), "+v"(xlo1[1]), "+v"(xhi1[1])
                 :
                 : "memory");
    f16x8 xb0[2], xb1[2];
    tr_take(xlo0, xhi0, xb0);
    tr_take(xlo1, xhi1, xb1);

    // -- per particle tile: element-wise on a K half, its split, GEMM2 of that half
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
      f16x8 ga[2];
      half_tile(acc[pt], 0, yv0, pt, ga);
#pragma unroll
      for (int t = 0; t < 3; ++t) gwacc[pt] = GLMH_MFMA2(ga[TA[t]], xb0[TB[t]], gwacc[pt]);
      half_tile(acc[pt], 1, yv1, pt, ga);
#pragma unroll
      for (int t = 0; t < 3; ++t) gwacc[pt] = GLMH_MFMA2(ga[TA[t]], xb1[TB[t]], gwacc[pt]);
    }
    gq += grid;
    bi = bi + 1 == NB ? 0 : bi + 1;
  }
  wait_vmcnt<0>();
  renorm();
  __syncthreads();

  // ---- block reduction over the four row tiles in a fixed order, one partial record in the format of
  //      glm.hip; the power-of-two scales come out here (exact) --------------------------------------
  constexpr int REC = NPT * 1024 + 2 * NPT * 32;
  static_assert((NPT * 1024 + 2 * NPT * 64) * 4 <= C::LDS_BYTES - C::OFS_RING, "LDS too small");
  float* red = reinterpret_cast<float*>(smem + C::OFS_RING);
  float* red2 = red + NPT * 1024;
  const float g_dsc = 1.0f / GLMH_GSCALE;
  for (int rr = 0; rr < C::NRT; ++rr) {
    if (wave == rr) {
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        const float s_lg = (float)(e_t[pt][0] + e_t[pt][1]) +
                           (__builtin_amdgcn_logf(p_t[pt][0]) + __builtin_amdgcn_logf(p_t[pt][1]));
        const float ll_acc = 0.69314718055994530942f *
                             ((s_yl[pt][0] + s_yl[pt][1]) * g_dsc - 0.5f * (s_abs[pt][0] + s_abs[pt][1]) - s_lg);
        const float gb_acc = (s_g[pt][0] + s_g[pt][1]) * g_dsc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int idx = (pt * 16 + r) * 64 + lane;
          red[idx] = (rr == 0 ? 0.0f : red[idx]) + ldexpf(gwacc[pt][r], -(14 + kx_l));
        }
        const int i0 = (2 * pt) * 64 + lane, i1 = (2 * pt + 1) * 64 + lane;
        red2[i0] = (rr == 0 ? 0.0f : red2[i0]) + ll_acc;
        red2[i1] = (rr == 0 ? 0.0f : red2[i1]) + gb_acc;
      }
    }
    __syncthreads();
  }
  float* rec = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * REC;
  for (int i = threadIdx.x; i < NPT * 1024; i += 256) rec[i] = red[i];
  for (int i = threadIdx.x; i < 2 * NPT * 32; i += 256) {
    const int qq = i >> 5, j = i & 31;
    rec[NPT * 1024 + i] = red2[qq * 64 + j] + red2[qq * 64 + 32 + j];
  }
  if (tstamps != nullptr && threadIdx.x == 0)
    __hip_atomic_fetch_max(&tstamps[1], (unsigned long long)wall_clock64(), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace pa
