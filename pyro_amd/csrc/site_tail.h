// site_tail.h -- one mean-field site of the chained tail of an SVI step in ONE pass.
//
// The generic form (chain.hip) runs, per site and one after the other: the value gradient of the
// site's prior entry, the chained contribution of the guide density, the known extra term (the fused
// GLM site's gradient), the guide entry's parameter gradients, the guide draw's backward (sums over
// the particles) and Adam on the site's parameters -- six passes of table-driven code, 2-5 us each
// on one workgroup (a wave alone on its SIMD issues an instruction every ~5 cycles and the code of
// every pass starts cold).  For the shape every AutoNormal guide produces -- prior entry of any
// element-wise family scoring the latent z[P, n], guide entry Normal(loc[n], scale[n]) chained on the
// same z, nothing masked -- this file does all of it with each element loaded once and every partial
// result in registers or LDS:
//   before the wait: dz = w_h * d prior/dz + w_g * d guide/dz, the guide's parameter gradients
//                    d loc_out[c], d scale[c] (sums over the particles), eps and the Adam operands;
//   after the wait : dz += xw * extra, the backward sums  sum_p dz, sum_p dz * eps, and Adam.
// The arithmetic per number (operation order, roundings, reduction trees) is the generic form's --
// the results are bit-identical to the separate launches (tests/test_chain_gpu.py).
#pragma once
#include "multisite_dev.h"
#include "optim_dev.h"

namespace pa {

enum { SITE_LAYOUT_COLS = 0, SITE_LAYOUT_SCALAR = 1 };

struct SiteAdam {
  float *p, *g, *m, *v;
  const int64_t* step_dev;
  double lr, b1, b2, eps, wd, clip, lrd;
  int clipped, zero_grad;
};

// value-gradient accumulation steps of combined_pass / extras_pass, spelled once
__device__ __forceinline__ float site_acc(float old, float wT, float gv) {
#pragma clang fp contract(off)
  return old + wT * gv + 0.0f * 0.0f;
}

// eh: the prior entry (family F, value z), eg: the guide entry (Normal, chained on z), ms: the
// site of the guide draw; layout COLS: frame [P, n] (thread (c, g) owns rows g, g + ng, ...),
// layout SCALAR: n = 1, frame [1, P].  `wait` blocks until the extra term may be read.
// red: 4 * GRAD_THREADS doubles, xch: 4 * 64 floats of LDS (the caller's: one copy for all families)
template <int F, typename Wait>
__device__ __forceinline__ void site_fast(const EntryDev& eh, const EntryDev& eg,
                                          const MfSiteDev& ms, int layout, double coef_all,
                                          int64_t off_loc, int64_t off_rho, const SiteAdam& ad,
                                          double* red, float* xch, Wait wait, int64_t* step_out) {
#pragma clang fp contract(off)
  const uint32_t t = threadIdx.x;
  const uint32_t R = (uint32_t)eh.rows, C = (uint32_t)eh.cols;
  const uint32_t n = (uint32_t)ms.n;
  const double w_h = coef_all * eh.coef, w_g = coef_all * eg.coef;
  const float wTh = (float)w_h, wTg = (float)w_g;
  const float* eps = (const float*)ms.eps;
  const float* xg = (const float*)eh.xg;
  const float xw = (float)(coef_all * eh.xcoef);
  float* dv = (float*)eh.dv;
  float* da = (float*)eg.da;
  float* db = (float*)eg.db;

  // combined_pass's element -> thread map on the head's frame
  const uint32_t tk = C < GRAD_THREADS ? C : GRAD_THREADS, ng = row_groups(tk);
  const uint32_t c0 = t % tk, g = t / tk;
  const bool okc = g < ng && c0 < C;
  // the thread that finishes column `cfin` (backward epilogue + Adam): g == 0 in the backward's map
  const bool fin_thread = layout == SITE_LAYOUT_COLS ? (g == 0 && c0 < n) : (t == 0);
  const uint32_t cfin = layout == SITE_LAYOUT_COLS ? c0 : 0u;

  // ---- operands, all requested before anything is consumed -------------------------------------
  // (the guide entry scores the same z, and its loc / scale do not depend on the particle: two
  //  loads per thread, not two per element)
  Elem<float> xh[UN];
  float ev[UN];
  bool ok[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const uint32_t r = g + (uint32_t)u * ng;
    ok[u] = okc && r < R;
    xh[u] = load_elem<F, float>(eh, r, c0, ok[u]);
    ev[u] = eps[ok[u] ? r * C + c0 : 0u];
  }
  const uint32_t cq = okc ? c0 : 0u;
  const float a_q = ((const float*)eg.a)[cq * (int32_t)eg.asc];
  const float b_q = ((const float*)eg.b)[cq * (int32_t)eg.bsc];
  const int64_t il = off_loc + cfin, ir = off_rho + cfin;
  float g_l = ad.g[il], p_l = ad.p[il], m_l = ad.m[il], v_l = ad.v[il];
  float g_r = ad.g[ir], p_r = ad.p[ir], m_r = ad.m[ir], v_r = ad.v[ir];
  const float v_rho = ((const float*)ms.rho)[cfin];
  const int64_t step = ad.step_dev[0] + 1;
  *step_out = step;

  // ---- the two densities' gradients -------------------------------------------------------------
  float dz[UN];
  float col_a = 0.0f, col_b = 0.0f;
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    float gv, ga, gb;
    elem_grad<F, float>(xh[u], gv, ga, gb);
    gv = xh[u].keep ? gv : 0.0f;
    float d = site_acc(0.0f, wTh, gv);
    Elem<float> xq;
    xq.v = xh[u].v;
    xq.a = a_q;
    xq.b = b_q;
    xq.keep = ok[u];
    elem_grad<PA_DIST_NORMAL, float>(xq, gv, ga, gb);
    gv = xq.keep ? gv : 0.0f;
    ga = xq.keep ? ga : 0.0f;
    gb = xq.keep ? gb : 0.0f;
    dz[u] = site_acc(d, wTg, gv);
    col_a += ga;
    col_b += gb;
  }
  // ---- the guide entry's parameter gradients: sums over the particles ---------------------------
  float v_dlo = 0.0f, v_dsc = 0.0f;
  if (layout == SITE_LAYOUT_COLS) {       // combined_pass, pattern ROWRED
    red[t] = (double)col_a;
    red[GRAD_THREADS + t] = (double)col_b;
    __syncthreads();
    if (fin_thread) {
      double sa = 0.0, sb = 0.0;
      for (uint32_t j = 0; j < ng; ++j) {
        sa += red[j * tk + c0];
        sb += red[GRAD_THREADS + j * tk + c0];
      }
      v_dlo = (float)(w_g * sa);
      v_dsc = (float)(w_g * sb);
      da[c0] = v_dlo;
      db[c0] = v_dsc;
    }
  } else {                                // operand_pass, pattern COLRED on the frame [1, P]
    // element l's (ga, gb) sit in thread l (its only element: col_a, col_b): hand them to the
    // summing threads
    if (t < 64) {
      xch[t] = col_a;
      xch[64 + t] = col_b;
    }
    __syncthreads();
    const uint32_t ngc = row_groups(1u);                  // 8 summing threads, element l = t, t + 8, ...
    float acc_a = 0.0f, acc_b = 0.0f;
    if (t < ngc)
      for (uint32_t l = t; l < C; l += ngc) {
        acc_a += xch[l];
        acc_b += xch[64 + l];
      }
    __syncthreads();
    red[t] = (double)acc_a;
    red[GRAD_THREADS + t] = (double)acc_b;
    __syncthreads();
    if (t == 0) {
      double sa = 0.0, sb = 0.0;
      for (uint32_t j = 0; j < ngc; ++j) {
        sa += red[j];
        sb += red[GRAD_THREADS + j];
      }
      v_dlo = (float)(w_g * sa);
      v_dsc = (float)(w_g * sb);
      da[0] = v_dlo;
      db[0] = v_dsc;
    }
  }

  wait();

  // ---- the extra term, then the backward of the draw ---------------------------------------------
  float ex[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const uint32_t r = g + (uint32_t)u * ng;
    ex[u] = xg != nullptr ? xg[ok[u] ? r * C + c0 : 0u] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const uint32_t r = g + (uint32_t)u * ng;
    if (xg != nullptr) dz[u] = dz[u] + xw * ex[u];
    if (ok[u]) dv[r * C + c0] = dz[u];
  }
  double sl = 0.0, ss = 0.0;
  if (layout == SITE_LAYOUT_COLS) {       // the backward's map IS this one: same thread, same rows
    float al = 0.0f, as = 0.0f;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      al += ok[u] ? dz[u] : 0.0f;
      as += ok[u] ? dz[u] * ev[u] : 0.0f;
    }
    __syncthreads();
    red[2 * GRAD_THREADS + t] = (double)al;
    red[3 * GRAD_THREADS + t] = (double)as;
    __syncthreads();
    if (fin_thread)
      for (uint32_t j = 0; j < ng; ++j) {
        sl += red[2 * GRAD_THREADS + j * tk + c0];
        ss += red[3 * GRAD_THREADS + j * tk + c0];
      }
  } else {                                // n = 1: particle p's dz sits in thread p; thread g sums
    __syncthreads();                      // p = g, g + 8, ...
    if (t < 64) {
      xch[2 * 64 + t] = ok[0] ? dz[0] : 0.0f;
      xch[3 * 64 + t] = ok[0] ? ev[0] : 0.0f;
    }
    __syncthreads();
    const uint32_t ngc = row_groups(1u);
    float al = 0.0f, as = 0.0f;
    if (t < ngc)
      for (uint32_t p = t; p < C; p += ngc) {
        const float gz = xch[2 * 64 + p], e1 = xch[3 * 64 + p];
        al += gz;
        as += gz * e1;
      }
    __syncthreads();
    red[2 * GRAD_THREADS + t] = (double)al;
    red[3 * GRAD_THREADS + t] = (double)as;
    __syncthreads();
    if (t == 0)
      for (uint32_t j = 0; j < ngc; ++j) {
        sl += red[2 * GRAD_THREADS + j];
        ss += red[3 * GRAD_THREADS + j];
      }
  }
  if (fin_thread) {
    ss += (double)v_dsc;
    sl += (double)v_dlo;
    const double sig = softplus_slope<float>(v_rho);
    const float gl = (float)sl + g_l, gr = (float)(ss * sig) + g_r;
    const AdamStep st = adam_step_scalars(step, ad.lr, ad.b1, ad.b2, ad.lrd, ad.clipped);
    adam_update<float>(gl, p_l, m_l, v_l, st, ad.b1, ad.b2, ad.eps, ad.wd, ad.clip, ad.clipped);
    adam_update<float>(gr, p_r, m_r, v_r, st, ad.b1, ad.b2, ad.eps, ad.wd, ad.clip, ad.clipped);
    ad.m[il] = m_l; ad.v[il] = v_l; ad.p[il] = p_l;
    ad.m[ir] = m_r; ad.v[ir] = v_r; ad.p[ir] = p_r;
    ad.g[il] = ad.zero_grad ? 0.0f : gl;
    ad.g[ir] = ad.zero_grad ? 0.0f : gr;
  }
}

}  // namespace pa
