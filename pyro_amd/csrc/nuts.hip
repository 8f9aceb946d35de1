// nuts.hip -- one full NUTS transition per chain, one wavefront per chain, for the
// closed-form Gaussian potential U(z) = 0.5 z^T Lambda z (BASELINE config 3).
//
// Reference semantics restated (per chain):  pyro/infer/mcmc/nuts.py
//   sample()          :367-522   doubling loop, slice / multinomial weights, accept draw
//   _build_tree()     :250-365   recursive; here ITERATIVE (leaf index bits = merge schedule)
//   _build_basetree() :197-248   one leapfrog (pyro/ops/integrator.py:45-65) + energies
//   _is_turning()     :184-195
//   _logaddexp()      :15-17
//   momentum          : hmc.py:231-248 + BlockMassMatrix.scale/unscale/kinetic_grad
//                       (adaptation.py:328-392, diagonal mass)
// The reference has no chain batch (one Python process per chain, api.py:239-351); here a
// chain is a wavefront, so the data-dependent tree depth costs nothing to the other chains:
// all control flow is wave-uniform (s_cbranch), the state vector lives in VGPRs (lane i owns
// coordinates i and i+64, D <= 128), Lambda is staged once per workgroup in LDS and read
// column-wise (Lambda is symmetric) so every ds_read is conflict-free, z_j is broadcast with
// v_readlane.  The pending-subtree stack of the iterative tree build (<= max_tree_depth
// entries of {first momentum, momentum sum, proposal}) is wave-private LDS.
//
// Randomness: every draw is a pure function of (seed, chain, t, slot) through Philox4x32-10,
// so the iterative build consumes exactly the draws the recursive reference formulation
// would (oracle/nuts.py restates the recursion with the same keyed draws):
//   counter_lo = t * 2^20 + slot, counter_hi = chain
//   slot [0,1024)       momentum normals (4 per block f32 / 2 per block f64)
//   slot 1024           slice variable  (-log u)
//   slot 1025 + j       .x(.y): direction draw of doubling j;  .z(.w): accept draw
//   slot 2048 + 2^j + id  merge draw of doubling j, id = 2^(j-k) + (i >> k) for the merge
//                        producing the level-k subtree that ends at leaf i.
#include "nuts_common.h"

namespace pa {

constexpr int NUTS_DMAX = 128;

__device__ __forceinline__ float bcast_lane(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ double bcast_lane(double v, int j) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, j);
  hi = __builtin_amdgcn_readlane(hi, j);
  return __hiloint2double(hi, lo);
}
template <typename T> struct P2 {  // a lane's two coordinates (i, i+64)
  T a, b;
};
template <typename T> __device__ __forceinline__ P2<T> operator+(P2<T> x, P2<T> y) {
  return P2<T>{x.a + y.a, x.b + y.b};
}
template <typename T> __device__ __forceinline__ T dot(P2<T> x, P2<T> y) {
  // fixed contraction (round the .b product, then one fma) so every kernel variant that inlines
  // this computes bit-identical energies whatever the surrounding code looks like
  T p;
  {
#pragma clang fp contract(off)
    p = x.b * y.b;
  }
  // (f32 chains reduce in f32: __builtin_fma on floats is the DOUBLE fma -- until round 4 every dot of
  //  an f32 chain ran a float64 DPP tree: two half-rate adds, two DPP moves, two zero fills and two
  //  v_readlane per step instead of one v_add_f32_dpp; four dots per leapfrog)
  if constexpr (sizeof(T) == 4) return uni(wave_sum(__builtin_fmaf(x.a, y.a, p)));
  else return uni(wave_sum(__builtin_fma(x.a, y.a, p)));
}

// ---- potentials: g = Lambda z (Lambda symmetric, read as columns), pe = 0.5 z.g ---------------
// PotLds: Lambda staged once per workgroup in LDS (any dtype, D <= 128).
template <typename T>
struct PotLds {
  const T* Ls;
  int D, lane;
  // Four partial sums per column (rows j = k mod 4), combined as (s0 + s1) + (s2 + s3): four
  // independent FMA chains instead of one of length D -- with one wave per SIMD nothing else hides
  // the FMA latency.  PotReg below uses the same partition and order: bit-identical results.
  __device__ __forceinline__ void operator()(P2<T> z, P2<T>& g, T& pe) const {
    T s0[4] = {T(0), T(0), T(0), T(0)}, s1[4] = {T(0), T(0), T(0), T(0)};
    const int d0 = D < 64 ? D : 64;
#pragma unroll 4
    for (int j = 0; j < d0; ++j) {
      const T zj = bcast_lane(z.a, j);
      s0[j & 3] = __builtin_fma(Ls[j * D + lane], zj, s0[j & 3]);
      s1[j & 3] = __builtin_fma(Ls[j * D + lane + 64], zj, s1[j & 3]);
    }
#pragma unroll 4
    for (int j = 64; j < D; ++j) {
      const T zj = bcast_lane(z.b, j - 64);
      s0[j & 3] = __builtin_fma(Ls[j * D + lane], zj, s0[j & 3]);
      s1[j & 3] = __builtin_fma(Ls[j * D + lane + 64], zj, s1[j & 3]);
    }
    const T g0 = (s0[0] + s0[1]) + (s0[2] + s0[3]);
    const T g1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    g.a = lane < D ? g0 : T(0);
    g.b = lane + 64 < D ? g1 : T(0);
    pe = T(0.5) * dot(z, g);
  }
};

// PotReg: f32, the lane's two columns of Lambda live in VGPRs for the whole launch (2*DPAD
// registers): the mat-vec is DPAD x {v_readlane -> SGPR, 2 v_fmac with register operands}, no LDS
// traffic and no load latency on the critical path of the leapfrog.  Same summation order as
// PotLds (j ascending), so both give bit-identical f32 results.
template <int DPAD>
struct PotReg {
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f L[DPAD];          // {Lambda[j][lane], Lambda[j][lane + 64]}: one register pair per row j
  float* zs;            // 128 floats of LDS owned by this wave: z broadcast to every lane
  int D, lane;
  __device__ __forceinline__ void load(const float* __restrict__ Lambda, int D_, int lane_,
                                       float* zs_) {
    D = D_;
    lane = lane_;
    zs = zs_;
#pragma unroll
    for (int j = 0; j < DPAD; ++j) {
      L[j][0] = (j < D && lane < D) ? Lambda[j * D + lane] : 0.0f;
      L[j][1] = (j < D && lane + 64 < D) ? Lambda[j * D + lane + 64] : 0.0f;
    }
  }
  // z goes through LDS once (2 ds_write_b32 per lane) and comes back as wave-uniform 16-byte
  // reads (4 coordinates per ds_read_b128, every lane the same address: a broadcast), and the two
  // columns of a lane advance together in ONE v_pk_fma_f32 per row: DPAD/4 LDS reads + DPAD packed
  // FMAs per mat-vec instead of DPAD x {v_readlane, 2 v_fmac}.  Each component is an ordinary fma,
  // partial sums and their order as in PotLds: bit-identical results.
  __device__ __forceinline__ void operator()(P2<float> z, P2<float>& g, float& pe) const {
    zs[lane] = z.a;
    zs[lane + 64] = z.b;
    // One v_pk_fma_f32 per row advances both columns of the lane: {a[k], b[k]} += L[j] * z_j with z_j
    // BROADCAST from the low or the high half of the register pair the ds_read_b128 delivered
    // (op_sel / op_sel_hi on src1) -- written out: left to itself the compiler forms the same packed
    // FMAs but copies every second z_j into a fresh register first (26 v_mov per mat-vec).  Each
    // component is an ordinary fma, partial sums and their order as in PotLds: bit-identical results.
    v2f acc2[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};   // rows j mod 4 (as PotLds)
    const float4* z4 = reinterpret_cast<const float4*>(zs);
#pragma unroll
    for (int j4 = 0; j4 < DPAD / 4; ++j4) {
      const float4 q = z4[j4];
      const v2f zlo = {q.x, q.y}, zhi = {q.z, q.w};
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[0]) : "v"(L[4 * j4 + 0]), "v"(zlo));
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
          : "+v"(acc2[1]) : "v"(L[4 * j4 + 1]), "v"(zlo));
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[2]) : "v"(L[4 * j4 + 2]), "v"(zhi));
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
          : "+v"(acc2[3]) : "v"(L[4 * j4 + 3]), "v"(zhi));
    }
    const float a[4] = {acc2[0][0], acc2[1][0], acc2[2][0], acc2[3][0]};
    const float b[4] = {acc2[0][1], acc2[1][1], acc2[2][1], acc2[3][1]};
    const float acc[2] = {(a[0] + a[1]) + (a[2] + a[3]), (b[0] + b[1]) + (b[2] + b[3])};
    g.a = acc[0];
    g.b = acc[1];
    pe = 0.5f * dot(z, g);
  }
};

template <typename T>
__device__ __forceinline__ bool is_turning(P2<T> r_first, P2<T> r_last, P2<T> r_sum) {
  // nuts.py:184-195 (symmetric in left/right)
  P2<T> rho{r_sum.a - (r_first.a + r_last.a) / T(2), r_sum.b - (r_first.b + r_last.b) / T(2)};
  const T a1 = dot(r_first, rho), a2 = dot(r_last, rho);
  return (a1 <= T(0)) || (a2 <= T(0));
}

template <typename T>
struct Edge {
  P2<T> z, r, ru, g;
};

template <typename T>
struct TransitionStats {
  T accept_prob;
  int n_leapfrog, depth, diverged, accepted;
};

// One NUTS transition of one chain (nuts.py:367-522), state (zc, pe_c, gc) updated in place.
template <typename T, typename Pot>
__device__ __forceinline__ void nuts_transition(const Pot& potential, P2<T>& zc, T& pe_c, P2<T>& gc,
                                                P2<T> v, P2<T> sq, P2<T> isq, T eps, bool va,
                                                bool vb, int lane, int max_depth, int multinomial,
                                                uint64_t seed, uint64_t t, uint64_t cid, T* stk,
                                                T* stk_s, TransitionStats<T>& out) {
  const uint64_t ctr_base = t << 20;
  // ---- momentum: r_unscaled ~ N(0, I), r = M^{1/2} r_unscaled (hmc.py:231-248) -----------
  P2<T> ru0;
  ru0.a = va ? philox_normal_t<T>(seed, ctr_base, (uint64_t)lane, cid) : T(0);
  ru0.b = vb ? philox_normal_t<T>(seed, ctr_base, (uint64_t)(lane + 64), cid) : T(0);
  P2<T> r0{ru0.a * isq.a, ru0.b * isq.b};
  const T energy_current = T(0.5) * dot(ru0, ru0) + pe_c;  // nuts.py:380

  T log_slice;
  if (multinomial) {
    log_slice = -energy_current;
  } else {
    const u32x4 b = philox4x32_10(seed, ctr_base + 1024, cid);
    const T u = uni(uniform_from<T>(b, 0));
    log_slice = -energy_current - (-Num<T>::log_(u));  // Exponential(1) draw, nuts.py:403-410
  }

  Edge<T> EL{zc, r0, ru0, gc}, ER{zc, r0, ru0, gc};
  P2<T> r_sum = ru0;
  T sum_accept = T(0);
  int num_prop = 0;
  T tree_weight = multinomial ? T(0) : T(1);
  int accepted = 0, diverged = 0;
  int tree_depth = 0;
  bool moved = false;

  while (tree_depth < max_depth) {
    const int j = tree_depth;
    const u32x4 bj = philox4x32_10(seed, ctr_base + 1025 + (uint64_t)j, cid);
    const int dir = uni(uniform_from<T>(bj, 0)) < T(0.5) ? 1 : -1;  // Bernoulli(0.5), nuts.py:429
    const T eps_d = dir == 1 ? eps : -eps;

    // cursor starts from the edge leaf in the direction of travel
    P2<T> zq, rq, ruq, gq;
    if (dir == 1) { zq = ER.z; rq = ER.r; gq = ER.g; ruq = ER.ru; }
    else          { zq = EL.z; rq = EL.r; gq = EL.g; ruq = EL.ru; }

    // the subtree under construction
    P2<T> c_first = ruq, c_sum = ruq, c_prop = zq;
    T c_w = T(0), c_pe = T(0);
    bool turning = false, diverging = false;
    T new_sum_accept = T(0);
    int new_num_prop = 0;

    const int nleaf = 1 << j;
    for (int i = 0; i < nleaf; ++i) {
      // ---- one leapfrog step (integrator.py:45-65) ----------------------------------------
      const T hk = T(0.5) * eps_d;
      rq.a = rq.a + hk * (-gq.a);
      rq.b = rq.b + hk * (-gq.b);
      zq.a = zq.a + eps_d * (v.a * rq.a);
      zq.b = zq.b + eps_d * (v.b * rq.b);
      T pe_q;
      potential(zq, gq, pe_q);
      rq.a = rq.a + hk * (-gq.a);
      rq.b = rq.b + hk * (-gq.b);
      // ---- base tree (nuts.py:197-248) ----------------------------------------------------
      ruq = P2<T>{rq.a * sq.a, rq.b * sq.b};
      T energy_new = pe_q + T(0.5) * dot(ruq, ruq);
      if (energy_new != energy_new) energy_new = Num<T>::inf();
      const T sliced = energy_new + log_slice;
      const bool leaf_div = sliced > T(1000);
      const T delta = energy_new - energy_current;
      T ap = Num<T>::exp_(-delta);
      ap = ap > T(1) ? T(1) : ap;
      new_sum_accept += ap;
      new_num_prop += 1;
      P2<T> b_first = ruq, b_sum = ruq, b_prop = zq;
      T b_w = multinomial ? -sliced : (sliced <= T(0) ? T(1) : T(0));
      T b_pe = pe_q;
      if (leaf_div) { diverging = true; c_first = b_first; c_sum = b_sum; c_prop = b_prop;
                      c_w = b_w; c_pe = b_pe; break; }
      // ---- merge with pending left siblings (nuts.py:285-342) -----------------------------
      int k = 0;
      while ((i >> k) & 1) {
        T* e = stk + k * 3 * 128;
        const P2<T> h_first{e[lane], e[lane + 64]};
        const P2<T> h_sum{e[128 + lane], e[128 + lane + 64]};
        const P2<T> h_prop{e[256 + lane], e[256 + lane + 64]};
        const T h_w = stk_s[2 * k], h_pe = stk_s[2 * k + 1];
        T w, prob_other;
        if (multinomial) {
          w = logaddexp_ref(h_w, b_w);
          prob_other = Num<T>::exp_(b_w - w);
        } else {
          w = h_w + b_w;
          prob_other = w > T(0) ? b_w / w : T(0);
        }
        const P2<T> s = h_sum + b_sum;
        const uint64_t id = ((uint64_t)1 << (j - (k + 1))) + (uint64_t)(i >> (k + 1));
        const u32x4 bm =
            philox4x32_10(seed, ctr_base + 2048 + ((uint64_t)1 << j) + id, cid);
        const bool is_other = uni(uniform_from<T>(bm, 0)) < prob_other;
        if (!is_other) { b_prop = h_prop; b_pe = h_pe; }
        b_first = h_first;
        b_sum = s;
        b_w = w;
        ++k;
        // r_last of the merged subtree is the current leaf's unscaled momentum
        if (is_turning(b_first, ruq, b_sum)) { turning = true; break; }
      }
      c_first = b_first; c_sum = b_sum; c_prop = b_prop; c_w = b_w; c_pe = b_pe;
      if (turning) break;
      if (i + 1 < nleaf) {  // park the finished level-k subtree until its sibling is built
        T* e = stk + k * 3 * 128;
        e[lane] = b_first.a; e[lane + 64] = b_first.b;
        e[128 + lane] = b_sum.a; e[128 + lane + 64] = b_sum.b;
        e[256 + lane] = b_prop.a; e[256 + lane + 64] = b_prop.b;
        if (lane == 0) { stk_s[2 * k] = b_w; stk_s[2 * k + 1] = b_pe; }
      }
    }

    // leaf bookkeeping of the outer loop (nuts.py:436-468)
    if (dir == 1) { ER.z = zq; ER.r = rq; ER.ru = ruq; ER.g = gq; }
    else          { EL.z = zq; EL.r = rq; EL.ru = ruq; EL.g = gq; }
    sum_accept += new_sum_accept;
    num_prop += new_num_prop;
    if (diverging) { diverged = 1; break; }
    if (turning) break;
    tree_depth += 1;

    T new_tree_prob;
    if (multinomial) new_tree_prob = Num<T>::exp_(c_w - tree_weight);
    else new_tree_prob = c_w / tree_weight;
    const T rnd = uni(uniform_from<T>(bj, 1));
    if (rnd < new_tree_prob) {  // nuts.py:482-492
      accepted = 1;
      moved = true;
      zc = c_prop;
      pe_c = c_pe;
    }
    r_sum = r_sum + c_sum;
    if (is_turning(EL.ru, ER.ru, r_sum)) break;
    if (multinomial) tree_weight = logaddexp_ref(tree_weight, c_w);
    else tree_weight = tree_weight + c_w;
  }

  // gradient at the returned position (the reference caches it with the proposal; it is a
  // deterministic function of z, recomputed here instead of carrying D more words per subtree)
  if (moved) {
    T pe_chk;
    potential(zc, gc, pe_chk);
  }
  out.accept_prob = sum_accept / (T)num_prop;  // nuts.py:510
  out.n_leapfrog = num_prop;
  out.depth = tree_depth;
  out.diverged = diverged;
  out.accepted = accepted;
}

// Optional per-chain adaptation executed between the transitions of one launch
// (WarmupAdapter.step without its window-end branch, adaptation.py:166-185):
//   da[chain*5 + {0: x_avg, 1: g_avg, 2: t, 3: prox_center, 4: x_t}]  dual averaging of log step
//   wf[chain*2*D + {0..D-1: mean, D..2D-1: m2}]                        Welford of z, count wf_n0 + k
struct RunArgs {
  int64_t num_transitions;
  double target_accept;      // dual averaging statistic H = target - accept_prob
  double da_t0, da_kappa, da_gamma;
  int64_t wf_n0;             // Welford samples seen before this launch
  int64_t mean_n0;           // transitions already averaged into mean_accept
  int count_accepts;         // sampling phase: accumulate accepted / record divergences
};

template <typename T, int WPB, typename Pot, bool LDS_LAMBDA>
__device__ __forceinline__ void nuts_run_body(
    Pot& potential, T* stk, T* stk_s, int chain, int lane, T* __restrict__ z_io,
    T* __restrict__ pe_io, T* __restrict__ grad_io, const T* __restrict__ inv_mass,
    T* __restrict__ step, int C, int D, int max_depth, int multinomial, uint64_t seed, uint64_t t0,
    uint64_t chain_offset, RunArgs ra, T* __restrict__ da, T* __restrict__ wf,
    T* __restrict__ samples, T* __restrict__ mean_accept, int64_t* __restrict__ counters,
    int8_t* __restrict__ div_flags, T* __restrict__ accept_prob_out,
    int32_t* __restrict__ nleap_out, int32_t* __restrict__ depth_out,
    int32_t* __restrict__ div_out, int32_t* __restrict__ acc_out) {
  const bool va = lane < D, vb = lane + 64 < D;
  const int64_t row = (int64_t)chain * D;
  P2<T> zc{va ? z_io[row + lane] : T(0), vb ? z_io[row + lane + 64] : T(0)};
  P2<T> gc{va ? grad_io[row + lane] : T(0), vb ? grad_io[row + lane + 64] : T(0)};
  T pe_c = pe_io[chain];
  // diagonal inverse mass v; mass_matrix_sqrt_inverse = sqrt(v); mass_matrix_sqrt = 1/sqrt(v)
  const P2<T> v{va ? inv_mass[row + lane] : T(1), vb ? inv_mass[row + lane + 64] : T(1)};
  const P2<T> sq{Num<T>::sqrt_(v.a), Num<T>::sqrt_(v.b)};
  const P2<T> isq{T(1) / sq.a, T(1) / sq.b};
  T eps = step[chain];
  const uint64_t cid = chain_offset + (uint64_t)chain;

  // adaptation state in registers for the whole launch
  T x_avg = T(0), g_avg = T(0), da_t = T(0), prox = T(0), x_t = T(0);
  if (da) { x_avg = da[chain * 5]; g_avg = da[chain * 5 + 1]; da_t = da[chain * 5 + 2];
            prox = da[chain * 5 + 3]; x_t = da[chain * 5 + 4]; }
  P2<T> w_mean{T(0), T(0)}, w_m2{T(0), T(0)};
  if (wf) {
    const int64_t wrow = (int64_t)chain * 2 * D;
    w_mean = P2<T>{va ? wf[wrow + lane] : T(0), vb ? wf[wrow + lane + 64] : T(0)};
    w_m2 = P2<T>{va ? wf[wrow + D + lane] : T(0), vb ? wf[wrow + D + lane + 64] : T(0)};
  }
  T mean_ap = mean_accept ? mean_accept[chain] : T(0);
  int64_t nleap_tot = 0, depth_tot = 0, acc_tot = 0;
  TransitionStats<T> st{T(0), 0, 0, 0, 0};

  for (int64_t k = 0; k < ra.num_transitions; ++k) {
    nuts_transition<T>(potential, zc, pe_c, gc, v, sq, isq, eps, va, vb, lane, max_depth,
                       multinomial, seed, t0 + (uint64_t)k, cid, stk, stk_s, st);
    nleap_tot += st.n_leapfrog;
    depth_tot += st.depth;
    T ap = st.accept_prob;
    if (ap != ap) ap = T(0);   // NaN acceptance counts as 0 (as exp(-inf) would)
    mean_ap += (ap - mean_ap) / (T)(ra.mean_n0 + k + 1);
    if (ra.count_accepts) {
      acc_tot += st.accepted;
      if (div_flags && lane == 0) div_flags[k * (int64_t)C + chain] = (int8_t)st.diverged;
    }
    if (da) {  // DualAveraging.step (pyro/ops/dual_averaging.py:55-78) on H = target - ap
      const T g = (T)ra.target_accept - ap;
      da_t += T(1);
      g_avg = (T(1) - T(1) / (da_t + (T)ra.da_t0)) * g_avg + g / (da_t + (T)ra.da_t0);
      x_t = prox - Num<T>::sqrt_(da_t) / (T)ra.da_gamma * g_avg;
      const T weight = Num<T>::exp_(-(T)ra.da_kappa * Num<T>::log_(da_t));
      x_avg = (T(1) - weight) * x_avg + weight * x_t;
      eps = Num<T>::exp_(x_t);
    }
    if (wf) {  // WelfordCovariance.update, diagonal (pyro/ops/welford.py:27-38)
      const T n = (T)(ra.wf_n0 + k + 1);
      const P2<T> pre{zc.a - w_mean.a, zc.b - w_mean.b};
      w_mean = P2<T>{w_mean.a + pre.a / n, w_mean.b + pre.b / n};
      w_m2 = P2<T>{w_m2.a + pre.a * (zc.a - w_mean.a), w_m2.b + pre.b * (zc.b - w_mean.b)};
    }
    if (samples) {
      T* srow = samples + (k * (int64_t)C + chain) * D;
      if (va) srow[lane] = zc.a;
      if (vb) srow[lane + 64] = zc.b;
    }
  }

  if (va) { z_io[row + lane] = zc.a; grad_io[row + lane] = gc.a; }
  if (vb) { z_io[row + lane + 64] = zc.b; grad_io[row + lane + 64] = gc.b; }
  if (wf) {
    const int64_t wrow = (int64_t)chain * 2 * D;
    if (va) { wf[wrow + lane] = w_mean.a; wf[wrow + D + lane] = w_m2.a; }
    if (vb) { wf[wrow + lane + 64] = w_mean.b; wf[wrow + D + lane + 64] = w_m2.b; }
  }
  if (lane == 0) {
    pe_io[chain] = pe_c;
    if (da) { da[chain * 5] = x_avg; da[chain * 5 + 1] = g_avg; da[chain * 5 + 2] = da_t;
              da[chain * 5 + 4] = x_t; step[chain] = eps; }
    if (mean_accept) mean_accept[chain] = mean_ap;
    if (counters) {
      counters[chain] += nleap_tot;
      counters[(int64_t)C + chain] += depth_tot;
      counters[2 * (int64_t)C + chain] += acc_tot;
    }
    accept_prob_out[chain] = st.accept_prob;
    nleap_out[chain] = st.n_leapfrog;
    depth_out[chain] = st.depth;
    div_out[chain] = st.diverged;
    acc_out[chain] = st.accepted;
  }
}

#define PA_NUTS_RUN_PARAMS                                                                        \
  T *__restrict__ z_io, T *__restrict__ pe_io, T *__restrict__ grad_io,                           \
      const T *__restrict__ Lambda, const T *__restrict__ inv_mass, T *__restrict__ step, int C,  \
      int D, int max_depth, int multinomial, uint64_t seed, uint64_t t0, uint64_t chain_offset,   \
      RunArgs ra, T *__restrict__ da, T *__restrict__ wf, T *__restrict__ samples,                \
      T *__restrict__ mean_accept, int64_t *__restrict__ counters, int8_t *__restrict__ div_flags, \
      T *__restrict__ accept_prob_out, int32_t *__restrict__ nleap_out,                           \
      int32_t *__restrict__ depth_out, int32_t *__restrict__ div_out, int32_t *__restrict__ acc_out
#define PA_NUTS_RUN_ARGS                                                                           \
  z_io, pe_io, grad_io, inv_mass, step, C, D, max_depth, multinomial, seed, t0, chain_offset, ra,  \
      da, wf, samples, mean_accept, counters, div_flags, accept_prob_out, nleap_out, depth_out,    \
      div_out, acc_out

// Lambda in LDS: WPB waves (= chains) per workgroup share one copy.
template <typename T, int WPB>
__global__ __launch_bounds__(64 * WPB) void nuts_gaussian_kernel(PA_NUTS_RUN_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ls = reinterpret_cast<T*>(smem_raw);  // [D*D + 128]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ls_elems = D * D + 128;
  // wave-private stack: [NUTS_MAX_DEPTH][3][128] vectors then [NUTS_MAX_DEPTH][2] scalars
  T* stk = Ls + ls_elems + wave * (NUTS_MAX_DEPTH * (3 * 128 + 2));
  T* stk_s = stk + NUTS_MAX_DEPTH * 3 * 128;
  for (int i = threadIdx.x; i < ls_elems; i += 64 * WPB) Ls[i] = i < D * D ? Lambda[i] : T(0);
  __syncthreads();
  const int chain = blockIdx.x * WPB + wave;
  if (chain >= C) return;  // wave-uniform; no block-level sync after this point
  PotLds<T> pot{Ls, D, lane};
  nuts_run_body<T, WPB, PotLds<T>, true>(pot, stk, stk_s, chain, lane, PA_NUTS_RUN_ARGS);
}

// Lambda columns in VGPRs (f32): one wave = one chain = one workgroup, up to 512 VGPRs per lane.
template <int DPAD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void nuts_gaussian_reg_kernel(float* __restrict__ z_io, float* __restrict__ pe_io,
                              float* __restrict__ grad_io, const float* __restrict__ Lambda,
                              const float* __restrict__ inv_mass, float* __restrict__ step, int C,
                              int D, int max_depth, int multinomial, uint64_t seed, uint64_t t0,
                              uint64_t chain_offset, RunArgs ra, float* __restrict__ da,
                              float* __restrict__ wf, float* __restrict__ samples,
                              float* __restrict__ mean_accept, int64_t* __restrict__ counters,
                              int8_t* __restrict__ div_flags, float* __restrict__ accept_prob_out,
                              int32_t* __restrict__ nleap_out, int32_t* __restrict__ depth_out,
                              int32_t* __restrict__ div_out, int32_t* __restrict__ acc_out) {
  using T = float;
  __shared__ float stack_s[NUTS_MAX_DEPTH * (3 * 128 + 2)];
  const int lane = threadIdx.x;
  const int chain = blockIdx.x;
  __shared__ __attribute__((aligned(16))) float zb_s[128];
  PotReg<DPAD> pot;
  pot.load(Lambda, D, lane, zb_s);
  nuts_run_body<float, 1, PotReg<DPAD>, false>(pot, stack_s, stack_s + NUTS_MAX_DEPTH * 3 * 128,
                                               chain, lane, PA_NUTS_RUN_ARGS);
}

// ---- reasonable step size per chain (reference: hmc.py:170-229; Stan's heuristic) ----------------
// One wave = one chain.  A trial = fresh momentum, ONE leapfrog with the candidate step size, the
// change of the Hamiltonian; the first trial fixes the direction (double while the one-step
// acceptance stays above 0.8, else halve), then the step is scaled until the direction flips or
// the step leaves (min, max).  The host version ran the same loop for all chains in lock step with
// ~15 launches and one host synchronisation per trial; here every chain runs its own loop in one
// launch.  Lambda is read from global memory (L2): a few dozen mat-vecs per chain, once per window.
template <typename T>
__global__ __launch_bounds__(64) void nuts_gaussian_find_step_kernel(
    const T* __restrict__ z_in, const T* __restrict__ pe_in, const T* __restrict__ grad_in,
    const T* __restrict__ Lambda, const T* __restrict__ inv_mass, T* __restrict__ step, int C,
    int D, uint64_t seed, uint64_t key, uint64_t chain_offset, T min_step, T max_step,
    T direction_threshold) {
  const int lane = threadIdx.x, chain = blockIdx.x;
  const bool va = lane < D, vb = lane + 64 < D;
  const int64_t row = (int64_t)chain * D;
  const P2<T> z{va ? z_in[row + lane] : T(0), vb ? z_in[row + lane + 64] : T(0)};
  const P2<T> g{va ? grad_in[row + lane] : T(0), vb ? grad_in[row + lane + 64] : T(0)};
  const T pe = pe_in[chain];
  const P2<T> v{va ? inv_mass[row + lane] : T(1), vb ? inv_mass[row + lane + 64] : T(1)};
  const P2<T> sq{Num<T>::sqrt_(v.a), Num<T>::sqrt_(v.b)};
  const P2<T> isq{T(1) / sq.a, T(1) / sq.b};
  const uint64_t cid = chain_offset + (uint64_t)chain;
  PotLds<T> pot{Lambda, D, lane};
  T eps = step[chain];

  auto trial = [&](T e, uint64_t it) -> int {
    const uint64_t ctr = (key + it) << 20;
    P2<T> ru{va ? philox_normal_t<T>(seed, ctr, (uint64_t)lane, cid) : T(0),
             vb ? philox_normal_t<T>(seed, ctr, (uint64_t)(lane + 64), cid) : T(0)};
    P2<T> r{ru.a * isq.a, ru.b * isq.b};
    const T e0 = T(0.5) * dot(ru, ru) + pe;
    const T hk = T(0.5) * e;
    r.a = r.a + hk * (-g.a);
    r.b = r.b + hk * (-g.b);
    P2<T> z1{z.a + e * (v.a * r.a), z.b + e * (v.b * r.b)}, g1;
    T pe1;
    pot(z1, g1, pe1);
    r.a = r.a + hk * (-g1.a);
    r.b = r.b + hk * (-g1.b);
    const P2<T> ru1{r.a * sq.a, r.b * sq.b};
    const T delta = (T(0.5) * dot(ru1, ru1) + pe1) - e0;
    return (direction_threshold < -delta) ? 1 : -1;        // NaN compares false: direction -1
  };

  const int direction = trial(eps, 0);
  const T scale = direction > 0 ? T(2) : T(0.5);
  for (uint64_t it = 1; it <= 200; ++it) {
    if (!(eps > min_step && eps < max_step)) break;
    eps = eps * scale;
    if (trial(eps, it) != direction) break;
  }
  eps = eps < min_step ? min_step : (eps > max_step ? max_step : eps);
  if (lane == 0) step[chain] = eps;
}

template <typename T>
struct RunPtrs {
  T *z, *pe, *grad;
  const T* Lambda;
  const T* inv_mass;
  T* step;
  T *da, *wf, *samples, *mean_accept;
  int64_t* counters;
  int8_t* div_flags;
  T* ap;
  int32_t *nl, *dp, *dv, *ac;
};

template <typename T, int WPB>
static int nuts_launch_lds(const RunPtrs<T>& p, int C, int D, int max_depth, int multinomial,
                           uint64_t seed, uint64_t t0, uint64_t chain_offset, const RunArgs& ra,
                           hipStream_t s) {
  const size_t lds = ((size_t)D * D + 128 + (size_t)WPB * NUTS_MAX_DEPTH * (3 * 128 + 2)) *
                     sizeof(T);
  if (lds > 160 * 1024)
    return fail(PA_ERR_UNSUPPORTED, "nuts_gaussian: needs %zu B of LDS (> 160 KiB)", lds);
  auto k = nuts_gaussian_kernel<T, WPB>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess)
      return fail(PA_ERR_LAUNCH, "nuts_gaussian: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int grid = (C + WPB - 1) / WPB;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WPB), lds, s, p.z, p.pe, p.grad, p.Lambda,
                     p.inv_mass, p.step, C, D, max_depth, multinomial, seed, t0, chain_offset, ra,
                     p.da, p.wf, p.samples, p.mean_accept, p.counters, p.div_flags, p.ap, p.nl,
                     p.dp, p.dv, p.ac);
  return PA_OK;
}

template <int DPAD>
static void nuts_launch_reg(const RunPtrs<float>& p, int C, int D, int max_depth, int multinomial,
                            uint64_t seed, uint64_t t0, uint64_t chain_offset, const RunArgs& ra,
                            hipStream_t s) {
  hipLaunchKernelGGL((nuts_gaussian_reg_kernel<DPAD>), dim3(C), dim3(64), 0, s, p.z, p.pe, p.grad,
                     p.Lambda, p.inv_mass, p.step, C, D, max_depth, multinomial, seed, t0,
                     chain_offset, ra, p.da, p.wf, p.samples, p.mean_accept, p.counters,
                     p.div_flags, p.ap, p.nl, p.dp, p.dv, p.ac);
}

static int g_nuts_force_lds = 0;  // test hook: pa_nuts_gaussian_set_variant

template <typename T>
static int nuts_dispatch(const RunPtrs<T>& p, int C, int D, int max_depth, int multinomial,
                         uint64_t seed, uint64_t t0, uint64_t chain_offset, const RunArgs& ra,
                         hipStream_t s) {
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_NUTS, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  int rc = PA_OK;
  if constexpr (sizeof(T) == 4) {
    if (!g_nuts_force_lds) {
      if (D <= 32) nuts_launch_reg<32>(p, C, D, max_depth, multinomial, seed, t0, chain_offset, ra, s);
      else if (D <= 64) nuts_launch_reg<64>(p, C, D, max_depth, multinomial, seed, t0, chain_offset, ra, s);
      else if (D <= 96) nuts_launch_reg<96>(p, C, D, max_depth, multinomial, seed, t0, chain_offset, ra, s);
      else if (D <= 100) nuts_launch_reg<100>(p, C, D, max_depth, multinomial, seed, t0, chain_offset, ra, s);
      else if (D <= 104) nuts_launch_reg<104>(p, C, D, max_depth, multinomial, seed, t0, chain_offset, ra, s);
      else nuts_launch_reg<128>(p, C, D, max_depth, multinomial, seed, t0, chain_offset, ra, s);
    } else {
      rc = nuts_launch_lds<T, 4>(p, C, D, max_depth, multinomial, seed, t0, chain_offset, ra, s);
    }
  } else {
    const size_t lds2 = ((size_t)D * D + 128 + 2 * (size_t)NUTS_MAX_DEPTH * (3 * 128 + 2)) * 8;
    if (lds2 <= 160 * 1024)
      rc = nuts_launch_lds<T, 2>(p, C, D, max_depth, multinomial, seed, t0, chain_offset, ra, s);
    else
      rc = nuts_launch_lds<T, 1>(p, C, D, max_depth, multinomial, seed, t0, chain_offset, ra, s);
  }
  if (br) (void)hipEventRecord(ev1, s);
  if (rc != PA_OK) return rc;
  return check_launch("nuts_gaussian_kernel");
}

}  // namespace pa

extern "C" {

int pa_nuts_gaussian_set_variant(int force_lds) {
  pa::g_nuts_force_lds = force_lds ? 1 : 0;
  return PA_OK;
}

static int nuts_common_checks(int dtype, int64_t C, int64_t D, int max_tree_depth, uint64_t t) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "nuts_gaussian: bad dtype %d", dtype);
  PA_REQUIRE(C >= 0 && D >= 1, "nuts_gaussian: bad shape C=%lld D=%lld", (long long)C,
             (long long)D);
  if (D > pa::NUTS_DMAX)
    return pa::fail(PA_ERR_UNSUPPORTED, "nuts_gaussian: D=%lld > %d", (long long)D, pa::NUTS_DMAX);
  PA_REQUIRE(max_tree_depth >= 1 && max_tree_depth <= pa::NUTS_MAX_DEPTH,
             "nuts_gaussian: max_tree_depth must be in [1,%d]", pa::NUTS_MAX_DEPTH);
  PA_REQUIRE(C < (1 << 30) && t < ((uint64_t)1 << 43), "nuts_gaussian: C or t too large");
  return PA_OK;
}

int pa_nuts_gaussian_run(int dtype, void* z, void* pe, void* grad, const void* Lambda,
                         const void* inv_mass, void* step, int64_t C, int64_t D,
                         int max_tree_depth, int use_multinomial, uint64_t seed, uint64_t t0,
                         int64_t num_transitions, uint64_t chain_offset, void* da_state,
                         double target_accept, void* welford, int64_t welford_n0, void* samples,
                         void* mean_accept, int64_t mean_n0, int64_t* counters, int count_accepts,
                         int8_t* div_flags, void* accept_prob, int32_t* n_leapfrog, int32_t* depth,
                         int32_t* diverging, int32_t* accepted, pa_stream_t stream) {
  int rc = nuts_common_checks(dtype, C, D, max_tree_depth, t0 + (uint64_t)num_transitions);
  if (rc != PA_OK) return rc;
  PA_REQUIRE(num_transitions >= 0, "nuts_gaussian_run: negative num_transitions");
  if (C == 0 || num_transitions == 0) return PA_OK;
  PA_REQUIRE(z && pe && grad && Lambda && inv_mass && step && accept_prob && n_leapfrog && depth &&
                 diverging && accepted,
             "nuts_gaussian: NULL pointer");
  pa::RunArgs ra;
  ra.num_transitions = num_transitions;
  ra.target_accept = target_accept;
  ra.da_t0 = 10.0; ra.da_kappa = 0.75; ra.da_gamma = 0.05;   // DualAveraging defaults
  ra.wf_n0 = welford_n0;
  ra.mean_n0 = mean_n0;
  ra.count_accepts = count_accepts;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32) {
    pa::RunPtrs<float> p{(float*)z, (float*)pe, (float*)grad, (const float*)Lambda,
                         (const float*)inv_mass, (float*)step, (float*)da_state, (float*)welford,
                         (float*)samples, (float*)mean_accept, counters, div_flags,
                         (float*)accept_prob, n_leapfrog, depth, diverging, accepted};
    return pa::nuts_dispatch<float>(p, (int)C, (int)D, max_tree_depth, use_multinomial, seed, t0,
                                    chain_offset, ra, s);
  }
  pa::RunPtrs<double> p{(double*)z, (double*)pe, (double*)grad, (const double*)Lambda,
                        (const double*)inv_mass, (double*)step, (double*)da_state, (double*)welford,
                        (double*)samples, (double*)mean_accept, counters, div_flags,
                        (double*)accept_prob, n_leapfrog, depth, diverging, accepted};
  return pa::nuts_dispatch<double>(p, (int)C, (int)D, max_tree_depth, use_multinomial, seed, t0,
                                   chain_offset, ra, s);
}

int pa_nuts_gaussian_transition(int dtype, void* z, void* pe, void* grad, const void* Lambda,
                                const void* inv_mass, const void* step, int64_t C, int64_t D,
                                int max_tree_depth, int use_multinomial, uint64_t seed, uint64_t t,
                                uint64_t chain_offset, void* accept_prob, int32_t* n_leapfrog,
                                int32_t* depth, int32_t* diverging, int32_t* accepted,
                                pa_stream_t stream) {
  // one transition, no adaptation, no sample buffer: `step` is only read
  return pa_nuts_gaussian_run(dtype, z, pe, grad, Lambda, inv_mass, const_cast<void*>(step), C, D,
                              max_tree_depth, use_multinomial, seed, t, 1, chain_offset, nullptr,
                              0.8, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr,
                              accept_prob, n_leapfrog, depth, diverging, accepted, stream);
}

int pa_nuts_gaussian_find_step(int dtype, const void* z, const void* pe, const void* grad,
                               const void* Lambda, const void* inv_mass, void* step, int64_t C,
                               int64_t D, uint64_t seed, uint64_t key, uint64_t chain_offset,
                               double min_step, double max_step, double direction_threshold,
                               pa_stream_t stream) {
  int rc = nuts_common_checks(dtype, C, D, 1, 0);
  if (rc != PA_OK) return rc;
  PA_REQUIRE(key < ((uint64_t)1 << 43), "nuts_gaussian_find_step: key too large");
  if (C == 0) return PA_OK;
  PA_REQUIRE(z && pe && grad && Lambda && inv_mass && step, "nuts_gaussian_find_step: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::nuts_gaussian_find_step_kernel<float>), dim3((unsigned)C), dim3(64), 0, s,
                       (const float*)z, (const float*)pe, (const float*)grad, (const float*)Lambda,
                       (const float*)inv_mass, (float*)step, (int)C, (int)D, seed, key, chain_offset,
                       (float)min_step, (float)max_step, (float)direction_threshold);
  else
    hipLaunchKernelGGL((pa::nuts_gaussian_find_step_kernel<double>), dim3((unsigned)C), dim3(64), 0,
                       s, (const double*)z, (const double*)pe, (const double*)grad,
                       (const double*)Lambda, (const double*)inv_mass, (double*)step, (int)C, (int)D,
                       seed, key, chain_offset, min_step, max_step, direction_threshold);
  return pa::check_launch("nuts_gaussian_find_step_kernel");
}

}  // extern "C"
