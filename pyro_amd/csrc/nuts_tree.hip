// nuts_tree.hip -- NUTS for ARBITRARY potentials, vectorised over chains: the tree logic of
// pyro/infer/mcmc/nuts.py as a device-resident per-chain state machine.
//
// Reference semantics restated (per chain), same citations as nuts.hip:
//   sample() :367-522, _build_tree() :250-365 (here iterative: the bits of the leaf index are
//   the merge schedule), _build_basetree() :197-248, _is_turning() :184-195, _logaddexp :15-17,
//   leapfrog pyro/ops/integrator.py:45-65, momentum hmc.py:231-248 with a diagonal mass matrix
//   (BlockMassMatrix.scale/unscale/kinetic_grad, adaptation.py:328-392).
//
// Division of labour.  The potential energy and its gradient are whatever the caller computes
// for ALL chains at once at the cursor positions zq[C,D] (a fused likelihood kernel such as
// pa_glm_bernoulli_fwd_bwd with P = C, or torch autograd of a chain-batched model); between two
// such evaluations ONE launch of nuts_tree_advance_kernel does, for every chain independently,
//   second half-kick -> leaf energies / divergence / accept prob -> merges with the parked
//   sibling subtrees (U-turn checks, multinomial or slice proposal choice) -> doubling
//   bookkeeping (accept draw, U-turn of the whole tree, next direction) -> first half-kick and
//   drift of the chain's NEXT leapfrog,
// so the host loop per leapfrog is {potential(zq) ; advance} with no host decision inside.
// A chain that has finished its transition is inactive: its cursor no longer moves and its
// (z, pe, grad) hold the accepted state; the host polls n_active.
//
// One workgroup (NW waves) per chain; thread t owns coordinates t, t+NT, ... (NPL of them, in
// VGPRs), every [C,D] array is read/written fully coalesced, reductions are wave butterflies
// (+ LDS across waves when NW > 1) in a fixed order => deterministic.  The pending-subtree
// stack lives in HBM/L2 (4 vectors + 2 scalars per level per chain).
//
// Randomness: the SAME keyed Philox contract as nuts.hip (so both kernels, and the recursive
// oracle, draw identical numbers): counter_lo = t*2^20 + slot, counter_hi = global chain id.
#include "nuts_common.h"
#include "dist_fam.h"

namespace pa {

template <typename T, int NPL> struct Vec { T x[NPL]; };

enum { FS_ENERGY = 0, FS_LOGSLICE = 1, FS_TREEW = 2, FS_SUMACC = 3, FS_COUNT = 4 };
enum { IS_ACTIVE = 0, IS_DIR = 1, IS_DEPTH = 2, IS_LEAF = 3, IS_NPROP = 4, IS_ACCEPTED = 5,
       IS_DIVERGED = 6, IS_COUNT = 8 };

// workspace layout (element counts of T unless noted)
template <typename T>
struct TreeWs {
  T* edges;     // [2][3][C*D]  left{z,r,g}, right{z,r,g}
  T* r_sum;     // [C*D]
  T* stack;     // [max_depth][4][C*D]  first_ru, sum_ru, prop_z, prop_g
  T* stack_s;   // [max_depth][2][C]    weight, prop_pe
  T* fscal;     // [FS_COUNT][C]
  int32_t* iscal;  // [IS_COUNT][C]
};
static size_t tree_ws_elems(int64_t C, int64_t D, int max_depth) {
  return (size_t)(6 * C * D + C * D + (int64_t)max_depth * 4 * C * D + (int64_t)max_depth * 2 * C +
                  FS_COUNT * C);
}
template <typename T>
static TreeWs<T> tree_ws(void* base, int64_t C, int64_t D, int max_depth) {
  TreeWs<T> w;
  T* p = (T*)base;
  w.edges = p; p += 6 * C * D;
  w.r_sum = p; p += C * D;
  w.stack = p; p += (int64_t)max_depth * 4 * C * D;
  w.stack_s = p; p += (int64_t)max_depth * 2 * C;
  w.fscal = p; p += FS_COUNT * C;
  w.iscal = (int32_t*)p;
  return w;
}

// Where slot s, coordinate d of the cursor buffer handed to the potential lives.  n_sites == 0: row-major
// [n_slots, D] (a potential that takes the flat state).  Otherwise SITE-MAJOR: the sites of the flat layout
// (ascending offsets covering [0, D)) as contiguous blocks [n_slots, len_s] one after the other, so that a
// kernel that takes one site as its operand (the GLM kernel: weights [P, D_w], bias [P]) reads it in place.
constexpr int TREE_MAX_SITES = 8;
struct SlotLayout {
  int n_sites;
  int off[TREE_MAX_SITES], len[TREE_MAX_SITES];
};
__device__ __forceinline__ int64_t slot_index(const SlotLayout& L, int64_t n_slots, int64_t slot, int D, int d) {
  if (L.n_sites == 0) return slot * D + d;
  int s = 0;
#pragma unroll
  for (int k = 1; k < TREE_MAX_SITES; ++k)
    if (k < L.n_sites && d >= L.off[k]) s = k;
  return n_slots * L.off[s] + slot * L.len[s] + (d - L.off[s]);
}

template <typename T, int NW, int NPL>
struct Chain {
  static constexpr int NT = 64 * NW;
  int tid, D;
  int64_t row;  // chain * D
  T* red;       // LDS [2*NW]
  __device__ __forceinline__ bool ok(int m) const { return tid + m * NT < D; }
  __device__ __forceinline__ Vec<T, NPL> ld(const T* p) const {
    Vec<T, NPL> v;
#pragma unroll
    for (int m = 0; m < NPL; ++m) v.x[m] = ok(m) ? p[row + tid + m * NT] : T(0);
    return v;
  }
  __device__ __forceinline__ void st(T* p, const Vec<T, NPL>& v) const {
#pragma unroll
    for (int m = 0; m < NPL; ++m)
      if (ok(m)) p[row + tid + m * NT] = v.x[m];
  }
  // the same at another row of a [*, D] array (a SLOT of a compacted round, see TreeRun)
  __device__ __forceinline__ Vec<T, NPL> ld_at(const T* p, int64_t r) const {
    Vec<T, NPL> v;
#pragma unroll
    for (int m = 0; m < NPL; ++m) v.x[m] = ok(m) ? p[r + tid + m * NT] : T(0);
    return v;
  }
  __device__ __forceinline__ void st_at(T* p, int64_t r, const Vec<T, NPL>& v) const {
#pragma unroll
    for (int m = 0; m < NPL; ++m)
      if (ok(m)) p[r + tid + m * NT] = v.x[m];
  }
  // slot `slot` of a cursor buffer laid out by `L`
  __device__ __forceinline__ void st_slot(T* p, const SlotLayout& L, int64_t n_slots, int64_t slot,
                                          const Vec<T, NPL>& v) const {
#pragma unroll
    for (int m = 0; m < NPL; ++m)
      if (ok(m)) p[slot_index(L, n_slots, slot, D, tid + m * NT)] = v.x[m];
  }
  // block-wide sums of two per-thread values, identical in every thread, fixed order
  __device__ __forceinline__ void sum2(T& a, T& b) const {
    a = wave_sum(a);
    b = wave_sum(b);
    if constexpr (NW == 1) {
      a = uni(a);
      b = uni(b);
    } else {
      const int lane = tid & 63, w = tid >> 6;
      __syncthreads();
      if (lane == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
      __syncthreads();
      T ta = T(0), tb = T(0);
#pragma unroll
      for (int i = 0; i < NW; ++i) { ta += red[2 * i]; tb += red[2 * i + 1]; }
      a = ta;
      b = tb;
    }
  }
  __device__ __forceinline__ T dot(const Vec<T, NPL>& a, const Vec<T, NPL>& b) const {
    T s = T(0), z = T(0);
#pragma unroll
    for (int m = 0; m < NPL; ++m) s += a.x[m] * b.x[m];
    sum2(s, z);
    return s;
  }
  // nuts.py:184-195 (symmetric in first/last)
  __device__ __forceinline__ bool is_turning(const Vec<T, NPL>& r_first, const Vec<T, NPL>& r_last,
                                             const Vec<T, NPL>& r_sum) const {
    T a1 = T(0), a2 = T(0);
#pragma unroll
    for (int m = 0; m < NPL; ++m) {
      const T rho = r_sum.x[m] - (r_first.x[m] + r_last.x[m]) / T(2);
      a1 += r_first.x[m] * rho;
      a2 += r_last.x[m] * rho;
    }
    sum2(a1, a2);
    return (a1 <= T(0)) || (a2 <= T(0));
  }
};

template <typename T, int NPL>
__device__ __forceinline__ Vec<T, NPL> vmul(const Vec<T, NPL>& a, const Vec<T, NPL>& b) {
  Vec<T, NPL> c;
#pragma unroll
  for (int m = 0; m < NPL; ++m) c.x[m] = a.x[m] * b.x[m];
  return c;
}
template <typename T, int NPL>
__device__ __forceinline__ Vec<T, NPL> vadd(const Vec<T, NPL>& a, const Vec<T, NPL>& b) {
  Vec<T, NPL> c;
#pragma unroll
  for (int m = 0; m < NPL; ++m) c.x[m] = a.x[m] + b.x[m];
  return c;
}

// first half of a leapfrog on the cursor: r <- r - 0.5 eps g ; z <- z + eps (v . r)
template <typename T, int NPL>
__device__ __forceinline__ void kick_drift(Vec<T, NPL>& z, Vec<T, NPL>& r, const Vec<T, NPL>& g,
                                           const Vec<T, NPL>& v, T eps_d) {
  const T hk = T(0.5) * eps_d;
#pragma unroll
  for (int m = 0; m < NPL; ++m) {
    r.x[m] = r.x[m] + hk * (-g.x[m]);
    z.x[m] = z.x[m] + eps_d * (v.x[m] * r.x[m]);
  }
}

// ---- asynchronous chains (pa_nuts_tree_run_*) ------------------------------------------------
// A SPAN is K transitions per chain with fixed adaptation bookkeeping (no warm-up window ends
// inside).  Within a span a chain that finishes its transition does the per-transition half of
// the adaptation itself (dual averaging of its step size, Welford update of its mass estimate,
// running acceptance mean, counters, sample store -- what HMC._after_transition /
// WarmupAdapter.step do on the host between the reference's transitions, adaptation.py:166-185)
// and BEGINS ITS NEXT TRANSITION in the same launch: no chain waits for the slowest tree of a
// transition, every potential evaluation serves C live cursors, and the host only counts
// finished chains.  The Philox keys are (transition index, slot, chain), so a chain draws the
// same numbers as under the lock-step schedule: same chains, bit for bit.
//
// The span's parameters live in DEVICE memory (ctl, int64 words) so that one captured hipGraph
// of rounds serves every span of a run.
enum { RC_T0 = 0, RC_K = 1, RC_MEAN_N0 = 2, RC_WF_N0 = 3, RC_FLAGS = 4, RC_SAMPLES = 5,
       RC_DIV = 6, RC_ROW0 = 7, RC_COUNT = 8 };
enum { RF_ADAPT_STEP = 1, RF_WELFORD = 2, RF_COUNT_ACCEPTS = 4 };

// The potential of a FLAT model assembled inside the tree kernel (pa_nuts_tree_run_advance_direct): every
// latent site is scored by a fused family at parameters that do not depend on other latents (the reference
// evaluates them site by site, pyro/infer/mcmc/util.py:264-286 -> trace.log_prob_sum), through the identity
// or the exp transform of its support; one or more observed sites were evaluated outside (the GLM kernel)
// and hand over their log-likelihood per slot and its gradient per site.
//   U(z)  = -( ll_ext[slot] + sum_sites sum_j [ log p_s(v_j) + log|dv_j/du_j| ] ),   v = T_s(u)
//   dU/du = -( (d log p_s/dv + g_ext) dv/du + d log|dv/du| / du )
// A HIERARCHICAL prior: a parameter of site s may be the (constrained) VALUE of another latent site q of the same
// chain -- w ~ Normal(mu, tau) with mu, tau latent (pyro/infer/mcmc/util.py:264-286 scores it through the handlers
// and autograd carries d log p_s / d parameter back to q).  Encoded without a new argument: p == NULL and stride
// s = -(q + 1); element j of s reads element j % len_q of q (a parent broadcast over leading plate dims).  The
// gradient d log p_s[j] / d parameter is added to q's coordinates through LDS, in a fixed order.
struct DirectSite {
  int dist, transform;            // PA_DIST_*; 0: v = u, 1: v = lower + exp(u)
  const void *p0, *p1;            // the family's parameters, element j of the site at p[j * stride]
  int64_t s0, s1;                 // (p == NULL, s < 0: the value of site -s - 1, see above)
  const void* g_ext;              // [n_slots, len]: d ll_ext / d v of this site, or NULL
  double lower;
};
struct TreeDirect {
  int n_sites;                    // 0: (peq, gq) come from the caller
  const void* ll_ext;             // [n_slots] or NULL
  DirectSite s[TREE_MAX_SITES];
};

template <typename T>
struct TreeRun {
  const int64_t* ctl;     // [RC_COUNT]
  T* step;                // [C], rewritten by the dual averaging
  T* da;                  // [C,5] {x_avg, g_avg, t, prox_center, x_t}      (nuts.hip RunArgs)
  T* wf;                  // [C,2,D] {mean, m2}
  T* mean_accept;         // [C]
  int64_t* counters;      // [3,C] leapfrogs, depth, accepted
  int32_t* tc;            // [C] transitions this chain has completed in the span
  int32_t* n_done;        // chains that completed the span
  int64_t* done_flag;     // set to 1 by the last chain to complete (a step gate's abort word), or NULL
  double target_accept, da_t0, da_kappa, da_gamma;
  // COMPACTED rounds (late in a span few chains are still building trees): the potential is evaluated at
  // n_slots < C cursor rows zq_slot[n_slots, D]; slot s belongs to chain slot2chain[s] (-1: empty).  The
  // launch has one workgroup per slot, (peq, gq) are slot-indexed, the chain's next cursor is written to its
  // slot row as well as to its own row.  NULL: slot == chain (the full round).
  const int32_t* slot2chain;
  T* zq_slot;
  SlotLayout lay;         // layout of zq_slot (row-major when lay.n_sites == 0)
  int64_t n_slots;
};

// tree state of a chain at the start of transition t (nuts.py:367-434): momentum draw, energies,
// both edges at the current state, first direction, first half leapfrog into (zq, rq)
template <typename T, int NW, int NPL>
__device__ __forceinline__ void tree_begin_chain(
    const Chain<T, NW, NPL>& c, int chain, int64_t C, const Vec<T, NPL>& zc, const Vec<T, NPL>& gc,
    T pe_c, const Vec<T, NPL>& v, T eps, int multinomial, uint64_t seed, uint64_t t, uint64_t cid,
    const TreeWs<T>& ws, T* __restrict__ zq, T* __restrict__ rq, T* __restrict__ zq_slot = nullptr,
    int64_t slot = 0, const SlotLayout* lay = nullptr, int64_t n_slots = 0) {
  const int64_t CD = C * c.D;
  const uint64_t ctr_base = t << 20;
  Vec<T, NPL> isq, ru0;
#pragma unroll
  for (int m = 0; m < NPL; ++m) {
    const int d = c.tid + m * c.NT;
    isq.x[m] = T(1) / Num<T>::sqrt_(v.x[m]);  // mass_matrix_sqrt (adaptation.py:270-282)
    ru0.x[m] = c.ok(m) ? philox_normal_t<T>(seed, ctr_base, (uint64_t)d, cid) : T(0);
  }
  Vec<T, NPL> r0 = vmul(ru0, isq);                              // scale(), adaptation.py:349-373
  const T energy_current = T(0.5) * c.dot(ru0, ru0) + pe_c;     // nuts.py:380
  T log_slice;
  if (multinomial) {
    log_slice = -energy_current;
  } else {
    const u32x4 b = philox4x32_10(seed, ctr_base + 1024, cid);
    log_slice = -energy_current - (-Num<T>::log_(uniform_from<T>(b, 0)));  // nuts.py:403-410
  }
  // both edges start at the current state
  for (int e = 0; e < 2; ++e) {
    c.st(ws.edges + (e * 3 + 0) * CD, zc);
    c.st(ws.edges + (e * 3 + 1) * CD, r0);
    c.st(ws.edges + (e * 3 + 2) * CD, gc);
  }
  c.st(ws.r_sum, ru0);
  // first doubling: direction draw, cursor = edge, first half leapfrog
  const u32x4 bj = philox4x32_10(seed, ctr_base + 1025, cid);
  const int dir = uniform_from<T>(bj, 0) < T(0.5) ? 1 : -1;
  Vec<T, NPL> zn = zc, rn = r0;
  kick_drift(zn, rn, gc, v, dir == 1 ? eps : -eps);
  c.st(zq, zn);
  c.st(rq, rn);
  if (zq_slot != nullptr) c.st_slot(zq_slot, *lay, n_slots, slot, zn);
  if (threadIdx.x == 0) {
    ws.fscal[FS_ENERGY * C + chain] = energy_current;
    ws.fscal[FS_LOGSLICE * C + chain] = log_slice;
    ws.fscal[FS_TREEW * C + chain] = multinomial ? T(0) : T(1);
    ws.fscal[FS_SUMACC * C + chain] = T(0);
    ws.iscal[IS_ACTIVE * C + chain] = 1;
    ws.iscal[IS_DIR * C + chain] = dir;
    ws.iscal[IS_DEPTH * C + chain] = 0;
    ws.iscal[IS_LEAF * C + chain] = 0;
    ws.iscal[IS_NPROP * C + chain] = 0;
    ws.iscal[IS_ACCEPTED * C + chain] = 0;
    ws.iscal[IS_DIVERGED * C + chain] = 0;
  }
}

// t_ctl == nullptr: the lock-step protocol (transition index t); else the first transition of a
// span (index ctl[RC_T0]) and the span's per-chain / global counters are reset
template <typename T, int NW, int NPL>
__global__ __launch_bounds__(64 * NW) void nuts_tree_begin_kernel(
    const T* __restrict__ z, const T* __restrict__ pe, const T* __restrict__ grad,
    T* __restrict__ zq, T* __restrict__ rq, const T* __restrict__ inv_mass, int64_t im_stride,
    const T* __restrict__ step, int64_t C, int D, int multinomial, uint64_t seed, uint64_t t,
    uint64_t chain_offset, TreeWs<T> ws, const int64_t* __restrict__ t_ctl,
    int32_t* __restrict__ tc, int32_t* __restrict__ n_done, int64_t* __restrict__ done_flag) {
  __shared__ T red[2 * NW];
  const int chain = blockIdx.x;
  Chain<T, NW, NPL> c{(int)threadIdx.x, D, (int64_t)chain * D, red};
  if (t_ctl != nullptr) {
    t = (uint64_t)t_ctl[RC_T0];
    if (threadIdx.x == 0) {
      tc[chain] = 0;
      if (chain == 0) {
        *n_done = 0;
        if (done_flag != nullptr) *done_flag = 0;
      }
    }
  }
  Vec<T, NPL> zc = c.ld(z), gc = c.ld(grad), v;
#pragma unroll
  for (int m = 0; m < NPL; ++m)
    v.x[m] = c.ok(m) ? inv_mass[(int64_t)chain * im_stride + c.tid + m * c.NT] : T(1);
  tree_begin_chain<T, NW, NPL>(c, chain, C, zc, gc, pe[chain], v, step[chain], multinomial, seed, t,
                               chain_offset + (uint64_t)chain, ws, zq, rq);
}

template <int DIST, typename T>
__device__ __forceinline__ T elem_site_lp(T v, T a, T b, T& dv, T& da, T& db) {
  Fam<DIST, T>::grad(v, a, b, dv, da, db);
  return Fam<DIST, T>::lp(v, a, b);
}
// the families a continuous latent site of a flat model can have (real or positive support)
#define PA_TREE_FAMILY(DIST_ID, CALL)                                                      \
  switch (DIST_ID) {                                                                       \
    case PA_DIST_NORMAL: { constexpr int D_ = PA_DIST_NORMAL; CALL; } break;               \
    case PA_DIST_HALF_CAUCHY: { constexpr int D_ = PA_DIST_HALF_CAUCHY; CALL; } break;     \
    case PA_DIST_LOG_NORMAL: { constexpr int D_ = PA_DIST_LOG_NORMAL; CALL; } break;       \
    case PA_DIST_EXPONENTIAL: { constexpr int D_ = PA_DIST_EXPONENTIAL; CALL; } break;     \
    case PA_DIST_HALF_NORMAL: { constexpr int D_ = PA_DIST_HALF_NORMAL; CALL; } break;     \
    case PA_DIST_GAMMA: { constexpr int D_ = PA_DIST_GAMMA; CALL; } break;                 \
    default: break;                                                                        \
  }

// (pe, gradient) of the flat model at this thread's coordinates of the cursor, see TreeDirect.  `ucoord(d)` = the
// unconstrained value of coordinate d of THIS chain's cursor (any coordinate: parents are read through it);
// par_lds = 2 x TREE_DIRECT_MAXD values of LDS (used only when some site has a parent-valued parameter).
constexpr int TREE_DIRECT_MAXD = 512;
template <typename T, int NW, int NPL, typename UCoord>
__device__ __forceinline__ void direct_potential(const Chain<T, NW, NPL>& c, const TreeDirect& dp,
                                                 const SlotLayout& lay, int64_t slot, const Vec<T, NPL>& zq,
                                                 Vec<T, NPL>& gq, T& pe_q, UCoord ucoord, T* par_lds) {
  T lp_sum = T(0), zero = T(0);
  bool any_parent = false;
#pragma unroll
  for (int k = 0; k < TREE_MAX_SITES; ++k)
    if (k < lay.n_sites && ((dp.s[k].p0 == nullptr && dp.s[k].s0 < 0) || (dp.s[k].p1 == nullptr && dp.s[k].s1 < 0)))
      any_parent = true;
  // the constrained value of element i of site q of this chain
  auto parent_value = [&](int q, int i) -> T {
    const T u = ucoord(lay.off[q] + i);
    return dp.s[q].transform == 1 ? (T)dp.s[q].lower + Num<T>::exp_(u) : u;
  };
  T dvdu_own[NPL];
#pragma unroll
  for (int m = 0; m < NPL; ++m) {
    gq.x[m] = T(0);
    dvdu_own[m] = T(1);
    const int d = c.tid + m * c.NT;
    if (!c.ok(m)) continue;
    int si = 0;
#pragma unroll
    for (int k = 1; k < TREE_MAX_SITES; ++k)
      if (k < lay.n_sites && d >= lay.off[k]) si = k;
    const DirectSite& st = dp.s[si];
    const int j = d - lay.off[si];
    const T u = zq.x[m];
    const int q0 = (st.p0 == nullptr && st.s0 < 0) ? (int)(-st.s0 - 1) : -1;
    const int q1 = (st.p1 == nullptr && st.s1 < 0) ? (int)(-st.s1 - 1) : -1;
    const T a = q0 >= 0 ? parent_value(q0, j % lay.len[q0]) : (st.p0 != nullptr ? ((const T*)st.p0)[j * st.s0] : T(0));
    const T b = q1 >= 0 ? parent_value(q1, j % lay.len[q1]) : (st.p1 != nullptr ? ((const T*)st.p1)[j * st.s1] : T(0));
    T v = u, dvdu = T(1), ladj = T(0), dladj = T(0);
    if (st.transform == 1) {                 // support (lower, inf): biject_to = exp then shift
      dvdu = Num<T>::exp_(u);
      v = (T)st.lower + dvdu;
      ladj = u;
      dladj = T(1);
    }
    dvdu_own[m] = dvdu;
    T lp = T(0), dv = T(0), da = T(0), db = T(0);
    PA_TREE_FAMILY(st.dist, (lp = elem_site_lp<D_, T>(v, a, b, dv, da, db)));
    const T ge = st.g_ext != nullptr ? ((const T*)st.g_ext)[slot * lay.len[si] + j] : T(0);
    gq.x[m] = -((dv + ge) * dvdu + dladj);
    lp_sum += lp + ladj;
    if (any_parent) {
      par_lds[d] = da;
      par_lds[TREE_DIRECT_MAXD + d] = db;
    }
  }
  if (any_parent) {
    // d U / d v_q[i] = - sum over the sites s that take q as a parameter, over their elements j = i (mod len_q),
    // of d log p_s[j] / d that parameter -- summed in increasing (s, j): the same bits on every run
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NPL; ++m) {
      const int d = c.tid + m * c.NT;
      if (!c.ok(m)) continue;
      int q = 0;
#pragma unroll
      for (int k = 1; k < TREE_MAX_SITES; ++k)
        if (k < lay.n_sites && d >= lay.off[k]) q = k;
      const int i = d - lay.off[q], lq = lay.len[q];
      T acc = T(0);
      for (int k = 0; k < lay.n_sites; ++k) {
        const DirectSite& st = dp.s[k];
        const bool t0 = st.p0 == nullptr && st.s0 == -(int64_t)(q + 1);
        const bool t1 = st.p1 == nullptr && st.s1 == -(int64_t)(q + 1);
        if (!t0 && !t1) continue;
        for (int j = i; j < lay.len[k]; j += lq) {
          if (t0) acc += par_lds[lay.off[k] + j];
          if (t1) acc += par_lds[TREE_DIRECT_MAXD + lay.off[k] + j];
        }
      }
      gq.x[m] -= acc * dvdu_own[m];
    }
    __syncthreads();
  }
  c.sum2(lp_sum, zero);
  pe_q = -(dp.ll_ext != nullptr ? ((const T*)dp.ll_ext)[slot] : T(0)) - lp_sum;
}

// the same arithmetic on its own (pa_nuts_direct_potential): (U, dU/du) at n_slots cursors of a site-major pack,
// one workgroup per slot -- what the tree kernel computes in registers for its kick, written out
template <int NW, int NPL>
__global__ __launch_bounds__(64 * NW) void nuts_direct_potential_kernel(
    const float* __restrict__ zq_pack, int64_t n_slots, int D, SlotLayout lay, TreeDirect direct,
    float* __restrict__ pe_out, float* __restrict__ grad_out) {
  __shared__ float red[2 * NW];
  const int64_t slot = blockIdx.x;
  Chain<float, NW, NPL> c{(int)threadIdx.x, D, slot * D, red};
  Vec<float, NPL> zq, gq;
#pragma unroll
  for (int m = 0; m < NPL; ++m)
    zq.x[m] = c.ok(m) ? zq_pack[slot_index(lay, n_slots, slot, D, c.tid + m * c.NT)] : 0.0f;
  float pe_q;
  __shared__ float par_lds[2 * TREE_DIRECT_MAXD];
  auto ucoord = [&](int d) -> float { return zq_pack[slot_index(lay, n_slots, slot, D, d)]; };
  direct_potential<float, NW, NPL>(c, direct, lay, slot, zq, gq, pe_q, ucoord, par_lds);
  c.st(grad_out, gq);
  if (c.tid == 0) pe_out[slot] = pe_q;
}

template <typename T, int NW, int NPL, bool RUN, bool DIRECT = false>
__global__ __launch_bounds__(64 * NW) void nuts_tree_advance_kernel(
    T* __restrict__ z_io, T* __restrict__ pe_io, T* __restrict__ grad_io, T* __restrict__ zq_io,
    T* __restrict__ rq_io, const T* __restrict__ gq_in, const T* __restrict__ peq_in,
    const T* __restrict__ inv_mass, int64_t im_stride, const T* step, int64_t C,
    int D, int max_depth, int multinomial, uint64_t seed, uint64_t t,
    const uint64_t* __restrict__ t_dev, uint64_t chain_offset,
    TreeWs<T> ws, T* __restrict__ accept_prob_out, int32_t* __restrict__ nleap_out,
    int32_t* __restrict__ depth_out, int32_t* __restrict__ div_out, int32_t* __restrict__ acc_out,
    int32_t* __restrict__ n_active, TreeRun<T> run, TreeDirect direct = TreeDirect{}) {
  __shared__ T red[2 * NW];
  const int slot = blockIdx.x;
  int chain = slot;
  if constexpr (RUN) {
    if (run.slot2chain != nullptr) {
      chain = run.slot2chain[slot];
      if (chain < 0) return;                         // an empty slot of a compacted round
    }
  }
  if (ws.iscal[IS_ACTIVE * C + chain] == 0) return;  // block-uniform
  Chain<T, NW, NPL> c{(int)threadIdx.x, D, (int64_t)chain * D, red};
  const int64_t slot_row = (int64_t)slot * D;
  T* const zq_slot = RUN ? run.zq_slot : nullptr;
  const int64_t CD = C * D;
  // the transition index keys the Philox draws; a launch that is replayed from a hipGraph reads it
  // from device memory (pa_nuts_tree_advance_tdev) instead of its (captured) argument; a chain of
  // a span is at its own transition ctl[RC_T0] + tc[chain]
  int span_k = 0;
  uint64_t tt = t_dev != nullptr ? *t_dev : t;
  if constexpr (RUN) {
    span_k = run.tc[chain];
    tt = (uint64_t)run.ctl[RC_T0] + (uint64_t)span_k;
  }
  const uint64_t ctr_base = tt << 20, cid = chain_offset + (uint64_t)chain;

  const int dir = ws.iscal[IS_DIR * C + chain];
  int tree_depth = ws.iscal[IS_DEPTH * C + chain];
  const int i = ws.iscal[IS_LEAF * C + chain];
  int num_prop = ws.iscal[IS_NPROP * C + chain];
  const T energy_current = ws.fscal[FS_ENERGY * C + chain];
  const T log_slice = ws.fscal[FS_LOGSLICE * C + chain];
  T tree_weight = ws.fscal[FS_TREEW * C + chain];
  T sum_accept = ws.fscal[FS_SUMACC * C + chain];
  const T eps = step[chain];
  const T eps_d = dir == 1 ? eps : -eps;
  const int j = tree_depth;

  Vec<T, NPL> v, sq;
#pragma unroll
  for (int m = 0; m < NPL; ++m) {
    v.x[m] = c.ok(m) ? inv_mass[(int64_t)chain * im_stride + c.tid + m * c.NT] : T(1);
    sq.x[m] = Num<T>::sqrt_(v.x[m]);  // mass_matrix_sqrt_inverse
  }
  Vec<T, NPL> zq = c.ld(zq_io), rq = c.ld(rq_io), gq;
  T pe_q;
  if constexpr (DIRECT) {
    // (the chain's cursor row in the row-major [C, D] buffer: a parent-valued parameter reads other coordinates)
    __shared__ T par_lds[2 * TREE_DIRECT_MAXD];
    const T* zq_row = zq_io + c.row;
    auto ucoord = [&](int d) -> T { return zq_row[d]; };
    direct_potential<T, NW, NPL>(c, direct, run.lay, slot, zq, gq, pe_q, ucoord, par_lds);
  } else {
    gq = c.ld_at(gq_in, slot_row);
    pe_q = peq_in[slot];
  }

  // ---- second half-kick (integrator.py:62-63) and the base tree (nuts.py:197-248) ----------
  {
    const T hk = T(0.5) * eps_d;
#pragma unroll
    for (int m = 0; m < NPL; ++m) rq.x[m] = rq.x[m] + hk * (-gq.x[m]);
  }
  const Vec<T, NPL> ruq = vmul(rq, sq);
  T energy_new = pe_q + T(0.5) * c.dot(ruq, ruq);
  if (energy_new != energy_new) energy_new = Num<T>::inf();
  const T sliced = energy_new + log_slice;
  const bool leaf_div = sliced > T(1000);
  T ap = Num<T>::exp_(-(energy_new - energy_current));
  ap = ap > T(1) ? T(1) : ap;
  sum_accept += ap;
  num_prop += 1;

  Vec<T, NPL> b_first = ruq, b_sum = ruq, b_prop = zq, b_propg = gq;
  T b_w = multinomial ? -sliced : (sliced <= T(0) ? T(1) : T(0));
  T b_pe = pe_q;
  bool turning = false;
  bool finished = false;
  bool moved = false;      // this round's doubling accepted its proposal: (b_prop, b_propg, b_pe)
  int diverged = 0;
  int accepted = ws.iscal[IS_ACCEPTED * C + chain];

  if (leaf_div) {
    diverged = 1;
    finished = true;
  } else {
    // ---- merge with the parked left siblings (nuts.py:285-342) -----------------------------
    int k = 0;
    while ((i >> k) & 1) {
      const T* e = ws.stack + (int64_t)k * 4 * CD;
      const Vec<T, NPL> h_first = c.ld(e), h_sum = c.ld(e + CD);
      const T h_w = ws.stack_s[(2 * k) * C + chain], h_pe = ws.stack_s[(2 * k + 1) * C + chain];
      T w, prob_other;
      if (multinomial) {
        w = logaddexp_ref(h_w, b_w);
        prob_other = Num<T>::exp_(b_w - w);
      } else {
        w = h_w + b_w;
        prob_other = w > T(0) ? b_w / w : T(0);
      }
      const uint64_t id = ((uint64_t)1 << (j - (k + 1))) + (uint64_t)(i >> (k + 1));
      const u32x4 bm = philox4x32_10(seed, ctr_base + 2048 + ((uint64_t)1 << j) + id, cid);
      const bool is_other = uniform_from<T>(bm, 0) < prob_other;
      if (!is_other) {
        b_prop = c.ld(e + 2 * CD);
        b_propg = c.ld(e + 3 * CD);
        b_pe = h_pe;
      }
      b_first = h_first;
      b_sum = vadd(h_sum, b_sum);
      b_w = w;
      ++k;
      if (c.is_turning(b_first, ruq, b_sum)) { turning = true; break; }
    }
    if (turning) {
      finished = true;
    } else if (i + 1 < (1 << j)) {
      // park the finished level-k subtree until its right sibling is built; go on leaping
      T* e = ws.stack + (int64_t)k * 4 * CD;
      c.st(e, b_first);
      c.st(e + CD, b_sum);
      c.st(e + 2 * CD, b_prop);
      c.st(e + 3 * CD, b_propg);
      if (threadIdx.x == 0) {
        ws.stack_s[(2 * k) * C + chain] = b_w;
        ws.stack_s[(2 * k + 1) * C + chain] = b_pe;
        ws.iscal[IS_LEAF * C + chain] = i + 1;
      }
      kick_drift(zq, rq, gq, v, eps_d);
      c.st(zq_io, zq);
      c.st(rq_io, rq);
      if (zq_slot != nullptr) c.st_slot(zq_slot, run.lay, run.n_slots, slot, zq);
    } else {
      // ---- the doubling is complete (nuts.py:436-503) ---------------------------------------
      const int e_dir = dir == 1 ? 1 : 0;
      c.st(ws.edges + (e_dir * 3 + 0) * CD, zq);
      c.st(ws.edges + (e_dir * 3 + 1) * CD, rq);
      c.st(ws.edges + (e_dir * 3 + 2) * CD, gq);
      tree_depth += 1;
      const T new_tree_prob = multinomial ? Num<T>::exp_(b_w - tree_weight) : b_w / tree_weight;
      const u32x4 bj = philox4x32_10(seed, ctr_base + 1025 + (uint64_t)j, cid);
      if (uniform_from<T>(bj, 1) < new_tree_prob) {  // nuts.py:482-492
        accepted = 1;
        moved = true;
        c.st(z_io, b_prop);
        c.st(grad_io, b_propg);
        if (threadIdx.x == 0) pe_io[chain] = b_pe;
      }
      Vec<T, NPL> r_sum = vadd(c.ld(ws.r_sum), b_sum);
      const Vec<T, NPL> ru_other = vmul(c.ld(ws.edges + ((1 - e_dir) * 3 + 1) * CD), sq);
      if (c.is_turning(ru_other, ruq, r_sum)) {
        finished = true;
      } else {
        tree_weight = multinomial ? logaddexp_ref(tree_weight, b_w) : tree_weight + b_w;
        if (tree_depth >= max_depth) {
          finished = true;
        } else {
          // next doubling: direction draw, cursor = the edge in that direction
          c.st(ws.r_sum, r_sum);
          const u32x4 bn = philox4x32_10(seed, ctr_base + 1025 + (uint64_t)tree_depth, cid);
          const int ndir = uniform_from<T>(bn, 0) < T(0.5) ? 1 : -1;
          const int ne = ndir == 1 ? 1 : 0;
          Vec<T, NPL> zn, rn, gn;
          if (ne == e_dir) { zn = zq; rn = rq; gn = gq; }
          else {
            zn = c.ld(ws.edges + (ne * 3 + 0) * CD);
            rn = c.ld(ws.edges + (ne * 3 + 1) * CD);
            gn = c.ld(ws.edges + (ne * 3 + 2) * CD);
          }
          kick_drift(zn, rn, gn, v, ndir == 1 ? eps : -eps);
          c.st(zq_io, zn);
          c.st(rq_io, rn);
          if (zq_slot != nullptr) c.st_slot(zq_slot, run.lay, run.n_slots, slot, zn);
          if (threadIdx.x == 0) {
            ws.iscal[IS_DIR * C + chain] = ndir;
            ws.iscal[IS_LEAF * C + chain] = 0;
          }
        }
      }
    }
  }

  if constexpr (RUN) {
    if (finished) {
      // ---- the transition is over: bookkeeping of HMC._after_transition / WarmupAdapter.step ---
      const int64_t flags = run.ctl[RC_FLAGS];
      const int64_t K = run.ctl[RC_K];
      const T ap_raw = sum_accept / (T)num_prop;            // nuts.py:510
      T ap = ap_raw;
      if (ap != ap) ap = T(0);   // NaN acceptance counts as 0 (as exp(-inf) would)
      const Vec<T, NPL> zc = moved ? b_prop : c.ld(z_io);
      const Vec<T, NPL> gc = moved ? b_propg : c.ld(grad_io);
      const T pe_c = moved ? b_pe : pe_io[chain];
      T eps_next = eps;
      if (flags & RF_ADAPT_STEP) {  // DualAveraging.step (pyro/ops/dual_averaging.py:55-78), H = target - ap
        T x_avg = run.da[chain * 5], g_avg = run.da[chain * 5 + 1], da_t = run.da[chain * 5 + 2];
        const T prox = run.da[chain * 5 + 3];
        const T g = (T)run.target_accept - ap;
        da_t += T(1);
        g_avg = (T(1) - T(1) / (da_t + (T)run.da_t0)) * g_avg + g / (da_t + (T)run.da_t0);
        const T x_t = prox - Num<T>::sqrt_(da_t) / (T)run.da_gamma * g_avg;
        const T weight = Num<T>::exp_(-(T)run.da_kappa * Num<T>::log_(da_t));
        x_avg = (T(1) - weight) * x_avg + weight * x_t;
        eps_next = Num<T>::exp_(x_t);
        if constexpr (NW > 1) __syncthreads();    // every thread has read the record
        if (threadIdx.x == 0) {
          run.da[chain * 5] = x_avg; run.da[chain * 5 + 1] = g_avg; run.da[chain * 5 + 2] = da_t;
          run.da[chain * 5 + 4] = x_t;
          run.step[chain] = eps_next;
        }
      }
      if (flags & RF_WELFORD) {  // WelfordCovariance.update, diagonal (pyro/ops/welford.py:27-38)
        const T n = (T)(run.ctl[RC_WF_N0] + span_k + 1);
        T* wrow = run.wf + (int64_t)chain * 2 * D;
#pragma unroll
        for (int m = 0; m < NPL; ++m) {
          const int d = c.tid + m * c.NT;
          if (c.ok(m)) {
            T mean = wrow[d], m2 = wrow[D + d];
            const T pre = zc.x[m] - mean;
            mean = mean + pre / n;
            m2 = m2 + pre * (zc.x[m] - mean);
            wrow[d] = mean;
            wrow[D + d] = m2;
          }
        }
      }
      T* samples = (T*)run.ctl[RC_SAMPLES];
      if (samples != nullptr)
        c.st(samples + (run.ctl[RC_ROW0] + span_k) * CD, zc);
      if (threadIdx.x == 0) {
        const T mean_ap = run.mean_accept[chain];
        run.mean_accept[chain] = mean_ap + (ap - mean_ap) / (T)(run.ctl[RC_MEAN_N0] + span_k + 1);
        run.counters[chain] += num_prop;
        run.counters[C + chain] += tree_depth;
        if (flags & RF_COUNT_ACCEPTS) {
          run.counters[2 * C + chain] += accepted;
          int8_t* div = (int8_t*)run.ctl[RC_DIV];
          if (div != nullptr) div[(run.ctl[RC_ROW0] + span_k) * C + chain] = (int8_t)diverged;
        }
        accept_prob_out[chain] = ap_raw;
        nleap_out[chain] = num_prop;
        depth_out[chain] = tree_depth;
        div_out[chain] = diverged;
        acc_out[chain] = accepted;
        run.tc[chain] = span_k + 1;
      }
      if ((int64_t)span_k + 1 < K) {
        // the chain's next transition starts here (the scalars of the finished tree are dead)
        tree_begin_chain<T, NW, NPL>(c, chain, C, zc, gc, pe_c, v, eps_next, multinomial, seed,
                                     tt + 1, cid, ws, zq_io, rq_io, zq_slot, slot, &run.lay, run.n_slots);
      } else if (threadIdx.x == 0) {
        ws.iscal[IS_ACTIVE * C + chain] = 0;
        ws.iscal[IS_DIVERGED * C + chain] = diverged;
        const int32_t old = atomicAdd(run.n_done, 1);
        if (old + 1 == (int32_t)C && run.done_flag != nullptr) *run.done_flag = 1;
      }
      return;
    }
  }
  if (threadIdx.x == 0) {
    ws.iscal[IS_DEPTH * C + chain] = tree_depth;
    ws.iscal[IS_NPROP * C + chain] = num_prop;
    ws.iscal[IS_ACCEPTED * C + chain] = accepted;
    ws.fscal[FS_TREEW * C + chain] = tree_weight;
    ws.fscal[FS_SUMACC * C + chain] = sum_accept;
    if (finished) {
      ws.iscal[IS_ACTIVE * C + chain] = 0;
      ws.iscal[IS_DIVERGED * C + chain] = diverged;
      accept_prob_out[chain] = sum_accept / (T)num_prop;  // nuts.py:510
      nleap_out[chain] = num_prop;
      depth_out[chain] = tree_depth;
      div_out[chain] = diverged;
      acc_out[chain] = accepted;
    } else if constexpr (!RUN) {
      atomicAdd(n_active, 1);
    }
  }
}

// slots of a compacted round: the chains still building a tree, in ascending order, then -1; their cursor
// rows gathered.  One workgroup (the map is a few thousand entries at most; this runs when the host switches
// to a smaller round, a handful of times per span).
template <typename T>
__global__ __launch_bounds__(1024) void nuts_tree_compact_kernel(const int32_t* __restrict__ iscal, int64_t C,
                                                               int D, const T* __restrict__ zq,
                                                               int32_t* __restrict__ slot2chain,
                                                               T* __restrict__ zq_slot, int n_slots,
                                                               int32_t* __restrict__ n_placed, SlotLayout lay,
                                                               int identity) {
  __shared__ int32_t counts[1024];
  __shared__ int32_t total;
  const int tid = threadIdx.x;
  if (identity) {
    // the FULL round in the layout `lay` (slot == chain, every chain placed): the cursor rows re-laid only
    for (int64_t e = tid; e < C * D; e += 1024) {
      const int64_t ch = e / D;
      zq_slot[slot_index(lay, C, ch, D, (int)(e % D))] = zq[e];
    }
    if (tid == 0) *n_placed = (int32_t)C;
    return;
  }
  const int64_t per = (C + 1023) / 1024, lo = tid * per, hi = lo + per < C ? lo + per : C;
  int32_t mine = 0;
  for (int64_t ch = lo; ch < hi; ++ch) mine += iscal[IS_ACTIVE * C + ch] != 0;
  counts[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    int32_t run = 0;
    for (int i = 0; i < 1024; ++i) { const int32_t v = counts[i]; counts[i] = run; run += v; }
    total = run;
    *n_placed = run < n_slots ? run : n_slots;
  }
  __syncthreads();
  int32_t at = counts[tid];
  for (int64_t ch = lo; ch < hi; ++ch)
    if (iscal[IS_ACTIVE * C + ch] != 0) {
      if (at < n_slots) slot2chain[at] = (int32_t)ch;
      ++at;
    }
  for (int s = total + tid; s < n_slots; s += 1024) slot2chain[s] = -1;
  __syncthreads();
  const int filled = total < n_slots ? total : n_slots;
  for (int64_t e = tid; e < (int64_t)filled * D; e += 1024) {
    const int s = (int)(e / D), d = (int)(e % D);
    zq_slot[slot_index(lay, n_slots, s, D, d)] = zq[(int64_t)slot2chain[s] * D + d];
  }
}

struct TreePlan { int nw, npl; };
static bool tree_plan(int64_t D, TreePlan* p) {
  if (D <= 128) { *p = {1, 2}; return true; }
  if (D <= 512) { *p = {1, 8}; return true; }
  if (D <= 2048) { *p = {4, 8}; return true; }
  return false;
}

#define PA_TREE_DISPATCH(T, CALL)                                   \
  do {                                                              \
    if (pl.nw == 1 && pl.npl == 2) { CALL(T, 1, 2); }               \
    else if (pl.nw == 1 && pl.npl == 8) { CALL(T, 1, 8); }          \
    else { CALL(T, 4, 8); }                                         \
  } while (0)

template <typename T>
static int tree_begin(const void* z, const void* pe, const void* grad, void* zq, void* rq,
                      const void* inv_mass, int64_t im_stride, const void* step, int64_t C,
                      int64_t D, int max_depth, int multinomial, uint64_t seed, uint64_t t,
                      uint64_t chain_offset, void* workspace, hipStream_t s,
                      const int64_t* t_ctl = nullptr, int32_t* tc = nullptr,
                      int32_t* n_done = nullptr, int64_t* done_flag = nullptr) {
  TreePlan pl;
  tree_plan(D, &pl);
  TreeWs<T> ws = tree_ws<T>(workspace, C, D, max_depth);
#define PA_CALL(TT, NW, NPL)                                                                     \
  hipLaunchKernelGGL((nuts_tree_begin_kernel<TT, NW, NPL>), dim3((unsigned)C), dim3(64 * NW), 0, \
                     s, (const TT*)z, (const TT*)pe, (const TT*)grad, (TT*)zq, (TT*)rq,          \
                     (const TT*)inv_mass, im_stride, (const TT*)step, C, (int)D, multinomial,    \
                     seed, t, chain_offset, ws, t_ctl, tc, n_done, done_flag)
  PA_TREE_DISPATCH(T, PA_CALL);
#undef PA_CALL
  return check_launch("nuts_tree_begin_kernel");
}

template <typename T>
static int tree_advance(void* z, void* pe, void* grad, void* zq, void* rq, const void* gq,
                        const void* peq, const void* inv_mass, int64_t im_stride, const void* step,
                        int64_t C, int64_t D, int max_depth, int multinomial, uint64_t seed,
                        uint64_t t, const uint64_t* t_dev, uint64_t chain_offset, void* accept_prob,
                        int32_t* nl, int32_t* dp, int32_t* dv, int32_t* ac, int32_t* n_active,
                        void* workspace, hipStream_t s) {
  TreePlan pl;
  tree_plan(D, &pl);
  TreeWs<T> ws = tree_ws<T>(workspace, C, D, max_depth);
  if (hipMemsetAsync(n_active, 0, sizeof(int32_t), s) != hipSuccess)
    return fail(PA_ERR_LAUNCH, "nuts_tree_advance: memset failed");
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_NUTS, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
#define PA_CALL(TT, NW, NPL)                                                                      \
  hipLaunchKernelGGL((nuts_tree_advance_kernel<TT, NW, NPL, false>), dim3((unsigned)C),            \
                     dim3(64 * NW), 0, s, (TT*)z, (TT*)pe, (TT*)grad, (TT*)zq, (TT*)rq,           \
                     (const TT*)gq, (const TT*)peq, (const TT*)inv_mass, im_stride,               \
                     (const TT*)step, C, (int)D, max_depth, multinomial, seed, t, t_dev,          \
                     chain_offset, ws, (TT*)accept_prob, nl, dp, dv, ac, n_active, TreeRun<TT>{})
  PA_TREE_DISPATCH(T, PA_CALL);
#undef PA_CALL
  if (br) (void)hipEventRecord(ev1, s);
  return check_launch("nuts_tree_advance_kernel");
}

template <typename T>
static int tree_run_advance(void* z, void* pe, void* grad, void* zq, void* rq, const void* gq,
                            const void* peq, const void* inv_mass, int64_t im_stride, int64_t C,
                            int64_t D, int max_depth, int multinomial, uint64_t seed,
                            uint64_t chain_offset, TreeRun<T> run, int64_t n_slots, void* accept_prob,
                            int32_t* nl, int32_t* dp, int32_t* dv, int32_t* ac, void* workspace,
                            hipStream_t s, const TreeDirect* direct = nullptr) {
  TreePlan pl;
  tree_plan(D, &pl);
  TreeWs<T> ws = tree_ws<T>(workspace, C, D, max_depth);
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_NUTS, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  if (direct != nullptr) {
    if constexpr (sizeof(T) == 4) {
      // (float32, D <= 512: what the GLM kernels feed)
#define PA_CALLD(NW, NPL)                                                                          \
  hipLaunchKernelGGL((nuts_tree_advance_kernel<float, NW, NPL, true, true>), dim3((unsigned)n_slots), \
                     dim3(64 * NW), 0, s, (float*)z, (float*)pe, (float*)grad, (float*)zq, (float*)rq, \
                     (const float*)nullptr, (const float*)nullptr, (const float*)inv_mass, im_stride, \
                     (const float*)run.step, C, (int)D, max_depth, multinomial, seed, (uint64_t)0,  \
                     (const uint64_t*)nullptr, chain_offset, ws, (float*)accept_prob, nl, dp, dv, ac, \
                     (int32_t*)nullptr, run, *direct)
      if (pl.nw == 1 && pl.npl == 2) { PA_CALLD(1, 2); }
      else if (pl.nw == 1 && pl.npl == 8) { PA_CALLD(1, 8); }
      else return fail(PA_ERR_UNSUPPORTED, "nuts_tree_run_advance_direct: D > 512");
#undef PA_CALLD
      if (br) (void)hipEventRecord(ev1, s);
      return check_launch("nuts_tree_run_advance_direct_kernel");
    } else {
      return fail(PA_ERR_UNSUPPORTED, "nuts_tree_run_advance_direct: float32 only");
    }
  }
#define PA_CALL(TT, NW, NPL)                                                                      \
  hipLaunchKernelGGL((nuts_tree_advance_kernel<TT, NW, NPL, true>), dim3((unsigned)n_slots),       \
                     dim3(64 * NW), 0, s, (TT*)z, (TT*)pe, (TT*)grad, (TT*)zq, (TT*)rq,           \
                     (const TT*)gq, (const TT*)peq, (const TT*)inv_mass, im_stride,               \
                     (const TT*)run.step, C, (int)D, max_depth, multinomial, seed, (uint64_t)0,   \
                     (const uint64_t*)nullptr, chain_offset, ws, (TT*)accept_prob, nl, dp, dv, ac, \
                     (int32_t*)nullptr, run)
  PA_TREE_DISPATCH(T, PA_CALL);
#undef PA_CALL
  if (br) (void)hipEventRecord(ev1, s);
  return check_launch("nuts_tree_run_advance_kernel");
}

}  // namespace pa

extern "C" {

size_t pa_nuts_tree_workspace(int dtype, int64_t C, int64_t D, int max_tree_depth) {
  if (C < 0 || D < 1 || D > 2048 || max_tree_depth < 1 || max_tree_depth > pa::NUTS_MAX_DEPTH)
    return 0;
  const size_t esz = dtype == PA_F64 ? 8 : 4;
  size_t bytes = pa::tree_ws_elems(C, D, max_tree_depth) * esz;
  bytes = (bytes + 15) & ~(size_t)15;
  return bytes + (size_t)pa::IS_COUNT * C * sizeof(int32_t) + 16;
}

#define PA_TREE_COMMON_CHECKS(who)                                                                \
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, who ": bad dtype %d", dtype);                    \
  PA_REQUIRE(C >= 0 && D >= 1, who ": bad shape C=%lld D=%lld", (long long)C, (long long)D);      \
  if (D > 2048)                                                                                   \
    return pa::fail(PA_ERR_UNSUPPORTED, who ": D=%lld > 2048", (long long)D);                     \
  PA_REQUIRE(max_tree_depth >= 1 && max_tree_depth <= pa::NUTS_MAX_DEPTH,                         \
             who ": max_tree_depth must be in [1,%d]", pa::NUTS_MAX_DEPTH);                       \
  PA_REQUIRE(C < (1 << 30) && t < ((uint64_t)1 << 43), who ": C or t too large");                 \
  PA_REQUIRE(im_stride_row == 0 || im_stride_row == D, who ": inv_mass must be [D] or [C,D]");    \
  if (C == 0) return PA_OK;                                                                       \
  PA_REQUIRE(workspace && workspace_bytes >= pa_nuts_tree_workspace(dtype, C, D, max_tree_depth), \
             who ": workspace too small");

int pa_nuts_tree_begin(int dtype, const void* z, const void* pe, const void* grad, void* zq,
                       void* rq, const void* inv_mass, int64_t im_stride_row, const void* step,
                       int64_t C, int64_t D, int max_tree_depth, int use_multinomial,
                       uint64_t seed, uint64_t t, uint64_t chain_offset, void* workspace,
                       size_t workspace_bytes, pa_stream_t stream) {
  PA_TREE_COMMON_CHECKS("nuts_tree_begin")
  PA_REQUIRE(z && pe && grad && zq && rq && inv_mass && step, "nuts_tree_begin: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    return pa::tree_begin<float>(z, pe, grad, zq, rq, inv_mass, im_stride_row, step, C, D,
                                 max_tree_depth, use_multinomial, seed, t, chain_offset, workspace,
                                 s);
  return pa::tree_begin<double>(z, pe, grad, zq, rq, inv_mass, im_stride_row, step, C, D,
                                max_tree_depth, use_multinomial, seed, t, chain_offset, workspace,
                                s);
}

int pa_nuts_tree_advance(int dtype, void* z, void* pe, void* grad, void* zq, void* rq,
                         const void* gq, const void* peq, const void* inv_mass,
                         int64_t im_stride_row, const void* step, int64_t C, int64_t D,
                         int max_tree_depth, int use_multinomial, uint64_t seed, uint64_t t,
                         uint64_t chain_offset, void* accept_prob, int32_t* n_leapfrog,
                         int32_t* depth, int32_t* diverging, int32_t* accepted, int32_t* n_active,
                         void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  PA_TREE_COMMON_CHECKS("nuts_tree_advance")
  PA_REQUIRE(z && pe && grad && zq && rq && gq && peq && inv_mass && step && accept_prob &&
                 n_leapfrog && depth && diverging && accepted && n_active,
             "nuts_tree_advance: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    return pa::tree_advance<float>(z, pe, grad, zq, rq, gq, peq, inv_mass, im_stride_row, step, C,
                                   D, max_tree_depth, use_multinomial, seed, t, nullptr,
                                   chain_offset, accept_prob, n_leapfrog, depth, diverging,
                                   accepted, n_active, workspace, s);
  return pa::tree_advance<double>(z, pe, grad, zq, rq, gq, peq, inv_mass, im_stride_row, step, C,
                                  D, max_tree_depth, use_multinomial, seed, t, nullptr,
                                  chain_offset, accept_prob, n_leapfrog, depth, diverging,
                                  accepted, n_active, workspace, s);
}

int pa_nuts_tree_advance_tdev(int dtype, void* z, void* pe, void* grad, void* zq, void* rq,
                              const void* gq, const void* peq, const void* inv_mass,
                              int64_t im_stride_row, const void* step, int64_t C, int64_t D,
                              int max_tree_depth, int use_multinomial, uint64_t seed,
                              const uint64_t* t_dev, uint64_t chain_offset, void* accept_prob,
                              int32_t* n_leapfrog, int32_t* depth, int32_t* diverging,
                              int32_t* accepted, int32_t* n_active, void* workspace,
                              size_t workspace_bytes, pa_stream_t stream) {
  const uint64_t t = 0;                 // (the checks below also bound the scalar index)
  PA_TREE_COMMON_CHECKS("nuts_tree_advance_tdev")
  PA_REQUIRE(z && pe && grad && zq && rq && gq && peq && inv_mass && step && accept_prob &&
                 n_leapfrog && depth && diverging && accepted && n_active && t_dev,
             "nuts_tree_advance_tdev: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    return pa::tree_advance<float>(z, pe, grad, zq, rq, gq, peq, inv_mass, im_stride_row, step, C,
                                   D, max_tree_depth, use_multinomial, seed, 0, t_dev,
                                   chain_offset, accept_prob, n_leapfrog, depth, diverging,
                                   accepted, n_active, workspace, s);
  return pa::tree_advance<double>(z, pe, grad, zq, rq, gq, peq, inv_mass, im_stride_row, step, C,
                                  D, max_tree_depth, use_multinomial, seed, 0, t_dev,
                                  chain_offset, accept_prob, n_leapfrog, depth, diverging,
                                  accepted, n_active, workspace, s);
}

int pa_nuts_tree_run_begin(int dtype, const void* z, const void* pe, const void* grad, void* zq,
                           void* rq, const void* inv_mass, int64_t im_stride_row, const void* step,
                           int64_t C, int64_t D, int max_tree_depth, int use_multinomial,
                           uint64_t seed, uint64_t chain_offset, const int64_t* ctl, int32_t* tc,
                           int32_t* n_done, int64_t* done_flag, void* workspace,
                           size_t workspace_bytes, pa_stream_t stream) {
  const uint64_t t = 0;
  PA_TREE_COMMON_CHECKS("nuts_tree_run_begin")
  PA_REQUIRE(z && pe && grad && zq && rq && inv_mass && step && ctl && tc && n_done,
             "nuts_tree_run_begin: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    return pa::tree_begin<float>(z, pe, grad, zq, rq, inv_mass, im_stride_row, step, C, D,
                                 max_tree_depth, use_multinomial, seed, 0, chain_offset, workspace,
                                 s, ctl, tc, n_done, done_flag);
  return pa::tree_begin<double>(z, pe, grad, zq, rq, inv_mass, im_stride_row, step, C, D,
                                max_tree_depth, use_multinomial, seed, 0, chain_offset, workspace,
                                s, ctl, tc, n_done, done_flag);
}

int pa_nuts_tree_run_advance(int dtype, void* z, void* pe, void* grad, void* zq, void* rq,
                             const void* gq, const void* peq, const void* inv_mass,
                             int64_t im_stride_row, void* step, int64_t C, int64_t D,
                             int max_tree_depth, int use_multinomial, uint64_t seed,
                             uint64_t chain_offset, const int64_t* ctl, void* da_state,
                             double target_accept, void* welford, void* mean_accept,
                             int64_t* counters, int32_t* tc, int32_t* n_done, int64_t* done_flag,
                             const int32_t* slot2chain, void* zq_slot, int64_t n_slots,
                             void* accept_prob, int32_t* n_leapfrog, int32_t* depth,
                             int32_t* diverging, int32_t* accepted, void* workspace,
                             size_t workspace_bytes, pa_stream_t stream) {
  const uint64_t t = 0;
  PA_TREE_COMMON_CHECKS("nuts_tree_run_advance")
  PA_REQUIRE((slot2chain == nullptr) == (zq_slot == nullptr), "nuts_tree_run_advance: slot2chain and zq_slot go together");
  if (slot2chain == nullptr) n_slots = C;
  PA_REQUIRE(n_slots >= 1 && n_slots <= C, "nuts_tree_run_advance: n_slots=%lld outside [1, C]", (long long)n_slots);
  PA_REQUIRE(z && pe && grad && zq && rq && gq && peq && inv_mass && step && ctl && da_state &&
                 welford && mean_accept && counters && tc && n_done && accept_prob && n_leapfrog &&
                 depth && diverging && accepted,
             "nuts_tree_run_advance: NULL pointer");
  PA_REQUIRE(target_accept > 0.0 && target_accept < 1.0,
             "nuts_tree_run_advance: target_accept=%g outside (0, 1)", target_accept);
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32) {
    pa::TreeRun<float> run{ctl, (float*)step, (float*)da_state, (float*)welford,
                           (float*)mean_accept, counters, tc, n_done, done_flag, target_accept,
                           10.0, 0.75, 0.05,   // DualAveraging defaults (ops/dual_averaging.py)
                           slot2chain, (float*)zq_slot, pa::SlotLayout{}, n_slots};
    return pa::tree_run_advance<float>(z, pe, grad, zq, rq, gq, peq, inv_mass, im_stride_row, C, D,
                                       max_tree_depth, use_multinomial, seed, chain_offset, run, n_slots,
                                       accept_prob, n_leapfrog, depth, diverging, accepted,
                                       workspace, s);
  }
  pa::TreeRun<double> run{ctl, (double*)step, (double*)da_state, (double*)welford,
                          (double*)mean_accept, counters, tc, n_done, done_flag, target_accept,
                          10.0, 0.75, 0.05, slot2chain, (double*)zq_slot, pa::SlotLayout{}, n_slots};
  return pa::tree_run_advance<double>(z, pe, grad, zq, rq, gq, peq, inv_mass, im_stride_row, C, D,
                                      max_tree_depth, use_multinomial, seed, chain_offset, run, n_slots,
                                      accept_prob, n_leapfrog, depth, diverging, accepted,
                                      workspace, s);
}

static int tree_layout(int n_sites, const int32_t* site_off, const int32_t* site_len, int64_t D,
                       pa::SlotLayout* lay) {
  *lay = pa::SlotLayout{};
  if (n_sites == 0) return PA_OK;
  PA_REQUIRE(n_sites >= 1 && n_sites <= pa::TREE_MAX_SITES && site_off && site_len,
             "nuts_tree: 1..%d sites", pa::TREE_MAX_SITES);
  int64_t at = 0;
  for (int k = 0; k < n_sites; ++k) {
    PA_REQUIRE(site_off[k] == at && site_len[k] >= 1, "nuts_tree: sites must tile [0, D) in ascending order");
    lay->off[k] = site_off[k];
    lay->len[k] = site_len[k];
    at += site_len[k];
  }
  PA_REQUIRE(at == D, "nuts_tree: the sites cover %lld of D = %lld coordinates", (long long)at, (long long)D);
  lay->n_sites = n_sites;
  return PA_OK;
}

int pa_nuts_tree_compact(int dtype, const void* zq, int64_t C, int64_t D, int max_tree_depth,
                         int32_t* slot2chain, void* zq_slot, int64_t n_slots, int32_t* n_placed,
                         int n_sites, const int32_t* site_off, const int32_t* site_len,
                         void* workspace, size_t workspace_bytes, pa_stream_t stream) {
  const uint64_t t = 0;
  const int64_t im_stride_row = 0;
  PA_TREE_COMMON_CHECKS("nuts_tree_compact")
  PA_REQUIRE(zq && zq_slot && n_placed && n_slots >= 1 && n_slots <= C, "nuts_tree_compact: bad arguments");
  PA_REQUIRE(slot2chain != nullptr || n_slots == C, "nuts_tree_compact: the identity map is the full round");
  pa::SlotLayout lay;
  const int rc = tree_layout(n_sites, site_off, site_len, D, &lay);
  if (rc != PA_OK) return rc;
  const int identity = slot2chain == nullptr;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32) {
    pa::TreeWs<float> ws = pa::tree_ws<float>(workspace, C, D, max_tree_depth);
    hipLaunchKernelGGL((pa::nuts_tree_compact_kernel<float>), dim3(1), dim3(1024), 0, s, ws.iscal, C, (int)D,
                       (const float*)zq, slot2chain, (float*)zq_slot, (int)n_slots, n_placed, lay, identity);
  } else {
    pa::TreeWs<double> ws = pa::tree_ws<double>(workspace, C, D, max_tree_depth);
    hipLaunchKernelGGL((pa::nuts_tree_compact_kernel<double>), dim3(1), dim3(1024), 0, s, ws.iscal, C, (int)D,
                       (const double*)zq, slot2chain, (double*)zq_slot, (int)n_slots, n_placed, lay, identity);
  }
  return pa::check_launch("nuts_tree_compact_kernel");
}

// parent-valued parameters of the direct potential (DirectSite): p == NULL with stride -(q + 1)
static int tree_check_parents(const char* who, int n_sites, const int32_t* site_len, const void* const* site_p0,
                              const int64_t* site_s0, const void* const* site_p1, const int64_t* site_s1, int64_t D) {
  for (int k = 0; k < n_sites; ++k) {
    const void* const pp[2] = {site_p0[k], site_p1[k]};
    const int64_t ss[2] = {site_s0[k], site_s1[k]};
    for (int t = 0; t < 2; ++t) {
      if (pp[t] != nullptr) {
        PA_REQUIRE(ss[t] >= 0, "%s: negative stride of a parameter tensor", who);
        continue;
      }
      if (ss[t] >= 0) continue;                               // unused parameter
      const int64_t q = -ss[t] - 1;
      PA_REQUIRE(q >= 0 && q < n_sites && q != k, "%s: site %d takes its parameter from site %lld", who, k, (long long)q);
      PA_REQUIRE(site_len[k] % site_len[q] == 0, "%s: site %d (%d elements) is not a whole number of copies of its "
                 "parent site %lld (%d elements)", who, k, site_len[k], (long long)q, site_len[q]);
      PA_REQUIRE(D <= pa::TREE_DIRECT_MAXD, "%s: parent-valued parameters need D <= %d", who, pa::TREE_DIRECT_MAXD);
    }
  }
  return PA_OK;
}

int pa_nuts_tree_run_advance_direct(void* z, void* pe, void* grad, void* zq, void* rq, const void* inv_mass,
                                    int64_t im_stride_row, void* step, int64_t C, int64_t D,
                                    int max_tree_depth, int use_multinomial, uint64_t seed,
                                    uint64_t chain_offset, const int64_t* ctl, void* da_state,
                                    double target_accept, void* welford, void* mean_accept,
                                    int64_t* counters, int32_t* tc, int32_t* n_done, int64_t* done_flag,
                                    const int32_t* slot2chain, void* zq_pack, int64_t n_slots,
                                    int n_sites, const int32_t* site_off, const int32_t* site_len,
                                    const int32_t* site_dist, const int32_t* site_transform,
                                    const double* site_lower, const void* const* site_p0,
                                    const int64_t* site_s0, const void* const* site_p1,
                                    const int64_t* site_s1, const void* const* site_g_ext,
                                    const void* ll_ext, void* accept_prob, int32_t* n_leapfrog,
                                    int32_t* depth, int32_t* diverging, int32_t* accepted, void* workspace,
                                    size_t workspace_bytes, pa_stream_t stream) {
  const int dtype = PA_F32;
  const uint64_t t = 0;
  PA_TREE_COMMON_CHECKS("nuts_tree_run_advance_direct")
  PA_REQUIRE(z && pe && grad && zq && rq && inv_mass && step && ctl && da_state && welford && mean_accept &&
                 counters && tc && n_done && accept_prob && n_leapfrog && depth && diverging && accepted &&
                 zq_pack && site_dist && site_transform && site_lower && site_p0 && site_s0 && site_p1 &&
                 site_s1 && site_g_ext,
             "nuts_tree_run_advance_direct: NULL pointer");
  PA_REQUIRE(target_accept > 0.0 && target_accept < 1.0, "nuts_tree_run_advance_direct: target_accept");
  if (slot2chain == nullptr) n_slots = C;
  PA_REQUIRE(n_slots >= 1 && n_slots <= C, "nuts_tree_run_advance_direct: n_slots=%lld outside [1, C]",
             (long long)n_slots);
  PA_REQUIRE(n_sites >= 1, "nuts_tree_run_advance_direct: at least one site");
  pa::SlotLayout lay;
  int rc = tree_layout(n_sites, site_off, site_len, D, &lay);
  if (rc != PA_OK) return rc;
  rc = tree_check_parents("nuts_tree_run_advance_direct", n_sites, site_len, site_p0, site_s0, site_p1, site_s1, D);
  if (rc != PA_OK) return rc;
  pa::TreeDirect dp{};
  dp.n_sites = n_sites;
  dp.ll_ext = ll_ext;
  for (int k = 0; k < n_sites; ++k) {
    PA_REQUIRE(site_transform[k] == 0 || site_transform[k] == 1, "nuts_tree_run_advance_direct: transform");
    dp.s[k] = pa::DirectSite{site_dist[k], site_transform[k], site_p0[k], site_p1[k], site_s0[k], site_s1[k],
                             site_g_ext[k], site_lower[k]};
  }
  pa::TreeRun<float> run{ctl, (float*)step, (float*)da_state, (float*)welford, (float*)mean_accept, counters,
                         tc, n_done, done_flag, target_accept, 10.0, 0.75, 0.05, slot2chain, (float*)zq_pack,
                         lay, n_slots};
  return pa::tree_run_advance<float>(z, pe, grad, zq, rq, nullptr, nullptr, inv_mass, im_stride_row, C, D,
                                     max_tree_depth, use_multinomial, seed, chain_offset, run, n_slots,
                                     accept_prob, n_leapfrog, depth, diverging, accepted, workspace,
                                     pa::as_stream(stream), &dp);
}

int pa_nuts_direct_potential(const void* zq_pack, int64_t n_slots, int64_t D, int n_sites,
                             const int32_t* site_off, const int32_t* site_len, const int32_t* site_dist,
                             const int32_t* site_transform, const double* site_lower,
                             const void* const* site_p0, const int64_t* site_s0, const void* const* site_p1,
                             const int64_t* site_s1, const void* const* site_g_ext, const void* ll_ext,
                             void* pe_out, void* grad_out, pa_stream_t stream) {
  PA_REQUIRE(zq_pack && pe_out && grad_out && site_off && site_len && site_dist && site_transform &&
                 site_lower && site_p0 && site_s0 && site_p1 && site_s1 && site_g_ext,
             "nuts_direct_potential: NULL pointer");
  PA_REQUIRE(n_slots >= 0 && n_slots < (1ll << 31), "nuts_direct_potential: n_slots=%lld", (long long)n_slots);
  PA_REQUIRE(D >= 1 && D <= 512, "nuts_direct_potential: D=%lld outside [1, 512]", (long long)D);
  PA_REQUIRE(n_sites >= 1, "nuts_direct_potential: at least one site");
  if (n_slots == 0) return PA_OK;
  pa::SlotLayout lay;
  int rc = tree_layout(n_sites, site_off, site_len, D, &lay);
  if (rc != PA_OK) return rc;
  rc = tree_check_parents("nuts_direct_potential", n_sites, site_len, site_p0, site_s0, site_p1, site_s1, D);
  if (rc != PA_OK) return rc;
  pa::TreeDirect dp{};
  dp.n_sites = n_sites;
  dp.ll_ext = ll_ext;
  for (int k = 0; k < n_sites; ++k) {
    PA_REQUIRE(site_transform[k] == 0 || site_transform[k] == 1, "nuts_direct_potential: transform");
    dp.s[k] = pa::DirectSite{site_dist[k], site_transform[k], site_p0[k], site_p1[k], site_s0[k], site_s1[k],
                             site_g_ext[k], site_lower[k]};
  }
  hipStream_t s = pa::as_stream(stream);
  if (D <= 128)
    hipLaunchKernelGGL((pa::nuts_direct_potential_kernel<1, 2>), dim3((unsigned)n_slots), dim3(64), 0, s,
                       (const float*)zq_pack, n_slots, (int)D, lay, dp, (float*)pe_out, (float*)grad_out);
  else
    hipLaunchKernelGGL((pa::nuts_direct_potential_kernel<1, 8>), dim3((unsigned)n_slots), dim3(64), 0, s,
                       (const float*)zq_pack, n_slots, (int)D, lay, dp, (float*)pe_out, (float*)grad_out);
  return pa::check_launch("nuts_direct_potential_kernel");
}

}  // extern "C"
