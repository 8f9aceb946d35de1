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
  return uni(wave_sum(x.a * y.a + x.b * y.b));
}

// g = Lambda z (Lambda symmetric, read as columns), pe = 0.5 z.g
template <typename T>
__device__ __forceinline__ void potential(const T* __restrict__ Ls, int D, int lane, P2<T> z,
                                          P2<T>& g, T& pe) {
  T g0 = T(0), g1 = T(0);
  const int d0 = D < 64 ? D : 64;
#pragma unroll 4
  for (int j = 0; j < d0; ++j) {
    const T zj = bcast_lane(z.a, j);
    g0 += Ls[j * D + lane] * zj;
    g1 += Ls[j * D + lane + 64] * zj;
  }
#pragma unroll 4
  for (int j = 64; j < D; ++j) {
    const T zj = bcast_lane(z.b, j - 64);
    g0 += Ls[j * D + lane] * zj;
    g1 += Ls[j * D + lane + 64] * zj;
  }
  g.a = lane < D ? g0 : T(0);
  g.b = lane + 64 < D ? g1 : T(0);
  pe = T(0.5) * dot(z, g);
}

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

template <typename T, int WPB>
__global__ __launch_bounds__(64 * WPB) void nuts_gaussian_kernel(
    T* __restrict__ z_io, T* __restrict__ pe_io, T* __restrict__ grad_io,
    const T* __restrict__ Lambda, const T* __restrict__ inv_mass, const T* __restrict__ step,
    int C, int D, int max_depth, int multinomial, uint64_t seed, uint64_t t,
    uint64_t chain_offset, T* __restrict__ accept_prob_out, int32_t* __restrict__ nleap_out,
    int32_t* __restrict__ depth_out, int32_t* __restrict__ div_out, int32_t* __restrict__ acc_out) {
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

  const bool va = lane < D, vb = lane + 64 < D;
  auto ld = [&](const T* p) {
    return P2<T>{va ? p[(int64_t)chain * D + lane] : T(0),
                 vb ? p[(int64_t)chain * D + lane + 64] : T(0)};
  };
  P2<T> zc = ld(z_io);
  P2<T> gc = ld(grad_io);
  T pe_c = pe_io[chain];
  // diagonal inverse mass v; mass_matrix_sqrt_inverse = sqrt(v); mass_matrix_sqrt = 1/sqrt(v)
  P2<T> v{va ? inv_mass[(int64_t)chain * D + lane] : T(1),
          vb ? inv_mass[(int64_t)chain * D + lane + 64] : T(1)};
  const P2<T> sq{Num<T>::sqrt_(v.a), Num<T>::sqrt_(v.b)};
  const P2<T> isq{T(1) / sq.a, T(1) / sq.b};
  const T eps = step[chain];

  const uint64_t ctr_base = t << 20;
  const uint64_t cid = chain_offset + (uint64_t)chain;

  // ---- momentum: r_unscaled ~ N(0, I), r = M^{1/2} r_unscaled (hmc.py:231-248) -----------
  P2<T> ru0;
  if constexpr (sizeof(T) == 4) {
    ru0.a = va ? philox_normal_f32(seed, ctr_base, (uint64_t)lane, cid) : 0.f;
    ru0.b = vb ? philox_normal_f32(seed, ctr_base, (uint64_t)(lane + 64), cid) : 0.f;
  } else {
    ru0.a = va ? philox_normal_f64(seed, ctr_base, (uint64_t)lane, cid) : 0.0;
    ru0.b = vb ? philox_normal_f64(seed, ctr_base, (uint64_t)(lane + 64), cid) : 0.0;
  }
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
      potential(Ls, D, lane, zq, gq, pe_q);
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
  T pe_chk;
  potential(Ls, D, lane, zc, gc, pe_chk);
  if (va) { z_io[(int64_t)chain * D + lane] = zc.a; grad_io[(int64_t)chain * D + lane] = gc.a; }
  if (vb) { z_io[(int64_t)chain * D + lane + 64] = zc.b;
            grad_io[(int64_t)chain * D + lane + 64] = gc.b; }
  if (lane == 0) {
    pe_io[chain] = pe_c;
    accept_prob_out[chain] = sum_accept / (T)num_prop;  // nuts.py:510
    nleap_out[chain] = num_prop;
    depth_out[chain] = tree_depth;
    div_out[chain] = diverged;
    acc_out[chain] = accepted;
  }
}

template <typename T, int WPB>
static int nuts_launch(T* z, T* pe, T* grad, const T* Lambda, const T* inv_mass, const T* step,
                       int C, int D, int max_depth, int multinomial, uint64_t seed, uint64_t t,
                       uint64_t chain_offset, T* ap, int32_t* nl, int32_t* dp, int32_t* dv, int32_t* ac, hipStream_t s) {
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
  hipEvent_t ev0, ev1;
  const bool br = take_bracket(PA_KERNEL_NUTS, &ev0, &ev1);
  if (br) (void)hipEventRecord(ev0, s);
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WPB), lds, s, z, pe, grad, Lambda, inv_mass, step, C,
                     D, max_depth, multinomial, seed, t, chain_offset, ap, nl, dp, dv, ac);
  if (br) (void)hipEventRecord(ev1, s);
  return check_launch("nuts_gaussian_kernel");
}

}  // namespace pa

extern "C" {

int pa_nuts_gaussian_transition(int dtype, void* z, void* pe, void* grad, const void* Lambda,
                                const void* inv_mass, const void* step, int64_t C, int64_t D,
                                int max_tree_depth, int use_multinomial, uint64_t seed, uint64_t t,
                                uint64_t chain_offset, void* accept_prob, int32_t* n_leapfrog, int32_t* depth,
                                int32_t* diverging, int32_t* accepted, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "nuts_gaussian: bad dtype %d", dtype);
  PA_REQUIRE(C >= 0 && D >= 1, "nuts_gaussian: bad shape C=%lld D=%lld", (long long)C,
             (long long)D);
  if (D > pa::NUTS_DMAX)
    return pa::fail(PA_ERR_UNSUPPORTED, "nuts_gaussian: D=%lld > %d", (long long)D, pa::NUTS_DMAX);
  PA_REQUIRE(max_tree_depth >= 1 && max_tree_depth <= pa::NUTS_MAX_DEPTH,
             "nuts_gaussian: max_tree_depth must be in [1,%d]", pa::NUTS_MAX_DEPTH);
  PA_REQUIRE(C < (1 << 30) && t < ((uint64_t)1 << 43), "nuts_gaussian: C or t too large");
  if (C == 0) return PA_OK;
  PA_REQUIRE(z && pe && grad && Lambda && inv_mass && step && accept_prob && n_leapfrog && depth &&
                 diverging && accepted,
             "nuts_gaussian: NULL pointer");
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    return pa::nuts_launch<float, 4>((float*)z, (float*)pe, (float*)grad, (const float*)Lambda,
                                     (const float*)inv_mass, (const float*)step, (int)C, (int)D,
                                     max_tree_depth, use_multinomial, seed, t, chain_offset,
                                     (float*)accept_prob,
                                     n_leapfrog, depth, diverging, accepted, s);
  const size_t lds2 = ((size_t)D * D + 128 + 2 * (size_t)pa::NUTS_MAX_DEPTH * (3 * 128 + 2)) * 8;
  if (lds2 <= 160 * 1024)
    return pa::nuts_launch<double, 2>((double*)z, (double*)pe, (double*)grad,
                                      (const double*)Lambda, (const double*)inv_mass,
                                      (const double*)step, (int)C, (int)D, max_tree_depth,
                                      use_multinomial, seed, t, chain_offset, (double*)accept_prob,
                                      n_leapfrog,
                                      depth, diverging, accepted, s);
  return pa::nuts_launch<double, 1>((double*)z, (double*)pe, (double*)grad, (const double*)Lambda,
                                    (const double*)inv_mass, (const double*)step, (int)C, (int)D,
                                    max_tree_depth, use_multinomial, seed, t, chain_offset,
                                    (double*)accept_prob,
                                    n_leapfrog, depth, diverging, accepted, s);
}

}  // extern "C"
