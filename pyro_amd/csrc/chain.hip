// chain.hip -- the CHAINED TAIL of an SVI step: finalize -> ELBO assembly -> guide backward -> Adam
// as phases of ONE launch.
//
// Reference path replaced: after the likelihood the reference's step is ~40 more small kernels
// (per-site sums, autograd duals, AccumulateGrad, per-parameter Adam: pyro/infer/trace_elbo.py:
// 130-159, pyro/infer/svi.py:134-162, pyro/optim/optim.py:117-155).  Round 2 had them down to four
// dependent launches of ours (glm_finalize, multi_sum_grad, meanfield_sample_bwd, adam) -- 24.5 us
// of a 106 us step, of which ~4 us each is the dispatch of a dependent graph node.  Here the four
// become phases of one kernel:
//   * the entry points record their arguments while pa_chain_begin() is in force (chain.h);
//   * one launch of `chain_kernel`: worker workgroups (256 threads) walk the virtual workgroups of
//     each phase with the SAME device code and thread geometry as the stand-alone kernels -- the
//     results are bit-identical to the unchained launches -- plus one 1024-thread workgroup for the
//     ELBO total (the geometry multi_sum_grad_kernel gives it);
//   * phases are separated by a device-wide barrier: every workgroup fences, bumps the phase's
//     counter (release, agent scope) and spins on it (acquire) before reading what the previous
//     phase wrote.  All workgroups are co-resident (grid <= #CUs), so the spin cannot deadlock; a
//     spin that outlives 4 s traps instead of hanging the device.
#include "chain.h"
#include "glm_finalize.h"
#include "multisite_dev.h"
#include "optim_dev.h"
#include "site_tail.h"

namespace pa {

constexpr int CH_FIN = 0, CH_MULTI = 1, CH_MF = 2, CH_ADAM = 3, CH_PHASES = 4;
constexpr int CH_MF_SITES = 8;

struct MfArgs8 {          // same prefix layout as MfArgs: the body addresses sites by offset
  int nsites;
  MfSiteDev s[CH_MF_SITES];
};
static_assert(offsetof(MfArgs8, s) == offsetof(MfArgs, s), "layout");

constexpr int CH_TAIL_ENTRIES = 6;
struct TailSite {
  int n_entries;
  int entries[CH_TAIL_ENTRIES];   // indices into the ELBO assembly's table, ascending
  int extras_mask;                // bit q: entries[q] carries a known extra value gradient
  int64_t off_loc, off_rho;       // element offsets of the site's loc / rho in the flat buffers
  int fast;                       // 1: the one-pass form of site_tail.h applies (entries[0] = the
  int layout;                     //    prior entry, entries[1] = the guide entry); its layout
};

struct ChainArgs {
  MultiArgs multi;        // tables first; read through the kernarg segment (multisite_dev.h)
  MfArgs8 mf;
  // --- finalize
  const float* fin_part;
  float *fin_ll, *fin_gw, *fin_gb;
  double fin_scale, fin_ll_offset;
  int fin_nblocks, fin_npass, fin_D, fin_P, fin_DT, fin_PT;
  // --- ELBO assembly
  float* multi_out;
  const float* multi_g;
  double multi_coef_all;
  int multi_accumulate, multi_n;
  // --- guide backward
  int64_t mf_P;
  int mf_gy, mf_nsites;
  // --- Adam
  float *ad_p, *ad_g, *ad_m, *ad_v;
  int64_t ad_n;
  double ad_lr, ad_b1, ad_b2, ad_eps, ad_wd, ad_clip, ad_lrd;
  int64_t* ad_step;
  int ad_clipped, ad_zero;
  AdamPublish ad_pub;
  // --- control
  uint32_t* sync;         // CH_PHASES counters, zero between launches
  int have[CH_PHASES];
  int grid[CH_PHASES];    // virtual workgroups per phase
  int last;               // the last present phase (its final arrival resets the counters)
  uint64_t* stamps;       // developer hook (pa_chain_debug_stamps): 32 wall-clock stamps, or NULL
  const int64_t* gate;    // the step gate's abort word (pa_gate_scope), or NULL
  uint32_t jitter;        // race hunting (pa_chain_tune bits 8..): seed of pseudo-random per-workgroup
                          // delays in front of every arrival and after every wait; 0 = off
  // --- the fused form (chain_tail_kernel): per mean-field site the entries whose gradients only
  //     that site's backward consumes, and where the site's parameters live in the flat buffers
  int tail_nsites;
  uint32_t fin_dep_mask;  // entries of the ELBO assembly that read the finalize phase's outputs
  TailSite tail[CH_MF_SITES];
};
static_assert(sizeof(ChainArgs) <= 6144, "kernel arguments grew unexpectedly");

// Arrival at the end of phase p (called by the workgroups that had work in it, all threads).  The
// workgroup barrier orders every wave's stores (acknowledged by the XCD's L2) before thread 0's
// agent-scope release -- ONE L2 write-back per workgroup, not one per wave: with a fence in every
// wave the 255 workgroups of the finalize phase queued 1020 write-backs on the L2s and the phase
// took 20 us -- then ONE relaxed add.  The final arrival of the launch's last phase re-arms the
// counters: by then every wait of the launch has completed (a workgroup only waits in front of a
// phase it takes part in, and arrives at that phase afterwards).
// race hunting: workgroup- and phase-dependent delays of 0..17 us shuffle the order in which the
// workgroups reach their arrivals and leave their waits (tests/test_chain_gpu.py runs the step under
// many seeds and compares bit for bit with the separate launches)
__device__ __forceinline__ void chain_jitter(uint32_t seed, int p, uint32_t salt) {
  if (seed == 0u) return;
  uint32_t hsh = seed * 2654435761u ^ (blockIdx.x + 1u) * 40503u ^ (uint32_t)(p + 1) * 2246822519u ^ salt;
  hsh ^= hsh >> 15;
  hsh *= 2654435761u;
  hsh ^= hsh >> 13;
  const uint32_t n = (hsh >> 8) & 63u;
  if ((hsh & 3u) == 0u) return;                      // a quarter of the workgroups: no delay at all
  for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
}

__device__ __forceinline__ void chain_signal(uint32_t jitter, uint32_t* sync, int p, uint32_t expected,
                                             bool last) {
  chain_jitter(jitter, p, 0x9e3779b9u);
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t old =
        __hip_atomic_fetch_add(&sync[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (last && old == expected - 1u) {
#pragma unroll
      for (int q = 0; q < CH_PHASES; ++q)
        __hip_atomic_store(&sync[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Wait until all `expected` workgroups of phase p have arrived.  The spin reads with RELAXED
// agent-scope loads (an acquire load per iteration would invalidate the XCD's L2 every time, for
// every workgroup that spins); ONE acquire fence by the spinning thread follows the loop (it
// invalidates the CU's vector L1 and the stale lines of the L2 for every wave of the workgroup),
// then the workgroup barrier.
__device__ __forceinline__ void chain_wait(uint32_t jitter, uint32_t* sync, int p, uint32_t expected) {
  if (threadIdx.x == 0) {
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(&sync[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 400000000ull) __builtin_trap();   // 4 s of the 100 MHz clock
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  chain_jitter(jitter, p, 0x85ebca6bu);
}

// SUB virtual workgroups of 256 threads side by side in one physical workgroup of 256 * SUB threads
// (workgroup `me` of `nw` takes the virtual workgroups SUB * me + sub, + SUB * nw, ...)
template <int DT, int PT, int SUB>
__device__ __forceinline__ void chain_fin(const ChainArgs& a, int me, int nw) {
  __shared__ double sm[SUB][FIN_GROUPS][FIN_OUT];
  const int sub = SUB == 1 ? 0 : (int)(threadIdx.x >> 8), tid = (int)(threadIdx.x & 255);
  for (int64_t base = (int64_t)me * SUB; base < a.grid[CH_FIN]; base += (int64_t)nw * SUB) {
    // (a virtual workgroup past the end does nothing but keep the barriers aligned)
    glm_finalize_body<DT, PT>(base + sub, a.fin_part, a.fin_nblocks, a.fin_npass, a.fin_D, a.fin_P,
                              a.fin_scale, a.fin_ll, a.fin_gw, a.fin_gb, a.fin_ll_offset, tid,
                              sm[sub]);
    __syncthreads();      // the body's LDS staging is reused by the next virtual workgroup
  }
}
template <int SUB>
__device__ __forceinline__ void chain_fin_any(const ChainArgs& a, int me, int nw) {
  if (a.fin_DT == 1 && a.fin_PT == 2) chain_fin<1, 2, SUB>(a, me, nw);
  else if (a.fin_DT == 1 && a.fin_PT == 1) chain_fin<1, 1, SUB>(a, me, nw);
  else if (a.fin_DT == 2 && a.fin_PT == 1) chain_fin<2, 1, SUB>(a, me, nw);
  else if (a.fin_DT == 2 && a.fin_PT == 2) chain_fin<2, 2, SUB>(a, me, nw);      // plane image, D <= 64
  else if (a.fin_DT == 4 && a.fin_PT == 2) chain_fin<4, 2, SUB>(a, me, nw);      // plane image, D <= 128
  else chain_fin<4, 1, SUB>(a, me, nw);
}

#define PA_CHAIN_STAMP(i)                                                     \
  do {                                                                        \
    if (a.stamps != nullptr && threadIdx.x == 0 && (me == 0 || total_wg))     \
      a.stamps[(total_wg ? 16 : 0) + (i)] = wall_clock64();                   \
  } while (0)

__global__ __launch_bounds__(1024) void chain_kernel(const ChainArgs a) {
  if (a.gate != nullptr && *a.gate != 0) return;    // the step gate gave this replay up (pa_gate)
  const bool total_wg = blockIdx.x == gridDim.x - 1;
  // worker workgroups run code written for 256 threads: the surplus waves leave before any barrier
  if (!total_wg && threadIdx.x >= 256) return;
  const int nw = (int)gridDim.x - 1, me = (int)blockIdx.x;
  uint32_t* sync = a.sync;
  // workgroups that take part in a phase = min(its virtual workgroups, workers) (+ the total's)
  const int part_fin = a.grid[CH_FIN] < nw ? a.grid[CH_FIN] : nw;
  const int part_multi = (a.multi_n < nw ? a.multi_n : nw) + 1;
  const int part_mf = a.grid[CH_MF] < nw ? a.grid[CH_MF] : nw;
  int prev = -1, prev_n = 0;
  PA_CHAIN_STAMP(0);

  if (a.have[CH_FIN]) {
    if (!total_wg && me < part_fin) {
      chain_fin_any<1>(a, me, nw);
      chain_signal(a.jitter, sync, CH_FIN, (uint32_t)part_fin, a.last == CH_FIN);
    }
    prev = CH_FIN;
    prev_n = part_fin;
  }
  PA_CHAIN_STAMP(1);
  if (a.have[CH_MULTI]) {
    constexpr uint32_t KB = (uint32_t)offsetof(ChainArgs, multi);
    if (total_wg || me < part_multi - 1) {
      if (prev >= 0) chain_wait(a.jitter, sync, prev, (uint32_t)prev_n);
      PA_CHAIN_STAMP(2);
      if (total_wg) {
        multi_sum_body<float, MULTI_THREADS>(KB, a.multi_out, a.multi_coef_all, a.multi_accumulate);
      } else {
        for (int vb = me; vb < a.multi_n; vb += nw) {
          multi_grad_body<float>(KB, vb, a.multi_g, a.multi_coef_all);
          __syncthreads();
        }
      }
      chain_signal(a.jitter, sync, CH_MULTI, (uint32_t)part_multi, a.last == CH_MULTI);
    }
    prev = CH_MULTI;
    prev_n = part_multi;
  }
  PA_CHAIN_STAMP(3);
  if (total_wg) return;           // the total's workgroup has no further part
  if (a.have[CH_MF]) {
    constexpr uint32_t KB = (uint32_t)offsetof(ChainArgs, mf);
    if (me < part_mf) {
      if (prev >= 0) chain_wait(a.jitter, sync, prev, (uint32_t)prev_n);
      PA_CHAIN_STAMP(4);
      for (int vb = me; vb < a.grid[CH_MF]; vb += nw) {
        meanfield_sample_bwd_body<float>(KB, (uint32_t)(vb % a.mf_nsites),
                                         (uint32_t)(vb / a.mf_nsites), (uint32_t)a.mf_gy, a.mf_P);
        __syncthreads();
      }
      chain_signal(a.jitter, sync, CH_MF, (uint32_t)part_mf, a.last == CH_MF);
    }
    prev = CH_MF;
    prev_n = part_mf;
  }
  PA_CHAIN_STAMP(5);
  if (a.have[CH_ADAM]) {
    const int part_adam = a.grid[CH_ADAM] < nw ? a.grid[CH_ADAM] : nw;
    if (me < part_adam) {
      if (prev >= 0) chain_wait(a.jitter, sync, prev, (uint32_t)prev_n);
      PA_CHAIN_STAMP(6);
      for (int64_t vb = me; vb < a.grid[CH_ADAM]; vb += nw)
        adam_body<float>(vb, (int64_t)a.grid[CH_ADAM], a.ad_p, a.ad_g, a.ad_m, a.ad_v, a.ad_n,
                         a.ad_lr, a.ad_b1, a.ad_b2, a.ad_eps, a.ad_wd, a.ad_clip, a.ad_lrd,
                         a.ad_clipped, a.ad_step, a.ad_zero, a.ad_pub);
      chain_signal(a.jitter, sync, CH_ADAM, (uint32_t)part_adam, a.last == CH_ADAM);
    }
  }
  PA_CHAIN_STAMP(7);
}

// ---- the fused form ------------------------------------------------------------------------------
// When every gradient the ELBO assembly produces is consumed by the backward of ONE mean-field site,
// and the sites' parameters tile the optimizer's flat buffer, the three phases after the finalize
// step have no cross-workgroup dependence at all: workgroup k runs the assembly's gradient code for
// site k's entries, the site's backward (sum over the particles) and Adam on the site's two slices
// of the flat buffers, one after the other, separated by workgroup barriers only.  The ELBO total
// runs beside them in its own workgroup; the last of the nsites + 1 arrivals advances the step
// counter and hands the loss to the host.  Same device code and thread geometry per element as the
// separate launches: bit-identical results.
__global__ __launch_bounds__(1024) void chain_tail_kernel(const ChainArgs a) {
  if (a.gate != nullptr && *a.gate != 0) return;    // the step gate gave this replay up (pa_gate)
  const bool total_wg = blockIdx.x == gridDim.x - 1;
  const int nw = (int)gridDim.x - 1, me = (int)blockIdx.x;
  uint32_t* sync = a.sync;
  // roles: workgroups [0, nsites) = one per mean-field site (code written for 256 threads: the
  // surplus waves leave at once; they start on their own entries at once), [nsites, nw) = the
  // finalize phase's workers (all 1024 threads: four virtual workgroups side by side, so that the
  // phase is ONE round on a quarter of the chip and 68 arrivals instead of 272), the last one = the
  // ELBO total
  if (me < a.tail_nsites && threadIdx.x >= 256) return;
  const int nfw = nw - a.tail_nsites;                       // finalize workers (>= 1 when needed)
  const int fin_vwg = (a.grid[CH_FIN] + 3) / 4;
  const int part_fin = fin_vwg < nfw ? fin_vwg : nfw;
  PA_CHAIN_STAMP(0);
  if (!total_wg && me >= a.tail_nsites) {
    const int fme = me - a.tail_nsites;
    if (a.have[CH_FIN] && fme < part_fin) {
      chain_fin_any<4>(a, fme, nfw);
      // (the counters are re-armed by the tail's last arrival, below)
      chain_signal(a.jitter, sync, CH_FIN, (uint32_t)part_fin, false);
    }
    return;
  }
  PA_CHAIN_STAMP(1);
  constexpr uint32_t KB = (uint32_t)offsetof(ChainArgs, multi);
  constexpr uint32_t KMF = (uint32_t)offsetof(ChainArgs, mf);
  // (the site record is read field by field through the kernarg segment: indexing a local copy of
  //  its entry list with a run-time index would put the copy into scratch memory)
  const uint32_t KT = (uint32_t)offsetof(ChainArgs, tail) + (uint32_t)me * (uint32_t)sizeof(TailSite);
  auto tail_entry = [&](int q) -> int {
    return kernarg_load<int>(KT + (uint32_t)offsetof(TailSite, entries) + 4u * (uint32_t)q);
  };
  int ts_n = 0, ts_mask = 0, ts_fast = 0;
  int64_t step = 0;
  if (!total_wg) ts_fast = kernarg_load<int>(KT + (uint32_t)offsetof(TailSite, fast));
  if (!total_wg && ts_fast) {
    // ---- the one-pass form: everything of this site with each element loaded once --------------
    __shared__ double fast_red[4 * GRAD_THREADS];
    __shared__ float fast_xch[4 * 64];
    const int layout = kernarg_load<int>(KT + (uint32_t)offsetof(TailSite, layout));
    const EntryDev eh = kernarg_load<EntryDev>(KB + (uint32_t)offsetof(MultiArgs, e) +
                                               (uint32_t)tail_entry(0) * (uint32_t)sizeof(EntryDev));
    const EntryDev eg = kernarg_load<EntryDev>(KB + (uint32_t)offsetof(MultiArgs, e) +
                                               (uint32_t)tail_entry(1) * (uint32_t)sizeof(EntryDev));
    const MfSiteDev ms = kernarg_load<MfSiteDev>(KMF + (uint32_t)offsetof(MfArgs, s) +
                                                 (uint32_t)me * (uint32_t)sizeof(MfSiteDev));
    const int64_t off_loc = kernarg_load<int64_t>(KT + (uint32_t)offsetof(TailSite, off_loc));
    const int64_t off_rho = kernarg_load<int64_t>(KT + (uint32_t)offsetof(TailSite, off_rho));
    const SiteAdam ad{a.ad_p, a.ad_g, a.ad_m, a.ad_v, a.ad_step, a.ad_lr, a.ad_b1, a.ad_b2, a.ad_eps,
                      a.ad_wd, a.ad_clip, a.ad_lrd, a.ad_clipped, a.ad_zero};
    auto wait = [&]() {
      PA_CHAIN_STAMP(7);
      if (a.have[CH_FIN]) chain_wait(a.jitter, sync, CH_FIN, (uint32_t)part_fin);
      else __syncthreads();
      PA_CHAIN_STAMP(2);
    };
// (Normal priors only: instantiating the other families here makes the kernel spill
    //  registers; their sites take the generic form -- chain_plan_tail)
    site_fast<PA_DIST_NORMAL>(eh, eg, ms, layout, a.multi_coef_all, off_loc, off_rho, ad, fast_red,
                              fast_xch, wait, &step);
    PA_CHAIN_STAMP(5);
  } else {
  double total_acc = 0.0;
  if (total_wg) {
    // the entries that do not read what the finalize phase writes, while that phase is running
    total_acc = multi_sum_partial<float, MULTI_THREADS>(KB, a.fin_dep_mask, 0u);
  }
  if (!total_wg) {
    // everything that does not depend on the finalize phase runs BEFORE the wait: the gradients of
    // the site's own entries (prior, guide density); what the big kernel contributes (the extra
    // term of the value gradient) is added afterwards
    ts_n = kernarg_load<int>(KT + (uint32_t)offsetof(TailSite, n_entries));
    ts_mask = kernarg_load<int>(KT + (uint32_t)offsetof(TailSite, extras_mask));
    for (int q = 0; q < ts_n; ++q) {
      multi_grad_body<float>(KB, tail_entry(q), a.multi_g, a.multi_coef_all, GRAD_NO_EXTRAS);
      __syncthreads();
    }
  }
  PA_CHAIN_STAMP(7);
  if (a.have[CH_FIN]) chain_wait(a.jitter, sync, CH_FIN, (uint32_t)part_fin);
  PA_CHAIN_STAMP(2);
  if (total_wg) {
    total_acc += multi_sum_partial<float, MULTI_THREADS>(KB, 0u, a.fin_dep_mask);
    multi_sum_finish<float>(total_acc, a.multi_out, a.multi_coef_all, a.multi_accumulate);
    if (threadIdx.x == 0) step = a.ad_step[0] + 1;
  } else {
    for (int q = 0; q < ts_n; ++q)
      if (ts_mask >> q & 1)
        multi_grad_body<float>(KB, tail_entry(q), a.multi_g, a.multi_coef_all, GRAD_EXTRAS_ONLY);
    __syncthreads();
    PA_CHAIN_STAMP(3);
    // all column tiles of the site in this workgroup (tile 0 of 1: the body strides over them)
    meanfield_sample_bwd_body<float>(KMF, (uint32_t)me, 0u, 1u, a.mf_P);
    __syncthreads();
    PA_CHAIN_STAMP(4);
    const MfSiteDev ms = kernarg_load<MfSiteDev>(KMF + (uint32_t)offsetof(MfArgs, s) +
                                                 (uint32_t)me * (uint32_t)sizeof(MfSiteDev));
    const int64_t off_loc = kernarg_load<int64_t>(KT + (uint32_t)offsetof(TailSite, off_loc));
    const int64_t off_rho = kernarg_load<int64_t>(KT + (uint32_t)offsetof(TailSite, off_rho));
    adam_two_ranges<float>(off_loc, off_rho, ms.n, a.ad_p, a.ad_g, a.ad_m, a.ad_v, a.ad_step,
                           a.ad_lr, a.ad_b1, a.ad_b2, a.ad_eps, a.ad_wd, a.ad_clip, a.ad_lrd,
                           a.ad_clipped, a.ad_zero, &step);
    PA_CHAIN_STAMP(5);
  }
  }   // (generic form)
  // arrival: the workgroup's writes are released, then the ticket; the last of the nsites + 1
  // arrivals (acquire: it reads the total another workgroup wrote) ends the step
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned long long ticket = __hip_atomic_fetch_add(
        reinterpret_cast<unsigned long long*>(a.ad_step + 1), 1ull, __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (unsigned long long)a.tail_nsites) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      a.ad_step[1] = 0;
      a.ad_step[0] = step;
#pragma unroll
      for (int q = 0; q < CH_PHASES; ++q)
        __hip_atomic_store(&sync[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      publish_to_host(a.ad_pub);
    }
  }
  PA_CHAIN_STAMP(6);
}

static int g_chain_fast_sites = 1;     // pa_chain_tune bit 1

// Does the recorded chain have the fused form?  Fills a.tail / a.tail_nsites.
static bool chain_plan_tail(ChainArgs& a) {
  if (!(a.have[CH_MULTI] && a.have[CH_MF] && a.have[CH_ADAM])) return false;
  if (a.multi_g != nullptr || a.mf_nsites < 1 || a.mf_nsites > CH_MF_SITES) return false;
  const float* g0 = a.ad_g;
  const float* g1 = a.ad_g + a.ad_n;
  int owner[PA_MULTI_MAX_ENTRIES];
  for (int e = 0; e < a.multi.n; ++e) owner[e] = -1;
  int64_t covered = 0;
  int64_t lo[2 * CH_MF_SITES], hi[2 * CH_MF_SITES];
  for (int k = 0; k < a.mf_nsites; ++k) {
    const MfSiteDev& s = a.mf.s[k];
    TailSite& t = a.tail[k];
    t.n_entries = 0;
    t.extras_mask = 0;
    if (s.n < 1 || s.n > 4096 || !s.accumulate || s.d_loc == nullptr || s.d_rho == nullptr) return false;
    const float* dl = (const float*)s.d_loc;
    const float* dr = (const float*)s.d_rho;
    if (dl < g0 || dl + s.n > g1 || dr < g0 || dr + s.n > g1) return false;
    t.off_loc = dl - g0;
    t.off_rho = dr - g0;
    lo[2 * k] = t.off_loc; hi[2 * k] = t.off_loc + s.n;
    lo[2 * k + 1] = t.off_rho; hi[2 * k + 1] = t.off_rho + s.n;
    covered += 2 * s.n;
    for (int e = 0; e < a.multi.n; ++e) {
      const EntryDev& en = a.multi.e[e];
      const bool own_v = (en.need & PA_NEED_VALUE) && en.dv && !(en.need & PA_VALUE_BY_CHAIN);
      const bool own_a = (en.need & PA_NEED_P0) && en.da, own_b = (en.need & PA_NEED_P1) && en.db;
      int hits = 0, outs = 0;
      if (own_v) { ++outs; hits += (en.dv == s.d_z); }
      if (own_a) { ++outs; hits += (en.da == s.d_loc_out || en.da == s.d_scale); }
      if (own_b) { ++outs; hits += (en.db == s.d_loc_out || en.db == s.d_scale); }
      if (hits == 0) continue;
      if (hits != outs || owner[e] >= 0) return false;     // an output goes somewhere else
      if (t.n_entries == CH_TAIL_ENTRIES) return false;
      owner[e] = k;
      if (t.n_entries == 0) t.extras_mask = 0;
      if (own_v && en.xg != nullptr) t.extras_mask |= 1 << t.n_entries;
      t.entries[t.n_entries++] = e;
    }
  }
  // every entry that writes a gradient belongs to a site
  for (int e = 0; e < a.multi.n; ++e) {
    const EntryDev& en = a.multi.e[e];
    const bool writes = ((en.need & PA_NEED_VALUE) && en.dv && !(en.need & PA_VALUE_BY_CHAIN)) ||
                        ((en.need & PA_NEED_P0) && en.da) || ((en.need & PA_NEED_P1) && en.db);
    if (writes && owner[e] < 0) return false;
  }
  // every gradient a site's backward reads is written by one of its entries
  for (int k = 0; k < a.mf_nsites; ++k) {
    const MfSiteDev& s = a.mf.s[k];
    const void* wants[3] = {s.d_z, s.d_scale, s.d_loc_out};
    for (const void* w : wants) {
      if (w == nullptr) continue;
      bool found = false;
      for (int q = 0; q < a.tail[k].n_entries && !found; ++q) {
        const EntryDev& en = a.multi.e[a.tail[k].entries[q]];
        found = en.dv == w || en.da == w || en.db == w;
      }
      if (!found) return false;
    }
  }
  // the sites' slices tile the flat buffer exactly
  if (covered != a.ad_n) return false;
  const int m = 2 * a.mf_nsites;
  for (int i = 0; i < m; ++i)
    for (int j = i + 1; j < m; ++j)
      if (lo[i] < hi[j] && lo[j] < hi[i]) return false;
  // which entries read what the finalize phase writes (all of them when a wave of the total's
  // workgroup owns several entries: the split sum is then not the same sequence of additions)
  a.fin_dep_mask = 0u;
  if (a.have[CH_FIN]) {
    auto in_fin = [&](const void* p) {
      const float* f = (const float*)p;
      return p != nullptr &&
             ((f >= a.fin_ll && f < a.fin_ll + a.fin_P) || (f >= a.fin_gb && f < a.fin_gb + a.fin_P) ||
              (f >= a.fin_gw && f < a.fin_gw + (int64_t)a.fin_P * a.fin_D));
    };
    for (int e = 0; e < a.multi.n; ++e) {
      const EntryDev& en = a.multi.e[e];
      if (in_fin(en.v) || in_fin(en.a) || in_fin(en.b) || in_fin(en.m)) a.fin_dep_mask |= 1u << e;
    }
    if (a.multi.n > MULTI_THREADS / 64) a.fin_dep_mask = 0xffffffffu;
  }
  // which sites have the shape site_tail.h does in one pass
  for (int k = 0; k < a.mf_nsites; ++k) {
    const MfSiteDev& s = a.mf.s[k];
    TailSite& t = a.tail[k];
    t.fast = 0;
    t.layout = SITE_LAYOUT_COLS;
    if (!g_chain_fast_sites || t.n_entries != 2) continue;
    const EntryDev& h = a.multi.e[t.entries[0]];
    const EntryDev& q = a.multi.e[t.entries[1]];
    const bool family = h.dist == PA_DIST_NORMAL;
    const bool shape =
        family && q.dist == PA_DIST_NORMAL && h.m == nullptr &&
        q.m == nullptr && h.chain_next == t.entries[1] && q.chain_next < 0 &&
        (h.need & PA_NEED_VALUE) && !(h.need & PA_VALUE_BY_CHAIN) && h.dv == s.d_z &&
        !((h.need & PA_NEED_P0) && h.da) && !((h.need & PA_NEED_P1) && h.db) &&
        (q.need & PA_VALUE_BY_CHAIN) && (q.need & PA_NEED_P0) && (q.need & PA_NEED_P1) &&
        q.da == s.d_loc_out && q.db == s.d_scale && q.da != nullptr && q.db != nullptr &&
        q.v == h.v && q.vsr == h.vsr && q.vsc == h.vsc && q.rows == h.rows && q.cols == h.cols &&
        s.d_z != nullptr && s.eps != nullptr && s.rho != nullptr;
    if (!shape) continue;
    const int64_t P = a.mf_P;
    if (s.n >= 2 && s.n <= GRAD_THREADS && h.rows == P && h.cols == s.n && h.vsr == s.n &&
        h.vsc == 1 && q.asr == 0 && q.asc == 1 && q.bsr == 0 && q.bsc == 1) {
      const int64_t ng = GRAD_THREADS / s.n > 8 ? 8 : GRAD_THREADS / s.n;
      if (P <= UN * ng) { t.fast = 1; t.layout = SITE_LAYOUT_COLS; }
    } else if (s.n == 1 && h.rows == 1 && h.cols == P && P >= 2 && P <= 64 && h.vsc == 1 &&
               q.asc == 0 && q.bsc == 0) {
      t.fast = 1;
      t.layout = SITE_LAYOUT_SCALAR;
    }
  }
  a.tail_nsites = a.mf_nsites;
  return true;
}

// ---- host side: the recording ------------------------------------------------------------------
struct ChainState {
  bool on = false;
  hipStream_t stream = nullptr;
  int top = -1;            // highest phase kind recorded so far (-1: nothing pending)
  int launches = 0;        // chain launches since pa_chain_begin (diagnostics)
  int phases = 0;          // phases those launches carried
  int fused = 0;           // launches that took the fused form
  ChainArgs a;
};
// One recording per process, NOT per thread: the backward half of a step (the guide's backward,
// and with it pa_meanfield_normal_sample_bwd) runs on torch's autograd thread while the recording
// was started on the caller's.  The entry points are serialised by the host language (the GIL).
static ChainState g_chain;
static uint64_t* g_chain_stamps = nullptr;
static int g_chain_fuse = 1;
static uint32_t g_chain_jitter = 0;

static int chain_launch() {
  ChainState& c = g_chain;
  if (c.top < 0) return PA_OK;
  ChainArgs& a = c.a;
  int maxgrid = 1, np = 0;
  for (int p = 0; p < CH_PHASES; ++p)
    if (a.have[p]) {
      a.last = p;
      ++np;
      if (a.grid[p] > maxgrid) maxgrid = a.grid[p];
    }
  a.stamps = g_chain_stamps;
  gate_emit_deferred(c.stream);       // a late gate goes in front of the step's first chained kernel
  a.gate = gate_word();
  a.jitter = g_chain_jitter;
  gate_aware_launch();
  int nw = cu_count() - 1;
  if (nw < 1) nw = 1;
  if (maxgrid < nw) nw = maxgrid;
  const hipStream_t s = c.stream;
  c.top = -1;                                   // (before the launch: as_stream() must not recurse)
  a.tail_nsites = 0;
  if (g_chain_fuse && cu_count() - 1 > CH_MF_SITES && chain_plan_tail(a)) {
    // the sites' workgroups + the finalize workers (as many as that phase has virtual workgroups,
    // within what is co-resident)
    nw = a.tail_nsites + (a.have[CH_FIN] ? (a.grid[CH_FIN] + 3) / 4 : 0);
    if (nw > cu_count() - 1) nw = cu_count() - 1;
    hipLaunchKernelGGL(chain_tail_kernel, dim3((unsigned)nw + 1), dim3(1024), 0, s, a);
    c.fused += 1;
  } else {
    hipLaunchKernelGGL(chain_kernel, dim3((unsigned)nw + 1), dim3(1024), 0, s, a);
  }
  c.launches += 1;
  c.phases += np;
  for (int p = 0; p < CH_PHASES; ++p) a.have[p] = 0;
  return check_launch("chain_kernel");
}

// a phase of kind `k` is about to be recorded: possible iff recording on this stream; phases are
// recorded in kind order, one of each -- anything else closes the pending chain first
static bool chain_open(pa_stream_t stream, int k, int* rc) {
  ChainState& c = g_chain;
  *rc = PA_OK;
  if (!c.on) return false;
  if ((hipStream_t)stream != c.stream) {
    *rc = chain_launch();
    return false;
  }
  if (c.top >= k) *rc = chain_launch();
  if (*rc != PA_OK) return false;
  if (c.top < 0) {
    for (int p = 0; p < CH_PHASES; ++p) c.a.have[p] = 0, c.a.grid[p] = 0;
  }
  c.top = k;
  c.a.have[k] = 1;
  return true;
}

int chain_record_fin(pa_stream_t stream, int DT, int PT, const float* part, int nblocks, int npass,
                     int D, int P, double scale, float* ll, float* gw, float* gb, double ll_offset) {
  const bool shape_ok = (PT == 1 || PT == 2) && (DT == 1 || DT == 2 || DT == 4);
  if (!g_chain.on || !shape_ok) return 0;
  int rc;
  if (!chain_open(stream, CH_FIN, &rc)) return rc;
  ChainArgs& a = g_chain.a;
  a.fin_part = part; a.fin_ll = ll; a.fin_gw = gw; a.fin_gb = gb;
  a.fin_scale = scale; a.fin_ll_offset = ll_offset;
  a.fin_nblocks = nblocks; a.fin_npass = npass; a.fin_D = D; a.fin_P = P;
  a.fin_DT = DT; a.fin_PT = PT;
  const int64_t J = (int64_t)P * D + 2 * P;
  a.grid[CH_FIN] = (int)((J + FIN_OUT - 1) / FIN_OUT);
  return 1;
}

int chain_record_multi(pa_stream_t stream, const MultiArgs& args, float* out, const float* g,
                       double coef_all, int accumulate) {
  if (!g_chain.on) return 0;
  int rc;
  if (!chain_open(stream, CH_MULTI, &rc)) return rc;
  ChainArgs& a = g_chain.a;
  a.multi = args;
  a.multi_out = out; a.multi_g = g; a.multi_coef_all = coef_all; a.multi_accumulate = accumulate;
  a.multi_n = args.n;
  a.grid[CH_MULTI] = args.n;
  return 1;
}

int chain_record_mf_bwd(pa_stream_t stream, const MfArgs& args, int nsites, int64_t P, int gy) {
  if (!g_chain.on || nsites > CH_MF_SITES) return 0;
  int rc;
  if (!chain_open(stream, CH_MF, &rc)) return rc;
  ChainArgs& a = g_chain.a;
  a.mf.nsites = nsites;
  for (int k = 0; k < nsites; ++k) a.mf.s[k] = args.s[k];
  a.mf_P = P; a.mf_gy = gy; a.mf_nsites = nsites;
  a.grid[CH_MF] = nsites * gy;
  return 1;
}

int chain_record_adam(pa_stream_t stream, float* p, float* g, float* m, float* v, int64_t n,
                      double lr, double b1, double b2, double eps, double wd, double clip,
                      double lrd, int clipped, int64_t* step_dev, int zero_grad,
                      const AdamPublish& pub) {
  if (!g_chain.on) return 0;
  int rc;
  if (!chain_open(stream, CH_ADAM, &rc)) return rc;
  ChainArgs& a = g_chain.a;
  a.ad_p = p; a.ad_g = g; a.ad_m = m; a.ad_v = v; a.ad_n = n;
  a.ad_lr = lr; a.ad_b1 = b1; a.ad_b2 = b2; a.ad_eps = eps; a.ad_wd = wd; a.ad_clip = clip;
  a.ad_lrd = lrd; a.ad_step = step_dev; a.ad_clipped = clipped; a.ad_zero = zero_grad;
  a.ad_pub = pub;
  int64_t grid = (n + 255) / 256;
  const int64_t cap = (int64_t)cu_count() * 8;
  if (grid > cap) grid = cap;
  a.grid[CH_ADAM] = (int)grid;
  return 1;
}

// every launcher of the library converts its stream argument here: whatever it is about to launch
// may read what the pending phases write, so they go first
// ---- the parked guide draw (chain.h) -------------------------------------------------------------------
struct PendingDraw {
  bool have = false;
  MfArgs args;
  int nsites = 0;
  int64_t P = 0;
  uint64_t seed = 0;
  const uint64_t* offset_dev = nullptr;
};
static PendingDraw g_draw;
static int g_draw_enabled = 1;            // pa_chain_tune bit 2 switches the fusion off

// the launch of multisite.hip (defined there): the draw as its own kernel
int meanfield_sample_launch(const MfArgs& args, int nsites, int64_t P, uint64_t seed,
                            const uint64_t* offset_dev, hipStream_t s);

int chain_flush_draw() {
  if (!g_draw.have) return PA_OK;
  g_draw.have = false;
  return meanfield_sample_launch(g_draw.args, g_draw.nsites, g_draw.P, g_draw.seed, g_draw.offset_dev,
                                 g_chain.stream);
}

int chain_park_draw(pa_stream_t stream, const MfArgs& args, int nsites, int64_t P, uint64_t seed,
                    const uint64_t* offset_dev) {
  if (!g_chain.on || !g_draw_enabled || (hipStream_t)stream != g_chain.stream || g_chain.top >= 0) return 0;
  int rc = chain_flush_draw();
  if (rc != PA_OK) return rc;
  g_draw.have = true;
  g_draw.args = args;
  g_draw.nsites = nsites;
  g_draw.P = P;
  g_draw.seed = seed;
  g_draw.offset_dev = offset_dev;
  return 1;
}

bool glm_take_pending_draw(pa_stream_t stream, const float* w, const float* b, int64_t P, int64_t D,
                           GlmDraw* out) {
  if (!g_draw.have || (hipStream_t)stream != g_chain.stream || g_draw.P != P) return false;
  int kw = -1, kb = -1;
  for (int k = 0; k < g_draw.nsites; ++k) {
    const MfSiteDev& sd = g_draw.args.s[k];
    if (sd.z == (const void*)w && sd.n == D) kw = k;
    if (b != nullptr && sd.z == (const void*)b && sd.n == 1) kb = k;
  }
  if (kw < 0 || (b != nullptr && kb < 0)) return false;
  const MfSiteDev& sw = g_draw.args.s[kw];
  out->loc_w = (const float*)sw.loc; out->rho_w = (const float*)sw.rho;
  out->z_w = (float*)sw.z; out->eps_w = (float*)sw.eps; out->scale_w = (float*)sw.scale;
  out->lout_w = (float*)sw.loc_out; out->off_w = sw.offset;
  out->have_b = kb >= 0;
  if (kb >= 0) {
    const MfSiteDev& sb = g_draw.args.s[kb];
    out->loc_b = (const float*)sb.loc; out->rho_b = (const float*)sb.rho;
    out->z_b = (float*)sb.z; out->eps_b = (float*)sb.eps; out->scale_b = (float*)sb.scale;
    out->lout_b = (float*)sb.loc_out; out->off_b = sb.offset;
  } else {
    out->loc_b = out->rho_b = nullptr;
    out->z_b = out->eps_b = out->scale_b = out->lout_b = nullptr;
    out->off_b = 0;
  }
  out->seed = g_draw.seed;
  out->offset_dev = g_draw.offset_dev;
  // the sites the GLM kernel does not draw keep their own (smaller) launch
  MfArgs rest;
  rest.nsites = 0;
  for (int k = 0; k < g_draw.nsites; ++k)
    if (k != kw && k != kb) rest.s[rest.nsites++] = g_draw.args.s[k];
  g_draw.have = false;
  if (rest.nsites > 0)
    (void)meanfield_sample_launch(rest, rest.nsites, g_draw.P, g_draw.seed, g_draw.offset_dev, g_chain.stream);
  return true;
}

hipStream_t as_stream(pa_stream_t s) {
  if (g_draw.have) (void)chain_flush_draw();
  if (g_chain.on && g_chain.top >= 0) (void)chain_launch();
  return (hipStream_t)s;
}

}  // namespace pa

extern "C" {

int pa_chain_begin(pa_stream_t stream, void* sync_words, size_t sync_bytes) {
  PA_REQUIRE(!pa::g_chain.on, "pa_chain_begin: already recording on this thread");
  PA_REQUIRE(sync_words != nullptr && sync_bytes >= PA_CHAIN_SYNC_BYTES,
             "pa_chain_begin: needs %d zeroed bytes of device memory", PA_CHAIN_SYNC_BYTES);
  pa::g_chain.on = true;
  pa::g_chain.stream = (hipStream_t)stream;
  pa::g_chain.top = -1;
  pa::g_chain.launches = pa::g_chain.phases = pa::g_chain.fused = 0;
  for (int p = 0; p < pa::CH_PHASES; ++p) pa::g_chain.a.have[p] = 0;
  pa::g_chain.a.sync = (uint32_t*)sync_words;
  return PA_OK;
}

int pa_chain_flush(void) {
  if (!pa::g_chain.on) return PA_OK;
  int rc = pa::chain_flush_draw();
  if (rc != PA_OK) return rc;
  return pa::chain_launch();
}

int pa_chain_end(int* launches, int* phases) {
  int rc = PA_OK;
  if (pa::g_chain.on) {
    rc = pa::chain_flush_draw();
    const int rc2 = pa::chain_launch();
    if (rc == PA_OK) rc = rc2;
  }
  pa::g_chain.on = false;
  if (launches) *launches = pa::g_chain.launches;
  if (phases) *phases = pa::g_chain.phases;
  return rc;
}

int pa_chain_tune(int fuse_tail) {
  // bit 0: the per-site fused form; bit 1 (with bit 0): NOT the one-pass site code of site_tail.h
  pa::g_chain_fuse = (fuse_tail & 1) ? 1 : 0;
  pa::g_chain_fast_sites = (fuse_tail & 2) ? 0 : 1;
  pa::g_chain_jitter = (uint32_t)fuse_tail >> 8;         // race hunting: see chain_jitter()
  pa::g_draw_enabled = (fuse_tail & 4) ? 0 : 1;          // bit 2: the guide draw stays its own launch
  return PA_OK;
}

int pa_chain_fused_launches(void) { return pa::g_chain.fused; }

int pa_chain_debug_stamps(void* stamps32) {
  pa::g_chain_stamps = (uint64_t*)stamps32;
#ifdef PA_CHAIN_LATENCY_PROBE
  // (64 words in this build: [32..61] are stamps taken inside the bodies by workgroup 0)
  uint64_t* p = (uint64_t*)stamps32;
  int zero = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(pa::pa_dbg_stamps), &p, sizeof(p));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(pa::pa_dbg_next), &zero, sizeof(zero));
#endif
  return PA_OK;
}

int pa_chain_pending(void) {
  if (!pa::g_chain.on || pa::g_chain.top < 0) return 0;
  int n = 0;
  for (int p = 0; p < pa::CH_PHASES; ++p) n += pa::g_chain.a.have[p];
  return n;
}

}  // extern "C"
