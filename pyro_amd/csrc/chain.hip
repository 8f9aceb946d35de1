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

namespace pa {

constexpr int CH_FIN = 0, CH_MULTI = 1, CH_MF = 2, CH_ADAM = 3, CH_PHASES = 4;
constexpr int CH_MF_SITES = 8;

struct MfArgs8 {          // same prefix layout as MfArgs: the body addresses sites by offset
  int nsites;
  MfSiteDev s[CH_MF_SITES];
};
static_assert(offsetof(MfArgs8, s) == offsetof(MfArgs, s), "layout");

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
};
static_assert(sizeof(ChainArgs) <= 4096, "kernel arguments are limited to 4 KiB");

// Arrival at the end of phase p (called by the workgroups that had work in it, all threads).  The
// workgroup barrier orders every wave's stores (acknowledged by the XCD's L2) before thread 0's
// agent-scope release -- ONE L2 write-back per workgroup, not one per wave: with a fence in every
// wave the 255 workgroups of the finalize phase queued 1020 write-backs on the L2s and the phase
// took 20 us -- then ONE relaxed add.  The final arrival of the launch's last phase re-arms the
// counters: by then every wait of the launch has completed (a workgroup only waits in front of a
// phase it takes part in, and arrives at that phase afterwards).
__device__ __forceinline__ void chain_signal(uint32_t* sync, int p, uint32_t expected, bool last) {
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
__device__ __forceinline__ void chain_wait(uint32_t* sync, int p, uint32_t expected) {
  if (threadIdx.x == 0) {
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(&sync[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 400000000ull) __builtin_trap();   // 4 s of the 100 MHz clock
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int DT, int PT>
__device__ __forceinline__ void chain_fin(const ChainArgs& a, int me, int nw) {
  for (int64_t vb = me; vb < a.grid[CH_FIN]; vb += nw) {
    glm_finalize_body<DT, PT>(vb, a.fin_part, a.fin_nblocks, a.fin_npass, a.fin_D, a.fin_P,
                              a.fin_scale, a.fin_ll, a.fin_gw, a.fin_gb, a.fin_ll_offset);
    __syncthreads();      // the body's LDS staging is reused by the next virtual workgroup
  }
}

#define PA_CHAIN_STAMP(i)                                                     \
  do {                                                                        \
    if (a.stamps != nullptr && threadIdx.x == 0 && (me == 0 || total_wg))     \
      a.stamps[(total_wg ? 16 : 0) + (i)] = wall_clock64();                   \
  } while (0)

__global__ __launch_bounds__(1024) void chain_kernel(const ChainArgs a) {
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
      if (a.fin_DT == 1 && a.fin_PT == 2) chain_fin<1, 2>(a, me, nw);
      else if (a.fin_DT == 1 && a.fin_PT == 1) chain_fin<1, 1>(a, me, nw);
      else if (a.fin_DT == 2 && a.fin_PT == 1) chain_fin<2, 1>(a, me, nw);
      else chain_fin<4, 1>(a, me, nw);
      chain_signal(sync, CH_FIN, (uint32_t)part_fin, a.last == CH_FIN);
    }
    prev = CH_FIN;
    prev_n = part_fin;
  }
  PA_CHAIN_STAMP(1);
  if (a.have[CH_MULTI]) {
    constexpr uint32_t KB = (uint32_t)offsetof(ChainArgs, multi);
    if (total_wg || me < part_multi - 1) {
      if (prev >= 0) chain_wait(sync, prev, (uint32_t)prev_n);
      PA_CHAIN_STAMP(2);
      if (total_wg) {
        multi_sum_body<float, MULTI_THREADS>(KB, a.multi_out, a.multi_coef_all, a.multi_accumulate);
      } else {
        for (int vb = me; vb < a.multi_n; vb += nw) {
          multi_grad_body<float>(KB, vb, a.multi_g, a.multi_coef_all);
          __syncthreads();
        }
      }
      chain_signal(sync, CH_MULTI, (uint32_t)part_multi, a.last == CH_MULTI);
    }
    prev = CH_MULTI;
    prev_n = part_multi;
  }
  PA_CHAIN_STAMP(3);
  if (total_wg) return;           // the total's workgroup has no further part
  if (a.have[CH_MF]) {
    constexpr uint32_t KB = (uint32_t)offsetof(ChainArgs, mf);
    if (me < part_mf) {
      if (prev >= 0) chain_wait(sync, prev, (uint32_t)prev_n);
      PA_CHAIN_STAMP(4);
      for (int vb = me; vb < a.grid[CH_MF]; vb += nw) {
        meanfield_sample_bwd_body<float>(KB, (uint32_t)(vb % a.mf_nsites),
                                         (uint32_t)(vb / a.mf_nsites), (uint32_t)a.mf_gy, a.mf_P);
        __syncthreads();
      }
      chain_signal(sync, CH_MF, (uint32_t)part_mf, a.last == CH_MF);
    }
    prev = CH_MF;
    prev_n = part_mf;
  }
  PA_CHAIN_STAMP(5);
  if (a.have[CH_ADAM]) {
    const int part_adam = a.grid[CH_ADAM] < nw ? a.grid[CH_ADAM] : nw;
    if (me < part_adam) {
      if (prev >= 0) chain_wait(sync, prev, (uint32_t)prev_n);
      PA_CHAIN_STAMP(6);
      for (int64_t vb = me; vb < a.grid[CH_ADAM]; vb += nw)
        adam_body<float>(vb, (int64_t)a.grid[CH_ADAM], a.ad_p, a.ad_g, a.ad_m, a.ad_v, a.ad_n,
                         a.ad_lr, a.ad_b1, a.ad_b2, a.ad_eps, a.ad_wd, a.ad_clip, a.ad_lrd,
                         a.ad_clipped, a.ad_step, a.ad_zero, a.ad_pub);
      chain_signal(sync, CH_ADAM, (uint32_t)part_adam, a.last == CH_ADAM);
    }
  }
  PA_CHAIN_STAMP(7);
}

// ---- host side: the recording ------------------------------------------------------------------
struct ChainState {
  bool on = false;
  hipStream_t stream = nullptr;
  int top = -1;            // highest phase kind recorded so far (-1: nothing pending)
  int launches = 0;        // chain launches since pa_chain_begin (diagnostics)
  int phases = 0;          // phases those launches carried
  ChainArgs a;
};
// One recording per process, NOT per thread: the backward half of a step (the guide's backward,
// and with it pa_meanfield_normal_sample_bwd) runs on torch's autograd thread while the recording
// was started on the caller's.  The entry points are serialised by the host language (the GIL).
static ChainState g_chain;
static uint64_t* g_chain_stamps = nullptr;

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
  int nw = cu_count() - 1;
  if (nw < 1) nw = 1;
  if (maxgrid < nw) nw = maxgrid;
  const hipStream_t s = c.stream;
  c.top = -1;                                   // (before the launch: as_stream() must not recurse)
  hipLaunchKernelGGL(chain_kernel, dim3((unsigned)nw + 1), dim3(1024), 0, s, a);
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
  const bool shape_ok = (DT == 1 && (PT == 1 || PT == 2)) || (PT == 1 && (DT == 2 || DT == 4));
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
hipStream_t as_stream(pa_stream_t s) {
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
  pa::g_chain.launches = pa::g_chain.phases = 0;
  for (int p = 0; p < pa::CH_PHASES; ++p) pa::g_chain.a.have[p] = 0;
  pa::g_chain.a.sync = (uint32_t*)sync_words;
  return PA_OK;
}

int pa_chain_flush(void) {
  if (!pa::g_chain.on) return PA_OK;
  return pa::chain_launch();
}

int pa_chain_end(int* launches, int* phases) {
  int rc = PA_OK;
  if (pa::g_chain.on) rc = pa::chain_launch();
  pa::g_chain.on = false;
  if (launches) *launches = pa::g_chain.launches;
  if (phases) *phases = pa::g_chain.phases;
  return rc;
}

int pa_chain_debug_stamps(void* stamps32) {
  pa::g_chain_stamps = (uint64_t*)stamps32;
  return PA_OK;
}

int pa_chain_pending(void) {
  if (!pa::g_chain.on || pa::g_chain.top < 0) return 0;
  int n = 0;
  for (int p = 0; p < pa::CH_PHASES; ++p) n += pa::g_chain.a.have[p];
  return n;
}

}  // extern "C"
