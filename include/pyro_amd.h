/*
 * pyro_amd.h -- C-ABI of the MI355X (gfx950) numerics backend for Pyro's two
 * data-parallel hot paths (SVI ELBO gradient under pyro.plate; HMC/NUTS leapfrog).
 *
 * Everything here is `extern "C"`, takes plain device pointers + sizes + a HIP
 * stream handle (as void*), and has no torch / C++ types in any signature.
 * All pointers are DEVICE pointers unless a parameter says "host".
 * Every entry point returns PA_OK (0) or a negative error code; the message for
 * the last error on the calling thread is available from pa_last_error().
 * Argument errors are detected BEFORE any launch (reference convention: shape
 * errors are Python exceptions raised before numerics run, trace_struct.py:228-235);
 * numerical problems are NOT errors: NaN/inf propagate to the outputs
 * (pyro/util.py:107-146 warn_if_nan; hmc.py:406-414 NaN -> divergence).
 *
 * Citations `path:line` are relative to the reference checkout of pyro-ppl/pyro 1.9.1;
 * "torch:" citations point into the third-party PyTorch that implements the
 * arithmetic of the reference's distributions.
 */
#ifndef PYRO_AMD_H
#define PYRO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_ABI_VERSION 8

enum { PA_OK = 0, PA_ERR_INVALID = -1, PA_ERR_UNSUPPORTED = -2, PA_ERR_LAUNCH = -3 };

/* element types of floating-point buffers */
enum { PA_F32 = 0, PA_F64 = 1 };

/* distribution families of the element-wise site kernels.
 * NORMAL(p0=loc,p1=scale)        torch: torch/distributions/normal.py:88-103
 * BERNOULLI_LOGITS(p0=logits)    torch: torch/distributions/bernoulli.py:121-125
 * HALF_CAUCHY(p0=scale)          torch: torch/distributions/half_cauchy.py:74-83
 * LOG_NORMAL(p0=loc,p1=scale)    torch: torch/distributions/log_normal.py + transforms.ExpTransform
 * EXPONENTIAL(p0=rate)           torch: torch/distributions/exponential.py
 * HALF_NORMAL(p0=scale)          torch: torch/distributions/half_normal.py
 */
enum {
  PA_DIST_NORMAL = 0,
  PA_DIST_BERNOULLI_LOGITS = 1,
  PA_DIST_HALF_CAUCHY = 2,
  PA_DIST_LOG_NORMAL = 3,
  PA_DIST_EXPONENTIAL = 4,
  PA_DIST_HALF_NORMAL = 5,
  PA_DIST_GAMMA = 6,           /* p0 = concentration, p1 = rate          (torch gamma.py log_prob) */
  PA_DIST_BETA = 7,            /* p0 = concentration1, p1 = concentration0 (torch beta.py/dirichlet.py) */
  PA_DIST_POISSON = 8,         /* p0 = rate                               (torch poisson.py) */
  PA_DIST_BINOMIAL_LOGITS = 9, /* p0 = logits, p1 = total_count (pyro/distributions/torch.py:83-101) */
  /* the two halves of -KL(Normal(lq, sq) || Normal(lp, sp)) (torch/distributions/kl.py
   * _kl_normal_normal; pyro/infer/trace_mean_field_elbo.py:121-137 replaces log p - log q by it):
   *   KL_NORMAL_LOC  (value = lq, p0 = lp, p1 = sp):  -(lq - lp)^2 / (2 sp^2) - log sp
   *   KL_NORMAL_SCALE(value = sq, p0 = sp):            log sq + 1/2 - sq^2 / (2 sp^2)
   * so that an analytic-KL site is two ordinary entries of the multi-site launch. */
  PA_DIST_KL_NORMAL_LOC = 10,
  PA_DIST_KL_NORMAL_SCALE = 11,
  PA_DIST_COUNT = 12
};

typedef void* pa_stream_t; /* hipStream_t; NULL = the null stream */

/* A 2-D strided view [rows, cols]; strides in ELEMENTS, 0 = broadcast.
 * This is how ExpandedDistribution's stride-0 views (torch_distribution.py:483-488)
 * are handed over without materialising them. ptr may be NULL when unused. */
typedef struct {
  const void* ptr;
  int64_t stride_row;
  int64_t stride_col;
} pa_view2d;

int pa_abi_version(void);
const char* pa_last_error(void);
/* number of compute units of the current device (host-side query used for grid sizing) */
int pa_device_cu_count(void);

/* Measurement hook: the NEXT launch (on the calling thread) of the dominant kernel named by
 * `kernel_tag` is bracketed by hipEventRecord(ev_start) / hipEventRecord(ev_stop) on the
 * stream it is launched on (only that kernel, not its finalize / helper launches).
 * ev_start / ev_stop are hipEvent_t handles owned by the caller. Used by bench.py to
 * measure kernel time inside the timed region without a profiler. */
enum { PA_KERNEL_GLM = 1, PA_KERNEL_NUTS = 2, PA_KERNEL_LDA = 3, PA_KERNEL_SITE_SUM = 4 };
int pa_profile_bracket_next(int kernel_tag, void* ev_start, void* ev_stop);

/* ------------------------------------------------------------------------------------
 * RNG: counter-based Philox4x32-10. out[i] depends only on (seed, offset, i) so a draw
 * is reproducible for any launch geometry and shardable across ranks by offset.
 * Replaces the torch.normal()/_standard_normal draw inside rsample
 * (torch: torch/distributions/normal.py:83-86; called from
 * pyro/distributions/torch_distribution.py:48-52, pyro/poutine/runtime.py:345) and the
 * momentum draw of HMC (pyro/infer/mcmc/hmc.py:231-248).
 * f32: one Philox block -> 4 normals (2 Box-Muller pairs); f64: one block -> 2 normals.
 * If offset_dev != NULL the 64-bit offset is read from device memory (graph-replay safe)
 * and `offset` is added to it.
 * ---------------------------------------------------------------------------------- */
int pa_philox_normal(void* out, int64_t n, int dtype, uint64_t seed, uint64_t offset,
                     const uint64_t* offset_dev, pa_stream_t stream);
int pa_philox_uniform(void* out, int64_t n, int dtype, uint64_t seed, uint64_t offset,
                      const uint64_t* offset_dev, pa_stream_t stream);
/* *counter += inc, executed on the stream (keeps a device-resident Philox offset moving
 * under hipGraph replay). */
int pa_counter_add(uint64_t* counter, uint64_t inc, pa_stream_t stream);
/* End-of-step node of a captured SVI step (pyro/infer/svi.py:134-162 returns the loss as a Python
 * float, i.e. synchronises every step): counter[0] += inc, then the scalar at `src` (device,
 * `dtype`) is stored as a double to `host_value` and a NEW sequence number to `*host_seq` behind a
 * system-scope release fence.  host_value / host_seq are PINNED host memory (device mapped): the
 * host polls host_seq for a change instead of calling a stream synchronisation or a D2H copy.
 * counter: device uint64[2] = {Philox block counter, publish sequence}; the sequence number
 * written is ++counter[1].  counter may be NULL: the sequence number is then *host_seq + 1, read
 * back over the bus (ABI 1 behaviour; a PCIe round trip at the end of every step). */
int pa_publish_scalar(int dtype, const void* src, double* host_value, uint64_t* host_seq,
                      uint64_t* counter, uint64_t inc, pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * A latent with support (lower, inf) under a mean-field Normal guide (AutoNormal.forward,
 * pyro/infer/autoguide/guides.py:494-519: value = biject_to(support)(u), log-density of the Delta site =
 * transform.inv.log_abs_det_jacobian(value, u) summed over the event dims):
 *   value[r,c] = lower + exp(u[r,c]),   log_density[r] = - sum_c u[r,c]       (one launch)
 *   g_u[r,c]   = g_value[r,c] exp(u[r,c]) - g_log_density[r]                  (one launch; NULL gradient = 0)
 * u / value contiguous [rows, cols], cols = the product of the site's event dims.  log_density == NULL: the value
 * alone (a PARAMETER under a positive / greater-than constraint: transform_to(constraint)(unconstrained),
 * pyro/params/param_store.py:99-119 -- exp, mul, add and their duals per access otherwise).
 * ---------------------------------------------------------------------------------- */
int pa_exp_site_fwd(int dtype, const void* u, int64_t rows, int64_t cols, double lower, void* value,
                    void* log_density, pa_stream_t stream);
int pa_exp_site_bwd(int dtype, const void* value, const void* g_value, const void* g_log_density, int64_t rows,
                    int64_t cols, double lower, void* g_u, pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The score of a mean-field Normal guide site whose value is the guide's own reparameterised draw
 * z[p,c] = loc[c] + scale[c] eps[p,c]  (Trace_ELBO scores it with Normal.log_prob and differentiates through
 * z, loc and scale: pyro/infer/trace_elbo.py:142-160).  With eps fixed the three paths sum to d/d loc = 0,
 * d/d scale = -1/scale per element, so:
 *   sum_b partial[b] = coef * sum_{p,c} log Normal(z[p,c]; loc[c], scale[c])   (pa_meanfield_score_blocks(P, n)
 *                      partial sums, each reduced in double in a fixed order)
 *   gscale[c]        = -coef * P / scale[c]     = the TOTAL derivative of that sum w.r.t. scale[c]
 * z contiguous [P, n]; loc, scale, gscale [n].
 * ---------------------------------------------------------------------------------- */
int64_t pa_meanfield_score_blocks(int64_t P, int64_t n);
int pa_meanfield_score(int dtype, const void* z, const void* loc, const void* scale, int64_t P, int64_t n,
                       double coef, void* partial, void* gscale, pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The step gate: a captured SVI step enqueued BEFORE the host asks for it.
 * (pyro/infer/svi.py:134-162: step() returns the loss, so the reference's loop has the host between
 * every two steps; here the launch latency of step k+1 overlaps the execution of step k.)
 *
 * pa_gate launches a one-thread kernel as the first node of a captured step.  `gate` = device
 * int64[2] {number of the last step that ran, abort flag of the current replay}; `go` / `ack` = PINNED
 * host int64[1] each.  The kernel numbers its replay n = gate[0] + 1 and spins on *go:
 *   *go >= n          -> the step runs: gate[0] = n, gate[1] = 0;
 *   *go == -n         -> cancelled by the host;
 *   timeout_us passed -> given up (the host did not come back: it may be waiting on this very stream);
 * in the last two cases gate[1] = 1, *ack = n (system-scope release) and every gate-aware kernel of the
 * replay returns at once, so the replay changes nothing.  A replay launched the ordinary way finds
 * *go >= n already and passes through.
 *
 * pa_gate_scope(gate) .. pa_gate_scope(NULL): launches of gate-aware kernels made in between read
 * `gate[1]` at their start (guide draw, the plane-image GLM kernel, the chained tail).
 * pa_gate_stats: launches this library made since the scope opened / how many of them were
 * gate-aware -- a captured step may only be pre-enqueued when the two are equal (and torch launched
 * nothing). */
int pa_gate(const int64_t* go, int64_t* gate, int64_t* ack, int64_t timeout_us, pa_stream_t stream);
int pa_gate_scope(int64_t* gate);
int pa_gate_stats(int64_t* launches, int64_t* aware);
/* The LATE gate: the same node, placed in front of the step's chained tail instead of first.  What the step
 * launches before it -- its forward pass: the plane-image GLM kernel with the guide draw in its prologue --
 * runs as soon as the previous step's tail has finished, i.e. WHILE the host is still reading that step's
 * loss and calling step() again; it reads parameters, Philox position and data in the state that step left
 * and writes only scratch that every replay rewrites (records, the draw).  Everything that changes
 * persistent state (gradients, Adam, Philox position, loss hand-over) sits behind the gate and is given up
 * with it.  pa_gate_defer registers the gate at the start of a capture (instead of pa_gate + pa_gate_scope;
 * the scope opens when the node is emitted, pa_gate_scope(NULL) closes it); pa_gate_defer_stats: launches
 * made before the gate node, how many of them were NOT the plane-image GLM kernel (a capture is armable this
 * way only if none), whether the node was emitted at all (a step without a chained tail has no place for it). */
int pa_gate_defer(const int64_t* go, int64_t* gate, int64_t* ack, int64_t timeout_us);
int pa_gate_defer_stats(int64_t* pre, int64_t* pre_other, int* emitted);

/* ------------------------------------------------------------------------------------
 * Element-wise site kernels (SURVEY 8a rows a1,a3,a4,a5).
 * ---------------------------------------------------------------------------------- */

/* out[r,c] = log_prob(value[r,c]; p0[r,c], p1[r,c]); out is contiguous [rows, cols].
 * == site["fn"].log_prob(site["value"]) (pyro/poutine/trace_struct.py:264). */
int pa_dist_log_prob(int dist, int dtype, void* out, pa_view2d value, pa_view2d p0, pa_view2d p1,
                     int64_t rows, int64_t cols, pa_stream_t stream);

/* Fused log_prob -> scale_and_mask -> plate-dim sum (trace_struct.py:264-278,
 * pyro/distributions/util.py:311-328):
 *   out_rowsum[r] = sum_c  mask[r,c] ? scale * log_prob(...)[r,c] : 0
 * mask.ptr == NULL means "no mask"; mask elements are uint8 (torch.bool).
 *   out_total     = sum_r out_rowsum[r]           (optional, NULL to skip) -- the site's
 *                   log_prob_sum (trace_struct.py:278) without a separate reduction launch.
 * The reduction is deterministic (fixed tree, fp64 across threads).  Sites of up to 32768
 * elements (global latents) take ONE launch and need no workspace
 * (pa_dist_log_prob_sum_workspace returns 0); larger ones a two-stage reduction through
 * `workspace` (at least pa_dist_log_prob_sum_workspace(rows, cols) bytes). */
size_t pa_dist_log_prob_sum_workspace(int64_t rows, int64_t cols);
int pa_dist_log_prob_sum(int dist, int dtype, void* out_rowsum, void* out_total, pa_view2d value,
                         pa_view2d p0, pa_view2d p1, pa_view2d mask, double scale, int64_t rows,
                         int64_t cols, void* workspace, size_t workspace_bytes,
                         pa_stream_t stream);

/* Backward of both entry points above: given the upstream gradient g[r,c] (a strided view,
 * so a per-row gradient is stride_col = 0) writes, for every non-NULL output,
 *   d_x[r,c] = (mask ? scale : 0) * g[r,c] * d log_prob / d x   (x in value, p0, p1)
 * as contiguous [rows, cols]. The autograd dual of trace_struct.py:264-278. */
int pa_dist_log_prob_grad(int dist, int dtype, void* d_value, void* d_p0, void* d_p1, pa_view2d g,
                          pa_view2d value, pa_view2d p0, pa_view2d p1, pa_view2d mask, double scale,
                          int64_t rows, int64_t cols, pa_stream_t stream);

/* The same site sum and gradient for operands that broadcast along a MIDDLE dimension of the
 * site's frame (a plated latent under the particle plate: w[P, G, D] scored against mu[P, 1, D]),
 * which are not 2-D strided views: a frame of ndim <= 4 dims `sizes`, every operand with its own
 * element strides (0 = broadcast; NULL strides for an absent operand), fewer than 2^31 elements.
 *   out_total      = sum over the frame of (mask ? scale : 0) * log_prob
 *   d_x[flat i]    = (mask ? scale : 0) * g[0] * d log_prob / d x, contiguous over the frame
 * (trace_struct.py:248-288 with the expanded parameters of torch_distribution.py:483-488). */
size_t pa_dist_log_prob_sum_nd_workspace(void);
int pa_dist_log_prob_sum_nd(int dist, int dtype, void* out_total, int ndim, const int64_t* sizes,
                            const void* value, const int64_t* value_strides, const void* p0,
                            const int64_t* p0_strides, const void* p1, const int64_t* p1_strides,
                            const uint8_t* mask, const int64_t* mask_strides, double scale,
                            void* workspace, size_t workspace_bytes, pa_stream_t stream);
int pa_dist_log_prob_grad_nd(int dist, int dtype, void* d_value, void* d_p0, void* d_p1,
                             const void* g, int ndim, const int64_t* sizes, const void* value,
                             const int64_t* value_strides, const void* p0,
                             const int64_t* p0_strides, const void* p1, const int64_t* p1_strides,
                             const uint8_t* mask, const int64_t* mask_strides, double scale,
                             pa_stream_t stream);
/* out[A, B] = sum over R of in[A, R, B] (contiguous): brings an un-reduced gradient back to the
 * shape of an operand that was broadcast along leading (A = 1), middle or trailing (B = 1) dims --
 * the autograd dual of an expand (torch sum_to_size).  fp64 accumulation, fixed order.
 * A < 65536; workspace of pa_sum_to_nd_workspace(A, R, B) bytes (0 for small R). */
size_t pa_sum_to_nd_workspace(int64_t A, int64_t R, int64_t B);
int pa_sum_to_nd(int dtype, const void* in, void* out, int64_t A, int64_t R, int64_t B,
                 void* workspace, size_t workspace_bytes, pa_stream_t stream);
/* Two tensors of one shape (the two parameter gradients of a site) reduced by the same launches; results
 * bitwise those of two pa_sum_to_nd calls.  Workspace: twice pa_sum_to_nd_workspace(A, R, B). */
int pa_sum_to_nd_pair(int dtype, const void* in0, void* out0, const void* in1, void* out1, int64_t A,
                      int64_t R, int64_t B, void* workspace, size_t workspace_bytes, pa_stream_t stream);

/* Reparameterised Normal draw fused with the affine map (torch: normal.py:83-86):
 *   eps[r,c] = Philox normal (as pa_philox_normal with i = r*cols + c),
 *   out[r,c] = loc[r,c] + scale[r,c] * eps[r,c].
 * eps_out (contiguous, may be NULL) receives eps for the backward pass. */
int pa_normal_rsample(int dtype, void* out, void* eps_out, pa_view2d loc, pa_view2d scale,
                      int64_t rows, int64_t cols, uint64_t seed, uint64_t offset,
                      const uint64_t* offset_dev, pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * ELBO assembly over many SMALL sites in one launch.
 *
 * The reference scores every sample site separately (Trace.compute_log_prob,
 * trace_struct.py:248-288: log_prob, scale_and_mask, .sum() per site) and adds the sums up on the
 * host side of autograd (trace_elbo.py:82-112).  For global latents these are tiny tensors: the
 * cost is one launch per operation.  Here every small element-wise site of the model and guide
 * traces (priors, guide densities) plus the already-reduced terms of the big sites (entries of
 * kind PA_SITE_IDENTITY: log_prob(value) = value) are summed by ONE workgroup:
 *     out_total = coef_all * sum_e coef_e * sum_{r,c} mask_e ? log_prob_e(value; p0, p1)[r,c] : 0
 * (coef_e = sign in the ELBO times the site's scale), fp64 accumulation in a fixed order.
 * The backward entry point produces, for every requested operand of every entry, the gradient
 * ALREADY REDUCED to the operand's own broadcast shape: contiguous [rows or 1, cols or 1], a
 * stride of 0 over a dimension of size > 1 meaning "summed over it" -- the autograd duals of
 * expand + log_prob + scale_and_mask + sum in one launch (one workgroup per entry).
 * Limits: n <= PA_MULTI_MAX_ENTRIES per call (chain calls with accumulate = 1), rows*cols <=
 * PA_MULTI_MAX_ELEMS per entry; entries are host structs, copied into the kernel arguments.
 * ---------------------------------------------------------------------------------- */
#define PA_SITE_IDENTITY 100   /* log_prob(value) = value       (d/dvalue = 1) */
#define PA_SITE_NONE 101       /* log_prob(value) = 0: only a carrier of `extra_grad` */
#define PA_MULTI_MAX_ENTRIES 16
#define PA_MULTI_MAX_ELEMS 65536
/* `need` bits of pa_site_entry (backward) */
#define PA_NEED_VALUE 1
#define PA_NEED_P0 2
#define PA_NEED_P1 4
#define PA_VALUE_BY_CHAIN 16   /* this entry's value gradient is produced by another entry's chain */
typedef struct {
  int32_t dist;          /* PA_DIST_*, PA_SITE_IDENTITY or PA_SITE_NONE */
  int32_t need;          /* backward: PA_NEED_* | PA_VALUE_BY_CHAIN */
  int64_t rows, cols;
  pa_view2d value, p0, p1, mask; /* mask: uint8, ptr NULL = none; p1.ptr NULL when unused */
  double coef;
  void* d_value;         /* backward outputs (NULL when not wanted) */
  void* d_p0;
  void* d_p1;
  /* Several entries may score the SAME value tensor (a latent's prior in the model and its density
   * in the guide).  chain_next links them (index of the next entry, -1 = end): the workgroup of
   * the chain head writes the SUM of their value gradients into the head's d_value; the linked
   * entries carry PA_VALUE_BY_CHAIN.  All members must have the head's rows/cols and an un-reduced
   * value operand. */
  int32_t chain_next;
  int32_t reserved;
  /* A term of the total whose gradient w.r.t. this entry's value tensor is already known (the
   * fused GLM site returns its log-likelihood together with d ll / d w): added as
   * d_value[i] += g * coef_all * extra_coef * extra_grad[i]  (contiguous [rows, cols]; the value
   * operand must be un-reduced).  NULL = none. */
  const void* extra_grad;
  double extra_coef;
} pa_site_entry;
int pa_multi_log_prob_sum(int dtype, void* out_total, const pa_site_entry* entries, int n,
                          double coef_all, int accumulate, pa_stream_t stream);
/* g: device pointer to the upstream gradient of out_total (one element of `dtype`). */
int pa_multi_log_prob_grad(int dtype, const void* g, const pa_site_entry* entries, int n,
                           double coef_all, pa_stream_t stream);
/* Both of the above in ONE launch, for a caller that differentiates the total immediately
 * (Trace_ELBO.loss_and_grads: surrogate_loss.backward() right after the forward,
 * pyro/infer/trace_elbo.py:153-157): out_total as pa_multi_log_prob_sum, the operand gradients as
 * pa_multi_log_prob_grad with the upstream gradient g (NULL = 1). */
int pa_multi_log_prob_sum_grad(int dtype, void* out_total, const void* g,
                               const pa_site_entry* entries, int n, double coef_all,
                               int accumulate, pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Mean-field Normal guide: all latent sites drawn in one launch (AutoNormal,
 * pyro/infer/autoguide/guides.py:415-603: per site  scale = softplus(rho) (the
 * softplus_positive constraint), z = Normal(loc, scale).rsample() under P vectorised particles).
 *   eps_s[p,i] = Philox normal number p*n_s + i of the stream (seed, offset_s)   -- exactly the
 *                draws of pa_normal_rsample called site by site with the same offsets
 *   scale_s[i] = softplus(rho_s[i]),  z_s[p,i] = loc_s[i] + scale_s[i] * eps_s[p,i]
 * Backward (one launch, one workgroup per site), given d_z_s[P,n_s] and d_scale_s[n_s] (the
 * gradient w.r.t. the scale output used by the guide's own density; may be NULL) and d_loc_out_s:
 *   d_loc_s[i] = sum_p d_z[p,i] + d_loc_out[i]
 *   d_rho_s[i] = (sum_p d_z[p,i] eps[p,i] + d_scale[i]) * sigmoid(rho[i])
 * ---------------------------------------------------------------------------------- */
#define PA_MF_MAX_SITES 16
typedef struct {
  const void* loc;       /* [n] */
  const void* rho;       /* [n] unconstrained scale */
  void* z;               /* [P, n] out */
  void* scale;           /* [n] out */
  void* loc_out;         /* [n] out: copy of loc (the guide's density takes its loc from here, so
                            that every gradient of loc arrives through ONE backward) */
  void* eps;             /* [P, n] out (kept for the backward) */
  int64_t n;
  uint64_t offset;       /* Philox block offset of this site's draws */
  int32_t accumulate;    /* backward: 1 = d_loc / d_rho are ADDED to (they point into the
                            optimizer's flat gradient buffer), 0 = overwritten */
  int32_t reserved;
  const void* d_z;       /* backward inputs (each may be NULL = zero) / outputs */
  const void* d_scale;
  const void* d_loc_out;
  void* d_loc;
  void* d_rho;
} pa_mf_site;
int pa_meanfield_normal_sample(int dtype, const pa_mf_site* sites, int nsites, int64_t P,
                               uint64_t seed, const uint64_t* offset_dev, pa_stream_t stream);
int pa_meanfield_normal_sample_bwd(int dtype, const pa_mf_site* sites, int nsites, int64_t P,
                                   pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused plated Bernoulli-logits GLM likelihood: forward AND gradient in ONE pass over X.
 * Replaces, for an observed site  obs ~ Bernoulli(logits = w @ X^T + b)  under
 * pyro.plate("data", N) with P vectorised particles (pyro/infer/elbo.py:186-216):
 *   matmul -> Bernoulli.log_prob (= -BCE-with-logits, torch: bernoulli.py:121-125)
 *   -> scale_and_mask (pyro/distributions/util.py:311-328) -> .sum() (trace_struct.py:278)
 *   and the autograd duals of all of them (pyro/infer/trace_elbo.py:153-157).
 * Inputs : X[N,D] row-major f32, y[N] f32 (0/1), w[P,D] f32, b[P] f32 (NULL = 0),
 *          mask[N] uint8 (NULL = all true), scale (plate size / subsample size).
 * Outputs: ll[P]   = scale * sum_n mask_n (y_n l_pn - softplus(l_pn)),  l_pn = w_p.x_n + b_p
 *          gw[P,D] = scale * sum_n mask_n (y_n - sigmoid(l_pn)) x_n     (= d ll / d w)
 *          gb[P]   = scale * sum_n mask_n (y_n - sigmoid(l_pn))         (= d ll / d b)
 * Deterministic reduction (per-block partials in workspace, fp64 finalize).
 * Supported: f32, 1 <= D <= 128. Otherwise PA_ERR_UNSUPPORTED and the
 * caller uses the unfused path (matmul + pa_dist_log_prob_sum).
 *
 * Arithmetic variants, selected process-wide by pa_glm_set_variant:
 *   0 (default, automatic)
 *     - P <= 4 (incl. the reference's default num_particles = 1), D % 4 == 0, 8 <= D, aligned X, w:
 *       a vector-ALU streaming kernel (plain f32 fmaf chains; the pass is HBM-bound there and
 *       the matrix-core tiling only adds latency);
 *     - otherwise, D <= 64, D % 4 == 0 and a 16-byte aligned X: the contractions run on the bf16
 *       matrix cores with every f32 operand split exactly into three bf16 pieces and the six
 *       piece products of order >= 2^-16 accumulated in f32 (error O(2^-23) per product:
 *       f32-roundoff class, not bit-identical to an fmaf chain);
 *     - other layouts take variant 1.
 *   1 exact f32 MFMA (v_mfma_f32_32x32x2_f32): bit-for-bit an f32 fmaf chain per product sum.
 *   2 as 0 but never the few-particle kernel (the matrix-core kernels at every P).
 * ---------------------------------------------------------------------------------- */
int pa_glm_set_variant(int variant);
/* Chain rule of the two gradient outputs with the upstream gradient g[P] of ll[P] (the autograd
 * dual of the fused site), one launch: dw[p, :] = g[p] * gw[p, :] (W floats per particle: D, or
 * G*D for the grouped variant), db[p] = g[p] * gb[p].  dw / db may be NULL. */
int pa_glm_chain(const float* g, const float* gw, const float* gb, int64_t P, int64_t W, float* dw,
                 float* db, pa_stream_t stream);
size_t pa_glm_bernoulli_workspace(int64_t N, int64_t D, int64_t P);
int pa_glm_bernoulli_fwd_bwd(const float* X, const float* y, const float* w, const float* b,
                             const uint8_t* mask, double scale, int64_t N, int64_t D, int64_t P,
                             float* ll, float* gw, float* gb, void* workspace,
                             size_t workspace_bytes, pa_stream_t stream);

/* ---- the hierarchical GLM site on the plane image (BASELINE config 5) -------------------------
 * Rows sorted by group, cut into segments seg[nseg][3] = {row_begin, row_end, group} (device
 * int64; as pa_glm_bernoulli_grouped_fwd_bwd).  The image places every segment on a super-tile
 * (64-row) boundary: segment s owns the super-tiles [st_off[s], st_off[s+1]) (device int64[nseg+1],
 * st_off[s+1] - st_off[s] = ceil(rows_s / 64), nst_total = st_off[nseg]); the observations y are
 * stored with it in the same padded row order, so y is part of the image (re-pack when y changes).
 * pa_glm_bernoulli_grouped_planes_fwd_bwd = pa_glm_bernoulli_grouped_fwd_bwd (same outputs:
 * ll[P], gw[P,G,D], gb[P]; no mask) streaming the image instead of re-splitting X every step:
 * one workgroup per segment with w[:, group, :].  Replaces, per ELBO-gradient step, the gather
 * w[..., g_n, :] + matmul + Bernoulli.log_prob + scale_and_mask + sum and their autograd duals
 * (pyro/poutine/trace_struct.py:264-278, pyro/poutine/subsample_messenger.py:159-174 for the scale). */
size_t pa_glm_grouped_planes_bytes(int format, int64_t nst_total, int64_t D);
int pa_glm_pack_planes_grouped(int format, const float* X, const float* y, int64_t N, int64_t D,
                               const int64_t* seg, const int64_t* st_off, int64_t nseg,
                               int64_t nst_total, void* planes, size_t planes_bytes,
                               pa_stream_t stream);
/* The same image from rows that are NOT sorted by group -- SURVEY 8(d) config 5 as the reference
 * writes it: g = randint(0, G, (N,)), logits = (w[..., g, :] * X).sum(-1) + b, an advanced-index
 * gather + product + reduction + Bernoulli.log_prob + sum and their autograd duals
 * (torch: aten index / index_backward; pyro/poutine/trace_struct.py:264-278).
 * pa_group_rows_build: a STABLE counting sort of the row indices by group id (integer work,
 * bit-exact: rows[offsets[k] .. offsets[k+1]) = the n with g[n] == k in ascending n = numpy
 * argsort(g, kind="stable"); offsets = exclusive cumulative bincount), device int64 outputs
 * offsets[G+1], rows[N]; *n_out_of_range (device int64) = number of ids outside [0, G) (torch raises
 * IndexError for those; the caller does too).  G <= 16384, N < 2^31, else PA_ERR_UNSUPPORTED
 * (pa_group_rows_workspace returns 0).
 * pa_glm_pack_planes_grouped_rows: pa_glm_pack_planes_grouped reading X[row_of[i]], y[row_of[i]]
 * for image row i (seg / st_off describe the SORTED order; row_of = rows above; NULL = identity).
 * No sorted copy of X is made; everything downstream (pa_glm_bernoulli_grouped_planes_fwd_bwd) is
 * unchanged because the likelihood sum does not depend on the row order. */
size_t pa_group_rows_workspace(int64_t N, int64_t G);
int pa_group_rows_build(const int64_t* g, int64_t N, int64_t G, int64_t* offsets, int64_t* rows,
                        int64_t* n_out_of_range, void* workspace, size_t workspace_bytes,
                        pa_stream_t stream);
int pa_glm_pack_planes_grouped_rows(int format, const float* X, const float* y, const int64_t* row_of,
                                    int64_t N, int64_t D, const int64_t* seg, const int64_t* st_off,
                                    int64_t nseg, int64_t nst_total, void* planes, size_t planes_bytes,
                                    pa_stream_t stream);
size_t pa_glm_bernoulli_grouped_planes_workspace(int64_t nseg, int64_t P);
int pa_glm_bernoulli_grouped_planes_fwd_bwd(int format, const void* planes, const float* w,
                                            const float* b,
                                            double scale, int64_t N, int64_t D, int64_t P, int64_t G,
                                            const int64_t* seg, const int64_t* st_off, int64_t nseg,
                                            const int64_t* group_seg_off, int64_t nst_total,
                                            float* ll, float* gw, float* gb, void* workspace,
                                            size_t workspace_bytes, pa_stream_t stream);

/* The same pass with the design matrix kept in HBM as its exact 3-way bf16 decomposition
 * (x = x1 + x2 + x3, the split the matrix-core kernel of variant 0 performs on the fly), packed ONCE
 * per data set by pa_glm_pack_planes into a tile image (per 32-row tile three [32][32] bf16 planes,
 * rows >= N and columns >= D zero, padded to whole 128-row groups) that pa_glm_bernoulli_planes_fwd_bwd
 * streams HBM -> LDS by DMA.  X never changes between ELBO-gradient steps
 * (pyro/infer/svi.py:134-162 passes the same data tensor every step), so the split, the staging
 * registers and the LDS writes of the on-the-fly kernel leave the per-step work; the arithmetic and
 * the outputs are those of pa_glm_bernoulli_fwd_bwd variant 0 (no mask argument: masked plates take
 * the entry above).  D <= 32 (format F16X2: D <= 128, with 2 or 4 feature tiles of 32 columns per row tile and a
 * 1-KiB trailer of 128 column maxima / exponents, csrc/glm_planes16d.h); any P (64 particles per pass over the image).
 * pa_glm_planes_tune(ring depth code 3..13, workgroups per CU; 0 = default) is a measurement knob (11: the 2 x 2 wave
 * geometry whatever P -- by default, from ~48 row tiles per workgroup on (N >= 393 k on 256 CUs), 65..128 particles run
 * 128 and more run 256 per pass over the image; 12: at most 128 per pass, at every N; 13: the default's choice at every N).
 *
 * Two image formats (`format`, the same value when an image is sized, packed and used):
 *   PA_GLM_PLANES_BF16X3  three bf16 planes, x = x1 + x2 + x3 EXACTLY; six piece products per
 *                         element product (dropped terms O(2^-24)); 6 B per element;
 *   PA_GLM_PLANES_F16X2   two f16 planes of X with every COLUMN scaled by its own power of two,
 *                         chosen from the column's max |x| (found on the device at pack time, kept in
 *                         the image's trailer: u32[32] maxima, i32[32] exponents), x ~= x1 + x2 to
 *                         2^-22 relative (elements below 2^-13 of their column's maximum: 2^-40 of
 *                         that maximum, absolute) -- columns of very different magnitude keep their
 *                         precision;
 *                         three piece products; W and the gradient operand are split the same way
 *                         inside the kernel with per-particle power-of-two scales.  Error per logit
 *                         <= 3 * 2^-22 sum_d |x_d w_d| -- inside the 32 * 2^-24 bound of an f32 FMA
 *                         chain over 32 features; 4 B per element (the f32 matrix's own size), 13
 *                         instead of 25 matrix instructions per 32x32 tile (csrc/glm_planes16.h). */
#define PA_GLM_PLANES_BF16X3 0
#define PA_GLM_PLANES_F16X2 1
size_t pa_glm_planes_bytes(int format, int64_t N, int64_t D);
int pa_glm_pack_planes(int format, const float* X, int64_t N, int64_t D, void* planes,
                       size_t planes_bytes, pa_stream_t stream);
int pa_glm_planes_tune(int ring_depth, int blocks_per_cu);
/* 0 (default): the separate finalize launch; 1: pa_glm_bernoulli_planes_fwd_bwd reduces its
 * per-workgroup partial records INSIDE the kernel (two levels of last-arriver sums, bit-identical
 * results).  Measured slower at large plates (one workgroup per sum has too little memory-level
 * parallelism): a developer switch.  The workspace holds the fp64 level-1 partials in both modes
 * (pa_glm_bernoulli_planes_workspace). */
int pa_glm_planes_finalize_mode(int in_kernel);
/* Measurement hook: later launches of the plane-image kernel record {earliest workgroup entry,
 * latest workgroup exit} on the device's 100 MHz wall clock into two_u64[0..1] with min / max
 * atomics (initialise to {UINT64_MAX, 0}); NULL = off.  The kernel's duration inside a captured
 * hipGraph, where HIP events do not time their node (bench.py roofline.kernel_ms). */
int pa_glm_planes_stamps(void* two_u64);
/* The label-linear part of the site's log-likelihood is a dot product with two DATA moments,
 *   sum_n (y_n - 1/2) (x_n . w_p + b_p) = c . w_p + c0 b_p,   c[d] = sum_n (y_n - 1/2) x[n,d],
 *                                                             c0   = sum_n (y_n - 1/2):
 * pa_glm_label_moments computes moments[33] = {c[0..31] (0 beyond D), c0} in float64 with a fixed
 * summation order, once per (X, y) (the host caches it beside the image).  Handed to
 * pa_glm_bernoulli_planes_fwd_bwd (format F16X2, default tuning) it replaces one fma per (row, particle)
 * element of the kernel's loop; NULL = the kernel sums the term itself.  The gradient never takes
 * this route.  (pyro/distributions/torch.py Bernoulli.log_prob = -BCE-with-logits, restated.) */
size_t pa_glm_label_moments_workspace(int64_t N);
int pa_glm_label_moments(const float* X, const float* y, int64_t N, int64_t D, double* moments,
                         void* workspace, size_t workspace_bytes, pa_stream_t stream);
size_t pa_glm_bernoulli_planes_workspace(int64_t N, int64_t D, int64_t P);
int pa_glm_bernoulli_planes_fwd_bwd(int format, const void* planes, const float* y, const float* w,
                                    const float* b, double scale, int64_t N, int64_t D, int64_t P,
                                    float* ll, float* gw, float* gb, void* workspace,
                                    size_t workspace_bytes, const double* moments, pa_stream_t stream);

/* Hierarchical variant (BASELINE config 5: logit_n = x_n . w_{g(n)} + b with per-group weights
 * w[P,G,D] under pyro.plate("groups", G)): rows of X are SORTED BY GROUP; the caller describes
 * the work as segments seg[nseg][3] = {row_begin, row_end, group} (device int64; each segment
 * lies inside one group, <= max_seg_rows rows, segments of a group are contiguous) and
 * group_seg_off[G+1] (device int64: first segment of every group).  One workgroup per segment
 * runs the same MFMA pipeline as pa_glm_bernoulli_fwd_bwd with that group's weights.
 * Outputs: ll[P], gb[P] as above, gw[P,G,D] = d ll / d w (groups without rows get 0). */
size_t pa_glm_bernoulli_grouped_workspace(int64_t nseg, int64_t D, int64_t P);
int pa_glm_bernoulli_grouped_fwd_bwd(const float* X, const float* y, const float* w,
                                     const float* b, const uint8_t* mask, double scale, int64_t N,
                                     int64_t D, int64_t P, int64_t G, const int64_t* seg,
                                     int64_t nseg, const int64_t* group_seg_off,
                                     int64_t max_seg_rows, float* ll, float* gw, float* gb,
                                     void* workspace, size_t workspace_bytes, pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * HMC / NUTS.
 * ---------------------------------------------------------------------------------- */

/* The two momentum/position updates of _single_step_verlet (pyro/ops/integrator.py:45-65)
 * for C chains of dimension D, row-major [C,D]; step[c] is per-chain (step_stride 0/1):
 *   pa_leapfrog_kick_drift : r <- r - 0.5*eps*grad ; z <- z + eps * (inv_mass (.) r)
 *   pa_leapfrog_kick       : r <- r - 0.5*eps*grad
 * inv_mass is diagonal, [D] (im_stride_row = 0) or per chain [C,D]
 * (BlockMassMatrix.kinetic_grad, pyro/infer/mcmc/adaptation.py:328-347). */
int pa_leapfrog_kick_drift(int dtype, void* z, void* r, const void* grad, const void* inv_mass,
                           int64_t im_stride_row, const void* step, int64_t step_stride, int64_t C,
                           int64_t D, pa_stream_t stream);
int pa_leapfrog_kick(int dtype, void* r, const void* grad, const void* step, int64_t step_stride,
                     int64_t C, int64_t D, pa_stream_t stream);

/* One full NUTS transition (pyro/infer/mcmc/nuts.py:367-522, _build_tree :250-365,
 * _build_basetree :197-248, _is_turning :184-195) for C independent chains on the
 * closed-form potential U(z) = 0.5 z^T Lambda z (BASELINE config 3), one wavefront per
 * chain, iterative tree doubling with wave-uniform control flow, all randomness from
 * Philox keyed by (seed, chain_offset + chain, transition t, draw kind, tree position).
 * State (in/out): z[C,D], pe[C] (potential energy at z), grad[C,D] (Lambda z).
 * Params: Lambda[D,D] symmetric row-major; inv_mass[C,D] diagonal per chain;
 *         step[C]; max_tree_depth <= 10; use_multinomial (1) or slice sampling (0).
 * Outputs per chain: accept_prob[C] (= sum_accept_probs/num_proposals, nuts.py:510),
 *         n_leapfrog[C] int32, depth[C] int32, diverging[C] int32, accepted[C] int32.
 * Supported: D <= 128. */
int pa_nuts_gaussian_transition(int dtype, void* z, void* pe, void* grad, const void* Lambda,
                                const void* inv_mass, const void* step, int64_t C, int64_t D,
                                int max_tree_depth, int use_multinomial, uint64_t seed, uint64_t t,
                                uint64_t chain_offset, void* accept_prob, int32_t* n_leapfrog, int32_t* depth,
                                int32_t* diverging, int32_t* accepted, pa_stream_t stream);

/* Reasonable step size per chain for the closed-form Gaussian potential (pyro/infer/mcmc/hmc.py:
 * 170-229: double / halve the step until the one-step acceptance probability crosses 0.8), every
 * chain running its own loop in ONE launch: step[C] is read as the starting point and overwritten.
 * Momenta are keyed Philox draws (seed, key + trial index, chain_offset + chain). */
int pa_nuts_gaussian_find_step(int dtype, const void* z, const void* pe, const void* grad,
                               const void* Lambda, const void* inv_mass, void* step, int64_t C,
                               int64_t D, uint64_t seed, uint64_t key, uint64_t chain_offset,
                               double min_step, double max_step, double direction_threshold,
                               pa_stream_t stream);

/* Persistent form of the same kernel: num_transitions consecutive transitions t0, t0+1, ... per
 * launch, each wavefront staying with its chain (the data-dependent tree sizes of the individual
 * transitions average out over the launch instead of making every launch as long as its longest
 * tree), with the per-transition part of the warm-up adaptation done in-kernel, per chain:
 *   da_state[C][5] = {x_avg, g_avg, t, prox_center, x_t}: DualAveraging.step on
 *       H = target_accept - accept_prob, step[c] <- exp(x_t)  (pyro/ops/dual_averaging.py:55-78,
 *       pyro/infer/mcmc/adaptation.py:115-121); NULL = fixed step size;
 *   welford[C][2][D] = {mean, m2}: WelfordCovariance.update of z, diagonal, the sample count is
 *       welford_n0 + k (pyro/ops/welford.py:27-38); NULL = off;
 * (window-end events -- new mass matrix, step-size search -- stay on the host between launches.)
 * samples[num_transitions][C][D] (NULL = not recorded) receives z after every transition;
 * mean_accept[C] is the running mean of accept_prob with mean_n0 transitions already in it;
 * counters[3][C] (int64, accumulated): leapfrog steps, tree depths, accepted proposals (the last
 * only with count_accepts); div_flags[num_transitions][C] (int8, with count_accepts, NULL = off).
 * accept_prob ... accepted report the LAST transition.  f32, D <= 128: Lambda's columns are held
 * in VGPRs (one workgroup = one wave = one chain); otherwise Lambda is staged in LDS. */
int pa_nuts_gaussian_run(int dtype, void* z, void* pe, void* grad, const void* Lambda,
                         const void* inv_mass, void* step, int64_t C, int64_t D,
                         int max_tree_depth, int use_multinomial, uint64_t seed, uint64_t t0,
                         int64_t num_transitions, uint64_t chain_offset, void* da_state,
                         double target_accept, void* welford, int64_t welford_n0, void* samples,
                         void* mean_accept, int64_t mean_n0, int64_t* counters, int count_accepts,
                         int8_t* div_flags, void* accept_prob, int32_t* n_leapfrog, int32_t* depth,
                         int32_t* diverging, int32_t* accepted, pa_stream_t stream);
/* Test hook: force_lds != 0 makes the f32 path use the LDS-resident Lambda variant too. */
int pa_nuts_gaussian_set_variant(int force_lds);

/* NUTS for ARBITRARY potentials, vectorised over chains: the
 * tree logic of pyro/infer/mcmc/nuts.py:184-522 as a device-resident per-chain state machine.
 * The caller evaluates the potential energy and its gradient for all chains at the cursor
 * positions zq[C,D] (a fused likelihood kernel or torch autograd of a chain-batched model,
 * replacing potential_grad, pyro/ops/integrator.py:68-94); between two evaluations ONE
 * pa_nuts_tree_advance launch performs, per chain: second half-kick, leaf energies, merges
 * with parked sibling subtrees (U-turn / divergence / proposal draws), doubling bookkeeping and
 * the first half-kick + drift of the chain's next leapfrog.  Protocol per transition t:
 *     pa_nuts_tree_begin(...)                      -> zq, rq hold the first cursor
 *     do { (peq, gq) = U(zq), dU/dz(zq);  pa_nuts_tree_advance(...); } while (*n_active > 0)
 * Chains that finished are inactive (cursor frozen); on exit (z, pe, grad) hold each chain's
 * next state and accept_prob / n_leapfrog / depth / diverging / accepted its statistics.
 * inv_mass: diagonal, [D] (im_stride_row = 0) or per chain [C,D] (im_stride_row = D).
 * Randomness: the keyed Philox contract of pa_nuts_gaussian_transition with the chain id
 * chain_offset + c (so a rank of a chain-sharded job passes its first global chain index).
 * Supported: D <= 2048. workspace >= pa_nuts_tree_workspace bytes, preserved between calls
 * of one transition. n_active: device int32, overwritten by every advance call. */
size_t pa_nuts_tree_workspace(int dtype, int64_t C, int64_t D, int max_tree_depth);
int pa_nuts_tree_begin(int dtype, const void* z, const void* pe, const void* grad, void* zq,
                       void* rq, const void* inv_mass, int64_t im_stride_row, const void* step,
                       int64_t C, int64_t D, int max_tree_depth, int use_multinomial,
                       uint64_t seed, uint64_t t, uint64_t chain_offset, void* workspace,
                       size_t workspace_bytes, pa_stream_t stream);
int pa_nuts_tree_advance(int dtype, void* z, void* pe, void* grad, void* zq, void* rq,
                         const void* gq, const void* peq, const void* inv_mass,
                         int64_t im_stride_row, const void* step, int64_t C, int64_t D,
                         int max_tree_depth, int use_multinomial, uint64_t seed, uint64_t t,
                         uint64_t chain_offset, void* accept_prob, int32_t* n_leapfrog,
                         int32_t* depth, int32_t* diverging, int32_t* accepted, int32_t* n_active,
                         void* workspace, size_t workspace_bytes, pa_stream_t stream);
/* pa_nuts_tree_advance with the transition index read from device memory (*t_dev): the launch
 * can sit in a hipGraph that is replayed for every leapfrog step of every transition (a captured
 * scalar argument would freeze t, i.e. the Philox keys). */
int pa_nuts_tree_advance_tdev(int dtype, void* z, void* pe, void* grad, void* zq, void* rq,
                              const void* gq, const void* peq, const void* inv_mass,
                              int64_t im_stride_row, const void* step, int64_t C, int64_t D,
                              int max_tree_depth, int use_multinomial, uint64_t seed,
                              const uint64_t* t_dev, uint64_t chain_offset, void* accept_prob,
                              int32_t* n_leapfrog, int32_t* depth, int32_t* diverging,
                              int32_t* accepted, int32_t* n_active, void* workspace,
                              size_t workspace_bytes, pa_stream_t stream);

/* ASYNCHRONOUS CHAINS: a span of K transitions per chain without a host decision inside.
 * Replaces, per chain and per transition, what the reference does on the host between two trees:
 * HMC._after_transition / WarmupAdapter.step without its window-end branch
 * (pyro/infer/mcmc/hmc.py:425-438, adaptation.py:166-185: DualAveraging.step on H = target -
 * accept_prob, pyro/ops/dual_averaging.py:55-78; diagonal WelfordCovariance.update,
 * pyro/ops/welford.py:27-38), the running mean of the acceptance probability, the
 * leapfrog / depth / accept counters, the store of the draw (api.py:405-651 collects it) -- and
 * then NUTS.sample's prologue for the chain's NEXT transition (nuts.py:367-434), in the launch that
 * finished the tree.  No chain waits for the slowest tree of a transition; every potential
 * evaluation serves C live cursors.  Philox keys are (transition, slot, chain): the chains are
 * those of the lock-step protocol above, bit for bit.
 *
 *     ctl (device int64[8]) = {t0, K, mean_n0, welford_n0, flags, samples, div_flags, row0}
 *     pa_nuts_tree_run_begin(...)       transition t0 of every chain begins; tc[c] = 0, *n_done = 0
 *     do { (peq, gq) = U(zq), dU/dz(zq);  pa_nuts_tree_run_advance(...); } while (*n_done < C)
 *
 * ctl lives in device memory so that ONE captured hipGraph of rounds serves every span:
 *   flags   PA_NUTS_RUN_ADAPT_STEP | PA_NUTS_RUN_WELFORD | PA_NUTS_RUN_COUNT_ACCEPTS;
 *   samples device address of T[rows][C][D] (0 = do not store): transition t0 + k of chain c is
 *           stored at row row0 + k; div_flags: device address of int8[rows][C] (0 = none), written
 *           when COUNT_ACCEPTS is set;
 *   mean_n0 / welford_n0: transitions already averaged into mean_accept / draws already in welford.
 * da_state[C,5] = {x_avg, g_avg, t, prox_center, x_t}, welford[C,2,D] = {mean, m2}, mean_accept[C],
 * counters[3,C] = {leapfrogs, tree depths, accepted}, step[C] (rewritten by the dual averaging),
 * tc[C] int32 (transitions completed in the span).  The statistics outputs hold each chain's LAST
 * finished transition.  done_flag (may be NULL): set to 1 by the last chain to complete its span --
 * passed as a step gate's abort word (pa_gate_scope) it makes the gate-aware kernels of the rounds
 * still queued return at once.  inv_mass is fixed during a span. */
#define PA_NUTS_RUN_ADAPT_STEP 1
#define PA_NUTS_RUN_WELFORD 2
#define PA_NUTS_RUN_COUNT_ACCEPTS 4
int pa_nuts_tree_run_begin(int dtype, const void* z, const void* pe, const void* grad, void* zq,
                           void* rq, const void* inv_mass, int64_t im_stride_row, const void* step,
                           int64_t C, int64_t D, int max_tree_depth, int use_multinomial,
                           uint64_t seed, uint64_t chain_offset, const int64_t* ctl, int32_t* tc,
                           int32_t* n_done, int64_t* done_flag, void* workspace,
                           size_t workspace_bytes, pa_stream_t stream);
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
                             size_t workspace_bytes, pa_stream_t stream);
/* COMPACTED rounds.  Late in a span few chains are still building trees (per-chain step sizes differ; the
 * reference's chain processes simply finish at different times, api.py:239-351); evaluating the potential
 * for all C cursors then serves mostly finished chains.  pa_nuts_tree_compact lists the chains still active
 * (ascending; -1 pads) in slot2chain[n_slots] and gathers their cursors into zq_slot[n_slots, D];
 * *n_placed = number of slots filled (the caller picks n_slots >= the number of active chains).  From then
 * on the caller evaluates the potential at zq_slot and passes (slot2chain, zq_slot, n_slots) to
 * pa_nuts_tree_run_advance: peq / gq are slot-indexed, the launch has one workgroup per slot, a chain writes
 * its next cursor to its slot row too.  NULL / NULL: slot == chain, n_slots ignored (the full round). */
int pa_nuts_tree_compact(int dtype, const void* zq, int64_t C, int64_t D, int max_tree_depth,
                         int32_t* slot2chain, void* zq_slot, int64_t n_slots, int32_t* n_placed,
                         int n_sites, const int32_t* site_off, const int32_t* site_len,
                         void* workspace, size_t workspace_bytes, pa_stream_t stream);
/* A FLAT model's potential assembled inside the tree kernel.  When every latent site of the model is scored
 * by one of the fused families at parameters that do not depend on other latents, and what remains are
 * observed sites with their own fused kernels (the Bernoulli-logits GLM site over the latent weights / bias),
 * the reference's potential (pyro/infer/mcmc/util.py:264-286: the conditioned model run under the handlers,
 * trace.log_prob_sum, autograd backward -- ~25 operators per evaluation) is
 *     U(u) = -( ll_ext[slot] + sum_sites sum_j [ log p_s(v_j; p0, p1) + log |dv_j/du_j| ] ),  v = T_s(u),
 * T_s the identity (transform 0) or v = lower + exp(u) (transform 1: biject_to of a positive /
 * greater-than support), and its gradient -( (d log p_s/dv + g_ext) dv/du + d log|dv/du|/du ).
 * pa_nuts_tree_run_advance_direct is pa_nuts_tree_run_advance (float32) computing (peq, gq) that way:
 * n_sites sites tile the flat coordinates [0, D) in ascending order (site_off / site_len); site k has family
 * site_dist[k] (PA_DIST_NORMAL, _HALF_CAUCHY, _LOG_NORMAL, _EXPONENTIAL, _HALF_NORMAL, _GAMMA), parameters
 * site_p0/p1[k] (device, element j at p[j * stride]; NULL: unused) and -- when an external kernel scored an
 * observed site against it -- site_g_ext[k] = d ll_ext / d v [n_slots, len] (else NULL); ll_ext[n_slots] (or
 * NULL).  The cursor handed to the external kernels, zq_pack, is SITE-MAJOR: site k's block [n_slots, len_k]
 * starts at element n_slots * site_off[k] (the GLM kernel reads weights [P, D_w] and bias [P] in place);
 * pa_nuts_tree_compact with the same (n_sites, site_off, site_len) fills it (slot2chain NULL and
 * n_slots == C: the full round, identity map).  n_sites == 0 there: the row-major [n_slots, D] buffer of a
 * potential that takes the flat state.
 * HIERARCHICAL priors (ABI 7): a parameter of site k may be the constrained VALUE of another latent site q of the
 * same chain (w ~ Normal(mu, tau) with mu / tau latent): site_p0[k] == NULL with site_s0[k] = -(q + 1) (likewise
 * p1 / s1); element j of site k takes element j % site_len[q] of site q (a parent broadcast over leading plate
 * dims; site_len[k] must be a multiple of site_len[q]); d log p_k / d parameter is added to q's gradient inside
 * the kernel, in a fixed order.  D <= 512 as for the direct form in general. */
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
                                    size_t workspace_bytes, pa_stream_t stream);
/* The same potential on its own: (U, dU/du) of the flat model at n_slots cursors of a site-major zq_pack
 * (float32, D <= 512; arguments as above), pe_out[n_slots], grad_out[n_slots, D] row-major.  Replaces one
 * evaluation of the reference's potential_fn + autograd at given unconstrained points
 * (pyro/infer/mcmc/util.py:264-286, pyro/ops/integrator.py:68-94); the tree kernel runs this arithmetic in
 * registers, this entry writes it out (initial state of a chain, step-size search, parity tests against
 * the reference's potential_fn values). */
int pa_nuts_direct_potential(const void* zq_pack, int64_t n_slots, int64_t D, int n_sites,
                             const int32_t* site_off, const int32_t* site_len, const int32_t* site_dist,
                             const int32_t* site_transform, const double* site_lower,
                             const void* const* site_p0, const int64_t* site_s0, const void* const* site_p1,
                             const int64_t* site_s1, const void* const* site_g_ext, const void* ll_ext,
                             void* pe_out, void* grad_out, pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Enumerated Categorical-Categorical mixture factor of examples/lda.py:53-71 under
 * TraceEnum_ELBO:
 *   out = sum_{d < B} sum_{w < Wd} logsumexp_t( log_theta[d,t] + log_phi[t, words[w,d]] )
 * plus its gradient (the adjoint of pyro/ops/einsum/torch_log.py:14-55 and the plate
 * products of pyro/ops/contract.py:79-160):
 *   post[t | w,d] = softmax_t(...),  g_theta[d,t] = sum_w post,  g_phi[t,v] = sum_{w,d: words=v} post
 * words: int64 [Wd,B] (layout of examples/lda.py: words plate dim -2, documents dim -1),
 * log_theta: [B,T], log_phi: [T,V]. T <= 64.
 * out_doc[B] receives the per-document sums (deterministic); g_phi is accumulated in
 * per-workgroup LDS histograms (LDS float atomics: summation order within a workgroup is
 * not fixed, so g_phi is reproducible to rounding, not bitwise) and reduced over workgroups
 * by a finalize kernel through `workspace` (>= pa_lda_factor_workspace bytes).
 * A word id outside [0,V) is a support violation: the result is unspecified but memory safe.
 * ---------------------------------------------------------------------------------- */
size_t pa_lda_factor_workspace(int dtype, int64_t B, int64_t T, int64_t V);
int pa_lda_factor_fwd_bwd(int dtype, const int64_t* words, const void* log_theta,
                          const void* log_phi, int64_t Wd, int64_t B, int64_t T, int64_t V,
                          void* out_doc, void* g_theta, void* g_phi, void* workspace,
                          size_t workspace_bytes, pa_stream_t stream);

/* The same factor through an INVERTED INDEX of the corpus, built once (the data of
 * examples/lda.py:133-141 does not change between steps): the scatter into g_phi becomes a gather
 * over each word's document list -- no atomics, every sum in a fixed order (g_phi bitwise
 * reproducible), ~8x less time per step at 1e5 documents x 64 words.
 * Index image (int32): header {magic, Wd, B, V, ntasks, task capacity, Wd*B, 0}, off[V+1],
 * first_task[V+1], task_v/task_start/task_len[capacity], docs[Wd*B]:
 *   docs[off[v] .. off[v+1]) = the document d of every pair (w, d) with words[w,d] == v, in
 *   ascending order of w*B + d (a stable counting sort; ids outside [0,V) are filed under 0 and
 *   flagged by the step); tasks cut every word's list into segments of <= 2048 pairs.
 * pa_lda_index_bytes returns 0 when no index exists for the shape (Wd*B >= 2^31, V > 15360):
 * callers keep the atomic entry point above for those. */
size_t pa_lda_index_bytes(int64_t Wd, int64_t B, int64_t V);
size_t pa_lda_index_workspace(int64_t Wd, int64_t B, int64_t V);
int pa_lda_build_index(const int64_t* words, int64_t Wd, int64_t B, int64_t V, void* index,
                       size_t index_bytes, void* workspace, size_t workspace_bytes,
                       pa_stream_t stream);
size_t pa_lda_factor_indexed_workspace(int dtype, int64_t Wd, int64_t B, int64_t T, int64_t V);
int pa_lda_factor_indexed_fwd_bwd(int dtype, const int64_t* words, const void* index,
                                  size_t index_bytes, const void* log_theta, const void* log_phi,
                                  int64_t Wd, int64_t B, int64_t T, int64_t V, void* out_doc,
                                  void* g_theta, void* g_phi, void* workspace,
                                  size_t workspace_bytes, pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Dirichlet log-density rows (torch/distributions/dirichlet.py log_prob; the simplex-valued sites of
 * examples/lda.py:45-60):  out[r] = sum_k xlogy(c[r,k]-1, x[r,k]) + lgamma(sum_k c[r,k]) - sum_k lgamma(c[r,k]).
 * value / concentration: 2-D strided views [rows, K] (row stride 0 = one concentration vector for
 * all rows).  Gradients (g[rows] upstream, outputs contiguous [rows, K], NULL = not wanted):
 *   d_value[r,k] = g[r] (c-1)/x,  d_concentration[r,k] = g[r] (log x + psi(sum c) - psi(c_k)).
 * ---------------------------------------------------------------------------------- */
int pa_dirichlet_log_prob(int dtype, void* out, pa_view2d value, pa_view2d concentration,
                          int64_t rows, int64_t K, pa_stream_t stream);
int pa_dirichlet_log_prob_grad(int dtype, const void* g, pa_view2d value, pa_view2d concentration,
                               int64_t rows, int64_t K, void* d_value, void* d_concentration,
                               pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * One elimination step of the plated sum-product in log space (SURVEY 8a row a16, 8f rank 4;
 * pyro/ops/contract.py:79-160 _contract_component, pyro/ops/einsum/torch_log.py:14-55):
 *     out[kept] = logsumexp over frame dim `rdim` of  sum_k term_k[frame]
 * `sizes[ndim]` is the frame (the union of the terms' dims after alignment), every term a strided
 * view over it (stride 0 = the term does not depend on that dim: nothing is materialised), `out`
 * contiguous over the frame without `rdim`.  An all -inf column gives -inf.
 * The gradient entry point writes the one tensor every term's gradient is a reduction of:
 *     G[frame] (contiguous) = g_out[kept] * exp( sum_k term_k[frame] - out[kept] )
 * (the posterior weights of the eliminated variable; d out / d term_k = G summed over the dims the
 * term does not have: pa_sum_to_nd).
 * ---------------------------------------------------------------------------------- */
#define PA_LSE_MAX_TERMS 4
#define PA_LSE_MAX_DIMS 6
typedef struct {
  const void* ptr;
  int64_t strides[PA_LSE_MAX_DIMS];   /* in elements, over the frame's dims */
} pa_lse_term;
int pa_logsumexp_terms(int dtype, void* out, int nterms, const pa_lse_term* terms, int ndim,
                       const int64_t* sizes, int rdim, pa_stream_t stream);
int pa_logsumexp_terms_grad(int dtype, void* G, const void* g_out, const void* out, int nterms,
                            const pa_lse_term* terms, int ndim, const int64_t* sizes, int rdim,
                            pa_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Flat multi-tensor Adam / ClippedAdam step (pyro/optim/optim.py:117-155,
 * pyro/optim/clipped_adam.py:52-100). One launch over the flat parameter buffer;
 * `step_dev` points to TWO device-resident int64: [0] the step counter, advanced by the kernel
 * itself (the last workgroup to finish does it: one launch per step, graph safe), [1] the ticket
 * counter of that hand-off, which must be 0 on entry and is 0 again on exit.
 * clipped = 0: torch.optim.Adam update; clipped = 1: ClippedAdam (element-wise gradient clamp
 * to [-clip_norm, clip_norm], lr *= lrd every step, its own denominator form). */
int pa_adam_step(int dtype, void* param, void* grad, void* exp_avg, void* exp_avg_sq, int64_t n,
                 double lr, double beta1, double beta2, double eps, double weight_decay,
                 double clip_norm, double lrd, int clipped, int64_t* step_dev, int zero_grad,
                 pa_stream_t stream);

/* Full-covariance Normal guide (AutoMultivariateNormal, pyro/infer/autoguide/guides.py:855-965:
 * MultivariateNormal(loc, scale_tril = softplus(rho)[:, None] * (tril(A, -1) + I)); rsample and
 * log_prob of torch/distributions/multivariate_normal.py) for P vectorised particles in one launch:
 *   eps[P, n] ~ N(0, 1) (Philox stream (seed, offset [+ *offset_dev]); eps_given != 0: read instead),
 *   z[p] = loc + S * (L eps[p]),  logq[p] = -|eps[p]|^2/2 - sum_i log S_i - n/2 log(2 pi),
 * with S = softplus(rho) [n], L = tril(A, -1) + I, A [n, n] row-major (unconstrained values).
 * 1 <= n <= 4096. */
int pa_mvn_tril_sample(int dtype, const void* loc, const void* rho, const void* A, int64_t n,
                       int64_t P, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                       int eps_given, void* eps, void* z, void* logq, pa_stream_t stream);
/* Its backward: (d_z [P, n] or NULL, d_logq [P] or NULL) -> gradients of the unconstrained
 * parameters d_loc [n], d_rho [n], d_A [n, n] (zero on and above the diagonal); each may be NULL;
 * accumulate != 0 adds into the buffers (the optimizer's flat gradient views). */
int pa_mvn_tril_sample_bwd(int dtype, const void* loc, const void* rho, const void* eps,
                           const void* z, const void* d_z, const void* d_logq, int64_t n,
                           int64_t P, void* d_loc, void* d_rho, void* d_A, int accumulate,
                           pa_stream_t stream);

/* A chain of T enumerated discrete variables (K <= 64 states each) summed out for B independent
 * batch elements -- the forward algorithm and its forward-backward gradient in one launch:
 *   log_z[b]              = log sum_{v_0..v_{T-1}} exp(sum_t unary[b,t,v_t] + sum_t pair[b,t,v_t,v_{t+1}])
 *   grad_unary[b,t,j]     = d log_z[b] / d unary[b,t,j]       (posterior marginal of v_t)
 *   grad_pairwise[b,t,i,j]= d log_z[b] / d pair[b,t,i,j]      (posterior of the pair), [B,T-1,K,K]
 * unary [B,T,K] contiguous; pair[b,t] = pairwise + b*pair_stride_batch + t*pair_stride_step
 * (elements; 0 = shared), row-major [K,K].  Replaces the pairwise log-space contractions of
 * pyro/ops/contract.py:79-202 + pyro/ops/einsum/torch_log.py:14-55 and their autograd duals for
 * models written with pyro.markov (examples/hmm.py).  K > 64: PA_ERR_UNSUPPORTED.
 * workspace: pa_logchain_workspace(dtype, B, T, K) bytes. */
size_t pa_logchain_workspace(int dtype, int64_t B, int64_t T, int64_t K);
int pa_logchain_fwd_bwd(int dtype, const void* unary, const void* pairwise,
                        int64_t pair_stride_batch, int64_t pair_stride_step, int64_t B,
                        int64_t T, int64_t K, void* log_z, void* grad_unary, void* grad_pairwise,
                        void* workspace, size_t workspace_bytes, pa_stream_t stream);

/* Per-chain dense matrix x vector: y[c] = M[c] x[c] (transpose = 0) or M[c]^T x[c]
 * (transpose = 1); M[C, D, D] row-major with chain stride m_stride_chain elements (0 = one matrix
 * shared by all chains), x, y [C, D] contiguous, x != y.  Replaces the dense-block products of
 * BlockMassMatrix.kinetic_grad / scale / unscale (pyro/infer/mcmc/adaptation.py:328-392), which
 * the reference runs one chain per process; here every chain carries its own adapted matrix. */
int pa_chain_matvec(int dtype, const void* M, int64_t m_stride_chain, const void* x, void* y,
                    int64_t C, int64_t D, int transpose, pa_stream_t stream);

/* pa_adam_step (n > 0) whose last workgroup also does what pa_publish_scalar does: advance the
 * Philox block counter (may be NULL) and hand the scalar `src` (the step's loss) to the host
 * through the pinned (host_value, host_seq) mailbox.  A captured SVI step (pyro/infer/svi.py:
 * 133-159: loss_and_grads, optimizer, zero_grads, return loss) then ends in this one launch. */
int pa_adam_step_publish(int dtype, void* param, void* grad, void* exp_avg, void* exp_avg_sq,
                         int64_t n, double lr, double beta1, double beta2, double eps,
                         double weight_decay, double clip_norm, double lrd, int clipped,
                         int64_t* step_dev, int zero_grad, int src_dtype, const void* src,
                         double* host_value, uint64_t* host_seq, uint64_t* counter, uint64_t inc,
                         pa_stream_t stream);

/* ---- first layer of an amortised guide over word histograms (examples/lda.py:76-92,113-121) -------
 * The reference builds counts = zeros(V, B).scatter_add(0, data, ones) on every step and runs
 * nn.Linear(V, H) on its transpose (dense 410 MB at V = 1024, B = 1e5, four f32 GEMMs per step with
 * the backward).  The histogram is a pure function of the corpus: it is built once (by the caller:
 * pyro_amd/kernels.py::bow_images, bit-exact integer work) as two bf16 images in MFMA-operand order
 *   image_a [ceil(B/32)][V/16][64 lanes][8 bf16]: lane l -> document 32 mt + (l & 31), words 16 kt + 8 (l >> 5) + 0..7
 *   image_b [V/32][Bp/16][64 lanes][8 bf16]    : lane l -> word 32 vt + (l & 31), documents 16 kt + 8 (l >> 5) + 0..7
 * (Bp = B rounded up to 32, missing documents zero; counts must be <= 256: exact in bf16).
 * pa_bow_linear_fwd: out[B, H] = bias + C W^T with W[H, V] f32 split exactly into three bf16 planes
 * per call; pa_bow_linear_bwd: dW[H, V] = d_out^T C with d_out[B, H] split likewise, the sum over the
 * documents split over workgroups and reduced in a fixed order.  f32-class results (exact piece
 * products, f32 accumulation).  V % 128 == 0, H <= 128; workspace: pa_bow_workspace bytes. */
size_t pa_bow_workspace(int64_t B, int64_t V, int64_t H);
int pa_bow_linear_fwd(const void* image_a, const float* W, const float* bias, int64_t B, int64_t V,
                      int64_t H, float* out, void* workspace, size_t workspace_bytes,
                      pa_stream_t stream);
int pa_bow_linear_bwd(const void* image_b, const float* d_out, int64_t B, int64_t V, int64_t H,
                      float* dW, void* workspace, size_t workspace_bytes, pa_stream_t stream);
/* The same layer with the Sigmoid that follows it in examples/lda.py:84-87 (nn.Sequential(Linear, Sigmoid,
 * ...): torch runs sigmoid / sigmoid_backward as two more passes over [B, H] in each direction).
 * pa_bow_linear_fwd_act: sigmoid_out != 0 stores y = 1 / (1 + exp(-(bias + C W^T))).
 * pa_bow_linear_bwd_act: y_mul != NULL (that y) makes d_out the gradient of the ACTIVATION: what is split
 * and multiplied is d_out * (1 - y) * y (sigmoid_backward, never written); db_partial != NULL
 * ([4 * Bp / 16][32] floats) receives the bias gradient's partial sums -- entry [(t * Bp/16 + k)][c] is
 * the sum over documents 16 k .. 16 k + 15 for hidden unit 32 t + c -- which the caller sums over k. */
int pa_bow_linear_fwd_act(const void* image_a, const float* W, const float* bias, int64_t B, int64_t V,
                          int64_t H, int sigmoid_out, float* out, void* workspace,
                          size_t workspace_bytes, pa_stream_t stream);
int pa_bow_linear_bwd_act(const void* image_b, const float* d_out, const float* y_mul, int64_t B, int64_t V,
                          int64_t H, float* dW, float* db_partial, void* workspace, size_t workspace_bytes,
                          pa_stream_t stream);

/* out[M, N] = A^T X for tall f32 operands A[B, M], X[B, N] (M, N <= 128, B large): the weight
 * gradient of a Linear layer over a large batch, dW = d_out^T input (examples/lda.py:76-92 with
 * B = 1e5 documents; torch: mm backward -> a rocBLAS 100 x 100 x 1e5 product that does not split
 * the long dimension).  Both operands are split exactly into three bf16 pieces, the six products
 * of order >= 2^-16 run on the matrix cores with f32 accumulation, the long dimension is split
 * over workgroups and reduced in a fixed order (f32-class result, bitwise reproducible). */
size_t pa_tsgemm_tn_workspace(int64_t B, int64_t M, int64_t N);
int pa_tsgemm_tn(const float* A, const float* X, int64_t B, int64_t M, int64_t N, float* out,
                 void* workspace, size_t workspace_bytes, pa_stream_t stream);

/* A Linear layer over a tall batch (B rows >> 128 features; the inner layers of examples/lda.py:76-92's
 * predictor over 1e5 documents) without rocBLAS and without operand-split passes (csrc/tall.hip).
 *   pa_tall_linear: Y[B, C] = G[B, R] Wm[R, C] (+ bias[C] when not NULL), R, C <= 128, G / Y contiguous;
 *     Wm[r][c] = W[r * w_row_stride + c * w_col_stride]: F.linear's forward is Wm = weight^T (strides 1,
 *     in_features), autograd's dx = g weight is Wm = weight (strides in_features, 1).
 *   pa_tall_wgrad:  dW[R, K] = G^T X and, when db != NULL, db[R] = sum_b G[b, :] for G[B, R], X[B, K]
 *     (R, K <= 128): autograd's weight and bias gradients of F.linear in one pass over the batch; the
 *     long dimension is split over waves and reduced in a fixed order (bitwise reproducible).
 * Operands are split exactly into three bf16 pieces in registers, six piece products on the matrix
 * cores, f32 accumulation (f32-class).  Replaces torch.nn.functional.linear and its AddmmBackward /
 * MmBackward products + the bias reduce. */
int pa_tall_linear(const float* G, int64_t B, int64_t R, const float* W, int64_t w_row_stride,
                   int64_t w_col_stride, int64_t C, const float* bias, float* Y, pa_stream_t stream);
size_t pa_tall_wgrad_workspace(int64_t B, int64_t R, int64_t K);
int pa_tall_wgrad(const float* G, const float* X, int64_t B, int64_t R, int64_t K, float* dW, float* db,
                  void* workspace, size_t workspace_bytes, pa_stream_t stream);
/* ... with the Sigmoid behind the layer fused (see pa_bow_linear_fwd_act): pa_tall_linear_act stores
 * sigmoid(G Wm + bias) when sigmoid_out != 0 and reads its first operand as G * (1 - y) * y when
 * y_mul[B, R] != NULL (the input gradient THROUGH the previous layer's Sigmoid); pa_tall_wgrad_act
 * likewise takes G * (1 - y) * y for G (dW and db of a layer whose output went through a Sigmoid). */
int pa_tall_linear_act(const float* G, int64_t B, int64_t R, const float* W, int64_t w_row_stride,
                       int64_t w_col_stride, int64_t C, const float* bias, const float* y_mul,
                       int sigmoid_out, float* Y, pa_stream_t stream);
int pa_tall_wgrad_act(const float* G, const float* X, const float* y_mul, int64_t B, int64_t R, int64_t K,
                      float* dW, float* db, void* workspace, size_t workspace_bytes, pa_stream_t stream);

/* Reparameterised standard-Gamma draws out[i] ~ Gamma(alpha[i], 1) on the keyed Philox stream and,
 * when d_alpha != NULL, the implicit reparameterisation gradient d out[i] / d alpha[i].  Replaces
 * torch._standard_gamma + torch._standard_gamma_grad behind torch.distributions.Gamma.rsample
 * (gamma.py:80-88), which pyro's Gamma / Beta / Dirichlet draw through (pyro/distributions/torch.py;
 * examples/lda.py:107-109).  Element i reads the Philox blocks (offset + *offset_dev + i, tag | k),
 * k = attempt of the Marsaglia-Tsang rejection loop (budget 24): reproducible for any launch
 * geometry and replay-safe in a hipGraph.  alpha: view on the [rows, cols] frame; out, d_alpha
 * contiguous.  pa_gamma_implicit_grad: the gradient alone, at given (alpha, value). */
int pa_gamma_rsample(int dtype, void* out, void* d_alpha, pa_view2d alpha, int64_t rows, int64_t cols,
                     uint64_t seed, uint64_t offset, const uint64_t* offset_dev, pa_stream_t stream);
int pa_gamma_implicit_grad(int dtype, void* d_alpha, pa_view2d alpha, pa_view2d value, int64_t rows,
                           int64_t cols, pa_stream_t stream);

/* ---- the chained tail of an SVI step ------------------------------------------------------------
 * Replaces the dependent small launches that end the reference's step -- per-site sums and their
 * autograd duals (pyro/infer/trace_elbo.py:130-159), AccumulateGrad + per-parameter optimizer steps
 * (pyro/optim/optim.py:117-155), zero_grads (pyro/infer/util.py:85-91) -- which even as four
 * launches of this library cost ~5 us of graph-node dispatch each.
 *
 * Between pa_chain_begin(stream, sync, bytes) and pa_chain_end() on the calling thread, these entry
 * points called with `stream` and float32 data RECORD their work instead of launching it:
 *     the finalize step of pa_glm_bernoulli_fwd_bwd / pa_glm_bernoulli_planes_fwd_bwd,
 *     pa_multi_log_prob_sum_grad, pa_meanfield_normal_sample_bwd (<= 8 sites),
 *     pa_adam_step / pa_adam_step_publish,
 * in that order, at most one of each.  The recorded phases run as ONE kernel (device-wide barriers
 * between phases, bit-identical results) when pa_chain_end / pa_chain_flush is called, when a
 * recordable call does not continue the order, or when ANY other entry point of this library is
 * about to launch (it may read what the phases write).  The caller guarantees that nothing outside
 * this library touches the phases' inputs or outputs on the device between the recording and the
 * flush (pyro_amd/kernels.py: a TorchDispatchMode flushes before every kernel-launching torch
 * operator), and keeps every recorded buffer alive until then.
 * sync: PA_CHAIN_SYNC_BYTES of zero-initialised device memory that stays allocated as long as a
 * captured graph holding a chain launch may be replayed; launches leave it zero. */
#define PA_CHAIN_SYNC_BYTES 64
int pa_chain_begin(pa_stream_t stream, void* sync_words, size_t sync_bytes);
/* launch what is pending; recording continues */
int pa_chain_flush(void);
/* launch what is pending and stop recording; optional out-params: chain launches made since
 * pa_chain_begin and the number of phases they carried */
int pa_chain_end(int* launches, int* phases);
/* phases recorded and not yet launched */
int pa_chain_pending(void);
/* fuse_tail = 3: as 1 below, without the one-pass code for AutoNormal-shaped sites (site_tail.h):
 * each site runs the generic entry / backward / Adam code in its workgroup.
 * fuse_tail = 1 (default): when every gradient of the recorded ELBO assembly feeds the backward
 * of exactly one mean-field site and the sites' parameters tile the optimizer's flat buffer, the
 * assembly / guide-backward / Adam phases run per site inside one workgroup each (no device-wide
 * barrier between them); 0: always the generic phase-by-phase form.  Same results either way.
 * Bit 2 (value 4): the guide draw in front of a plane-image GLM launch stays its own launch (default: inside a
 * recording pa_meanfield_normal_sample parks its launch and the GLM kernel whose weights / bias are two of its
 * sites' draws makes them in its own prologue -- the same numbers, one graph node less).
 * Bits 8 and up: race hunting -- a non-zero seed makes every workgroup of the chain kernels sleep a
 * pseudo-random 0..17 us in front of each phase arrival and after each phase wait, which shuffles the
 * order of the device-wide arrivals (the kernels must give bit-identical results under any seed). */
int pa_chain_tune(int fuse_tail);
/* chain launches since pa_chain_begin that took the fused form */
int pa_chain_fused_launches(void);
/* developer hook: 32 x uint64 of device memory that later chain launches fill with wall-clock
 * (100 MHz) stamps at their phase boundaries (workgroup 0: [0..7], the total's workgroup:
 * [16..23]); NULL switches it off */
int pa_chain_debug_stamps(void* stamps32);

/* ---- run-time compiled element-wise kernels (pyro_amd/ops/fuser.py) ------------------------------
 * Replaces the runs of small ATen operators between the fused sites of a step (constraint
 * transforms, a model's own normalisations, their autograd duals: pyro/infer/traceenum_elbo.py:112-214
 * over examples/lda.py:78-122 dispatches ~110 of them per step).  The host emits one HIP source per run;
 * pa_rtc_compile builds it for gfx950 with hiprtc (-ffp-contract=off: a product and a sum stay two
 * roundings, as in the operators they replace) and returns the kernel `kernel_name` of it; the kernel's
 * single parameter is a struct of at most PA_RTC_MAX_POINTERS device pointers by value (shapes and strides are
 * constants of the source).  pa_rtc_launch: grid x block threads on `stream`, pointers[0..n). */
#define PA_RTC_MAX_POINTERS 384
int pa_rtc_compile(const char* source, const char* kernel_name, void** function_out);
/* The same with a PERSISTENT cache: `cache_file` (or NULL) names the code object of this source -- the caller
 * derives the name from a digest of the source, the compiler's version (pa_rtc_version) and the options.  An
 * existing file is loaded with hipModuleLoadData and hiprtc is not called (*compiled_out = 0); otherwise the
 * source is compiled (*compiled_out = 1) and the code object written there (private name + rename).  A second
 * process warms up a step without a single hiprtc call. */
int pa_rtc_compile_cached(const char* source, const char* kernel_name, const char* cache_file,
                          void** function_out, int* compiled_out);
int pa_rtc_version(int* hiprtc_major, int* hiprtc_minor, int* runtime_version);
/* Launches made while their stream is being captured keep a parameter block alive for the captured graph.
 * pa_rtc_blocks_begin opens a scope on the calling thread that OWNS the blocks of such launches until
 * pa_rtc_blocks_end (*n_blocks_out: how many it holds); pa_rtc_blocks_free releases them -- call it when the
 * captured graph is destroyed.  Without a scope a block lives as long as the process. */
void* pa_rtc_blocks_begin(void);
int pa_rtc_blocks_end(void* scope, int64_t* n_blocks_out);
int pa_rtc_blocks_free(void* scope);
int pa_rtc_launch(void* function, uint32_t grid, uint32_t block, const void* const* pointers,
                  int n_pointers, pa_stream_t stream);

/* ---- csrc/mixture.hip: an observed site under an enumerated assignment, forward and backward in one pass ----
 * Replaces, for the leaf of a plated mixture under TraceEnum_ELBO (pyro/infer/traceenum_elbo.py:112-214): the
 * observed site's [K, N] log_prob (pyro/poutine/trace_struct.py:248-288), the sum-product's add + logsumexp over the
 * enumerated variable and the plate sum (pyro/ops/contract.py:79-160), and the autograd duals of all three.
 *   S[b] = sum_n log sum_k exp(a[b][k] + log p(x[n] | p0[b][k], p1[b][k]))       (family `dist`, K <= 64, b < B)
 * for B parameter sets over the SAME data (B = 1: one model evaluation; B = the vectorised chains of HMC / NUTS or
 * the vectorised particles of an ELBO).  a[b][k] = a[b * a_batch_stride + k]; p0[b][k] = p0[b * p0_batch_stride +
 * k * p0_stride] (stride 0: shared by the components / by the sets), p1 likewise or NULL for one-parameter families.
 * out[b * (1 + 3 K) + ...]: [0] = S, [1 + k] = dS/da_k, [1 + K + k] = dS/dp0_k, [1 + 2 K + k] = dS/dp1_k (doubles in
 * device memory; parameter gradients per (b, k) -- a shared parameter takes their sum).  x, a, p0, p1 are `dtype`
 * (PA_F32 / PA_F64).  Families: Normal, LogNormal, Exponential, Bernoulli(logits), Poisson, Gamma.  Workspace:
 * pa_mixture_workspace(K, B) bytes.  Bit-reproducible. */
size_t pa_mixture_workspace(int K, int64_t B);
int pa_mixture_fwd_bwd(int dtype, int dist, const void* x, int64_t N, int K, int64_t B, const void* a,
                       int64_t a_batch_stride, const void* p0, int64_t p0_stride, int64_t p0_batch_stride,
                       const void* p1, int64_t p1_stride, int64_t p1_batch_stride, void* workspace,
                       size_t workspace_bytes, double* out, pa_stream_t stream);

/* The same leaf for EVENT-SHAPED observations: a diagonal Normal over D <= 8 features
 * (`Normal(locs[z], scale).to_event(1)` under the data plate: the multi-dimensional Gaussian mixture) --
 *   S[b] = sum_n log sum_k exp(a[b][k] + sum_d log N(x[n][d] | loc[b][k][d], scale[b][k][d])),
 * the sum over the features inside the logsumexp (pyro/distributions/torch.py Independent.log_prob ->
 * torch/distributions/independent.py:96-98).  x [N, D] row-major; loc[b][k][d] = loc[b * loc_batch_stride +
 * k * loc_stride_k + d * loc_stride_d], scale likewise (stride 0: shared).  The output is PADDED: with
 * J = pa_mixture_diag_normal_layout(K, D, &KP, &DD) doubles per set, out[b * J + ...]: [0] = S, [1 + k] = dS/da_k,
 * [1 + KP + k * DD + d] = dS/dloc_kd, [1 + KP + KP * DD + k * DD + d] = dS/dscale_kd (entries past K / D are zero).
 * Workspace: pa_mixture_diag_normal_workspace(K, D, B) bytes.  Bit-reproducible. */
int pa_mixture_diag_normal_layout(int K, int D, int* kp_out, int* dd_out);
size_t pa_mixture_diag_normal_workspace(int K, int D, int64_t B);
int pa_mixture_diag_normal_fwd_bwd(int dtype, const void* x, int64_t N, int D, int K, int64_t B, const void* a,
                                   int64_t a_batch_stride, const void* loc, int64_t loc_stride_k,
                                   int64_t loc_stride_d, int64_t loc_batch_stride, const void* scale,
                                   int64_t scale_stride_k, int64_t scale_stride_d, int64_t scale_batch_stride,
                                   void* workspace, size_t workspace_bytes, double* out_padded, pa_stream_t stream);

/* ---- csrc/replay.hip: a captured step that is a short chain of kernels, launched as kernels ----------------
 * Replaces nothing of the reference's (its SVI.step, pyro/infer/svi.py:134-162, re-runs the model): it is the
 * replay path of this package's captured step, opt-in on the host side: measured on config 2 a step that waits for
 * its loss takes 72.2 us through hipGraphLaunch and 75.7 us as two launches; with replays queued ahead of the host
 * the launches follow each other more closely (62.5 against 66.5 us per step).
 * pa_graph_direct_plan inspects a hipGraph_t (`hip_graph`; it must outlive the plan: the plan points into the
 * nodes' argument blocks): when the graph is ONE chain of at most `max_nodes` kernel nodes launched from host
 * functions, *plan_out receives a plan, otherwise NULL (not an error: the caller keeps hipGraphLaunch);
 * *n_nodes_out = the graph's node count.  pa_graph_direct_launch enqueues the plan's kernels in chain order into
 * `stream` with the captured grids and arguments -- the same device work as one hipGraphLaunch. */
int pa_graph_direct_plan(void* hip_graph, int max_nodes, void** plan_out, int* n_nodes_out);
int pa_graph_direct_launch(void* plan, pa_stream_t stream);
int pa_graph_direct_free(void* plan);

#ifdef __cplusplus
}
#endif
#endif /* PYRO_AMD_H */
