// multisite.hip -- C-ABI launchers of the many-small-sites kernels (device code: multisite_dev.h)
#include "multisite_dev.h"
#include "chain.h"

namespace pa {

// the guide draw as its own kernel (also what a parked draw falls back to, chain.hip)
int meanfield_sample_launch_any(int dtype, const MfArgs& args, int nsites, int64_t P, uint64_t seed,
                                const uint64_t* offset_dev, hipStream_t s) {
  int64_t maxn = 0;
  for (int k = 0; k < nsites; ++k)
    if (args.s[k].n > maxn) maxn = args.s[k].n;
  // small sites: one element per thread (latency); from 64 K elements on: one Philox block = 4 f32 /
  // 2 f64 elements per thread and trip (a quarter / half of the generator work) -- the same numbers
  const bool big = P * maxn >= (int64_t(1) << 16);
  const int64_t per = !big ? 1 : (dtype == PA_F32 ? 4 : 2);
  int64_t gx = ((P * maxn + per - 1) / per + 255) / 256;
  if (gx < 1) gx = 1;
  const int64_t cap = (int64_t)cu_count() * 4;
  if (gx > cap) gx = cap;
  const dim3 grid((unsigned)gx, (unsigned)nsites);
  if (dtype == PA_F32 && big)
    hipLaunchKernelGGL((meanfield_sample_block_kernel<float>), grid, dim3(256), 0, s, args, P, seed,
                       offset_dev, gate_word());
  else if (dtype == PA_F32)
    hipLaunchKernelGGL((meanfield_sample_kernel<float>), grid, dim3(256), 0, s, args, P, seed,
                       offset_dev, gate_word());
  else if (big)
    hipLaunchKernelGGL((meanfield_sample_block_kernel<double>), grid, dim3(256), 0, s, args, P, seed,
                       offset_dev, gate_word());
  else
    hipLaunchKernelGGL((meanfield_sample_kernel<double>), grid, dim3(256), 0, s, args, P, seed,
                       offset_dev, gate_word());
  gate_aware_launch();
  return check_launch("meanfield_sample_kernel");
}
int meanfield_sample_launch(const MfArgs& args, int nsites, int64_t P, uint64_t seed,
                            const uint64_t* offset_dev, hipStream_t s) {
  return meanfield_sample_launch_any(PA_F32, args, nsites, P, seed, offset_dev, s);
}

}  // namespace pa

extern "C" {

int pa_multi_log_prob_sum(int dtype, void* out_total, const pa_site_entry* entries, int n,
                          double coef_all, int accumulate, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_multi_log_prob_sum: bad dtype %d", dtype);
  PA_REQUIRE(out_total != nullptr, "pa_multi_log_prob_sum: NULL output");
  pa::MultiArgs args;
  int rc = pa::to_dev(entries, n, &args, "pa_multi_log_prob_sum");
  if (rc != PA_OK) return rc;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::multi_sum_kernel<float>), dim3(1), dim3(pa::MULTI_THREADS), 0, s,
                       args, (float*)out_total, coef_all, accumulate);
  else
    hipLaunchKernelGGL((pa::multi_sum_kernel<double>), dim3(1), dim3(pa::MULTI_THREADS), 0, s,
                       args, (double*)out_total, coef_all, accumulate);
  return pa::check_launch("multi_sum_kernel");
}

int pa_multi_log_prob_grad(int dtype, const void* g, const pa_site_entry* entries, int n,
                           double coef_all, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_multi_log_prob_grad: bad dtype %d", dtype);
  PA_REQUIRE(g != nullptr, "pa_multi_log_prob_grad: NULL upstream gradient");
  pa::MultiArgs args;
  int rc = pa::to_dev(entries, n, &args, "pa_multi_log_prob_grad");
  if (rc != PA_OK) return rc;
  if (n == 0) return PA_OK;
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::multi_grad_kernel<float>), dim3((unsigned)n), dim3(pa::GRAD_THREADS), 0, s,
                       args, (const float*)g, coef_all);
  else
    hipLaunchKernelGGL((pa::multi_grad_kernel<double>), dim3((unsigned)n), dim3(pa::GRAD_THREADS), 0, s,
                       args, (const double*)g, coef_all);
  return pa::check_launch("multi_grad_kernel");
}

int pa_multi_log_prob_sum_grad(int dtype, void* out_total, const void* g,
                               const pa_site_entry* entries, int n, double coef_all,
                               int accumulate, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_multi_log_prob_sum_grad: bad dtype %d", dtype);
  PA_REQUIRE(out_total != nullptr, "pa_multi_log_prob_sum_grad: NULL output");
  pa::MultiArgs args;
  int rc = pa::to_dev(entries, n, &args, "pa_multi_log_prob_sum_grad");
  if (rc != PA_OK) return rc;
  if (dtype == PA_F32 && n > 0) {
    rc = pa::chain_record_multi(stream, args, (float*)out_total, (const float*)g, coef_all,
                                accumulate);
    if (rc != 0) return rc < 0 ? rc : PA_OK;     // recorded as a phase of the step's chained tail
  }
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    // f32: 16 waves for the total's workgroup (the gradient code fits the 128-VGPR budget of a
    // 1024-thread launch); f64 gradient code needs more registers: 256-thread launch
    hipLaunchKernelGGL((pa::multi_sum_grad_kernel<float, pa::MULTI_THREADS>), dim3((unsigned)n + 1),
                       dim3(pa::MULTI_THREADS), 0, s, args, (float*)out_total, (const float*)g,
                       coef_all, accumulate);
  else
    hipLaunchKernelGGL((pa::multi_sum_grad_kernel<double, pa::GRAD_THREADS>),
                       dim3((unsigned)n + 1), dim3(pa::GRAD_THREADS), 0, s, args,
                       (double*)out_total, (const double*)g, coef_all, accumulate);
  return pa::check_launch("multi_sum_grad_kernel");
}

int pa_meanfield_normal_sample(int dtype, const pa_mf_site* sites, int nsites, int64_t P,
                               uint64_t seed, const uint64_t* offset_dev, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_meanfield_normal_sample: bad dtype %d", dtype);
  PA_REQUIRE(P >= 1, "pa_meanfield_normal_sample: P = %lld", (long long)P);
  pa::MfArgs args;
  int rc = pa::mf_to_dev(sites, nsites, &args, "pa_meanfield_normal_sample");
  if (rc != PA_OK) return rc;
  if (nsites == 0) return PA_OK;
  for (int k = 0; k < nsites; ++k)
    PA_REQUIRE(sites[k].n == 0 || (sites[k].loc && sites[k].rho && sites[k].z && sites[k].scale &&
                                   sites[k].loc_out && sites[k].eps),
               "pa_meanfield_normal_sample: site %d: NULL pointer", k);
  if (dtype == PA_F32) {
    // inside a recorded step the draw waits for the launch that consumes it (chain.h: the plane-image
    // GLM kernel draws its own weights)
    rc = pa::chain_park_draw(stream, args, nsites, P, seed, offset_dev);
    if (rc != 0) return rc < 0 ? rc : PA_OK;
  }
  return pa::meanfield_sample_launch_any(dtype, args, nsites, P, seed, offset_dev, pa::as_stream(stream));
}

int pa_meanfield_normal_sample_bwd(int dtype, const pa_mf_site* sites, int nsites, int64_t P,
                                   pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "pa_meanfield_normal_sample_bwd: bad dtype %d",
             dtype);
  PA_REQUIRE(P >= 1, "pa_meanfield_normal_sample_bwd: P = %lld", (long long)P);
  pa::MfArgs args;
  int rc = pa::mf_to_dev(sites, nsites, &args, "pa_meanfield_normal_sample_bwd");
  if (rc != PA_OK) return rc;
  if (nsites == 0) return PA_OK;
  int64_t maxn = 0;
  for (int k = 0; k < nsites; ++k) {
    PA_REQUIRE(sites[k].n == 0 || (sites[k].rho && sites[k].eps),
               "pa_meanfield_normal_sample_bwd: site %d: NULL pointer", k);
    if (sites[k].n > maxn) maxn = sites[k].n;
  }
  int64_t gy = (maxn >= pa::MF_BWD_WIDE_N && P >= 16) ? (maxn + 63) / 64 : (maxn + 255) / 256;
  if (gy < 1) gy = 1;
  if (gy > 4096) gy = 4096;
  if (dtype == PA_F32) {
    rc = pa::chain_record_mf_bwd(stream, args, nsites, P, (int)gy);
    if (rc != 0) return rc < 0 ? rc : PA_OK;     // recorded as a phase of the step's chained tail
  }
  hipStream_t s = pa::as_stream(stream);
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::meanfield_sample_bwd_kernel<float>), dim3((unsigned)nsites, (unsigned)gy),
                       dim3(256), 0, s, args, P);
  else
    hipLaunchKernelGGL((pa::meanfield_sample_bwd_kernel<double>),
                       dim3((unsigned)nsites, (unsigned)gy), dim3(256), 0, s, args, P);
  return pa::check_launch("meanfield_sample_bwd_kernel");
}

}  // extern "C"
