// optim.hip -- flat multi-tensor Adam / ClippedAdam step.
// The reference keeps one torch optimizer object per parameter and loops over them in Python
// (pyro/optim/optim.py:117-155) and re-allocates zero gradients every step
// (pyro/infer/util.py:85-91). Here all unconstrained parameters live in one flat buffer:
// one launch updates everything, optionally zeroes the gradient in the same pass, and the
// step counter lives in device memory so the launch can be replayed from a hipGraph.
#include "optim_dev.h"
#include "chain.h"

namespace pa {

__global__ void adam_bump_kernel(int64_t* step_dev) { step_dev[0] += 1; }

static int adam_launch(int dtype, void* param, void* grad, void* exp_avg, void* exp_avg_sq,
                       int64_t n, double lr, double beta1, double beta2, double eps,
                       double weight_decay, double clip_norm, double lrd, int clipped,
                       int64_t* step_dev, int zero_grad, const AdamPublish& pub,
                       pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "adam_step: bad dtype %d", dtype);
  PA_REQUIRE(n > 0, "adam_step: n <= 0");
  PA_REQUIRE(step_dev != nullptr, "adam_step: NULL step counter");
  PA_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: NULL buffer");
  if (dtype == PA_F32) {
    const int rc = chain_record_adam(stream, (float*)param, (float*)grad, (float*)exp_avg,
                                     (float*)exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                                     clip_norm, lrd, clipped, step_dev, zero_grad, pub);
    if (rc != 0) return rc < 0 ? rc : PA_OK;     // recorded as a phase of the step's chained tail
  }
  hipStream_t s = as_stream(stream);
  int64_t grid = (n + 255) / 256;
  const int64_t cap = (int64_t)cu_count() * 8;
  if (grid > cap) grid = cap;
  if (dtype == PA_F32)
    hipLaunchKernelGGL((adam_kernel<float>), dim3((unsigned)grid), dim3(256), 0, s, (float*)param,
                       (float*)grad, (float*)exp_avg, (float*)exp_avg_sq, n, lr, beta1, beta2, eps,
                       weight_decay, clip_norm, lrd, clipped, step_dev, zero_grad, pub);
  else
    hipLaunchKernelGGL((adam_kernel<double>), dim3((unsigned)grid), dim3(256), 0, s,
                       (double*)param, (double*)grad, (double*)exp_avg, (double*)exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, clip_norm, lrd, clipped, step_dev,
                       zero_grad, pub);
  return check_launch("adam_kernel");
}

}  // namespace pa

extern "C" {

int pa_adam_step(int dtype, void* param, void* grad, void* exp_avg, void* exp_avg_sq, int64_t n,
                 double lr, double beta1, double beta2, double eps, double weight_decay,
                 double clip_norm, double lrd, int clipped, int64_t* step_dev, int zero_grad,
                 pa_stream_t stream) {
  PA_REQUIRE(n >= 0, "adam_step: n < 0");
  PA_REQUIRE(step_dev != nullptr, "adam_step: NULL step counter");
  if (n > 0)
    return pa::adam_launch(dtype, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                           weight_decay, clip_norm, lrd, clipped, step_dev, zero_grad,
                           pa::AdamPublish{nullptr, 0, nullptr, nullptr, nullptr, 0}, stream);
  hipLaunchKernelGGL(pa::adam_bump_kernel, dim3(1), dim3(1), 0, pa::as_stream(stream), step_dev);
  return pa::check_launch("adam_bump_kernel");
}

int pa_adam_step_publish(int dtype, void* param, void* grad, void* exp_avg, void* exp_avg_sq,
                         int64_t n, double lr, double beta1, double beta2, double eps,
                         double weight_decay, double clip_norm, double lrd, int clipped,
                         int64_t* step_dev, int zero_grad, int src_dtype, const void* src,
                         double* host_value, uint64_t* host_seq, uint64_t* counter, uint64_t inc,
                         pa_stream_t stream) {
  PA_REQUIRE(n > 0, "adam_step_publish: n <= 0");
  PA_REQUIRE(src_dtype == PA_F32 || src_dtype == PA_F64, "adam_step_publish: bad scalar dtype %d",
             src_dtype);
  PA_REQUIRE(src && host_value && host_seq, "adam_step_publish: NULL pointer");
  return pa::adam_launch(dtype, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                         weight_decay, clip_norm, lrd, clipped, step_dev, zero_grad,
                         pa::AdamPublish{src, src_dtype, host_value, host_seq, counter, inc},
                         stream);
}

}  // extern "C"
