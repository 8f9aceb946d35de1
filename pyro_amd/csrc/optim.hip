// optim.hip -- flat multi-tensor Adam / ClippedAdam step.
// The reference keeps one torch optimizer object per parameter and loops over them in Python
// (pyro/optim/optim.py:117-155) and re-allocates zero gradients every step
// (pyro/infer/util.py:85-91). Here all unconstrained parameters live in one flat buffer:
// one launch updates everything, optionally zeroes the gradient in the same pass, and the
// step counter lives in device memory so the launch can be replayed from a hipGraph.
#include "common.h"

namespace pa {

// Optional epilogue of the update launch (pa_adam_step_publish): what pa_publish_scalar does,
// run by the last workgroup to finish, so a captured SVI step needs no separate 1-thread node.
struct AdamPublish {
  const void* src;        // device scalar (the step's loss), NULL = nothing to publish
  int src_dtype;
  double* host_value;     // pinned
  uint64_t* host_seq;     // pinned
  uint64_t* counter;      // device Philox block counter, may be NULL
  uint64_t inc;
};

template <typename T>
__global__ __launch_bounds__(256) void adam_kernel(T* __restrict__ p, T* __restrict__ g,
                                                   T* __restrict__ m, T* __restrict__ v, int64_t n,
                                                   double lr, double b1, double b2, double eps,
                                                   double wd, double clip, double lrd, int clipped,
                                                   int64_t* __restrict__ step_dev,
                                                   int zero_grad, AdamPublish pub) {
  // step_dev[0] = steps taken so far, step_dev[1] = workgroups of THIS launch that have finished.
  // Every workgroup reads the step count when it starts; the last one to finish (all others have
  // read it by then) advances it and resets the ticket: no separate "bump" launch.
  const int64_t step = step_dev[0] + 1;
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  // ClippedAdam multiplies lr by lrd before every step (clipped_adam.py:63)
  const double lr_t = clipped ? lr * pow(lrd, (double)step) : lr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    T gi = g[i];
    if (clipped && clip > 0.0) {  // element-wise clamp, clipped_adam.py:69
      gi = gi > (T)clip ? (T)clip : (gi < (T)(-clip) ? (T)(-clip) : gi);
    }
    if (wd != 0.0) gi = gi + (T)wd * p[i];
    const T mi = (T)b1 * m[i] + (T)(1.0 - b1) * gi;
    const T vi = (T)b2 * v[i] + (T)(1.0 - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    T upd;
    if (clipped) {  // clipped_adam.py:91-97
      const T denom = sqrt(vi) + (T)eps;
      upd = (T)(lr_t * sqrt(bc2) / bc1) * (mi / denom);
    } else {        // torch.optim.Adam (single-tensor path)
      const T denom = sqrt(vi) / (T)sqrt(bc2) + (T)eps;
      upd = (T)(lr_t / bc1) * (mi / denom);
    }
    p[i] = p[i] - upd;
    if (zero_grad) g[i] = T(0);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long ticket =
        atomicAdd(reinterpret_cast<unsigned long long*>(step_dev + 1), 1ull);
    if (ticket == (unsigned long long)gridDim.x - 1) {
      step_dev[1] = 0;
      step_dev[0] = step;
      if (pub.counter != nullptr) *pub.counter += pub.inc;
      if (pub.src != nullptr) {
        const double v = pub.src_dtype == PA_F32 ? (double)*static_cast<const float*>(pub.src)
                                                 : *static_cast<const double*>(pub.src);
        __hip_atomic_store(pub.host_value, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();                  // value visible to the host before the flag
        const uint64_t seq =
            __hip_atomic_load(pub.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(pub.host_seq, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

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
