// optim_dev.h -- device code of the flat Adam / ClippedAdam step (launchers: optim.hip; also a phase
// of the chained tail of an SVI step, chain.hip).
#pragma once
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

// (vb, nvb) = this workgroup's index and the number of workgroups of the launch (the chained tail
// of an SVI step, chain.hip, walks several virtual workgroups per physical one)
template <typename T>
__device__ __forceinline__ void adam_body(int64_t vb, int64_t nvb, T* __restrict__ p,
                                          T* __restrict__ g, T* __restrict__ m, T* __restrict__ v,
                                          int64_t n, double lr, double b1, double b2, double eps,
                                          double wd, double clip, double lrd, int clipped,
                                          int64_t* __restrict__ step_dev, int zero_grad,
                                          const AdamPublish& pub) {
  // step_dev[0] = steps taken so far, step_dev[1] = workgroups of THIS launch that have finished.
  // Every workgroup reads the step count when it starts; the last one to finish (all others have
  // read it by then) advances it and resets the ticket: no separate "bump" launch.
  const int64_t step = step_dev[0] + 1;
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  // ClippedAdam multiplies lr by lrd before every step (clipped_adam.py:63)
  const double lr_t = clipped ? lr * pow(lrd, (double)step) : lr;
  for (int64_t i = vb * 256 + threadIdx.x; i < n; i += nvb * 256) {
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
    if (ticket == (unsigned long long)nvb - 1) {
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

template <typename T>
__global__ __launch_bounds__(256) void adam_kernel(T* __restrict__ p, T* __restrict__ g,
                                                   T* __restrict__ m, T* __restrict__ v, int64_t n,
                                                   double lr, double b1, double b2, double eps,
                                                   double wd, double clip, double lrd, int clipped,
                                                   int64_t* __restrict__ step_dev,
                                                   int zero_grad, AdamPublish pub) {
  adam_body<T>((int64_t)blockIdx.x, (int64_t)gridDim.x, p, g, m, v, n, lr, b1, b2, eps, wd, clip, lrd,
               clipped, step_dev, zero_grad, pub);
}

}  // namespace pa
