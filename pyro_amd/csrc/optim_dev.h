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
  uint64_t* counter;      // device uint64[2]: {Philox block counter, publish sequence}, may be NULL
  uint64_t inc;
};

// counter[0] += inc; the scalar goes to the pinned mailbox: value first, then -- behind a
// system-scope fence -- the sequence number the host polls.  The sequence number is counter[1] + 1
// (device memory); only without a counter is it read back from the host word, which costs a PCIe
// round trip (~2 us) at the very end of every step.
__device__ __forceinline__ void publish_to_host(const AdamPublish& pub) {
  uint64_t seq = 0;
  if (pub.counter != nullptr) {
    const uint64_t c0 = pub.counter[0], c1 = pub.counter[1];
    pub.counter[0] = c0 + pub.inc;
    if (pub.src != nullptr) pub.counter[1] = seq = c1 + 1;
  }
  if (pub.src != nullptr) {
    const double v = pub.src_dtype == PA_F32 ? (double)*static_cast<const float*>(pub.src)
                                             : *static_cast<const double*>(pub.src);
    __hip_atomic_store(pub.host_value, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();                  // value visible to the host before the flag
    if (pub.counter == nullptr)
      seq = __hip_atomic_load(pub.host_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
    __hip_atomic_store(pub.host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the step-dependent scalars of one launch
struct AdamStep {
  double bc1, bc2, lr_t;
};
// b^n by repeated squaring: a dozen instructions in a loop.  libm's pow() is ~2 KB of straight-line
// code per call site, and the update launch is a few hundred bytes of data: with three pow() the
// kernel spent most of its 6 us FETCHING cold instructions (~1.6 us per KB right after the big
// likelihood kernel has swept the caches).  Relative error <= 2 log2(n) ulp.
__device__ __forceinline__ double pow_int(double b, int64_t n) {
  double r = 1.0, x = b;
  while (n > 0) {
    if (n & 1) r *= x;
    x *= x;
    n >>= 1;
  }
  return r;
}
__device__ __forceinline__ AdamStep adam_step_scalars(int64_t step, double lr, double b1, double b2,
                                                      double lrd, int clipped) {
  AdamStep s;
  s.bc1 = 1.0 - pow_int(b1, step);
  s.bc2 = 1.0 - pow_int(b2, step);
  // ClippedAdam multiplies lr by lrd before every step (clipped_adam.py:63)
  s.lr_t = clipped ? lr * pow_int(lrd, step) : lr;
  return s;
}

// one element: (gradient, parameter, moments) -> new (parameter, moments)
template <typename T>
__device__ __forceinline__ void adam_update(T gi, T& pi, T& mi, T& vi, const AdamStep& st, double b1,
                                            double b2, double eps, double wd, double clip,
                                            int clipped) {
  // lr == 0 without weight decay: the no-op optimizer (pyro_amd.optim.NoUpdate -- a step that computes the loss
  // and the gradients, hands the loss over and zeroes the gradients): parameter AND moments stay as they are
  // (0 * m / denom would also turn an overflowed moment into a NaN parameter)
  if (st.lr_t == 0.0 && wd == 0.0) return;
  if (clipped && clip > 0.0) {  // element-wise clamp, clipped_adam.py:69
    gi = gi > (T)clip ? (T)clip : (gi < (T)(-clip) ? (T)(-clip) : gi);
  }
  if (wd != 0.0) gi = gi + (T)wd * pi;
  mi = (T)b1 * mi + (T)(1.0 - b1) * gi;
  vi = (T)b2 * vi + (T)(1.0 - b2) * gi * gi;
  T upd;
  if (clipped) {  // clipped_adam.py:91-97
    const T denom = sqrt(vi) + (T)eps;
    upd = (T)(st.lr_t * sqrt(st.bc2) / st.bc1) * (mi / denom);
  } else {        // torch.optim.Adam (single-tensor path)
    const T denom = sqrt(vi) / (T)sqrt(st.bc2) + (T)eps;
    upd = (T)(st.lr_t / st.bc1) * (mi / denom);
  }
  pi = pi - upd;
}

// elements first + k * stride (k = 0, 1, ...) below n, by this thread.  The operands of the first
// element are requested BEFORE the step counter is consumed: the loads and the three fp64 pow() of
// the step scalars overlap instead of queueing behind one another (these launches move a few
// hundred bytes; what they cost is dependent memory round trips).
template <typename T>
__device__ __forceinline__ void adam_range(int64_t first, int64_t stride, int64_t n,
                                           T* __restrict__ p, T* __restrict__ g, T* __restrict__ m,
                                           T* __restrict__ v, const int64_t* __restrict__ step_dev,
                                           double lr, double b1, double b2, double eps, double wd,
                                           double clip, double lrd, int clipped, int zero_grad,
                                           int64_t* step_out) {
  const bool ok0 = first < n;
  const int64_t i0 = ok0 ? first : 0;
  T g0 = g[i0], p0 = p[i0], m0 = m[i0], v0 = v[i0];
  const int64_t step = step_dev[0] + 1;
  *step_out = step;
  const AdamStep st = adam_step_scalars(step, lr, b1, b2, lrd, clipped);
  if (ok0) {
    adam_update<T>(g0, p0, m0, v0, st, b1, b2, eps, wd, clip, clipped);
    m[i0] = m0; v[i0] = v0; p[i0] = p0;
    if (zero_grad) g[i0] = T(0);
  }
  for (int64_t i = first + stride; i < n; i += stride) {
    T pi = p[i], mi = m[i], vi = v[i];
    adam_update<T>(g[i], pi, mi, vi, st, b1, b2, eps, wd, clip, clipped);
    m[i] = mi; v[i] = vi; p[i] = pi;
    if (zero_grad) g[i] = T(0);
  }
}

// Two slices [o0, o0 + n) and [o1, o1 + n) of the flat buffers by one 256-thread workgroup (a
// mean-field site's loc and rho in the fused tail of an SVI step, chain.hip): adam_range's
// arithmetic per element, one evaluation of the step scalars.
template <typename T>
__device__ __forceinline__ void adam_two_ranges(int64_t o0, int64_t o1, int64_t n,
                                                T* __restrict__ p, T* __restrict__ g,
                                                T* __restrict__ m, T* __restrict__ v,
                                                const int64_t* __restrict__ step_dev, double lr,
                                                double b1, double b2, double eps, double wd,
                                                double clip, double lrd, int clipped, int zero_grad,
                                                int64_t* step_out) {
  const int64_t t = threadIdx.x;
  const bool ok = t < n;
  const int64_t i0 = o0 + (ok ? t : 0), i1 = o1 + (ok ? t : 0);
  T ga = g[i0], pa_ = p[i0], ma = m[i0], va = v[i0];
  T gb = g[i1], pb = p[i1], mb = m[i1], vb = v[i1];
  const int64_t step = step_dev[0] + 1;
  *step_out = step;
  const AdamStep st = adam_step_scalars(step, lr, b1, b2, lrd, clipped);
  if (ok) {
    adam_update<T>(ga, pa_, ma, va, st, b1, b2, eps, wd, clip, clipped);
    m[i0] = ma; v[i0] = va; p[i0] = pa_;
    adam_update<T>(gb, pb, mb, vb, st, b1, b2, eps, wd, clip, clipped);
    m[i1] = mb; v[i1] = vb; p[i1] = pb;
    if (zero_grad) { g[i0] = T(0); g[i1] = T(0); }
  }
  for (int64_t k = t + 256; k < n; k += 256) {
    for (int h = 0; h < 2; ++h) {
      const int64_t i = (h == 0 ? o0 : o1) + k;
      T pi = p[i], mi = m[i], vi = v[i];
      adam_update<T>(g[i], pi, mi, vi, st, b1, b2, eps, wd, clip, clipped);
      m[i] = mi; v[i] = vi; p[i] = pi;
      if (zero_grad) g[i] = T(0);
    }
  }
}

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
  int64_t step;
  adam_range<T>(vb * 256 + threadIdx.x, nvb * 256, n, p, g, m, v, step_dev, lr, b1, b2, eps, wd, clip,
                lrd, clipped, zero_grad, &step);
  __syncthreads();
  if (threadIdx.x == 0) {
    bool last = true;
    if (nvb > 1) {       // (a one-workgroup launch is its own last arrival: no atomic round trip)
      const unsigned long long ticket =
          atomicAdd(reinterpret_cast<unsigned long long*>(step_dev + 1), 1ull);
      last = ticket == (unsigned long long)nvb - 1;
      if (last) step_dev[1] = 0;
    }
    if (last) {
      step_dev[0] = step;
      publish_to_host(pub);
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
