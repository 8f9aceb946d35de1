// nuts_common.h -- scalar helpers shared by the NUTS kernels (nuts.hip: fused Gaussian
// potential, one wave per chain; nuts_tree.hip: tree state machine for arbitrary potentials).
#pragma once
#include "common.h"

namespace pa {

constexpr int NUTS_MAX_DEPTH = 10;

template <typename T> struct Num;
template <> struct Num<float> {
  // f32 chains: the hardware transcendentals (v_exp_f32 / v_log_f32, ~1 ulp) instead of the
  // ~60-instruction library routines -- these feed acceptance probabilities and log-weights that
  // are compared with uniform draws, every lane of the wave computes them redundantly, and a tree
  // merge needs three of them.  The float64 kernels (the ones held to the oracle chain for chain)
  // keep the exact library functions.
  static __device__ __forceinline__ float exp_(float x) {
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
  }
  static __device__ __forceinline__ float log_(float x) {
    return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
  }
  static __device__ __forceinline__ float log1p_(float x) { return log_(1.0f + x); }
  static __device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
  static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
};
template <> struct Num<double> {
  static __device__ __forceinline__ double exp_(double x) { return exp(x); }
  static __device__ __forceinline__ double log_(double x) { return log(x); }
  static __device__ __forceinline__ double log1p_(double x) { return log1p(x); }
  static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
  static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
};

__device__ __forceinline__ float uni(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ double uni(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <typename T>
__device__ __forceinline__ T uniform_from(const u32x4& b, int second) {
  if constexpr (sizeof(T) == 4)
    return u32_to_unit_f32(second ? b.z : b.x);
  else
    return second ? u32x2_to_unit_f64(b.z, b.w) : u32x2_to_unit_f64(b.x, b.y);
}

template <typename T>
__device__ __forceinline__ T logaddexp_ref(T x, T y) {  // nuts.py:15-17
  const T mn = x < y ? x : y, mx = x < y ? y : x;
  return Num<T>::log1p_(Num<T>::exp_(mn - mx)) + mx;
}

template <typename T>
__device__ __forceinline__ T philox_normal_t(uint64_t seed, uint64_t offset, uint64_t i,
                                             uint64_t stream_id) {
  if constexpr (sizeof(T) == 4) return philox_normal_f32(seed, offset, i, stream_id);
  else return philox_normal_f64(seed, offset, i, stream_id);
}

}  // namespace pa
