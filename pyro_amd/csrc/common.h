// common.h -- shared device/host helpers for the gfx950 kernels of libpyro_amd.so.
// CDNA4 only: wavefront = 64 lanes, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pyro_amd.h"

namespace pa {

// 1 / (1 + exp(-h)) on v_exp_f32 / v_rcp_f32 (1 ulp each; exp2 of a large argument is +inf and rcp(inf) = 0, so
// both limits come out right): the activation in the epilogue of the Linear-layer kernels (tall.hip, bow.hip)
__device__ __forceinline__ float fast_sigmoid(float h) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * h));
}


constexpr int WAVE = 64;

// ---- host-side error plumbing ------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
// hipGetLastError() after a launch -> PA_ERR_LAUNCH
int check_launch(const char* what);
int cu_count();
// the step gate (pa_gate_scope): the word gate-aware kernels poll at their start, or nullptr; a
// launcher of such a kernel calls gate_aware_launch() once per launch it passes gate_word() to
const int64_t* gate_word();
void gate_aware_launch();
// emits a gate registered with pa_gate_defer (no-op otherwise); called in front of the chained tail
void gate_emit_deferred(hipStream_t s);
// true (and the two events) if pa_profile_bracket_next(tag, ...) is pending on this thread
bool take_bracket(int tag, hipEvent_t* start, hipEvent_t* stop);

#define PA_REQUIRE(cond, ...)                                  \
  do {                                                         \
    if (!(cond)) return ::pa::fail(PA_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// ---- wave-level reductions (64 lanes) ------------------------------------------------------------
// Sum over the wave in a FIXED order, result uniform in every lane.  Within a row of 16 lanes the
// butterfly runs on DPP (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: ~8 cycles a step);
// the four row sums are then read with v_readlane and added as scalars.  (__shfl_xor lowers to
// ds_bpermute: six dependent LDS round trips, ~500 cycles per reduction -- the NUTS transition
// kernel does half a dozen dot products per leapfrog step.)
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __int_as_float(dpp_mov<CTRL>(__float_as_int(v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
  const int lo = dpp_mov<CTRL>(__double2loint(v)), hi = dpp_mov<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_get(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double lane_get(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  if constexpr (sizeof(T) == 4 || sizeof(T) == 8) {
    v += dpp_get<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_get<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_get<0x141>(v);   // row_half_mirror: the other quad of the 8
    v += dpp_get<0x140>(v);   // row_mirror: the other half of the row of 16
    const T s0 = lane_get(v, 0), s1 = lane_get(v, 16), s2 = lane_get(v, 32), s3 = lane_get(v, 48);
    return (s0 + s1) + (s2 + s3);
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  }
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T u = __shfl_xor(v, o, 64);
    v = u > v ? u : v;
  }
  return v;
}

// Sum of doubles over the first ``nw`` waves of the block (all of them take part; any further
// waves must have left the kernel); result valid in thread 0.
__device__ __forceinline__ double block_sum_f64_waves(double v, double* smem /* >= nw doubles */,
                                                      int nw) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) t += smem[i];  // fixed order -> deterministic
  return t;
}

// Block-wide sum of doubles for blocks of up to 1024 threads; result valid in thread 0.
__device__ __forceinline__ double block_sum_f64(double v, double* smem /* >= 16 doubles */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) t += smem[i];  // fixed order -> deterministic
  return t;
}

// ---- Philox4x32-10 (Salmon et al. 2011), the same stream the oracle restates ---------------
struct u32x4 {
  uint32_t x, y, z, w;
};
__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint64_t seed, uint64_t ctr_lo,
                                                        uint64_t ctr_hi) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
  uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return u32x4{c0, c1, c2, c3};
}

// u32 -> uniform in the OPEN interval (0,1): 24 random bits, centred bins (f32 exact).
__host__ __device__ __forceinline__ float u32_to_unit_f32(uint32_t x) {
  return (float)(x >> 8) * 5.9604644775390625e-08f + 2.98023223876953125e-08f;
}
// two u32 -> uniform in (0,1) with 53 random bits.
__host__ __device__ __forceinline__ double u32x2_to_unit_f64(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 1.1102230246251565e-16 +
         5.551115123125783e-17;
}

// Box-Muller. f32: (u1,u2) -> two normals.
__device__ __forceinline__ void box_muller_f32(float u1, float u2, float& n0, float& n1) {
  float r = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincosf(6.283185307179586f * u2, &s, &c);
  n0 = r * c;
  n1 = r * s;
}
__device__ __forceinline__ void box_muller_f64(double u1, double u2, double& n0, double& n1) {
  double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincos(6.283185307179586 * u2, &s, &c);
  n0 = r * c;
  n1 = r * s;
}

// Standard normal number `i` of the (seed, offset) stream. f32: block i/4, lane i%4
// (lanes 0,1 = cos,sin of pair (x,y); lanes 2,3 = cos,sin of pair (z,w)).
__device__ __forceinline__ float philox_normal_f32(uint64_t seed, uint64_t offset, uint64_t i,
                                                   uint64_t stream_id = 0) {
  u32x4 b = philox4x32_10(seed, offset + (i >> 2), stream_id);
  const int l = (int)(i & 3);
  float u1 = u32_to_unit_f32(l < 2 ? b.x : b.z), u2 = u32_to_unit_f32(l < 2 ? b.y : b.w);
  float n0, n1;
  box_muller_f32(u1, u2, n0, n1);
  return (l & 1) ? n1 : n0;
}
// f64: block i/2, lane i%2.
__device__ __forceinline__ double philox_normal_f64(uint64_t seed, uint64_t offset, uint64_t i,
                                                    uint64_t stream_id = 0) {
  u32x4 b = philox4x32_10(seed, offset + (i >> 1), stream_id);
  double u1 = u32x2_to_unit_f64(b.x, b.y), u2 = u32x2_to_unit_f64(b.z, b.w);
  double n0, n1;
  box_muller_f64(u1, u2, n0, n1);
  return (i & 1) ? n1 : n0;
}

template <typename T>
struct ViewT {
  const T* p;
  int64_t sr, sc;
  __device__ __forceinline__ T at(int64_t r, int64_t c) const { return p[r * sr + c * sc]; }
};
template <typename T>
inline ViewT<T> as_view(const pa_view2d& v) {
  return ViewT<T>{(const T*)v.ptr, v.stride_row, v.stride_col};
}

// The stream a launcher is about to launch on.  Defined in chain.hip: phases recorded for the chained
// tail of an SVI step (chain.h) are launched first -- whatever follows may read what they write.
hipStream_t as_stream(pa_stream_t s);

}  // namespace pa
