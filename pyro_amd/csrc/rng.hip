// rng.hip -- counter-based Philox4x32-10 draws (normal / uniform), see include/pyro_amd.h.
// Memory-bound element-wise fill: 16 B stores per lane, grid-stride, one Philox block per
// 4 (f32) or 2 (f64) outputs so out[i] never depends on the launch geometry.
#include "common.h"
#include "optim_dev.h"

namespace pa {

template <bool NORMAL>
__global__ __launch_bounds__(256) void philox_fill_f32(float* __restrict__ out, int64_t n,
                                                       uint64_t seed, uint64_t offset,
                                                       const uint64_t* __restrict__ offset_dev) {
  if (offset_dev) offset += *offset_dev;
  const int64_t nblk = (n + 3) >> 2;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk;
       b += (int64_t)gridDim.x * blockDim.x) {
    u32x4 x = philox4x32_10(seed, offset + (uint64_t)b, 0);
    float v0, v1, v2, v3;
    if (NORMAL) {
      box_muller_f32(u32_to_unit_f32(x.x), u32_to_unit_f32(x.y), v0, v1);
      box_muller_f32(u32_to_unit_f32(x.z), u32_to_unit_f32(x.w), v2, v3);
    } else {
      v0 = u32_to_unit_f32(x.x); v1 = u32_to_unit_f32(x.y);
      v2 = u32_to_unit_f32(x.z); v3 = u32_to_unit_f32(x.w);
    }
    const int64_t i = b << 2;
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
      *reinterpret_cast<float4*>(out + i) = make_float4(v0, v1, v2, v3);
    } else {
      if (i < n) out[i] = v0;
      if (i + 1 < n) out[i + 1] = v1;
      if (i + 2 < n) out[i + 2] = v2;
      if (i + 3 < n) out[i + 3] = v3;
    }
  }
}

template <bool NORMAL>
__global__ __launch_bounds__(256) void philox_fill_f64(double* __restrict__ out, int64_t n,
                                                       uint64_t seed, uint64_t offset,
                                                       const uint64_t* __restrict__ offset_dev) {
  if (offset_dev) offset += *offset_dev;
  const int64_t nblk = (n + 1) >> 1;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk;
       b += (int64_t)gridDim.x * blockDim.x) {
    u32x4 x = philox4x32_10(seed, offset + (uint64_t)b, 0);
    double v0, v1;
    if (NORMAL) {
      box_muller_f64(u32x2_to_unit_f64(x.x, x.y), u32x2_to_unit_f64(x.z, x.w), v0, v1);
    } else {
      v0 = u32x2_to_unit_f64(x.x, x.y);
      v1 = u32x2_to_unit_f64(x.z, x.w);
    }
    const int64_t i = b << 1;
    if (i < n) out[i] = v0;
    if (i + 1 < n) out[i + 1] = v1;
  }
}

__global__ void counter_add_kernel(uint64_t* c, uint64_t inc) { *c += inc; }

// End-of-step node of a captured training step: advances the Philox base counter of the graph
// and hands the step's scalar result to the host through pinned (device-mapped) memory: the value
// first, then -- behind a system-scope fence -- a sequence number the host polls.  The host reads
// the result a few microseconds after the kernel ran, without a stream synchronisation or a
// device-to-host copy call.
template <typename T>
__global__ void publish_scalar_kernel(const T* __restrict__ src, double* host_value,
                                      uint64_t* host_seq, uint64_t* counter, uint64_t inc) {
  publish_to_host(AdamPublish{src, sizeof(T) == 4 ? PA_F32 : PA_F64, host_value, host_seq, counter,
                              inc});
}

template <bool NORMAL>
static int fill(void* out, int64_t n, int dtype, uint64_t seed, uint64_t offset,
                const uint64_t* offset_dev, pa_stream_t stream) {
  PA_REQUIRE(n >= 0, "philox: n=%lld < 0", (long long)n);
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "philox: bad dtype %d", dtype);
  if (n == 0) return PA_OK;
  PA_REQUIRE(out != nullptr, "philox: out is NULL");
  const int64_t per = dtype == PA_F32 ? 4 : 2;
  const int64_t nblk = (n + per - 1) / per;
  int64_t grid = (nblk + 255) / 256;
  const int64_t cap = (int64_t)cu_count() * 8;
  if (grid > cap) grid = cap;
  if (dtype == PA_F32)
    hipLaunchKernelGGL(philox_fill_f32<NORMAL>, dim3((unsigned)grid), dim3(256), 0,
                       as_stream(stream), (float*)out, n, seed, offset, offset_dev);
  else
    hipLaunchKernelGGL(philox_fill_f64<NORMAL>, dim3((unsigned)grid), dim3(256), 0,
                       as_stream(stream), (double*)out, n, seed, offset, offset_dev);
  return check_launch("philox_fill");
}

}  // namespace pa

extern "C" {

int pa_philox_normal(void* out, int64_t n, int dtype, uint64_t seed, uint64_t offset,
                     const uint64_t* offset_dev, pa_stream_t stream) {
  return pa::fill<true>(out, n, dtype, seed, offset, offset_dev, stream);
}
int pa_philox_uniform(void* out, int64_t n, int dtype, uint64_t seed, uint64_t offset,
                      const uint64_t* offset_dev, pa_stream_t stream) {
  return pa::fill<false>(out, n, dtype, seed, offset, offset_dev, stream);
}
int pa_publish_scalar(int dtype, const void* src, double* host_value, uint64_t* host_seq,
                      uint64_t* counter, uint64_t inc, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "publish_scalar: bad dtype %d", dtype);
  PA_REQUIRE(src && host_value && host_seq, "publish_scalar: NULL pointer");
  if (dtype == PA_F32)
    hipLaunchKernelGGL((pa::publish_scalar_kernel<float>), dim3(1), dim3(1), 0, pa::as_stream(stream),
                       (const float*)src, host_value, host_seq, counter, inc);
  else
    hipLaunchKernelGGL((pa::publish_scalar_kernel<double>), dim3(1), dim3(1), 0,
                       pa::as_stream(stream), (const double*)src, host_value, host_seq, counter, inc);
  return pa::check_launch("publish_scalar");
}

int pa_counter_add(uint64_t* counter, uint64_t inc, pa_stream_t stream) {
  PA_REQUIRE(counter != nullptr, "counter_add: NULL counter");
  hipLaunchKernelGGL(pa::counter_add_kernel, dim3(1), dim3(1), 0, pa::as_stream(stream), counter,
                     inc);
  return pa::check_launch("counter_add");
}

}  // extern "C"
