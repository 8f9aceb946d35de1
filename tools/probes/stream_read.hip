// Probe (developer tool): achievable HBM read bandwidth of the GLM kernel's access pattern
// (persistent waves, one 4 KB tile = 4 x float4 per lane, prefetched one tile ahead) against a
// plain grid-stride float4 read, for several block counts / bytes in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int NLD, int DEPTH>
__global__ __launch_bounds__(256) void persist(const float4* __restrict__ X, int64_t nvec, int64_t iters, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t stride = (int64_t)gridDim.x * 4;
  float4 st[DEPTH][NLD];
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      int64_t e = (tile + d * stride) * (NLD * 64) + j * 64 + lane;
      st[d][j] = X[e < nvec ? e : 0];
    }
  for (int64_t it = 0; it < iters; it += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int j = 0; j < NLD; ++j) acc += st[d][j].x + st[d][j].y + st[d][j].z + st[d][j].w;
#pragma unroll
      for (int j = 0; j < NLD; ++j) {
        int64_t e = (tile + (int64_t)(it + d + DEPTH) * stride) * (NLD * 64) + j * 64 + lane;
        st[d][j] = X[e < nvec ? e : 0];
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void gridstride(const float4* __restrict__ X, int64_t nvec, float* out) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float4 v = X[i];
    acc += v.x + v.y + v.z + v.w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class F>
static float timeit(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / 20;
}

int main() {
  for (int64_t MB : {128, 1280}) {
    const int64_t nvec = MB * 1000000 / 16;
    float4* X; float* out;
    (void)hipMalloc(&X, nvec * 16); (void)hipMalloc(&out, (size_t)65536 * 256 * 4);
    (void)hipMemset(X, 0, nvec * 16);
    printf("---- %ld MB\n", (long)MB);
    for (int nb : {256, 512, 768, 1024, 2048}) {
      const int64_t ntiles = (nvec + 255) / 256;
      const int64_t iters = (ntiles + nb * 4 - 1) / (nb * 4);
      float t1 = timeit([&] { hipLaunchKernelGGL((persist<4, 1>), dim3(nb), dim3(256), 0, 0, X, nvec, iters, out); });
      float t2 = timeit([&] { hipLaunchKernelGGL((persist<4, 2>), dim3(nb), dim3(256), 0, 0, X, nvec, (iters + 1) / 2 * 2, out); });
      float t4 = timeit([&] { hipLaunchKernelGGL((persist<4, 4>), dim3(nb), dim3(256), 0, 0, X, nvec, (iters + 3) / 4 * 4, out); });
      printf("persist nb=%4d: depth1 %7.1f us %5.2f TB/s | depth2 %7.1f us %5.2f TB/s | depth4 %7.1f us %5.2f TB/s\n", nb,
             t1, nvec * 16 / t1 / 1e6, t2, nvec * 16 / t2 / 1e6, t4, nvec * 16 / t4 / 1e6);
    }
    for (int nb : {1024, 4096, 16384, 65536}) {
      float t = timeit([&] { hipLaunchKernelGGL(gridstride, dim3(nb), dim3(256), 0, 0, X, nvec, out); });
      printf("gridstride nb=%6d: %7.1f us %5.2f TB/s\n", nb, t, nvec * 16 / t / 1e6);
    }
    (void)hipFree(X); (void)hipFree(out);
  }
  return 0;
}
