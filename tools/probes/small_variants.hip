// Probe (developer tool): where do ~10 us go in a two-workgroup reduction kernel?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct Site { const float* rho; const float* eps; const float* dz; const float* dsc; float* dloc; float* drho; long n; int accumulate; };
struct Args { int ns; Site s[16]; };

template <typename S>
__device__ __forceinline__ S kernarg_load(uint32_t off) {
  typedef __attribute__((address_space(4))) const uint32_t* cptr;
  typedef __attribute__((address_space(4))) const char* cbytes;
  cptr p = (cptr)((cbytes)__builtin_amdgcn_kernarg_segment_ptr() + off);
  union { S s; uint32_t w[sizeof(S) / 4]; } u;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(S) / 4); ++i) u.w[i] = p[i];
  return u.s;
}

template <int V>
__global__ __launch_bounds__(256) void k(const Args a_by_value, long P) {
  __shared__ double red_l[256], red_s[256];
  const Site s = kernarg_load<Site>(offsetof(Args, s) + blockIdx.x * sizeof(Site));
  const uint32_t n = (uint32_t)s.n, t = threadIdx.x, PP = (uint32_t)P;
  const uint32_t tk = n < 256 ? n : 256, ng = 256 / tk, c0 = t % tk, g = t / tk;
  if (V == 3) { if (t == 0) s.dloc[0] = 1.f; return; }
  for (uint32_t cb = 0; cb < n; cb += tk) {
    const uint32_t c = cb + c0;
    const bool okc = g < ng && c < n;
    float al = 0.f, as = 0.f;
    if (V != 2)
      for (uint32_t pb = g; pb < PP; pb += 8 * ng) {
        float gz[8], ev[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t p = pb + u * ng;
          const uint32_t o = (okc && p < PP) ? p * n + c : 0u;
          gz[u] = s.dz[o]; ev[u] = s.eps[o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const bool ok = okc && (pb + u * ng) < PP; al += ok ? gz[u] : 0.f; as += ok ? gz[u] * ev[u] : 0.f; }
      }
    if (V == 1) { if (okc) { s.dloc[c] = al; s.drho[c] = as; } continue; }
    __syncthreads();
    red_l[t] = (double)al; red_s[t] = (double)as;
    __syncthreads();
    if (g == 0 && c < n) {
      double sl = 0.0, ss = 0.0;
      for (uint32_t j = 0; j < ng; ++j) { sl += red_l[j * tk + c0]; ss += red_s[j * tk + c0]; }
      if (V != 4) ss += (double)s.dsc[c];
      const float x = s.rho[c];
      const double sig = (V == 5) ? 0.5 : (x > 20.f ? 1.0 : (double)(1.f / (1.f + expf(-x))));
      s.dloc[c] = (float)sl;
      s.drho[c] = (float)(ss * sig);
    }
  }
}

template <int V> float run(const Args& a, long P, int ns) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<V>, dim3(ns), dim3(256), 0, 0, a, P);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(k<V>, dim3(ns), dim3(256), 0, 0, a, P);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / 500;
}
__global__ void busy(float* x, int iters) { float v = x[threadIdx.x]; for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.1f; x[threadIdx.x] = v; }

int main() {
  const long P = 64; float* buf; (void)hipMalloc(&buf, 1 << 20); (void)hipMemset(buf, 0, 1 << 20);
  Args a; a.ns = 2;
  a.s[0] = Site{buf, buf + 4096, buf + 8192, buf + 12288, buf + 16384, buf + 20480, 32, 0};
  a.s[1] = Site{buf + 100, buf + 4196, buf + 8292, buf + 12388, buf + 16484, buf + 20580, 1, 0};
  printf("V0 full %.2f us\n", run<0>(a, P, 2));
  printf("V1 no LDS reduce %.2f us\n", run<1>(a, P, 2));
  printf("V2 no loads %.2f us\n", run<2>(a, P, 2));
  printf("V3 kernarg only %.2f us\n", run<3>(a, P, 2));
  printf("V4 no dsc load %.2f us\n", run<4>(a, P, 2));
  printf("V5 no expf %.2f us\n", run<5>(a, P, 2));
  printf("V0 one site %.2f us\n", run<0>(a, P, 1));
  return 0;
}
