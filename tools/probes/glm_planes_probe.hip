// Phase anatomy of glm_planes_kernel (developer tool): the product kernel header compiled with
// PA_GLMP_STAMP, one wave per workgroup accumulates shader-clock time per loop phase.
#define PA_GLMP_STAMP 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../pyro_amd/csrc/glm_planes.h"
namespace pa { int cu_count() { return 256; } }

template <int NB, int OCC>
static void run(const unsigned char* img, const float* y, const float* w, const float* b, int64_t N, int D, int P,
                float* part, int bpc) {
  using namespace pa;
  auto k = glm_planes_kernel<2, NB, OCC>;
  constexpr int lds = GlmPlCfg<2, NB>::LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int64_t nst = ((N + 31) / 32 + 1) / 2;
  const int nblocks = 256 * bpc;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(nblocks, 1), dim3(256), lds, 0, img, y, w, b, N, D, P, nst, part, 256, pa::GlmFinArgs{});
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(nblocks, 1), dim3(256), lds, 0, img, y, w, b, N, D, P, nst, part, 256, pa::GlmFinArgs{});
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> dbg((size_t)nblocks * 4 * 16);
  (void)hipMemcpy(dbg.data(), reinterpret_cast<uint64_t*>(part) + (1 << 20), dbg.size() * 8, hipMemcpyDeviceToHost);
  double ph[8] = {0}, tot = 0, wall = 0, cnt = 0;
  for (int i = 0; i < nblocks * 4; ++i) {
    for (int j = 0; j < 8; ++j) ph[j] += (double)dbg[(size_t)i * 16 + j];
    tot += (double)dbg[(size_t)i * 16 + 8]; wall += (double)dbg[(size_t)i * 16 + 9]; cnt += (double)dbg[(size_t)i * 16 + 10];
  }
  const double nw = nblocks * 4.0;
  printf("ring=%d wg/CU=%d: %7.1f us/launch; per wave: %.0f tiles, loop span %.0f shader cycles = %.1f us of the 100 MHz clock -> %.2f GHz\n",
         NB, bpc, ms * 100.0, cnt / nw, tot / nw, wall / nw / 100.0, (tot / nw) / (wall / nw / 100.0) / 1e3);
  {
    uint64_t t0 = ~0ull;
    for (int i = 0; i < nblocks * 4; ++i) t0 = dbg[(size_t)i * 16 + 11] < t0 ? dbg[(size_t)i * 16 + 11] : t0;
    double mx[4] = {0, 0, 0, 0}, av[4] = {0, 0, 0, 0};
    for (int i = 0; i < nblocks * 4; ++i)
      for (int j = 0; j < 4; ++j) {
        const double v = (double)(dbg[(size_t)i * 16 + 11 + j] - t0) / 100.0;
        av[j] += v / (nblocks * 4); mx[j] = v > mx[j] ? v : mx[j];
      }
    printf("    wave timeline (us after the first wave's entry), mean / max: entry %.1f / %.1f, loop start %.1f / %.1f, loop end %.1f / %.1f, exit %.1f / %.1f\n",
           av[0], mx[0], av[1], mx[1], av[2], mx[2], av[3], mx[3]);
  }
  {
    uint64_t t0 = ~0ull;
    for (int i = 0; i < nblocks * 4; ++i) t0 = dbg[(size_t)i * 16 + 11] < t0 ? dbg[(size_t)i * 16 + 11] : t0;
    for (int slot = 0; slot < bpc; ++slot) {
      double e = 0, mn = 1e9, mx = 0; int n = 0;
      for (int b = slot * 256; b < (slot + 1) * 256; ++b)
        for (int wv = 0; wv < 4; ++wv) {
          const double v = (double)(dbg[(size_t)(b * 4 + wv) * 16 + 13] - t0) / 100.0;
          e += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; ++n;
        }
      printf("    blocks %4d..%4d: loop end mean %.1f min %.1f max %.1f us\n", slot * 256, slot * 256 + 255, e / n, mn, mx);
    }
    for (int x = 0; x < 8; ++x) {
      double e = 0; int n = 0;
      for (int b = x; b < nblocks; b += 8) { e += (double)(dbg[(size_t)(b * 4) * 16 + 13] - t0) / 100.0; ++n; }
      printf("    XCD %d: loop end mean %.1f us\n", x, e / n);
    }
  }
  const char* names[8] = {"wait vmcnt", "barrier", "dma issue+GEMM1", "elem kh0", "GEMM2 kh0", "elem kh1", "GEMM2 kh1", "loop overhead"};
  for (int j = 0; j < 8; ++j) printf("    %-18s %8.0f cycles per tile (%4.1f %%)\n", names[j], ph[j] / cnt, 100.0 * ph[j] / tot);
}

int main() {
  const int64_t N = 1000000; const int D = 32, P = 64;
  std::vector<float> hX(N * D), hy(N), hw(P * D), hb(P);
  srand(1);
  for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 3.4f;
  for (auto& v : hy) v = rand() & 1;
  for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
  for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f);
  float *X, *y, *w, *b, *part; unsigned char* img;
  const int64_t nt = ((N + 31) / 32 + 3) / 4 * 4;
  (void)hipMalloc(&X, hX.size() * 4); (void)hipMalloc(&y, hy.size() * 4); (void)hipMalloc(&w, hw.size() * 4); (void)hipMalloc(&b, hb.size() * 4);
  (void)hipMalloc(&img, (size_t)nt * pa::GLMP_TILE);
  (void)hipMalloc(&part, (size_t)64 << 20);
  (void)hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(pa::glm_pack_planes_kernel, dim3((unsigned)((nt * 128 + 255) / 256)), dim3(256), 0, 0, X, N, D, nt, img);
  run<3, 3>(img, y, w, b, N, D, P, part, 3);
  run<4, 2>(img, y, w, b, N, D, P, part, 2);
  run<3, 3>(img, y, w, b, N, D, P, part, 1);
  return 0;
}
