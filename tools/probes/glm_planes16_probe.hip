// Timing ablations of glm_planes_f16_kernel (developer tool): the product kernel header compiled with
// one part of the per-tile work removed (-DPA_GLMH_ABL_*), N = 1e6, D = 32, P = 64, 512 workgroups.
// The numbers a variant produces are wrong on purpose; only its duration means anything.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../pyro_amd/csrc/glm_planes16w.h"
namespace pa { int cu_count() { return 256; } }

int main(int argc, char** argv) {
  using namespace pa;
  const int64_t N = 1000000; const int D = 32, P = 64;
  const int bpc = argc > 1 ? atoi(argv[1]) : 2;
  std::vector<float> hX(N * D), hy(N), hw(P * D), hb(P);
  srand(1);
  for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 3.4f;
  for (auto& v : hy) v = rand() & 1;
  for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
  for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f);
  float *X, *y, *w, *b, *part; unsigned char* img;
  const int64_t nt = ((N + 31) / 32 + 3) / 4 * 4;
  (void)hipMalloc(&X, hX.size() * 4); (void)hipMalloc(&y, hy.size() * 4); (void)hipMalloc(&w, hw.size() * 4); (void)hipMalloc(&b, hb.size() * 4);
  (void)hipMalloc(&img, (size_t)nt * GLMH_TILE + GLMH_TRAILER);
  (void)hipMalloc(&part, (size_t)64 << 20);
  (void)hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  uint32_t* trailer = reinterpret_cast<uint32_t*>(img + (size_t)nt * GLMH_TILE);
  (void)hipMemset(trailer, 0, GLMH_TRAILER);
  hipLaunchKernelGGL(glm_absmax_kernel, dim3(1024), dim3(256), 0, 0, X, N, D, trailer);
  hipLaunchKernelGGL(glm_pack_planes_f16_kernel, dim3((unsigned)((nt * 128 + 255) / 256)), dim3(256), 0, 0, X, N, D, nt, img, trailer);
#ifndef PROBE_NB
#define PROBE_NB 3
#endif
#ifndef PROBE_PRIV
#define PROBE_PRIV false
#endif
#ifndef PROBE_OCC
#define PROBE_OCC 3
#endif
#ifdef PROBE_WIDE
  auto k = glm_planes_f16w_kernel<PROBE_NB>;
  constexpr int lds = GlmWCfg<PROBE_NB>::LDS_BYTES;
#else
  auto k = glm_planes_f16_kernel<PROBE_NB, PROBE_OCC, false, PROBE_PRIV>;
  constexpr int lds = GlmHCfg<PROBE_NB, PROBE_PRIV>::LDS_BYTES;
#endif
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int64_t nst = argc > 3 ? atoll(argv[3]) : ((N + 31) / 32 + 1) / 2;   // (0: prologue + epilogue only, NODMA builds)
  const int nblocks = 256 * bpc;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto launch = [&]() {
#ifdef PROBE_WIDE
    const int64_t ngrp = argc > 3 ? atoll(argv[3]) : (N + 127) / 128;
    hipLaunchKernelGGL(k, dim3(nblocks, 1), dim3(256), lds, 0, img, y, w, b, N, D, P, ngrp, part, trailer,
                       (unsigned long long*)nullptr, (const int64_t*)nullptr);
#else
    hipLaunchKernelGGL(k, dim3(nblocks, 1), dim3(256), lds, 0, img, y, w, b, N, D, P, nst, part, 256, trailer,
                       (unsigned long long*)nullptr, GlmGroupArgs{nullptr, nullptr, 1}, (const int64_t*)nullptr, (const double*)nullptr, GlmDraw{});
#endif
  };
  for (int i = 0; i < 5; ++i) launch();
  float best = 1e9f, tot = 0;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best; tot += ms;
  }
  printf("%-28s wg/CU=%d: %6.1f us/launch (best of 5 x 20 back-to-back), mean %6.1f\n", argc > 2 ? argv[2] : "base", bpc,
         best * 50.0, tot * 10.0);
  return 0;
}
