// One timing unit of tools/probes/glm_variants (developer tool): the product kernel header compiled
// with one PA_GLM_PROBE_* macro that removes a part of the per-tile work, to read off marginal costs.
#include <hip/hip_runtime.h>
#include "../../pyro_amd/csrc/glm_bf16.h"
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
extern "C" float CAT(run_, PROBE_NAME)(const float* X, const float* y, const float* w, const float* b,
                                       int64_t N, int D, int P, float* part, int nblocks, int reps) {
  using namespace pa;
  auto k = glm_bernoulli_bf16_kernel<1, 2, false, false>;
  constexpr int lds = GlmBfCfg<1, 2>::LDS_BYTES;
  const int64_t ntiles = (N + 31) / 32;
  const int64_t iters = (ntiles + (int64_t)nblocks * GLMB_WAVES - 1) / ((int64_t)nblocks * GLMB_WAVES);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL(k, dim3(nblocks, 1), dim3(256), lds, 0, X, y, w, b, (const uint8_t*)nullptr, N, D, P, iters, part, (const int64_t*)nullptr, 1);
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(k, dim3(nblocks, 1), dim3(256), lds, 0, X, y, w, b, (const uint8_t*)nullptr, N, D, P, iters, part, (const int64_t*)nullptr, 1);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}
