// glm_finalize.h -- the deterministic fp64 reduction of the GLM kernels' per-workgroup partial records
// (shared by glm.hip's launchers and the chained tail of an SVI step, chain.hip).
#pragma once
#include "common.h"

namespace pa {

// floats in one block's partial record: raw MFMA accumulator tiles + ll + gb
template <int DT, int PT>
constexpr int glm_record_floats() { return PT * DT * 1024 + 2 * PT * 32; }

// out[j] = scale * sum_blocks partial[block][slot(j)], fp64 accumulation, fixed order.
// Output order: gw[P,D] then ll[P] then gb[P].  One workgroup = FIN_OUT outputs x FIN_GROUPS
// record groups (thread (j, s) sums records s, s + FIN_GROUPS, ...; the groups are then combined
// through LDS in a fixed order): many small workgroups so that the ~4 MB of partial records are
// pulled by the whole chip rather than by a few dozen CUs.
constexpr int FIN_OUT = 8, FIN_GROUPS = 32;
// (tid = 0..255 within the virtual workgroup `vb`; sm = its FIN_GROUPS x FIN_OUT doubles of LDS: a
//  1024-thread workgroup of the chained tail runs four virtual workgroups side by side)
template <int DT, int PT>
__device__ __forceinline__ void glm_finalize_body(
    int64_t vb, const float* __restrict__ part, int nblocks, int npass, int D, int P, double scale,
    float* __restrict__ ll, float* __restrict__ gw, float* __restrict__ gb, double ll_offset, int tid,
    double (*sm)[FIN_OUT]) {
  constexpr int REC = glm_record_floats<DT, PT>();
  const int jj = tid % FIN_OUT, s = tid / FIN_OUT;
  const int64_t J = (int64_t)P * D + 2 * P;
  const int64_t j = vb * FIN_OUT + jj;
  double acc = 0.0;
  if (j < J) {
    int p, slot;
    if (j < (int64_t)P * D) {
      p = (int)(j / D);
      const int d = (int)(j % D);
      const int pl = p % (32 * PT), pt = pl >> 5, i = pl & 31, dt = d >> 5, c = d & 31;
      const int hh = (i >> 2) & 1, reg = (i & 3) + 4 * (i >> 3);
      slot = ((pt * DT + dt) * 16 + reg) * 64 + c + 32 * hh;
    } else {
      const int64_t k = j - (int64_t)P * D;
      const int which = k >= P ? 1 : 0;
      p = (int)(k - (int64_t)which * P);
      const int pl = p % (32 * PT), pt = pl >> 5, i = pl & 31;
      slot = PT * DT * 1024 + (2 * pt + which) * 32 + i;
    }
    const int pass = p / (32 * PT);
    const float* base = part + (int64_t)pass * nblocks * REC + slot;
    // the records are summed in increasing order whatever the batching: what a batch buys is that
    // its loads are in flight together (a thread's 24 records of a 768-workgroup launch: ONE
    // memory round trip instead of three)
    int blk = s;
    {
      float v[24];
      for (; blk + 23 * FIN_GROUPS < nblocks; blk += 24 * FIN_GROUPS) {
#pragma unroll
        for (int u = 0; u < 24; ++u) v[u] = base[(int64_t)(blk + u * FIN_GROUPS) * REC];
#pragma unroll
        for (int u = 0; u < 24; ++u) acc += (double)v[u];
      }
    }
#ifndef PA_FIN_NO16
    {
      // (a thread's 16 records of the 512-workgroup default launch: one round trip, not two)
      float v[16];
      for (; blk + 15 * FIN_GROUPS < nblocks; blk += 16 * FIN_GROUPS) {
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = base[(int64_t)(blk + u * FIN_GROUPS) * REC];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (double)v[u];
      }
    }
#endif
    float v[8];
    for (; blk + 7 * FIN_GROUPS < nblocks; blk += 8 * FIN_GROUPS) {   // 8 loads in flight
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(blk + u * FIN_GROUPS) * REC];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; blk < nblocks; blk += FIN_GROUPS) acc += (double)base[(int64_t)blk * REC];
  }
  sm[s][jj] = acc;
  __syncthreads();
  if (s == 0 && j < J) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_GROUPS; ++k) t += sm[k][jj];
    const float v = (float)(t * scale);
    if (j < (int64_t)P * D) gw[j] = v;
    else if (j < (int64_t)P * D + P) ll[j - (int64_t)P * D] = (float)((t + ll_offset) * scale);
    else gb[j - (int64_t)P * D - P] = v;
  }
}

template <int DT, int PT>
__global__ __launch_bounds__(FIN_OUT * FIN_GROUPS) void glm_finalize_kernel(
    const float* __restrict__ part, int nblocks, int npass, int D, int P, double scale,
    float* __restrict__ ll, float* __restrict__ gw, float* __restrict__ gb, double ll_offset) {
  __shared__ double sm[FIN_GROUPS][FIN_OUT];
  glm_finalize_body<DT, PT>((int64_t)blockIdx.x, part, nblocks, npass, D, P, scale, ll, gw, gb,
                            ll_offset, (int)threadIdx.x, sm);
}

}  // namespace pa
