// chain_linalg.hip -- per-chain dense matrix x vector products for the dense ("full_mass")
// mass matrix of HMC/NUTS.
//
// Reference path replaced: BlockMassMatrix.kinetic_grad / scale / unscale with a dense block
// (pyro/infer/mcmc/adaptation.py:328-392: `inverse_mass_matrix.matmul(r_flat)` etc.), one chain
// per process.  Here every chain owns its matrix ([C, D, D], adapted per chain) and one launch
// does the product for all chains.  The work is a single streaming read of C*D*D elements (each
// used once): HBM-bound, one workgroup per chain, rows/columns coalesced along the contiguous
// matrix dimension, the vector staged in LDS.
#include "common.h"

namespace pa {

constexpr int CMV_THREADS = 256;

// TRANS = false: y[c, i] = sum_j M[c, i, j] x[c, j]     (M x)
// TRANS = true : y[c, j] = sum_i M[c, i, j] x[c, i]     (M^T x)
template <typename T, bool TRANS>
__global__ __launch_bounds__(CMV_THREADS) void chain_matvec_kernel(
    const T* __restrict__ M, int64_t m_stride_chain, const T* __restrict__ x, T* __restrict__ y,
    int D) {
  extern __shared__ unsigned char smem_raw[];
  T* xs = reinterpret_cast<T*>(smem_raw);            // [D]
  T* part = xs + D;                                  // TRANS: [4][64] partial column sums
  const int c = blockIdx.x;
  const T* Mc = M + (int64_t)c * m_stride_chain;
  const T* xc = x + (int64_t)c * D;
  T* yc = y + (int64_t)c * D;
  for (int j = threadIdx.x; j < D; j += CMV_THREADS) xs[j] = xc[j];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (!TRANS) {
    for (int i = wave; i < D; i += CMV_THREADS / 64) {       // wave-uniform
      const T* row = Mc + (int64_t)i * D;
      T acc = T(0);
      for (int j = lane; j < D; j += 64) acc += row[j] * xs[j];
      acc = wave_sum(acc);
      if (lane == 0) yc[i] = acc;
    }
  } else {
    for (int jb = 0; jb < D; jb += 64) {
      const int j = jb + lane;
      T acc = T(0);
      if (j < D)
        for (int i = wave; i < D; i += CMV_THREADS / 64) acc += Mc[(int64_t)i * D + j] * xs[i];
      part[wave * 64 + lane] = acc;
      __syncthreads();
      if (wave == 0 && j < D)
        yc[j] = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
      __syncthreads();
    }
  }
}

}  // namespace pa

extern "C" {

int pa_chain_matvec(int dtype, const void* M, int64_t m_stride_chain, const void* x, void* y,
                    int64_t C, int64_t D, int transpose, pa_stream_t stream) {
  PA_REQUIRE(dtype == PA_F32 || dtype == PA_F64, "chain_matvec: bad dtype %d", dtype);
  PA_REQUIRE(C >= 0 && D >= 0, "chain_matvec: negative size");
  if (C == 0 || D == 0) return PA_OK;
  PA_REQUIRE(M && x && y, "chain_matvec: NULL pointer");
  PA_REQUIRE(x != y, "chain_matvec: x and y must not alias");
  PA_REQUIRE(D <= 4096, "chain_matvec: D=%lld > 4096", (long long)D);
  PA_REQUIRE(m_stride_chain == 0 || m_stride_chain >= D * D, "chain_matvec: bad chain stride");
  PA_REQUIRE(C <= 0x7fffffffLL, "chain_matvec: too many chains");
  hipStream_t s = pa::as_stream(stream);
  const size_t es = dtype == PA_F32 ? 4 : 8;
  const size_t shmem = ((size_t)D + 256) * es;
#define PA_CMV(T, TR)                                                                            \
  hipLaunchKernelGGL((pa::chain_matvec_kernel<T, TR>), dim3((unsigned)C), dim3(pa::CMV_THREADS), \
                     shmem, s, (const T*)M, m_stride_chain, (const T*)x, (T*)y, (int)D)
  if (dtype == PA_F32) {
    if (transpose) PA_CMV(float, true); else PA_CMV(float, false);
  } else {
    if (transpose) PA_CMV(double, true); else PA_CMV(double, false);
  }
#undef PA_CMV
  return pa::check_launch("chain_matvec_kernel");
}

}  // extern "C"
