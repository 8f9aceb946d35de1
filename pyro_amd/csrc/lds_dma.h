// lds_dma.h -- HBM/L2 -> LDS copies without registers (global_load_lds_*) and counted waits, shared by
// the kernels that stream operand images (glm_planes.h, glm_planes16.h, bow.hip).
#pragma once
#include "common.h"

namespace pa {

// LDS-DMA as inline asm: 64 lanes x 16 B (or 4 B) from per-lane global addresses to the wave-uniform
// LDS byte address `lds_dst` + lane * 16 (4).  Written as asm on purpose: hipcc orders LDS reads
// behind an LDS-DMA it can see with s_waitcnt vmcnt(0) (it cannot tell which bytes the DMA writes),
// which would drain the prefetch ring in every iteration; the loop below waits with counted
// s_waitcnt vmcnt(N) + s_barrier instead.  M0 (the LDS base of the DMA) is compiler-reserved:
// saved, set and restored inside the one statement.
// The GLM image is read exactly once per launch (198 MB at the headline size): the loads carry the
// non-temporal hint so that the stream does not push everything else -- the parameters, the code and
// operands of the small kernels around this one -- out of the L2s (PA_GLMP_NT=0: plain loads).
#ifndef PA_GLMP_NT
#define PA_GLMP_NT 1
#endif
#if PA_GLMP_NT
#define PA_GLMP_NT_STR " nt"
#else
#define PA_GLMP_NT_STR ""
#endif
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" PA_GLMP_NT_STR "\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// the same without the hint: operands that many workgroups re-read from the L2 (bow.hip's W planes)
__device__ __forceinline__ void dma16_cached(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

}  // namespace pa
