// Probe (developer tool): issue cost of VALU / transcendental / packed instructions next to
// v_mfma_f32_32x32x16_bf16 on gfx950, as a function of the fillers per MFMA and of the waves per SIMD.
// Every loop iteration issues 8 MFMAs (two independent accumulators, or ONE dependent chain) and
// after each MFMA `K` filler instructions of one kind on 16 independent registers.
// Output: cycles per MFMA group at the measured clock (s_memtime is not used: wall time x 2.4 GHz
// would assume a clock, so the table prints ns per group and the pure-MFMA line calibrates it).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MBF(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ab), "v"(bb))
enum { F_FMA = 0, F_ADD, F_PKFMA, F_PKADD, F_EXP, F_CVT, F_AND, F_BFI, F_DOT2C, F_FMAMIX, F_EXPMOD, F_ADDABS, F_SUB, F_LSHL, F_NONE };

template <int KIND>
__device__ __forceinline__ void filler(float& x, f32x2& p, float a, float b) {
  if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
  if (KIND == F_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  if (KIND == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
  if (KIND == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(p));
  if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  if (KIND == F_AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x));
  if (KIND == F_DOT2C) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  if (KIND == F_FMAMIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(a), "v"(b));
  if (KIND == F_EXPMOD) asm volatile("v_exp_f32_e64 %0, -|%0|" : "+v"(x));
  if (KIND == F_ADDABS) asm volatile("v_add_f32_e64 %0, |%1|, %0" : "+v"(x) : "v"(a));
  if (KIND == F_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  if (KIND == F_LSHL) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x));
  if (KIND == F_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(a), "v"(b));
}

template <int KIND, int K, int NMF, bool CHAIN>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x16 acc0 = {0}, acc1 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[16];
  f32x2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = a + i; p[i] = f32x2{a + i, a - i}; }
  s16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8}, bb = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (NMF) { if (CHAIN || (m & 1) == 0) MBF(acc0); else MBF(acc1); }
#pragma unroll
      for (int j = 0; j < K; ++j) filler<KIND>(v[(m * K + j) & 15], p[(m * K + j) & 15], a, b);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + p[i].x + p[i].y + acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int K, int NMF, bool CHAIN>
static float run(int blocks) {
  static float* out = nullptr;
  if (!out) (void)hipMalloc(&out, (size_t)1024 * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<KIND, K, NMF, CHAIN><<<blocks, 256>>>(out, 50);
  (void)hipEventRecord(e0);
  probe<KIND, K, NMF, CHAIN><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters / 8;   // ns per (MFMA + K fillers) group, per wave
}

template <int KIND>
static void table(const char* name) {
  printf("%-10s fillers/MFMA:      0      2      4      5      6      8     12   | no MFMA, 8 fillers | chain K=4\n", name);
  for (int wps : {1, 2, 3, 4}) {
    const int blocks = 256 * wps;
    printf("  waves/SIMD=%d     %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f   | %6.1f             | %6.1f   (ns per group per wave)\n", wps,
           run<KIND, 0, 1, false>(blocks), run<KIND, 2, 1, false>(blocks), run<KIND, 4, 1, false>(blocks),
           run<KIND, 5, 1, false>(blocks), run<KIND, 6, 1, false>(blocks), run<KIND, 8, 1, false>(blocks),
           run<KIND, 12, 1, false>(blocks), run<KIND, 8, 0, false>(blocks), run<KIND, 4, 1, true>(blocks));
  }
}

int main(int argc, char** argv) {
  if (argc > 1) {
    table<F_DOT2C>("v_dot2c_bf16");
    table<F_FMAMIX>("v_fma_mix");
    table<F_EXPMOD>("v_exp -|x|");
    table<F_ADDABS>("v_add |x|");
    table<F_SUB>("v_sub");
    table<F_LSHL>("v_lshl");
    return 0;
  }
  table<F_FMA>("v_fma");
  table<F_ADD>("v_add");
  table<F_PKFMA>("v_pk_fma");
  table<F_PKADD>("v_pk_add");
  table<F_EXP>("v_exp");
  table<F_CVT>("v_cvt_pk");
  table<F_AND>("v_and");
  table<F_BFI>("v_bfi");
  return 0;
}
