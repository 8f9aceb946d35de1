// Probe: can VALU work overlap (a) f32-input MFMA, (b) bf16 MFMA on gfx950?
// Inline asm fixes the instruction mix and order exactly.  Per loop iteration:
//   f32 : 8 x v_mfma_f32_32x32x2_f32   (8 x 64 = 512 issue cycles on one SIMD)
//   bf16: 16 x v_mfma_f32_32x32x16_bf16 (16 x 32 = 512)
//   valu: 128 x v_fma_f32 on 16 independent chains (128 x 4 = 512)
//   exp : 32 x v_exp_f32 (quarter rate: 32 x 16 = 512)
// "+" modes interleave them 1 MFMA : k VALU.  One workgroup of 256 threads per CU (1 wave/SIMD)
// or two (2 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define MF32(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MBF(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ab), "v"(bb))
#define VF(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a))
#define VE(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define V16() VF(v0); VF(v1); VF(v2); VF(v3); VF(v4); VF(v5); VF(v6); VF(v7); VF(v8); VF(v9); VF(v10); VF(v11); VF(v12); VF(v13); VF(v14); VF(v15)
#define V8a() VF(v0); VF(v1); VF(v2); VF(v3); VF(v4); VF(v5); VF(v6); VF(v7)
#define V8b() VF(v8); VF(v9); VF(v10); VF(v11); VF(v12); VF(v13); VF(v14); VF(v15)
#define E4a() VE(v0); VE(v1); VE(v2); VE(v3)
#define E4b() VE(v4); VE(v5); VE(v6); VE(v7)

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x16 acc0 = {0}, acc1 = {0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
  float v8 = a, v9 = a + 1, v10 = a + 2, v11 = a + 3, v12 = a + 4, v13 = a + 5, v14 = a + 6, v15 = a + 7;
  s16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8}, bb = {1, 1, 1, 1, 1, 1, 1, 1};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { for (int k = 0; k < 4; ++k) { MF32(acc0); MF32(acc1); } }
    if (MODE == 1) { for (int k = 0; k < 8; ++k) { V16(); } }
    if (MODE == 2) { for (int k = 0; k < 4; ++k) { MF32(acc0); V16(); MF32(acc1); V16(); } }
    if (MODE == 3) { for (int k = 0; k < 8; ++k) { MBF(acc0); MBF(acc1); } }
    if (MODE == 4) { for (int k = 0; k < 8; ++k) { MBF(acc0); V8a(); MBF(acc1); V8b(); } }
    if (MODE == 5) { for (int k = 0; k < 4; ++k) { E4a(); E4b(); } }
    if (MODE == 6) { for (int k = 0; k < 4; ++k) { MF32(acc0); E4a(); MF32(acc1); E4b(); } }
    if (MODE == 7) { for (int k = 0; k < 8; ++k) { MBF(acc0); E4a(); MBF(acc1); } }
  }
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + v8 + v9 + v10 + v11 + v12 + v13 + v14 + v15;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int blocks) {
  float* out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(out, 100);
  (void)hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-36s waves/SIMD=%d : %8.1f ns/iter\n", name, blocks / 256, ms * 1e6 / iters);
  (void)hipFree(out);
}

int main() {
  for (int blocks : {256, 512}) {
    run<0>("8 f32 mfma", blocks);
    run<1>("128 v_fma", blocks);
    run<2>("8 f32 mfma + 128 v_fma interleaved", blocks);
    run<3>("16 bf16 mfma", blocks);
    run<4>("16 bf16 mfma + 128 v_fma interleaved", blocks);
    run<5>("32 v_exp", blocks);
    run<6>("8 f32 mfma + 32 v_exp interleaved", blocks);
    run<7>("16 bf16 mfma + 32 v_exp interleaved", blocks);
  }
  return 0;
}
