// Probe (developer tool): lane/element mapping of ds_read_b64_tr_b16 on gfx950.
// LDS holds the u16 value i at byte 2*i; lane l passes the address 8*l (its own 4 consecutive
// u16: 4l..4l+3).  Prints, per lane, the 4 u16 it receives.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) v4i16* lptr;
  lptr p = (lptr)(uint32_t)(uintptr_t)(lds + 4 * threadIdx.x);
  v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
  unsigned short* d; (void)hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  return 0;
}
