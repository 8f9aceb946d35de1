// rtc.hip -- run-time compiled element-wise kernels: the native half of pyro_amd/ops/fuser.py.
//
// Reference path replaced: the long tail of small ATen kernels around the fused sites of a step --
// constraint transforms of parameters, normalisations in a model's / guide's own torch text, the
// autograd duals of all of them (pyro/infer/traceenum_elbo.py:112-214 + examples/lda.py:78-122 run
// ~110 such operators per step; each is a graph node that costs its dispatch whatever it computes).
// The host records runs of such operators and emits ONE HIP kernel per run with the intermediates in
// registers; this file compiles that source for the running device (hiprtc), loads it and launches
// it on the caller's stream.  Nothing here is a fallback: a source that fails to compile is an error.
#include "common.h"

#include <hip/hiprtc.h>

#include <string>

namespace pa {

static int rtc_fail(const char* what, hiprtcResult r) {
  return fail(PA_ERR_LAUNCH, "%s: %s", what, hiprtcGetErrorString(r));
}

}  // namespace pa

extern "C" {

int pa_rtc_compile(const char* source, const char* kernel_name, void** function_out) {
  PA_REQUIRE(source && kernel_name && function_out, "rtc_compile: NULL pointer");
  hiprtcProgram prog = nullptr;
  hiprtcResult r = hiprtcCreateProgram(&prog, source, "pyro_amd_fused.hip", 0, nullptr, nullptr);
  if (r != HIPRTC_SUCCESS) return pa::rtc_fail("hiprtcCreateProgram", r);
  // -ffp-contract=off: a*b+c stays a multiply and an add, as the two ATen kernels it replaces compute it
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
  r = hiprtcCompileProgram(prog, 4, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    (void)hiprtcGetProgramLogSize(prog, &n);
    std::string log(n + 1, '\0');
    if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
    (void)hiprtcDestroyProgram(&prog);
    return pa::fail(PA_ERR_LAUNCH, "rtc_compile(%s): %s\n%.400s", kernel_name, hiprtcGetErrorString(r),
                    log.c_str());
  }
  size_t size = 0;
  r = hiprtcGetCodeSize(prog, &size);
  if (r != HIPRTC_SUCCESS) { (void)hiprtcDestroyProgram(&prog); return pa::rtc_fail("hiprtcGetCodeSize", r); }
  std::string code(size, '\0');
  r = hiprtcGetCode(prog, &code[0]);
  (void)hiprtcDestroyProgram(&prog);
  if (r != HIPRTC_SUCCESS) return pa::rtc_fail("hiprtcGetCode", r);
  hipModule_t mod = nullptr;
  hipError_t e = hipModuleLoadData(&mod, code.data());
  if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "rtc_compile: hipModuleLoadData: %s", hipGetErrorString(e));
  hipFunction_t fn = nullptr;
  e = hipModuleGetFunction(&fn, mod, kernel_name);
  if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "rtc_compile: no kernel '%s': %s", kernel_name, hipGetErrorString(e));
  *function_out = (void*)fn;        // (the module lives as long as the process: kernels are cached by source)
  return PA_OK;
}

int pa_rtc_launch(void* function, uint32_t grid, uint32_t block, const void* const* pointers, int n_pointers,
                  pa_stream_t stream) {
  PA_REQUIRE(function && grid > 0 && block > 0 && block <= 1024, "rtc_launch: bad launch geometry");
  PA_REQUIRE(n_pointers >= 0 && n_pointers <= PA_RTC_MAX_POINTERS && (pointers || n_pointers == 0),
             "rtc_launch: %d pointers (at most %d)", n_pointers, PA_RTC_MAX_POINTERS);
  // the kernel's ONE parameter is a struct of at most PA_RTC_MAX_POINTERS pointers, by value.  While the stream is
  // being captured the runtime reads the parameter block when the capture ENDS (a block on this stack frame
  // crashed hipStreamEndCapture): such launches get a block that lives as long as the process (3 KB per
  // captured launch)
  struct Block { const void* table[PA_RTC_MAX_POINTERS]; void* params[1]; };
  Block local{};
  Block* blk = &local;
  hipStream_t s = pa::as_stream(stream);
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) blk = new Block{};
  for (int i = 0; i < n_pointers; ++i) blk->table[i] = pointers[i];
  blk->params[0] = (void*)blk->table;
  hipError_t e = hipModuleLaunchKernel((hipFunction_t)function, grid, 1, 1, block, 1, 1, 0, s, blk->params, nullptr);
  if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "rtc_launch: %s", hipGetErrorString(e));
  return pa::check_launch("rtc_kernel");
}

}  // extern "C"
