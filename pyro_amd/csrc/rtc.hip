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

#include <unistd.h>

#include <string>
#include <vector>

namespace pa {

struct RtcBlock { const void* table[PA_RTC_MAX_POINTERS]; void* params[1]; };
struct RtcScope { std::vector<RtcBlock*> blocks; };
static thread_local RtcScope* t_rtc_scope = nullptr;

static int rtc_fail(const char* what, hiprtcResult r) {
  return fail(PA_ERR_LAUNCH, "%s: %s", what, hiprtcGetErrorString(r));
}

}  // namespace pa

extern "C" {

static int rtc_load(const std::string& code, const char* kernel_name, void** function_out) {
  hipModule_t mod = nullptr;
  hipError_t e = hipModuleLoadData(&mod, code.data());
  if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "rtc_compile: hipModuleLoadData: %s", hipGetErrorString(e));
  hipFunction_t fn = nullptr;
  e = hipModuleGetFunction(&fn, mod, kernel_name);
  if (e != hipSuccess) {
    (void)hipModuleUnload(mod);
    return pa::fail(PA_ERR_LAUNCH, "rtc_compile: no kernel '%s': %s", kernel_name, hipGetErrorString(e));
  }
  *function_out = (void*)fn;        // (the module lives as long as the process: kernels are cached by source)
  return PA_OK;
}

int pa_rtc_version(int* hiprtc_major, int* hiprtc_minor, int* runtime_version) {
  PA_REQUIRE(hiprtc_major && hiprtc_minor && runtime_version, "rtc_version: NULL pointer");
  hiprtcResult r = hiprtcVersion(hiprtc_major, hiprtc_minor);
  if (r != HIPRTC_SUCCESS) return pa::rtc_fail("hiprtcVersion", r);
  if (hipRuntimeGetVersion(runtime_version) != hipSuccess) *runtime_version = 0;
  return PA_OK;
}

int pa_rtc_compile_cached(const char* source, const char* kernel_name, const char* cache_file,
                          void** function_out, int* compiled_out) {
  PA_REQUIRE(source && kernel_name && function_out, "rtc_compile: NULL pointer");
  if (compiled_out) *compiled_out = 0;
  if (cache_file != nullptr) {
    // a code object an earlier process compiled from this very source (the caller's file name is a digest of
    // source + compiler version + options): loaded as is -- no hiprtc call in this process
    FILE* f = fopen(cache_file, "rb");
    if (f != nullptr) {
      std::string code;
      char buf[1 << 16];
      size_t n;
      while ((n = fread(buf, 1, sizeof(buf), f)) > 0) code.append(buf, n);
      fclose(f);
      // (hiprtc hands out an ELF code object; a clang offload bundle is what other versions wrote)
      if (code.size() > 64 && (code.compare(1, 3, "ELF") == 0 || code.compare(0, 8, "__CLANG_") == 0)) {
        void* fn = nullptr;
        if (rtc_load(code, kernel_name, &fn) == PA_OK) {
          *function_out = fn;
          return PA_OK;
        }
      }
      // (a truncated or foreign file: compile, and replace it below)
    }
  }
  hiprtcProgram prog = nullptr;
  hiprtcResult r = hiprtcCreateProgram(&prog, source, "pyro_amd_fused.hip", 0, nullptr, nullptr);
  if (r != HIPRTC_SUCCESS) return pa::rtc_fail("hiprtcCreateProgram", r);
  // -ffp-contract=off: a*b+c stays a multiply and an add, as the two ATen kernels it replaces compute it
  // (pyro_amd/ops/fuser.py::RTC_OPTIONS names the same list in the cache key)
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
  if (compiled_out) *compiled_out = 1;
  if (cache_file != nullptr) {
    // written under a private name and renamed: a reader never sees a partial file, concurrent writers of the
    // same digest write the same bytes (a cache that cannot be written is not an error)
    std::string tmp = std::string(cache_file) + ".tmp." + std::to_string((long long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (f != nullptr) {
      const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
      if (fclose(f) == 0 && ok) {
        if (rename(tmp.c_str(), cache_file) != 0) (void)remove(tmp.c_str());
      } else {
        (void)remove(tmp.c_str());
      }
    }
  }
  return rtc_load(code, kernel_name, function_out);
}

int pa_rtc_compile(const char* source, const char* kernel_name, void** function_out) {
  return pa_rtc_compile_cached(source, kernel_name, nullptr, function_out, nullptr);
}

int pa_rtc_launch(void* function, uint32_t grid, uint32_t block, const void* const* pointers, int n_pointers,
                  pa_stream_t stream) {
  PA_REQUIRE(function && grid > 0 && block > 0 && block <= 1024, "rtc_launch: bad launch geometry");
  PA_REQUIRE(n_pointers >= 0 && n_pointers <= PA_RTC_MAX_POINTERS && (pointers || n_pointers == 0),
             "rtc_launch: %d pointers (at most %d)", n_pointers, PA_RTC_MAX_POINTERS);
  // the kernel's ONE parameter is a struct of at most PA_RTC_MAX_POINTERS pointers, by value.  While the stream is
  // being captured the runtime reads the parameter block when the capture ENDS (a block on this stack frame
  // crashed hipStreamEndCapture): such launches get a block that lives as long as the process (3 KB per
  // captured launch, owned by the host's capture scope)
  using pa::RtcBlock;
  RtcBlock local{};
  RtcBlock* blk = &local;
  hipStream_t s = pa::as_stream(stream);
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
    blk = new RtcBlock{};
    // owned by the capture scope the host opened (pa_rtc_blocks_begin): freed with the captured graph; with
    // no scope open it lives as long as the process (the behaviour before ABI 7)
    if (pa::t_rtc_scope != nullptr) pa::t_rtc_scope->blocks.push_back(blk);
  }
  for (int i = 0; i < n_pointers; ++i) blk->table[i] = pointers[i];
  blk->params[0] = (void*)blk->table;
  hipError_t e = hipModuleLaunchKernel((hipFunction_t)function, grid, 1, 1, block, 1, 1, 0, s, blk->params, nullptr);
  if (e != hipSuccess) return pa::fail(PA_ERR_LAUNCH, "rtc_launch: %s", hipGetErrorString(e));
  return pa::check_launch("rtc_kernel");
}

void* pa_rtc_blocks_begin(void) {
  pa::RtcScope* sc = new pa::RtcScope{};
  pa::t_rtc_scope = sc;
  return (void*)sc;
}

int pa_rtc_blocks_end(void* scope, int64_t* n_blocks_out) {
  PA_REQUIRE(scope != nullptr, "rtc_blocks_end: NULL scope");
  if (pa::t_rtc_scope == (pa::RtcScope*)scope) pa::t_rtc_scope = nullptr;
  if (n_blocks_out) *n_blocks_out = (int64_t)((pa::RtcScope*)scope)->blocks.size();
  return PA_OK;
}

int pa_rtc_blocks_free(void* scope) {
  if (scope == nullptr) return PA_OK;
  pa::RtcScope* sc = (pa::RtcScope*)scope;
  if (pa::t_rtc_scope == sc) pa::t_rtc_scope = nullptr;
  for (pa::RtcBlock* b : sc->blocks) delete b;
  delete sc;
  return PA_OK;
}

}  // extern "C"
