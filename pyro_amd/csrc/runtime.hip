// runtime.hip -- host-side plumbing of the C-ABI: error strings, device queries.
#include "common.h"

#include <string.h>

namespace pa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PA_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return PA_OK;
}

int cu_count() {
  static int cached = 0;
  if (cached > 0) return cached;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
    return 256;
  cached = n;
  return n;
}

}  // namespace pa

extern "C" {

int pa_abi_version(void) { return PA_ABI_VERSION; }
const char* pa_last_error(void) { return pa::g_err; }
int pa_device_cu_count(void) { return pa::cu_count(); }

}  // extern "C"
