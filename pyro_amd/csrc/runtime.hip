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

// one-shot event bracket requested through pa_profile_bracket_next()
static thread_local int g_br_tag = 0;
static thread_local hipEvent_t g_br_start = nullptr, g_br_stop = nullptr;

bool take_bracket(int tag, hipEvent_t* start, hipEvent_t* stop) {
  if (g_br_tag != tag || g_br_start == nullptr) return false;
  *start = g_br_start;
  *stop = g_br_stop;
  g_br_tag = 0;
  g_br_start = g_br_stop = nullptr;
  return true;
}

}  // namespace pa

extern "C" {

int pa_profile_bracket_next(int kernel_tag, void* ev_start, void* ev_stop) {
  PA_REQUIRE(kernel_tag >= 1 && kernel_tag <= 4, "profile_bracket_next: unknown kernel tag %d",
             kernel_tag);
  PA_REQUIRE(ev_start && ev_stop, "profile_bracket_next: NULL event");
  pa::g_br_tag = kernel_tag;
  pa::g_br_start = (hipEvent_t)ev_start;
  pa::g_br_stop = (hipEvent_t)ev_stop;
  return PA_OK;
}

int pa_abi_version(void) { return PA_ABI_VERSION; }
const char* pa_last_error(void) { return pa::g_err; }
int pa_device_cu_count(void) { return pa::cu_count(); }

}  // extern "C"
