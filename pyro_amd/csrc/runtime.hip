// runtime.hip -- host-side plumbing of the C-ABI: error strings, device queries.
#include "common.h"

#include <stdlib.h>
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

// the step gate: see include/pyro_amd.h.  (entry points are serialised by the host language)
static int64_t* g_gate = nullptr;
static int64_t g_gate_launches = 0, g_gate_aware = 0;
// a LATE gate (pa_gate_defer): registered at the start of a capture, emitted in front of the step's chained
// tail; what is launched before it runs unconditionally (the forward pass of a step enqueued ahead of time)
struct GateDeferred {
  bool pending = false, emitted = false;
  const int64_t* go = nullptr;
  int64_t* gate = nullptr;
  int64_t* ack = nullptr;
  unsigned long long ticks = 0;
  int64_t pre = 0, pre_other = 0;        // launches before the gate node / of them not the plane-image GLM
};
static GateDeferred g_gate_late;
const int64_t* gate_word() { return g_gate == nullptr ? nullptr : g_gate + 1; }
void gate_aware_launch() {
  if (!g_gate_late.pending) g_gate_aware += 1;
}

int check_launch(const char* what) {
  static const bool trace = getenv("PYRO_AMD_TRACE_LAUNCHES") != nullptr;   // developer aid: every launch by name
  if (trace) fprintf(stderr, "pyro_amd launch: %s%s\n", what, g_gate_late.pending ? "  (in front of a late gate)" : "");
  if (g_gate_late.pending) {
    g_gate_late.pre += 1;
    if (strncmp(what, "glm_planes_kernel", 17) != 0 || strchr(what, '<') != nullptr) g_gate_late.pre_other += 1;
  } else {
    g_gate_launches += 1;
  }
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

__global__ void gate_kernel(const int64_t* go, int64_t* gate, int64_t* ack, unsigned long long ticks) {
  const int64_t n = gate[0] + 1;
  const unsigned long long t0 = wall_clock64();
  int aborted = 0;
  for (unsigned iter = 0;; ++iter) {
    const int64_t v = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v >= n) break;
    // (the iteration cap bounds the wait even if the clock should ever stand still)
    if (v == -n || wall_clock64() - t0 > ticks || iter > (1u << 22)) {
      aborted = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  gate[1] = aborted;
  if (!aborted) {
    gate[0] = n;
  } else {
    __hip_atomic_store(ack, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// called by the chained tail's launcher right before it emits its kernel
void gate_emit_deferred(hipStream_t s) {
  GateDeferred& d = g_gate_late;
  if (!d.pending) return;
  d.pending = false;
  d.emitted = true;
  g_gate = d.gate;                       // the scope opens here: everything from now on polls gate[1]
  hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, s, d.go, d.gate, d.ack, d.ticks);
  g_gate_aware += 1;
  (void)check_launch("gate_kernel");
}

}  // namespace pa

extern "C" {

int pa_gate_defer(const int64_t* go, int64_t* gate, int64_t* ack, int64_t timeout_us) {
  PA_REQUIRE(go && gate && ack, "pa_gate_defer: NULL pointer");
  PA_REQUIRE(timeout_us > 0 && timeout_us <= 100000, "pa_gate_defer: timeout_us=%lld outside (0, 1e5]",
             (long long)timeout_us);
  pa::g_gate = nullptr;
  pa::g_gate_launches = pa::g_gate_aware = 0;
  pa::g_gate_late = pa::GateDeferred{};
  pa::g_gate_late.pending = true;
  pa::g_gate_late.go = go;
  pa::g_gate_late.gate = gate;
  pa::g_gate_late.ack = ack;
  pa::g_gate_late.ticks = (unsigned long long)timeout_us * 100ull;
  return PA_OK;
}

int pa_gate_defer_stats(int64_t* pre, int64_t* pre_other, int* emitted) {
  if (pre) *pre = pa::g_gate_late.pre;
  if (pre_other) *pre_other = pa::g_gate_late.pre_other;
  if (emitted) *emitted = pa::g_gate_late.emitted ? 1 : 0;
  return PA_OK;
}

int pa_gate(const int64_t* go, int64_t* gate, int64_t* ack, int64_t timeout_us, pa_stream_t stream) {
  PA_REQUIRE(go && gate && ack, "pa_gate: NULL pointer");
  PA_REQUIRE(timeout_us > 0 && timeout_us <= 100000, "pa_gate: timeout_us=%lld outside (0, 1e5]",
             (long long)timeout_us);
  // wall_clock64 counts at 100 MHz
  hipLaunchKernelGGL(pa::gate_kernel, dim3(1), dim3(1), 0, pa::as_stream(stream), go, gate, ack,
                     (unsigned long long)timeout_us * 100ull);
  pa::g_gate_aware += 1;
  return pa::check_launch("gate_kernel");
}

int pa_gate_scope(int64_t* gate) {
  pa::g_gate = gate;
  pa::g_gate_launches = pa::g_gate_aware = 0;
  pa::g_gate_late.pending = false;       // (a deferred gate that was never emitted is dropped; its counts stay readable)
  return PA_OK;
}

int pa_gate_stats(int64_t* launches, int64_t* aware) {
  if (launches) *launches = pa::g_gate_launches;
  if (aware) *aware = pa::g_gate_aware;
  return PA_OK;
}

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
