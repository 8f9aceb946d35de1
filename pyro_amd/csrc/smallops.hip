// smallops.hip -- the small element-wise torch operators of a captured step, run as ONE kernel.
//
// Reference path: everything a model / guide text computes with torch on small tensors -- constraint
// transforms of parameters (pyro/params/param_store.py:186-206), Dirichlet normalisations, clamps,
// scalings -- and the autograd duals of those operators: one ATen launch each.  In a captured step
// of config 4 (examples/lda.py) 60 of 153 graph nodes are such operators on <= 64 K elements; a node
// costs ~4.8 us of dependent dispatch whatever it computes (profiles/r03_cfg4_operator_attribution.txt).
//
// Here the host side (pyro_amd/ops/smallops.py, a TorchDispatchMode active while a step is being
// captured) does not launch those operators: it allocates the output and RECORDS an instruction
// {opcode, destination, <= 3 sources with broadcast strides over a <= 4-d frame, two immediates}.
// The pending program is emitted as one launch of smallops_kernel -- one workgroup of 1024 threads
// that interprets the instructions in order, with a workgroup barrier in front of an instruction that
// touches memory an earlier one of the program wrote -- when anything else is about to run: any other
// torch operator (the mode flushes), any launch of this library (as_stream() flushes), a recorded
// chain phase (chain_open() flushes), or 28 instructions are pending.  The arithmetic is torch's:
// IEEE add / sub / mul / div, the same libm exp / log, clamp and where with torch's NaN behaviour, so
// a captured step computes what the eager step computes.
#include "chain.h"
#include "common.h"
#include "multisite_dev.h"

#include <cstring>

namespace pa {

struct SmallOpDev {
  uint32_t op, ndim, numel, barrier;
  uint32_t shape[4];
  int32_t sd[4], s0[4], s1[4], s2[4];     // element strides of dst / sources along the frame dims
  float imm, imm2;
  void* dst;
  const void* src0;
  const void* src1;
  const void* src2;
};
static_assert(sizeof(SmallOpDev) == 136, "layout shared with include/pyro_amd.h pa_smallop");

struct SmallProgram {
  SmallOpDev ops[PA_SMALLOPS_MAX];
  int n;
};

__device__ __forceinline__ float so_clamp(float v, float lo, float hi) {
  return v != v ? v : fminf(fmaxf(v, lo), hi);          // at::native clamp: NaN stays NaN
}

constexpr int SO_BATCH = 8;       // elements per thread whose loads are in flight together

__global__ __launch_bounds__(1024) void smallops_kernel(const SmallProgram prog) {
  const int n = kernarg_load<int>((uint32_t)offsetof(SmallProgram, n));
  for (int i = 0; i < n; ++i) {
    const SmallOpDev o = kernarg_load<SmallOpDev>((uint32_t)(i * sizeof(SmallOpDev)));
    if (o.barrier & 1u) __syncthreads();  // (global writes of this workgroup are visible to it afterwards)
    // `barrier` bit 1 (set by the host): every operand is either dense in the frame's row-major order
    // or a scalar -- offsets are e times 0 or 1, no index decode
    const bool linear = (o.barrier & 2u) != 0;
    const int64_t l0 = o.s0[3], l1 = o.s1[3], l2 = o.s2[3];       // (then: 0 / 1 per operand)
    const bool u8op = o.op == PA_SO_AND_U8;
    const bool f0 = o.src0 != nullptr && !u8op, f1 = o.src1 != nullptr && !u8op;      // (wave-uniform)
    const float* a = (const float*)o.src0;
    const float* b = (const float*)o.src1;
    const uint8_t* a8 = (const uint8_t*)o.src0;
    const uint8_t* b8 = (const uint8_t*)o.src1;
    const uint8_t* c8 = (const uint8_t*)o.src2;
    // A thread's elements are taken SO_BATCH at a time: all loads of a batch are requested before the
    // first store (one memory round trip per batch; element by element the loop ran at one round trip
    // per element, 8-16 us per instruction on an [8, 1024] operand).  An operator's destination may
    // alias a source only element for element (in place), which this order preserves.
    for (uint32_t base = 0; base < o.numel; base += SO_BATCH * 1024) {
      int64_t od[SO_BATCH];
      float av[SO_BATCH], bv[SO_BATCH];
      uint32_t cv[SO_BATCH];
      bool ok[SO_BATCH];
#pragma unroll
      for (int j = 0; j < SO_BATCH; ++j) {
        const uint32_t e0 = base + j * 1024 + threadIdx.x;
        ok[j] = e0 < o.numel;
        const uint32_t e = ok[j] ? e0 : 0;                      // (clamped: unconditional loads)
        int64_t o0 = e * l0, o1 = e * l1, o2 = e * l2;
        od[j] = e;
        if (!linear) {
          // coordinates of element e in the frame (row-major), offsets through the strides
          uint32_t rem = e;
          od[j] = o0 = o1 = o2 = 0;
#pragma unroll
          for (int d = 3; d >= 0; --d) {
            if (d < (int)o.ndim) {
              const uint32_t c = rem % o.shape[d];
              rem /= o.shape[d];
              od[j] += (int64_t)c * o.sd[d];
              o0 += (int64_t)c * o.s0[d];
              o1 += (int64_t)c * o.s1[d];
              o2 += (int64_t)c * o.s2[d];
            }
          }
        }
        av[j] = f0 ? a[o0] : (u8op ? (float)a8[o0] : 0.0f);
        bv[j] = f1 ? b[o1] : (u8op ? (float)b8[o1] : 0.0f);
        cv[j] = c8 != nullptr ? c8[o2] : 0u;
      }
#pragma unroll
      for (int j = 0; j < SO_BATCH; ++j) {
        if (!ok[j]) continue;
        const float x = av[j], y = bv[j];
        float r = 0.0f;
        uint32_t q = 0u;
        switch (o.op) {
          case PA_SO_ADD: r = x + y; break;
          case PA_SO_SUB: r = x - y; break;
          case PA_SO_MUL: r = x * y; break;
          case PA_SO_DIV: r = x / y; break;
          case PA_SO_ADD_IMM: r = x + o.imm; break;
          case PA_SO_MUL_IMM: r = x * o.imm; break;
          case PA_SO_DIV_IMM: r = x * (1.0f / o.imm); break;   // ATen: a * (1 / scalar), reciprocal in f32
          case PA_SO_RSUB_IMM: r = o.imm - x; break;
          case PA_SO_RDIV_IMM: r = o.imm / x; break;
          case PA_SO_NEG: r = -x; break;
          case PA_SO_EXP: r = expf(x); break;
          case PA_SO_LOG: r = logf(x); break;
          case PA_SO_RECIP: r = 1.0f / x; break;
          case PA_SO_SQRT: r = sqrtf(x); break;
          case PA_SO_CLAMP: r = so_clamp(x, o.imm, o.imm2); break;
          case PA_SO_COPY: r = x; break;
          case PA_SO_FILL: r = o.imm; break;
          case PA_SO_WHERE: r = cv[j] ? x : y; break;
          case PA_SO_GE_IMM: q = x >= o.imm; break;
          case PA_SO_LE_IMM: q = x <= o.imm; break;
          case PA_SO_GT_IMM: q = x > o.imm; break;
          case PA_SO_LT_IMM: q = x < o.imm; break;
          case PA_SO_AND_U8: q = (x != 0.0f) && (y != 0.0f); break;
          default: break;
        }
        if (o.op >= PA_SO_GE_IMM) ((uint8_t*)o.dst)[od[j]] = (uint8_t)q;
        else ((float*)o.dst)[od[j]] = r;
      }
    }
  }
}

struct SmallState {
  bool on = false;
  hipStream_t stream = nullptr;
  SmallProgram prog;
  int launches = 0, recorded = 0;
};
static SmallState g_small;      // process-global: backward operators are recorded from autograd's thread

int smallops_flush_pending() {
  SmallState& s = g_small;
  if (!s.on || s.prog.n == 0) return PA_OK;
  const int n = s.prog.n;
  s.prog.n = 0;                 // (before the launch: nothing below may recurse into a flush)
  SmallProgram p = s.prog;
  p.n = n;
  hipLaunchKernelGGL(smallops_kernel, dim3(1), dim3(1024), 0, s.stream, p);
  s.launches += 1;
  return check_launch("smallops_kernel");
}

}  // namespace pa

extern "C" {

int pa_smallops_begin(pa_stream_t stream) {
  PA_REQUIRE(!pa::g_small.on, "pa_smallops_begin: already recording");
  pa::g_small.on = true;
  pa::g_small.stream = (hipStream_t)stream;
  pa::g_small.prog.n = 0;
  pa::g_small.launches = pa::g_small.recorded = 0;
  return PA_OK;
}

int pa_smallops_record(const pa_smallop* op) {
  pa::SmallState& s = pa::g_small;
  PA_REQUIRE(s.on, "pa_smallops_record: not recording");
  PA_REQUIRE(op && op->op >= 1 && op->op < PA_SO_COUNT && op->ndim <= 4 && op->dst, "pa_smallops_record: bad instruction");
  // a recorded chain phase may write what this instruction reads: it goes first
  int rc = pa::chain_flush_pending();
  if (rc != PA_OK) return rc;
  if (s.prog.n == PA_SMALLOPS_MAX) {
    rc = pa::smallops_flush_pending();
    if (rc != PA_OK) return rc;
  }
  static_assert(sizeof(pa_smallop) == sizeof(pa::SmallOpDev), "pa_smallop layout");
  std::memcpy(&s.prog.ops[s.prog.n], op, sizeof(pa::SmallOpDev));
  if (s.prog.n == 0) s.prog.ops[0].barrier &= ~1u;
  s.prog.n += 1;
  s.recorded += 1;
  return PA_OK;
}

int pa_smallops_flush(void) { return pa::smallops_flush_pending(); }

int pa_smallops_end(int* launches, int* recorded) {
  int rc = pa::smallops_flush_pending();
  if (launches) *launches = pa::g_small.launches;
  if (recorded) *recorded = pa::g_small.recorded;
  pa::g_small.on = false;
  return rc;
}

}  // extern "C"
