// torch_ops.cpp -- TORCH_LIBRARY registration of the fused GLM site over the extern "C" launchers of
// libpyro_amd.so (include/pyro_amd.h): the ops appear in torch's dispatcher as pyro_amd::*, so that
// torch.jit.trace / torch.compile record them as graph nodes instead of losing the ctypes calls.
//
// Reference seam: pyro/ops/jit.py:104-109 (torch.jit.trace of a loss function over the unconstrained
// parameters) and pyro/infer/trace_elbo.py:162-257 (JitTrace_ELBO.differentiable_loss); the reference
// registers nothing because it has no native ops -- its traced graph is made of ATen nodes.  Here the
// observed GLM site (pyro/poutine/trace_struct.py:264-278 at the Bernoulli-logits site of SURVEY
// 8(d)'s model) is ONE node whose three outputs are the per-particle log-likelihood and its exact
// gradient factors; the autograd formula (registered from Python, pyro_amd/ops/torch_library.py) is a
// second node, pyro_amd::glm_chain.
//
// Host-only C++ (no kernels here): built by csrc/build.py into lib/libpyro_amd_torch.so and loaded
// with torch.ops.load_library.  Device memory comes from torch's allocator, the launches go to
// torch's current stream.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <optional>
#include <tuple>

#include "pyro_amd.h"

namespace {

void check(int rc, const char* who) {
  TORCH_CHECK(rc == PA_OK, "pyro_amd::", who, ": ", pa_last_error());
}

pa_stream_t current_stream() { return (pa_stream_t)c10::hip::getCurrentHIPStream().stream(); }

const float* f32_ptr(const std::optional<at::Tensor>& t) {
  return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr;
}

void require_f32_gpu(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(), "pyro_amd: ", name,
              " must be a contiguous float32 tensor on the GPU");
}

// X[N,D] -> the uint8 plane image of pa_glm_pack_planes
at::Tensor glm_pack_planes(const at::Tensor& X, int64_t format) {
  require_f32_gpu(X, "X");
  TORCH_CHECK(X.dim() == 2, "pyro_amd::glm_pack_planes: X must be [N, D]");
  const int64_t N = X.size(0), D = X.size(1);
  const size_t nbytes = pa_glm_planes_bytes((int)format, N, D);
  TORCH_CHECK(nbytes > 0 || N == 0, "pyro_amd::glm_pack_planes: no plane image for D = ", D);
  at::Tensor out = at::empty({(int64_t)(nbytes < 16 ? 16 : nbytes)}, X.options().dtype(at::kByte));
  check(pa_glm_pack_planes((int)format, X.data_ptr<float>(), N, D, out.data_ptr(), nbytes, current_stream()),
        "glm_pack_planes");
  return out;
}

// (ll, gw, gb, ws): the workspace is an OUTPUT so that the caller can keep it alive -- inside a chained
// tail (pa_chain_begin) the launcher only RECORDS the finalize phase, which reads the partial records
// in ws when the chain is flushed, after this function has returned
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> glm_bernoulli_planes(
    const at::Tensor& planes, const at::Tensor& y, const at::Tensor& w,
    const std::optional<at::Tensor>& b, double scale, int64_t N, int64_t D, int64_t format,
    const std::optional<at::Tensor>& moments) {
  require_f32_gpu(y, "y");
  require_f32_gpu(w, "w");
  TORCH_CHECK(planes.is_cuda() && planes.scalar_type() == at::kByte, "pyro_amd: planes must be a uint8 image");
  TORCH_CHECK(w.dim() == 2 && w.size(1) == D && y.dim() == 1 && y.size(0) == N,
              "pyro_amd::glm_bernoulli_planes: shapes y[N], w[P, D]");
  const int64_t P = w.size(0);
  if (b.has_value() && b->defined()) {
    require_f32_gpu(*b, "b");
    TORCH_CHECK(b->numel() == P, "pyro_amd::glm_bernoulli_planes: b[P]");
  }
  const size_t ws_bytes = pa_glm_bernoulli_planes_workspace(N, D, P);
  TORCH_CHECK(ws_bytes > 0, "pyro_amd::glm_bernoulli_planes: unsupported shape");
  at::Tensor ws = at::empty({(int64_t)ws_bytes}, w.options().dtype(at::kByte));
  at::Tensor ll = at::empty({P}, w.options()), gw = at::empty({P, D}, w.options()),
             gb = at::empty({P}, w.options());
  const double* mom = nullptr;          // pa_glm_label_moments of (X, y), float64[33] (or none)
  if (moments.has_value() && moments->defined()) {
    TORCH_CHECK(moments->is_cuda() && moments->scalar_type() == at::kDouble && moments->numel() == 33 &&
                    moments->is_contiguous(), "pyro_amd::glm_bernoulli_planes: moments = float64[33]");
    mom = moments->data_ptr<double>();
  }
  check(pa_glm_bernoulli_planes_fwd_bwd((int)format, planes.data_ptr(), y.data_ptr<float>(),
                                        w.data_ptr<float>(), f32_ptr(b), scale, N, D, P,
                                        ll.data_ptr<float>(), gw.data_ptr<float>(), gb.data_ptr<float>(),
                                        ws.data_ptr(), ws_bytes, mom, current_stream()),
        "glm_bernoulli_planes");
  return {ll, gw, gb, ws};
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> glm_bernoulli(
    const at::Tensor& X, const at::Tensor& y, const at::Tensor& w, const std::optional<at::Tensor>& b,
    const std::optional<at::Tensor>& mask, double scale) {
  require_f32_gpu(X, "X");
  require_f32_gpu(y, "y");
  require_f32_gpu(w, "w");
  TORCH_CHECK(X.dim() == 2 && w.dim() == 2 && w.size(1) == X.size(1) && y.dim() == 1 && y.size(0) == X.size(0),
              "pyro_amd::glm_bernoulli: shapes X[N, D], y[N], w[P, D]");
  const int64_t N = X.size(0), D = X.size(1), P = w.size(0);
  const uint8_t* m = nullptr;
  if (mask.has_value() && mask->defined()) {
    TORCH_CHECK(mask->is_cuda() && mask->is_contiguous() && mask->numel() == N &&
                    (mask->scalar_type() == at::kBool || mask->scalar_type() == at::kByte),
                "pyro_amd::glm_bernoulli: mask[N] bool");
    m = (const uint8_t*)mask->data_ptr();
  }
  if (b.has_value() && b->defined()) require_f32_gpu(*b, "b");
  const size_t ws_bytes = pa_glm_bernoulli_workspace(N, D, P);
  TORCH_CHECK(ws_bytes > 0, "pyro_amd::glm_bernoulli: unsupported shape");
  at::Tensor ws = at::empty({(int64_t)ws_bytes}, w.options().dtype(at::kByte));
  at::Tensor ll = at::empty({P}, w.options()), gw = at::empty({P, D}, w.options()),
             gb = at::empty({P}, w.options());
  check(pa_glm_bernoulli_fwd_bwd(X.data_ptr<float>(), y.data_ptr<float>(), w.data_ptr<float>(), f32_ptr(b),
                                 m, scale, N, D, P, ll.data_ptr<float>(), gw.data_ptr<float>(),
                                 gb.data_ptr<float>(), ws.data_ptr(), ws_bytes, current_stream()),
        "glm_bernoulli");
  return {ll, gw, gb, ws};
}

// (g[P] * gw[P, W], g[P] * gb[P]): the backward of the site
std::tuple<at::Tensor, at::Tensor> glm_chain(const at::Tensor& g, const at::Tensor& gw, const at::Tensor& gb) {
  require_f32_gpu(gw, "gw");
  require_f32_gpu(gb, "gb");
  at::Tensor gc = g.reshape({-1}).contiguous();
  const int64_t P = gc.numel();
  TORCH_CHECK(gc.scalar_type() == at::kFloat && gw.size(0) == P && gb.numel() == P, "pyro_amd::glm_chain: shapes");
  const int64_t W = P > 0 ? gw.numel() / P : 0;
  at::Tensor dw = at::empty_like(gw), db = at::empty_like(gb);
  check(pa_glm_chain(gc.data_ptr<float>(), gw.data_ptr<float>(), gb.data_ptr<float>(), P, W,
                     dw.data_ptr<float>(), db.data_ptr<float>(), current_stream()),
        "glm_chain");
  return {dw, db};
}

// pa_adam_step over the flat parameter buffer (pyro/optim/optim.py:117-155 + clipped_adam.py:52-100 +
// pyro/infer/util.py:85-91 zero_grads in one launch); every tensor argument is updated in place
void adam_step(at::Tensor param, at::Tensor grad, at::Tensor exp_avg, at::Tensor exp_avg_sq, at::Tensor step,
               double lr, double beta1, double beta2, double eps, double weight_decay, double clip_norm,
               double lrd, bool clipped, bool zero_grad) {
  TORCH_CHECK(param.is_cuda() && param.is_contiguous() && grad.is_contiguous() && exp_avg.is_contiguous() &&
                  exp_avg_sq.is_contiguous(), "pyro_amd::adam_step: contiguous GPU buffers");
  TORCH_CHECK(param.scalar_type() == at::kFloat || param.scalar_type() == at::kDouble,
              "pyro_amd::adam_step: float32 / float64");
  TORCH_CHECK(grad.scalar_type() == param.scalar_type() && exp_avg.scalar_type() == param.scalar_type() &&
                  exp_avg_sq.scalar_type() == param.scalar_type() && grad.numel() == param.numel() &&
                  exp_avg.numel() == param.numel() && exp_avg_sq.numel() == param.numel(),
              "pyro_amd::adam_step: param / grad / moments of one dtype and size");
  TORCH_CHECK(step.is_cuda() && step.scalar_type() == at::kLong && step.numel() == 2,
              "pyro_amd::adam_step: step = int64[2] {step counter, ticket}");
  check(pa_adam_step(param.scalar_type() == at::kFloat ? PA_F32 : PA_F64, param.data_ptr(), grad.data_ptr(),
                     exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param.numel(), lr, beta1, beta2, eps,
                     weight_decay, clip_norm, lrd, clipped ? 1 : 0, step.data_ptr<int64_t>(), zero_grad ? 1 : 0,
                     current_stream()),
        "adam_step");
}

}  // namespace

TORCH_LIBRARY(pyro_amd, m) {
  m.def("glm_pack_planes(Tensor X, int format) -> Tensor");
  m.def("glm_bernoulli_planes(Tensor planes, Tensor y, Tensor w, Tensor? b, float scale, int N, int D, "
        "int format, Tensor? moments=None) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("glm_bernoulli(Tensor X, Tensor y, Tensor w, Tensor? b, Tensor? mask, float scale) -> "
        "(Tensor, Tensor, Tensor, Tensor)");
  m.def("adam_step(Tensor(a!) param, Tensor(b!) grad, Tensor(c!) exp_avg, Tensor(d!) exp_avg_sq, "
        "Tensor(e!) step, float lr, float beta1, float beta2, float eps, float weight_decay, float clip_norm, "
        "float lrd, bool clipped, bool zero_grad) -> ()");
  m.def("glm_chain(Tensor g, Tensor gw, Tensor gb) -> (Tensor, Tensor)");
}

// HIP tensors carry the CUDA dispatch key in a ROCm build of torch
TORCH_LIBRARY_IMPL(pyro_amd, CUDA, m) {
  m.impl("glm_pack_planes", &glm_pack_planes);
  m.impl("glm_bernoulli_planes", &glm_bernoulli_planes);
  m.impl("glm_bernoulli", &glm_bernoulli);
  m.impl("glm_chain", &glm_chain);
  m.impl("adam_step", &adam_step);
}
