"""The recorder of pyro_amd/ops/fuser.py on a machine without a GPU: a program that exercises every recorded
operator family is recorded with host tensors standing in for device tensors, scheduled into launch levels, and
every generated HIP source is compiled for gfx950 with hiprtc (tools/fuser_dry.py).  Nothing is launched: the
numbers are the GPU tests' business (tests/test_fuser_gpu.py); this pins the code generator's syntax and types
for both dtypes, and what the recorder takes."""
import os

import pytest
import torch


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/libhiprtc.so"), reason="needs hiprtc")
def test_every_generated_source_compiles_for_gfx950():
    from pyro_amd.ops import fuser
    from tools import fuser_dry

    fuser.UNFUSED.clear()
    del fuser_dry.SOURCES[:]
    with fuser_dry.dry():
        for dt in (torch.float32, torch.float64):
            before = dict(fuser.STATS)
            with fuser.Fuser():
                res = fuser_dry.program(dt)
            del res
            d = {k: fuser.STATS[k] - before[k] for k in fuser.STATS}
            assert d["recorded"] >= 150 and d["dead"] > 0, d
    # one launch per level, not per operator; the families' expressions, gathers, softmax, joins and long sums in
    assert 4 <= len(fuser_dry.SOURCES) <= 40
    text = "\n".join(fuser_dry.SOURCES)
    for needle in ("pa::Fam<", "fam_g<", "__shfl_xor", "long long x_", "__builtin_bit_cast(double", "_Pragma(\"unroll 4\")",
                   "PA_GROUP_SUM(s, 64)"):
        assert needle in text, needle
    # only what the program draws / creates with operators outside the recorder's reach is left to ATen
    assert set(fuser.UNFUSED) <= {"aten::randn", "aten::rand", "aten::sgn", "aten::arange", "aten::randint"}, \
        fuser.UNFUSED
    assert fuser._dev(torch.empty(1)) is False                 # (the stand-ins are gone)


@pytest.mark.parametrize("block", range(8))
def test_the_schedule_respects_every_dependence_of_random_programs(block):
    """Random programs (element-wise chains, in-place writes through views, short and long sums, joins,
    gathers / accumulate scatters, softmax, operators the recorder does not know, dropped references) re-run
    IN THE RECORDER'S SCHEDULE on the host (tools/fuser_dry.py::replaying: levels, merged kernels, partial
    flushes, sums with their operand's kernel, stores declared dead poisoned with NaN) give the eager run's
    numbers bit for bit."""
    from pyro_amd.ops import fuser
    from tools import fuser_dry

    for seed in range(block * 12, block * 12 + 12):
        run = fuser_dry.random_program(seed)
        want = run()
        before = dict(fuser.STATS)
        with fuser_dry.replaying():
            with fuser.Fuser():
                got = run()
        assert fuser.STATS["recorded"] - before["recorded"] > 10, seed
        assert len(got) == len(want), seed
        for a, b in zip(got, want):
            assert a.shape == b.shape and a.dtype == b.dtype, seed
            assert torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0)), (seed, a, b)


@pytest.mark.parametrize("block", range(6))
def test_the_schedule_of_forward_and_autograd_duals(block):
    """The same check with gradients: the programs' leaves require grad, a loss over the last results is
    differentiated inside the scope (the duals are recorded on the autograd thread), values and leaf gradients
    equal the eager run's bit for bit."""
    from pyro_amd.ops import fuser
    from tools import fuser_dry

    for seed in range(1000 + block * 10, 1000 + block * 10 + 10):
        run = fuser_dry.random_program(seed, n_ops=45, grad=True)
        want = run()
        with fuser_dry.replaying():
            with fuser.Fuser():
                got = run()
        assert len(got) == len(want), seed
        for a, b in zip(got, want):
            assert a.shape == b.shape and a.dtype == b.dtype, seed
            assert torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0)), (seed, a, b)


def _close(a, b, tol):
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_close(x, y, tol) for x, y in zip(a, b))
    if not isinstance(a, torch.Tensor):
        return True
    if a.dtype == torch.bool:
        return torch.equal(a, b)
    return a.shape == b.shape and torch.allclose(a, b, rtol=tol, atol=tol, equal_nan=True)


@pytest.mark.skipif(os.system("g++ --version > /dev/null 2>&1") != 0, reason="needs g++")
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.float64, 1e-11)])
def test_generated_code_executed_on_the_host_gives_the_operators_numbers(dtype, tol):
    """The generated HIP source ITSELF, compiled by g++ behind a page of shims and run on the host tensors
    (tools/fuser_dry.py::hosting): the program of the compile test -- element-wise chains, broadcasting, views,
    in-place writes, short and long sums, gathers / scatters, softmax, joins, transposed results, scalars in the
    argument table, the element-wise families with csrc/dist_fam.h's expressions, and the autograd duals of all
    of it -- against the eager run."""
    from pyro_amd.ops import fuser
    from tools import fuser_dry

    torch.manual_seed(11)
    want = fuser_dry.program(dtype)
    torch.manual_seed(11)
    before = dict(fuser.STATS)
    with fuser_dry.hosting():
        with fuser.Fuser():
            got = fuser_dry.program(dtype)
    assert fuser.STATS["kernels"] - before["kernels"] >= 4
    assert _close(got, want, tol)


@pytest.mark.skipif(os.system("g++ --version > /dev/null 2>&1") != 0, reason="needs g++")
@pytest.mark.parametrize("grad", [False, True], ids=["forward", "with_duals"])
def test_generated_code_of_random_programs_on_the_host(grad):
    from pyro_amd.ops import fuser
    from tools import fuser_dry

    for seed in range(5000, 5006) if not grad else range(6000, 6004):
        run = fuser_dry.random_program(seed, n_ops=60 if not grad else 45, grad=grad, long_sums=True)
        want = run()
        with fuser_dry.hosting():
            with fuser.Fuser():
                got = run()
        assert _close(got, want, 3e-5), seed


def _flush_after_differentiation(monkeypatch):
    """On the device a result reaches the host through a copy operator, which materialises what it reads; a host
    tensor's .numpy() is no operator at all: materialise explicitly before a case reads its gradients."""
    from pyro_amd.ops import fuser
    grad, backward = torch.autograd.grad, torch.Tensor.backward

    def grad_then_flush(*a, **kw):
        out = grad(*a, **kw)
        if fuser.active() is not None:
            fuser.active().flush()
        return out

    def backward_then_flush(self, *a, **kw):
        backward(self, *a, **kw)
        if fuser.active() is not None:
            fuser.active().flush()
    monkeypatch.setattr(torch.autograd, "grad", grad_then_flush)
    monkeypatch.setattr(torch.Tensor, "backward", backward_then_flush)
    import numpy as np
    compare = np.testing.assert_allclose

    def flush_then_compare(*a, **kw):
        if fuser.active() is not None:
            fuser.active().flush()
        return compare(*a, **kw)
    monkeypatch.setattr(np.testing, "assert_allclose", flush_then_compare)


@pytest.fixture
def _cpu_backend(oracle_backend):
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


@pytest.mark.skipif(os.system("g++ --version > /dev/null 2>&1") != 0, reason="needs g++")
@pytest.mark.parametrize("which", [1, 3])
def test_hmm_under_markov_through_the_recorder_matches_the_reference(_cpu_backend, monkeypatch, which):
    """examples/hmm.py model_1 / model_3 under pyro.markov with TraceEnum_ELBO, on the host: the package's
    kernels answered by the oracle, every other operator of the time steps -- and of the backward pass --
    recorded, batched level by level and EXECUTED from the generated code (tools/fuser_dry.py::hosting); loss
    and gradients against the unmodified reference's (tests/golden/hmm.npz), as the plain host run's."""
    import numpy as np
    from pyro_amd.ops import fuser
    from tests import enum_cases as ec
    from tools import fuser_dry

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hmm.npz"), allow_pickle=False)
    _flush_after_differentiation(monkeypatch)
    before = dict(fuser.STATS)
    with fuser_dry.hosting():
        with fuser.Fuser():
            ec.run_hmm(g, torch.device("cpu"), which, fused_chain=True)
    d = {k: fuser.STATS[k] - before[k] for k in fuser.STATS}
    assert d["recorded"] > 200 and d["kernels"] < d["recorded"] // 4, d


@pytest.mark.skipif(os.system("g++ --version > /dev/null 2>&1") != 0, reason="needs g++")
@pytest.mark.parametrize("case", ["lda_fused", "lda_generic", "gmm", "gmm_subsampled"])
def test_enumerated_models_through_the_recorder_match_the_reference(_cpu_backend, monkeypatch, case):
    """examples/lda.py (word topics summed out: the fused factor kernel answered by the oracle, or the generic
    log-space contraction) and the Gaussian mixture with its assignments enumerated, TraceEnum_ELBO on the
    host through the recorder with the generated code executed: loss and gradients of the unmodified reference
    (tests/golden/enum.npz)."""
    import numpy as np
    from pyro_amd.ops import fuser
    from tests import enum_cases as ec
    from tools import fuser_dry

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "enum.npz"), allow_pickle=False)
    _flush_after_differentiation(monkeypatch)
    if case == "lda_generic":
        import pyro_amd.ops.contract as c
        monkeypatch.setattr(c, "_try_fused_lda", lambda *a: None)
    before = dict(fuser.STATS)
    with fuser_dry.hosting():
        with fuser.Fuser():
            if case.startswith("lda"):
                ec.run_lda(g, torch.device("cpu"), monkeypatch, expect_fused=case == "lda_fused")
            else:
                ec.run_gmm(g, torch.device("cpu"), monkeypatch, case == "gmm_subsampled")
    assert fuser.STATS["recorded"] - before["recorded"] > 20


@pytest.mark.skipif(os.system("g++ --version > /dev/null 2>&1") != 0, reason="needs g++")
@pytest.mark.parametrize("case", ["eight_schools", "logreg_f64", "logreg_fused", "scale_mask", "hier", "hier_fused"])
def test_svi_models_through_the_recorder_match_the_reference(_cpu_backend, monkeypatch, case):
    """Trace_ELBO on the host through the recorder (generated code executed, small sites and constrained
    parameters recorded instead of launched): eight schools' loss, gradients and 30-step Adam trajectory, the
    logistic regressions (materialised logits and the GLM site), scale / mask / subsampling, the hierarchical
    model -- against the unmodified reference's golden vectors, at the plain host run's tolerances."""
    import numpy as np
    from pyro_amd.ops import fuser
    from tests import models
    from tools import fuser_dry

    def load(name):
        return np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"), allow_pickle=False)
    _flush_after_differentiation(monkeypatch)
    cpu = torch.device("cpu")
    before = dict(fuser.STATS)
    with fuser_dry.hosting():
        with fuser.Fuser():
            if case == "eight_schools":
                models.run_eight_schools(load("eight_schools"), cpu, monkeypatch, rtol=1e-9)
            elif case.startswith("logreg"):
                models.run_logreg(load("logreg_f64"), cpu, monkeypatch, fused=case.endswith("fused"),
                                  dtype=torch.float64, rtol=1e-9)
            elif case == "scale_mask":
                models.run_scale_mask(load("scale_mask"), cpu, monkeypatch, rtol=1e-9)
            else:
                models.run_hier(load("hier"), cpu, monkeypatch, fused=case.endswith("fused"), rtol=1e-9)
    assert fuser.STATS["recorded"] - before["recorded"] > 10


@pytest.mark.skipif(os.system("g++ --version > /dev/null 2>&1") != 0, reason="needs g++")
@pytest.mark.skipif(not os.environ.get("PYRO_AMD_SLOW_TESTS"),
                    reason="~40 s each (a g++ build per distinct generated kernel): PYRO_AMD_SLOW_TESTS=1 runs them")
@pytest.mark.parametrize("case", ["enum_potential", "constrained_potentials"])
def test_model_potentials_through_the_recorder_match_the_reference(_cpu_backend, monkeypatch, case):
    """The potential of a model as HMC / NUTS evaluate it (the conditioned model under the handlers, the transforms'
    Jacobians, autograd), through the recorder on the host: discrete latents summed out of the potential, and
    constrained supports (positive, unit interval, simplex, ...) -- values and gradients against the unmodified
    reference's (tests/golden/mcmc_*.npz), as the plain host run's."""
    from pyro_amd.ops import fuser
    from tests import mcmc_cases as mc
    from tools import fuser_dry

    _flush_after_differentiation(monkeypatch)
    before = dict(fuser.STATS)
    with fuser_dry.hosting():
        with fuser.Fuser():
            if case == "enum_potential":
                mc.run_enum_potential_vs_reference(torch.device("cpu"))
            else:
                mc.run_constrained_potentials_vs_reference(torch.device("cpu"))
    assert fuser.STATS["recorded"] - before["recorded"] > 10


def test_persistent_cache_key_covers_source_compiler_and_options(tmp_path, monkeypatch):
    """The name of a cached code object is a digest of everything the
    object depends on: the generated source, the compiler's version (hiprtc major.minor + HIP runtime), the
    options (which name the architecture) and the kernel's name; where it lives follows PYRO_AMD_RTC_CACHE."""
    from pyro_amd.ops import fuser
    src = 'extern "C" __global__ void k(T t) { }'
    v = (7, 2, 70253)
    k0 = fuser.rtc_cache_key(src, v)
    assert len(k0) == 64 and k0 == fuser.rtc_cache_key(src, v)
    others = {fuser.rtc_cache_key(src + " ", v), fuser.rtc_cache_key(src, (7, 3, 70253)),
              fuser.rtc_cache_key(src, (7, 2, 70254)),
              fuser.rtc_cache_key(src, v, options=("--offload-arch=gfx942",) + fuser.RTC_OPTIONS[1:]),
              fuser.rtc_cache_key(src, v, kernel="k2")}
    assert len(others) == 5 and k0 not in others
    # length-prefixed fields: moving a character between two of them changes the digest
    assert fuser.rtc_cache_key("ab", v, kernel="c") != fuser.rtc_cache_key("a", v, kernel="bc")
    assert fuser.RTC_OPTIONS[0] == "--offload-arch=gfx950"
    # ... and the options in the key are the ones the native side compiles with
    import os
    rtc = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "pyro_amd", "csrc", "rtc.hip")).read()
    assert ", ".join('"%s"' % o for o in fuser.RTC_OPTIONS) in rtc
    monkeypatch.setenv("PYRO_AMD_RTC_CACHE", str(tmp_path / "c"))
    assert fuser.rtc_cache_dir() == str(tmp_path / "c") and (tmp_path / "c").is_dir()
    for off in ("0", "off", ""):
        monkeypatch.setenv("PYRO_AMD_RTC_CACHE", off)
        assert fuser.rtc_cache_dir() is None
    monkeypatch.delenv("PYRO_AMD_RTC_CACHE")
    monkeypatch.setenv("HOME", str(tmp_path))
    assert fuser.rtc_cache_dir() == str(tmp_path / ".cache" / "pyro_amd" / "rtc")
