"""The C-ABI shared library loads (no GPU needed) and exports exactly what include/pyro_amd.h
declares; the ctypes binding covers every declared entry point; the product fails loudly
without a GPU."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "pyro_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", src)))


def test_header_vs_binding_vs_library():
    from pyro_amd import _lib
    from pyro_amd.csrc.build import build_library

    build_library()
    declared = header_functions()
    assert declared, "no declarations parsed"
    assert declared == _lib.exported_symbols()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True,
                         check=True).stdout
    exported = sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\b", out)))
    assert exported == declared
    lib = _lib.load()
    assert lib.pa_abi_version() == _lib.ABI_VERSION == 8


def test_no_torch_types_in_abi():
    src = open(os.path.join(ROOT, "include", "pyro_amd.h")).read()
    assert "at::" not in src and "torch" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert 'extern "C"' in src


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_on_cpu_tensors():
    import pyro_amd.distributions as dist

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dist.Normal(torch.zeros(3), torch.ones(3)).log_prob(torch.zeros(3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dist.Normal(torch.zeros(3), torch.ones(3)).rsample()


def test_argument_errors_before_launch():
    """Error convention: invalid arguments -> ValueError (PA_ERR_INVALID) without touching a device."""
    import ctypes

    from pyro_amd import _lib

    lib = _lib.load()
    rc = lib.pa_philox_normal(None, -1, 0, 0, 0, None, None)
    assert rc == _lib.PA_ERR_INVALID and b"n=" in lib.pa_last_error()
    rc = lib.pa_glm_bernoulli_fwd_bwd(None, None, None, None, None, 1.0, 10, 200, 4, None, None, None,
                                      None, 0, None)
    assert rc == _lib.PA_ERR_UNSUPPORTED
    with pytest.raises(_lib.Unsupported):
        _lib.check(rc)
    assert lib.pa_glm_bernoulli_workspace(10, 200, 4) == 0
    assert lib.pa_dist_log_prob_sum_workspace(4, 100) == 0       # single-launch small-site path
    assert lib.pa_dist_log_prob_sum_workspace(4, 100000) > 0


def test_torch_library_shim_loads_and_registers_the_schemas():
    """csrc/torch_ops.cpp: the TORCH_LIBRARY registration over the C-ABI loads without a GPU and the
    dispatcher knows the four ops (no compute here)."""
    import torch

    from pyro_amd.csrc.build import build_torch_ops
    from pyro_amd.ops import torch_library
    build_torch_ops()
    assert torch_library.available()
    # (ll, gw, gb, workspace): the workspace is an output so that a chained tail can keep it alive
    for name, nret in (("glm_pack_planes", 1), ("glm_bernoulli_planes", 4), ("glm_bernoulli", 4),
                       ("glm_chain", 2), ("adam_step", 0)):
        schema = getattr(torch.ops.pyro_amd, name).default._schema
        assert len(schema.returns) == nret, schema
    # shape functions (register_fake): meta tensors flow through without touching a device
    w = torch.empty((64, 32), device="meta")
    ll, gw, gb, ws = torch.ops.pyro_amd.glm_bernoulli_planes(
        torch.empty((16,), dtype=torch.uint8, device="meta"), torch.empty((100,), device="meta"), w,
        None, 1.0, 100, 32, 1)
    assert ll.shape == (64,) and gw.shape == (64, 32) and gb.shape == (64,) and ws.dtype == torch.uint8
    # every kernel-backed autograd Function of the package is a NAMED dispatcher op pair (its trampoline:
    # pyro_amd::fn_<name>) ...
    import pyro_amd  # noqa: F401
    names = torch_library.registered_ops()
    assert "pyro_amd::adam_step" in names
    for op in ("meanfield_normal_sample", "multi_log_prob_sum", "dist_log_prob_sum", "dist_log_prob",
               "lda_factor", "logsumexp_terms", "logchain", "normal_rsample", "standard_gamma",
               "mvn_tril_sample", "glm_bernoulli_ll", "glm_bernoulli_grouped_ll"):
        assert "pyro_amd::fn_" + op in names and "pyro_amd::fn_" + op + "_bwd" in names, op
        assert str(getattr(torch.ops.pyro_amd, "fn_" + op).default._schema).endswith(
            "(Tensor[] tensors, int spec) -> Tensor[]")
    # ... and the ten a maintainer would bind first carry TYPED schemas in C++ (precedent:
    # the reference binds its one native file through torch's C++ extension loader,
    # pyro/distributions/spanning_tree.py:225-241): real argument lists, no process-local spec table
    typed = {
        "dist_log_prob_sum": "(int dist, Tensor value, Tensor? p0, Tensor? p1, Tensor? mask, float scale) -> "
                             "(Tensor rowsum, Tensor total)",
        "multi_log_prob_sum": "(int[] dist, Tensor[] value, Tensor?[] p0, Tensor?[] p1, float[] coef, "
                              "float coef_all) -> Tensor",
        "meanfield_normal_sample": "(Tensor[] loc, Tensor[] rho, int P, int seed, int[] offsets, "
                                   "Tensor? offset_dev) -> (Tensor[] z, Tensor[] scale, Tensor[] eps)",
        "exp_site": "(Tensor u, float lower) -> (Tensor value, Tensor log_density)",
        "mvn_tril_sample": "(Tensor loc, Tensor rho, Tensor A, int P, int seed, int offset, Tensor? offset_dev) -> "
                           "(Tensor eps, Tensor z, Tensor logq)",
        "logsumexp_terms": "(Tensor[] terms, int[] sizes, int rdim) -> Tensor",
        "logchain": "(Tensor unary, Tensor pairwise) -> (Tensor log_z, Tensor grad_unary, Tensor grad_pairwise)",
        "mixture_fwd_bwd": "(int dist, Tensor x, Tensor a, Tensor p0, Tensor? p1) -> Tensor",
        "lda_factor_indexed": "(Tensor words, Tensor index, Tensor log_theta, Tensor log_phi) -> "
                              "(Tensor out_doc, Tensor g_theta, Tensor g_phi)",
        "tall_linear_act": "(Tensor G, Tensor weight, Tensor? bias, Tensor? y_mul, bool sigmoid_out, "
                           "bool transpose_weight) -> Tensor",
    }
    for op, sig in typed.items():
        assert "pyro_amd::" + op in names, op
        assert str(getattr(torch.ops.pyro_amd, op).default._schema) == "pyro_amd::" + op + sig, op
    adv = torch.ops.pyro_amd.nuts_tree_run_advance.default._schema
    assert len(adv.arguments) == 27 and len(adv.returns) == 0 and adv.arguments[0].alias_info.is_write
    assert set(torch_library.TYPED_OPS) >= set(typed)
    # their shape functions answer on meta tensors
    m = lambda *sh, **kw: torch.empty(sh, device="meta", **kw)  # noqa: E731
    lz, gu, gp = torch.ops.pyro_amd.logchain(m(5, 7, 3), m(1, 6, 3, 3))
    assert lz.shape == (5,) and gu.shape == (5, 7, 3) and gp.shape == (5, 6, 3, 3)
    z, sc, eps = torch.ops.pyro_amd.meanfield_normal_sample([m(4), m(1)], [m(4), m(1)], 8, 0, [0, 64], None)
    assert [t.shape for t in z] == [(8, 4), (8, 1)] and sc[0].shape == (4,) and eps[1].shape == (8, 1)
    assert torch.ops.pyro_amd.logsumexp_terms([m(3, 1, 4), m(1, 5, 4)], [3, 5, 4], 2).shape == (3, 5)
    assert torch.ops.pyro_amd.tall_linear_act(m(100, 16), m(24, 16), m(24), None, True, True).shape == (100, 24)
    mx = torch.ops.pyro_amd.mixture_fwd_bwd(0, m(1000), m(4, 5), m(4, 5), m(1, 1))
    assert mx.shape == (4, 16) and mx.dtype == torch.float64
    rs, tot = torch.ops.pyro_amd.dist_log_prob_sum(0, m(6, 9), m(9), m(1), None, 1.0)
    assert rs.shape == (6,) and tot.shape == ()
