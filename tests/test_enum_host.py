"""TraceEnum_ELBO host logic + plated sum-product on CPU (kernels answered by the oracle), against
loss/gradients of the unmodified reference (tests/golden/enum.npz)."""
import os

import numpy as np
import pytest
import torch

from tests import enum_cases as ec

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def test_golden_lda_loss_is_the_exact_marginal():
    g = load("enum")
    np.testing.assert_allclose(ec.lda_brute_force_loss(g), float(g["lda/loss"]), rtol=1e-7)


@pytest.fixture
def _cpu_backend(oracle_backend):
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


def test_lda_generic_contraction(_cpu_backend, monkeypatch):
    # the generic message passing (broadcast add + logsumexp + plate sums) with the LDA-shaped
    # leaf's fused route switched off, as tests/test_enum_gpu.py does on the device
    import pyro_amd.ops.contract as c
    monkeypatch.setattr(c, "_try_fused_lda", lambda *a: None)
    ec.run_lda(load("enum"), torch.device("cpu"), monkeypatch, expect_fused=False)


@pytest.mark.parametrize("sub", [False, True])
def test_gmm(_cpu_backend, monkeypatch, sub):
    ec.run_gmm(load("enum"), torch.device("cpu"), monkeypatch, sub)


def test_lda_fused_route_with_oracle_kernel(_cpu_backend, monkeypatch):
    """Force the fused route on CPU tensors (oracle LDA kernel): checks the pattern matcher, the
    (table, index) extraction and the autograd wiring of _LdaFactor."""
    ec.run_lda(load("enum"), torch.device("cpu"), monkeypatch, expect_fused=True)


@pytest.mark.parametrize("fused_chain", [True, False], ids=["fused_chain", "generic"])
@pytest.mark.parametrize("which", [1, 3])
def test_hmm_under_markov_matches_reference(_cpu_backend, which, fused_chain):
    ec.run_hmm(load("hmm"), torch.device("cpu"), which, fused_chain=fused_chain)


def test_discrete_hmm_matches_reference(_cpu_backend):
    ec.run_discrete_hmm(load("discrete_hmm"), torch.device("cpu"))


def test_hmm_vectorised_over_time_equals_markov_model(_cpu_backend):
    ec.run_hmm_vectorised_equals_markov(torch.device("cpu"))


def test_sequential_enumeration_in_the_model_raises(_cpu_backend):
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.infer import TraceEnum_ELBO

    def model():
        pyro.sample("z", dist.Categorical(torch.ones(3) / 3), infer={"enumerate": "sequential"})

    def guide():
        pass

    with pytest.raises(NotImplementedError):
        TraceEnum_ELBO(max_plate_nesting=0).loss_and_grads(model, guide)


def test_sequential_guide_enumeration_equals_parallel(_cpu_backend):
    from tests import enum_kat_cases
    enum_kat_cases.run_sequential_equals_parallel(torch.device("cpu"))


# ---- the reference's hand-vs-auto enumeration KATs (tests/enum_kat_cases.py) ---------------------
from tests import enum_kat_cases as ekc   # noqa: E402

CPU = torch.device("cpu")


@pytest.mark.parametrize("scale", [1, 10])
@pytest.mark.parametrize("variant", [1, 2, 3])
def test_elbo_enumerate_chain(_cpu_backend, variant, scale):
    ekc.run_enumerate_chain(CPU, variant, scale)


@pytest.mark.parametrize("scale", [1, 10])
@pytest.mark.parametrize("num_samples,num_masked", [(1, 1), (2, 2), (3, 2)],
                         ids=["single", "batch", "masked"])
@pytest.mark.parametrize("variant", [1, 2, 3])
def test_elbo_enumerate_plate(_cpu_backend, variant, num_samples, num_masked, scale):
    ekc.run_enumerate_plate(CPU, variant, num_samples, num_masked, scale)


@pytest.mark.parametrize("scale", [1, 10])
@pytest.mark.parametrize("variant", [1, 2])
def test_elbo_enumerate_plates(_cpu_backend, variant, scale):
    ekc.run_enumerate_plates(CPU, variant, scale)


@pytest.mark.parametrize("history", [1, 2, 3])
def test_markov_history_equals_brute_force(_cpu_backend, history):
    ekc.run_markov_history(CPU, history)


def test_guide_side_markov_enumeration_is_exact(_cpu_backend):
    ekc.run_guide_side_markov(CPU)


def test_guide_enumeration_is_the_exact_expectation(_cpu_backend):
    ekc.run_guide_enumeration_closed_form(CPU)


def test_guide_enumeration_and_dice_match_reference(_cpu_backend):
    ec.run_guide_enum_vs_reference(load("guide_enum"), CPU)


@pytest.mark.parametrize("scale", [1, 10])
@pytest.mark.parametrize("method", ["loss", "differentiable_loss", "loss_and_grads"])
@pytest.mark.parametrize("enumerate1", ["sequential", "parallel"])
def test_elbo_bern(_cpu_backend, method, enumerate1, scale):
    ekc.run_elbo_bern(CPU, method, enumerate1, scale)


@pytest.mark.parametrize("method", ["differentiable_loss", "loss_and_grads"])
@pytest.mark.parametrize("enums", [("parallel",) * 3, ("sequential",) * 3,
                                   ("sequential", "parallel", "sequential"),
                                   ("parallel", "sequential", "parallel")], ids="-".join)
def test_elbo_berns(_cpu_backend, method, enums):
    ekc.run_elbo_berns(CPU, method, enums)


@pytest.mark.parametrize("max_plate_nesting", [0, 1])
@pytest.mark.parametrize("enums", [("parallel",) * 3, ("sequential",) * 3,
                                   ("sequential", "parallel", "parallel"),
                                   ("parallel", "parallel", "sequential")], ids="-".join)
def test_elbo_categoricals(_cpu_backend, enums, max_plate_nesting):
    ekc.run_elbo_categoricals(CPU, enums, max_plate_nesting)


@pytest.mark.parametrize("elbo", ["Trace_ELBO", "TraceGraph_ELBO", "TraceEnum_ELBO"])
def test_vectorized_num_particles(_cpu_backend, elbo):
    ekc.run_vectorized_num_particles(CPU, elbo)


@pytest.mark.parametrize("enumerate_,expand", [(None, False), ("sequential", False), ("sequential", True),
                                               ("parallel", False), ("parallel", True)])
@pytest.mark.parametrize("num_particles", [1, 50])
def test_enum_discrete_vectorized_num_particles(_cpu_backend, enumerate_, expand, num_particles):
    ekc.run_enum_discrete_vectorized_num_particles(CPU, enumerate_, expand, num_particles)


@pytest.mark.parametrize("enums", [("sequential",) * 4, ("parallel",) * 4,
                                   ("parallel", "sequential", "parallel", "sequential"),
                                   ("sequential", "parallel", "sequential", "parallel"),
                                   ("parallel", "parallel", "sequential", "sequential"),
                                   ("sequential", "sequential", "parallel", "parallel"),
                                   ("parallel", "sequential", "sequential", "parallel")], ids="-".join)
def test_elbo_plate_plate(_cpu_backend, enums):
    ekc.run_elbo_plate_plate(CPU, enums)


@pytest.mark.parametrize("tmc,expand", [("diagonal", False), ("mixture", False), ("diagonal", True)])
def test_local_monte_carlo_sampling(_cpu_backend, tmc, expand):
    ekc.run_local_sampling(CPU, tmc=tmc, expand=expand)


def test_local_sampling_of_a_reparameterised_site(_cpu_backend):
    ekc.run_local_sampling_of_a_reparameterised_site(CPU)


def test_config_enumerate_arguments(_cpu_backend):
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd import poutine
    from pyro_amd.infer import config_enumerate
    for bad in (dict(default="bogus"), dict(expand=None), dict(num_samples=0),
                dict(default="sequential", num_samples=3)):
        with pytest.raises(ValueError):
            config_enumerate(lambda: None, **bad)

    def guide():
        pyro.sample("a", dist.Bernoulli(torch.tensor(0.5)))
        pyro.sample("b", dist.Normal(torch.tensor(0.0), 1.0))
        pyro.sample("c", dist.Bernoulli(torch.tensor(0.5)), infer={"enumerate": "sequential"})

    tr = poutine.trace(config_enumerate(guide)).get_trace()
    assert tr.nodes["a"]["infer"] == {"enumerate": "parallel", "expand": False}
    assert tr.nodes["b"]["infer"] == {}
    assert tr.nodes["c"]["infer"] == {"enumerate": "sequential", "expand": False}
    tr = poutine.trace(config_enumerate(guide, num_samples=7, tmc="full")).get_trace()
    assert tr.nodes["b"]["infer"] == {"enumerate": "parallel", "num_samples": 7, "expand": True,
                                      "tmc": "full"}


# ---- posterior of the enumerated sites: the reference's compute_marginals / sample_posterior tests
#      (tests/infer/test_enum.py:3725-4010) on the host logic -----------------------------------
from tests import posterior_kat_cases as pkc   # noqa: E402

_PRIORS = [("bernoulli", 0.2), ("categorical", [0.2, 0.8]), ("categorical", [0.2, 0.3, 0.5]),
           ("categorical", [0.2, 0.3, 0.3, 0.2]), ("onehot", [0.2, 0.8]), ("onehot", [0.2, 0.3, 0.5]),
           ("onehot", [0.2, 0.3, 0.3, 0.2])]
_RESTRICT = [(True, None, 1, False), (False, "sequential", 1, False), (False, "parallel", 1, False),
             (False, None, 2, False), (False, None, 2, True)]


@pytest.mark.parametrize("which,prior", _PRIORS)
def test_compute_marginals_single(_cpu_backend, which, prior):
    pkc.run_marginals_single(CPU, which, prior)


@pytest.mark.parametrize("ok,enumerate_guide,num_particles,vectorize_particles", _RESTRICT)
@pytest.mark.parametrize("what", ["marginals", "posterior"])
def test_posterior_restrictions(_cpu_backend, ok, enumerate_guide, num_particles, vectorize_particles,
                                what):
    pkc.run_marginals_restrictions(CPU, ok, enumerate_guide, num_particles, vectorize_particles, what)


@pytest.mark.parametrize("size", [1, 2, 3, 4, 10, 20])
def test_compute_marginals_hmm(_cpu_backend, size):
    pkc.run_marginals_hmm(CPU, size)


@pytest.mark.parametrize("observed", ["", "a", "b", "ab"])
def test_marginals_2678(_cpu_backend, observed):
    pkc.run_marginals_2678(CPU, observed)


def test_compute_marginals_matches_reference_on_a_plated_mixture(_cpu_backend):
    pkc.run_marginals_plated_golden(CPU, np.load(os.path.join(os.path.dirname(__file__), "golden",
                                                              "marginals.npz")), 1e-9)


@pytest.mark.parametrize("data", [[None, None], [0.0, None], [None, 0], [0.0, 0]])
def test_backwardsample_posterior_smoke(_cpu_backend, data):
    pkc.run_backwardsample_smoke(CPU, data)


def test_backwardsample_posterior_statistics(_cpu_backend):
    pkc.run_backwardsample_2(CPU, 4000)
    pkc.run_backwardsample_3(CPU, 4000)
    pkc.run_backwardsample_hmm_joint(CPU, 4, 3000)
