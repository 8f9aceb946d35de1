"""The reference's own ELBO-gradient known-answer tests (tests/kat_cases.py) on the MI355X: the HIP
kernels under the unchanged host logic."""
import pytest

from tests import kat_cases as kc

pytestmark = pytest.mark.gpu
RSAMPLE = [(True, None), (True, False), (True, True), (False, None)]
IDS = ["reparam", "reparam-False", "reparam-True", "nonreparam"]


@pytest.mark.parametrize("reparameterized,has_rsample", RSAMPLE, ids=IDS)
@pytest.mark.parametrize("elbo", ["Trace_ELBO", "TraceEnum_ELBO"])
def test_particle_gradient(gpu, elbo, reparameterized, has_rsample):
    kc.run_particle_gradient(gpu, elbo, reparameterized, has_rsample)


@pytest.mark.parametrize("scale", [1.0, 2.0], ids=["unscaled", "scaled"])
@pytest.mark.parametrize("reparameterized,has_rsample", RSAMPLE, ids=IDS)
@pytest.mark.parametrize("subsample", [False, True], ids=["full", "subsample"])
@pytest.mark.parametrize("elbo", ["Trace_ELBO", "DiffTrace_ELBO", "TraceGraph_ELBO", "TraceMeanField_ELBO",
                                  "TraceEnum_ELBO"])
def test_subsample_gradient(gpu, elbo, reparameterized, has_rsample, subsample, scale):
    try:
        kc.run_subsample_gradient(gpu, elbo, reparameterized, has_rsample, subsample, scale)
    except NotImplementedError as e:       # the reference test: `with xfail_if_not_implemented()`
        pytest.xfail(str(e))


@pytest.mark.parametrize("reparameterized", [True, False], ids=["reparam", "nonreparam"])
@pytest.mark.parametrize("elbo", ["Trace_ELBO", "TraceGraph_ELBO", "TraceEnum_ELBO"])
def test_plate(gpu, elbo, reparameterized):
    kc.run_plate(gpu, elbo, reparameterized)


@pytest.mark.parametrize("reparameterized", [True, False], ids=["reparam", "nonreparam"])
@pytest.mark.parametrize("elbo", ["Trace_ELBO", "TraceEnum_ELBO"])
def test_plate_elbo_vectorized_particles(gpu, elbo, reparameterized):
    kc.run_plate(gpu, elbo, reparameterized, vectorized_elbo=True)


def test_plating_sums(gpu):
    kc.run_plating_sums(gpu)


@pytest.mark.parametrize("elbo,reparameterized,n_steps", [("Trace_ELBO", True, 5000),
                                                          ("TraceMeanField_ELBO", True, 3000),
                                                          ("Trace_ELBO", False, 15000)],
                         ids=["reparameterized", "analytic_kl", "nonreparameterized"])
def test_normal_normal_convergence(gpu, elbo, reparameterized, n_steps):
    kc.run_normal_normal(gpu, elbo, reparameterized, n_steps)


@pytest.mark.parametrize("reparameterized,n_steps,prec,baseline", [
    (True, 1500, 0.02, None), (False, 5000, 0.05, None),
    (False, 5000, 0.05, {"use_decaying_avg_baseline": True, "baseline_beta": 0.9})],
    ids=["reparameterized", "nonreparameterized", "decaying_avg_baseline"])
def test_tracegraph_normal_normal(gpu, reparameterized, n_steps, prec, baseline):
    kc.run_tracegraph_normal_normal(gpu, reparameterized, n_steps, prec, baseline)


@pytest.mark.parametrize("reparameterized,vectorized,n_steps", [(True, False, 10000), (False, False, 10000),
                                                                (True, True, 5000)],
                         ids=["reparameterized", "nonreparameterized", "reparameterized_vectorized"])
def test_bernoulli_beta_convergence(gpu, reparameterized, vectorized, n_steps):
    kc.run_bernoulli_beta(gpu, reparameterized, n_steps, vectorized, hip_graph=True)


def test_poisson_gamma_convergence(gpu):
    kc.run_poisson_gamma(gpu, True, 10000, hip_graph=True)
