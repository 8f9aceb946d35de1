"""tests/infer/mcmc/test_mcmc_api.py of the reference on the MI355X (tests/mcmc_api_kat_cases.py)."""
import pytest

from pyro_amd.infer.mcmc import HMC, NUTS
from tests import mcmc_api_kat_cases as mk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("num_draws", [None, 1800, 2200])
@pytest.mark.parametrize("group_by_chain", [False, True])
@pytest.mark.parametrize("num_chains", [1, 2])
def test_mcmc_interface(gpu, num_draws, group_by_chain, num_chains):
    mk.run_mcmc_interface(gpu, num_draws, group_by_chain, num_chains)


@pytest.mark.parametrize("kernel", [HMC, NUTS])
@pytest.mark.parametrize("jit", [False, True])
@pytest.mark.parametrize("num_chains", [1, 2])
def test_null_model_with_hook(gpu, kernel, jit, num_chains):
    mk.run_null_model_with_hook(kernel, jit, num_chains)


@pytest.mark.parametrize("num_chains", [1, 2])
def test_mcmc_diagnostics(gpu, num_chains):
    mk.run_mcmc_diagnostics(gpu, num_chains)


def test_model_with_potential_fn(gpu):
    mk.run_model_with_potential_fn(gpu)


@pytest.mark.parametrize("save_params", ["xy", "x", "y"])
@pytest.mark.parametrize("Kernel,options", [(HMC, {}), (NUTS, {"max_tree_depth": 2})])
def test_save_params(gpu, save_params, Kernel, options):
    mk.run_save_params(gpu, save_params, Kernel, options)
