"""tests/infer/mcmc/test_mcmc_api.py of the reference restated against the drop-in API: the MCMC
driver with a user-written MCMCKernel, sample selection (num_draws / group_by_chain), hooks with a
model that has no latent site, diagnostics, a bare potential_fn, save_params.  (StreamingMCMC is
not part of the scoped path.)"""
from functools import partial

import numpy as np
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd import poutine
from pyro_amd.infer.mcmc import HMC, MCMC, NUTS, MCMCKernel, initialize_model


class PriorKernel(MCMCKernel):
    """Disregards the value of the current trace and samples from the prior (test_mcmc_api.py:25-69)."""

    def __init__(self, model):
        self.model = model
        self.data = None
        self._initial_params = None
        self._prototype_trace = None
        self.transforms = None

    def setup(self, warmup_steps, data):
        self.data = data
        init_params, potential_fn, transforms, model_trace = initialize_model(
            self.model, model_args=(data,))
        if self._initial_params is None:
            self._initial_params = init_params
        if self.transforms is None:
            self.transforms = transforms
        self._prototype_trace = model_trace

    def diagnostics(self):
        return {"dummy_key": "dummy_value"}

    @property
    def initial_params(self):
        return self._initial_params

    @initial_params.setter
    def initial_params(self, params):
        self._initial_params = params

    def cleanup(self):
        self.data = None

    def sample_params(self):
        trace = poutine.trace(self.model).get_trace(self.data)
        return {k: v["value"] for k, v in trace.iter_stochastic_nodes()}

    def sample(self, params):
        new_params = self.sample_params()
        assert params.keys() == new_params.keys()
        for k, v in params.items():
            assert new_params[k].shape == v.shape
        return new_params


def make_normal_normal(device):
    def normal_normal_model(data):
        x = torch.tensor([0.0], device=device)
        y = pyro.sample("y", dist.Normal(x, torch.ones(data.shape, device=device)))
        pyro.sample("obs", dist.Normal(y, torch.tensor([1.0], device=device)), obs=data)
    return normal_normal_model


def run_mcmc_interface(device, num_draws, group_by_chain, num_chains):
    pyro.set_rng_seed(0)
    num_samples = 2000
    data = torch.tensor([1.0], device=device)
    model = make_normal_normal(device)
    initial_params, _, transforms, _ = initialize_model(model, model_args=(data,), num_chains=num_chains)
    kernel = PriorKernel(model)
    mcmc = MCMC(kernel, num_samples=num_samples, warmup_steps=100, initial_params=initial_params,
                num_chains=num_chains, mp_context="spawn", transforms=transforms)
    mcmc.run(data)
    samples = mcmc.get_samples(num_draws, group_by_chain=group_by_chain)
    expected = num_draws if num_draws is not None else num_samples
    if group_by_chain:
        shape = (mcmc.num_chains, expected, 1)
    elif num_draws is not None:
        shape = (expected, 1)
    else:
        shape = (mcmc.num_chains * expected, 1)
    assert tuple(samples["y"].shape) == shape
    if group_by_chain:
        samples = {k: v.reshape((-1,) + v.shape[2:]) for k, v in samples.items()}
    assert abs(float(samples["y"].mean())) < 0.1
    assert abs(float(samples["y"].std()) - 1.0) < 0.1


def _empty_model():
    return torch.tensor(1)


def _hook(iters, kernel, samples, stage, i):
    assert samples == {}
    iters.append((stage, i))


def run_null_model_with_hook(kernel_cls, jit, num_chains):
    num_warmup, num_samples = 10, 10
    initial_params, potential_fn, transforms, _ = initialize_model(_empty_model, num_chains=num_chains)
    iters = []
    kern = kernel_cls(potential_fn=potential_fn, transforms=transforms, jit_compile=jit)
    mcmc = MCMC(kern, num_samples=num_samples, warmup_steps=num_warmup, initial_params=initial_params,
                hook_fn=partial(_hook, iters), num_chains=num_chains)
    mcmc.run()
    assert mcmc.get_samples() == {}
    if num_chains == 1:
        assert iters == [("Warmup", i) for i in range(num_warmup)] + \
            [("Sample", i) for i in range(num_samples)]


def run_mcmc_diagnostics(device, num_chains):
    data = torch.tensor([2.0], device=device).repeat(3)
    model = make_normal_normal(device)
    initial_params, _, transforms, _ = initialize_model(model, model_args=(data,), num_chains=num_chains)
    mcmc = MCMC(PriorKernel(model), num_samples=10, warmup_steps=10, num_chains=num_chains,
                mp_context="spawn", initial_params=initial_params, transforms=transforms)
    mcmc.run(data)
    diagnostics = mcmc.diagnostics()
    assert diagnostics["y"]["n_eff"].shape == data.shape
    assert diagnostics["y"]["r_hat"].shape == data.shape
    assert diagnostics["dummy_key"] == {"chain {}".format(i): "dummy_value" for i in range(num_chains)}


def run_model_with_potential_fn(device):
    init_params = {"z": torch.tensor(0.0, device=device)}

    def potential_fn(params):
        return params["z"]

    mcmc = MCMC(HMC(potential_fn=potential_fn), num_samples=10, warmup_steps=10,
                initial_params=init_params)
    mcmc.run()
    assert tuple(mcmc.get_samples()["z"].shape) == (10,)


def run_save_params(device, save_params, Kernel, options):
    save_params = list(save_params)
    z = lambda v: torch.tensor(v, device=device)   # noqa: E731

    def model():
        x = pyro.sample("x", dist.Normal(z(0.0), 1.0))
        with pyro.plate("plate", 2):
            y = pyro.sample("y", dist.Normal(x, 1.0))
            pyro.sample("obs", dist.Normal(y, 1.0), obs=torch.zeros(2, device=device))

    mcmc = MCMC(Kernel(model, **options), warmup_steps=2, num_samples=4, save_params=save_params)
    mcmc.run()
    samples = mcmc.get_samples()
    assert set(samples.keys()) == set(save_params)
    diagnostics = {k: v for k, v in mcmc.diagnostics().items() if k in "xy"}
    assert set(diagnostics.keys()) == set(save_params)
    mcmc.summary()      # smoke test
