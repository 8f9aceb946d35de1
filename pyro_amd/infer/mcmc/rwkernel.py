"""Gradient-free random-walk Metropolis in the unconstrained space of a model (what
pyro/infer/mcmc/rwkernel.py offers), built from this package's pieces: the state is the flat [D] vector of
``Layout``, the accept test is done in log space, and the step size follows a Robbins-Monro schedule during
warm-up.  It is a plain ``MCMCKernel``, so the driver runs it chain after chain."""
import math
from collections import OrderedDict

import torch

from .mcmc_kernel import MCMCKernel
from .util import Layout, initialize_model


class _RobbinsMonroStep:
    """log step size += gain_t * (accept_prob - target), gain_t = max(1e-3, 0.1 / sqrt(1 + t)), while
    warming up; afterwards frozen."""

    def __init__(self, initial, target):
        self.log_step, self.target = math.log(initial), target

    @property
    def value(self):
        return math.exp(self.log_step)

    def update(self, t, accept_prob):
        self.log_step += max(1e-3, 0.1 / math.sqrt(1.0 + t)) * (accept_prob - self.target)


class _RunningRate:
    """Mean acceptance probability of the current phase and the count of accepted post-warm-up moves."""

    def __init__(self):
        self.mean, self.accepted = 0.0, 0

    def add(self, n, accept_prob, accepted, counted):
        self.mean += (accept_prob - self.mean) / n
        self.accepted += int(accepted and counted)


class RandomWalkKernel(MCMCKernel):
    """``RandomWalkKernel(model, init_step_size=0.1, target_accept_prob=0.234)``: isotropic Gaussian
    proposals ``z' = z + step * eps`` accepted with probability ``min(1, exp(U(z) - U(z')))``."""

    def __init__(self, model, init_step_size=0.1, target_accept_prob=0.234):
        if not (isinstance(init_step_size, float) and init_step_size > 0.0):
            raise ValueError("init_step_size must be a positive float.")
        if not (isinstance(target_accept_prob, float) and 0.0 < target_accept_prob < 1.0):
            raise ValueError("target_accept_prob must be a float in the interval (0, 1).")
        super().__init__()
        self.model = model
        self.init_step_size, self.target_accept_prob = init_step_size, target_accept_prob
        self._initial_params = None
        self._begin(0)

    def _begin(self, warmup_steps):
        self._t, self._warmup_steps = 0, warmup_steps
        self._step = _RobbinsMonroStep(self.init_step_size, self.target_accept_prob)
        self._rate = _RunningRate()

    # ---- MCMCKernel interface ---------------------------------------------------------------------------
    def setup(self, warmup_steps, *args, **kwargs):
        self._begin(warmup_steps)
        start, self.potential_fn, self.transforms, self._prototype_trace = initialize_model(
            self.model, model_args=args, model_kwargs=kwargs)
        self._initial_params = start
        self._layout = Layout({name: v.shape for name, v in start.items()})
        self._energy = self.potential_fn(start).detach()

    def sample(self, params):
        z = self._layout.flatten(params, 1, False)                                  # [1, D]
        proposal = self._layout.unflatten(z + self._step.value * torch.randn_like(z), False)
        energy = self.potential_fn(proposal).detach()
        log_ratio = float(self._energy - energy)                                    # log of the accept ratio
        accept_prob = 1.0 if log_ratio >= 0.0 else math.exp(log_ratio)
        accepted = math.log(max(float(torch.rand(())), 1e-300)) < log_ratio
        if accepted:
            params, self._energy = proposal, energy
        in_warmup = self._t <= self._warmup_steps
        if in_warmup:
            self._step.update(self._t, accept_prob)
        self._t += 1
        sampling = self._t > self._warmup_steps
        self._rate.add(self._t - self._warmup_steps if sampling else self._t, accept_prob, accepted, sampling)
        return dict(params)

    @property
    def initial_params(self):
        return self._initial_params

    @initial_params.setter
    def initial_params(self, params):
        self._initial_params = params

    def logging(self):
        return OrderedDict(
            (("step size", format(self._step.value, ".2e")), ("acc. prob", format(self._rate.mean, ".3f"))))

    def diagnostics(self):
        kept = max(self._t - self._warmup_steps, 1)
        return {"acceptance rate": self._rate.accepted / kept}
