"""Gradient-free random-walk Metropolis in the unconstrained space of a model (the role of
pyro/infer/mcmc/rwkernel.py): isotropic Gaussian proposals whose step size is adapted during warm-up towards
a target acceptance probability.  A plain MCMCKernel: the driver runs it chain after chain."""
import math
from collections import OrderedDict

import torch

from .mcmc_kernel import MCMCKernel
from .util import initialize_model


class RandomWalkKernel(MCMCKernel):
    def __init__(self, model, init_step_size=0.1, target_accept_prob=0.234):
        if not isinstance(init_step_size, float) or init_step_size <= 0.0:
            raise ValueError("init_step_size must be a positive float.")
        if not isinstance(target_accept_prob, float) or not 0.0 < target_accept_prob < 1.0:
            raise ValueError("target_accept_prob must be a float in the interval (0, 1).")
        self.model = model
        self.init_step_size = init_step_size
        self.target_accept_prob = target_accept_prob
        self._restart()
        super().__init__()

    def _restart(self):
        self._t = 0
        self._log_step_size = math.log(self.init_step_size)
        self._accept_cnt = 0
        self._mean_accept_prob = 0.0

    def setup(self, warmup_steps, *args, **kwargs):
        self._restart()
        self._warmup_steps = warmup_steps
        self._initial_params, self.potential_fn, self.transforms, self._prototype_trace = \
            initialize_model(self.model, model_args=args, model_kwargs=kwargs)
        self._energy_last = self.potential_fn(self._initial_params)

    def sample(self, params):
        step_size = math.exp(self._log_step_size)
        proposal = {k: v + step_size * torch.randn(v.shape, dtype=v.dtype, device=v.device)
                    for k, v in params.items()}
        energy = self.potential_fn(proposal)
        accept_prob = float((self._energy_last - energy).exp().clamp(max=1.0))
        accepted = float(torch.rand(())) < accept_prob
        if accepted:
            params, self._energy_last = proposal, energy
        warming_up = self._t <= self._warmup_steps
        if warming_up:
            speed = max(0.001, 0.1 / math.sqrt(1 + self._t))
            self._log_step_size += speed * (accept_prob - self.target_accept_prob)
        self._t += 1
        if self._t > self._warmup_steps:
            n = self._t - self._warmup_steps
            self._accept_cnt += int(accepted)
        else:
            n = self._t
        self._mean_accept_prob += (accept_prob - self._mean_accept_prob) / n
        return dict(params)

    @property
    def initial_params(self):
        return self._initial_params

    @initial_params.setter
    def initial_params(self, params):
        self._initial_params = params

    def logging(self):
        return OrderedDict([("step size", "{:.2e}".format(math.exp(self._log_step_size))),
                            ("acc. prob", "{:.3f}".format(self._mean_accept_prob))])

    def diagnostics(self):
        return {"acceptance rate": self._accept_cnt / max(self._t - self._warmup_steps, 1)}
