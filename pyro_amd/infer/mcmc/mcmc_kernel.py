"""MCMCKernel: the interface MCMC drives (reference seam 3(iii):
pyro/infer/mcmc/mcmc_kernel.py:7-80)."""
from abc import ABCMeta, abstractmethod


class MCMCKernel(object, metaclass=ABCMeta):
    def setup(self, warmup_steps, *args, **kwargs):
        """Optional: set up anything needed before the first ``sample`` call."""
        pass

    def cleanup(self):
        pass

    def logging(self):
        """An OrderedDict of name -> formatted value shown while sampling."""
        return None

    def diagnostics(self):
        """A dict of diagnostics available when the run completes."""
        return {}

    def end_warmup(self):
        pass

    @property
    def initial_params(self):
        raise NotImplementedError

    @initial_params.setter
    def initial_params(self, params):
        raise NotImplementedError

    @abstractmethod
    def sample(self, params):
        """One transition: params (dict of tensors) -> new params."""
        raise NotImplementedError

    def __call__(self, params):
        return self.sample(params)
