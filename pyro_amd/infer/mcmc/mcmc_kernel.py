"""What the MCMC driver asks of a transition kernel (the interface of pyro.infer.mcmc.MCMCKernel, so that
kernels written for the reference plug in unchanged).

Only ``sample`` is mandatory.  HMC and NUTS of this package additionally run all chains as ONE batch on the
device (``chain_batched = True``); a kernel that does not say so -- every user-written one -- is driven chain
after chain through exactly the calls below, in this order::

    setup(warmup_steps, *model_args, **model_kwargs)      once per chain
    params = initial_params                               the starting point (a dict name -> tensor)
    params = sample(params)  x (warmup_steps + num_samples), with logging() polled for the progress line
    diagnostics()                                         merged into MCMC.diagnostics()
    cleanup()
"""
import abc


class MCMCKernel(abc.ABC):
    chain_batched = False

    # ---- the one thing a kernel must provide ------------------------------------------------------------
    @abc.abstractmethod
    def sample(self, params):
        """One transition from ``params`` (dict of tensors); returns the next state in the same form."""

    def __call__(self, params):
        return self.sample(params)

    # ---- the starting point: stored here unless a kernel computes its own ----------------------------------
    def _get_initial_params(self):
        raise NotImplementedError("{} does not define initial_params".format(type(self).__name__))

    def _set_initial_params(self, params):
        raise NotImplementedError("{} does not accept initial_params".format(type(self).__name__))

    initial_params = property(lambda self: self._get_initial_params(),
                              lambda self, params: self._set_initial_params(params))

    # ---- optional hooks, all no-ops by default ---------------------------------------------------------------
    def setup(self, warmup_steps, *args, **kwargs):
        """Before the first transition of a chain; receives the model's arguments."""

    def end_warmup(self):
        """Between the last warm-up transition and the first kept one."""

    def cleanup(self):
        """After the last transition of a chain."""

    def logging(self):
        """Progress information: an ordered mapping name -> already formatted value, or None."""
        return None

    def diagnostics(self):
        """Whatever the kernel wants reported once the run is over."""
        return {}
