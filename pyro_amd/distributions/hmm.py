"""DiscreteHMM: hidden Markov model with discrete latent state and arbitrary observation
distribution, time included in the event shape (reference: pyro/distributions/hmm.py:243-420).

The reference eliminates the time axis with a parallel-scan of log-space matrix products
(_sequential_logmatmulexp, O(log T) steps of [K, K] x [K, K] products and their autograd duals).
Here log_prob is ONE launch of the forward-backward kernel (pa_logchain_fwd_bwd): the initial state
is variable 0 of a chain of T + 1 variables, step t contributes the transition as the pairwise
potential and the observation log-likelihood as the unary potential of variable t + 1; the launch
returns the log-likelihood of every batch element and, as its gradient, the posterior marginals.
"""
import torch

from torch.distributions import constraints

from .base import TorchDistribution


class DiscreteHMM(TorchDistribution):
    arg_constraints = {"initial_logits": constraints.real, "transition_logits": constraints.real}
    has_rsample = False

    def __init__(self, initial_logits, transition_logits, observation_dist, validate_args=None,
                 duration=None):
        if initial_logits.dim() < 1:
            raise ValueError("expected initial_logits to have at least one dim, actual shape = {}"
                             .format(initial_logits.shape))
        if transition_logits.dim() < 2:
            raise ValueError("expected transition_logits to have at least two dims, actual shape = {}"
                             .format(transition_logits.shape))
        if len(observation_dist.batch_shape) < 1:
            raise ValueError("expected observation_dist to have at least one batch dim, actual "
                             ".batch_shape = {}".format(observation_dist.batch_shape))
        shape = torch.broadcast_shapes(initial_logits.shape[:-1] + (1,), transition_logits.shape[:-2],
                                       observation_dist.batch_shape[:-1])
        batch_shape, time_shape = shape[:-1], shape[-1:]
        if duration is not None and time_shape[0] not in (1, duration):
            raise ValueError("duration {} does not match the time axis of size {}".format(
                duration, time_shape[0]))
        event_shape = time_shape + observation_dist.event_shape
        self.initial_logits = initial_logits - initial_logits.logsumexp(-1, True)
        self.transition_logits = transition_logits - transition_logits.logsumexp(-1, True)
        self.observation_dist = observation_dist
        self.duration = duration
        super().__init__(torch.Size(batch_shape), torch.Size(event_shape), validate_args=validate_args)

    @constraints.dependent_property(event_dim=2)
    def support(self):
        return constraints.independent(self.observation_dist.support, 1)

    def expand(self, batch_shape, _instance=None):
        new = self._get_checked_instance(DiscreteHMM, _instance)
        batch_shape = torch.Size(torch.broadcast_shapes(self.batch_shape, tuple(batch_shape)))
        # the batch shape is the broadcast of all three inputs: expanding one of them is enough
        new.initial_logits = self.initial_logits.expand(batch_shape + (-1,))
        new.transition_logits = self.transition_logits
        new.observation_dist = self.observation_dist
        new.duration = self.duration
        super(DiscreteHMM, new).__init__(batch_shape, self.event_shape, validate_args=False)
        new._validate_args = self.__dict__.get("_validate_args")
        return new

    def _validate_sample(self, value):
        pass        # time may be longer than event_shape[0] for time-homogeneous parameters

    def log_prob(self, value):
        from .. import kernels
        from ..ops.contract import _LogChain
        obs = self.observation_dist
        value = value.unsqueeze(-1 - len(obs.event_shape))
        obs_logits = obs.log_prob(value)                              # [..., T, K]
        T, K = obs_logits.shape[-2:]
        trans, init = self.transition_logits, self.initial_logits
        if trans.dim() == 2:                       # [K, K]: shared by every step and batch element
            trans = trans.unsqueeze(0)
        batch = torch.broadcast_shapes(init.shape[:-1], trans.shape[:-3], obs_logits.shape[:-2])
        if trans.shape[-3] not in (1, T):
            raise ValueError("transition_logits has {} time steps, the data {}".format(
                trans.shape[-3], T))
        if not kernels.on_device(obs_logits):
            raise RuntimeError("pyro_amd: DiscreteHMM.log_prob needs device tensors (there is no "
                               "CPU implementation in this package)")
        fused = K <= 64 and obs_logits.dtype in (torch.float32, torch.float64)
        if not fused:
            # more than 64 states (or a dtype the kernel does not take): the forward recursion
            # step by step with device tensor ops
            a = init
            for t in range(T):
                tr = trans[..., t if trans.shape[-3] > 1 else 0, :, :]
                a = torch.logsumexp(a.unsqueeze(-1) + tr, dim=-2) + obs_logits[..., t, :]
            return torch.logsumexp(a, dim=-1)
        B = 1
        for n in batch:
            B *= int(n)
        unary = torch.cat([init.expand(batch + (K,)).unsqueeze(-2),
                           obs_logits.expand(batch + (T, K))], dim=-2).reshape(B, T + 1, K).contiguous()
        lead = trans.shape[:-3]
        if all(n == 1 for n in lead):                 # shared by the batch: [K, K] or [T, K, K]
            pair = trans.reshape(trans.shape[-3:])
            pair = pair[0].contiguous() if pair.shape[0] == 1 else pair.contiguous()
        else:
            pair = trans.expand(batch + (T, K, K)).reshape(B, T, K, K).contiguous()
        return _LogChain.invoke(unary, pair).reshape(batch)

    def sample(self, sample_shape=torch.Size()):
        raise NotImplementedError("DiscreteHMM.sample is not built in this backend (log_prob only)")
