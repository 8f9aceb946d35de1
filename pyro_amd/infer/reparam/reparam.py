"""The reparameteriser interface (reference: pyro/infer/reparam/reparam.py)."""
import torch


class Reparam:
    """``apply(msg) -> msg`` over ``{"name", "fn", "value", "is_observed"}``; may call ``pyro.sample``
    for the auxiliary sites it introduces."""

    def apply(self, msg):
        raise NotImplementedError

    def __call__(self, name, fn, obs):          # the pre-1.7 interface of the reference
        msg = self.apply({"name": name, "fn": fn, "value": obs, "is_observed": obs is not None})
        return msg["fn"], msg["value"]

    @staticmethod
    def _unwrap(fn):
        event_dim = fn.event_dim
        while isinstance(fn, torch.distributions.Independent):
            fn = fn.base_dist
        return fn, event_dim

    @staticmethod
    def _wrap(fn, event_dim):
        if not hasattr(fn, "to_event"):
            # a plain torch.distributions object (the base of a torch TransformedDistribution):
            # the same distribution as this package's class of that name
            from ... import distributions as dist
            cls = getattr(dist, type(fn).__name__)
            fn = cls(**{k: getattr(fn, k) for k in fn.arg_constraints})
        if fn.event_dim < event_dim:
            fn = fn.to_event(event_dim - fn.event_dim)
        assert fn.event_dim == event_dim
        return fn
