"""The reparameteriser interface (role of pyro/infer/reparam/reparam.py) and the one mechanism both provided
strategies share: the site is re-expressed through an AUXILIARY site in other coordinates plus a bijection
back, and from then on carries no density of its own."""
import torch


class Reparam:
    """``apply(msg) -> msg`` over ``{"name", "fn", "value", "is_observed"}``.  ``args_kwargs`` holds the
    arguments of the reparameterised callable while it runs (``poutine.reparam`` used as a decorator)."""

    args_kwargs = None

    def apply(self, msg):
        raise NotImplementedError

    def __call__(self, name, fn, obs):
        # the (fn, obs) -> (fn, obs) calling convention older user code still has
        out = self.apply({"name": name, "fn": fn, "value": obs, "is_observed": obs is not None})
        return out["fn"], out["value"]

    # ---- helpers for strategies -------------------------------------------------------------------------
    @staticmethod
    def _unwrap(fn):
        """(innermost distribution, event_dim of the site) with every ``Independent`` layer peeled off."""
        site_event_dim = fn.event_dim
        inner = fn
        while isinstance(inner, torch.distributions.Independent):
            inner = inner.base_dist
        return inner, site_event_dim

    @staticmethod
    def _wrap(fn, event_dim):
        """``fn`` declared with ``event_dim`` event dims.  A plain torch.distributions object (the base of a
        torch TransformedDistribution) is first rebuilt as this package's class of the same name."""
        if not hasattr(fn, "to_event"):
            from ... import distributions as dist
            fn = getattr(dist, type(fn).__name__)(**{k: getattr(fn, k) for k in fn.arg_constraints})
        missing = event_dim - fn.event_dim
        assert missing >= 0
        return fn.to_event(missing) if missing else fn

    def _through_auxiliary(self, msg, aux_name, aux_fn, event_dim, to_aux, from_aux):
        """Sample (or, when the site's value is already known, score) the auxiliary site ``aux_name`` and
        return the message of the original site: a Delta at ``from_aux(aux value)`` that scores nothing.
        ``to_aux`` / ``from_aux`` are the two directions of the bijection."""
        from ... import distributions as dist
        from ...primitives import sample
        value = msg["value"]
        known = None if value is None else to_aux(value)
        aux = sample(aux_name, aux_fn, obs=known, infer={"is_observed": msg["is_observed"]})
        if value is None:
            value = from_aux(aux)
        return {"fn": dist.Delta(value, event_dim=event_dim).mask(False), "value": value,
                "is_observed": True}
