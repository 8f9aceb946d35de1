"""A latent ``TransformedDistribution`` sampled in the coordinates of its base distribution (role of
pyro/infer/reparam/transform.py): useful when the posterior is simple before the transforms."""
import functools

import torch

from .reparam import Reparam


def _compose(transforms):
    return lambda x: functools.reduce(lambda acc, t: t(acc), transforms, x)


class TransformReparam(Reparam):
    """``<name>_base ~ base_dist`` is the auxiliary site; the site itself becomes the image of that draw under
    the distribution's transforms.  For latent sites only."""

    def apply(self, msg):
        td, event_dim = self._unwrap(msg["fn"])
        assert isinstance(td, torch.distributions.TransformedDistribution), type(td)
        forward = list(td.transforms)
        backward = [t.inv for t in reversed(forward)]
        # every transform may change how many rightmost dims form one event
        base_event_dim = event_dim + sum(t.domain.event_dim - t.codomain.event_dim for t in forward)
        return self._through_auxiliary(msg, msg["name"] + "_base", self._wrap(td.base_dist, base_event_dim),
                                       event_dim, _compose(backward), _compose(forward))
