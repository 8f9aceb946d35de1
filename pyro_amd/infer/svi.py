"""SVI: the drop-in training-step API (reference: pyro/infer/svi.py:38-162)."""
import warnings

import torch

from .. import poutine
from ..params import _PARAM_STORE
from ..util import torch_isnan, zero_grads
from .elbo import ELBO


class SVI:
    def __init__(self, model, guide, optim, loss, loss_and_grads=None, num_samples=0, num_steps=0,
                 **kwargs):
        if num_steps or num_samples:
            warnings.warn("num_steps / num_samples are ignored (TracePosterior is not part of "
                          "this backend)")
        self.model, self.guide, self.optim = model, guide, optim
        if isinstance(loss, ELBO):
            self.loss = loss.loss
            self.loss_and_grads = loss.loss_and_grads
        else:
            if loss_and_grads is None:
                def _loss_and_grads(model, guide, *args, **kwargs):
                    loss_val = loss(model, guide, *args, **kwargs)
                    if getattr(loss_val, "requires_grad", False):
                        loss_val.backward(retain_graph=True)
                    return loss_val

                loss_and_grads = _loss_and_grads
            self.loss, self.loss_and_grads = loss, loss_and_grads

    def evaluate_loss(self, *args, **kwargs):
        with torch.no_grad():
            loss = self.loss(self.model, self.guide, *args, **kwargs)
            return loss.item() if isinstance(loss, torch.Tensor) else loss

    def step(self, *args, **kwargs):
        """One gradient step: loss_and_grads, optimizer update on every touched param, zero grads."""
        with poutine.trace(param_only=True) as param_capture:
            loss = self.loss_and_grads(self.model, self.guide, *args, **kwargs)
        params = set(site["value"].unconstrained() if hasattr(site["value"], "unconstrained")
                     else getattr(site["value"], "_pyro_unconstrained_param", site["value"])
                     for site in param_capture.trace.nodes.values())
        self.optim(params)
        if not getattr(self.optim, "zeroes_grads", False):
            zero_grads(params)
        if isinstance(loss, tuple):
            return type(loss)(map(lambda x: x.item() if isinstance(x, torch.Tensor) else x, loss))
        return loss.item() if isinstance(loss, torch.Tensor) else loss
