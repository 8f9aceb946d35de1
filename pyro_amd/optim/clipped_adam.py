"""ClippedAdam as a torch ``Optimizer`` (the semantics of pyro.optim.clipped_adam.ClippedAdam): Adam on
gradients clamped element-wise to ``[-clip_norm, clip_norm]``, with each group's learning rate multiplied
by ``lrd`` at every step.  This per-parameter form serves ``pyro_amd.optim.ClippedAdam`` when it is given
callable arguments or ``clip_args``; the flat-buffer form (all parameters in one ``pa_adam_step`` launch)
does the same arithmetic.  Written over whole parameter lists with torch's ``_foreach`` operators: one
fused sequence per group instead of a Python loop per parameter."""
import math

import torch
from torch.optim.optimizer import Optimizer

_DEFAULTS = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, clip_norm=10.0, lrd=1.0)


class ClippedAdam(Optimizer):
    def __init__(self, params, **hyper):
        unknown = set(hyper) - set(_DEFAULTS)
        if unknown:
            raise TypeError("unexpected ClippedAdam arguments: {}".format(sorted(unknown)))
        super().__init__(params, {**_DEFAULTS, **hyper})

    def _moments(self, p):
        slot = self.state[p]
        if not slot:
            slot.update(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
        return slot

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            group["lr"] *= group["lrd"]                     # the k-th step runs at lr * lrd ** k
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            b1, b2 = group["betas"]
            bound = group["clip_norm"]
            grads = [p.grad.clamp(-bound, bound) for p in live]
            if group["weight_decay"]:
                torch._foreach_add_(grads, live, alpha=group["weight_decay"])
            slots = [self._moments(p) for p in live]
            first = [s["exp_avg"] for s in slots]
            second = [s["exp_avg_sq"] for s in slots]
            torch._foreach_mul_(first, b1)
            torch._foreach_add_(first, grads, alpha=1 - b1)
            torch._foreach_mul_(second, b2)
            torch._foreach_addcmul_(second, grads, grads, value=1 - b2)
            denominators = torch._foreach_sqrt(second)
            torch._foreach_add_(denominators, group["eps"])
            for p, s, m, d in zip(live, slots, first, denominators):
                s["step"] += 1                              # parameters may have joined at different times
                k = s["step"]
                p.addcdiv_(m, d, value=-group["lr"] * math.sqrt(1 - b2 ** k) / (1 - b1 ** k))
        return loss
