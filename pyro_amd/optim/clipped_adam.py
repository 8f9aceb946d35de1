"""ClippedAdam as a torch ``Optimizer`` (the semantics of pyro/optim/clipped_adam.py:14-100): Adam
with every gradient element clamped to [-clip_norm, clip_norm] and the learning rate of each group
multiplied by ``lrd`` after every step.  Used per parameter (``pyro_amd.optim.ClippedAdam`` with
callable arguments); the fused flat-buffer form is pa_adam_step with the same arithmetic."""
import math

import torch
from torch.optim.optimizer import Optimizer


class ClippedAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, clip_norm=10.0,
                 lrd=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      clip_norm=clip_norm, lrd=lrd))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            group["lr"] *= group["lrd"]                 # decay first: step k uses lr * lrd^k
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad.clamp(-group["clip_norm"], group["clip_norm"])
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(grad)
                    state["exp_avg_sq"] = torch.zeros_like(grad)
                state["step"] += 1
                if group["weight_decay"] != 0:
                    grad = grad.add(p, alpha=group["weight_decay"])
                m, v = state["exp_avg"], state["exp_avg_sq"]
                m.mul_(beta1).add_(grad, alpha=1 - beta1)
                v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
                k = state["step"]
                step_size = group["lr"] * math.sqrt(1 - beta2 ** k) / (1 - beta1 ** k)
                p.addcdiv_(m, v.sqrt().add_(group["eps"]), value=-step_size)
        return loss
