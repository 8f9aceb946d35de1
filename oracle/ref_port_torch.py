"""TEST INFRASTRUCTURE / cpu_baseline -- the reference's config-2 SVI step restated with the
SAME torch CPU operators the reference executes, minus its Python effect-handler overhead.

The reference's arithmetic on this path *is* third-party torch (torch.distributions on ATen CPU
kernels, autograd, torch.optim.Adam): Normal.rsample / log_prob (torch: normal.py:83-103),
Bernoulli(logits).log_prob = -binary_cross_entropy_with_logits (bernoulli.py:121-125),
Independent sums (independent.py:119-121), scale_and_mask (pyro/distributions/util.py:311-328),
the surrogate of pyro/infer/trace_elbo.py:82-159 and SVI.step (pyro/infer/svi.py:134-162).
Because the reference (pure Python) cannot travel to the GPU box, bench.py times THIS port on
the box's host cores as the ``cpu_baseline`` ("kind": "port").  It is pinned against the golden
loss/gradients of the unmodified reference in tests/test_oracle_vs_golden.py; being free of the
handler overhead it is, if anything, faster than the real reference (a conservative baseline).
"""
import torch
import torch.distributions as td
import torch.nn.functional as F


class LogRegAutoNormalPort:
    def __init__(self, X, y, num_particles, init_scale=0.1, lr=0.01):
        self.X, self.y, self.P = X, y, num_particles
        D = X.shape[1]
        dt = X.dtype
        self.loc_w = torch.zeros(D, dtype=dt, requires_grad=True)
        self.loc_b = torch.zeros((), dtype=dt, requires_grad=True)
        rho = torch.tensor(init_scale, dtype=dt)
        rho = rho + torch.log(-torch.expm1(-rho))  # softplus^-1 (AutoNormal.scale_constraint)
        self.rho_w = rho.expand(D).clone().requires_grad_(True)
        self.rho_b = rho.clone().requires_grad_(True)
        self.params = [self.loc_w, self.rho_w, self.loc_b, self.rho_b]
        # the reference keeps one torch optimizer per parameter (pyro/optim/optim.py:117-155)
        self.optims = [torch.optim.Adam([p], lr=lr) for p in self.params]

    def loss_and_grads(self, eps_w=None, eps_b=None):
        X, y, P = self.X, self.y, self.P
        D = X.shape[1]
        s_w, s_b = F.softplus(self.rho_w), F.softplus(self.rho_b)
        qw = td.Independent(td.Normal(self.loc_w, s_w).expand([P, 1, D]), 1)
        qb = td.Normal(self.loc_b, s_b).expand([P, 1])
        if eps_w is None:
            w, b = qw.rsample(), qb.rsample()
        else:
            w = self.loc_w + s_w * eps_w.reshape(P, 1, D)
            b = self.loc_b + s_b * eps_b.reshape(P, 1)
        log_q = qw.log_prob(w).sum() + qb.log_prob(b).sum()
        log_p = td.Independent(td.Normal(torch.zeros(D, dtype=X.dtype), 1.0), 1).log_prob(w).sum() \
            + td.Normal(torch.zeros((), dtype=X.dtype), 1.0).log_prob(b).sum()
        logits = (w @ X.t()).squeeze(-2) + b          # [P, N]
        ll = td.Bernoulli(logits=logits).log_prob(y).sum()
        surrogate = -(log_p + ll - log_q) / P
        surrogate.backward()
        return float(surrogate.detach())

    def step(self):
        loss = self.loss_and_grads()
        for o in self.optims:
            o.step()
        for p in self.params:  # zero_grads re-allocates zeros (pyro/infer/util.py:85-91)
            p.grad = torch.zeros_like(p)
        return loss
