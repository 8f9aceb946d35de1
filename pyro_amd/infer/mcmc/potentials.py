"""Potential-energy callables with closed-form gradients.

A potential in the reference is ``potential_fn(dict[str, Tensor]) -> scalar Tensor``
differentiated by autograd on every leapfrog step (pyro/ops/integrator.py:68-94).  The same
callables work here; a potential may additionally provide

    potential_and_grad(z_flat[C, D]) -> (pe[C], grad[C, D])

to bypass autograd, and the NUTS kernel recognises ``GaussianPotential`` and runs the whole
transition in one fused HIP launch (pa_nuts_gaussian_transition).
"""
import torch


class GaussianPotential:
    """U(z) = 0.5 z^T Lambda z for one site (BASELINE config 3).

    Callable in the reference's style -- ``potential({"x": z})`` with ``z`` of shape [D] or
    chain-batched [C, D] -- and carrying the precision matrix for the fused kernel."""

    def __init__(self, precision, site="x"):
        P = 0.5 * (precision + precision.t())   # the kernels read Lambda column-wise
        self.precision = P.contiguous()
        self.site = site

    def __call__(self, z):
        x = z[self.site]
        g = x @ self.precision
        return 0.5 * (x * g).sum(-1)

    def potential_and_grad(self, z_flat):
        g = z_flat @ self.precision
        return 0.5 * (z_flat * g).sum(-1), g
