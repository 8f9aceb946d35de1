"""Velocity-Verlet integrator (reference seam 3(ii): pyro/ops/integrator.py:14-94).

Same function-level contract -- dicts keyed by site name in, ``(z_next, r_next, z_grads,
potential_energy)`` out -- with the two momentum/position updates done by the HIP kernels
pa_leapfrog_kick_drift / pa_leapfrog_kick when the kinetic gradient is a diagonal mass matrix
(``kinetic_grad`` exposes ``inverse_mass_diag(site)``; BlockMassMatrix does), and by torch
element-wise ops for any other ``kinetic_grad`` callable.  Tensors may carry a leading chain
dim: ``step_size`` is then a per-chain tensor.
"""
import torch
from torch.autograd import grad

from .. import kernels

_EXCEPTION_HANDLERS = {}


def register_exception_handler(name, handler, warn_on_overwrite=True):
    """reference: pyro/ops/integrator.py:97-117."""
    if name in _EXCEPTION_HANDLERS and warn_on_overwrite:
        import warnings
        warnings.warn("Overwriting exception handler '{}'".format(name), RuntimeWarning)
    _EXCEPTION_HANDLERS[name] = handler


def _handle_linalg(exception):
    return type(exception) is RuntimeError and "singular" in str(exception)


register_exception_handler("torch_singular", _handle_linalg, warn_on_overwrite=False)


def potential_grad(potential_fn, z):
    """(z_grads, potential_energy) by autograd (reference: integrator.py:68-94).  A potential
    that returns one energy per chain ([C]) is differentiated through its sum: chains are
    independent, so each row of the gradient is that chain's own gradient."""
    z_keys, z_nodes = zip(*z.items())
    for node in z_nodes:
        node.requires_grad_(True)
    try:
        potential_energy = potential_fn(z)
    except Exception as e:  # noqa: BLE001
        if any(h(e) for h in _EXCEPTION_HANDLERS.values()):
            grads = {k: v.new_zeros(v.shape) for k, v in z.items()}
            for node in z_nodes:
                node.requires_grad_(False)
            return grads, z_nodes[0].new_tensor(float("nan"))
        raise
    total = potential_energy if potential_energy.dim() == 0 else potential_energy.sum()
    grads = grad(total, z_nodes)
    for node in z_nodes:
        node.requires_grad_(False)
    return dict(zip(z_keys, grads)), potential_energy.detach()


def _as_2d(t, chain_batched):
    return t.reshape(t.shape[0], -1) if chain_batched else t.reshape(1, -1)


def _step_tensor(step_size, like, chain_batched):
    if isinstance(step_size, torch.Tensor):
        s = step_size.to(like.dtype)
        return s.reshape(-1).contiguous() if chain_batched else s.reshape(1).contiguous()
    return torch.full((1,), float(step_size), dtype=like.dtype, device=like.device)


def _single_step_verlet(z, r, potential_fn, kinetic_grad, step_size, z_grads=None):
    """One leapfrog step; modifies the z / r dicts in place (reference: integrator.py:45-65)."""
    z_grads = potential_grad(potential_fn, z)[0] if z_grads is None else z_grads
    diag = getattr(kinetic_grad, "inverse_mass_diag", None)
    on_gpu = all(v.is_cuda for v in z.values())
    if diag is not None and on_gpu:
        chain_batched = isinstance(step_size, torch.Tensor) and step_size.dim() == 1
        steps = {}
        for name in z:
            zz = _as_2d(z[name].contiguous().clone(), chain_batched)
            rr = _as_2d(r[name].contiguous().clone(), chain_batched)
            gg = _as_2d(z_grads[name].contiguous(), chain_batched)
            st = _step_tensor(step_size, zz, chain_batched)
            im = diag(name).to(zz.dtype).contiguous()
            im = im.reshape(zz.shape) if im.numel() == zz.numel() and zz.shape[0] > 1 \
                else im.reshape(-1)
            kernels.leapfrog_kick_drift(zz, rr, gg, im, st)
            z[name] = zz.reshape(z[name].shape)
            r[name] = rr.reshape(r[name].shape)
            steps[name] = st
        z_grads, potential_energy = potential_grad(potential_fn, z)
        for name in r:
            rr = _as_2d(r[name], chain_batched)
            kernels.leapfrog_kick(rr, _as_2d(z_grads[name].contiguous(), chain_batched),
                                  steps[name])
        return z, r, z_grads, potential_energy
    # generic kinetic_grad (dense / structured mass, CPU tensors in tests of the host logic)
    step = step_size
    if isinstance(step_size, torch.Tensor) and step_size.dim() == 1:
        step = None
    for name in r:
        s = step if step is not None else step_size.reshape((-1,) + (1,) * (r[name].dim() - 1))
        r[name] = r[name] + 0.5 * s * (-z_grads[name])
    r_grads = kinetic_grad(r)
    for name in z:
        s = step if step is not None else step_size.reshape((-1,) + (1,) * (z[name].dim() - 1))
        z[name] = z[name] + s * r_grads[name]
    z_grads, potential_energy = potential_grad(potential_fn, z)
    for name in r:
        s = step if step is not None else step_size.reshape((-1,) + (1,) * (r[name].dim() - 1))
        r[name] = r[name] + 0.5 * s * (-z_grads[name])
    return z, r, z_grads, potential_energy


def velocity_verlet(z, r, potential_fn, kinetic_grad, step_size, num_steps=1, z_grads=None):
    """Second-order symplectic integrator (reference: integrator.py:14-42)."""
    z_next = z.copy()
    r_next = r.copy()
    potential_energy = None
    for _ in range(num_steps):
        z_next, r_next, z_grads, potential_energy = _single_step_verlet(
            z_next, r_next, potential_fn, kinetic_grad, step_size, z_grads)
    return z_next, r_next, z_grads, potential_energy
