"""TEST INFRASTRUCTURE -- velocity-Verlet restated from pyro/ops/integrator.py:14-65, for a flat
position vector, a diagonal inverse mass (BlockMassMatrix.kinetic_grad,
pyro/infer/mcmc/adaptation.py:328-347) and a potential given as a function returning
(potential_energy, gradient)."""
import numpy as np


def single_step_verlet(z, r, potential_and_grad, inv_mass, step_size, z_grads=None):
    """integrator.py:45-65."""
    if z_grads is None:
        _, z_grads = potential_and_grad(z)
    r = r + 0.5 * step_size * (-z_grads)          # r(n+1/2)
    v = inv_mass * r if np.ndim(inv_mass) < 2 else inv_mass @ r   # kinetic_grad (adaptation.py:328-347)
    z = z + step_size * v                         # z(n+1)
    pe, z_grads = potential_and_grad(z)
    r = r + 0.5 * step_size * (-z_grads)          # r(n+1)
    return z, r, z_grads, pe


def velocity_verlet(z, r, potential_and_grad, inv_mass, step_size, num_steps=1, z_grads=None):
    """integrator.py:14-42."""
    pe = None
    for _ in range(num_steps):
        z, r, z_grads, pe = single_step_verlet(z, r, potential_and_grad, inv_mass, step_size,
                                               z_grads)
    return z, r, z_grads, pe


def gaussian_potential(Lambda):
    """U(z) = 0.5 z^T Lambda z (BASELINE config 3); gradient as autograd gives it for a
    symmetric Lambda."""
    Lambda = np.asarray(Lambda)

    def fn(z):
        g = Lambda @ z
        return 0.5 * float(z @ g), g

    return fn
