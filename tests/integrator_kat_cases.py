"""The reference's integrator known-answer tests (tests/ops/test_integrator.py:40-180: harmonic
oscillator, circular planetary motion, quartic oscillator; trajectory end points, energy
conservation, time reversibility) restated against pyro_amd.ops.integrator.velocity_verlet, with
the kinetic gradient given either as a plain callable (generic path) or as an object exposing its
diagonal inverse mass (the HIP leapfrog kernels)."""
import math

import numpy as np
import torch

from pyro_amd.ops.integrator import velocity_verlet

SYSTEMS = {
    # name: (potential, energy, step_size, num_steps, q_i, p_i, q_f, p_f, prec)
    "harmonic": (lambda q: 0.5 * q["x"] ** 2,
                 lambda q, p: 0.5 * p["x"] ** 2 + 0.5 * q["x"] ** 2,
                 0.01, 100, {"x": [0.0]}, {"x": [1.0]}, {"x": [math.sin(1.0)]}, {"x": [math.cos(1.0)]},
                 1e-4),
    "circular": (lambda q: -1.0 / torch.pow(q["x"] ** 2 + q["y"] ** 2, 0.5),
                 lambda q, p: 0.5 * p["x"] ** 2 + 0.5 * p["y"] ** 2
                 - 1.0 / torch.pow(q["x"] ** 2 + q["y"] ** 2, 0.5),
                 0.01, 628, {"x": [1.0], "y": [0.0]}, {"x": [0.0], "y": [1.0]},
                 {"x": [1.0], "y": [0.0]}, {"x": [0.0], "y": [1.0]}, 5e-3),
    "quartic": (lambda q: 0.25 * torch.pow(q["x"], 4.0),
                lambda q, p: 0.5 * p["x"] ** 2 + 0.25 * torch.pow(q["x"], 4.0),
                0.1, 1810, {"x": [0.02]}, {"x": [0.0]}, {"x": [-0.02]}, {"x": [0.0]}, 1e-4),
}


class UnitMass:
    """kinetic_grad(p) = p, announcing its diagonal inverse mass so that the drift runs in
    pa_leapfrog_kick_drift."""

    def __init__(self, device, dtype):
        self.one = torch.ones(1, dtype=dtype, device=device)

    def __call__(self, p):
        return dict(p)

    def inverse_mass_diag(self, site):
        return self.one


def _dev(d, device, dtype):
    return {k: torch.tensor(v, dtype=dtype, device=device) for k, v in d.items()}


def run_system(name, device, kernel_path, dtype=torch.float64):
    pot, energy, eps, n, q_i, p_i, q_f, p_f, prec = SYSTEMS[name]
    kg = UnitMass(device, dtype) if kernel_path else (lambda p: dict(p))
    potential = lambda q: pot(q).sum()      # noqa: E731
    qi, pi = _dev(q_i, device, dtype), _dev(p_i, device, dtype)
    qf, pf, _, _ = velocity_verlet(qi, pi, potential, kg, eps, n)
    for k in q_f:                                                     # test_trajectory
        np.testing.assert_allclose(qf[k].cpu().numpy(), q_f[k], atol=prec)
        np.testing.assert_allclose(pf[k].cpu().numpy(), p_f[k], atol=prec)
    e0, e1 = energy(qi, pi), energy(qf, pf)                           # test_energy_conservation
    np.testing.assert_allclose(e1.cpu().numpy(), e0.cpu().numpy(), atol=1e-5, rtol=1e-5)
    p_rev = {k: -v for k, v in pf.items()}                            # test_time_reversibility
    qb, _, _, _ = velocity_verlet(qf, p_rev, potential, kg, eps, n)
    for k in q_i:
        np.testing.assert_allclose(qb[k].cpu().numpy(), q_i[k], atol=1e-5)
