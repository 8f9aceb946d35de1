"""Step-size controller state for HMC / NUTS warm-up: Nesterov's primal-dual averaging as Stan and
the reference use it (pyro/ops/dual_averaging.py:5-85 states the recurrences).

The controller is written around the record the persistent NUTS kernel keeps per chain
(``pa_nuts_gaussian_transition``: ``[C, 5] = {x_avg, g_avg, t, prox_center, x_t}``): the host
object is that record plus its transition rule, so a warm-up window can run on the device and
hand the record back (``to_record`` / ``from_record``).  Fields hold python floats for one chain
or ``[C]`` tensors for vectorised chains -- one independent controller per chain, no host sync.
"""
import torch

RECORD_WIDTH = 5      # {x_avg, g_avg, t, prox_center, x_t}


def _mix(old, new, weight):
    """Convex combination (1 - weight) old + weight new, floats or tensors."""
    return (1.0 - weight) * old + weight * new


class DualAveraging:
    """Minimises E[g(x)] for a noisy statistic g (here: target accept probability minus the
    observed one, x = log step size).  ``step(g)`` feeds one observation; ``get_state()`` returns
    (the iterate to try next, the averaged iterate to keep at the end of the window)."""

    def __init__(self, prox_center=0, t0=10, kappa=0.75, gamma=0.05):
        self.prox_center = prox_center      # the point the iterates are pulled towards
        self.t0, self.kappa, self.gamma = t0, kappa, gamma
        self.reset()

    def reset(self):
        self.count = 0
        self.grad_mean = 0      # running mean of g with t0 pseudo-observations of 0 in front
        self.iterate = 0        # x_t
        self.average = 0        # the kappa-weighted mean of the iterates

    def step(self, g):
        self.count += 1
        n = self.count
        self.grad_mean = _mix(self.grad_mean, g, 1.0 / (n + self.t0))
        self.iterate = self.prox_center - self.grad_mean * (n ** 0.5 / self.gamma)
        self.average = _mix(self.average, self.iterate, float(n) ** (-self.kappa))

    def get_state(self):
        return self.iterate, self.average

    # ---- the kernel's record ----------------------------------------------------------------
    def to_record(self, chains, dtype, device):
        rec = torch.zeros((chains, RECORD_WIDTH), dtype=dtype, device=device)
        rec[:, 0] = self.average
        rec[:, 1] = self.grad_mean
        rec[:, 2] = float(self.count)
        rec[:, 3] = self.prox_center
        rec[:, 4] = self.iterate
        return rec

    def from_record(self, rec, steps_taken):
        """Adopt the record after the device advanced it by ``steps_taken`` observations."""
        self.average, self.grad_mean, self.iterate = rec[:, 0].clone(), rec[:, 1].clone(), rec[:, 4].clone()
        self.count += steps_taken
