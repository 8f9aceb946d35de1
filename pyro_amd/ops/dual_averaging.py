"""Dual averaging for step-size adaptation (reference: pyro/ops/dual_averaging.py:5-85).

Identical recurrences; the state may be python floats (one chain, as the reference) or tensors
of shape [C] (one independent scheme per vectorised chain, updated on the device with no host
synchronisation)."""


class DualAveraging:
    def __init__(self, prox_center=0, t0=10, kappa=0.75, gamma=0.05):
        self.prox_center = prox_center
        self.t0 = t0
        self.kappa = kappa
        self.gamma = gamma
        self.reset()

    def reset(self):
        self._x_avg = 0   # average of the primal sequence
        self._g_avg = 0   # average of the dual sequence
        self._t = 0

    def step(self, g):
        self._t += 1
        self._g_avg = (1 - 1 / (self._t + self.t0)) * self._g_avg + g / (self._t + self.t0)
        self._x_t = self.prox_center - (self._t ** 0.5) / self.gamma * self._g_avg
        weight_t = self._t ** (-self.kappa)
        self._x_avg = (1 - weight_t) * self._x_avg + weight_t * self._x_t

    def get_state(self):
        return self._x_t, self._x_avg
