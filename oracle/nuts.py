"""TEST INFRASTRUCTURE -- NUTS transition restated from pyro/infer/mcmc/nuts.py (recursive
formulation, exactly as the reference: _build_basetree :197-248, _build_tree :250-365,
_is_turning :184-195, sample :367-522, _logaddexp :15-17) for a flat position vector with a
diagonal (inv_mass[D]) or dense (inv_mass[D, D], full_mass=True) mass matrix
(pyro/infer/mcmc/adaptation.py:238-392; hmc.py:152-156 kinetic energy, :231-248 momentum draw).

Randomness is abstracted behind a ``draws`` object so the same code can be
 * pinned against the unmodified reference (SequenceDraws: the reference's own pyro.sample
   calls are intercepted and answered from the same sequence, tests/golden/make_golden.py), and
 * compared with the HIP kernel (KeyedDraws: Philox draws keyed by tree position, the
   contract documented in pyro_amd/csrc/nuts.hip).
"""
import math
from collections import namedtuple

import numpy as np

from . import philox
from .integrator import single_step_verlet

MAX_SLICED_ENERGY = 1000.0  # nuts.py:182

Tree = namedtuple("Tree", [
    "z_left", "r_left", "r_left_unscaled", "z_left_grads",
    "z_right", "r_right", "r_right_unscaled", "z_right_grads",
    "z_proposal", "z_proposal_pe", "z_proposal_grads",
    "r_sum", "weight", "turning", "diverging", "sum_accept_probs", "num_proposals"])


def logaddexp(x, y):
    """nuts.py:15-17."""
    mn, mx = (x, y) if x < y else (y, x)
    return math.log1p(math.exp(mn - mx)) + mx if mx > -math.inf else mx


class KeyedDraws:
    """Philox draws keyed by (seed, chain, t, slot); see the slot table in nuts.hip."""

    def __init__(self, seed, chain, t, dtype):
        self.seed, self.chain, self.dtype = seed, chain, np.dtype(dtype).type
        self.base = t << 20

    def momentum(self, D):
        return philox.normal(D, self.dtype, self.seed, self.base, self.chain).astype(self.dtype)

    def _u(self, slot, second=False):
        b = philox.block(self.seed, self.base + slot, self.chain)
        return float(philox.unit_from_block(b, self.dtype, second))

    def slice_exp(self):
        return -math.log(self._u(1024))

    def direction(self, j):
        return self._u(1025 + j)

    def accept(self, j):
        return self._u(1025 + j, second=True)

    def merge(self, j, k, m):
        return self._u(2048 + (1 << j) + (1 << (j - k)) + m)


class SequenceDraws:
    """Answers draws from explicit sequences, in the reference's call order."""

    def __init__(self, momentum, uniforms, slice_exps=()):
        self._mom = momentum
        self._u = iter(uniforms)
        self._s = iter(slice_exps)

    def momentum(self, D):
        return np.asarray(self._mom)

    def slice_exp(self):
        return float(next(self._s))

    def direction(self, j):
        return float(next(self._u))

    def accept(self, j):
        return float(next(self._u))

    def merge(self, j, k, m):
        return float(next(self._u))


class _Nuts:
    def __init__(self, potential_and_grad, inv_mass, step_size, draws, multinomial, dtype):
        self.pg = potential_and_grad
        self.dtype = np.dtype(dtype).type
        self.v = np.asarray(inv_mass, dtype=self.dtype)
        # BlockMassMatrix.inverse_mass_matrix.setter (adaptation.py:270-282)
        if self.v.ndim == 1:
            self.sqrt_inv = np.sqrt(self.v)                   # mass_matrix_sqrt_inverse
            self.sqrt = (self.dtype(1) / self.sqrt_inv)       # mass_matrix_sqrt
        else:
            # dense: sqrt_inverse = cholesky(inverse mass)^T (upper), sqrt = its triangular inverse
            self.sqrt_inv = np.linalg.cholesky(self.v).T.astype(self.dtype)
            self.sqrt = np.linalg.inv(self.sqrt_inv).astype(self.dtype)
        self.eps = self.dtype(step_size)
        self.draws = draws
        self.multinomial = multinomial
        self.n_grad_evals = 0

    def scale(self, r_unscaled):
        """adaptation.py:349-373."""
        return self.sqrt * r_unscaled if self.v.ndim == 1 else self.sqrt @ r_unscaled

    def unscale(self, r):
        """adaptation.py:375-392."""
        return self.sqrt_inv * r if self.v.ndim == 1 else self.sqrt_inv @ r

    def kinetic(self, r_unscaled):
        return self.dtype(0.5) * self.dtype(r_unscaled.dot(r_unscaled))   # hmc.py:152-156

    def is_turning(self, r_left_unscaled, r_right_unscaled, r_sum):
        """nuts.py:184-195."""
        rho = r_sum - (r_left_unscaled + r_right_unscaled) / 2
        return bool(r_left_unscaled.dot(rho) <= 0) or bool(r_right_unscaled.dot(rho) <= 0)

    def basetree(self, z, r, z_grads, log_slice, direction, energy_current):
        """nuts.py:197-248."""
        step = self.eps if direction == 1 else -self.eps
        z_new, r_new, z_grads, pe = single_step_verlet(z, r, self.pg, self.v, step, z_grads)
        self.n_grad_evals += 1
        r_new_unscaled = self.unscale(r_new)
        energy_new = self.dtype(pe) + self.kinetic(r_new_unscaled)
        if math.isnan(energy_new):
            energy_new = self.dtype(math.inf)
        sliced = energy_new + log_slice
        diverging = bool(sliced > MAX_SLICED_ENERGY)
        delta = energy_new - energy_current
        with np.errstate(over="ignore"):
            accept_prob = min(float(np.exp(-delta)), 1.0)
        if self.multinomial:
            weight = -sliced
        else:
            weight = 1.0 if sliced <= 0 else 0.0
        return Tree(z_new, r_new, r_new_unscaled, z_grads, z_new, r_new, r_new_unscaled, z_grads,
                    z_new, pe, z_grads, r_new_unscaled, weight, False, diverging, accept_prob, 1)

    def build_tree(self, z, r, z_grads, log_slice, direction, depth, energy_current, j, offset):
        """nuts.py:250-365. (j, offset) locate the subtree inside doubling j for keyed draws."""
        if depth == 0:
            return self.basetree(z, r, z_grads, log_slice, direction, energy_current)
        half = self.build_tree(z, r, z_grads, log_slice, direction, depth - 1, energy_current, j,
                               offset)
        z_prop, z_prop_pe, z_prop_grads = half.z_proposal, half.z_proposal_pe, \
            half.z_proposal_grads
        if half.turning or half.diverging:
            return half
        if direction == 1:
            z, r, z_grads = half.z_right, half.r_right, half.z_right_grads
        else:
            z, r, z_grads = half.z_left, half.r_left, half.z_left_grads
        other = self.build_tree(z, r, z_grads, log_slice, direction, depth - 1, energy_current, j,
                                offset + (1 << (depth - 1)))
        if self.multinomial:
            weight = logaddexp(half.weight, other.weight)
        else:
            weight = half.weight + other.weight
        sum_accept = half.sum_accept_probs + other.sum_accept_probs
        num_prop = half.num_proposals + other.num_proposals
        r_sum = half.r_sum + other.r_sum
        if self.multinomial:
            other_prob = math.exp(other.weight - weight) if weight > -math.inf else float("nan")
        else:
            other_prob = other.weight / weight if weight > 0 else 0.0
        u = self.draws.merge(j, depth, offset >> depth)
        if u < other_prob:   # Bernoulli(probs).sample() == 1  <=>  rand < probs
            z_prop, z_prop_pe, z_prop_grads = other.z_proposal, other.z_proposal_pe, \
                other.z_proposal_grads
        if direction == 1:
            left, right = half, other
        else:
            left, right = other, half
        turning = other.turning or self.is_turning(left.r_left_unscaled, right.r_right_unscaled,
                                                   r_sum)
        return Tree(left.z_left, left.r_left, left.r_left_unscaled, left.z_left_grads,
                    right.z_right, right.r_right, right.r_right_unscaled, right.z_right_grads,
                    z_prop, z_prop_pe, z_prop_grads, r_sum, weight, turning, other.diverging,
                    sum_accept, num_prop)


def nuts_transition(z, pe, z_grads, potential_and_grad, inv_mass, step_size, draws,
                    max_tree_depth=10, multinomial=True, dtype=np.float64):
    """nuts.py:367-522 for one chain. Returns a dict with the next state and statistics."""
    n = _Nuts(potential_and_grad, inv_mass, step_size, draws, multinomial, dtype)
    dt = n.dtype
    z = np.asarray(z, dtype=dt)
    z_grads = np.asarray(z_grads, dtype=dt)
    r_unscaled = np.asarray(draws.momentum(z.shape[0]), dtype=dt)
    r = n.scale(r_unscaled)                                   # adaptation.py:349-373 scale()
    energy_current = n.kinetic(r_unscaled) + dt(pe)
    if multinomial:
        log_slice = -energy_current
    else:
        log_slice = -energy_current - dt(draws.slice_exp())
    z_left = z_right = z
    r_left = r_right = r
    r_left_u = r_right_u = r_unscaled
    g_left = g_right = z_grads
    accepted = False
    diverged = False
    r_sum = r_unscaled
    sum_accept, num_prop = 0.0, 0
    tree_weight = 0.0 if multinomial else 1.0
    tree_depth = 0
    pe_out = pe
    while tree_depth < max_tree_depth:
        j = tree_depth
        direction = 1 if draws.direction(j) < 0.5 else -1    # Bernoulli(0.5).sample()
        if direction == 1:
            new = n.build_tree(z_right, r_right, g_right, log_slice, direction, tree_depth,
                               energy_current, j, 0)
            z_right, r_right, r_right_u, g_right = new.z_right, new.r_right, \
                new.r_right_unscaled, new.z_right_grads
        else:
            new = n.build_tree(z_left, r_left, g_left, log_slice, direction, tree_depth,
                               energy_current, j, 0)
            z_left, r_left, r_left_u, g_left = new.z_left, new.r_left, new.r_left_unscaled, \
                new.z_left_grads
        sum_accept += new.sum_accept_probs
        num_prop += new.num_proposals
        if new.diverging:
            diverged = True
            break
        if new.turning:
            break
        tree_depth += 1
        if multinomial:
            new_prob = math.exp(new.weight - tree_weight)
        else:
            new_prob = new.weight / tree_weight
        rand = draws.accept(j)
        if rand < new_prob:
            accepted = True
            z, z_grads, pe_out = new.z_proposal, new.z_proposal_grads, new.z_proposal_pe
        r_sum = r_sum + new.r_sum
        if n.is_turning(r_left_u, r_right_u, r_sum):
            break
        if multinomial:
            tree_weight = logaddexp(tree_weight, new.weight)
        else:
            tree_weight = tree_weight + new.weight
    return {"z": z, "pe": pe_out, "grad": z_grads, "accept_prob": sum_accept / num_prop,
            "n_leapfrog": num_prop, "depth": tree_depth, "diverging": diverged,
            "accepted": accepted}
