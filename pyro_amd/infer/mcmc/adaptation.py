"""Warm-up adaptation of step size and mass matrix, vectorised over chains
(reference: pyro/infer/mcmc/adaptation.py:23-202 WarmupAdapter, :238-392 BlockMassMatrix).

Every chain runs ITS OWN Stan-style scheme exactly as a single reference chain would -- its own
dual-averaging state, its own Welford estimator, its own step size and diagonal inverse mass --
held as tensors with a leading chain dim and updated on the device without host round trips
(the reference keeps python floats and synchronises once per transition, adaptation.py:176).
Statistics are NOT pooled across chains (that would be a deviation from the reference).
"""
import math
from collections import namedtuple

import torch

from ... import kernels
from ...ops.dual_averaging import DualAveraging
from ...ops.welford import WelfordCovariance

adapt_window = namedtuple("adapt_window", ["start", "end"])


def build_adaptation_schedule(warmup_steps, start_buffer=75, end_buffer=50, initial_window=25):
    """Stan's windowed warm-up (reference: adaptation.py:65-103): a fast window for the step size only,
    slow windows for the mass matrix that double in length while at least three of the current length
    still fit before the closing fast window (the last slow window takes whatever is left), and a closing
    fast window.  Returns inclusive (start, end) pairs that tile 0 .. warmup_steps - 1."""
    W = int(warmup_steps)
    if W < 20:                                   # too short to split: one window
        return [adapt_window(0, W - 1)]
    head, tail, width = start_buffer, end_buffer, initial_window
    if head + tail + width > W:                  # the defaults do not fit: 15 % / 75 % / 10 %
        head, tail = int(0.15 * W), int(0.1 * W)
        width = W - head - tail
    slow_end = W - tail                          # first transition of the closing window
    cuts = [0, head]                             # window boundaries (each the start of a window)
    while cuts[-1] < slow_end:
        left = slow_end - cuts[-1]
        if 3 * width <= left:
            cuts.append(cuts[-1] + width)
            width *= 2
        else:
            cuts.append(slow_end)
    cuts.append(W)
    return [adapt_window(a, b - 1) for a, b in zip(cuts[:-1], cuts[1:])]


class DiagMassMatrix:
    """Per-chain diagonal mass matrix: inverse mass v[C, D] and the derived square roots
    (reference: BlockMassMatrix with one diagonal block, adaptation.py:270-392)."""

    def __init__(self, C, D, dtype, device, init_scale=1.0, adapt=True):
        self.C, self.D = C, D
        self._scheme = WelfordCovariance(diagonal=True) if adapt else None
        self.inverse_mass_matrix = torch.full((C, D), float(init_scale), dtype=dtype,
                                              device=device)

    @property
    def inverse_mass_matrix(self):
        return self._v

    @inverse_mass_matrix.setter
    def inverse_mass_matrix(self, v):
        if self._scheme is not None:
            self._scheme.reset()
        self._v = v.contiguous()
        self._sqrt_inv = self._v.sqrt()             # mass_matrix_sqrt_inverse
        self._sqrt = self._sqrt_inv.reciprocal()    # mass_matrix_sqrt

    def update(self, z):
        self._scheme.update(z.detach())

    def end_adaptation(self):
        self.inverse_mass_matrix = self._scheme.get_covariance(regularize=True)

    def kinetic_grad(self, r):          # adaptation.py:328-347
        return self._v * r

    def scale(self, r_unscaled):        # adaptation.py:349-373
        return self._sqrt * r_unscaled

    def unscale(self, r):               # adaptation.py:375-392
        return self._sqrt_inv * r


class DenseMassMatrix:
    """Per-chain dense (or block-structured) mass matrix: inverse mass V[C, D, D] and its
    Cholesky factor L (V = L L^T) (reference: BlockMassMatrix with dense blocks,
    adaptation.py:238-392; ``full_mass=True`` or a list of site-name tuples, hmc.py:296-333).

    ``mask`` [D, D] (0/1) encodes the block structure over the flat layout: entries outside the
    dense blocks (and off the diagonal of the remaining sites) are held at zero, which is what
    adapting every block with its own Welford estimator amounts to.

    The reference's three products are provided (kinetic_grad = V r, scale = L^-T eps,
    unscale = L^T r), but the samplers do not call them per leapfrog step: they run in the
    WHITENED coordinates z' = L^-1 z, r' = L^T r, where the same Hamiltonian has unit mass
        U'(z') = U(L z'),  grad' = L^T grad,  K = r'.r'/2,
    so the leapfrog / tree kernels written for a diagonal mass are reused unchanged, and the
    per-step cost of the dense mass is the two per-chain products ``color`` and ``pull``
    (kernels.chain_matvec, one launch each for all chains).  ``r'`` is exactly the reference's
    "unscaled" momentum (unscale(r) = L^T r), so the U-turn test, the kinetic energy and the
    momentum draw r' ~ N(0, I) are the reference's own, term for term."""

    def __init__(self, C, D, dtype, device, mask=None, init_scale=1.0, adapt=True):
        self.C, self.D = C, D
        self._scheme = WelfordCovariance(diagonal=False) if adapt else None
        self._mask = None if mask is None else mask.to(dtype=dtype, device=device)
        self.version = 0
        self.unit_diag = torch.ones((C, D), dtype=dtype, device=device)
        eye = torch.eye(D, dtype=dtype, device=device) * float(init_scale)
        self.inverse_mass_matrix = eye.expand(C, D, D)

    @property
    def inverse_mass_matrix(self):
        return self._v

    @inverse_mass_matrix.setter
    def inverse_mass_matrix(self, v):
        if self._scheme is not None:
            self._scheme.reset()
        if v.dim() == 2:
            v = v.expand(self.C, self.D, self.D)
        # adaptation.py:270-282: sqrt_inverse = cholesky(V)^T = L^T, sqrt = triu_inverse = L^-T
        L = torch.linalg.cholesky(v)
        eye = torch.eye(self.D, dtype=v.dtype, device=v.device).expand_as(L)
        Linv = torch.linalg.solve_triangular(L, eye, upper=False)
        if getattr(self, "_L", None) is not None and self._L.shape == L.shape \
                and self._L.dtype == L.dtype and self._L.device == L.device:
            # same buffers, new contents: a captured potential (jit_compile) keeps pointing at them
            self._v.copy_(v)
            self._L.copy_(L)
            self._Linv.copy_(Linv)
        else:
            self._v, self._L, self._Linv = v.contiguous().clone(), L.contiguous(), Linv.contiguous()
        self.version += 1

    def update(self, z):
        self._scheme.update(z.detach())

    def end_adaptation(self):
        cov = self._scheme.get_covariance(regularize=True)
        if self._mask is not None:
            cov = cov * self._mask
        self.inverse_mass_matrix = cov

    # ---- the reference's products (adaptation.py:328-392) --------------------------------------
    def kinetic_grad(self, r):
        return kernels.chain_matvec(self._v, r.contiguous())

    def scale(self, r_unscaled):
        return kernels.chain_matvec(self._Linv, r_unscaled.contiguous(), transpose=True)

    def unscale(self, r):
        return kernels.chain_matvec(self._L, r.contiguous(), transpose=True)

    # ---- whitened coordinates --------------------------------------------------------------------
    def whiten(self, z):            # z' = L^-1 z
        return kernels.chain_matvec(self._Linv, z.contiguous())

    def color(self, zw):            # z = L z'
        return kernels.chain_matvec(self._L, zw.contiguous())

    def pull(self, grad):           # grad' = L^T grad
        return kernels.chain_matvec(self._L, grad.contiguous(), transpose=True)


class SymmArrowhead(namedtuple("SymmArrowhead", ["top", "bottom_diag"])):
    """(top, bottom_diag) of a symmetric arrowhead matrix (pyro/ops/arrowhead.py:10).  Also answers the
    reference's access through the block key -- ``adapter.mass_matrix[("w", "y", "x", "z")]`` -- with the
    chain dim dropped when there is one chain."""

    __slots__ = ()

    def __getitem__(self, key):
        if isinstance(key, tuple) and key and all(isinstance(k, str) for k in key):
            top, bottom = tuple(self)
            if top.dim() == 3 and top.shape[0] == 1:
                return SymmArrowhead(top[0], bottom[0])
            return self
        return super().__getitem__(key)


class ArrowheadMassMatrix:
    """What the user assigns -- ``kernel.mass_matrix_adapter = ArrowheadMassMatrix()`` before
    ``MCMC.run`` (reference: adaptation.py:395-580, tests/infer/mcmc/test_nuts.py:506-546): the MASS
    matrix is adapted, from the potential's GRADIENTS, with arrowhead structure -- the sites named in
    the kernel's ``full_mass`` form the dense head (their rows / columns are full), every other
    coordinate keeps only its diagonal entry.  ``WarmupAdapter.configure`` turns it into the
    per-chain device object below."""

    def __init__(self, init_scale=1.0):
        self._init_scale = float(init_scale)


def head_indices(layout, dense_mass):
    """Flat indices of the arrowhead's head in the order the sites are named in ``full_mass``
    (adaptation.py:458-474: dense blocks first, concatenated), then of the tail."""
    head = []
    if dense_mass is True:
        head = list(range(layout.D))
    elif dense_mass:
        for block in dense_mass:
            for name in block:
                a, b = layout.slices[name]
                head.extend(range(a, b))
    taken = set(head)
    tail = [i for i in range(layout.D) if i not in taken]
    return head, tail


class ArrowheadDenseMassMatrix(DenseMassMatrix):
    """Per-chain arrowhead mass matrix M[C, D, D] (zero outside the head's rows / columns and the
    diagonal), adapted as the regularised covariance of the potential's gradients
    (WelfordArrowheadCovariance, pyro/ops/welford.py:55-101: the entries an arrowhead keeps are the
    same entries of the full Welford estimate).  The samplers only need V = M^-1 and its Cholesky
    factor, which the dense machinery above provides (whitened coordinates, pa_chain_matvec): the
    arrowhead is held as a dense [D, D] per chain -- O(D^2) memory instead of the reference's
    O(D x head), same transitions.  If the masked estimate is not positive definite the head-tail
    block is halved and the factorisation retried, up to six times (pyro/ops/arrowhead.py:30-52)."""

    uses_grad = True

    def __init__(self, C, D, dtype, device, head, tail, init_scale=1.0, adapt=True):
        self._head = torch.tensor(head, dtype=torch.int64, device=device)
        self._tail = torch.tensor(tail, dtype=torch.int64, device=device)
        mask = torch.eye(D, dtype=dtype, device=device)
        if head:
            mask[self._head, :] = 1.0
            mask[:, self._head] = 1.0
        self._arrow_mask = mask
        cross = torch.zeros((D, D), dtype=dtype, device=device)      # the head-tail block B and B^T
        if head and tail:
            cross[self._head.unsqueeze(-1), self._tail.unsqueeze(0)] = 1.0
            cross[self._tail.unsqueeze(-1), self._head.unsqueeze(0)] = 1.0
        self._cross = cross
        self._mass = None
        super().__init__(C, D, dtype, device, mask=None, init_scale=1.0, adapt=adapt)
        eye = torch.eye(D, dtype=dtype, device=device) * float(init_scale)
        self.mass_matrix_dense = eye.expand(C, D, D)

    @property
    def mass_matrix_dense(self):
        return self._mass

    @mass_matrix_dense.setter
    def mass_matrix_dense(self, M):
        M = (M * self._arrow_mask).contiguous()
        self._mass = M                 # as assigned (the reference keeps the un-halved matrix too)
        for attempt in range(6):
            _, info = torch.linalg.cholesky_ex(M)
            bad = info != 0
            if not bool(bad.any()):
                break
            # halve the head-tail block of the chains whose matrix is not positive definite
            M = torch.where(bad.reshape(-1, 1, 1) & (self._cross > 0), M * 0.5, M)
        else:
            raise RuntimeError("Singular schur complement in computing Cholesky of the input "
                               "arrowhead matrix")
        self.inverse_mass_matrix = torch.cholesky_inverse(torch.linalg.cholesky(M))

    @property
    def mass_matrix(self):
        """SymmArrowhead(top [C, head, D], bottom_diag [C, D - head]) with the coordinates in the
        reference's order: head sites as named in ``full_mass``, then the others."""
        order = torch.cat([self._head, self._tail])
        M = self._mass[:, order][:, :, order]
        h = self._head.numel()
        return SymmArrowhead(M[:, :h, :], torch.diagonal(M, dim1=-2, dim2=-1)[:, h:])

    def update(self, z_grad):
        self._scheme.update(z_grad.detach())

    def end_adaptation(self):
        self.mass_matrix_dense = self._scheme.get_covariance(regularize=True)


def block_mask(layout, dense_mass):
    """0/1 matrix [D, D] of the mass-matrix structure over a flat Layout: ``dense_mass`` True =
    one dense block over all sites; a list of site-name tuples = one dense block per tuple, the
    remaining sites diagonal (hmc.py:296-326)."""
    D = layout.D
    if dense_mass is True:
        return None
    msg = "full_mass should be a list of tuples of site names."
    assert isinstance(dense_mass, list), msg
    mask = torch.eye(D)
    seen = set()
    for block in dense_mass:
        assert block and isinstance(block, tuple), msg
        idx = []
        for name in block:
            assert isinstance(name, str) and name in layout.slices, msg
            assert name not in seen, "Site names specified in full_mass are duplicated."
            seen.add(name)
            a, b = layout.slices[name]
            idx.extend(range(a, b))
        idx = torch.tensor(idx)
        mask[idx.unsqueeze(-1), idx.unsqueeze(0)] = 1.0
    return mask


class WarmupAdapter:
    def __init__(self, step_size=1, adapt_step_size=False, target_accept_prob=0.8,
                 adapt_mass_matrix=False, dense_mass=False):
        self.adapt_step_size = adapt_step_size
        self.adapt_mass_matrix = adapt_mass_matrix
        self.target_accept_prob = target_accept_prob
        self.dense_mass = dense_mass
        self.arrowhead = None          # an ArrowheadMassMatrix prototype assigned by the user
        self._init_step_size = 1 if step_size is None else step_size
        self.step_size = None           # tensor [C] after configure
        self._adaptation_disabled = not (adapt_step_size or adapt_mass_matrix)
        if adapt_step_size:
            self._step_size_adapt_scheme = DualAveraging()
        self._adapt_start_buffer = 75
        self._adapt_end_buffer = 50
        self._adapt_initial_window = 25
        self._warmup_steps = None
        self._adaptation_schedule = []
        self._find_reasonable_step_size = None
        self.mass_matrix_adapter = None

    def _build_adaptation_schedule(self):
        return build_adaptation_schedule(self._warmup_steps, self._adapt_start_buffer,
                                         self._adapt_end_buffer, self._adapt_initial_window)

    def configure(self, warmup_steps, C, D, dtype, device, initial_step_size=None,
                  find_reasonable_step_size_fn=None, layout=None):
        self._warmup_steps = warmup_steps
        s = self._init_step_size if initial_step_size is None else initial_step_size
        if isinstance(s, torch.Tensor):
            self.step_size = s.to(dtype=dtype, device=device).expand(C).contiguous().clone()
        else:
            self.step_size = torch.full((C,), float(s), dtype=dtype, device=device)
        self._find_reasonable_step_size = find_reasonable_step_size_fn
        if self.arrowhead is not None:
            head, tail = head_indices(layout, self.dense_mass)
            self.mass_matrix_adapter = ArrowheadDenseMassMatrix(
                C, D, dtype, device, head, tail, init_scale=self.arrowhead._init_scale,
                adapt=self.adapt_mass_matrix)
        elif self.dense_mass:
            mask = block_mask(layout, self.dense_mass) if layout is not None else None
            self.mass_matrix_adapter = DenseMassMatrix(C, D, dtype, device, mask=mask,
                                                       adapt=self.adapt_mass_matrix)
        else:
            self.mass_matrix_adapter = DiagMassMatrix(C, D, dtype, device,
                                                      adapt=self.adapt_mass_matrix)
        if not self._adaptation_disabled:
            self._adaptation_schedule = self._build_adaptation_schedule()
        self._current_window = 0
        if self.adapt_step_size:
            self._step_size_adapt_scheme.reset()

    def reset_step_size_adaptation(self, z):
        """New reasonable step size per chain + restart of dual averaging
        (reference: adaptation.py:105-113)."""
        if self._find_reasonable_step_size is not None:
            self.step_size = self._find_reasonable_step_size(z)
        self._step_size_adapt_scheme.prox_center = torch.log(10 * self.step_size)
        self._step_size_adapt_scheme.reset()

    def _update_step_size(self, accept_prob):
        H = self.target_accept_prob - accept_prob
        self._step_size_adapt_scheme.step(H)
        log_step_size, _ = self._step_size_adapt_scheme.get_state()
        self.step_size = torch.exp(log_step_size).contiguous()

    def _end_adaptation(self):
        if self.adapt_step_size:
            _, log_step_size_avg = self._step_size_adapt_scheme.get_state()
            self.step_size = torch.exp(log_step_size_avg).contiguous()

    def step(self, t, z, accept_prob, z_grad=None):
        """Called once per warm-up transition with t = 1, 2, ... (the reference passes the
        already-incremented counter, hmc.py:424-431); z [C, D], accept_prob [C]."""
        if t >= self._warmup_steps or self._adaptation_disabled:
            return
        w, last = self._current_window, len(self._adaptation_schedule) - 1
        slow = self.adapt_mass_matrix and 0 < w < last       # a mass-matrix window (not the two fast ones)
        if self.adapt_step_size:
            # NaN acceptance probabilities (diverged chains) count as 0, as exp(-inf) does
            self._update_step_size(torch.nan_to_num(accept_prob, nan=0.0))
        if slow:
            mm = self.mass_matrix_adapter
            mm.update(z_grad if getattr(mm, "uses_grad", False) else z)
        if t != self._adaptation_schedule[w].end:
            return
        # ---- the window closes with this transition ----
        self._current_window = w + 1
        if w == last:
            self._end_adaptation()               # averaged step size from here on
        elif slow:
            self.mass_matrix_adapter.end_adaptation()        # new metric: the step-size search restarts
            if self.adapt_step_size:
                self.reset_step_size_adaptation(z)

    # ---- bulk interface: the per-transition part of step() runs inside the persistent NUTS kernel
    # (pa_nuts_gaussian_run); the host only handles the window-end events -------------------------
    def bulk_span(self, t_done):
        """(k, adapting): the transitions t_done+1 .. t_done+k can share one launch -- no window
        end strictly inside -- and whether step() would adapt during them."""
        remaining = self._warmup_steps - t_done
        if self._adaptation_disabled or t_done + 1 >= self._warmup_steps:
            return remaining, False
        window = self._adaptation_schedule[self._current_window]
        k = max(1, min(window.end - t_done, self._warmup_steps - 1 - t_done))
        return k, True

    def in_mass_phase(self):
        num_windows = len(self._adaptation_schedule)
        return self.adapt_mass_matrix and (0 < self._current_window < num_windows - 1)

    def da_state(self):
        """Dual-averaging record [C, 5] of the persistent kernel (ops/dual_averaging.py)."""
        return self._step_size_adapt_scheme.to_record(self.step_size.shape[0], self.step_size.dtype,
                                                      self.step_size.device)

    def load_da_state(self, st, k):
        self._step_size_adapt_scheme.from_record(st, k)

    def welford_state(self):
        """([C, 2, D] {mean, scatter}, draws seen) of the mass-matrix estimator."""
        w = self.mass_matrix_adapter._scheme
        v = self.mass_matrix_adapter.inverse_mass_matrix
        return w.to_record(v.shape[0], v.shape[1], v.dtype, v.device), w.n_samples

    def load_welford_state(self, st, k):
        self.mass_matrix_adapter._scheme.from_record(st, k)

    def finish_span(self, t, z):
        """The window-end branch of step() (adaptation.py:186-202) for a span that ended at t."""
        if t >= self._warmup_steps or self._adaptation_disabled:
            return
        window = self._adaptation_schedule[self._current_window]
        if t != window.end:
            return
        num_windows = len(self._adaptation_schedule)
        mass_phase = self.in_mass_phase()
        if self._current_window == num_windows - 1:
            self._current_window += 1
            self._end_adaptation()
            return
        if self._current_window == 0:
            self._current_window += 1
            return
        if mass_phase:
            self.mass_matrix_adapter.end_adaptation()
            if self.adapt_step_size:
                self.reset_step_size_adaptation(z)
        self._current_window += 1

    @property
    def adaptation_schedule(self):
        return self._adaptation_schedule


__all__ = ["WarmupAdapter", "DiagMassMatrix", "DenseMassMatrix", "block_mask", "adapt_window", "build_adaptation_schedule",
           "math"]
