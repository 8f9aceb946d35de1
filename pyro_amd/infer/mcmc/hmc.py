"""Hamiltonian Monte Carlo, vectorised over chains on one GPU
(reference: pyro/infer/mcmc/hmc.py:31-452; same constructor, same MCMCKernel interface).

The reference runs one chain per process; here ``num_chains`` chains are rows of a flat state
``z[C, D]`` and every leapfrog step is two HIP launches (pa_leapfrog_kick_drift,
pa_leapfrog_kick) around ONE chain-batched potential evaluation.  Each chain keeps its own step
size, mass matrix and adaptation state (see adaptation.py).  Randomness comes from the Philox
stream (pyro_amd.rng) instead of pyro.sample sites.
"""
import math
from collections import OrderedDict

import torch

from ... import kernels, rng
from ..autoguide.initialization import init_to_uniform
from .adaptation import WarmupAdapter
from .mcmc_kernel import MCMCKernel
from .util import FlatPotential, GraphedPotential, Layout, initialize_model


class _UnitMassView:
    """What the kernels see when the sampler runs in the whitened coordinates of a dense mass
    matrix (adaptation.DenseMassMatrix): unit diagonal inverse mass, scale/unscale = identity."""

    def __init__(self, dense):
        self.inverse_mass_matrix = dense.unit_diag

    @staticmethod
    def scale(r_unscaled):
        return r_unscaled

    @staticmethod
    def unscale(r):
        return r


class _WhitenedPotential:
    """U'(z') = U(L z'), grad' = L^T grad for the CURRENT Cholesky factor of the adapter's dense
    inverse mass (two kernels.chain_matvec launches around the potential itself)."""

    def __init__(self, base, adapter):
        self.base, self.adapter = base, adapter

    def __call__(self, zw):
        mm = self.adapter.mass_matrix_adapter
        pe, grad = self.base(mm.color(zw))
        return pe, mm.pull(grad)


class HMC(MCMCKernel):
    def __init__(self, model=None, potential_fn=None, step_size=1, trajectory_length=None,
                 num_steps=None, adapt_step_size=True, adapt_mass_matrix=True, full_mass=False,
                 transforms=None, max_plate_nesting=None, jit_compile=False, jit_options=None,
                 ignore_jit_warnings=False, target_accept_prob=0.8,
                 init_strategy=init_to_uniform, *, min_stepsize=1e-10, max_stepsize=1e10):
        if not ((model is None) ^ (potential_fn is None)):
            raise ValueError("Only one of `model` or `potential_fn` must be specified.")
        self.model = model
        self.transforms = transforms
        self._max_plate_nesting = max_plate_nesting
        self._jit_compile = jit_compile
        self._init_strategy = init_strategy
        self._min_stepsize = min_stepsize
        self._max_stepsize = max_stepsize
        self.potential_fn = potential_fn
        if trajectory_length is not None:
            self.trajectory_length = trajectory_length
        elif num_steps is not None:
            self.trajectory_length = step_size * num_steps
        else:
            self.trajectory_length = 2 * math.pi  # from Stan
        self._direction_threshold = math.log(0.8)  # from Stan
        self._max_sliced_energy = 1000
        self.num_chains = 1
        self.chain_offset = 0      # first global chain index of this rank (chain-sharded runs)
        self._dense = bool(full_mass)
        self._full_mass = full_mass
        self._reset()
        self._adapter = WarmupAdapter(step_size, adapt_step_size=adapt_step_size,
                                      adapt_mass_matrix=adapt_mass_matrix,
                                      target_accept_prob=target_accept_prob,
                                      dense_mass=full_mass)
        super().__init__()

    # ---- bookkeeping -------------------------------------------------------------------------
    def _reset(self):
        self._t = 0
        self._accept_cnt = None
        self._mean_accept_prob = None
        self._divergences = []
        self._prototype_trace = None
        self._initial_params = None
        self._z = self._pe = self._grad = None
        self._zr, self._white_version = None, -1
        self._warmup_steps = None
        self._layout = None
        self._n_leapfrog_total = None

    @property
    def mass_matrix_adapter(self):
        return self._adapter.mass_matrix_adapter

    @mass_matrix_adapter.setter
    def mass_matrix_adapter(self, value):
        """``kernel.mass_matrix_adapter = ArrowheadMassMatrix()`` (reference: hmc.py:334-347): the
        head of the arrowhead is the sites of ``full_mass``; takes effect at the next setup."""
        from .adaptation import ArrowheadMassMatrix
        if not isinstance(value, ArrowheadMassMatrix):
            raise TypeError("mass_matrix_adapter accepts an ArrowheadMassMatrix()")
        self._adapter.arrowhead = value
        self._dense = True          # the samplers run in the whitened coordinates of M^-1

    @property
    def inverse_mass_matrix(self):
        """The reference's public form (hmc.py:349-351, adaptation.py:196-216): a dict from the tuple of
        site names of a mass-matrix block to that block of M^-1 -- 1-D (diagonal block) or 2-D (dense
        block).  ``full_mass=False`` / ``True``: one block over all sites (sorted by name);
        ``full_mass=[("a",), ("b", "c")]``: those dense blocks plus one diagonal block of the remaining
        sites; with an ``ArrowheadMassMatrix``: one block, head sites first.  With several chains every
        value has a leading chain dim (the chains adapt separately, as the reference's processes do).
        The flat tensor the kernels use is ``kernel.mass_matrix_adapter.inverse_mass_matrix``."""
        layout = getattr(self, "_layout", None)
        if layout is None:
            return {}
        V = self.mass_matrix_adapter.inverse_mass_matrix              # [C, D] or [C, D, D]
        dense = V.dim() == 3

        def flat(names):
            idx = []
            for name in names:
                a, b = layout.slices[name]
                idx.extend(range(a, b))
            return torch.tensor(idx, dtype=torch.int64, device=V.device)

        def block(names, as_dense):
            i = flat(names)
            if dense:
                sub = V[:, i][:, :, i]
                out = sub if as_dense else torch.diagonal(sub, dim1=-2, dim2=-1)
            else:
                out = V[:, i]
            return out if self._batched else out[0]

        full_mass = getattr(self, "_full_mass", False)
        blocks = full_mass if isinstance(full_mass, list) else []
        named = [name for b in blocks for name in b]
        rest = tuple(n for n in layout.names if n not in set(named))
        if getattr(self._adapter, "arrowhead", None) is not None:
            key = tuple(named) + rest if blocks else tuple(layout.names)
            return {key: block(key, True)}
        if not blocks:
            return {tuple(layout.names): block(layout.names, dense)}
        out = {tuple(b): block(b, True) for b in blocks}
        if rest:
            out[rest] = block(rest, False)
        return out

    @property
    def _mm_eff(self):
        """Mass-matrix object of the coordinates the kernels run in: the adapter itself (diagonal
        mass) or the unit mass of the whitened problem (dense mass)."""
        mm = self.mass_matrix_adapter
        if not self._dense:
            return mm
        view = getattr(self, "_unit_view", None)
        if view is None or view.inverse_mass_matrix is not mm.unit_diag:
            view = self._unit_view = _UnitMassView(mm)
        return view

    def _position(self):
        """The chains' positions in the model's (unconstrained) coordinates; ``self._z`` holds
        the whitened coordinates when the mass matrix is dense."""
        return self._zr if self._dense else self._z

    def _sync_coordinates(self):
        """Dense mass only: if the mass matrix was replaced since the state was whitened (a
        warm-up window ended, or the user assigned ``inverse_mass_matrix``), re-express it."""
        if self._dense and self.mass_matrix_adapter.version != self._white_version:
            self._rewhiten(self._zr)

    def _rewhiten(self, z_real):
        """(Re)express the state in the whitened coordinates of the current dense mass matrix."""
        self._zr = z_real.clone()
        self._white_version = self.mass_matrix_adapter.version
        zw = self.mass_matrix_adapter.whiten(z_real)
        pe, grad = self._potential(zw)
        if self._z is None or self._pe is None:
            self._z, self._pe, self._grad = zw, pe.detach().contiguous(), grad.detach().contiguous()
        else:       # in place: the NUTS tree workspace keeps references to these buffers
            self._z.copy_(zw)
            self._pe.copy_(pe.detach())
            self._grad.copy_(grad.detach())

    @property
    def step_size(self):
        return self._adapter.step_size

    @property
    def initial_params(self):
        return self._initial_params

    @initial_params.setter
    def initial_params(self, params):
        self._initial_params = params
        self._drawn_initial_params = False      # given by the caller: kept across runs

    # ---- setup -------------------------------------------------------------------------------
    def _initialize_model_properties(self, model_args, model_kwargs):
        init_params, potential_fn, transforms, trace = initialize_model(
            self.model, model_args, model_kwargs, transforms=self.transforms,
            max_plate_nesting=self._max_plate_nesting, num_chains=self.num_chains,
            init_strategy=self._init_strategy,
            # points drawn by an earlier run are not reused (the reference forgets them in
            # cleanup(), hmc.py:363-364): every run of a model kernel initialises afresh
            initial_params=None if getattr(self, "_drawn_initial_params", False)
            else self._initial_params)
        self._drawn_initial_params = getattr(self, "_drawn_initial_params", False) or \
            self._initial_params is None
        self.potential_fn = potential_fn
        self.transforms = transforms
        self._initial_params = init_params
        self._prototype_trace = trace

    def setup(self, warmup_steps, *args, **kwargs):
        # a kernel object that already ran starts over: transition counter (it indexes the
        # per-transition random streams and the adaptation schedule), statistics, whitened state
        self._t = 0
        self._divergences = []
        self._zr, self._white_version = None, -1
        self._warmup_steps = warmup_steps
        if self.model is not None:
            self._initialize_model_properties(args, kwargs)
        self._empty = not self._initial_params
        if self._empty:
            # a model without continuous latent sites: nothing to move (the reference's kernels run
            # such a model too, test_mcmc_api.py:236-286); transitions are no-ops, samples are {}
            self._warmup_steps = warmup_steps
            return
        params = self._initial_params
        C = self.num_chains
        first = next(iter(params.values()))
        # chain-batched parameters carry a leading dim of size num_chains; a single chain may
        # pass un-batched tensors exactly as with the reference
        self._batched = C > 1 or bool(getattr(self, "_force_batched", False))
        if C > 1:
            for k, v in params.items():
                if v.dim() == 0 or v.shape[0] != C:
                    raise ValueError("initial_params['{}'] must have a leading dim of size "
                                     "num_chains={} (got shape {})".format(k, C, tuple(v.shape)))
        shapes = {k: (v.shape[1:] if self._batched else v.shape) for k, v in params.items()}
        self._layout = Layout(shapes)
        self._potential = FlatPotential(self.potential_fn, self._layout, self._batched)
        z = self._layout.flatten({k: v.detach() for k, v in params.items()}, C, self._batched)
        kernels._require_gpu(z)
        self._adapter.configure(warmup_steps, C, self._layout.D, z.dtype, z.device,
                                find_reasonable_step_size_fn=self._find_reasonable_step_size,
                                layout=self._layout)
        if self._dense:
            self._potential = _WhitenedPotential(self._potential, self._adapter)
        if self._jit_compile:
            # the reference's jit_compile traces the potential with torch.jit; here the whole
            # evaluation (model + backward + mass-matrix products) becomes a hipGraph replay
            self._potential = GraphedPotential(self._potential)
        if self._dense:
            self._z = self._pe = None
            self._rewhiten(z)
        else:
            self._z = z.clone()
            pe, grad = self._potential(self._z)
            self._pe, self._grad = pe.detach().contiguous(), grad.detach().contiguous()
        self._accept_cnt = torch.zeros((C,), dtype=torch.int64, device=z.device)
        self._mean_accept_prob = torch.zeros((C,), dtype=z.dtype, device=z.device)
        self._n_leapfrog_total = torch.zeros((), dtype=torch.int64, device=z.device)
        self._seed = rng.current_seed()
        self._prepare_paths()
        if self._adapter.adapt_step_size:
            self._adapter.reset_step_size_adaptation(z)

    def _prepare_paths(self):
        """Hook: device fast paths that depend on the state laid out by setup() (NUTS: the fused
        closed-form Gaussian kernels) are chosen before the first step-size search."""

    def cleanup(self):
        self._reset()

    def release_graphs(self):
        """Drop captured hipGraphs (jit_compile) and their static buffers; the sampler state and
        its statistics stay.  The next setup() / transition re-captures."""
        pot = getattr(self, "_potential", None)
        if isinstance(pot, GraphedPotential):
            pot.graph, pot.calls = None, 0
            pot.z_static = pot.pe_static = pot.grad_static = None

    # ---- pieces of a transition --------------------------------------------------------------
    def _kinetic_energy(self, r_unscaled):
        return 0.5 * (r_unscaled * r_unscaled).sum(-1)       # hmc.py:152-156

    def _sample_r(self):
        r_unscaled = rng.normal(self._z.shape, self._z.dtype, self._z.device)
        return self._mm_eff.scale(r_unscaled), r_unscaled   # hmc.py:231-248

    def _leapfrog(self, z, r, grad, step):
        """One velocity-Verlet step for all chains, in place on (z, r); returns (pe, grad)."""
        step = step.contiguous()
        kernels.leapfrog_kick_drift(z, r, grad, self._mm_eff.inverse_mass_matrix, step)
        pe, grad = self._potential(z)
        grad = grad.contiguous()
        kernels.leapfrog_kick(r, grad, step)
        return pe, grad

    def _find_reasonable_step_size(self, z):
        """Per chain: double / halve the step size until the one-step acceptance probability
        crosses the target (reference: hmc.py:170-229), all chains in lock step under masks."""
        step = self.step_size.clone()
        if getattr(self, "_fused", False) and not self._dense \
                and getattr(self, "_Lambda", None) is not None and z.is_cuda:
            # closed-form Gaussian potential: every chain runs the loop on the device, one launch
            pe, grad = self._potential(z)
            self._find_step_calls = getattr(self, "_find_step_calls", 0) + 1
            key = (1 << 42) + 256 * self._find_step_calls       # disjoint from transition indices
            return kernels.nuts_gaussian_find_step(
                z, pe.detach(), grad.detach(), self._Lambda, self.mass_matrix_adapter.inverse_mass_matrix, step,
                self._seed if getattr(self, "_seed", None) is not None else rng.current_seed(),
                key, self.chain_offset, self._min_stepsize, self._max_stepsize,
                self._direction_threshold)
        if self._dense:
            z = self.mass_matrix_adapter.whiten(z)    # ``z`` arrives in model coordinates
        pe, grad = self._potential(z)
        grad = grad.contiguous()
        mm = self._mm_eff

        def trial(step):
            r, r_u = self._sample_r()
            e0 = self._kinetic_energy(r_u) + pe
            z1, r1 = z.clone(), r.contiguous().clone()
            pe1, _ = self._leapfrog(z1, r1, grad, step)
            e1 = self._kinetic_energy(mm.unscale(r1)) + pe1
            delta = e1 - e0
            # NaN (diverged trial) compares False -> direction -1, as in the reference
            return torch.where(self._direction_threshold < -delta, 1, -1)

        direction = trial(step)
        scale = torch.pow(torch.full_like(step, 2.0), direction.to(step.dtype))
        active = torch.ones_like(direction, dtype=torch.bool)
        for _ in range(200):
            active = active & (step > self._min_stepsize) & (step < self._max_stepsize)
            if not bool(active.any()):
                break
            step = torch.where(active, step * scale, step)
            active = active & (trial(step) == direction)
        return step.clamp(min=self._min_stepsize, max=self._max_stepsize).contiguous()

    def _after_transition(self, accept_prob, accepted, diverging):
        self._t += 1
        if self._dense:
            self._zr = self.mass_matrix_adapter.color(self._z)
        if self._t > self._warmup_steps:
            n = self._t - self._warmup_steps
            self._accept_cnt += accepted.to(torch.int64)
            self._divergences.append(diverging)
        else:
            n = self._t
            mm = self.mass_matrix_adapter
            z_grad = None
            if getattr(mm, "uses_grad", False):     # model-coordinates gradient: L^-T grad'
                z_grad = mm.scale(self._grad)
            self._adapter.step(self._t, self._position(), accept_prob, z_grad)
            self._sync_coordinates()      # a window may have ended: new mass, new coordinates
        self._mean_accept_prob += (torch.nan_to_num(accept_prob, nan=0.0)
                                   - self._mean_accept_prob) / n

    # ---- one transition for all chains -------------------------------------------------------
    def _transition(self):
        self._sync_coordinates()
        z, pe, grad = self._z, self._pe, self._grad
        mm = self._mm_eff
        r, r_u = self._sample_r()
        energy_current = self._kinetic_energy(r_u) + pe
        step = self.step_size
        num_steps = torch.clamp((self.trajectory_length / step).floor(), min=1)
        L = int(num_steps.max().item())
        z_new, r_new, g_new, pe_new = z.clone(), r.contiguous().clone(), grad, pe
        zero = torch.zeros_like(step)
        for i in range(L):
            st = torch.where(num_steps > i, step, zero)   # a chain that is done stands still
            pe_new, g_new = self._leapfrog(z_new, r_new, g_new, st)
        self._n_leapfrog_total += num_steps.sum().to(torch.int64)
        energy_proposal = self._kinetic_energy(mm.unscale(r_new)) + pe_new
        delta = energy_proposal - energy_current
        delta = torch.where(torch.isnan(delta), torch.full_like(delta, float("inf")), delta)
        diverging = delta > self._max_sliced_energy
        accept_prob = (-delta).exp().clamp(max=1.0)
        rand = rng.uniform(accept_prob.shape, accept_prob.dtype, accept_prob.device)
        accepted = rand < accept_prob
        m = accepted.unsqueeze(-1)
        self._z = torch.where(m, z_new, z).contiguous()
        self._grad = torch.where(m, g_new, grad).contiguous()
        self._pe = torch.where(accepted, pe_new, pe).contiguous()
        self._after_transition(accept_prob, accepted, diverging)

    def sample(self, params):
        """One transition; ``params`` is ignored in favour of the cached state when it is the
        state returned by the previous call (the reference caches the same way, hmc.py:371-379)."""
        if getattr(self, "_empty", False):
            return params
        if getattr(self, "_cache_cleared", False):
            self._cache_cleared = False
            self._restart_at(params)
        self._transition()
        return self._layout.unflatten(self._position().clone(), self._batched)

    def clear_cache(self):
        """Forget the cached state (position, potential, gradient): the next ``sample(params)`` starts
        from ``params`` and evaluates the potential there (reference: hmc.py:363-379)."""
        self._cache_cleared = True

    def _restart_at(self, params):
        z = self._layout.flatten(params, self.num_chains, self._batched)
        z = z.to(self._z.dtype) if self._z is not None else z
        if self._dense:
            self._rewhiten(z)
        else:                       # in place: device fast paths keep references to these buffers
            self._z.copy_(z)
            pe, grad = self._potential(self._z)
            self._pe.copy_(pe.detach())
            self._grad.copy_(grad.detach())

    # ---- reporting ---------------------------------------------------------------------------
    def logging(self):
        return OrderedDict([("step size", "{:.2e}".format(float(self.step_size.mean()))),
                            ("acc. prob", "{:.3f}".format(float(self._mean_accept_prob.mean())))])

    def diagnostics(self):
        if getattr(self, "_empty", False):
            return {}
        n = max(self._t - self._warmup_steps, 1)
        rate = (self._accept_cnt.to(torch.float64) / n).cpu().tolist()
        # one entry per chain, as the reference's driver assembles them (api.py:560-575)
        out = {"acceptance rate": {"chain {}".format(c): r for c, r in enumerate(rate)}}
        if self._divergences:
            div = torch.stack(self._divergences).to(torch.bool).cpu().numpy()    # [S, C]
            # one pass over the (few) divergent transitions instead of a nonzero() per chain
            per_chain = {c: [] for c in range(div.shape[1])}
            if div.any():
                import numpy as np
                for s_, c in zip(*np.nonzero(div)):
                    per_chain[int(c)].append(int(s_))
            out["divergences"] = {"chain {}".format(c): v for c, v in per_chain.items()}
        else:
            out["divergences"] = {}
        return out

    @property
    def num_leapfrog_steps(self):
        """Total leapfrog steps taken by all chains so far (device counter, host read)."""
        return int(self._n_leapfrog_total.item())
