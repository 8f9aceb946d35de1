"""No-U-Turn Sampler, vectorised over chains (reference: pyro/infer/mcmc/nuts.py:67-522; same
constructor and MCMCKernel interface).

Two device paths, both restating the reference's tree doubling chain by chain:

* closed-form Gaussian potential (``GaussianPotential``, D <= 128): ONE fused launch per
  transition, one wavefront per chain (pa_nuts_gaussian_transition);
* any other potential: the per-chain tree state machine pa_nuts_tree_begin/advance with one
  chain-batched potential evaluation per leapfrog step; chains whose tree is finished idle.

Both use the same keyed Philox draws, so they produce the same chains (and the same chains as
oracle/nuts.py) up to floating-point rounding of the potential.
"""
from collections import OrderedDict

import torch

from ... import kernels
from ...util import capture_scope
from ..autoguide.initialization import init_to_uniform
from .hmc import HMC
from .potentials import GaussianPotential

# Rounds of the tree path (potential at the cursors + tree bookkeeping) are replayed from a captured
# hipGraph: "auto" = whenever the chains live on the device (a capture that fails -- the potential
# synchronises with the host, say -- falls back to eager rounds), True / False = always / never.
# ``jit_compile=True`` asks for it whatever this says (and warns when the capture fails).
CAPTURE_ROUNDS = "auto"


class NUTS(HMC):
    def __init__(self, model=None, potential_fn=None, step_size=1, adapt_step_size=True,
                 adapt_mass_matrix=True, full_mass=False, use_multinomial_sampling=True,
                 transforms=None, max_plate_nesting=None, jit_compile=False, jit_options=None,
                 ignore_jit_warnings=False, target_accept_prob=0.8, max_tree_depth=10,
                 init_strategy=init_to_uniform):
        super().__init__(model, potential_fn, step_size, adapt_step_size=adapt_step_size,
                         adapt_mass_matrix=adapt_mass_matrix, full_mass=full_mass,
                         transforms=transforms, max_plate_nesting=max_plate_nesting,
                         jit_compile=jit_compile, jit_options=jit_options,
                         ignore_jit_warnings=ignore_jit_warnings,
                         target_accept_prob=target_accept_prob, init_strategy=init_strategy)
        self.use_multinomial_sampling = use_multinomial_sampling
        self._max_tree_depth = max_tree_depth
        self._max_sliced_energy = 1000
        self._tree = None
        self.sync_every = 4      # host polls of n_active in the generic tree loop
        self.use_fused_gaussian = True
        self.use_persistent = True   # many transitions per launch on the fused Gaussian path
        self.use_async_chains = True # model / generic potentials: spans of transitions without lock step
        self.rounds_per_replay = 16  # tree rounds in the captured graph of the asynchronous path
        self.use_direct_potential = True   # flat models: the potential without handlers / autograd (direct.py)
        self.compact_chains = True   # late in a span: rounds over the chains still active only
        self.min_slots = 64          # smallest compacted round (the GLM kernels take 64 columns per pass)
        self._launch_hook = None     # called before every fused launch (bench: event brackets)

    def release_graphs(self):
        super().release_graphs()
        self._round_graph = self._step_buf = self._mass_buf = None
        self._round_calls = 0
        self._span_graph = self._span_keep = None
        self._span_graphs = {}
        self._span_calls = 0

    def _prepare_paths(self):
        self._fused = (self.use_fused_gaussian and isinstance(self.potential_fn, GaussianPotential)
                       and self._layout.D <= 128 and len(self._layout.names) == 1
                       and not self._dense)   # dense mass: per-chain whitened potential, tree path
        self._Lambda = None
        if self._fused:
            self._Lambda = self.potential_fn.precision.to(self._z.dtype).contiguous()

    def setup(self, warmup_steps, *args, **kwargs):
        self._fused, self._Lambda = False, None
        super().setup(warmup_steps, *args, **kwargs)
        if getattr(self, "_empty", False):
            return
        self._tree = None
        self._step_buf = self._mass_buf = self._round_graph = None     # jit_compile round graph
        self._round_calls, self._round_failed = 0, False
        self._span_graph = self._span_keep = None                      # asynchronous spans
        self._span_graphs = {}                                         # rounds per replay size -> graph
        self._span_calls, self._span_failed, self._span_replays = 0, False, 0
        self._span_rounds = 0                                          # slot-rounds evaluated (occupancy)
        self._span_compactions = 0
        self._da_buf = self._wf_buf = None
        self._direct = None
        if self.use_direct_potential and self.model is not None and not self._dense and self.num_chains > 1 \
                and self._z.is_cuda:
            from . import direct
            maker = getattr(self.potential_fn, "__self__", None)       # the bound _PEMaker.potential_fn
            if maker is not None:
                self._direct = direct.recognise(maker, self._layout, self.transforms, self._initial_params,
                                                self.num_chains)
        self._tree_depth_sum = torch.zeros((), dtype=torch.int64, device=self._z.device)
        self._counters = torch.zeros((3, self.num_chains), dtype=torch.int64,
                                     device=self._z.device)

    def _transition(self):
        self._sync_coordinates()
        t = self._t
        step = self.step_size.contiguous()
        inv_mass = self._mm_eff.inverse_mass_matrix
        if self._fused:
            out = kernels.nuts_gaussian_transition(
                self._z, self._pe, self._grad, self._Lambda, inv_mass, step,
                self._max_tree_depth, self.use_multinomial_sampling,
                self._seed, t, self.chain_offset)
        else:
            out = self._tree_transition(t, step, inv_mass)
        self._n_leapfrog_total += out["n_leapfrog"].sum()
        self._tree_depth_sum += out["depth"].sum()
        self._last_stats = out
        self._after_transition(out["accept_prob"], out["accepted"] != 0, out["diverging"] != 0)

    @property
    def bulk_ready(self):
        """MCMC.run may drive this kernel through ``_transition_many`` (many transitions without a
        host decision): the fused Gaussian launches, or asynchronous chains on the tree path --
        diagonal mass only (a dense / arrowhead mass keeps its per-transition host step)."""
        if getattr(self, "_empty", False) or self._z is None:
            return False
        if self._fused:
            return bool(self.use_persistent)
        return bool(self.use_async_chains) and not self._dense and self._layout.D <= 2048 \
            and not getattr(self.mass_matrix_adapter, "uses_grad", False)

    def _transition_many(self, k, samples=None):
        """Up to ``k`` transitions per chain without a host decision inside: returns how many
        were done -- a span never crosses a warm-up window end or the warm-up/sampling border.
        The per-transition half of the adaptation (dual averaging, Welford) runs in the kernels."""
        if not self._fused:
            return self._span(k, samples)
        ad = self._adapter
        warm = self._t < self._warmup_steps
        da = wf = None
        wf_n0 = 0
        if warm:
            span, adapting = ad.bulk_span(self._t)
            k = min(k, span)
            if adapting and ad.adapt_step_size:
                da = ad.da_state()
            if adapting and ad.in_mass_phase():
                wf, wf_n0 = ad.welford_state()
            mean_n0 = self._t
        else:
            mean_n0 = self._t - self._warmup_steps
        step = ad.step_size
        if not step.is_contiguous():
            step = step.contiguous()
            ad.step_size = step
        div = None
        if not warm:
            div = torch.zeros((k, self.num_chains), dtype=torch.int8, device=self._z.device)
        if samples is not None:
            samples = samples[:k]
        if self._launch_hook is not None:
            self._launch_hook()
        out = kernels.nuts_gaussian_run(
            self._z, self._pe, self._grad, self._Lambda, self.mass_matrix_adapter.inverse_mass_matrix, step,
            self._max_tree_depth, self.use_multinomial_sampling, self._seed, self._t, k,
            self.chain_offset, da_state=da, target_accept=ad.target_accept_prob, welford=wf,
            welford_n0=wf_n0, samples=samples, mean_accept=self._mean_accept_prob,
            mean_n0=mean_n0, counters=self._counters, count_accepts=not warm, div_flags=div)
        self._last_stats = out
        self._t += k
        if warm:
            if da is not None:
                ad.load_da_state(da, k)
            if wf is not None:
                ad.load_welford_state(wf, k)
            ad.finish_span(self._t, self._z)
        else:
            self._divergences.extend(div[i] for i in range(k))
        return k

    @property
    def num_leapfrog_steps(self):
        return int(self._n_leapfrog_total.item()) + int(self._counters[0].sum().item())

    # ---- asynchronous chains on the tree path ------------------------------------------------
    def _persistent_step_and_mass(self, step, inv_mass):
        """The tree kernels of a captured round read step / inverse mass through pointers:
        persistent buffers, refreshed in place when adaptation hands over new tensors."""
        if getattr(self, "_step_buf", None) is None or self._step_buf.shape != step.shape \
                or self._mass_buf.shape != inv_mass.shape:
            self._step_buf, self._mass_buf = step.clone(), inv_mass.clone()
            self._round_graph, self._round_calls = None, 0
            self._span_graph, self._span_calls, self._span_graphs = None, 0, {}
            self._tree = None
        else:
            if step is not self._step_buf:
                self._step_buf.copy_(step)
            if inv_mass is not self._mass_buf:
                self._mass_buf.copy_(inv_mass)
        return self._step_buf, self._mass_buf

    def _span(self, k, samples=None):
        """``k`` transitions of every chain as ONE span of asynchronous rounds
        (kernels.NutsTree.run_*): a chain that finishes a tree adapts, stores its draw and starts
        its next tree in the same launch; the host replays a graph of rounds and counts finished
        chains.  Same chains as ``_transition`` (the Philox keys do not know the schedule)."""
        ad = self._adapter
        warm = self._t < self._warmup_steps
        T = kernels.NutsTree
        flags, da, wf, wf_n0 = 0, None, None, 0
        if warm:
            span, adapting = ad.bulk_span(self._t)
            k = min(k, span)
            if adapting and ad.adapt_step_size:
                da, flags = ad.da_state(), flags | T.RUN_ADAPT_STEP
            if adapting and ad.in_mass_phase():
                (wf, wf_n0), flags = ad.welford_state(), flags | T.RUN_WELFORD
            mean_n0 = self._t
        else:
            mean_n0, flags = self._t - self._warmup_steps, flags | T.RUN_COUNT_ACCEPTS
        C, D = self._z.shape
        step, inv_mass = self._persistent_step_and_mass(ad.step_size.contiguous(),
                                                       self._mm_eff.inverse_mass_matrix)
        tree = self._tree
        if tree is None:
            tree = self._tree = kernels.NutsTree(self._z, self._pe, self._grad, inv_mass, step,
                                                 self._max_tree_depth, self.use_multinomial_sampling,
                                                 self._seed, self.chain_offset)
        elif tree.inv_mass is not inv_mass or tree.step is not step:
            tree.inv_mass, tree.step = inv_mass, step
            tree.im_stride = tree.D if inv_mass.dim() == 2 else 0
            self._span_graph, self._span_graphs = None, {}
        if self._da_buf is None:
            self._da_buf = torch.zeros((C, 5), dtype=self._z.dtype, device=self._z.device)
            self._wf_buf = torch.zeros((C, 2, D), dtype=self._z.dtype, device=self._z.device)
        if da is not None:
            self._da_buf.copy_(da)
        if wf is not None:
            self._wf_buf.copy_(wf)
        div = None
        if not warm:
            div = torch.zeros((k, C), dtype=torch.int8, device=self._z.device)
        if samples is not None:
            samples = samples[:k]
        tree.set_span(self._t, k, mean_n0=mean_n0, welford_n0=wf_n0, flags=flags, samples=samples,
                      div_flags=div)
        tree.run_begin()
        self._run_rounds(tree)
        self._last_stats = tree.stats()
        self._t += k
        if warm:
            if da is not None:
                ad.step_size = step.clone()
                ad.load_da_state(self._da_buf.clone(), k)
            if wf is not None:
                ad.load_welford_state(self._wf_buf.clone(), k)
            ad.finish_span(self._t, self._z)
        else:
            self._divergences.extend(div[i] for i in range(k))
        return k

    def _span_round(self, tree, potential, slots=None):
        prog = self._direct
        if prog is not None:
            # a flat model: the observed site's kernel at the (site-major) cursors, everything else inside
            # the tree kernel -- three launches, no handlers, no autograd (infer/mcmc/direct.py)
            n_slots = self.num_chains if slots[0] is None else slots[0].numel()
            ll, ext = prog.glm_round(slots[1], n_slots)
            tree.run_advance_direct(prog, ll, ext, self._da_buf, self._adapter.target_accept_prob,
                                    self._wf_buf, self._mean_accept_prob, self._counters, slots)
            return ll, ext
        pe, grad = potential(tree.zq if slots is None else slots[1])
        tree.run_advance(pe.detach().contiguous(), grad.detach().contiguous(), self._da_buf,
                         self._adapter.target_accept_prob, self._wf_buf, self._mean_accept_prob,
                         self._counters, slots=slots)
        return pe, grad

    def _want_capture(self):
        if self._span_failed or not self._z.is_cuda:
            return False
        if self._jit_compile:
            return True
        return CAPTURE_ROUNDS is True or CAPTURE_ROUNDS == "auto"

    def _slot_sizes(self):
        """Round sizes a span may shrink to: C, C/2, C/4, ... down to ``min_slots``."""
        C, sizes = self.num_chains, []
        n = C
        while self.compact_chains and n % 2 == 0 and n // 2 >= self.min_slots:
            n //= 2
            sizes.append(n)
        return sizes

    def _run_rounds(self, tree):
        """Rounds until every chain has completed the span.  After a few eager rounds a block of
        ``rounds_per_replay`` rounds is captured into ONE hipGraph (the span's parameters reach the
        kernels through device memory, so the graph serves every later span too); the tree kernel
        of the round that completes the span sets the abort word of a step gate, and the
        gate-aware kernels of the rounds still queued behind it return at once.

        Per-chain step sizes differ, so chains complete a span at different times; once at most half
        of the round's slots hold a chain that is still building trees the round is COMPACTED: the
        potential is evaluated at the cursors of the active chains only (kernels.NutsTree.compact; one
        captured graph per round size C/2, C/4, ...)."""
        base = getattr(self._potential, "base", self._potential)
        C = self.num_chains
        sizes = self._slot_sizes()
        cur, slots = C, None
        if self._direct is not None:
            slots = tree.compact(C, self._direct)      # the full round's cursors in the site-major layout
        polls = 0
        while True:
            graph = self._span_graphs.get(cur)
            if graph is None and self._want_capture() and self._span_calls >= 4:
                graph = self._capture_span(tree, base, slots, cur)
            if graph is not None:
                graph.replay()
                self._span_replays += 1
                self._span_rounds += self.rounds_per_replay * cur
            else:
                for _ in range(self.sync_every):
                    self._span_round(tree, self._potential, slots)
                    self._span_calls += 1
                self._span_rounds += self.sync_every * cur
            polls += 1
            done = tree.chains_done()
            if done >= C:
                break
            active = C - done
            fit = [n for n in sizes if n >= active and n < cur]
            if fit:
                cur = min(fit)
                slots = tree.compact(cur, self._direct)
                self._span_compactions += 1
            if polls > (1 << 22):
                raise RuntimeError("pyro_amd: NUTS span did not complete")

    def _capture_span(self, tree, base, slots=None, size=None):
        import os
        import warnings
        from ...ops import fuser
        lib = kernels._lib.load()
        try:
            # one eager round under the fuser: the element-wise kernels of a round (the glue between the
            # model's fused sites and their autograd duals) are generated and compiled before the capture
            with fuser.scope():
                self._span_round(tree, base, slots)
            for attempt in (0, 1):
                compiled = fuser.STATS["loaded"]
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                keep = []
                kernels.check(lib.pa_gate_scope(kernels._ptr(tree.gate)))
                blocks = fuser.RtcBlocks()
                try:
                    with capture_scope(), blocks, torch.cuda.graph(graph):
                        for _ in range(self.rounds_per_replay):
                            with fuser.scope():
                                keep.append(self._span_round(tree, base, slots))
                    break
                except Exception:  # noqa: BLE001
                    # (a kernel generated during the capture loaded its module there: cached now, once more)
                    blocks.free()
                    if attempt == 1 or fuser.STATS["loaded"] == compiled:
                        raise
                finally:
                    lib.pa_gate_scope(None)
            self._span_graphs[size if size is not None else self.num_chains] = graph
            self._span_graph = graph
            self._span_keep = (getattr(self, "_span_keep", None) or []) + [keep, blocks]
            return graph
        except Exception as e:  # noqa: BLE001  (anything that synchronises inside the capture)
            if os.environ.get("PYRO_AMD_DEBUG_GRAPH"):
                raise
            if self._jit_compile:
                warnings.warn("pyro_amd: hipGraph capture of the NUTS rounds failed ({}: {}); "
                              "continuing with eager rounds".format(type(e).__name__, e))
            self._span_failed = True
            return None

    def _tree_transition(self, t, step, inv_mass):
        tree = self._tree
        replayable = bool(self._jit_compile)
        if replayable or self._step_buf is not None:
            step, inv_mass = self._persistent_step_and_mass(step, inv_mass)
            tree = self._tree
        if tree is None or tree.inv_mass is not inv_mass or tree.step is not step:
            if tree is None:
                tree = kernels.NutsTree(self._z, self._pe, self._grad, inv_mass, step,
                                        self._max_tree_depth, self.use_multinomial_sampling,
                                        self._seed, self.chain_offset)
                self._tree = tree
            else:   # adaptation replaced the step-size / mass tensors: re-point, keep workspace
                tree.inv_mass, tree.step = inv_mass, step
                tree.im_stride = tree.D if inv_mass.dim() == 2 else 0
        tree.begin(t)
        it = 0
        max_iter = (1 << self._max_tree_depth) + 1
        while True:
            if replayable:
                self._round(tree)
            else:
                pe, grad = self._potential(tree.zq)
                tree.advance(pe.contiguous(), grad.contiguous())
            it += 1
            if it >= max_iter:
                break
            # chains can only finish at leaf counts 2^j - 1 or by a U-turn/divergence inside a
            # doubling; polling every few leaves keeps the host out of the loop
            if (it & (it + 1)) == 0 or it % self.sync_every == 0:
                if tree.n_active() == 0:
                    break
        return tree.stats()

    def _round(self, tree):
        """One tree round = potential at the cursor + tree bookkeeping.  With jit_compile the pair
        is captured into ONE hipGraph (after a few eager rounds) and replayed: the transition index
        reaches the tree kernel through device memory, step size and mass through persistent
        buffers, so the same graph serves every leapfrog of every transition."""
        base = getattr(self._potential, "base", self._potential)
        graph = getattr(self, "_round_graph", None)
        if graph is None and not getattr(self, "_round_failed", False):
            self._round_calls = getattr(self, "_round_calls", 0) + 1
            if self._round_calls > 8:
                try:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with capture_scope(), torch.cuda.graph(graph):
                        pe, grad = base(tree.zq)
                        tree.advance_replayable(pe.detach().contiguous(), grad.detach().contiguous())
                    self._round_graph = graph
                    self._round_keep = (pe, grad)
                except Exception as e:  # noqa: BLE001
                    import os
                    import warnings
                    if os.environ.get("PYRO_AMD_DEBUG_GRAPH"):
                        raise
                    warnings.warn("pyro_amd: hipGraph capture of the NUTS round failed ({}: {}); "
                                  "continuing with eager rounds".format(type(e).__name__, e))
                    self._round_failed, graph = True, None
        if graph is not None:
            graph.replay()
            return
        pe, grad = self._potential(tree.zq)
        tree.advance_replayable(pe.contiguous(), grad.contiguous())

    def logging(self):
        out = super().logging()
        return OrderedDict(list(out.items()))

    def diagnostics(self):
        if getattr(self, "_empty", False):
            return {}
        self._accept_cnt = self._accept_cnt + self._counters[2]
        self._counters[2].zero_()
        out = super().diagnostics()
        if self._t:
            depth = float(self._tree_depth_sum.item()) + float(self._counters[1].sum().item())
            out["mean tree depth"] = depth / (self._t * self.num_chains)
        return out
