"""MCMC driver (reference: pyro/infer/mcmc/api.py:405-651 MCMC, util.py:507-616 diagnostics /
summary / select_samples).

Same constructor and methods.  ``num_chains`` chains run as ONE vectorised batch on this
process's GPU (the reference forks one process per chain, api.py:239-351).  When
torch.distributed is initialised the chains are sharded over the ranks (chains are independent:
no communication while sampling), each rank draws from its own keyed Philox streams
(global chain index), and ``get_samples`` all-gathers the per-rank samples.
"""
import warnings
from collections import OrderedDict

import torch
import torch.distributed as dist

from ... import poutine
from ...ops import stats
from .hmc import HMC


def select_samples(samples, num_samples=None, group_by_chain=False):
    """reference: pyro/infer/mcmc/util.py:740-770."""
    if num_samples is None:
        if not group_by_chain:
            samples = {k: v.reshape((-1,) + v.shape[2:]) for k, v in samples.items()}
        return samples
    if not samples:
        raise ValueError("No samples found from MCMC run.")
    first = next(iter(samples.values()))
    if group_by_chain:
        batch_dim = 1
    else:
        samples = {k: v.reshape((-1,) + v.shape[2:]) for k, v in samples.items()}
        first = next(iter(samples.values()))
        batch_dim = 0
    idxs = torch.randint(0, first.size(batch_dim), size=(num_samples,), device=first.device)
    return {k: v.index_select(batch_dim, idxs) for k, v in samples.items()}


def diagnostics(samples, group_by_chain=True):
    """n_eff and r_hat per site; samples[name]: [chain, sample, ...]."""
    out = OrderedDict()
    for site, support in samples.items():
        if not group_by_chain:
            support = support.unsqueeze(0)
        d = OrderedDict()
        try:
            d["n_eff"] = stats.effective_sample_size(support)
        except (AssertionError, RuntimeError):
            d["n_eff"] = torch.full(support.shape[2:], float("nan"))
        try:
            d["r_hat"] = stats.split_gelman_rubin(support)
        except (AssertionError, RuntimeError):
            d["r_hat"] = torch.full(support.shape[2:], float("nan"))
        out[site] = d
    return out


def summary(samples, prob=0.9, group_by_chain=True):
    """mean / std / median / credible interval / n_eff / r_hat per site
    (reference: util.py:531-570)."""
    if not group_by_chain:
        samples = {k: v.unsqueeze(0) for k, v in samples.items()}
    out = {}
    for name, value in samples.items():
        flat = value.reshape((-1,) + value.shape[2:])
        lo, hi = stats.hpdi(flat, prob, dim=0)
        d = OrderedDict([("mean", flat.mean(0)), ("std", flat.std(0)),
                         ("median", flat.median(0)[0]),
                         ("{:.1f}%".format(50 * (1 - prob)), lo),
                         ("{:.1f}%".format(50 * (1 + prob)), hi)])
        diag = diagnostics({name: value})[name]
        d["n_eff"], d["r_hat"] = diag["n_eff"], diag["r_hat"]
        out[name] = d
    return out


def print_summary(samples, prob=0.9, group_by_chain=True, max_rows=40):
    s = summary(samples, prob, group_by_chain)
    if not s:
        return
    cols = list(next(iter(s.values())).keys())
    print("{:>16}".format("") + "".join("{:>10}".format(c) for c in cols))
    rows = 0
    for name, d in s.items():
        n = d["mean"].numel()
        for i in range(n):
            if rows >= max_rows:
                print("  ... ({} more rows)".format(sum(v["mean"].numel() for v in s.values())
                                                    - rows))
                return
            label = name if n == 1 else "{}[{}]".format(name, i)
            print("{:>16}".format(label[:16]) + "".join(
                "{:>10.2f}".format(float(d[c].reshape(-1)[i])) for c in cols))
            rows += 1


class MCMC:
    def __init__(self, kernel, num_samples, warmup_steps=None, initial_params=None, num_chains=1,
                 hook_fn=None, mp_context=None, disable_progbar=True, disable_validation=True,
                 transforms=None, save_params=None, shard_chains=True):
        self.kernel = kernel
        self.num_samples = num_samples
        self.warmup_steps = num_samples if warmup_steps is None else warmup_steps  # Stan
        self.num_chains = num_chains
        self.transforms = transforms
        self.hook_fn = hook_fn
        self.disable_validation = disable_validation
        self.save_params = save_params
        self._samples = None
        self._diagnostics = None
        self._world = dist.get_world_size() if (shard_chains and dist.is_initialized()) else 1
        self._rank = dist.get_rank() if self._world > 1 else 0
        if num_chains % self._world != 0:
            raise ValueError("num_chains={} is not divisible by the number of ranks {}".format(
                num_chains, self._world))
        self._local_chains = num_chains // self._world
        if initial_params is not None:
            if num_chains > 1:
                for v in initial_params.values():
                    if v.shape[0] != num_chains:
                        raise ValueError("The leading dimension of tensors in `initial_params` "
                                         "must match the number of chains.")
                if self._world > 1:
                    lo = self._rank * self._local_chains
                    initial_params = {k: v[lo:lo + self._local_chains]
                                      for k, v in initial_params.items()}
            kernel.initial_params = initial_params
        self._given_initial_params = initial_params
        # mp_context (how the reference forks one process per chain) has nothing to configure here:
        # the chains are one vectorised batch on the device; accepted and ignored, silently, so that
        # scripts written for the reference run unchanged

    def run(self, *args, **kwargs):
        k = self.kernel
        k.num_chains = self._local_chains
        k.chain_offset = self._rank * self._local_chains
        if self._local_chains == 1 and self.num_chains > 1:
            k._force_batched = True
        from ...primitives import validation_enabled
        with validation_enabled(False if self.disable_validation else
                                poutine.settings.validation_enabled()):
            k.setup(self.warmup_steps, *args, **kwargs)
            if self.transforms is None:
                self.transforms = getattr(k, "transforms", None) or {}
            fast = isinstance(k, HMC) and not getattr(k, "_empty", False)
            S, C = self.num_samples, self._local_chains
            params = k.initial_params
            bulk = fast and bool(getattr(k, "bulk_ready", False)) and self.hook_fn is None
            per_chain_diag = None

            def hook(stage, i):
                if self.hook_fn is not None:
                    cur = k._layout.unflatten(k._position(), k._batched) if fast else {}
                    self.hook_fn(k, cur, stage, i)

            if bulk:
                # persistent launches: many transitions per kernel, samples written by the kernel
                buf = torch.empty((S, C, k._layout.D), dtype=k._z.dtype, device=k._z.device)
                done = 0
                while done < self.warmup_steps:
                    done += k._transition_many(self.warmup_steps - done)
                k.end_warmup()
                done = 0
                while done < S:
                    done += k._transition_many(S - done, samples=buf[done:])
                flat = buf.transpose(0, 1)    # [C, S, D]
                z_acc = {}
                for name in k._layout.names:
                    a, b = k._layout.slices[name]
                    z_acc[name] = flat[:, :, a:b].reshape((C, S) + tuple(k._layout.shapes[name]))
            elif fast:
                buf = torch.empty((S, C, k._layout.D), dtype=k._z.dtype, device=k._z.device)
                for i in range(self.warmup_steps):
                    k._transition()
                    hook("Warmup", i)
                k.end_warmup()
                for i in range(S):
                    k._transition()
                    buf[i].copy_(k._position())
                    hook("Sample", i)
                flat = buf.transpose(0, 1)    # [C, S, D]
                z_acc = {}
                for name in k._layout.names:
                    a, b = k._layout.slices[name]
                    z_acc[name] = flat[:, :, a:b].reshape((C, S) + tuple(k._layout.shapes[name]))
            elif isinstance(k, HMC):
                # a model without continuous latent sites: nothing moves, the hooks still run
                for i in range(self.warmup_steps):
                    hook("Warmup", i)
                for i in range(S):
                    hook("Sample", i)
                z_acc = {}
            else:
                # a user-written MCMCKernel knows nothing of vectorised chains: one chain after the
                # other through the MCMCKernel interface, as the reference's sequential sampler does
                # (api.py:212-237); the first chain reuses the setup() above
                given = self._given_initial_params
                chains, per_chain_diag = [], []
                for c in range(C):
                    if c > 0 or (given is not None and C > 1):
                        if given is not None:
                            k.initial_params = {n: (v[c] if C > 1 else v) for n, v in given.items()}
                        k.setup(self.warmup_steps, *args, **kwargs)
                    params = k.initial_params
                    for i in range(self.warmup_steps):
                        params = k.sample(params)
                        if self.hook_fn is not None:
                            self.hook_fn(k, params, "Warmup", i)
                    acc = None
                    for i in range(S):
                        params = k.sample(params)
                        if self.hook_fn is not None:
                            self.hook_fn(k, params, "Sample", i)
                        if acc is None:
                            acc = {n: [] for n in params}
                        for n, v in params.items():
                            acc[n].append(v.detach().clone())
                    chains.append({n: torch.stack(v, dim=0) for n, v in (acc or {}).items()})
                    per_chain_diag.append(k.diagnostics())
                    if c + 1 < C:
                        k.cleanup()
                z_acc = {n: torch.stack([ch[n] for ch in chains], dim=0) for n in chains[0]}
            if self.save_params is not None:
                z_acc = {n: v for n, v in z_acc.items() if n in self.save_params}
            # back to the constrained space of the model (api.py:600-603)
            for name, z in z_acc.items():
                if name in self.transforms:
                    z_acc[name] = self.transforms[name].inv(z)
            self._local_samples = z_acc
            self._samples = self._gather(z_acc)
            if per_chain_diag is not None:
                # {key: {"chain i": value}} as the reference's driver merges them (api.py:560-575)
                merged = {}
                for c, d in enumerate(per_chain_diag):
                    for key, value in (d or {}).items():
                        merged.setdefault(key, {})["chain {}".format(c)] = value
                self._diagnostics = merged
            else:
                self._diagnostics = k.diagnostics()
        # the reference's driver ends a run with kernel.cleanup() (api.py:170), which there drops the
        # jit-compiled potential; here the run's statistics stay readable on the kernel object
        # (leapfrog counts, adapted step sizes) and only the captured graphs are released
        release = getattr(k, "release_graphs", None)
        if release is not None:
            release()
        return self

    def _gather(self, z_acc):
        if self._world == 1:
            return z_acc
        out = {}
        for name in sorted(z_acc):
            v = z_acc[name].contiguous()
            parts = [torch.empty_like(v) for _ in range(self._world)]
            dist.all_gather(parts, v)
            out[name] = torch.cat(parts, dim=0)
        return out

    def get_samples(self, num_samples=None, group_by_chain=False):
        return select_samples(self._samples, num_samples, group_by_chain)

    def diagnostics(self):
        diag = diagnostics(self._samples)
        for name, value in (self._diagnostics or {}).items():
            diag[name] = value
        return diag

    def summary(self, prob=0.9):
        print_summary(self._samples, prob=prob)
        if self._diagnostics and "divergences" in self._diagnostics:
            print("Number of divergences: {}".format(
                sum(len(v) for v in self._diagnostics["divergences"].values())))
