"""TEST INFRASTRUCTURE -- the ITERATIVE per-chain NUTS tree state machine (the formulation of
pyro_amd/csrc/nuts_tree.hip) restated in numpy, one python object per chain batch.

Purpose: (1) tests/test_oracle_vs_golden.py checks it against the RECURSIVE restatement of the
reference (oracle/nuts.py, itself pinned against unmodified pyro's NUTS.sample) -- this pins the
recursion -> iteration transformation on the CPU; (2) tests/oracle_backend.py uses it as the
stand-in for kernels.NutsTree so the host MCMC driver can be exercised without a GPU.
Reference lines: pyro/infer/mcmc/nuts.py:184-522 (see oracle/nuts.py for the line map).
"""
import math

import numpy as np

from .nuts import MAX_SLICED_ENERGY, KeyedDraws, logaddexp


class _Chain:
    pass


class NutsTreeOracle:
    def __init__(self, z, pe, grad, inv_mass, step, max_tree_depth=10, use_multinomial=True,
                 seed=0, chain_offset=0, dtype=np.float64):
        self.z, self.pe, self.grad = z, pe, grad          # numpy arrays, updated in place
        self.inv_mass, self.step = inv_mass, step
        self.C, self.D = z.shape
        self.max_depth, self.multinomial = max_tree_depth, use_multinomial
        self.seed, self.chain_offset = seed, chain_offset
        self.dt = np.dtype(dtype).type
        self.zq = np.zeros_like(z)
        self.rq = np.zeros_like(z)
        self.accept_prob = np.zeros(self.C, dtype=z.dtype)
        self.ints = np.zeros((4, self.C), dtype=np.int32)
        self.chains = [None] * self.C

    def _v(self, c):
        v = self.inv_mass if self.inv_mass.ndim == 1 else self.inv_mass[c]
        return v.astype(self.dt)

    def begin(self, t):
        for c in range(self.C):
            self._begin_chain(c, t)

    def _begin_chain(self, c, t):
        dt = self.dt
        s = _Chain()
        s.draws = KeyedDraws(self.seed, self.chain_offset + c, t, dt)
        v = self._v(c)
        s.v, s.sq = v, np.sqrt(v)
        ru0 = np.asarray(s.draws.momentum(self.D), dtype=dt)
        r0 = ru0 * (dt(1) / s.sq)
        s.energy_current = dt(0.5) * dt(ru0.dot(ru0)) + dt(self.pe[c])
        s.log_slice = -s.energy_current if self.multinomial else \
            -s.energy_current - dt(s.draws.slice_exp())
        z0, g0 = self.z[c].astype(dt).copy(), self.grad[c].astype(dt).copy()
        s.edges = [(z0, r0, g0), (z0, r0, g0)]   # left, right
        s.r_sum = ru0.copy()
        s.tree_weight = dt(0) if self.multinomial else dt(1)
        s.sum_accept, s.num_prop = dt(0), 0
        s.accepted = s.diverged = False
        s.depth, s.leaf = 0, 0
        s.stack = {}
        s.active = True
        s.eps = dt(self.step[c])
        s.dir = 1 if s.draws.direction(0) < 0.5 else -1
        self._start_leapfrog(c, s, *s.edges[1 if s.dir == 1 else 0])
        self.chains[c] = s

    def _start_leapfrog(self, c, s, z, r, g):
        eps_d = s.eps if s.dir == 1 else -s.eps
        r = r + self.dt(0.5) * eps_d * (-g)
        z = z + eps_d * (s.v * r)
        self.zq[c], self.rq[c] = z, r

    def _is_turning(self, r_first, r_last, r_sum):
        rho = r_sum - (r_first + r_last) / 2
        return bool(r_first.dot(rho) <= 0) or bool(r_last.dot(rho) <= 0)

    def advance(self, peq, gq):
        for c in range(self.C):
            if self.chains[c].active:
                self._advance_chain(c, peq, gq)

    def _advance_chain(self, c, peq, gq):
        """One leaf of chain c's tree; True when the transition is over."""
        dt = self.dt
        s = self.chains[c]
        eps_d = s.eps if s.dir == 1 else -s.eps
        zq, g = self.zq[c].astype(dt), np.asarray(gq[c], dtype=dt)
        rq = self.rq[c].astype(dt) + dt(0.5) * eps_d * (-g)
        pe_q = dt(peq[c])
        ruq = rq * s.sq
        energy_new = pe_q + dt(0.5) * dt(ruq.dot(ruq))
        if math.isnan(energy_new):
            energy_new = dt(math.inf)
        sliced = energy_new + s.log_slice
        with np.errstate(over="ignore"):
            ap = min(float(np.exp(-(energy_new - s.energy_current))), 1.0)
        s.sum_accept += dt(ap)
        s.num_prop += 1
        b_first, b_sum, b_prop, b_propg, b_pe = ruq, ruq, zq, g, pe_q
        b_w = -sliced if self.multinomial else (1.0 if sliced <= 0 else 0.0)
        j, i = s.depth, s.leaf
        finished = turning = False
        if sliced > MAX_SLICED_ENERGY:
            s.diverged = True
            finished = True
        else:
            k = 0
            while (i >> k) & 1:
                h_first, h_sum, h_prop, h_propg, h_w, h_pe = s.stack[k]
                if self.multinomial:
                    w = logaddexp(h_w, b_w)
                    prob_other = math.exp(b_w - w) if w > -math.inf else float("nan")
                else:
                    w = h_w + b_w
                    prob_other = b_w / w if w > 0 else 0.0
                u = s.draws.merge(j, k + 1, i >> (k + 1))
                if not (u < prob_other):
                    b_prop, b_propg, b_pe = h_prop, h_propg, h_pe
                b_first, b_sum, b_w = h_first, h_sum + b_sum, w
                k += 1
                if self._is_turning(b_first, ruq, b_sum):
                    turning = True
                    break
            if turning:
                finished = True
            elif i + 1 < (1 << j):
                s.stack[k] = (b_first, b_sum, b_prop, b_propg, b_w, b_pe)
                s.leaf = i + 1
                self._start_leapfrog(c, s, zq, rq, g)
            else:
                e_dir = 1 if s.dir == 1 else 0
                s.edges[e_dir] = (zq, rq, g)
                s.depth += 1
                new_prob = math.exp(min(b_w - s.tree_weight, 700.0)) if self.multinomial else \
                    b_w / s.tree_weight
                if s.draws.accept(j) < new_prob:
                    s.accepted = True
                    self.z[c], self.grad[c], self.pe[c] = b_prop, b_propg, b_pe
                s.r_sum = s.r_sum + b_sum
                ru_other = s.edges[1 - e_dir][1] * s.sq
                if self._is_turning(ru_other, ruq, s.r_sum):
                    finished = True
                else:
                    s.tree_weight = logaddexp(s.tree_weight, b_w) if self.multinomial else \
                        s.tree_weight + b_w
                    if s.depth >= self.max_depth:
                        finished = True
                    else:
                        s.dir = 1 if s.draws.direction(s.depth) < 0.5 else -1
                        s.leaf = 0
                        self._start_leapfrog(c, s, *s.edges[1 if s.dir == 1 else 0])
        if finished:
            s.active = False
            self.accept_prob[c] = s.sum_accept / s.num_prop
            self.ints[:, c] = (s.num_prop, s.depth, int(s.diverged), int(s.accepted))
        return finished

    def n_active(self):
        return sum(1 for s in self.chains if s.active)

    # ---- asynchronous chains (pa_nuts_tree_run_begin / _advance): a span of K transitions per chain;
    # a chain whose tree finishes does the per-transition bookkeeping of HMC._after_transition /
    # WarmupAdapter.step (pyro/infer/mcmc/hmc.py:425-438, adaptation.py:166-185: DualAveraging.step,
    # pyro/ops/dual_averaging.py:55-78; WelfordCovariance.update, pyro/ops/welford.py:27-38) and begins
    # its next transition in the same round -------------------------------------------------------
    ADAPT_STEP, WELFORD, COUNT_ACCEPTS = 1, 2, 4

    def run_begin(self, t0, K, mean_n0=0, welford_n0=0, flags=0, samples=None, div_flags=None, row0=0):
        self.span = dict(t0=int(t0), K=int(K), mean_n0=int(mean_n0), wf_n0=int(welford_n0),
                         flags=int(flags), samples=samples, div=div_flags, row0=int(row0))
        self.tc = np.zeros(self.C, dtype=np.int64)
        self.n_done = 0
        self.begin(self.span["t0"])

    def run_advance(self, peq, gq, da, target_accept, wf, mean_accept, counters,
                    da_t0=10.0, da_kappa=0.75, da_gamma=0.05):
        """da [C,5] {x_avg, g_avg, t, prox_center, x_t}, wf [C,2,D] {mean, m2}, mean_accept [C],
        counters [3,C] int64: numpy arrays updated in place; self.step too (dual averaging)."""
        dt, sp = self.dt, self.span
        for c in range(self.C):
            if not self.chains[c].active:
                continue
            if not self._advance_chain(c, peq, gq):
                continue
            k = int(self.tc[c])
            s = self.chains[c]
            ap = dt(self.accept_prob[c])
            if math.isnan(ap):
                ap = dt(0)
            if sp["flags"] & self.ADAPT_STEP:
                x_avg, g_avg, t, prox = dt(da[c, 0]), dt(da[c, 1]), dt(da[c, 2]), dt(da[c, 3])
                g = dt(target_accept) - ap
                t = t + dt(1)
                g_avg = (dt(1) - dt(1) / (t + dt(da_t0))) * g_avg + g / (t + dt(da_t0))
                x_t = prox - dt(np.sqrt(t)) / dt(da_gamma) * g_avg
                weight = dt(np.exp(-dt(da_kappa) * dt(np.log(t))))
                x_avg = (dt(1) - weight) * x_avg + weight * x_t
                da[c, 0], da[c, 1], da[c, 2], da[c, 4] = x_avg, g_avg, t, x_t
                self.step[c] = dt(np.exp(x_t))
            zc = self.z[c].astype(dt)
            if sp["flags"] & self.WELFORD:
                n = dt(sp["wf_n0"] + k + 1)
                pre = zc - wf[c, 0].astype(dt)
                mean = wf[c, 0].astype(dt) + pre / n
                wf[c, 1] = wf[c, 1].astype(dt) + pre * (zc - mean)
                wf[c, 0] = mean
            if sp["samples"] is not None:
                sp["samples"][sp["row0"] + k, c] = zc
            mean_accept[c] = dt(mean_accept[c]) + (ap - dt(mean_accept[c])) / dt(sp["mean_n0"] + k + 1)
            counters[0, c] += s.num_prop
            counters[1, c] += s.depth
            if sp["flags"] & self.COUNT_ACCEPTS:
                counters[2, c] += int(s.accepted)
                if sp["div"] is not None:
                    sp["div"][sp["row0"] + k, c] = int(s.diverged)
            self.tc[c] = k + 1
            if k + 1 < sp["K"]:
                self._begin_chain(c, sp["t0"] + k + 1)
            else:
                self.n_done += 1

    def span_done(self):
        return self.n_done >= self.C
