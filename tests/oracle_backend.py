"""TEST-ONLY seam: replaces the functions of pyro_amd.kernels by numpy-oracle implementations so
the HOST logic (handlers, ELBO assembly, optimizers, MCMC drivers) can be exercised on a machine
without a GPU.  Installed by the ``oracle_backend`` fixture through monkeypatch; the product
package never imports this module and has no CPU path of its own.
"""
import numpy as np
import torch

from oracle import adam as o_adam
from oracle import dists as o_dists
from oracle import glm as o_glm
from oracle import integrator as o_int
from oracle import lda as o_lda
from oracle import nuts as o_nuts
from oracle import nuts_tree as o_tree
from oracle import philox as o_philox


def _np(t):
    if t is None:
        return None
    import pyro_amd.kernels as k
    for hook in k._PTR_HOOKS:          # (a recorder scope: what the stand-in reads is materialised first)
        hook(t)
    return t.detach().cpu().numpy()


def _bc(t, rows, cols):
    return None if t is None else np.broadcast_to(_np(t), (rows, cols))


def philox_normal(shape, dtype, device, seed, offset, offset_dev=None):
    n = int(np.prod(shape)) if len(shape) else 1
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    return torch.as_tensor(o_philox.normal(n, np_dt, seed, offset).reshape(shape), device=device)


def philox_uniform(shape, dtype, device, seed, offset, offset_dev=None):
    n = int(np.prod(shape)) if len(shape) else 1
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    return torch.as_tensor(o_philox.uniform(n, np_dt, seed, offset).reshape(shape), device=device)


def gamma_rsample(alpha, rows, cols, seed, offset, offset_dev=None, want_grad=True):
    from oracle import gamma as o_gamma
    a = np.ascontiguousarray(_bc(alpha, rows, cols), dtype=np.float64)
    np_dt = np.float32 if alpha.dtype == torch.float32 else np.float64
    x = o_gamma.standard_gamma(a, seed, offset)
    x = np.maximum(x, np.finfo(np_dt).tiny).astype(np_dt)
    out = torch.as_tensor(x, device=alpha.device)
    if not want_grad:
        return out, None
    g = o_gamma.implicit_grad(a, x.astype(np.float64)).astype(np_dt)
    return out, torch.as_tensor(g, device=alpha.device)


def dist_log_prob(dist_id, value, p0, p1, rows, cols):
    out = o_dists.LOG_PROB[dist_id](_bc(value, rows, cols).astype(np.float64),
                                    _bc(p0, rows, cols).astype(np.float64),
                                    None if p1 is None else _bc(p1, rows, cols).astype(np.float64))
    return torch.as_tensor(np.ascontiguousarray(out), dtype=value.dtype)


def dist_log_prob_sum(dist_id, value, p0, p1, mask, scale, rows, cols, want_total=False):
    out = o_dists.log_prob_sum(dist_id, _bc(value, rows, cols).astype(np.float64),
                               _bc(p0, rows, cols).astype(np.float64),
                               None if p1 is None else _bc(p1, rows, cols).astype(np.float64),
                               _bc(mask, rows, cols), scale)
    rs = torch.as_tensor(np.ascontiguousarray(out), dtype=value.dtype)
    return (rs, rs.sum()) if want_total else rs


def dist_log_prob_grad(dist_id, g, value, p0, p1, mask, scale, rows, cols, need):
    v = _bc(value, rows, cols).astype(np.float64)
    a = _bc(p0, rows, cols).astype(np.float64)
    b = None if p1 is None else _bc(p1, rows, cols).astype(np.float64)
    dv, da, db = o_dists.log_prob_grad(dist_id, v, a, b)
    w = _bc(g, rows, cols).astype(np.float64) * scale
    m = _bc(mask, rows, cols)
    outs = []
    for d, n in zip((dv, da, db), need):
        if not n:
            outs.append(None)
            continue
        x = w * np.broadcast_to(d, (rows, cols))
        if m is not None:
            x = np.where(m, x, 0.0)
        outs.append(torch.as_tensor(np.ascontiguousarray(x), dtype=value.dtype))
    return outs


def _bcn(t, shape):
    return None if t is None else np.broadcast_to(_np(t), shape)


def dist_log_prob_sum_nd(dist_id, shape, value, p0, p1, mask, scale):
    lp = o_dists.LOG_PROB[dist_id](_bcn(value, shape).astype(np.float64),
                                   _bcn(p0, shape).astype(np.float64),
                                   None if p1 is None else _bcn(p1, shape).astype(np.float64)) * scale
    m = _bcn(mask, shape)
    if m is not None:
        lp = np.where(m, lp, 0.0)
    return torch.as_tensor(lp.sum(), dtype=value.dtype)


def dist_log_prob_grad_nd(dist_id, shape, g, value, p0, p1, mask, scale, need):
    v = _bcn(value, shape).astype(np.float64)
    a = _bcn(p0, shape).astype(np.float64)
    b = None if p1 is None else _bcn(p1, shape).astype(np.float64)
    dv, da, db = o_dists.log_prob_grad(dist_id, v, a, b)
    w = float(_np(g).reshape(-1)[0]) * scale
    m = _bcn(mask, shape)
    outs = []
    for d, n in zip((dv, da, db), need):
        if not n:
            outs.append(None)
            continue
        x = w * np.broadcast_to(d, shape)
        if m is not None:
            x = np.where(m, x, 0.0)
        outs.append(torch.as_tensor(np.ascontiguousarray(x), dtype=value.dtype))
    return outs


def sum_to_nd(x, A, R, B):
    return torch.as_tensor(np.ascontiguousarray(_np(x).reshape(A, R, B).sum(1)), dtype=x.dtype)


def sum_to_nd_pair(x0, x1, A, R, B):
    return sum_to_nd(x0, A, R, B), sum_to_nd(x1, A, R, B)


def _entry_np(e):
    rows, cols = e["rows"], e["cols"]
    v = _bc(e["value"], rows, cols).astype(np.float64)
    a = None if e["p0"] is None else _bc(e["p0"], rows, cols).astype(np.float64)
    b = None if e["p1"] is None else _bc(e["p1"], rows, cols).astype(np.float64)
    return v, a, b, _bc(e["mask"], rows, cols)


def multi_log_prob_sum(entries, coef_all, dtype, device):
    """Numpy restatement of pa_multi_log_prob_sum (SITE_IDENTITY = 100: log_prob(value) = value)."""
    tot = 0.0
    for e in entries:
        v, a, b, m = _entry_np(e)
        if e["dist"] == 101:
            continue
        lp = v if e["dist"] == 100 else o_dists.LOG_PROB[e["dist"]](v, a, b)
        if m is not None:
            lp = np.where(m, lp, 0.0)
        tot += e["coef"] * lp.sum()
    return torch.as_tensor(coef_all * tot, dtype=dtype)


def multi_log_prob_grad(g, entries, coef_all, dtype, device):
    def operand_grad(e, j):
        rows, cols = e["rows"], e["cols"]
        v, a, b, m = _entry_np(e)
        if e["dist"] == 100:
            d = np.ones_like(v) if j == 0 else None
        elif e["dist"] == 101:
            d = np.zeros_like(v) if j == 0 else None
        else:
            d = o_dists.log_prob_grad(e["dist"], v, a, b)[j]
        x = float(g) * coef_all * e["coef"] * np.broadcast_to(d, (rows, cols))
        if m is not None:
            x = np.where(m, x, 0.0)
        src = (e["value"], e["p0"], e["p1"])[j]
        if (src.shape[0] == 1 or src.stride(0) == 0) and rows > 1:
            x = x.sum(0, keepdims=True)
        if (src.shape[1] == 1 or src.stride(1) == 0) and cols > 1:
            x = x.sum(1, keepdims=True)
        return x

    outs = []
    for e in entries:
        res = []
        for j, (need, src) in enumerate(zip(e["need"], (e["value"], e["p0"], e["p1"]))):
            if not need or src is None or (j == 0 and e.get("by_chain")):
                res.append(None)
                continue
            x = operand_grad(e, j)
            if j == 0:
                k = e.get("chain_next", -1)
                while k >= 0:
                    x = x + operand_grad(entries[k], 0)
                    k = entries[k].get("chain_next", -1)
                if e.get("extra_grad") is not None:
                    x = x + float(g) * coef_all * e["extra_coef"] * _np(e["extra_grad"]).reshape(x.shape)
            res.append(torch.as_tensor(np.ascontiguousarray(x), dtype=dtype))
        outs.append(tuple(res))
    return outs


def multi_log_prob_sum_grad(entries, coef_all, dtype, device):
    return (multi_log_prob_sum(entries, coef_all, dtype, device),
            multi_log_prob_grad(1.0, entries, coef_all, dtype, device))


def meanfield_normal_sample(locs, rhos, P, seed, offsets, offset_dev=None):
    zs, scales, louts, epss = [], [], [], []
    base = 0 if offset_dev is None else int(offset_dev.item())
    for loc, rho, off in zip(locs, rhos, offsets):
        n = loc.numel()
        np_dt = np.float32 if loc.dtype == torch.float32 else np.float64
        eps = o_philox.normal(P * n, np_dt, seed, base + int(off)).reshape(P, n)
        r = _np(rho).astype(np.float64)
        sc = np.where(r > 20, r, np.log1p(np.exp(np.minimum(r, 20))))
        z = _np(loc).astype(np.float64)[None, :] + sc[None, :] * eps
        zs.append(torch.as_tensor(z, dtype=loc.dtype))
        scales.append(torch.as_tensor(sc, dtype=loc.dtype))
        louts.append(loc.detach().clone())
        epss.append(torch.as_tensor(eps, dtype=loc.dtype))
    return zs, scales, louts, epss


def meanfield_normal_sample_bwd(rhos, epss, d_zs, d_scales, d_louts, P, sinks=None):
    d_locs, d_rhos = [], []
    for k, (rho, eps, dz, ds, dlo) in enumerate(zip(rhos, epss, d_zs, d_scales, d_louts)):
        r = _np(rho).astype(np.float64)
        sl = np.zeros_like(r) if dz is None else _np(dz).astype(np.float64).sum(0)
        ss = np.zeros_like(r) if dz is None else (_np(dz).astype(np.float64) * _np(eps)).sum(0)
        if ds is not None:
            ss = ss + _np(ds).astype(np.float64)
        if dlo is not None:
            sl = sl + _np(dlo).astype(np.float64)
        sig = np.where(r > 20, 1.0, 1.0 / (1.0 + np.exp(-r)))
        if sinks is not None and sinks[k] is not None:
            sinks[k][0].add_(torch.as_tensor(sl, dtype=rho.dtype))
            sinks[k][1].add_(torch.as_tensor(ss * sig, dtype=rho.dtype))
            d_locs.append(None); d_rhos.append(None)
            continue
        d_locs.append(torch.as_tensor(sl, dtype=rho.dtype))
        d_rhos.append(torch.as_tensor(ss * sig, dtype=rho.dtype))
    return d_locs, d_rhos


def glm_bernoulli_fwd_bwd(X, y, w, b, mask, scale):
    ll, gw, gb = o_glm.glm_bernoulli_fwd_bwd(_np(X), _np(y), _np(w), _np(b), _np(mask), scale)
    return (torch.as_tensor(ll, dtype=X.dtype), torch.as_tensor(gw, dtype=X.dtype),
            torch.as_tensor(gb, dtype=X.dtype))


def glm_chain(g, gw, gb, need_w=True, need_b=True):
    gg = g.reshape(-1)
    dw = gg.reshape((-1,) + (1,) * (gw.dim() - 1)) * gw if need_w else None
    db = gg * gb if (need_b and gb is not None) else None
    return dw, db


class GroupSegments:
    def __init__(self, group_offsets, device, target_segments=4096):
        off = np.asarray(group_offsets, dtype=np.int64)
        self.G, self.N, self.group_offsets = off.size - 1, int(off[-1]), off
        self.rows = self.ids = None
        self.ids_version = 0


def grouped_rows_of(g, G):
    """kernels.grouped_rows_of answered by oracle/glm.py::group_rows (no cache: host-logic tests)."""
    off, rows = o_glm.group_rows(_np(g), G)
    segs = GroupSegments(off, g.device)
    segs.rows, segs.ids, segs.ids_version = torch.as_tensor(rows), g, g._version
    return segs


def glm_grouped_rows_servable(X, y, mask, segs):
    return segs.rows is None or mask is None


def glm_bernoulli_grouped_fwd_bwd(X, y, w, b, mask, scale, segs):
    g_of = np.repeat(np.arange(segs.G), np.diff(segs.group_offsets)) if segs.ids is None else _np(segs.ids)
    ll, gw, gb = o_glm.glm_bernoulli_grouped_fwd_bwd(_np(X), _np(y), _np(w), g_of, _np(b), _np(mask),
                                                     scale)
    return (torch.as_tensor(ll, dtype=X.dtype), torch.as_tensor(gw, dtype=X.dtype),
            torch.as_tensor(gb, dtype=X.dtype))


def leapfrog_kick_drift(z, r, grad, inv_mass, step):
    st = _np(step).reshape(-1, 1) if step.dim() == 1 else _np(step)
    rn = _np(r) + 0.5 * st * (-_np(grad))
    r.copy_(torch.as_tensor(rn))
    z.copy_(torch.as_tensor(_np(z) + st * (_np(inv_mass) * rn)))


def leapfrog_kick(r, grad, step):
    st = _np(step).reshape(-1, 1) if step.dim() == 1 else _np(step)
    r.copy_(torch.as_tensor(_np(r) + 0.5 * st * (-_np(grad))))


def nuts_gaussian_transition(z, pe, grad, Lambda, inv_mass, step, max_tree_depth, use_multinomial,
                             seed, t, chain_offset=0):
    C, D = z.shape
    np_dt = np.float32 if z.dtype == torch.float32 else np.float64
    pg = o_int.gaussian_potential(_np(Lambda).astype(np.float64))
    ap = np.zeros(C)
    ints = np.zeros((4, C), np.int32)
    zn, pen, gn = _np(z).copy(), _np(pe).copy(), _np(grad).copy()
    for c in range(C):
        out = o_nuts.nuts_transition(zn[c].astype(np.float64), float(pen[c]), gn[c].astype(np.float64),
                                     pg, _np(inv_mass)[c].astype(np.float64), float(_np(step)[c]),
                                     o_nuts.KeyedDraws(seed, chain_offset + c, t, np_dt), max_tree_depth,
                                     bool(use_multinomial))
        zn[c], pen[c], gn[c] = out["z"], out["pe"], out["grad"]
        ap[c] = out["accept_prob"]
        ints[:, c] = (out["n_leapfrog"], out["depth"], out["diverging"], out["accepted"])
    z.copy_(torch.as_tensor(zn)); pe.copy_(torch.as_tensor(pen)); grad.copy_(torch.as_tensor(gn))
    ti = torch.as_tensor(ints)
    return {"accept_prob": torch.as_tensor(ap, dtype=z.dtype), "n_leapfrog": ti[0], "depth": ti[1],
            "diverging": ti[2], "accepted": ti[3]}


def nuts_gaussian_run(z, pe, grad, Lambda, inv_mass, step, max_tree_depth, use_multinomial, seed,
                      t0, num_transitions, chain_offset=0, da_state=None, target_accept=0.8,
                      welford=None, welford_n0=0, samples=None, mean_accept=None, mean_n0=0,
                      counters=None, count_accepts=False, div_flags=None):
    """Stand-in of the persistent kernel: the single-transition stand-in in a loop plus the
    per-transition adaptation recurrences (DualAveraging.step / WelfordCovariance.update)."""
    out = None
    for k in range(int(num_transitions)):
        out = nuts_gaussian_transition(z, pe, grad, Lambda, inv_mass, step, max_tree_depth,
                                       use_multinomial, seed, t0 + k, chain_offset)
        ap = torch.nan_to_num(out["accept_prob"], nan=0.0)
        if counters is not None:
            counters[0] += out["n_leapfrog"].to(torch.int64)
            counters[1] += out["depth"].to(torch.int64)
            if count_accepts:
                counters[2] += out["accepted"].to(torch.int64)
        if mean_accept is not None:
            mean_accept += (ap - mean_accept) / (mean_n0 + k + 1)
        if count_accepts and div_flags is not None:
            div_flags[k] = out["diverging"].to(torch.int8)
        if da_state is not None:
            g = target_accept - ap
            da_state[:, 2] += 1
            t = da_state[:, 2]
            da_state[:, 1] = (1 - 1 / (t + 10.0)) * da_state[:, 1] + g / (t + 10.0)
            da_state[:, 4] = da_state[:, 3] - t.sqrt() / 0.05 * da_state[:, 1]
            w = t ** (-0.75)
            da_state[:, 0] = (1 - w) * da_state[:, 0] + w * da_state[:, 4]
            step.copy_(da_state[:, 4].exp())
        if welford is not None:
            n = welford_n0 + k + 1
            pre = z - welford[:, 0]
            welford[:, 0] += pre / n
            welford[:, 1] += pre * (z - welford[:, 0])
        if samples is not None:
            samples[k] = z
    return out


class NutsTree:
    """Stand-in for kernels.NutsTree backed by oracle/nuts_tree.py (numpy, host)."""

    def __init__(self, z, pe, grad, inv_mass, step, max_tree_depth=10, use_multinomial=True,
                 seed=0, chain_offset=0):
        self.z, self.pe, self.grad, self.inv_mass, self.step = z, pe, grad, inv_mass, step
        self.C, self.D = z.shape
        self.im_stride = self.D if inv_mass.dim() == 2 else 0
        self._args = (max_tree_depth, bool(use_multinomial), seed, chain_offset)
        self.zq = torch.zeros_like(z)
        self.rq = torch.zeros_like(z)
        self._o = None

    def begin(self, t):
        np_dt = np.float32 if self.z.dtype == torch.float32 else np.float64
        self._zn, self._pn, self._gn = _np(self.z).copy(), _np(self.pe).copy(), _np(self.grad).copy()
        self._o = o_tree.NutsTreeOracle(self._zn, self._pn, self._gn, _np(self.inv_mass),
                                        _np(self.step), *self._args, dtype=np_dt)
        self._o.begin(t)
        self._sync()

    def _sync(self):
        self.zq.copy_(torch.as_tensor(self._o.zq))
        self.rq.copy_(torch.as_tensor(self._o.rq))
        self.z.copy_(torch.as_tensor(self._zn)); self.pe.copy_(torch.as_tensor(self._pn))
        self.grad.copy_(torch.as_tensor(self._gn))

    def advance(self, peq, gq):
        self._o.advance(_np(peq), _np(gq))
        self._sync()

    advance_replayable = advance

    def n_active(self):
        return self._o.n_active()

    # asynchronous chains (kernels.NutsTree.run_*): the span oracle of oracle/nuts_tree.py
    RUN_ADAPT_STEP, RUN_WELFORD, RUN_COUNT_ACCEPTS = 1, 2, 4
    gate = None

    def set_span(self, t0, K, mean_n0=0, welford_n0=0, flags=0, samples=None, div_flags=None, row0=0):
        self._span = dict(t0=t0, K=K, mean_n0=mean_n0, welford_n0=welford_n0, flags=flags, row0=row0)
        self._span_out = (samples, div_flags)

    def run_begin(self):
        np_dt = np.float32 if self.z.dtype == torch.float32 else np.float64
        self._zn, self._pn, self._gn = _np(self.z).copy(), _np(self.pe).copy(), _np(self.grad).copy()
        self._stepn = _np(self.step).copy()
        self._o = o_tree.NutsTreeOracle(self._zn, self._pn, self._gn, _np(self.inv_mass),
                                        self._stepn, *self._args, dtype=np_dt)
        samples, div = self._span_out
        self._sn = None if samples is None else _np(samples).copy()
        self._dn = None if div is None else _np(div).copy()
        self._o.run_begin(samples=self._sn, div_flags=self._dn, **self._span)
        self._sync()

    def compact(self, n_slots, program=None):
        """kernels.NutsTree.compact: the chains still building a tree (ascending, -1 pads), their cursors."""
        assert program is None          # (the direct potential is a device path: tests/test_mcmc_gpu.py)
        active = [c for c in range(self.C) if self._o.chains[c].active]
        assert len(active) <= n_slots
        s2c = torch.full((n_slots,), -1, dtype=torch.int32)
        s2c[:len(active)] = torch.tensor(active, dtype=torch.int32)
        zqs = torch.zeros((n_slots, self.D), dtype=self.z.dtype)
        zqs[:len(active)] = self.zq[active]
        return s2c, zqs

    def run_advance(self, peq, gq, da_state, target_accept, welford, mean_accept, counters, slots=None):
        arrs = [_np(x).copy() for x in (da_state, welford, mean_accept, counters)]
        if slots is not None:          # a compacted round: (peq, gq) per slot -> per chain
            s2c = slots[0].tolist()
            pf, gf = np.zeros(self.C, _np(peq).dtype), np.zeros((self.C, self.D), _np(gq).dtype)
            for s_, c in enumerate(s2c):
                if c >= 0:
                    pf[c], gf[c] = _np(peq)[s_], _np(gq)[s_]
            self._o.run_advance(pf, gf, arrs[0], target_accept, arrs[1], arrs[2], arrs[3])
        else:
            self._o.run_advance(_np(peq), _np(gq), arrs[0], target_accept, arrs[1], arrs[2], arrs[3])
        for t, a in zip((da_state, welford, mean_accept, counters), arrs):
            t.copy_(torch.as_tensor(a))
        self.step.copy_(torch.as_tensor(self._stepn))
        samples, div = self._span_out
        if samples is not None:
            samples.copy_(torch.as_tensor(self._sn))
        if div is not None:
            div.copy_(torch.as_tensor(self._dn))
        self._sync()
        if slots is not None:          # a chain writes its next cursor to its slot row too
            for s_, c in enumerate(slots[0].tolist()):
                if c >= 0:
                    slots[1][s_] = self.zq[c]

    def chains_done(self):
        return self._o.n_done

    def span_done(self):
        return self._o.span_done()

    def stats(self):
        ti = torch.as_tensor(self._o.ints)
        return {"accept_prob": torch.as_tensor(self._o.accept_prob, dtype=self.z.dtype),
                "n_leapfrog": ti[0], "depth": ti[1], "diverging": ti[2], "accepted": ti[3]}


def lda_factor_fwd_bwd(words, log_theta, log_phi):
    out, gt, gp = o_lda.lda_factor(_np(words), _np(log_theta), _np(log_phi))
    dt = log_theta.dtype
    return torch.as_tensor(out, dtype=dt), torch.as_tensor(gt, dtype=dt), torch.as_tensor(gp, dtype=dt)


def mixture_fwd_bwd(dist_id, x, a, p0, s0, p1, s1, p0_bs=0, p1_bs=0):
    from oracle import mixture as o_mix
    an = _np(a)
    K = an.shape[-1]

    def of(p, s_, bs, b):
        if p is None:
            return None
        flat = _np(p).reshape(-1)
        return np.array([flat[b * bs + k * s_] for k in range(K)])

    rows = []
    for b in range(an.shape[0] if an.ndim == 2 else 1):
        S, da, d0, d1 = o_mix.mixture_fwd_bwd(int(dist_id), _np(x), an[b] if an.ndim == 2 else an,
                                              of(p0, s0, p0_bs, b), of(p1, s1, p1_bs, b))
        rows.append(np.concatenate([[S], da, d0, d1]))
    return torch.as_tensor(np.stack(rows) if an.ndim == 2 else rows[0], dtype=torch.float64)


def mixture_diag_normal_fwd_bwd(x, a, loc, scale):
    from oracle import mixture as o_mix
    an = _np(a)
    B, K = an.shape
    D = x.shape[1]
    ln = np.broadcast_to(_np(loc), (B, K, D))
    sn = np.broadcast_to(_np(scale), (B, K, D))
    rows = [o_mix.mixture_diag_normal_fwd_bwd(_np(x), an[b], ln[b], sn[b]) for b in range(B)]
    f = lambda i: torch.as_tensor(np.stack([np.asarray(r[i]) for r in rows]), dtype=torch.float64)  # noqa: E731
    return f(0), f(1), f(2), f(3)


def logsumexp_terms(terms, frame, rdim):
    from oracle import logsumexp as o_lse
    out, _ = o_lse.logsumexp_terms([_np(t) for t in terms], tuple(frame), rdim)
    return torch.as_tensor(out, dtype=terms[0].dtype)


def logsumexp_terms_grad(terms, frame, rdim, out, g_out):
    from oracle import logsumexp as o_lse
    G = o_lse.logsumexp_terms_grad([_np(t) for t in terms], tuple(frame), rdim, _np(g_out))
    return torch.as_tensor(G, dtype=terms[0].dtype)


def adam_step(param, grad, exp_avg, exp_avg_sq, step_dev, lr, betas=(0.9, 0.999), eps=1e-8,
              weight_decay=0.0, clip_norm=0.0, lrd=1.0, clipped=False, zero_grad=True,
              publish=None):
    assert publish is None          # only a captured (GPU) step folds the hand-over in
    step = int(step_dev[0].item()) + 1
    p, m, v = o_adam.adam_step(_np(param), _np(grad), _np(exp_avg), _np(exp_avg_sq), step, lr, betas,
                               eps, weight_decay, clip_norm, lrd, clipped)
    param.data.copy_(torch.as_tensor(p)); exp_avg.copy_(torch.as_tensor(m))
    exp_avg_sq.copy_(torch.as_tensor(v))
    step_dev[0] += 1
    if zero_grad:
        grad.zero_()


def _softplus_np(x):
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))


def mvn_tril_sample(loc, rho, A, P, seed=0, offset=0, offset_dev=None, eps=None):
    """torch/distributions/multivariate_normal.py rsample + log_prob for scale_tril =
    softplus(rho)[:, None] * (tril(A, -1) + I), written out in numpy."""
    n = loc.numel()
    if eps is None:
        eps = philox_normal((P, n), loc.dtype, loc.device, seed, offset, offset_dev)
    e, lo, S, An = _np(eps), _np(loc), _softplus_np(_np(rho)), _np(A)
    L = np.tril(An, -1) + np.eye(n)
    T = S[:, None] * L
    z = lo + e @ T.T
    # log_prob through the triangular solve, as the reference computes it
    y = np.linalg.solve(T, (z - lo).T).T
    logq = -0.5 * (y * y).sum(-1) - np.log(np.diag(T)).sum() - 0.5 * n * np.log(2 * np.pi)
    return (torch.as_tensor(z, dtype=loc.dtype), torch.as_tensor(logq, dtype=loc.dtype), eps)


def mvn_tril_sample_bwd(loc, rho, eps, z, d_z, d_logq, sinks=None):
    P, n = eps.shape
    e, S, r = _np(eps), _softplus_np(_np(rho)), _np(rho)
    dz = np.zeros((P, n)) if d_z is None else _np(d_z)
    dq = np.zeros(P) if d_logq is None else _np(d_logq)
    u = (_np(z) - _np(loc)) / S
    d_loc = dz.sum(0)
    dS = (dz * u).sum(0) - dq.sum() / S
    d_rho = dS * np.where(r > 20, 1.0, 1.0 / (1.0 + np.exp(-r)))
    d_A = np.tril(S[:, None] * (dz.T @ e), -1)
    outs = [torch.as_tensor(np.ascontiguousarray(v), dtype=loc.dtype) for v in (d_loc, d_rho, d_A)]
    if sinks is not None:
        for sk, v in zip(sinks, outs):
            sk.add_(v.reshape(sk.shape))
        return None, None, None
    return tuple(outs)


def logchain_fwd_bwd(unary, pairwise):
    """Forward algorithm + forward-backward posteriors in numpy (float64)."""
    from scipy.special import logsumexp
    U = _np(unary).astype(np.float64)
    B, T, K = U.shape
    Pn = _np(pairwise).astype(np.float64) if T > 1 else np.zeros((B, 0, K, K))
    Pn = np.broadcast_to(Pn, (B, max(T - 1, 0), K, K))
    alpha = np.zeros((B, T, K))
    alpha[:, 0] = U[:, 0]
    for t in range(1, T):
        alpha[:, t] = U[:, t] + logsumexp(alpha[:, t - 1, :, None] + Pn[:, t - 1], axis=1)
    lz = logsumexp(alpha[:, -1], axis=1)
    beta = np.zeros((B, T, K))
    xi = np.zeros((B, max(T - 1, 0), K, K))
    for t in range(T - 2, -1, -1):
        w = U[:, t + 1] + beta[:, t + 1]
        beta[:, t] = logsumexp(Pn[:, t] + w[:, None, :], axis=2)
        xi[:, t] = np.exp(alpha[:, t, :, None] + Pn[:, t] + w[:, None, :] - lz[:, None, None])
    gamma = np.exp(alpha + beta - lz[:, None, None])
    mk = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=unary.dtype)   # noqa: E731
    return mk(lz), mk(gamma), mk(xi)


def chain_matvec(M, x, transpose=False):
    Mn, xn = _np(M), _np(x)
    if Mn.ndim == 2:
        Mn = np.broadcast_to(Mn, (xn.shape[0],) + Mn.shape)
    y = np.einsum("cij,ci->cj" if transpose else "cij,cj->ci", Mn, xn)
    return torch.as_tensor(np.ascontiguousarray(y), dtype=x.dtype)


def meanfield_score(z, loc, scale, P, coef):
    n = loc.numel()
    zz = _np(z).astype(np.float64).reshape(P, n)
    l, s = _np(loc).astype(np.float64).reshape(-1), _np(scale).astype(np.float64).reshape(-1)
    lp = -np.log(s)[None, :] - 0.5 * ((zz - l[None, :]) / s[None, :]) ** 2 - 0.5 * np.log(2 * np.pi)
    partial = torch.as_tensor(np.array([coef * lp.sum()]), dtype=z.dtype)
    return partial, torch.as_tensor(-coef * P / s, dtype=z.dtype)


def exp_site_fwd(u, cols, lower=0.0, want_ld=True):
    a = _np(u).astype(np.float64)
    value = lower + np.exp(a)
    ld = -a.reshape(-1, cols).sum(1)
    return torch.as_tensor(value, dtype=u.dtype), (torch.as_tensor(ld, dtype=u.dtype) if want_ld else None)


def exp_site_bwd(value, g_value, g_ld, cols, lower=0.0):
    v = _np(value).astype(np.float64)
    g = np.zeros_like(v)
    if g_value is not None:
        g = g + _np(g_value).astype(np.float64) * (v - lower)
    if g_ld is not None:
        g = (g.reshape(-1, cols) - _np(g_ld).astype(np.float64).reshape(-1, 1)).reshape(v.shape)
    return torch.as_tensor(g, dtype=value.dtype)


# ---- the Linear layers of an amortised guide over word histograms (csrc/bow.hip, csrc/tall.hip) -------
# The "images" of the stand-in are the word matrix itself; everything else restates the entry points'
# contracts in float64 numpy (oracle/lda.py holds the layout-exact restatements the GPU tests compare with).
def bow_images_of(words, V):
    return (words, words)


def _counts(words, V):
    w = _np(words)
    c = np.zeros((w.shape[1], V))
    for b in range(w.shape[1]):
        c[b] = np.bincount(w[:, b], minlength=V)
    return c


def _sig(h):
    return 1.0 / (1.0 + np.exp(-h))


def bow_linear_fwd(image_a, W, bias, B, sigmoid=False):
    Wn = _np(W).astype(np.float64)
    h = _counts(image_a, Wn.shape[1]) @ Wn.T + (0.0 if bias is None else _np(bias).astype(np.float64))
    return torch.as_tensor(_sig(h) if sigmoid else h, dtype=W.dtype, device=W.device)


def bow_linear_bwd(image_b, d_out, V, y_mul=None, want_bias=False):
    d = _np(d_out).astype(np.float64)
    if y_mul is not None:
        y = _np(y_mul).astype(np.float64)
        d = d * (1.0 - y) * y
    dW = torch.as_tensor(d.T @ _counts(image_b, V), dtype=d_out.dtype, device=d_out.device)
    if not want_bias:
        return dW
    B, H = d.shape
    nkt = (B + 31) // 32 * 2
    dp = np.zeros((nkt * 16, 128))
    dp[:B, :H] = d
    part = dp.reshape(nkt, 16, 4, 32).sum(1).transpose(1, 0, 2)          # [4, nkt, 32]
    return dW, torch.as_tensor(np.ascontiguousarray(part), dtype=d_out.dtype, device=d_out.device)


def tall_linear(g, W, w_row_stride, w_col_stride, C, bias=None, y_mul=None, sigmoid=False):
    gn = _np(g).astype(np.float64)
    if y_mul is not None:
        y = _np(y_mul).astype(np.float64)
        gn = gn * (1.0 - y) * y
    R = gn.shape[1]
    flat = _np(W).astype(np.float64).reshape(-1)
    Wm = flat[(np.arange(R)[:, None] * w_row_stride + np.arange(C)[None, :] * w_col_stride)]
    h = gn @ Wm + (0.0 if bias is None else _np(bias).astype(np.float64))
    return torch.as_tensor(_sig(h) if sigmoid else h, dtype=g.dtype, device=g.device)


def tall_wgrad(g, x, want_bias=True, y_mul=None):
    gn = _np(g).astype(np.float64)
    if y_mul is not None:
        y = _np(y_mul).astype(np.float64)
        gn = gn * (1.0 - y) * y
    dW = torch.as_tensor(gn.T @ _np(x).astype(np.float64), dtype=g.dtype, device=g.device)
    db = torch.as_tensor(gn.sum(0), dtype=g.dtype, device=g.device) if want_bias else None
    return dW, db


FUNCTIONS = ["bow_images_of", "bow_linear_fwd", "bow_linear_bwd", "tall_linear", "tall_wgrad", "exp_site_fwd", "exp_site_bwd", "meanfield_score", "sum_to_nd_pair", "philox_normal", "philox_uniform", "dist_log_prob", "dist_log_prob_sum",
             "dist_log_prob_grad", "glm_bernoulli_fwd_bwd", "leapfrog_kick_drift", "leapfrog_kick",
             "nuts_gaussian_transition", "nuts_gaussian_run", "lda_factor_fwd_bwd", "adam_step", "NutsTree", "GroupSegments",
             "glm_bernoulli_grouped_fwd_bwd", "grouped_rows_of", "glm_grouped_rows_servable", "multi_log_prob_sum", "multi_log_prob_grad", "multi_log_prob_sum_grad",
             "meanfield_normal_sample", "meanfield_normal_sample_bwd", "glm_chain", "chain_matvec", "mvn_tril_sample",
             "mvn_tril_sample_bwd", "logchain_fwd_bwd", "dist_log_prob_sum_nd", "dist_log_prob_grad_nd", "sum_to_nd",
             "logsumexp_terms", "logsumexp_terms_grad", "gamma_rsample", "mixture_fwd_bwd",
             "mixture_diag_normal_fwd_bwd"]


def install(monkeypatch):
    import pyro_amd.kernels as k
    g = globals()
    for name in FUNCTIONS:
        monkeypatch.setattr(k, name, g[name])
    # the product refuses CPU tensors; lift that check for host-logic tests only
    monkeypatch.setattr(k, "_require_gpu", lambda *a: None)
    monkeypatch.setattr(k, "on_device", lambda t: True)     # the oracle stands in for the device
    from pyro_amd.ops import lazy, torch_library
    monkeypatch.setattr(lazy, "_on_device", lambda t: True)  # ... for the lazy recognition too
    # the TORCH_LIBRARY ops launch HIP kernels from C++: host tensors take the autograd.Function route,
    # whose kernels.* calls are the stand-ins above
    monkeypatch.setattr(torch_library, "available", lambda: False)
