"""Execution trace of a probabilistic program and the per-site log-probability bookkeeping
(reference: pyro/poutine/trace_struct.py -- add_node / log_prob_sum :203-246 /
compute_log_prob :248-288 / compute_score_parts :290-328).

Difference from the reference, by design: when only the *sum* of a site's scaled+masked
log-probability is needed (Trace_ELBO), ``compute_log_prob_sums`` asks the site's distribution
for ``fused_log_prob_sum`` -- one HIP kernel instead of log_prob, scale, where, sum (and one
backward kernel instead of four).  The un-reduced ``site["log_prob"]`` is still available
through ``compute_log_prob`` for estimators that need it.
"""
from collections import OrderedDict

import torch

from ..distributions.util import is_identically_zero, scale_and_mask
from ..util import warn_if_inf, warn_if_nan
from . import settings


class Trace:
    def __init__(self, graph_type="flat"):
        assert graph_type in ("flat", "dense")
        self.graph_type = graph_type
        self.nodes = OrderedDict()
        self._succ = OrderedDict()         # the DAG over site names (dense graphs, trace_struct.py:148-201)
        self._pred = OrderedDict()

    def __contains__(self, name):
        return name in self.nodes

    def __iter__(self):
        return iter(self.nodes.keys())

    def __len__(self):
        return len(self.nodes)

    def add_node(self, site_name, **kwargs):
        if site_name in self.nodes:
            site = self.nodes[site_name]
            if site["type"] != kwargs["type"]:
                raise RuntimeError("{} is already in the trace as a {}".format(site_name,
                                                                               site["type"]))
            elif kwargs["type"] != "param":
                raise RuntimeError("Multiple {} sites named '{}'".format(kwargs["type"], site_name))
        self.nodes[site_name] = kwargs
        self._succ.setdefault(site_name, set())
        self._pred.setdefault(site_name, set())

    def add_edge(self, site1, site2):
        for site in (site1, site2):
            if site not in self.nodes:
                self.add_node(site)
        self._succ[site1].add(site2)
        self._pred[site2].add(site1)

    def remove_node(self, site_name):
        del self.nodes[site_name]
        for p in self._pred.pop(site_name, ()):
            self._succ[p].discard(site_name)
        for q in self._succ.pop(site_name, ()):
            self._pred[q].discard(site_name)

    def predecessors(self, site_name):
        return self._pred[site_name]

    def successors(self, site_name):
        return self._succ[site_name]

    @property
    def edges(self):
        for site, followers in self._succ.items():
            for follower in followers:
                yield site, follower

    def _dfs(self, site, visited):
        """Post-order walk of what is reachable from ``site`` and not yet visited."""
        if site in visited:
            return
        for follower in self._succ[site]:
            yield from self._dfs(follower, visited)
        visited.add(site)
        yield site

    def topological_sort(self, reverse=False):
        visited, post_order = set(), []
        for site in self._succ:
            post_order.extend(self._dfs(site, visited))
        return post_order if reverse else post_order[::-1]

    def copy(self):
        new = Trace(self.graph_type)
        for name, site in self.nodes.items():
            new.nodes[name] = dict(site)
        for name in self._succ:
            new._succ[name] = set(self._succ[name])
            new._pred[name] = set(self._pred[name])
        return new

    def detach_(self):
        for site in self.nodes.values():
            if site["type"] == "sample":
                site["value"] = site["value"].detach()

    # ---- node views -----------------------------------------------------------------------
    @property
    def stochastic_nodes(self):
        return [n for n, s in self.nodes.items() if s["type"] == "sample" and not s["is_observed"]]

    @property
    def observation_nodes(self):
        return [n for n, s in self.nodes.items() if s["type"] == "sample" and s["is_observed"]]

    @property
    def param_nodes(self):
        return [n for n, s in self.nodes.items() if s["type"] == "param"]

    @property
    def reparameterized_nodes(self):
        return [n for n, s in self.nodes.items() if s["type"] == "sample"
                and not s["is_observed"] and getattr(s["fn"], "has_rsample", False)]

    @property
    def nonreparam_stochastic_nodes(self):
        return [n for n, s in self.nodes.items() if s["type"] == "sample"
                and not s["is_observed"] and not getattr(s["fn"], "has_rsample", False)]

    def iter_stochastic_nodes(self):
        for name, node in self.nodes.items():
            if node["type"] == "sample" and not node["is_observed"]:
                yield name, node

    # ---- log-probabilities ----------------------------------------------------------------
    def _site_error(self, name, site, exc, what="log_prob"):
        shapes = self.format_shapes(last_site=name)
        return ValueError("Error while computing {} at site '{}':\n{}\n{}".format(
            what, name, exc, shapes))

    def compute_log_prob(self, site_filter=lambda name, site: True):
        """Un-reduced path: site["unscaled_log_prob"], ["log_prob"], ["log_prob_sum"]."""
        for name, site in self.nodes.items():
            if site["type"] != "sample" or not site_filter(name, site):
                continue
            if "log_prob" in site:
                continue
            try:
                log_p = site["fn"].log_prob(site["value"], *site["args"], **site["kwargs"])
            except ValueError as e:
                raise self._site_error(name, site, e) from e
            site["unscaled_log_prob"] = log_p
            log_p = scale_and_mask(log_p, site["scale"], site["mask"])
            site["log_prob"] = log_p
            if "log_prob_sum" not in site:
                site["log_prob_sum"] = log_p.sum()
            if settings.validation_enabled():
                warn_if_nan(site["log_prob_sum"], "log_prob_sum at site '{}'".format(name))
                warn_if_inf(site["log_prob_sum"], "log_prob_sum at site '{}'".format(name),
                            allow_neginf=True)

    def compute_log_prob_sums(self, site_filter=lambda name, site: True):
        """Fused path: only site["log_prob_sum"] (0-dim, differentiable)."""
        for name, site in self.nodes.items():
            if site["type"] != "sample" or not site_filter(name, site):
                continue
            if "log_prob_sum" in site:
                continue
            site["log_prob_sum"] = self._site_sum(name, site)
            if settings.validation_enabled():
                warn_if_nan(site["log_prob_sum"], "log_prob_sum at site '{}'".format(name))
                warn_if_inf(site["log_prob_sum"], "log_prob_sum at site '{}'".format(name),
                            allow_neginf=True)

    def collect_log_prob_sums(self, batch, sign, site_filter=lambda name, site: True):
        """Batched fused path: instead of one reduction launch per site, every small element-wise
        site is described to ``batch`` (distributions.fused.SiteBatch) and the whole signed sum is
        produced by one launch in ``batch.total()``.  Sites that cannot be described that way are
        reduced on their own and handed over as already-computed terms.  Returns the list of
        (sign, tensor) terms the batch could not take (other dtype / device)."""
        leftovers = []
        for name, site in self.nodes.items():
            if site["type"] != "sample" or not site_filter(name, site):
                continue
            term = site.get("log_prob_sum")
            if term is None:
                fn, value, scale, mask = site["fn"], site["value"], site["scale"], site["mask"]
                if mask is False:
                    continue
                if mask is True:
                    mask = None
                plain = not site["args"] and not site["kwargs"]
                entry_fn = getattr(fn, "fused_site_entry", None) if plain else None
                if entry_fn is not None and isinstance(value, torch.Tensor):
                    try:
                        entry = entry_fn(value, scale, mask)
                    except ValueError as e:
                        raise self._site_error(name, site, e, "log_prob_sum") from e
                    if entry is not None and batch.add_site(*entry, sign):
                        continue
                lin_fn = getattr(fn, "fused_linear_term", None) if plain else None
                if lin_fn is not None and torch.is_grad_enabled():
                    try:
                        lin = lin_fn(value, scale, mask)   # (ll, [(tensor, known gradient), ...])
                    except ValueError as e:
                        raise self._site_error(name, site, e, "log_prob_sum") from e
                    if lin is not None and batch.add_linear_term(lin[0], lin[1], sign):
                        continue
                score_fn = getattr(fn, "fused_score_term", None) if plain else None
                if score_fn is not None and torch.is_grad_enabled():
                    term = score_fn(value, scale, mask)          # a guide site at its own draw
                batch_fn = getattr(fn, "fused_log_prob_batch", None) if plain and term is None else None
                if batch_fn is not None:
                    try:
                        term = batch_fn(value, scale, mask)      # e.g. per-particle sums ll[P]
                    except ValueError as e:
                        raise self._site_error(name, site, e, "log_prob_sum") from e
                if term is None:
                    term = self._site_sum(name, site)
            if not batch.add_term(term, sign):
                leftovers.append((sign, term.sum() if term.dim() else term))
        return leftovers

    def _site_sum(self, name, site):
        fn, value, scale, mask = site["fn"], site["value"], site["scale"], site["mask"]
        if mask is False:
            return torch.zeros((), dtype=value.dtype if value.is_floating_point()
                               else torch.get_default_dtype(), device=value.device)
        if mask is True:
            mask = None
        fused = getattr(fn, "fused_log_prob_sum", None)
        if fused is not None and not site["args"] and not site["kwargs"]:
            try:
                out = fused(value, scale, mask)
            except ValueError as e:
                raise self._site_error(name, site, e, "log_prob_sum") from e
            if out is not None:
                return out
        try:
            log_p = fn.log_prob(value, *site["args"], **site["kwargs"])
        except ValueError as e:
            raise self._site_error(name, site, e, "log_prob_sum") from e
        return scale_and_mask(log_p, scale, mask).sum()

    def log_prob_sum(self, site_filter=lambda name, site: True):
        self.compute_log_prob_sums(site_filter)
        result = 0.0
        for name, site in self.nodes.items():
            if site["type"] == "sample" and site_filter(name, site):
                result = result + site["log_prob_sum"]
        return result

    def compute_score_parts(self):
        """Guide side: site["score_parts"] = (log_prob, score_function, entropy_term), scaled and
        masked, plus ["log_prob"], ["log_prob_sum"]."""
        for name, site in self.nodes.items():
            if site["type"] != "sample" or "score_parts" in site:
                continue
            try:
                value = site["fn"].score_parts(site["value"], *site["args"], **site["kwargs"])
            except ValueError as e:
                raise self._site_error(name, site, e, "score_parts") from e
            site["unscaled_log_prob"] = value.log_prob
            value = value.scale_and_mask(site["scale"], site["mask"])
            site["score_parts"] = value
            site["log_prob"] = value.log_prob
            site["log_prob_sum"] = value.log_prob.sum()
            if settings.validation_enabled():
                warn_if_nan(site["log_prob_sum"], "log_prob_sum at site '{}'".format(name))

    def format_shapes(self, title="Trace Shapes:", last_site=None):
        rows = [[title]]
        for name, site in self.nodes.items():
            if site["type"] == "sample":
                fn = site["fn"]
                batch = getattr(fn, "batch_shape", None)
                event = getattr(fn, "event_shape", None)
                rows.append(["{} dist".format(name), str(tuple(batch) if batch is not None else "?"),
                             "|", str(tuple(event) if event is not None else "?")])
                v = site["value"]
                if v is not None and hasattr(v, "shape"):
                    rows.append(["{} value".format(" " * len(name)), str(tuple(v.shape))])
            if name == last_site:
                break
        return "\n".join(" ".join(r) for r in rows)


__all__ = ["Trace", "is_identically_zero"]
