"""Plated sum-product in log space: marginalise enumerated discrete variables out of a set of
log-factors that live in (tree-structured) plate contexts
(reference: pyro/ops/contract.py:44-202 _partition_terms / _contract_component /
contract_tensor_tree, with LogRing of pyro/ops/rings.py:178-216, the log-space pairwise einsum of
pyro/ops/einsum/torch_log.py:14-55, and the packed, symbol-named tensors of
pyro/poutine/trace_struct.py:398-473 + pyro/ops/packed.py).

Representation: a factor is PACKED -- its leading dims are exactly the enumerated variables it
depends on, named by the variables' unique ids (``Term.ids``), followed by a right-aligned block of
``nplates`` plate dims (size 1 where the factor does not vary).  Naming by id rather than by tensor
position is what lets pyro.markov recycle enumeration dims: x_{t-2} and x_t may have occupied the
same tensor dim in the program, here they are different names.  No einsum symbol strings and no
contraction-path search are needed: variables are eliminated one at a time in a greedy min-size
order (a chain of T variables costs T small logsumexps, never a K^T tensor); a sum-contraction is
an aligned broadcast add + logsumexp over the eliminated names, a plate product is a sum over the
plate dims.  The LDA-shaped leaf (a factor constant along the inner plate + an observed
Categorical whose logits are gathered by the enumerated value) is contracted by ONE fused HIP
kernel (pa_lda_factor_fwd_bwd) that never materialises the [T, words, docs] tensor.
"""
from collections import OrderedDict

import torch

from .. import kernels
from .torch_library import dispatcher_op as _dispatcher_op


FUSED_CHAIN = True            # HMM-shaped components go through pa_logchain_fwd_bwd


class Term:
    """A packed log-factor: ``tensor`` [k_1..k_m, *plate block] (or a lazy gather), ``ids`` = the m
    enumerated variables of the leading dims, ``ordinal`` = the plate context (frozenset of
    vectorised CondIndepStackFrames) it lives in."""

    __slots__ = ("tensor", "ids", "ordinal", "lazy")

    def __init__(self, tensor, ids, ordinal, lazy=None):
        self.tensor, self.ids, self.ordinal, self.lazy = tensor, tuple(ids), ordinal, lazy

    @property
    def dims(self):
        return frozenset(self.ids)

    def dense(self):
        if self.tensor is None:
            self.tensor = self.lazy.materialize()
        return self.tensor


def pack(tensor, dim_to_id, nplates, ordinal):
    """Right-aligned program tensor (plates at -1..-nplates, enumeration dims to their left at the
    positions ``dim_to_id`` names) -> packed Term (trace_struct.py:428-473 pack_tensors)."""
    t = tensor
    if t.dim() < nplates:
        t = t.reshape((1,) * (nplates - t.dim()) + tuple(t.shape))
    n_enum = t.dim() - nplates
    ids, sizes = [], []
    for i in range(n_enum):
        if t.shape[i] > 1:
            d = i - t.dim()
            if d not in dim_to_id:
                raise ValueError("enumeration dim {} of a log-factor of shape {} is not known to the "
                                 "enumeration bookkeeping (is max_plate_nesting too small?)"
                                 .format(d, tuple(tensor.shape)))
            ids.append(dim_to_id[d])
            sizes.append(t.shape[i])
    if len(ids) != n_enum:
        t = t.reshape(tuple(sizes) + tuple(t.shape[n_enum:]))
    return Term(t, ids, ordinal)


def align(term, ids):
    """The term's tensor laid out over the names ``ids`` (a superset of its own): size-1 dims for
    the names it does not depend on, its own dims permuted into that order."""
    x = term.dense()
    m = len(term.ids)
    if tuple(term.ids) == tuple(ids):
        return x
    pos = {v: i for i, v in enumerate(ids)}
    order = sorted(range(m), key=lambda i: pos[term.ids[i]])
    if order != list(range(m)):
        x = x.permute(order + list(range(m, x.dim())))
    present = set(term.ids)
    index = tuple(slice(None) if v in present else None for v in ids)
    return x[index] if len(ids) != m else x


class LazyGather:
    """log-factor b[t, w, d] = table[t, index[w, d]] kept un-materialised (observed Categorical
    whose logits were gathered by an enumerated value: examples/lda.py:68-70)."""

    def __init__(self, table, index):
        self.table, self.index = table, index   # [T,V], int64 [W,D]

    def materialize(self):
        return self.table[:, self.index]                   # [T, W, D]: packed, two plates


class LazyFamily:
    """log-factor f[k, (b), n] = log p(x[n] | p0[k, b], p1[k, b]) of an OBSERVED element-wise site whose parameters
    were indexed by an enumerated value (``Normal(locs[z], scale)`` under ``plate(n)``: a plated mixture's
    likelihood), kept un-materialised: the leaf kernel evaluates it in registers (csrc/mixture.hip).  ``b``: an
    optional outer plate the parameters vary over (vectorised chains / particles).  ``p0`` / ``p1``: [K or 1,
    B or 1] views of the parameters; ``batch_dim``: the outer plate's tensor dim (negative) or None, ``B`` its size.
    ``packed``: a callable that makes the packed tensor the generic elimination needs when the pattern does not
    match."""

    def __init__(self, dist_id, value, K, p0, p1, packed, batch_dim=None, B=1, D=None):
        self.dist_id, self.value, self.K, self.p0, self.p1, self._packed = dist_id, value, K, p0, p1, packed
        self.batch_dim, self.B = batch_dim, B
        # D: the event size of a diagonal Normal over features (value [N, D], p0 / p1 [K | 1, B | 1, D | 1]); None:
        # scalar observations (p0 / p1 [K | 1, B | 1])
        self.D = D

    def materialize(self):
        return self._packed()


FUSED_MIXTURE = True          # the mixture leaf goes through pa_mixture_fwd_bwd


@_dispatcher_op("mixture_factor")
class _MixtureFactor(torch.autograd.Function):
    """S[b] = sum_n logsumexp_k(a[b, k] + log p(x[n] | p0[b, k], p1[b, k])) with its gradient from the same pass
    (pa_mixture_fwd_bwd): nothing of size K N is written.  a: [B, K]; p0 / p1: flat, addressed b * bs + k * s."""

    @staticmethod
    def forward(ctx, dist_id, s0, s1, bs0, bs1, x, a, p0, p1):
        out = kernels.mixture_fwd_bwd(dist_id, x, a, p0, s0, p1, s1, bs0, bs1)      # [B, 1 + 3 K]
        K = a.shape[-1]
        g = out[:, 1:].to(a.dtype)
        ctx.strides = (s0, s1, bs0, bs1)
        ctx.save_for_backward(g[:, :K], g[:, K:2 * K], g[:, 2 * K:])
        return out[:, 0].to(a.dtype)

    @staticmethod
    def backward(ctx, g):
        da, d0, d1 = ctx.saved_tensors
        s0, s1, bs0, bs1 = ctx.strides
        g = g.reshape(-1, 1)

        def param_grad(d, s_, bs):
            d = g * d                                   # [B, K]
            if s_ == 0:
                d = d.sum(1, keepdim=True)
            if bs == 0:
                d = d.sum(0, keepdim=True)
            return d.reshape(-1)

        return (None, None, None, None, None, None, g * da, param_grad(d0, s0, bs0),
                param_grad(d1, s1, bs1) if ctx.needs_input_grad[8] else None)


@_dispatcher_op("mixture_diag_factor")
class _MixtureDiagFactor(torch.autograd.Function):
    """The mixture leaf for a diagonal Normal over D features (pa_mixture_diag_normal_fwd_bwd): a [B, K];
    loc / scale [B | 1, K | 1, D | 1]."""

    @staticmethod
    def forward(ctx, x, a, loc, scale):
        S, da, dl, dc = kernels.mixture_diag_normal_fwd_bwd(x, a, loc, scale)
        dt = a.dtype
        ctx.shapes = (tuple(loc.shape), tuple(scale.shape))
        ctx.save_for_backward(da.to(dt), dl.to(dt), dc.to(dt))
        return S.to(dt)

    @staticmethod
    def backward(ctx, g):
        da, dl, dc = ctx.saved_tensors
        g1, g2 = g.reshape(-1, 1), g.reshape(-1, 1, 1)

        def to_shape(d, shape):
            d = g2 * d                                  # [B, K, D]
            for ax in range(3):
                if shape[ax] == 1 and d.shape[ax] != 1:
                    d = d.sum(ax, keepdim=True)
            return d

        return None, g1 * da, to_shape(dl, ctx.shapes[0]), to_shape(dc, ctx.shapes[1])


def _try_fused_mixture(terms, sum_ids, contract_frames):
    """The mixture leaf: one enumerated name k, ONE lazy observed-family term over (k, [b,] n), every other term a
    function of k (and b) alone -- constant along the data plate --, and the data plate n (dim -1) is contracted
    here; an outer plate b the parameters vary over (vectorised chains / particles) stays, or is summed too when the
    caller contracts it here as well.  Returns a tensor (0-dim, or the plate block with n summed) or None when the
    pattern does not match."""
    if not FUSED_MIXTURE or len(sum_ids) != 1 or len(terms) < 1:
        return None
    lazy = [t for t in terms if t.tensor is None and isinstance(t.lazy, LazyFamily)]
    dense = [t for t in terms if t.tensor is not None]
    if len(lazy) != 1 or len(lazy) + len(dense) != len(terms):
        return None
    (kid,) = sum_ids
    lz = lazy[0].lazy
    plates = {f.dim for f in lazy[0].ordinal}
    frames = {f.dim for f in contract_frames}
    want = {-1} if lz.batch_dim is None else {-1, lz.batch_dim}
    if lazy[0].ids != (kid,) or plates != want or -1 not in frames or not frames <= plates:
        return None
    x, K, B = lz.value, lz.K, lz.B
    if K > kernels.MIXTURE_MAX_K or K < 1 or B > 65535:
        return None
    nplates = None
    a = None
    for t in dense:
        if t.ids != (kid,) or not isinstance(t.tensor, torch.Tensor) or t.tensor.dtype != x.dtype \
                or not kernels.on_device(t.tensor) or t.tensor.shape[0] != K:
            return None
        blk = tuple(t.tensor.shape[1:])                # the plate block, right-aligned
        if nplates is None:
            nplates = len(blk)
        big = [i - len(blk) for i, s_ in enumerate(blk) if s_ > 1]
        if big and (big != [lz.batch_dim] or blk[lz.batch_dim] != B):
            return None                                # varies along the data plate (per-row weights) or elsewhere
        tk = t.tensor.reshape(K, -1).t()               # [B or 1, K]
        a = tk if a is None else a + tk
    if a is None:                                      # no assignment probabilities here
        a = x.new_zeros((1, K))
    a = a.expand(B, K).contiguous()

    def flat(p):
        # [Kp, Bp] view -> (flat tensor addressed b * bs + k * s, s, bs)
        if p is None:
            return None, 0, 0
        Kp, Bp = p.shape
        if Kp not in (1, K) or Bp not in (1, B):
            return False, 0, 0
        q = p.t().contiguous().reshape(-1)             # [Bp, Kp] row-major
        return q, (1 if Kp == K and K > 1 else 0), (Kp if Bp == B and B > 1 else 0)

    if lz.D is not None:                               # a diagonal Normal over features: value [N, D]
        if lz.p1 is None:
            return None
        S = _MixtureDiagFactor.invoke(x.contiguous(), a, lz.p0.transpose(0, 1).contiguous(),
                                      lz.p1.transpose(0, 1).contiguous())                          # [B]
    else:
        p0, s0, bs0 = flat(lz.p0)
        p1, s1, bs1 = flat(lz.p1)
        if p0 is False or p1 is False or p0 is None:
            return None
        S = _MixtureFactor.invoke(int(lz.dist_id), s0, s1, bs0, bs1, x.contiguous(), a, p0, p1)   # [B]
    if lz.batch_dim is None or lz.batch_dim in frames:
        return S.sum()                                 # every plate contracted here
    if nplates is None:
        nplates = -lz.batch_dim
    shape = [1] * nplates
    shape[lz.batch_dim] = B
    return S.reshape(shape)


@_dispatcher_op("lda_factor")
class _LdaFactor(torch.autograd.Function):
    """sum_{d,w} logsumexp_t(log_theta[d,t] + log_phi[t, words[w,d]]) and its gradient from ONE
    pass over the int64 word ids (pa_lda_factor_fwd_bwd)."""

    @staticmethod
    def forward(ctx, words, log_theta, log_phi):
        out_doc, g_theta, g_phi = kernels.lda_factor_fwd_bwd(words, log_theta, log_phi)
        ctx.save_for_backward(g_theta, g_phi)
        return out_doc.sum()

    @staticmethod
    def backward(ctx, g):
        g_theta, g_phi = ctx.saved_tensors
        return None, g * g_theta, g * g_phi


@_dispatcher_op("logchain")
class _LogChain(torch.autograd.Function):
    """log Z of a chain of enumerated variables for every batch element (forward algorithm) with
    the forward-backward posteriors as gradient, ONE launch (pa_logchain_fwd_bwd)."""

    @staticmethod
    def forward(ctx, unary, pairwise):
        log_z, g_u, g_p = kernels.logchain_fwd_bwd(unary, pairwise)
        ctx.save_for_backward(g_u, g_p)
        ctx.pair_shape = pairwise.shape
        return log_z

    @staticmethod
    def backward(ctx, g):
        from ..distributions.fused import _sum_to
        g_u, g_p = ctx.saved_tensors
        d_pair = g.reshape(-1, 1, 1, 1) * g_p
        if len(ctx.pair_shape) < 4:       # potentials shared by the batch (and by the steps)
            d_pair = _sum_to(d_pair, d_pair.new_empty(ctx.pair_shape))
        return g.reshape(-1, 1, 1) * g_u, d_pair


def _try_fused_chain(terms, sum_ids):
    """A component whose names form a simple path (every term mentions one or two of them, the
    two-name terms link consecutive variables: an HMM written with pyro.markov) is summed out by
    the fused forward-backward kernel.  Returns the packed tensor [*plate block] or None."""
    T = len(sum_ids)
    if T < 3 or not terms:
        return None
    proto = None
    for t in terms:
        if t.tensor is None or not (1 <= len(t.ids) <= 2) or not set(t.ids) <= sum_ids:
            return None
        proto = t.tensor if proto is None else proto
    if proto.dtype not in (torch.float32, torch.float64) or \
            not kernels.on_device(proto):
        return None
    adj = {v: set() for v in sum_ids}
    for t in terms:
        if len(t.ids) == 2:
            a, b = t.ids
            adj[a].add(b)
            adj[b].add(a)
    ends = sorted(v for v, n in adj.items() if len(n) == 1)
    if len(ends) != 2 or any(len(n) > 2 or len(n) == 0 for n in adj.values()):
        return None
    order, prev = [ends[0]], None
    while len(order) < T:
        nxt = [v for v in adj[order[-1]] if v != prev]
        if len(nxt) != 1:
            return None
        prev = order[-1]
        order.append(nxt[0])
    if len(set(order)) != T:
        return None
    pos = {v: i for i, v in enumerate(order)}
    nplates = None
    K = None
    batch = ()
    for t in terms:
        m = len(t.ids)
        x = t.tensor
        nplates = x.dim() - m if nplates is None else nplates
        if x.dim() - m != nplates or any(k != x.shape[0] for k in x.shape[:m]):
            return None
        K = x.shape[0] if K is None else K
        if x.shape[0] != K:
            return None
        batch = torch.broadcast_shapes(batch, tuple(x.shape[m:]))
    if K > 64:
        return None
    B = 1
    for n in batch:
        B *= int(n)
    unary = [None] * T
    pair = [None] * (T - 1)
    for t in terms:
        if len(t.ids) == 1:
            i = pos[t.ids[0]]
            unary[i] = t.tensor if unary[i] is None else unary[i] + t.tensor
        else:
            a, b = t.ids
            if abs(pos[a] - pos[b]) != 1:
                return None
            first = a if pos[a] < pos[b] else b
            x = align(t, (first, b if first == a else a))
            i = pos[first]
            pair[i] = x if pair[i] is None else pair[i] + x
    zeros_u = proto.new_zeros((K,) + (1,) * nplates)
    U = torch.stack([(zeros_u if u is None else u).expand((K,) + tuple(batch)) for u in unary])
    P = torch.stack([p.expand((K, K) + tuple(batch)) for p in pair])
    nb = len(batch)
    U = U.permute(tuple(range(2, 2 + nb)) + (0, 1)).reshape(B, T, K).contiguous()
    P = P.permute(tuple(range(3, 3 + nb)) + (0, 1, 2)).reshape(B, T - 1, K, K).contiguous()
    return _LogChain.invoke(U, P).reshape(tuple(batch))


FUSED_SUMPRODUCT = True       # eliminations go through pa_logsumexp_terms (one pass, no frame tensor)


@_dispatcher_op("logsumexp_terms")
class _LogSumExpTerms(torch.autograd.Function):
    """logsumexp over one dim of the sum of up to four broadcast log-factors: ONE forward launch
    reading every factor through its own strides (pa_logsumexp_terms), ONE backward launch writing
    the posterior weights G (pa_logsumexp_terms_grad); a factor's gradient is G summed over the
    dims the factor does not have (pa_sum_to_nd).  Replaces adds that materialise the frame,
    torch.logsumexp's three passes, and the autograd duals of all of them."""

    @staticmethod
    def forward(ctx, frame, rdim, *terms):
        out = kernels.logsumexp_terms(terms, frame, rdim)
        ctx.frame, ctx.rdim = frame, rdim
        ctx.save_for_backward(out, *terms)
        return out

    @staticmethod
    def backward(ctx, g):
        from ..distributions.fused import _sum_to
        out, terms = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        if torch.is_grad_enabled():
            # create_graph=True (Hessians, Newton steps through an enumerated model): the kernel's
            # gradients carry no autograd history, so the first-order gradient is re-derived from
            # the same expression written with torch operators and stays differentiable
            with torch.enable_grad():
                total = terms[0].expand(ctx.frame)
                for t in terms[1:]:
                    total = total + t
                lse = torch.logsumexp(total, ctx.rdim)
                inputs = [t for t, need in zip(terms, ctx.needs_input_grad[2:]) if need]
                got = iter(torch.autograd.grad(lse, inputs, g, create_graph=True, allow_unused=True))
            return (None, None) + tuple(next(got) if need else None
                                        for need in ctx.needs_input_grad[2:])
        G = kernels.logsumexp_terms_grad(terms, ctx.frame, ctx.rdim, out, g)
        grads = [_sum_to(G, t) if need else None
                 for t, need in zip(terms, ctx.needs_input_grad[2:])]
        return (None, None) + tuple(grads)


def _sumproduct(terms, sum_ids):
    """logsumexp over the names ``sum_ids`` of the aligned sum of the terms -> (tensor, ids)."""
    ids = sorted(set().union(*(t.ids for t in terms))) if terms else []
    drop = [i for i, v in enumerate(ids) if v in sum_ids]
    keep_ids = [v for v in ids if v not in sum_ids]
    aligned = [align(t, ids) for t in terms]
    if FUSED_SUMPRODUCT and len(drop) == 1 and 1 <= len(aligned) <= 4:
        x0 = aligned[0]
        ok = all(isinstance(x, torch.Tensor) and x.dtype == x0.dtype and x.device == x0.device
                 for x in aligned) and x0.dtype in (torch.float32, torch.float64) \
            and kernels.on_device(x0)
        if ok:
            nd = max(x.dim() for x in aligned)
            # right-align the plate blocks: every term is [its ids..., *plates]; pad between the
            # id dims and the plate dims so that all terms have the same rank
            m = len(ids)
            padded = [x if x.dim() == nd else x.reshape(tuple(x.shape[:m]) + (1,) * (nd - x.dim())
                                                        + tuple(x.shape[m:])) for x in aligned]
            frame = tuple(torch.broadcast_shapes(*(x.shape for x in padded)))
            if len(frame) <= 6 and all(s > 0 for s in frame):
                out = _LogSumExpTerms.invoke(frame, drop[0], *padded)
                return out, keep_ids
    total = None
    for x in aligned:
        total = x if total is None else total + x
    if drop:
        total = torch.logsumexp(total, dim=drop)
    return total, keep_ids


def _eliminate(terms, sum_ids):
    """Variable elimination: sum out ``sum_ids`` one name at a time, always the one whose
    neighbourhood (union of names over the terms that mention it) is smallest.  Returns the
    remaining terms (those that never mentioned an eliminated name + the messages)."""
    terms = list(terms)
    pending = set(sum_ids)
    while pending:
        best, best_cost = None, None
        for v in sorted(pending):
            nb = set()
            for t in terms:
                if v in t.ids:
                    nb.update(t.ids)
            if not nb:
                continue
            if best is None or len(nb) < best_cost:
                best, best_cost = v, len(nb)
        if best is None:
            break
        group = [t for t in terms if best in t.ids]
        terms = [t for t in terms if best not in t.ids]
        tensor, ids = _sumproduct(group, {best})
        terms.append(Term(tensor, ids, group[0].ordinal))
        pending.discard(best)
    return terms


def _product(tensor, frames):
    """Plate product in log space = sum over the plates' tensor dims (kept as size 1)."""
    for f in sorted(frames, key=lambda f: f.dim):
        if tensor.dim() >= -f.dim and tensor.shape[f.dim] > 1:
            tensor = tensor.sum(f.dim, keepdim=True)
        elif f.size > 1:
            tensor = tensor * f.size        # constant along the plate: size identical copies
    return tensor


def _try_fused_lda(terms, sum_ids, contract_frames):
    """The LDA leaf: one enumerated name t, one dense term a[t, (w), d] constant along the inner
    plate w, one lazy gather over (w, d), and both plates are contracted here.  Returns a 0-dim
    tensor or None when the pattern does not match."""
    if len(sum_ids) != 1 or len(terms) != 2:
        return None
    lazy = [t for t in terms if t.tensor is None and isinstance(t.lazy, LazyGather)]
    dense = [t for t in terms if t.tensor is not None]
    if len(lazy) != 1 or len(dense) != 1:
        return None
    (tid,) = sum_ids
    lz, a = lazy[0].lazy, dense[0].tensor
    if lazy[0].ids != (tid,) or dense[0].ids != (tid,) or lz.index.dim() != 2 \
            or not kernels.on_device(a):
        return None
    plate_dims = {f.dim for f in contract_frames}
    if plate_dims != {-1, -2} or {f.dim for f in lazy[0].ordinal} != {-1, -2}:
        return None
    W, D = lz.index.shape
    T = lz.table.shape[0]
    if T > 64 or a.dim() != 3 or a.shape[0] != T or a.shape[-1] != D:
        return None
    if a.shape[-2] != 1 and a.stride(-2) != 0:
        return None           # genuinely word-dependent prior over topics: generic path
    log_theta = a.select(-2, 0).t().contiguous()           # [D, T]
    if lz.table.dtype != log_theta.dtype:
        return None
    return _LdaFactor.invoke(lz.index.contiguous(), log_theta, lz.table.contiguous())


def _partition(terms, sum_ids):
    """Connected components of the bipartite graph terms <-> enumerated names
    (reference: contract.py:44-84)."""
    remaining = list(terms)
    components = []
    seen = set()
    for d in sorted(sum_ids):
        if d in seen:
            continue
        comp_ids, comp_terms, frontier = {d}, [], [d]
        while frontier:
            x = frontier.pop()
            for t in list(remaining):
                if x in t.dims:
                    remaining.remove(t)
                    comp_terms.append(t)
                    for d2 in t.dims & sum_ids:
                        if d2 not in comp_ids:
                            comp_ids.add(d2)
                            frontier.append(d2)
        seen |= comp_ids
        if comp_terms:
            components.append((comp_terms, comp_ids))
    for t in remaining:       # terms without any contraction name: components of their own
        components.append(([t], set()))
    return components


def _contract_component(tensor_tree, sum_ids, reduce_all=False, keep_dims=()):
    """Contract all ``sum_ids`` out of one connected component by message passing from the
    deepest plate context to the root (reference: contract.py:87-165 with target_dims = {})."""
    id_to_ordinal = {}
    for ordinal, terms in tensor_tree.items():
        for term in terms:
            for d in sum_ids & term.dims:
                id_to_ordinal[d] = id_to_ordinal.get(d, ordinal) & ordinal
    ids_tree = {}
    for d, ordinal in id_to_ordinal.items():
        ids_tree.setdefault(ordinal, set()).add(d)
    min_ordinal = frozenset.intersection(*tensor_tree)
    while any(ids_tree.values()):
        leaf = max(tensor_tree, key=len)
        leaf_terms = tensor_tree.pop(leaf)
        leaf_ids = ids_tree.pop(leaf, set())
        for terms, ids in _partition(leaf_terms, leaf_ids):
            if leaf == min_ordinal:
                parent = leaf
            else:
                pending = set().union(*(t.dims for t in terms)) & sum_ids - ids
                parents = [o for o, d in ids_tree.items() if d & pending]
                parent = frozenset.union(*parents) if parents else min_ordinal
                if parent == leaf:
                    raise NotImplementedError(
                        "Expected tree-structured plate nesting, but found dependencies on "
                        "independent plates [{}]".format(", ".join(f.name for f in leaf)))
            contract_frames = leaf - parent
            # at the root of the component the caller may ask for the plates to be reduced too
            # (the ELBO sums every contracted factor completely): lets the fused kernel cover
            # logsumexp AND both plate sums
            # (keep_dims: plate dims the caller does NOT sum -- the chain plate of a vectorised potential)
            fuse_frames = frozenset(f for f in leaf if f.dim not in keep_dims) \
                if (reduce_all and parent == leaf) else contract_frames
            fused = _try_fused_lda(terms, ids, fuse_frames)
            if fused is None:
                fused = _try_fused_mixture(terms, ids, fuse_frames)
            chain = _try_fused_chain(terms, ids) if fused is None and FUSED_CHAIN else None
            if fused is not None:
                tensor, new_ids = fused, []
            elif chain is not None:
                tensor, new_ids = _product(chain, contract_frames), []
            else:
                tensor, new_ids = _sumproduct(_eliminate(terms, ids), set())
                tensor = _product(tensor, contract_frames)
            tensor_tree.setdefault(parent, []).append(Term(tensor, new_ids, parent))
    assert len(tensor_tree) == 1
    ordinal, terms = tensor_tree.popitem()
    tensor, ids = _sumproduct(terms, set())
    return ordinal, Term(tensor, ids, ordinal)


def contract_tensor_tree(tensor_tree, sum_ids, reduce_all=False, keep_dims=()):
    """{ordinal: [Term]} -> {ordinal: [Term]} with every name of ``sum_ids`` summed out; plate dims
    are contracted only as far as the message passing requires (reference: contract.py:168-210).
    With ``reduce_all`` a factor may come back already summed over its remaining plates -- except those at the
    tensor dims ``keep_dims``."""
    assert isinstance(tensor_tree, OrderedDict)
    all_terms = [t for terms in tensor_tree.values() for t in terms]
    contracted = OrderedDict()
    for terms, ids in _partition(all_terms, set(sum_ids)):
        component = OrderedDict()
        for t in terms:
            component.setdefault(t.ordinal, []).append(t)
        ordinal, term = _contract_component(component, ids, reduce_all, keep_dims)
        contracted.setdefault(ordinal, []).append(term)
    return contracted
