"""Plated sum-product in log space: marginalise enumerated discrete variables out of a set of
log-factors that live in (tree-structured) plate contexts
(reference: pyro/ops/contract.py:44-202 _partition_terms / _contract_component /
contract_tensor_tree, with LogRing of pyro/ops/rings.py:178-216 and the log-space pairwise
einsum of pyro/ops/einsum/torch_log.py:14-55).

Differences in representation, not in algebra: factors stay UNPACKED (right-aligned tensor dims:
plates at -1..-max_plate_nesting, one fresh dim per enumerated variable to their left), so no
opt_einsum-style symbol table or contraction-path search is needed: a sum-contraction is a
broadcast add + logsumexp over the enumerated dims, a plate product is a sum over the plate dims.
The LDA-shaped leaf (a factor constant along the inner plate + an observed Categorical whose
logits are gathered by the enumerated value) is contracted by ONE fused HIP kernel
(pa_lda_factor_fwd_bwd) that never materialises the [T, words, docs] tensor.
"""
from collections import OrderedDict

import torch

from .. import kernels


_FUSED_NEEDS_DEVICE = True    # tests lift this to drive the fused route with the oracle kernel


class Term:
    """A log-factor: dense ``tensor`` (or a lazy gather), the enumerated dims it depends on and
    the plate context (``ordinal``: frozenset of vectorised CondIndepStackFrames) it lives in."""

    __slots__ = ("tensor", "dims", "ordinal", "lazy")

    def __init__(self, tensor, dims, ordinal, lazy=None):
        self.tensor, self.dims, self.ordinal, self.lazy = tensor, frozenset(dims), ordinal, lazy

    def dense(self):
        if self.tensor is None:
            self.tensor = self.lazy.materialize()
        return self.tensor


class LazyGather:
    """log-factor b[t, w, d] = table[t, index[w, d]] kept un-materialised (observed Categorical
    whose logits were gathered by an enumerated value: examples/lda.py:68-70)."""

    def __init__(self, table, index, enum_dim):
        self.table, self.index, self.enum_dim = table, index, enum_dim   # [T,V], int64 [W,D]

    def materialize(self):
        T = self.table.shape[0]
        out = self.table[:, self.index]                    # [T, W, D]
        extra = -self.enum_dim - out.dim()
        return out.reshape((T,) + (1,) * extra + tuple(self.index.shape))


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


def _sumproduct(terms, dims):
    """logsumexp over ``dims`` of the broadcast sum of the terms (dims kept as size 1 so the
    right-aligned layout survives)."""
    total = None
    for t in terms:
        x = t.dense()
        total = x if total is None else total + x
    if dims:
        total = torch.logsumexp(total, dim=sorted(dims), keepdim=True)
    return total


def _product(tensor, frames):
    """Plate product in log space = sum over the plates' tensor dims."""
    for f in sorted(frames, key=lambda f: f.dim):
        if tensor.dim() >= -f.dim and tensor.shape[f.dim] > 1:
            tensor = tensor.sum(f.dim, keepdim=True)
        elif f.size > 1:
            tensor = tensor * f.size        # constant along the plate: size identical copies
    return tensor


def _try_fused_lda(terms, dims, contract_frames):
    """The LDA leaf: dims == {t}, one dense term a[t, (w), d] constant along the inner plate w,
    one lazy gather over (w, d), and both plates are contracted here.  Returns a 0-dim tensor or
    None when the pattern does not match."""
    if len(dims) != 1 or len(terms) != 2:
        return None
    lazy = [t for t in terms if t.tensor is None and isinstance(t.lazy, LazyGather)]
    dense = [t for t in terms if t.tensor is not None]
    if len(lazy) != 1 or len(dense) != 1:
        return None
    (edim,) = dims
    lz, a = lazy[0].lazy, dense[0].tensor
    if lz.enum_dim != edim or lz.index.dim() != 2 or (_FUSED_NEEDS_DEVICE and not a.is_cuda):
        return None
    plate_dims = {f.dim for f in contract_frames}
    if plate_dims != {-1, -2} or {f.dim for f in lazy[0].ordinal} != {-1, -2}:
        return None
    W, D = lz.index.shape
    T = lz.table.shape[0]
    if T > 64 or a.dim() != -edim or a.shape[edim] != T or a.shape[-1] != D:
        return None
    if any(s != 1 for i, s in enumerate(a.shape) if i not in (a.dim() + edim, a.dim() - 1, a.dim() - 2)):
        return None
    if a.shape[-2] != 1 and a.stride(-2) != 0:
        return None           # genuinely word-dependent prior over topics: generic path
    a2 = a.select(-2, 0)                                   # [T, 1.., D]
    log_theta = a2.reshape(T, D).t().contiguous()          # [D, T]
    if lz.table.dtype != log_theta.dtype:
        return None
    return _LdaFactor.apply(lz.index.contiguous(), log_theta, lz.table.contiguous())


def _partition(terms, dims):
    """Connected components of the bipartite graph terms <-> enumerated dims
    (reference: contract.py:44-84)."""
    remaining = list(terms)
    components = []
    seen_dims = set()
    for d in sorted(dims, reverse=True):
        if d in seen_dims:
            continue
        comp_dims, comp_terms, frontier = {d}, [], [d]
        while frontier:
            x = frontier.pop()
            for t in list(remaining):
                if x in t.dims:
                    remaining.remove(t)
                    comp_terms.append(t)
                    for d2 in t.dims & dims:
                        if d2 not in comp_dims:
                            comp_dims.add(d2)
                            frontier.append(d2)
        seen_dims |= comp_dims
        if comp_terms:
            components.append((comp_terms, comp_dims))
    for t in remaining:       # terms without any contraction dim: components of their own
        components.append(([t], set()))
    return components


def _contract_component(tensor_tree, sum_dims, reduce_all=False):
    """Contract all ``sum_dims`` out of one connected component by message passing from the
    deepest plate context to the root (reference: contract.py:87-165 with target_dims = {})."""
    dim_to_ordinal = {}
    for ordinal, terms in tensor_tree.items():
        for term in terms:
            for d in sum_dims & term.dims:
                dim_to_ordinal[d] = dim_to_ordinal.get(d, ordinal) & ordinal
    dims_tree = {}
    for d, ordinal in dim_to_ordinal.items():
        dims_tree.setdefault(ordinal, set()).add(d)
    min_ordinal = frozenset.intersection(*tensor_tree)
    while any(dims_tree.values()):
        leaf = max(tensor_tree, key=len)
        leaf_terms = tensor_tree.pop(leaf)
        leaf_dims = dims_tree.pop(leaf, set())
        for terms, dims in _partition(leaf_terms, leaf_dims):
            if leaf == min_ordinal:
                parent = leaf
            else:
                pending = set().union(*(t.dims for t in terms)) & sum_dims - dims
                parents = [o for o, d in dims_tree.items() if d & pending]
                parent = frozenset.union(*parents) if parents else min_ordinal
                if parent == leaf:
                    raise NotImplementedError(
                        "Expected tree-structured plate nesting, but found dependencies on "
                        "independent plates [{}]".format(", ".join(f.name for f in leaf)))
            contract_frames = leaf - parent
            # at the root of the component the caller may ask for the plates to be reduced too
            # (the ELBO sums every contracted factor completely): lets the fused kernel cover
            # logsumexp AND both plate sums
            fuse_frames = leaf if (reduce_all and parent == leaf) else contract_frames
            fused = _try_fused_lda(terms, dims, fuse_frames)
            if fused is not None:
                tensor = fused
            else:
                tensor = _product(_sumproduct(terms, dims), contract_frames)
            new_dims = (set().union(*(t.dims for t in terms)) - dims) if terms else set()
            tensor_tree.setdefault(parent, []).append(Term(tensor, new_dims, parent))
    assert len(tensor_tree) == 1
    ordinal, terms = tensor_tree.popitem()
    tensor = _sumproduct(terms, set())
    return ordinal, Term(tensor, (), ordinal)


def contract_tensor_tree(tensor_tree, sum_dims, reduce_all=False):
    """{ordinal: [Term]} -> {ordinal: [Term]} with every enumerated dim summed out; plate dims are
    contracted only as far as the message passing requires (reference: contract.py:168-210).
    With ``reduce_all`` a factor may come back already summed over its remaining plates."""
    assert isinstance(tensor_tree, OrderedDict)
    all_terms = [t for terms in tensor_tree.values() for t in terms]
    contracted = OrderedDict()
    for terms, dims in _partition(all_terms, set(sum_dims)):
        component = OrderedDict()
        for t in terms:
            component.setdefault(t.ordinal, []).append(t)
        ordinal, term = _contract_component(component, dims, reduce_all)
        contracted.setdefault(ordinal, []).append(term)
    return contracted
