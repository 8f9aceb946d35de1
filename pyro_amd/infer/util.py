"""Inference helpers (reference: pyro/infer/util.py: MultiFrameTensor :122-171, plate stacks)."""
import torch

from ..distributions.util import is_identically_zero
from ..util import torch_item, zero_grads  # noqa: F401  (their reference home is pyro/infer/util.py)

_VALIDATION_ENABLED = __debug__     # pyro.infer's own switch (pyro/infer/util.py:23-36)


def enable_validation(is_validate):
    global _VALIDATION_ENABLED
    _VALIDATION_ENABLED = bool(is_validate)


def is_validation_enabled():
    return _VALIDATION_ENABLED


class MultiFrameTensor:
    """Per-site terms of an ELBO estimator filed by the set of vectorised plates they live in
    (the role of pyro/infer/util.py:122-171 in Trace_ELBO._compute_log_r, trace_elbo.py:20-29).

    ``add((stack, tensor), ...)`` files a tensor under the vectorised frames of its plate stack
    (terms of the same plate set are summed at once); ``sum_to(stack)`` returns the total of all
    filed terms with every plate that is NOT in ``stack`` summed out -- the downstream-cost a
    score-function site inside ``stack`` sees.  Plate dims are negative (counted from the right),
    so a filed tensor can be reduced with one keepdim sum over the foreign plates and its leading
    singleton dims dropped afterwards.
    """

    def __init__(self, *items):
        self._by_plates = {}
        self.add(*items)

    def __len__(self):
        return len(self._by_plates)

    def items(self):
        return self._by_plates.items()

    def add(self, *items):
        for stack, value in items:
            plates = frozenset(f for f in stack if f.vectorized)
            for f in plates:
                if not (-value.dim() <= f.dim < 0):
                    raise ValueError("term of shape {} does not reach plate {!r} (dim {})".format(
                        tuple(value.shape), f.name, f.dim))
            held = self._by_plates.get(plates)
            self._by_plates[plates] = value if held is None else held + value

    def sum_to(self, target_frames):
        keep = set(target_frames)
        total = None
        for plates, value in self._by_plates.items():
            foreign = [f.dim for f in plates if f not in keep and value.shape[f.dim] != 1]
            if foreign:
                value = value.sum(foreign, keepdim=True)
            lead = 0
            while lead < value.dim() and value.shape[lead] == 1:
                lead += 1
            if lead:
                value = value.reshape(value.shape[lead:])
            total = value if total is None else total + value
        return 0.0 if total is None else total


def get_plate_stacks(trace):
    return {name: [f for f in node["cond_indep_stack"] if f.vectorized]
            for name, node in trace.nodes.items()
            if node["type"] == "sample" and not _is_subsample(node)}


def _is_subsample(node):
    from ..poutine.util import site_is_subsample
    return site_is_subsample(node)


def check_fully_reparametrized(guide_site):
    log_prob, score_function_term, entropy_term = guide_site["score_parts"]
    fully_rep = (guide_site["fn"].has_rsample and not is_identically_zero(entropy_term)
                 and is_identically_zero(score_function_term))
    if not fully_rep:
        raise NotImplementedError("All distributions in the guide must be fully reparameterized.")
