"""Inference helpers (reference: pyro/infer/util.py: MultiFrameTensor :122-171, plate stacks)."""
import torch

from ..distributions.util import is_identically_zero


class MultiFrameTensor(dict):
    """Maps plate-stacks (tuples of frames) to tensors and sums them down to a target stack."""

    def __init__(self, *items):
        super().__init__()
        self.add(*items)

    def add(self, *items):
        for cond_indep_stack, value in items:
            frames = frozenset(f for f in cond_indep_stack if f.vectorized)
            assert all(f.dim < 0 and -value.dim() <= f.dim for f in frames)
            if frames in self:
                self[frames] = self[frames] + value
            else:
                self[frames] = value

    def sum_to(self, target_frames):
        total = None
        for frames, value in self.items():
            for f in frames:
                if f not in target_frames and value.shape[f.dim] != 1:
                    value = value.sum(f.dim, True)
            while value.shape and value.shape[0] == 1:
                value = value.squeeze(0)
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
