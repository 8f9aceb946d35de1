"""Process-wide counter-based RNG state for the HIP kernels (Philox4x32-10).

``set_rng_seed`` mirrors pyro.set_rng_seed (reference: pyro/util.py:37-45): it seeds torch,
python and numpy *and* the Philox stream used by the fused kernels.  Every draw advances a
host-side 64-bit block offset, so the sequence is reproducible and independent of launch
geometry; ranks of a multi-GPU job de-correlate by seeding with ``seed + rank`` exactly as the
reference de-correlates chains (pyro/infer/mcmc/api.py:107).
"""
import random

import numpy as np
import torch

from . import kernels

_STATE = {"seed": 0, "offset": 0}
# While a hipGraph is being captured the block offset of a draw cannot be a launch constant (every
# replay would repeat the same numbers): draws are addressed relative to a device-resident base
# counter that the graph itself advances (pa_counter_add) -- see GraphCapture below.
_CAPTURE = {"active": None}


def set_rng_seed(seed):
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    _STATE["seed"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _STATE["offset"] = 0


def get_rng_state():
    return {"torch": torch.get_rng_state(), "random": random.getstate(),
            "numpy": np.random.get_state(), "philox": dict(_STATE)}


def set_rng_state(state):
    torch.set_rng_state(state["torch"])
    random.setstate(state["random"])
    np.random.set_state(state["numpy"])
    _STATE.update(state["philox"])


def reserve(n_elements, dtype):
    """Reserve Philox blocks for ``n_elements`` draws; returns (seed, offset, offset_dev) where
    ``offset_dev`` is None (eager) or the device base counter of the active graph capture."""
    per = 4 if dtype == torch.float32 else 2
    off = _STATE["offset"]
    _STATE["offset"] = off + (int(n_elements) + per - 1) // per
    cap = _CAPTURE["active"]
    if cap is not None:
        return _STATE["seed"], off - cap.start, cap.base
    return _STATE["seed"], off, None


def reserve_blocks(n_blocks):
    """Reserve ``n_blocks`` whole Philox blocks (draws that key one block per element, such as the
    Gamma sampler's rejection loop); returns (seed, offset, offset_dev) like ``reserve``."""
    off = _STATE["offset"]
    _STATE["offset"] = off + int(n_blocks)
    cap = _CAPTURE["active"]
    if cap is not None:
        return _STATE["seed"], off - cap.start, cap.base
    return _STATE["seed"], off, None


def normal(shape, dtype, device):
    """Standard normal draws from the Philox stream (HIP kernel, GPU only)."""
    n = 1
    for s in shape:
        n *= int(s)
    seed, off, off_dev = reserve(n, dtype)
    return kernels.philox_normal(tuple(shape), dtype, device, seed, off, off_dev)


_default_normal = normal


def uniform(shape, dtype, device):
    n = 1
    for s in shape:
        n *= int(s)
    seed, off, off_dev = reserve(n, dtype)
    return kernels.philox_uniform(tuple(shape), dtype, device, seed, off, off_dev)


class GraphCapture:
    """Makes the Philox stream replay-safe: inside ``with GraphCapture(device) as cap`` every draw
    reads its block offset as ``*base + relative offset``; ``cap.finish()`` (still inside the
    capture) appends the node that advances ``*base`` by the blocks one replay consumes.  A replay
    therefore draws exactly the numbers the same step would draw eagerly: call
    ``cap.before_replay()`` to (re)synchronise the base with the host-side offset and
    ``cap.after_replay()`` to advance the host mirror."""

    def __init__(self, device):
        # device words {Philox block base, publish sequence}: pa_publish_scalar's `counter`
        self.base = torch.zeros((2,), dtype=torch.int64, device=device)
        self.start = None
        self.used = None
        self._base_value = None      # what *base holds on the device, if known

    def __enter__(self):
        assert _CAPTURE["active"] is None, "nested RNG graph captures are not supported"
        self.start = _STATE["offset"]
        _CAPTURE["active"] = self
        return self

    def finish(self, publish=None):
        """Append the node that advances ``*base`` by the blocks one replay consumes.  With
        ``publish`` = (device scalar, pinned float64[1], pinned int64[1]) the same node also hands
        that scalar (the step's loss) to the host (kernels.publish_scalar)."""
        self.used = _STATE["offset"] - self.start
        if publish is not None:
            kernels.publish_scalar(publish[0], publish[1], publish[2], self.base, self.used)
        else:
            kernels.counter_add(self.base, self.used)

    def finish_args(self, publish):
        """The arguments ``finish(publish)`` would launch with, for a caller that folds the node
        into its own last launch (the optimizer update: kernels.adam_step(publish=...))."""
        self.used = _STATE["offset"] - self.start
        return (publish[0], publish[1], publish[2], self.base, self.used)

    def __exit__(self, *exc):
        _CAPTURE["active"] = None
        _STATE["offset"] = self.start      # capturing executes nothing: no draws were consumed

    def before_replay(self):
        if self._base_value != _STATE["offset"]:
            self.base[:1].fill_(_STATE["offset"])
            self._base_value = _STATE["offset"]

    def after_replay(self):
        _STATE["offset"] += self.used
        self._base_value += self.used
