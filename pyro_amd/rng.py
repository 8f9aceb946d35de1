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
    """Reserve Philox blocks for ``n_elements`` draws; returns (seed, offset)."""
    per = 4 if dtype == torch.float32 else 2
    off = _STATE["offset"]
    _STATE["offset"] = off + (int(n_elements) + per - 1) // per
    return _STATE["seed"], off


def normal(shape, dtype, device):
    """Standard normal draws from the Philox stream (HIP kernel, GPU only)."""
    n = 1
    for s in shape:
        n *= int(s)
    seed, off = reserve(n, dtype)
    return kernels.philox_normal(tuple(shape), dtype, device, seed, off)


def uniform(shape, dtype, device):
    n = 1
    for s in shape:
        n *= int(s)
    seed, off = reserve(n, dtype)
    return kernels.philox_uniform(tuple(shape), dtype, device, seed, off)
