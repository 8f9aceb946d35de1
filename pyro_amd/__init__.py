"""pyro_amd -- MI355X-native backend for Pyro's SVI (Trace_ELBO / TraceEnum_ELBO under
pyro.plate) and HMC/NUTS hot paths.  The module layout and names mirror ``pyro`` for these
paths (``pyro_amd.sample``, ``pyro_amd.plate``, ``pyro_amd.infer.SVI`` ...) so a model written
for the reference runs after ``import pyro_amd as pyro``; the numerics are hand-written HIP
kernels for gfx950 behind the C-ABI in include/pyro_amd.h.  GPU tensors only: there is no CPU
fallback.
"""
from . import distributions, infer, nn, ops, optim, poutine, settings  # noqa: F401
from .primitives import (barrier, clear_param_store, deterministic, enable_validation, factor,  # noqa: F401
                         iarange, irange,
                         get_param_store, module, param, plate, plate_stack, random_module, sample, subsample,
                         set_rng_seed, validation_enabled)
from .poutine.handlers import condition, do, markov  # noqa: F401  (pyro/__init__.py:7)

__version__ = "0.1.0"

import logging as _logging  # noqa: E402

log = _logging.getLogger("pyro_amd")      # pyro.log (pyro/logger.py)
log.setLevel(_logging.INFO)
