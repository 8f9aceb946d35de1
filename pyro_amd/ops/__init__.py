"""Numerical building blocks of the HMC/NUTS path (reference: pyro/ops/integrator.py,
dual_averaging.py, welford.py, stats.py)."""
from . import contract, dual_averaging, integrator, stats, welford  # noqa: F401
