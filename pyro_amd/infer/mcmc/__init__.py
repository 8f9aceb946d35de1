"""HMC / NUTS, vectorised over chains on the GPU (reference: pyro/infer/mcmc/__init__.py)."""
from .adaptation import ArrowheadMassMatrix, WarmupAdapter  # noqa: F401
from .api import MCMC  # noqa: F401
from .hmc import HMC  # noqa: F401
from .mcmc_kernel import MCMCKernel  # noqa: F401
from .nuts import NUTS  # noqa: F401
from .potentials import GaussianPotential  # noqa: F401
from .rwkernel import RandomWalkKernel  # noqa: F401
from .util import initialize_model  # noqa: F401
