from .elbo import ELBO, ELBOModule  # noqa: F401
from .enum import config_enumerate  # noqa: F401
from .svi import SVI  # noqa: F401
from .trace_elbo import JitTrace_ELBO, Trace_ELBO  # noqa: F401
from .traceenum_elbo import TraceEnum_ELBO  # noqa: F401
from .tracegraph_elbo import TraceGraph_ELBO  # noqa: F401
from .trace_mean_field_elbo import TraceMeanField_ELBO  # noqa: F401
from .predictive import Predictive  # noqa: F401
from .mcmc import HMC, MCMC, NUTS, RandomWalkKernel  # noqa: F401

from .util import enable_validation, is_validation_enabled  # noqa: E402,F401


