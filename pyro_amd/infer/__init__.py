from .elbo import ELBO, ELBOModule  # noqa: F401
from .enum import config_enumerate  # noqa: F401
from .svi import SVI  # noqa: F401
from .trace_elbo import Trace_ELBO  # noqa: F401
from .traceenum_elbo import TraceEnum_ELBO  # noqa: F401
from .tracegraph_elbo import TraceGraph_ELBO  # noqa: F401
from .trace_mean_field_elbo import TraceMeanField_ELBO  # noqa: F401
from .predictive import Predictive  # noqa: F401
from .mcmc import HMC, MCMC, NUTS, RandomWalkKernel  # noqa: F401

from .util import enable_validation, is_validation_enabled  # noqa: E402,F401


# ---- the reference's Jit* estimators --------------------------------------------------------------
# pyro compiles the loss with torch.jit.trace; here whole-step capture is SVI(hip_graph=True) (HIP
# graphs, not a tracing compiler), so the Jit* names are the same estimators: code that asks for them
# keeps working, and gets the captured step by passing hip_graph=True to SVI.
class JitTrace_ELBO(Trace_ELBO):
    pass


class JitTraceGraph_ELBO(TraceGraph_ELBO):
    pass


class JitTraceEnum_ELBO(TraceEnum_ELBO):
    pass


class JitTraceMeanField_ELBO(TraceMeanField_ELBO):
    pass
