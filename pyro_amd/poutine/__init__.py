"""pyro_amd.poutine -- effect handlers (same surface as pyro.poutine for the hot paths)."""
from . import runtime, settings  # noqa: F401
from .handlers import (BlockMessenger, CondIndepStackFrame, ConditionMessenger,  # noqa: F401
                       DoMessenger, InferConfigMessenger, SubstituteMessenger, do, infer_config,
                       substitute, EnumMessenger, MarkovMessenger, MaskMessenger, PlateMessenger, ReplayMessenger,
                       ScaleMessenger, SeedMessenger, TraceMessenger, UnconditionMessenger, block,
                       condition, enum, get_mask, markov, mask, replay, scale, seed, trace, uncondition)
from .runtime import Messenger, NonlocalExit, apply_stack, effectful  # noqa: F401
from .trace import Trace  # noqa: F401
from .util import prune_subsample_sites, site_is_subsample  # noqa: F401
