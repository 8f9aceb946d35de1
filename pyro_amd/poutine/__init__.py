"""pyro_amd.poutine -- effect handlers (same surface as pyro.poutine for the hot paths)."""
from . import runtime, settings  # noqa: F401
from .handlers import (BlockMessenger, CondIndepStackFrame, ConditionMessenger,  # noqa: F401
                       DoMessenger, EqualizeMessenger, EscapeMessenger, ReparamMessenger, broadcast, equalize, reparam, InferConfigMessenger, LiftMessenger,
                       SubstituteMessenger, do, escape, infer_config, lift, queue, substitute, EnumMessenger, MarkovMessenger, MaskMessenger, PlateMessenger, ReplayMessenger,
                       ScaleMessenger, SeedMessenger, TraceMessenger, UnconditionMessenger, block,
                       condition, enum, get_mask, markov, mask, replay, scale, seed, trace, uncondition)
from .runtime import Messenger, NonlocalExit, apply_stack, block_messengers, effectful  # noqa: F401
from .plate_messenger import block_plate  # noqa: F401
from .trace import Trace  # noqa: F401
from . import util  # noqa: F401
from .util import prune_subsample_sites, site_is_subsample  # noqa: F401
from .messenger import unwrap  # noqa: E402,F401
from .util import enable_validation, is_validation_enabled  # noqa: E402,F401
