from .optim import (SGD, Adam, AdamW, ClippedAdam, PyroOptim, RMSprop, TorchAdam)  # noqa: F401
from .rccl import RcclOptimizer  # noqa: F401
