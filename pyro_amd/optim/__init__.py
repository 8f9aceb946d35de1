from .optim import (SGD, Adam, AdamW, ClippedAdam, NoUpdate, PyroLRScheduler, PyroOptim, RMSprop,  # noqa: F401
                    TorchAdam, _torch_wrappers)

for _name, _factory in _torch_wrappers().items():
    if _name not in globals():          # Adam / ClippedAdam / SGD ... keep the definitions above
        globals()[_name] = _factory
del _name, _factory
from .rccl import RcclOptimizer  # noqa: F401
