from .guides import (AutoCallable, AutoContinuous, AutoDelta, AutoDiagonalNormal, AutoGuide,  # noqa: F401
                     AutoGuideList, AutoLowRankMultivariateNormal, AutoMultivariateNormal, AutoNormal)
from .initialization import (InitMessenger, init_to_feasible, init_to_generated, init_to_mean,  # noqa: F401
                             init_to_median, init_to_sample, init_to_uniform, init_to_value)
