"""pyro.distributions.torch_distribution: the reference's module path for these names."""
from .base import (ExpandedDistribution, MaskedDistribution, TorchDistribution,  # noqa: F401
                   TorchDistributionMixin)
