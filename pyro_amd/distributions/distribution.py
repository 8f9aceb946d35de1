"""pyro.distributions.distribution: the reference's module path; ``Distribution`` is the root class users
test against with isinstance."""
import torch

Distribution = torch.distributions.Distribution
