import sys; sys.path.insert(0, ".")
import torch
from pyro_amd import kernels as k
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
P, N = 64, 1_000_000
logits = torch.randn((P, N), device=dev)
y = (torch.rand((1, N), device=dev) < 0.5).float()
us = timeit(lambda: k.dist_log_prob_sum(1, y, logits, None, None, 1.0, P, N))
print(f"site log_prob_sum Bernoulli [64,1e6]: {us:.1f} us {P*N*4/us/1e6:.3f} TB/s")
g = torch.ones((P, 1), device=dev)
us = timeit(lambda: k.dist_log_prob_grad(1, g, y, logits, None, None, 1.0, P, N, (False, True, False)))
print(f"site log_prob_grad Bernoulli [64,1e6]: {us:.1f} us {2*P*N*4/us/1e6:.3f} TB/s")
v = torch.randn((P, N), device=dev); loc = torch.randn((1, N), device=dev); sc = torch.rand((1, 1), device=dev) + 0.5
us = timeit(lambda: k.dist_log_prob_sum(0, v, loc, sc, None, 1.0, P, N))
print(f"site log_prob_sum Normal [64,1e6]: {us:.1f} us {P*N*4/us/1e6:.3f} TB/s")
