"""Plane-image GLM kernel with / without the label moments (developer tool): kernel + finalize per call
from a hipGraph of 10 calls."""
import sys
import torch
sys.path.insert(0, ".")
from pyro_amd import kernels as k
from tools.bench_glm_planes import graph_time

dev = torch.device("cuda:0")
N, D, P = 1_000_000, 32, 64
X = torch.randn((N, D), device=dev)
y = (torch.rand((N,), device=dev) < 0.5).float()
w = torch.randn((P, D), device=dev) * 0.2
b = torch.randn((P,), device=dev)
planes = k.glm_pack_planes(X, fmt=k.GLM_PLANES_F16X2)
mom = k.glm_label_moments(X, y)
for rep in range(3):
    a, _ = graph_time(lambda: k.glm_bernoulli_planes_fwd_bwd(planes, y, w, b, 1.0, N, D))
    c, _ = graph_time(lambda: k.glm_bernoulli_planes_fwd_bwd(planes, y, w, b, 1.0, N, D, moments=mom))
    print("kernel + finalize: sums the term itself %.1f us, with label moments %.1f us" % (a, c))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); k.glm_label_moments(X, y, out=mom); e.record(); torch.cuda.synchronize()
print("pa_glm_label_moments (once per (X, y)): %.1f us" % (s.elapsed_time(e) * 1e3))
