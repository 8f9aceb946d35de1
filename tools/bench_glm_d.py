"""GLM site at D = 64 / 128 (N = 1e6, P = 64): the plane-image kernel with feature tiles against the kernels
that split X on the fly (developer tool): kernel + finalize per call from a hipGraph of 10 calls."""
import sys
import torch
sys.path.insert(0, ".")
from pyro_amd import kernels as k
from tools.bench_glm_planes import graph_time

dev = torch.device("cuda:0")
N, P = 1_000_000, 64
for D in (64, 128, 48, 100):
    X = torch.randn((N, D), device=dev)
    y = (torch.rand((N,), device=dev) < 0.5).float()
    w = torch.randn((P, D), device=dev) * 0.1
    b = torch.randn((P,), device=dev)
    k.glm_set_planes_mode(k.GLM_PLANES_OFF)
    a, _ = graph_time(lambda: k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0))
    k.glm_set_planes_mode(k.GLM_PLANES_AUTO)
    planes = k.glm_pack_planes(X, fmt=k.GLM_PLANES_F16X2)
    c, _ = graph_time(lambda: k.glm_bernoulli_planes_fwd_bwd(planes, y, w, b, 1.0, N, D))
    alg = N * (4 * D + 4)
    print("D=%3d: on the fly %7.1f us (%.2f TB/s alg)   plane image with feature tiles %7.1f us (%.2f TB/s alg)"
          % (D, a, alg / a / 1e6, c, alg / c / 1e6))
    del X, planes
