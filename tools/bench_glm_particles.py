"""Developer tool: the plane-image GLM kernel (+ finalize) at P = 64 .. 1024 particles / chains, N = 1e5 and 1e6, under
the wave geometries of csrc/glm_planes16.h: the default's choice (wide from N ~ 4e5 on), wide at every N (2 x 2 up to
64, 2 x 4 up to 128, 1 x 8 above: pa_glm_planes_tune(13, 0)), at most 2 x 4 (tune 12) and 2 x 2 passes only (tune 11)."""
import sys
import torch
sys.path.insert(0, ".")
from pyro_amd import kernels as k, examples
dev = torch.device("cuda", 0)
for N in (100_000, 1_000_000):
    X, y = examples.synthetic_logreg_data(N, 32, dev, seed=0)
    planes = k.glm_pack_planes(X, fmt=k.GLM_PLANES_F16X2)
    mom = k.glm_label_moments(X, y)
    for P in (64, 128, 256, 512, 1024):
        w = torch.randn((P, 32), device=dev) * 0.1
        b = torch.randn((P,), device=dev)
        row = []
        for tune in (0, 13, 12, 11):
            k.glm_planes_tune(tune, 0)
            for _ in range(3):
                k.glm_bernoulli_planes_fwd_bwd(planes, y, w, b, 1.0, N, 32, moments=mom)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    k.glm_bernoulli_planes_fwd_bwd(planes, y, w, b, 1.0, N, 32, moments=mom)
                e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / 20 * 1e3)
            row.append(best)
        k.glm_planes_tune(0, 0)
        print("N=%d P=%4d: default %.1f us | wide %.1f us | at most 2 x 4: %.1f us | 2 x 2 passes: %.1f us" % (N, P, *row))
