"""Plane-image GLM kernel against the on-the-fly bf16x3 kernel (developer tool): kernel + finalize
time per call from a hipGraph of 10 calls (no host launch cost in the number), all ring depths and
workgroups-per-CU settings, and the error of every variant against float64."""
import sys

import torch

sys.path.insert(0, ".")
from pyro_amd import kernels as k

dev = torch.device("cuda:0")


def graph_time(fn, calls=10, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            out = fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * calls) * 1e3, out


def pmc_mode():
    """A handful of eager calls of both kernels (for rocprofv3 --pmc passes)."""
    N, D, P = 1_000_000, 32, 64
    X = torch.randn((N, D), device=dev)
    y = (torch.rand((N,), device=dev) < 0.5).float()
    w = torch.randn((P, D), device=dev) * 0.2
    b = torch.randn((P,), device=dev)
    planes = k.glm_pack_planes(X)
    k.glm_set_planes_mode(k.GLM_PLANES_OFF)
    for _ in range(4):
        k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
        k.glm_bernoulli_planes_fwd_bwd(planes, y, w, b, 1.0, N, D)
    torch.cuda.synchronize()


def main():
    if "--pmc" in sys.argv:
        return pmc_mode()
    shapes = [(1_000_000, 32, 64)]
    if "--more" in sys.argv:
        shapes += [(1_000_000, 32, 128), (1_000_000, 16, 64), (10_000_000, 32, 64), (100_000, 32, 64)]
    for (N, D, P) in shapes:
        X = torch.randn((N, D), device=dev)
        y = (torch.rand((N,), device=dev) < 0.5).float()
        w = torch.randn((P, D), device=dev) * 0.2
        b = torch.randn((P,), device=dev)
        if N <= 1_000_000:
            Xd, wd = X.double(), w.double()
            lg = wd @ Xd.t() + b.double()[:, None]
            llr = (y.double() * lg - torch.nn.functional.softplus(lg)).sum(1)
            gr = (y.double() - torch.sigmoid(lg)) @ Xd
            del Xd, lg
        else:
            llr = gr = None

        def err(out):
            if llr is None:
                return float("nan"), float("nan")
            return (((out[0].double() - llr).abs() / llr.abs()).max().item(),
                    ((out[1].double() - gr).abs().max() / gr.abs().max()).item())

        k.glm_set_planes_mode(k.GLM_PLANES_OFF)
        us, out = graph_time(lambda: k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0))
        print(f"N={N} D={D} P={P}")
        print(f"  on-the-fly bf16x3           {us:8.1f} us   rel err ll {err(out)[0]:.1e} gw {err(out)[1]:.1e}")
        for fmt, fname, settings in ((k.GLM_PLANES_BF16X3, "bf16x3", [(3, 3)]),
                                     (k.GLM_PLANES_F16X2, "f16x2", [(3, 2), (9, 2), (10, 2), (9, 1), (5, 2), (3, 2)])):
            planes = k.glm_pack_planes(X, fmt=fmt)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            k.glm_pack_planes(X, out=planes)
            e.record()
            torch.cuda.synchronize()
            print(f"  pack {fname} (once per X)    {s.elapsed_time(e) * 1e3:8.1f} us")
            for nb, bpc in settings:
                k.glm_planes_tune(nb, bpc)
                us, out = graph_time(lambda: k.glm_bernoulli_planes_fwd_bwd(planes, y, w, b, 1.0, N, D))
                print(f"  {fname} ring={nb} wg/CU={bpc}      {us:8.1f} us   {N*(4*D+4)/us/1e6:6.3f} TB/s(alg)"
                      f"   rel err ll {err(out)[0]:.1e} gw {err(out)[1]:.1e}")
        k.glm_planes_tune(0, 0)
        del X, planes


if __name__ == "__main__":
    main()
