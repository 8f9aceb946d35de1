"""Micro-benchmarks of the individual HIP kernels (developer tool, not the graded bench)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from pyro_amd import kernels as k


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def main():
    dev = torch.device("cuda:0")
    for (N, D, P) in [(1_000_000, 32, 64), (1_000_000, 32, 32), (1_000_000, 8, 64), (10_000_000, 32, 64)]:
        X = torch.randn((N, D), device=dev)
        y = (torch.rand((N,), device=dev) < 0.5).float()
        w = torch.randn((P, D), device=dev) * 0.2
        b = torch.randn((P,), device=dev)
        us = timeit(lambda: k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0))
        byts = N * (4 * D + 4)
        fl = 4.0 * P * N * D
        print(f"glm N={N} D={D} P={P}: {us:9.1f} us  {byts/us/1e6:7.3f} TB/s  {fl/us/1e6:7.2f} TFLOP/s(gemm)")
        # unfused level-A path: torch matmul + fused site kernel fwd + grad kernel + matmul
        def unfused():
            logits = torch.addmm(b[:, None], w, X.t())
            s = k.dist_log_prob_sum(1, y[None, :], logits, None, None, 1.0, P, N)
            g = torch.ones((P, 1), device=dev)
            _, dl, _ = k.dist_log_prob_grad(1, g, y[None, :], logits, None, None, 1.0, P, N, (False, True, False))
            return s, dl @ X, dl.sum(1)
        if N <= 1_000_000:
            us2 = timeit(unfused, n=5)
            print(f"    unfused (rocBLAS matmul + site kernels): {us2:9.1f} us")
        del X, y
    # elementwise site kernel bandwidth
    P, N = 64, 1_000_000
    logits = torch.randn((P, N), device=dev)
    y = (torch.rand((1, N), device=dev) < 0.5).float()
    us = timeit(lambda: k.dist_log_prob_sum(1, y, logits, None, None, 1.0, P, N))
    print(f"site log_prob_sum [64,1e6]: {us:.1f} us {P*N*4/us/1e6:.3f} TB/s")
    # NUTS
    C, Dn = 1024, 100
    A = torch.randn((Dn, Dn), dtype=torch.float64)
    Sigma = A @ A.T / Dn + 0.1 * torch.eye(Dn, dtype=torch.float64)
    Lam = torch.linalg.inv(Sigma)
    Lam = (0.5 * (Lam + Lam.T)).float().to(dev).contiguous()
    z = torch.zeros((C, Dn), device=dev)
    g = torch.zeros((C, Dn), device=dev)
    pe = torch.zeros((C,), device=dev)
    im = torch.ones((C, Dn), device=dev)
    st = torch.full((C,), 0.1, device=dev)
    tot = 0
    for t in range(20):
        out = k.nuts_gaussian_transition(z, pe, g, Lam, im, st, 10, True, 1, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(20, 120):
        out = k.nuts_gaussian_transition(z, pe, g, Lam, im, st, 10, True, 1, t)
        tot += out["n_leapfrog"].sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = int(tot)
    print(f"nuts C={C} D={Dn}: {tot} leapfrogs in {dt*1e3:.1f} ms -> {tot/dt/1e6:.2f} M leapfrog/s; mean tree size {tot/100/C:.1f}")


if __name__ == "__main__":
    main()
