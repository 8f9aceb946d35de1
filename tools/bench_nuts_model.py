"""NUTS on a MODEL (Bayesian logistic regression through the handlers, not a closed-form potential):
leapfrog steps/s of the SAMPLING phase (after warm-up: adapted step sizes, captured rounds), the run as
MCMC.run drives it.  Developer tool; bench.py's secondary_model_nuts reports the same measurement.

    python tools/bench_nuts_model.py [--n 100000] [--chains 256] [--samples 200] [--warmup 200] [--lockstep]
"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
import pyro_amd as pyro  # noqa: E402
from pyro_amd.infer.mcmc import NUTS  # noqa: E402
from tests import mcmc_cases as mc  # noqa: E402


def run(X, y, C, warmup, samples, max_tree_depth=6, lockstep=False, jit=False, seed=1, model=None, compact=True,
        rounds=0, generic=False):
    """-> dict(leapfrog_per_s, leapfrogs, seconds, rounds, mean_depth, step_size) of the sampling phase."""
    with pyro.validation_enabled(False):        # (MCMC.run's default: disable_validation=True)
        return _run(X, y, C, warmup, samples, max_tree_depth, lockstep, jit, seed, model, compact, rounds, generic)


def _run(X, y, C, warmup, samples, max_tree_depth, lockstep, jit, seed, model, compact, rounds, generic):
    pyro.set_rng_seed(seed)
    kernel = NUTS(model or mc.logreg_mcmc_model, max_tree_depth=max_tree_depth, jit_compile=jit)
    kernel.use_async_chains = not lockstep
    kernel.compact_chains = compact
    kernel.use_direct_potential = not generic
    if rounds:
        kernel.rounds_per_replay = rounds
    kernel.num_chains = C
    kernel.setup(warmup, X, y)
    dev = X.device
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kernel.bulk_ready:
        done = 0
        while done < warmup:
            done += kernel._transition_many(warmup - done)
    else:
        for _ in range(warmup):
            kernel._transition()
    kernel.end_warmup()
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t0
    n0 = kernel.num_leapfrog_steps
    r0 = getattr(kernel, "_span_replays", 0)
    buf = torch.empty((samples, C, kernel._layout.D), dtype=kernel._z.dtype, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kernel.bulk_ready:
        done = 0
        while done < samples:
            done += kernel._transition_many(samples - done, samples=buf[done:])
    else:
        for i in range(samples):
            kernel._transition()
            buf[i].copy_(kernel._position())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = kernel.num_leapfrog_steps - n0
    out = dict(leapfrog_per_s=n / dt, leapfrogs=n, seconds=dt, warmup_seconds=t_warm,
               replays=getattr(kernel, "_span_replays", 0) - r0,
               rounds_per_replay=kernel.rounds_per_replay,
               step_size=float(kernel.step_size.mean()), graphed=getattr(kernel, "_span_graph", None) is not None,
               compactions=getattr(kernel, "_span_compactions", 0),
               posterior_mean_w0=float(buf[:, :, 0].mean()))
    kernel.release_graphs()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--chains", type=int, default=256)
    ap.add_argument("--samples", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--lockstep", action="store_true")
    ap.add_argument("--no-compact", action="store_true")
    ap.add_argument("--rounds", type=int, default=0, help="tree rounds per graph replay (0: the kernel's default)")
    ap.add_argument("--generic", action="store_true", help="the potential through the handlers and autograd")
    ap.add_argument("--both", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, D, C = a.n, 32, a.chains
    g = torch.Generator().manual_seed(0)
    Xc = torch.randn((N, D), generator=g)
    yc = (torch.rand((N,), generator=g) < torch.sigmoid(Xc @ torch.randn(D, generator=g) * 0.3)).float()
    X, y = Xc.to(dev), yc.to(dev)
    for lock in ((True, False) if a.both else (a.lockstep,)):
        r = run(X, y, C, a.warmup, a.samples, a.depth, lockstep=lock, compact=not a.no_compact, rounds=a.rounds,
                generic=a.generic)
        print("N=%d C=%d %s: sampling %.3f s, %d leapfrogs, %.0f leapfrog/s (%.0f rounds/s if every round served all "
              "chains); warm-up %.2f s; %s" % (N, C, "lock-step" if lock else "async spans", r["seconds"],
                                               r["leapfrogs"], r["leapfrog_per_s"], r["leapfrog_per_s"] / C,
                                               r["warmup_seconds"], r))
