#!/usr/bin/env python
"""Benchmark of the SVI hot path on MI355X: ELBO-gradient steps/s of BASELINE config 2
(Bayesian logistic regression, plate N=1e6, D=32, Trace_ELBO with 64 vectorised particles per
GPU, AutoNormal guide, Adam), one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full ``SVI.step`` (guide sample, model replay, fused ELBO gradient kernel,
optimizer update) over the whole 1e6-row plate with 64 particles, inputs resident in HBM.
Multi-GPU: particles are sharded (64 per GPU, weak scaling; data replicated), gradients reduced
with ONE flat RCCL all-reduce per step; value = world_size * steps / max-over-ranks time, i.e.
64-particle ELBO-gradient evaluations per second over the whole job.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed inside the timed
region) and `cpu_baseline` (the torch-CPU port of the reference step on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_HBM_TBS = 8.0             # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--plate", type=int, default=1_000_000)
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--particles", type=int, default=64, help="particles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    return ap.parse_args()


def cpu_baseline(N, D, P, budget_s):
    """Time the torch-CPU port of the reference's SVI step (oracle/ref_port_torch.py) on the
    host cores, on the SAME workload shape, bounded to ~budget_s seconds of CPU work."""
    from oracle.ref_port_torch import LogRegAutoNormalPort

    g = torch.Generator().manual_seed(0)
    X = torch.randn((N, D), generator=g)
    w_true = torch.randn((D,), generator=g)
    y = (torch.rand((N,), generator=g) < torch.sigmoid(X @ w_true)).float()
    port = LogRegAutoNormalPort(X, y, P)
    port.step()
    t0 = time.perf_counter()
    port.step()
    one = time.perf_counter() - t0
    n = max(2, min(30, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        port.step()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "ELBO-grad steps/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d full SVI steps (N=%d, D=%d, P=%d, fp32) of oracle/ref_port_torch.py "
                      "(the reference's torch-CPU operators without its handler overhead)" % (n, N, D, P)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=dev)

    import pyro_amd as pyro
    from pyro_amd import _lib, examples, kernels
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    N, D, P = args.plate, args.features, args.particles
    X, y = examples.synthetic_logreg_data(N, D, dev, seed=0)
    pyro.clear_param_store()
    pyro.set_rng_seed(1234 + rank)          # different particles on every rank
    pyro.enable_validation(False)           # reported both ways in DESIGN.md; hot loop unvalidated
    guide = AutoNormal(examples.logreg_model, init_scale=0.1)
    optim = pyro.optim.Adam({"lr": 0.01})
    if world > 1:
        optim = pyro.optim.RcclOptimizer(optim)
    svi = SVI(examples.logreg_model, guide, optim,
              Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1))

    for _ in range(args.warmup):
        svi.step(X, y)
    timer = kernels.KernelTimer(_lib.KERNEL_GLM)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timer.arm()
        svi.step(X, y)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        kern_ms = timer.mean_ms()
        gemm_flops = 4.0 * P * N * D                      # two [P,D]x[D,N]-shaped contractions
        alg_bytes = N * (4 * D + 4)                        # X (f32) + y (f32), read once
        achieved_tflops = gemm_flops / (kern_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("glm_bernoulli_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "ELBO-grad steps/sec (SVI)", "value": world * args.steps / elapsed,
            "unit": "ELBO-grad steps/s (64 particles x 1e6-row plate per step)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: Bayesian logistic regression, plate=%d, "
                                   "D=%d, Trace_ELBO num_particles=%d per GPU (vectorised), AutoNormal, "
                                   "Adam; full SVI.step" % (N, D, P),
                       "parallelism": "particles sharded x%d, flat RCCL grad all-reduce" % world},
            "roofline": {"bound": "mfma", "achieved": achieved_tflops, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic, "kernel": "glm_bernoulli_kernel",
                         "kernel_ms": kern_ms, "flops_per_launch": gemm_flops,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "hbm_achieved_TBps": alg_bytes / (kern_ms * 1e-3) / 1e12,
                         "hbm_frac_of_%.1fTBps" % PEAK_HBM_TBS: alg_bytes / (kern_ms * 1e-3) / 1e12 / PEAK_HBM_TBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, D, P, args.cpu_budget_s)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
