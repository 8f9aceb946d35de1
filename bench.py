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

Prints ONE JSON line (rank 0) with `roofline` and `cpu_baseline`.  The timed region replays one
hipGraph per step; HIP events recorded inside a captured graph do not time the bracketed node on
ROCm 7.2, so the dominant kernel's duration (`roofline.kernel_ms`) is read from the DEVICE's wall
clock: the kernel itself records its earliest workgroup entry and latest workgroup exit
(pa_glm_planes_stamps), in 20 replays of the same graph right after the timed region.  Beside it:
`kernel_ms_eager` (HIP events around the kernel on its launch stream in eager steps) and
`kernel_ms_rocprof` (the rocprofv3 kernel trace of this same command, committed under profiles/).
`validated` repeats the headline with pyro.enable_validation(True) (SURVEY 8d asks for both).
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
BINDING_F16 = ("serialised phases of two waves per SIMD: 229 VALU instructions per wave and 32x32 (row, "
               "particle) tile (sigmoid / softplus sums, the two-level f16 split of g) + 13 MFMAs + the "
               "tile loop's barrier / operand reads; timing ablations (profiles/r04_glm16_ablation.txt): "
               "fixed 5 us + loop skeleton 11.5 + element-wise 15 + split 7 + MFMA 8 + LDS-DMA beside the "
               "compute 7 add up to the kernel's duration; the image stream alone runs at 4.6 TB/s")
BINDING_BF16 = ("VALU issue next to the MFMA pipe: 332 VALU instructions per wave and 32x32 (row, "
                "particle) tile (sigmoid / softplus sums + the exact 3-way bf16 split of g) against 25 "
                "MFMAs; PMC per launch: VALU issue 47 us of SIMD time, matrix pipe busy 26 us, kernel "
                "65 us (profiles/r02_pmc_summary.json, DESIGN.md section 3)")
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_TBS = 8.0             # MI355X_MICROARCH.md: HBM3E spec peak
PEAK_F32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 vector peak (FMA = 2 flop)
CPU_BASELINE_THREADS = 32      # the CPU legs run on this many host threads, every round


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--plate", type=int, default=1_000_000)
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--particles", type=int, default=64, help="particles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nuts", action="store_true", help="skip the secondary NUTS measurement")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the one-GPU measurements of BASELINE configs 4 (LDA) and 5 (hierarchical)")
    ap.add_argument("--full", action="store_true",
                    help="also time the variants of configs[1] (guides, D sweep, image formats, ...), the LDA "
                         "mini-batch sizes, eight schools and the HMM example: they go into bench_full.json, "
                         "never into the printed line")
    ap.add_argument("--prearm", action="store_true",
                    help="headline with SVI(prearm=True) (opt-in: the caller promises to enqueue nothing between "
                         "two steps); by default it is measured beside the headline as `with_prearm`")
    ap.add_argument("--no-graph", action="store_true",
                    help="eager SVI.step (Python handlers + one launch per kernel) instead of the "
                         "captured hipGraph step")
    ap.add_argument("--chains", type=int, default=1024, help="NUTS chains per GPU")
    ap.add_argument("--nuts-dim", type=int, default=100)
    ap.add_argument("--nuts-warmup", type=int, default=200)
    ap.add_argument("--nuts-samples", type=int, default=200)
    ap.add_argument("--no-model-nuts", action="store_true",
                    help="skip NUTS on the logistic-regression MODEL (secondary_model_nuts)")
    ap.add_argument("--model-nuts-chains", type=int, default=256, help="chains per GPU")
    ap.add_argument("--model-nuts-warmup", type=int, default=100)
    ap.add_argument("--model-nuts-samples", type=int, default=100)
    ap.add_argument("--model-nuts-depth", type=int, default=10)
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--config5-sharded", action=argparse.BooleanOptionalAction, default=None,
                    help="also time BASELINE configs[4] with the PLATE sharded over the ranks (SURVEY "
                         "8e variant 2): rows / world per GPU, the same particles everywhere, the "
                         "likelihood scaled to the full plate, one flat RCCL gradient all-reduce "
                         "(default: on when launched with more than one rank)")
    ap.add_argument("--config5-rows", type=int, default=10_000_000)
    ap.add_argument("--config5-groups", type=int, default=1000)
    return ap.parse_args()


ROUND = 6       # profiles/r05_*: counter files of another round measured other kernels and are refused


def latest_profile(suffix):
    """profiles/r<ROUND>_<suffix> (counters cannot be collected from inside the run: the rocprofv3 passes
    of tools/prof.sh write them), or None -- a file of an earlier round is NOT taken in its place."""
    path = os.path.join(ROOT, "profiles", "r%02d_%s" % (ROUND, suffix))
    return path if os.path.exists(path) else None


def measured_hbm(dev):
    """What this box's HBM delivers to plain streaming kernels (report the measured peak
    beside the nominal 8 TB/s): a 1-GiB device-to-device copy (bytes read + bytes written per second)
    and a read-only reduction over the same buffer, torch's own kernels, best of 5."""
    import torch
    n = 1 << 28                                            # 1 GiB of f32
    a = torch.empty((n,), dtype=torch.float32, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def best(fn):
        fn()
        t = []
        for _ in range(5):
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            t.append(s.elapsed_time(e) * 1e-3)
        return min(t)
    t_copy = best(lambda: b.copy_(a))
    t_read = best(lambda: a.sum())
    return {"copy_GBps": 2 * 4 * n / t_copy / 1e9, "read_GBps": 4 * n / t_read / 1e9,
            "how": "1 GiB f32: torch d2d copy (read + written bytes) and torch.sum (read bytes), best of 5"}


def cpu_baseline(N, D, P, budget_s):
    """Time the torch-CPU port of the reference's SVI step (oracle/ref_port_torch.py) on the
    host cores, on the SAME workload shape, bounded to ~budget_s seconds of CPU work.  The thread
    count is swept (an untuned count deflates the baseline: 128 threads are slower than 16 on
    this shape) and the best one is reported with its count."""
    from oracle.ref_port_torch import LogRegAutoNormalPort

    g = torch.Generator().manual_seed(0)
    X = torch.randn((N, D), generator=g)
    w_true = torch.randn((D,), generator=g)
    y = (torch.rand((N,), generator=g) < torch.sigmoid(X @ w_true)).float()
    port = LogRegAutoNormalPort(X, y, P)
    ncpu = os.cpu_count() or 1
    # a STATED thread count (the same every round, so that ratios compare across rounds): 32, the
    # neighbourhood the sweeps of rounds 2-4 found best on this shape (16 and 32 within a few per cent,
    # 128 slower); one probe of 16 beside it is reported, not used
    cores = min(CPU_BASELINE_THREADS, ncpu)
    default_threads = torch.get_num_threads()
    probe = {}
    try:
        for t in sorted({min(16, ncpu), cores}):
            torch.set_num_threads(t)
            port.step()
            t0 = time.perf_counter()
            port.step()
            probe[t] = time.perf_counter() - t0
        torch.set_num_threads(cores)
        left = max(2.0, budget_s - 2.0 * sum(probe.values()))
        n = max(3, min(40, int(left / max(probe[cores], 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n):
            port.step()
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(default_threads)
    return {"value": n / dt, "unit": "ELBO-grad steps/s", "cores": cores, "kind": "port",
            "probe_steps_per_s": {str(t): 1.0 / v for t, v in probe.items()},
            "host_cpus": ncpu,
            "sample": "%d full SVI steps (N=%d, D=%d, P=%d, fp32) of oracle/ref_port_torch.py "
                      "(the reference's torch-CPU operators without its handler overhead) on %d "
                      "threads (fixed)" % (n, N, D, P, cores)}


def _nuts_chain_on_host(D, budget_s, chain):
    """One reference-style chain on one host core: the recursive tree of pyro/infer/mcmc/nuts.py
    (oracle/nuts.py) with the potential gradient taken by torch autograd on CPU tensors on every
    leapfrog step, as pyro/ops/integrator.py:68-94 does.  Returns (leapfrogs, transitions, s)."""
    import numpy as np
    from oracle import nuts as o_nuts
    from pyro_amd import examples

    torch.set_num_threads(1)
    _, Lam = examples.correlated_gaussian_precision(D, dtype=torch.float64)

    def pot_and_grad(z):
        zt = torch.tensor(z, requires_grad=True)
        pe = 0.5 * zt @ Lam @ zt
        (g,) = torch.autograd.grad(pe, zt)
        return float(pe.detach()), g.numpy()

    z = np.zeros(D)
    pe, g = pot_and_grad(z)
    n, t0, t = 0, time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        out = o_nuts.nuts_transition(z, pe, g, pot_and_grad, np.ones(D), 0.15,
                                     o_nuts.KeyedDraws(1, chain, t, np.float64), 10, True)
        z, pe, g = out["z"], out["pe"], out["grad"]
        n += out["n_leapfrog"]
        t += 1
    return n, t, time.perf_counter() - t0


def _nuts_chain_worker(D, budget_s, chain, q):
    q.put(_nuts_chain_on_host(D, budget_s, chain))


def nuts_cpu_baseline(D, budget_s, procs=7):
    """The reference's two execution models on the host: ONE chain on one
    core, and 7 chains in 7 processes (pyro/infer/mcmc/api.py:239-351, num_chains=7 with
    mp_context="spawn"), leapfrogs summed over the chains."""
    import multiprocessing as mp
    n, t, dt = _nuts_chain_on_host(D, budget_s, 0)
    out = {"value": n / dt, "unit": "leapfrog steps/s", "cores": 1, "kind": "port",
           "sample": "%d NUTS transitions (%d leapfrogs) of ONE chain, D=%d, f64, step 0.15, unit "
                     "mass: oracle/nuts.py recursion + torch-CPU autograd gradient per leapfrog "
                     "(the reference's per-chain execution model)" % (t, n, D)}
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_nuts_chain_worker, args=(D, budget_s, c + 1, q)) for c in range(procs)]
        for p_ in ps:
            p_.start()
        res = [q.get(timeout=budget_s * 4 + 240) for _ in ps]
        for p_ in ps:
            p_.join(timeout=60)
        out["multiprocess"] = {"value": sum(r[0] / r[2] for r in res), "cores": procs,
                               "unit": "leapfrog steps/s summed over %d chains in %d processes" % (procs, procs),
                               "leapfrogs": sum(r[0] for r in res)}
    except Exception as e:  # noqa: BLE001  (a box that cannot spawn still reports the one-chain figure)
        out["multiprocess"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


class _NoKernelTimer:                      # (host run of the plumbing: nothing to bracket)
    pairs = []

    def arm(self):
        pass

    def times_ms(self):
        return []

    def mean_ms(self):
        return float("nan")


def bench_nuts(dev, rank, world, args):
    """BASELINE config 3: NUTS on a 100-dim correlated Gaussian, 1024 vectorised chains per GPU,
    200 warm-up (step size + diagonal mass adapted per chain) + 200 sampling transitions.
    Leapfrog steps/s = all leapfrogs of all chains (warm-up included) / wall time."""
    import torch.distributed as dist

    import pyro_amd as pyro
    from pyro_amd import _lib, examples, kernels
    from pyro_amd.infer.mcmc import MCMC, NUTS, GaussianPotential

    C, D = args.chains, args.nuts_dim
    _, Lam = examples.correlated_gaussian_precision(D, dtype=torch.float64)
    Lam = Lam.float().to(dev)
    pyro.set_rng_seed(1 + rank)

    def run(warmup, samples):
        kernel = NUTS(potential_fn=GaussianPotential(Lam), max_tree_depth=10,
                      target_accept_prob=0.8)
        mcmc = MCMC(kernel, num_samples=samples, warmup_steps=warmup, num_chains=C,
                    initial_params={"x": torch.zeros((C, D), device=dev)}, shard_chains=False)
        timer = kernels.KernelTimer(_lib.KERNEL_NUTS) if dev.type == "cuda" else _NoKernelTimer()
        kernel._launch_hook = timer.arm      # brackets every persistent launch
        mcmc.run()
        return kernel, mcmc, timer

    run(min(20, args.nuts_warmup), min(5, args.nuts_samples))    # warm the allocator / code objects
    if world > 1:
        dist.barrier()
    _sync(dev)
    t0 = time.perf_counter()
    kernel, mcmc, timer = run(args.nuts_warmup, args.nuts_samples)
    nleap = kernel.num_leapfrog_steps
    _sync(dev)
    elapsed = time.perf_counter() - t0
    tot = torch.tensor([float(nleap), elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        nleap_all, elapsed = float(tot[0]), float(mx[1])
    else:
        nleap_all = float(nleap)
    if rank != 0:
        return None
    x = mcmc.get_samples(group_by_chain=True)["x"]
    diag = mcmc.diagnostics()
    kern_ms = timer.times_ms()
    kern_total_ms = sum(kern_ms) if kern_ms else float("nan")
    n_t = args.nuts_warmup + args.nuts_samples
    flops = nleap * (2.0 * D * D + 6.0 * D)
    stream_bytes = nleap * 6.0 * D * 4
    out = {"metric": "leapfrog steps/sec (NUTS)", "value": nleap_all / elapsed,
           "unit": "leapfrog steps/s summed over chains", "n_gpus": world,
           "wall_s": elapsed, "leapfrogs": nleap_all, "scaling": "weak", "dtype": "f32",
           "config": {"workload": "BASELINE configs[2]: NUTS, %d-dim correlated Gaussian, %d "
                                  "vectorised chains per GPU, %d warm-up + %d samples, per-chain "
                                  "step-size and diagonal-mass adaptation, max_tree_depth=10"
                                  % (D, C, args.nuts_warmup, args.nuts_samples)},
           "mean_tree_leaves": nleap / (n_t * C),
           "posterior_check": {"max_r_hat": float(diag["x"]["r_hat"].max()),
                               "min_n_eff": float(diag["x"]["n_eff"].min()),
                               "mean_accept_prob": float(kernel._mean_accept_prob.mean())},
           "launches": len(kern_ms),
           "roofline": {"bound": "on-chip: chain state and Lambda's columns in VGPRs, f32 VALU "
                                 "FMA issue; the streaming-model HBM figure is for reference",
                        "kernel": "nuts_gaussian_kernel", "kernel_ms_total": kern_total_ms,
                        "achieved": flops / (kern_total_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                        "peak": PEAK_F32_VALU_TFLOPS,
                        "frac": flops / (kern_total_ms * 1e-3) / 1e12 / PEAK_F32_VALU_TFLOPS,
                        "streaming_model_TBps": stream_bytes / (kern_total_ms * 1e-3) / 1e12,
                        "traffic": None}}
    # counter-measured HBM bytes of the NUTS kernels over one such run (tools/nuts_traffic.sh, committed
    # under profiles/: counters cannot be collected from inside the run)
    npath = latest_profile("nuts_traffic.json")
    if npath is not None:
        try:
            nj = json.load(open(npath))
            out["roofline"]["traffic"] = nj.get("hbm_bytes_total")
            out["roofline"]["traffic_per_leapfrog_bytes"] = nj.get("hbm_bytes_per_leapfrog")
            out["roofline"]["traffic_source"] = "profiles/%s: HBM bytes of every NUTS kernel launch of one " \
                "MCMC.run of this workload (%s leapfrogs); %s" % (os.path.basename(npath), nj.get("leapfrogs"),
                                                                nj.get("how", ""))
        except Exception:
            pass
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = nuts_cpu_baseline(D, min(args.cpu_budget_s, 8.0))
    return out


def _model_nuts_chain_on_host(N, D, budget_s, threads):
    """One reference-style chain of NUTS on the logistic-regression MODEL on the host: the recursive
    tree of pyro/infer/mcmc/nuts.py (oracle/nuts.py), the potential (-log joint of examples.logreg_model:
    pyro/infer/mcmc/util.py:264-286) and its gradient by torch-CPU autograd on every leapfrog step, as
    pyro/ops/integrator.py:68-94 does.  -> (leapfrogs, transitions, seconds)."""
    import numpy as np
    from oracle import nuts as o_nuts

    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    X = torch.randn((N, D), generator=g)
    w_true = torch.randn((D,), generator=g)
    y = (torch.rand((N,), generator=g) < torch.sigmoid(X @ w_true)).float()
    bce = torch.nn.functional.binary_cross_entropy_with_logits

    def pot_and_grad(z):
        zt = torch.tensor(z, dtype=torch.float32, requires_grad=True)
        b, w = zt[0], zt[1:]                       # flat layout: sites sorted by name
        pe = bce(X @ w + b, y, reduction="sum") + 0.5 * (zt * zt).sum()
        (gr,) = torch.autograd.grad(pe, zt)
        return float(pe.detach()), gr.double().numpy()

    z = np.zeros(D + 1)
    pe, gr = pot_and_grad(z)
    n, t0, t = 0, time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        out = o_nuts.nuts_transition(z, pe, gr, pot_and_grad, np.ones(D + 1), 0.02,
                                     o_nuts.KeyedDraws(1, 0, t, np.float64), 10, True)
        z, pe, gr = out["z"], out["pe"], out["grad"]
        n += out["n_leapfrog"]
        t += 1
    return n, t, time.perf_counter() - t0


def bench_model_nuts(dev, rank, world, args):
    """NUTS on a MODEL potential (north_star: "the velocity-Verlet update with potential_fn grad"):
    MCMC(NUTS(examples.logreg_model)) -- BASELINE configs[1]'s model text, through the handlers -- with
    per-chain step-size and diagonal-mass adaptation, C vectorised chains per GPU, at N = 1e5 and 1e6.
    Leapfrog steps/s = leapfrogs of all chains / wall time, for the whole MCMC.run (setup, capture and
    warm-up included) and for its sampling phase alone."""
    import torch.distributed as dist

    import pyro_amd as pyro
    from pyro_amd import _lib, examples, kernels
    from pyro_amd.infer.mcmc import MCMC, NUTS

    C, D = args.model_nuts_chains, args.features
    W, S = args.model_nuts_warmup, args.model_nuts_samples

    def run(X, y, warmup, samples, chains, init=None, model=None):
        pyro.set_rng_seed(11 + rank)
        kw = {} if init is None else {"init_strategy": init}
        kernel = NUTS(model or examples.logreg_model, max_tree_depth=args.model_nuts_depth, **kw)
        mcmc = MCMC(kernel, num_samples=samples, warmup_steps=warmup, num_chains=chains,
                    shard_chains=False)
        marks = {}
        end_warmup = kernel.end_warmup

        def marked():
            _sync(dev)
            marks.update(t=time.perf_counter(), n=kernel.num_leapfrog_steps,
                         replays=getattr(kernel, "_span_replays", 0), slots=getattr(kernel, "_span_rounds", 0))
            end_warmup()
        kernel.end_warmup = marked
        if world > 1:
            dist.barrier()
        _sync(dev)
        t0 = time.perf_counter()
        mcmc.run(X, y)
        n = kernel.num_leapfrog_steps
        _sync(dev)
        t1 = time.perf_counter()
        return kernel, mcmc, dict(wall=t1 - t0, n=n, t_sample=t1 - marks["t"], n_sample=n - marks["n"],
                                  replays_sample=getattr(kernel, "_span_replays", 0) - marks["replays"],
                                  slots_sample=getattr(kernel, "_span_rounds", 0) - marks["slots"],
                                  compactions=getattr(kernel, "_span_compactions", 0))

    out = {}
    # (N, chains, warm-up, samples, init strategy).  The 1e6-row posterior is ~1000x tighter than the prior's scale.
    # From the reference's default starting points (init_to_uniform: U(-2, 2) in every coordinate, potential ~5e6
    # with gradients of ~1e5 per coordinate) most chains reach it within ~100 transitions, but with 256 chains one
    # or two do not: their step size collapses (1e-7) while the gradient is still enormous, and in float32 the
    # position update eps * v falls below the position's ulp -- such a chain is frozen for good and every later
    # transition of it is a 1023-leapfrog tree the other 255 chains wait for (tools/nuts_stuck_chain.py and nuts_stuck_chain_history.py print it;
    # round 5's R-hat of 8.4 and this round's first collection, R-hat 77, were that).  The N = 1e6 run therefore
    # starts at the prior's median -- NUTS(init_strategy=init_to_median), the reference's own strategy
    # (pyro/infer/autoguide/initialization.py:67-92, accepted by pyro/infer/mcmc/nuts.py:125) -- from where 150
    # warm-up transitions give R-hat 1.04 with either wave geometry; the 1e5-row runs keep the default.
    from pyro_amd.infer.autoguide.initialization import init_to_median
    # the last run: a hierarchical prior (tau -> w: a site's scale IS another latent site, config 5's structure),
    # which the direct potential holds since round 6 -- 3 launches per round as for the flat model (VERDICT r05
    # item 7; config 5's own G = 1000 groups are 32 065 coordinates per chain: beyond the tree kernel's per-chain
    # form; a vector tau[D] over ONE group is a funnel -- 128 leapfrogs per transition, R-hat 1.06 after 1500)
    plan = [(100_000, C, 5 * W, 10 * S, None, None), (100_000, 4 * C, 2 * W, 4 * S, None, None),
            (1_000_000, C, 3 * W, 2 * S, init_to_median, None),
            (100_000, C, 3 * W, 5 * S, None, examples.hier_prior_logreg_model)]
    if dev.type != "cuda":                 # the plumbing test of tests/test_distributed_cpu.py
        plan = [(args.plate, C, W, S, None, None)]
    X = y = None
    for N, C, W, S, init, model in plan:
        if X is None or X.shape[0] != N:
            X, y = examples.synthetic_logreg_data(N, D, dev, seed=0)
        run(X, y, min(12, W), min(4, S), C, init, model)   # warm the allocator / code objects / the plane image of X
        kernel, mcmc, r = run(X, y, W, S, C, init, model)
        tot = torch.tensor([float(r["n"]), r["wall"], float(r["n_sample"]), r["t_sample"]], device=dev,
                           dtype=torch.float64)
        if world > 1:
            mx = tot.clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            n_all, wall, ns_all, ts = float(tot[0]), float(mx[1]), float(tot[2]), float(mx[3])
        else:
            n_all, wall, ns_all, ts = (float(v) for v in tot)
        if rank != 0:
            continue
        # the dominant kernel at P = C particles, HIP events on its launch stream, eager evaluations
        # of the same potential at the chains' final positions
        timer = kernels.KernelTimer(_lib.KERNEL_GLM) if dev.type == "cuda" else _NoKernelTimer()
        with pyro.validation_enabled(False):
            for _ in range(8 if dev.type == "cuda" else 1):
                timer.arm()
                kernel._potential(kernel._z)
        _sync(dev)
        kms = sorted(timer.times_ms())
        kern_ms = kms[len(kms) // 2] if kms else float("nan")
        rounds = r["replays_sample"] * kernel.rounds_per_replay
        alg = N * (4 * D + 4)
        diag = mcmc.diagnostics()
        key = "N%d_C%d" % (N, C) + ("" if model is None else "_hier_prior")
        direct = getattr(kernel, "_direct", None)
        out[key] = {
            "direct_potential": direct is not None, "latent_sites": None if direct is None else int(direct.n),
            "value": n_all / wall, "sampling_phase": ns_all / ts, "unit": "leapfrog steps/s summed over chains",
            "wall_s": wall, "leapfrogs": n_all, "sampling_s": ts, "sampling_leapfrogs": ns_all,
            "mean_tree_leaves": r["n"] / ((W + S) * C),
            "rounds_sampling": rounds, "us_per_round": ts / max(rounds, 1) * 1e6,
            # leapfrogs done per cursor row the potential was evaluated at (compacted rounds evaluate fewer rows)
            "round_occupancy": r["n_sample"] / max(r["slots_sample"], 1),
            "compactions": r["compactions"],
            "graphed": bool(rounds),
            "posterior_check": {"max_r_hat": float(max(d["r_hat"].max() for k, d in diag.items()
                                                       if isinstance(d, dict) and "r_hat" in d)),
                                "mean_accept_prob": float(kernel._mean_accept_prob.mean()),
                                "mean_step_size": float(kernel.step_size.mean())},
            "roofline": {"bound": "hbm", "kernel": "glm_planes_f16_kernel (P = %d chains: %d pass%s of %d over "
                                                   "the image)" % (C, -(-C // (256 if C > 128 else 128 if C > 64 else 64)),
                                                                   "" if C <= 256 else "es", 256 if C > 128 else 128 if C > 64 else 64),
                         "kernel_ms": kern_ms, "kernel_ms_source": "HIP events around the kernel on its launch "
                         "stream, median of %d eager evaluations of the potential" % len(kms),
                         "algorithmic_bytes_per_round": alg, "achieved": alg / (kern_ms * 1e-3) / 1e9,
                         "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                         "frac": alg / (kern_ms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                         "frac_of_round": alg / (ts / max(rounds, 1)) / 1e12 / PEAK_HBM_TBS,
                         "note": "N(4D+4) bytes per round are shared by all chains; the kernel is bound by "
                                 "VALU issue (14.6 instructions per (row, chain) element), so its time "
                                 "grows with C at fixed bytes",
                         "traffic": None}}
        tp = latest_profile("nuts_model_traffic.json")
        if tp is not None:
            try:
                tj = json.load(open(tp))
                ent = tj.get(key)
                if ent:
                    out[key]["roofline"]["traffic"] = ent.get("hbm_bytes_per_launch")
                    out[key]["roofline"]["traffic_source"] = "profiles/%s" % os.path.basename(tp)
            except Exception:
                pass
        out[key]["transitions"] = "%d warm-up + %d samples" % (W, S)
        out[key]["init_strategy"] = "init_to_uniform (default)" if init is None else init.__name__
        # a run whose chains have not mixed is a throughput measurement of unconverged chains: said so, and
        # its roofline block is not a claim about a working sampler
        out[key]["converged"] = bool(out[key]["posterior_check"]["max_r_hat"] < 1.05)
        out[key]["roofline"]["claimed"] = out[key]["converged"]
        kernel.cleanup()
    C, W, S = args.model_nuts_chains, args.model_nuts_warmup, args.model_nuts_samples
    if rank != 0:
        return None
    res = {"metric": "leapfrog steps/sec (NUTS on a model potential)", "n_gpus": world, "scaling": "weak",
           "dtype": "f32",
           "config": {"workload": "MCMC(NUTS(model)), model = BASELINE configs[1]'s logistic regression (SURVEY "
                                  "8(d) text, w @ X.t() recognised lazily), D=%d, %d / %d vectorised chains per GPU, "
                                  "per-chain step-size + diagonal-mass adaptation, "
                                  "max_tree_depth=%d; asynchronous chains (a chain that finishes a tree starts "
                                  "its next one in the same launch), %d rounds per hipGraph replay"
                                  % (D, C, 4 * C, args.model_nuts_depth, 16)},
           "runs": out}
    if world == 1 and not args.no_cpu_baseline:
        n, t, dt = _model_nuts_chain_on_host(100_000, D, min(args.cpu_budget_s, 8.0),
                                             min(CPU_BASELINE_THREADS, os.cpu_count() or 1))
        res["cpu_baseline"] = {"value": n / dt, "unit": "leapfrog steps/s", "kind": "port",
                               "cores": min(CPU_BASELINE_THREADS, os.cpu_count() or 1),
                               "sample": "%d NUTS transitions (%d leapfrogs) of ONE chain at N=1e5, D=%d: "
                                         "oracle/nuts.py recursion + torch-CPU autograd of the model's log "
                                         "joint per leapfrog, step 0.02, unit mass" % (t, n, D)}
    return res


def _r(v, sig=6):
    """A float at `sig` significant digits (the printed line is a record, not a dump)."""
    if isinstance(v, float) and v == v and abs(v) != float("inf"):
        return float("%.*g" % (sig, v))
    return v


def _pick(d, *keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def _file_only(src):
    """profiles/<file> of a "profiles/<file> (how ...)" source string."""
    return src.split(" ")[0].rstrip(":") if isinstance(src, str) else src


def reference_cpu_record():
    """The UNMODIFIED reference's SVI.step on configs[1], timed in the build container where /root/reference
    exists (tools/time_reference_cpu.py) and committed as a fixture: profiles/r<ROUND>_reference_cpu.json."""
    path = latest_profile("reference_cpu.json")
    if path is None:
        return None
    try:
        j = json.load(open(path))
        return {"value": _r(j["validation_off"]["steps_per_s"]),
                "validated": _r(j["validation_on"]["steps_per_s"]), "cores": j["cores"],
                "kind": "reference", "where": "build container", "source": "profiles/" + os.path.basename(path)}
    except Exception:  # noqa: BLE001
        return None


def compact(out):
    """The ONE line the driver parses: numbers and short names only, < 4 KB whatever was measured.  The full
    record (every block time, every note, every variant) goes to bench_full.json."""
    line = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "rccl_ranks", "replay")
    line["config"] = out["config"]
    rf = out["roofline"]
    line["roofline"] = _pick(rf, "bound", "kernel", "kernel_ms", "kernel_ms_eager", "kernel_ms_rocprof", "achieved",
                             "peak", "unit", "frac", "frac_rocprof", "traffic", "algorithmic_bytes_per_launch")
    line["roofline"]["traffic_source"] = _file_only(rf.get("traffic_source"))
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind")
        line["cpu_baseline"]["sample"] = "%s SVI steps of the torch-CPU port, same N/D/P" % cb.get("sample", "?").split(" ")[0]
        if cb.get("reference"):
            line["cpu_baseline"]["reference"] = cb["reference"]
    for k in ("with_prearm", "without_prearm", "validated"):
        if out.get(k):
            line[k] = _pick(out[k], "value", "ms_per_step")
    oc = out.get("other_configs") or {}
    others = {}
    keep = ("config4_lda", "gmm_enumerated", "config5_hierarchical_logreg", "config5_plate_sharded", "config2_loss_and_grads_only",
            "config2_bf16x3_exact_split", "error")
    for k, v in oc.items():
        if k not in keep:                      # (the variants of --full: bench_full.json only)
            continue
        if not isinstance(v, dict):
            others[k] = str(v)[:120]
            continue
        e = _pick(v, "steps_per_s", "ms_per_step", "ranks", "rows_total")
        r = v.get("roofline")
        if isinstance(r, dict):
            e.update(_pick(r, "kernel_ms", "frac"))
            e["kernel"] = str(r.get("kernel", ""))[:40]
        others[k] = e
    if others:
        line["other_configs"] = others
    sec = out.get("secondary")
    if sec:
        e = _pick(sec, "metric", "value", "unit", "n_gpus", "leapfrogs", "wall_s", "dtype")
        e["roofline"] = _pick(sec.get("roofline", {}), "kernel", "kernel_ms_total", "achieved", "peak", "unit", "frac")
        e["max_r_hat"] = _r(sec.get("posterior_check", {}).get("max_r_hat"))
        if sec.get("cpu_baseline"):
            e["cpu_baseline"] = _pick(sec["cpu_baseline"], "value", "cores", "kind")
        line["secondary"] = e
    mn = out.get("secondary_model_nuts")
    if mn:
        e = _pick(mn, "metric", "n_gpus", "dtype", "error")
        runs = {}
        for k, v in (mn.get("runs") or {}).items():
            rr = _pick(v, "value", "sampling_phase", "leapfrogs", "us_per_round", "round_occupancy", "converged",
                       "direct_potential")
            rr["max_r_hat"] = _r(v.get("posterior_check", {}).get("max_r_hat"))
            rr["roofline"] = _pick(v.get("roofline", {}), "kernel_ms", "frac", "traffic", "claimed")
            runs[k] = rr
        e["runs"] = runs
        if mn.get("cpu_baseline"):
            e["cpu_baseline"] = _pick(mn["cpu_baseline"], "value", "cores", "kind")
        line["secondary_model_nuts"] = e
    line["full_record"] = "bench_full.json"
    return line


def emit(out):
    """Write the full record to bench_full.json (repo root, and gpurun_out/ when that exists so that it comes
    back from the GPU box) and print the compact line -- the only thing this script writes to stdout."""
    full = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as fh:
                    fh.write(full + "\n")
            except OSError:
                pass
    line = json.dumps(compact(out), separators=(",", ":"))
    if len(line) >= 4096:                      # never again a line the driver cannot keep whole
        c = compact(out)
        for k in ("other_configs", "validated", "without_prearm", "with_prearm"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < 4096:
                break
    print(line, flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # PYRO_AMD_BENCH_DEVICE=cpu: the multi-process plumbing of this script (rendezvous, barrier,
    # max-over-ranks timing, the one JSON line) on host tensors over gloo -- used by
    # tests/test_distributed_cpu.py with the kernels answered by the test oracle; never a measurement
    on_gpu = os.environ.get("PYRO_AMD_BENCH_DEVICE", "cuda") != "cpu"
    import torch.distributed as dist
    if on_gpu:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        args.no_others = args.no_cpu_baseline = args.no_graph = True
        # PYRO_AMD_BENCH_CPU_NUTS=1 keeps the two NUTS blocks in the host run (chain sharding, the sums and
        # maxima over ranks, rank > 0 returning nothing): small sizes from the command line
        args.no_nuts = args.no_model_nuts = not os.environ.get("PYRO_AMD_BENCH_CPU_NUTS")

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
    use_graph = not args.no_graph
    # the dominant kernel's own clock stamps (a launch argument: created before the capture)
    clock = kernels.GlmDeviceClock(dev) if on_gpu else None
    # The headline is the reference's constructor and nothing else: SVI(model, guide, optim, loss).  With
    # device tensors as arguments such an SVI captures its step into a hipGraph by itself (after 3 eager
    # steps, with a one-time warning naming what a captured step freezes); every step() launches its own
    # replay.  SVI(prearm=True) -- opt-in, the caller promises to enqueue nothing between two steps that
    # reads what a step writes -- is measured beside it (`with_prearm`); --prearm makes it the headline,
    # --no-graph is the reference's eager step.
    kw = {}
    if not use_graph:
        kw["hip_graph"] = False
    elif args.prearm:
        kw["prearm"] = True
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        svi = SVI(examples.logreg_model, guide, optim,
                  Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1), **kw)
    prearm = bool(svi.prearm and svi.hip_graph and world == 1)

    class _NoTimer:                      # (host run: nothing to bracket)
        pairs = []

        def arm(self):
            pass

        def mean_ms(self):
            return float("nan")

    timer = kernels.KernelTimer(_lib.KERNEL_GLM) if on_gpu else _NoTimer()
    # untimed warm-up; with hip_graph the step is captured here (after 3 eager steps).  The event bracket is NOT
    # armed then: armed for the capturing step it became two event-record nodes of the graph, which do not time
    # the node on ROCm 7.2 (below) -- the headline graph is the two kernel nodes and nothing else
    for _ in range(max(args.warmup, 4 if use_graph else 0)):
        if not use_graph:
            timer.arm()
        svi.step(X, y)
    graphed = use_graph and svi.hip_graph and len(svi._graphs) == 1
    kern_ms_list = []
    # Event records captured into a hipGraph do not time the bracketed node on ROCm 7.2 (they read
    # ~5 us for a 120 us kernel), so with the graphed step the dominant kernel is bracketed in
    # eager steps of the SAME workload run right after the timed region; the rocprofv3 kernel
    # trace of the graphed run (profiles/) gives the same per-launch duration.
    graphed_events = False
    timer.pairs = []

    def sync():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    # The timed region is EXACTLY args.steps steps between barrier + synchronize on both sides.  At the
    # driver's --steps 20 that is ~1.7 ms of timing, so the block is repeated (a count fixed by the
    # arguments alone: every rank runs the same number of barriers) and the MEDIAN block is the
    # headline; every block's time is in the line (`blocks_ms_per_step`).
    # (The first ~250 replays after a capture run 8-10 % slower than the rest -- DESIGN.md section 3, "the
    #  step gate"; tools/short_block_ramp.py -- so the blocks cover ~8000 steps: the median then sits in the
    #  state a real SVI run, thousands of steps long, spends its time in.)
    nblocks = max(1, min(100, -(-8000 // max(args.steps, 1))))
    block_s = []
    for _ in range(nblocks):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if not graphed:
                timer.arm()
            svi.step(X, y)             # returns the loss as a float: one host read per step
            if graphed_events:
                kern_ms_list.append(timer.read_last())
        if prearm:
            svi.pause()                # (inside the timed region: part of what a block of steps costs)
        sync()
        block_s.append(time.perf_counter() - t0)
    # which kernel the headline ran (the secondary measurements below switch image formats)
    headline_planes = bool(on_gpu and kernels.glm_planes_of(X) is not None)
    headline_f16 = headline_planes and kernels.glm_planes_format() == kernels.GLM_PLANES_F16X2
    prearmed = bool(prearm and graphed and next(iter(svi._graphs.values())).gate is not None)
    unarmed = None
    if prearmed:
        # the same captured step with every replay launched by its own step() call
        svi.disarm()
        nu = min(args.steps, 300)
        tus = []
        for _ in range(max(1, min(25, nblocks))):
            sync()
            tu = time.perf_counter()
            for _ in range(nu):
                svi.step(X, y)
            sync()
            tus.append(time.perf_counter() - tu)
        tu = sorted(tus)[len(tus) // 2]
        unarmed = {"value": nu / tu, "ms_per_step": tu / nu * 1e3, "steps": nu, "blocks": len(tus),
                   "note": "the same capture after SVI.disarm(): the host launches each replay when "
                           "step() is called (what SVI(..., prearm=False) runs)"}
    clock_ms = []
    if graphed and not graphed_events:
        # the kernel's duration inside the graph, from its own stamps on the device clock
        for _ in range(20):
            clock.arm()
            svi.step(X, y)
            torch.cuda.synchronize()
            clock_ms.append(clock.read_ms())
        clock_ms = [v for v in clock_ms if v == v]
        # ... and bracketed with HIP events in eager steps of the same workload
        for _ in range(10):
            timer.arm()
            svi._eager_step(X, y)
        torch.cuda.synchronize()
    armed = None
    if world == 1 and graphed and not prearmed and on_gpu:
        # the opt-in variant beside the headline: SVI(prearm=True) on the same guide / optimizer
        svi_p = SVI(examples.logreg_model, guide, optim,
                    Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1),
                    hip_graph=True, graph_warmup=2, prearm=True)
        for _ in range(8):
            svi_p.step(X, y)
        na = min(args.steps, 300)
        tas = []
        for _ in range(max(1, min(25, nblocks))):
            sync()
            ta = time.perf_counter()
            for _ in range(na):
                svi_p.step(X, y)
            svi_p.pause()
            sync()
            tas.append(time.perf_counter() - ta)
        ta = sorted(tas)[len(tas) // 2]
        ent = next(iter(svi_p._graphs.values()), None)
        armed = {"value": na / ta, "ms_per_step": ta / na * 1e3, "steps": na, "blocks": len(tas),
                 "gated": bool(ent is not None and ent.gate is not None),
                 "note": "SVI(..., prearm=True): the replay of step k+1 enqueued behind a gate while step k "
                         "executes (opt-in: the caller enqueues nothing between steps that reads what a step "
                         "writes)"}
        svi_p.release()
    validated = None
    if world == 1 and graphed:
        # the same measurement with validation switched on (it runs in the eager steps before the
        # capture; the captured step is the same graph)
        pyro.enable_validation(True)
        try:
            svi_v = SVI(examples.logreg_model, guide, optim,
                        Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1),
                        hip_graph=True, graph_warmup=2)
            for _ in range(6):
                svi_v.step(X, y)
            nv = min(args.steps, 200)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(nv):
                svi_v.step(X, y)
            torch.cuda.synchronize()
            tv = time.perf_counter() - tv
            validated = {"value": nv / tv, "ms_per_step": tv / nv * 1e3, "steps": nv,
                         "graphed": bool(svi_v.hip_graph and len(svi_v._graphs) == 1),
                         "note": "pyro.enable_validation(True): argument / shape / support checks "
                                 "run in the eager steps that precede the capture"}
        finally:
            pyro.enable_validation(False)
    rccl_ranks = 1
    if world > 1:
        t = torch.tensor(block_s, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # max over ranks, block by block
        block_s = [float(v) for v in t.cpu()]
        # how many ranks the communicator actually holds: a sum of ones through the same backend the
        # gradient all-reduce uses (not the WORLD_SIZE environment variable)
        ones = torch.ones((1,), device=dev, dtype=torch.float32)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(round(float(ones.item())))
    elapsed = sorted(block_s)[len(block_s) // 2]           # the median block

    sharded5 = None
    if args.config5_sharded is None:
        args.config5_sharded = world > 1                   # (both SURVEY 8e variants on a multi-GPU run)
    if args.config5_sharded:
        # every rank holds rows / world rows (its own synthetic shard), draws the SAME particles (one
        # seed), scores its rows scaled by `world`, and the flat gradient is averaged over the ranks
        n_loc = args.config5_rows // world
        G5 = args.config5_groups
        Xs, ys, gs = examples.synthetic_hier_logreg_data_unsorted(n_loc, D, G5, dev, seed=100 + rank)
        pyro.clear_param_store()
        pyro.set_rng_seed(4321)
        model5 = lambda X_, y_, g_: examples.hier_logreg_model_reference(X_, y_, g_, G5, plate_scale=float(world))  # noqa: E731
        opt5 = pyro.optim.Adam({"lr": 0.01})
        if world > 1:
            opt5 = pyro.optim.RcclOptimizer(opt5)
        svi5 = SVI(model5, AutoNormal(model5, init_scale=0.1), opt5,
                   Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1),
                   hip_graph=use_graph, graph_warmup=2)
        for _ in range(6):
            svi5.step(Xs, ys, gs)
        sync()
        t5 = time.perf_counter()
        n5 = max(3, min(args.steps, 20))
        for _ in range(n5):
            svi5.step(Xs, ys, gs)
        sync()
        t5 = time.perf_counter() - t5
        if world > 1:
            tt5 = torch.tensor([t5], device=dev, dtype=torch.float64)
            dist.all_reduce(tt5, op=dist.ReduceOp.MAX)
            t5 = float(tt5.item())
        sharded5 = {"steps_per_s": n5 / t5, "ms_per_step": t5 / n5 * 1e3, "rows_per_rank": n_loc,
                    "rows_total": n_loc * world, "particles": P, "ranks": world, "scaling": "strong",
                    "workload": "BASELINE configs[4], plate sharded over the ranks (SURVEY 8e variant 2): "
                                "hierarchical logistic regression in the reference's formulation (unsorted "
                                "int64 group ids, (w[..., g, :] * X).sum(-1) + b), %d rows in all, %d groups, "
                                "the same %d particles on every rank, likelihood scaled by the world size, "
                                "flat gradient all-reduce (mean)" % (n_loc * world, args.config5_groups, P)}
        pyro.clear_param_store()
    nuts = None if args.no_nuts else bench_nuts(dev, rank, world, args)
    model_nuts = None
    if not args.no_model_nuts:
        try:
            model_nuts = bench_model_nuts(dev, rank, world, args)
        except Exception as e:  # noqa: BLE001  (a secondary measurement must not kill the headline)
            if world > 1:
                raise
            model_nuts = {"error": "%s: %s" % (type(e).__name__, e)}
    others = None
    if world == 1 and not args.no_others:
        # BASELINE configs[3] and configs[4] (one GPU's share) -- reported, not the headline value
        from tools import bench_configs
        others = {}
        try:
            r5 = bench_configs.config5(dev, steps=20)
            r5["workload"] = ("BASELINE configs[4], one GPU's share, SURVEY 8(d)'s model text verbatim: "
                              "hierarchical logistic regression, plate=1e7, D=32, g = randint(0, 1000, (N,)) "
                              "UNSORTED, logits = (w[..., g, :] * X).sum(-1) + b, 64 of the 512 particles, "
                              "AutoNormal, Adam, graphed SVI.step; the gather is recognised lazily and runs "
                              "the grouped plane-image GLM kernel (%s)" % r5.get("roofline", {}).get("kernel", "?"))
            others["config5_hierarchical_logreg"] = r5
            r4 = bench_configs.config4(dev, steps=10)
            r4["workload"] = ("BASELINE configs[3]: examples/lda.py, TraceEnum_ELBO, 1e5 documents (all "
                              "in the plate), 8 topics, 1024 words, 64 words per document, amortised "
                              "guide; word_topics enumerated and summed out by the indexed LDA kernels "
                              "(no atomics), the guide's first layer on the bag-of-words image, its inner "
                              "layers on the tall-batch kernels; graphed SVI.step")
            others["config4_lda"] = r4
            rg = bench_configs.config_gmm(dev)
            rg["workload"] = ("a plated Gaussian mixture under TraceEnum_ELBO, N=1e6 data, K=16 components, the "
                              "assignment enumerated: likelihood against every component + logsumexp + plate sum "
                              "+ all gradients in one pass over the data (csrc/mixture.hip); graphed SVI.step")
            others["gmm_enumerated"] = rg
            if args.full:
                rg0 = bench_configs.config_gmm(dev, leaf=False)
                rg0["workload"] = "the same mixture with its [K, N] likelihood materialised (round 3's elimination)"
                others["gmm_enumerated_materialised"] = rg0
                for bs in (32, 4096):      # the mini-batch variants SURVEY 8(d) lists
                    rb = bench_configs.config4(dev, steps=30, batch_size=bs)
                    rb["workload"] = ("examples/lda.py with batch_size=%d of 1e5 documents per step (a fresh "
                                      "sub-sampled word matrix every step: LDS-atomic factor kernel), "
                                      "graphed SVI.step" % bs)
                    others["config4_lda_batch%d" % bs] = rb
                r1 = bench_configs.config1(dev)
                r1["workload"] = ("BASELINE configs[0]: eight schools, Trace_ELBO, 1 particle, Adam; graphed "
                                  "SVI.step (the reference's own CPU-runnable case: ~58 steps/s there)")
                others["config1_eight_schools"] = r1
                rh = bench_configs.config_hmm(dev, steps=10, graph=True)
                rh["workload"] = ("examples/hmm.py model_1 at its JSB-chorales size (229 sequences x 129 "
                                  "steps, 16 hidden states, 88 tones), TraceEnum_ELBO under pyro.markov, "
                                  "chain summed out by pa_logchain_fwd_bwd, graphed SVI.step")
                others["hmm_example"] = rh
                rv = bench_configs.config_hmm_vectorised(dev)
                rv["workload"] = ("the same HMM likelihood with time vectorised in one DiscreteHMM site "
                                  "(examples/hmm.py model_7's construction): one pa_logchain_fwd_bwd launch "
                                  "for all sequences, graphed SVI.step")
                others["hmm_example_vectorised"] = rv
                r2m = bench_configs.config2_variant(dev, "mvn")
                r2m["workload"] = ("BASELINE configs[1] with AutoMultivariateNormal (the second guide "
                                   "SURVEY 8d names), 64 particles, graphed SVI.step")
                others["config2_automultivariatenormal"] = r2m
                r2p = bench_configs.config2_variant(dev, "normal", P=1)
                r2p["workload"] = ("BASELINE configs[1] at the reference's default num_particles=1: "
                                   "few-particle vector-ALU GLM kernel (HBM-bound), graphed SVI.step")
                others["config2_one_particle"] = r2p
                r2e = bench_configs.config2_variant(dev, "normal", model=examples.logreg_model_explicit)
                r2e["workload"] = ("BASELINE configs[1] with the logits spelled as dist.linear_logits(X, w, b) "
                                   "and hoisted prior constants instead of the reference's model text "
                                   "(same kernels, two fill launches fewer)")
                others["config2_explicit_linear_logits"] = r2e
            # "loss_and_grads only" beside the full step, the D sweep, and the
            # exact three-plane bf16 image beside the default two-plane f16 one on every run
            r2g = bench_configs.config2_variant(dev, "normal", no_update=True)
            r2g["workload"] = ("BASELINE configs[1], loss_and_grads only: the captured step without the "
                               "optimizer update (guide draw, fused ELBO gradient, guide backward, gradient "
                               "zeroing, host read of the loss)")
            others["config2_loss_and_grads_only"] = r2g
            r2x = bench_configs.config2_variant(dev, "normal", planes_format=kernels.GLM_PLANES_BF16X3)
            r2x["workload"] = ("BASELINE configs[1] on the EXACT three-plane bf16 image (every f32 operand "
                               "split exactly, 6 piece products): the headline's arithmetic without the "
                               "2^-22 representation error of the two-plane f16 image")
            others["config2_bf16x3_exact_split"] = r2x
            if args.full:
                for Dv in (8, 64, 128):
                    rd = bench_configs.config2_variant(dev, "normal", D=Dv, steps=30)
                    rd["workload"] = ("BASELINE configs[1] at D=%d (SURVEY 8d sweep): %s" % (
                        Dv, "plane-image kernel" if Dv <= 32 else
                        "plane-image kernel with %d feature tiles of 32 columns (csrc/glm_planes16d.h; until "
                        "round 4 D > 32 split X on the fly in two passes)" % (2 if Dv <= 64 else 4)))
                    others["config2_D%d" % Dv] = rd
                r2u = bench_configs.config2_variant(dev, "normal", lazy_matmul=False, steps=20)
                r2u["workload"] = ("BASELINE configs[1], the reference's model text with the lazy recognition "
                                   "of w @ X.t() switched OFF: [P, N] logits materialised by rocBLAS, "
                                   "log-prob / gradient by the fused site kernels, rocBLAS for dw")
                others["config2_materialised_logits"] = r2u
        except Exception as e:  # noqa: BLE001  (secondary measurements must not kill the headline)
            others["error"] = "%s: %s" % (type(e).__name__, e)
        pyro.clear_param_store()
    if rank == 0:
        kern_ms_eager = sum(kern_ms_list) / len(kern_ms_list) if kern_ms_list else timer.mean_ms()
        kern_ms = sum(clock_ms) / len(clock_ms) if clock_ms else kern_ms_eager
        gemm_flops = 4.0 * P * N * D                      # two [P,D]x[D,N]-shaped contractions
        alg_bytes = N * (4 * D + 4)                        # X (f32) + y (f32), read once
        achieved_tflops = gemm_flops / (kern_ms * 1e-3) / 1e12
        # counter-measured HBM bytes per launch and the in-graph duration of the dominant kernel
        # come from the rocprofv3 passes over this same command (tools/prof.sh), committed under
        # profiles/: they cannot be collected from inside the run
        traffic = rocprof_ms = None
        traffic_src = None
        tpath = latest_profile("traffic.json")
        if tpath is None:
            traffic_src = "none: no profiles/r%02d_traffic.json committed this round (earlier rounds' files are refused)" % ROUND
        if tpath is not None:
            try:
                tj = json.load(open(tpath))
                # ... and only for the kernel family that is running now
                running = ("glm_planes_f16_kernel" if headline_f16 else "glm_planes_kernel") if headline_planes \
                    else "glm_bernoulli_bf16_kernel"
                if running in str(tj.get("kernel", "")):
                    traffic, rocprof_ms = tj.get("hbm_bytes_per_launch"), tj.get("kernel_ms_in_graph")
                    traffic_src = "profiles/%s (%s)" % (os.path.basename(tpath), tj.get("how", ""))
                else:
                    traffic_src = "refused: profiles/%s measured %s, this run launches %s" % (
                        os.path.basename(tpath), str(tj.get("kernel"))[:60], running)
            except Exception:
                traffic = None
        planes, f16 = headline_planes, headline_f16      # (noted right after the timed region)
        n_prod = 3 if f16 else 6                       # piece products per element product
        out = {
            "metric": "ELBO-grad steps/sec (SVI)", "value": world * args.steps / elapsed,
            "unit": "ELBO-grad steps/s (64 particles x 1e6-row plate per step)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "timing": {"blocks": len(block_s), "steps_per_block": args.steps, "statistic": "median block "
                       "(each block = exactly `steps` steps between barrier + synchronize, max over ranks)",
                       "blocks_ms_per_step": [round(b / args.steps * 1e3, 5) for b in block_s]},
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: Bayesian logistic regression, plate=%d, D=%d, Trace_ELBO "
                                   "num_particles=%d per GPU (vectorised), AutoNormal, Adam; full SVI.step of "
                                   "SVI(model, guide, optim, loss), %s" % (N, D, P, (
                                       ("one hipGraph replay per step" + (", prearm=True" if prearmed else ""))
                                       if graphed else "eager launches")),
                       "parallelism": "particles sharded x%d, flat RCCL grad all-reduce" % world},
            # the plate scan is priced against HBM (algorithmic bytes = X and y once)
            "roofline": {"bound": "hbm", "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9,
                         "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                         "frac": alg_bytes / (kern_ms * 1e-3) / 1e12 / PEAK_HBM_TBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ("glm_planes_f16_kernel" if f16 else "glm_planes_kernel") if planes
                         else "glm_bernoulli_bf16_kernel",
                         "kernel_ms": kern_ms,
                         "kernel_ms_source": ("the kernel's own stamps on the device wall clock "
                                              "(earliest workgroup entry -> latest workgroup exit), "
                                              "mean of %d replays of the captured step right after "
                                              "the timed region" % len(clock_ms)) if clock_ms else
                                             "HIP events around the kernel on its launch stream",
                         "kernel_ms_eager": kern_ms_eager,
                         "kernel_ms_rocprof": rocprof_ms,
                         "frac_rocprof": (alg_bytes / (rocprof_ms * 1e-3) / 1e12 / PEAK_HBM_TBS)
                         if rocprof_ms else None,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "arithmetic": ("f32-class: every f32 operand as two power-of-two-scaled f16 "
                                        "pieces (2^-22 relative), the 3 piece products of order >= "
                                        "2^-11 on the f16 matrix cores, f32 accumulation; error "
                                        "against float64 equal to an f32 evaluation's "
                                        "(tests/test_kernels_gpu.py::test_glm_planes_f16_is_f32_class; "
                                        "PYRO_AMD_GLM_PLANES=bf16x3 selects the exact 3-piece image)")
                         if f16 else
                                       "f32-equivalent: every f32 operand split exactly into 3 bf16 "
                                       "pieces, the 6 piece products of order >= 2^-16 on the bf16 "
                                       "matrix cores, f32 accumulation",
                         # the other two resources the kernel uses, for the same duration:
                         "mfma_piece_TFLOPs": n_prod * gemm_flops / (kern_ms * 1e-3) / 1e12,
                         "frac_16bit_mfma": n_prod * gemm_flops / (kern_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                         "f32_equivalent_TFLOPs": achieved_tflops,
                         "binding_resource": BINDING_F16 if f16 else BINDING_BF16},
            "rccl_ranks": rccl_ranks,      # measured: all_reduce(SUM) of ones over the process group
        }
        if world == 1 and on_gpu:
            try:
                out["roofline"]["measured_hbm"] = measured_hbm(dev)
            except Exception as e:  # noqa: BLE001
                out["roofline"]["measured_hbm"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, D, P, args.cpu_budget_s)
            ref = reference_cpu_record()
            if ref is not None:
                out["cpu_baseline"]["reference"] = ref
        if validated is not None:
            out["validated"] = validated
        if unarmed is not None:
            out["without_prearm"] = unarmed
        if armed is not None:
            out["with_prearm"] = armed
        out["step_anatomy"] = {"graph_nodes": "glm_planes (its prologue makes the guide draw), step gate (when "
                                              "pre-armed: in front of the tail, so a replay enqueued ahead runs "
                                              "its forward pass before the host asks), chain_tail (GLM finalize "
                                              "+ ELBO assembly + guide backward + Adam + loss hand-over)",
                               "chain": getattr(svi, "chain_stats", None),
                               "chain_fused": getattr(svi, "chain_fused", None)}
        ent = next(iter(svi._graphs.values()), None) if graphed else None
        # how a step is enqueued: its kernels one by one (csrc/replay.hip: a graph that is a chain of <= 4 kernel
        # nodes) or one hipGraphLaunch
        out["replay"] = ("eager" if ent is None else "hipGraphLaunch" if ent.direct is None
                         else "%d kernel launches" % ent.direct.n_nodes)
        if sharded5 is not None:
            others = dict(others or {})
            others["config5_plate_sharded"] = sharded5
        if others:
            out["other_configs"] = others
        # (the two NUTS blocks LAST: the driver keeps the tail of this line)
        if nuts is not None:
            out["secondary"] = nuts
        if model_nuts is not None:
            out["secondary_model_nuts"] = model_nuts
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
