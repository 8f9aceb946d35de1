"""Worker of the world-size-2 gloo tests (tests/test_distributed_cpu.py).  Kernels are answered
by the numpy oracle (tests/oracle_backend.py): this exercises the DISTRIBUTED host logic --
flat-gradient all-reduce, parameter broadcast, chain sharding with global chain ids, all_gather
of samples -- on CPU exactly as it runs over RCCL on the GPUs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


class _MP:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def _setup(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from tests import oracle_backend as ob
    ob.install(_MP())
    torch.set_default_dtype(torch.float64)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)


def svi_worker(rank, world, port, out_path, eps_banks, X, y, steps):
    """Particle-sharded SVI: each rank P particles with its own eps bank, RcclOptimizer(Adam)."""
    _setup(rank, world, port)
    import pyro_amd as pyro
    from pyro_amd import rng
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal
    from tests import models

    Xt, yt = torch.tensor(X), torch.tensor(y)
    bank = eps_banks[rank]
    P = bank[0].shape[0]
    pyro.clear_param_store()
    pyro.set_rng_seed(100 + rank)
    guide = AutoNormal(models.logreg_model_fused, init_scale=0.1)
    guide._setup_prototype(Xt, yt)          # initialisation draws happen before the banked ones
    rng.normal = models.EpsReplay(bank, torch.device("cpu"))
    optim = pyro.optim.Adam({"lr": 0.05})
    if world > 1:
        optim = pyro.optim.RcclOptimizer(optim)
    svi = SVI(models.logreg_model_fused, guide, optim,
              Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1))
    losses = [svi.step(Xt, yt) for _ in range(steps)]
    params = {k: v.detach().clone() for k, v in pyro.get_param_store().items()}
    torch.save({"losses": losses, "params": params}, out_path % rank)
    if world > 1:
        dist.destroy_process_group()


def mcmc_worker(rank, world, port, out_path, Lam, z0, num_chains, n_samples):
    """Chain-sharded NUTS: global chain ids key the Philox streams, samples are all-gathered."""
    _setup(rank, world, port)
    import pyro_amd as pyro
    from pyro_amd.infer.mcmc import MCMC, NUTS, GaussianPotential

    pyro.set_rng_seed(5)
    kernel = NUTS(potential_fn=GaussianPotential(torch.tensor(Lam)), step_size=0.25,
                  adapt_step_size=False, adapt_mass_matrix=False, max_tree_depth=5)
    mcmc = MCMC(kernel, num_samples=n_samples, warmup_steps=0, num_chains=num_chains,
                initial_params={"x": torch.tensor(z0)})
    mcmc.run()
    x = mcmc.get_samples(group_by_chain=True)["x"]
    torch.save({"x": x, "local_chains": kernel.num_chains, "offset": kernel.chain_offset},
               out_path % rank)
    if world > 1:
        dist.destroy_process_group()


def data_sharded_worker(rank, world, port, out_path, bank, X, y, steps):
    """Data-sharded SVI (SURVEY 8e variant 2): every rank scores ITS rows of the plate, scaled to
    the full plate (plate(N, subsample=rows of this rank)), with the same particles everywhere;
    the mean of the per-rank gradients is the full-data gradient.  The optimizer is a generic
    per-parameter one (TorchAdam), so RcclOptimizer takes its pack -> all-reduce -> unpack route."""
    _setup(rank, world, port)
    import pyro_amd as pyro
    import pyro_amd.distributions as pdist
    from pyro_amd import rng
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal
    from tests import models

    Xt, yt = torch.tensor(X), torch.tensor(y)
    N, D = Xt.shape
    rows = torch.arange(N)[rank::world]

    def model(Xs, ys, idx):
        w = pyro.sample("w", pdist.Normal(torch.zeros(D), 1.0).to_event(1))
        b = pyro.sample("b", pdist.Normal(torch.zeros(()), 1.0))
        with pyro.plate("data", N, subsample=idx):
            logits = (w @ Xs.t()).squeeze(-2) if w.dim() > 1 else w @ Xs.t()
            pyro.sample("obs", pdist.Bernoulli(logits=logits + b), obs=ys)

    P = bank[0].shape[0]
    pyro.clear_param_store()
    pyro.set_rng_seed(7)
    guide = AutoNormal(model, init_scale=0.1)
    guide._setup_prototype(Xt[rows], yt[rows], rows)
    rng.normal = models.EpsReplay(bank, torch.device("cpu"))
    optim = pyro.optim.TorchAdam({"lr": 0.05})
    if world > 1:
        optim = pyro.optim.RcclOptimizer(optim)
    svi = SVI(model, guide, optim, Trace_ELBO(num_particles=P, vectorize_particles=True,
                                              max_plate_nesting=1))
    losses = [svi.step(Xt[rows], yt[rows], rows) for _ in range(steps)]
    params = {k: v.detach().clone() for k, v in pyro.get_param_store().items()}
    torch.save({"losses": losses, "params": params}, out_path % rank)
    if world > 1:
        dist.destroy_process_group()


def bench_worker(rank, world, port, out_path, steps, nuts=False):
    """bench.py itself, launched the way the driver launches it for N > 1 (RANK / WORLD_SIZE /
    MASTER_* in the environment), on host tensors over gloo with the kernels answered by the oracle:
    the rendezvous, the barrier-bracketed timed region, the max-over-ranks time and the ONE JSON line
    of rank 0."""
    import contextlib
    import io
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                       "PYRO_AMD_BENCH_DEVICE": "cpu"})
    if nuts:
        os.environ["PYRO_AMD_BENCH_CPU_NUTS"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from tests import oracle_backend as ob
    ob.install(_MP())
    import bench
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", str(steps), "--warmup", "1",
                "--plate", "512", "--features", "8", "--particles", "4",
                "--config5-rows", "600", "--config5-groups", "5"]     # (--config5-sharded: on by default at N > 1)
    if nuts:      # the driver's 8-GPU shares at toy sizes: 1024 chains / 8 -> 2 per rank, 512 particles / 8 -> 4
        sys.argv += ["--chains", "2", "--nuts-dim", "5", "--nuts-warmup", "6", "--nuts-samples", "4",
                     "--model-nuts-chains", "2", "--model-nuts-warmup", "5", "--model-nuts-samples", "3",
                     "--model-nuts-depth", "3"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    torch.save({"stdout": buf.getvalue()}, out_path % rank)

