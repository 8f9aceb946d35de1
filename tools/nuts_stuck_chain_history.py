"""Developer probe: the warm-up history (distance from the final median, step size, potential) of the chain that
freezes in tools/nuts_stuck_chain.py, transition by transition (hook_fn: lock-step transitions)."""
import sys, torch
sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples, kernels
from pyro_amd.infer.mcmc import MCMC, NUTS
dev = torch.device("cuda", 0)
N = 1_000_000
X, y = examples.synthetic_logreg_data(N, 32, dev, seed=0)
ref = None
for W in (150,):
    pyro.set_rng_seed(11)
    k = NUTS(examples.logreg_model, max_tree_depth=10)
    hist = []
    def hook(kernel, samples, stage, i):
        z = kernel._z
        hist.append((stage, i, z[113].clone(), float(kernel.step_size[113]), float(kernel._pe[113]), float(kernel.step_size.median())))
    m = MCMC(k, num_samples=2, warmup_steps=W, num_chains=256, shard_chains=False, hook_fn=hook)
    m.run(X, y)
    med = k._z.median(0)[0]
    for stage, i, z, st, pe, stm in hist[:60]:
        print("%s %3d chain 113: max|z-med| %.3e  step %.3e (median step %.3e) pe %.4e" % (stage, i, float((z - med).abs().max()), st, stm, pe))
