import cProfile, pstats, sys, time, torch
sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer.mcmc import MCMC, NUTS, GaussianPotential
dev = torch.device("cuda:0")
C, D = 1024, 100
_, Lam = examples.correlated_gaussian_precision(D, dtype=torch.float64)
Lam = Lam.float().to(dev)
def run(w, s):
    kernel = NUTS(potential_fn=GaussianPotential(Lam), max_tree_depth=10, target_accept_prob=0.8)
    mcmc = MCMC(kernel, num_samples=s, warmup_steps=w, num_chains=C, initial_params={"x": torch.zeros((C, D), device=dev)}, shard_chains=False)
    mcmc.run(); return kernel
pyro.set_rng_seed(1); run(20, 5); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
k = run(200, 200); n = k.num_leapfrog_steps; torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(38)
