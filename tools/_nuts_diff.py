import sys, torch
sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples, kernels
from pyro_amd.infer.mcmc import MCMC, NUTS
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, y = examples.synthetic_logreg_data(N, 32, dev, seed=0)
def run(tune, W, compact=True):
    kernels.glm_planes_tune(tune, 0)
    pyro.set_rng_seed(11)
    k = NUTS(examples.logreg_model, max_tree_depth=10)
    k.compact_chains = compact
    m = MCMC(k, num_samples=2, warmup_steps=W, num_chains=256, shard_chains=False)
    m.run(X, y)
    kernels.glm_planes_tune(0, 0)
    return k._z.clone(), k.step_size.clone(), k._span_compactions, k.num_leapfrog_steps
for W in (3, 6, 12, 25, 50):
    a = run(5, W); b = run(0, W); c = run(0, W, compact=False)
    d = (a[0] - b[0]).abs().max(1)[0]
    d2 = (a[0] - c[0]).abs().max(1)[0]
    print("W=%d: priv vs default: max|dz| %.3e (chains differing > 1e-3: %s) leapfrogs %d vs %d, compactions %d vs %d; priv vs default-no-compact: max|dz| %.3e"
          % (W, float(d.max()), (d > 1e-3).nonzero().flatten().tolist()[:12], a[3], b[3], a[2], b[2], float(d2.max())))
    print("   step sizes min: priv %.3e default %.3e nocompact %.3e" % (float(a[1].min()), float(b[1].min()), float(c[1].min())))
