"""Developer probe: how much warm-up NUTS on BASELINE configs[1]'s model needs at N = 1e6 before R-hat < 1.05
(bench.py's secondary_model_nuts sizes its N = 1e6 run from this), and what a round costs once converged.

    python tools/nuts_model_converge.py [--n 1000000] [--chains 256] [--warmup 150] [--samples 50]
"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
import pyro_amd as pyro  # noqa: E402
from pyro_amd import examples  # noqa: E402
from pyro_amd.infer.mcmc import MCMC, NUTS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--chains", type=int, default=256)
ap.add_argument("--warmup", type=int, default=150)
ap.add_argument("--samples", type=int, default=50)
ap.add_argument("--depth", type=int, default=10)
ap.add_argument("--no-compact", action="store_true")
ap.add_argument("--eager", action="store_true", help="no captured rounds")
ap.add_argument("--generic", action="store_true", help="the handlers + autograd potential")
ap.add_argument("--init", default="uniform", help="uniform (reference default) | median | mean | sample")
ap.add_argument("--tune", type=int, default=0, help="pa_glm_planes_tune ring-depth code (measurement knob)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
if a.tune:
    from pyro_amd import kernels as _k
    _k.glm_planes_tune(a.tune, 0)
X, y = examples.synthetic_logreg_data(a.n, 32, dev, seed=0)
pyro.set_rng_seed(11)
from pyro_amd.infer.autoguide import initialization as _init  # noqa: E402
k = NUTS(examples.logreg_model, max_tree_depth=a.depth,
         init_strategy=getattr(_init, "init_to_" + a.init))
if a.no_compact:
    k.compact_chains = False
if a.generic:
    k.use_direct_potential = False
if a.eager:
    from pyro_amd.infer.mcmc import nuts as _nuts
    _nuts.CAPTURE_ROUNDS = False
m = MCMC(k, num_samples=a.samples, warmup_steps=a.warmup, num_chains=a.chains, shard_chains=False)
marks = {}
end_warmup = k.end_warmup


def marked():
    torch.cuda.synchronize()
    marks.update(t=time.perf_counter(), n=k.num_leapfrog_steps, slots=k._span_rounds, rep=k._span_replays)
    end_warmup()


k.end_warmup = marked
torch.cuda.synchronize()
t0 = time.perf_counter()
m.run(X, y)
torch.cuda.synchronize()
t1 = time.perf_counter()
d = m.diagnostics()
rh = max(float(v["r_hat"].max()) for v in d.values() if isinstance(v, dict) and "r_hat" in v)
n = k.num_leapfrog_steps
print("N=%d C=%d warmup=%d samples=%d: wall %.2f s (warm-up %.2f s), leapfrogs %d (sampling %d), "
      "sampling %.0f leapfrog/s, occupancy %.3f, max r_hat %.3f, step %.4g, accept %.3f, mean leaves/transition "
      "sampling %.1f" % (a.n, a.chains, a.warmup, a.samples, t1 - t0, marks["t"] - t0, n, n - marks["n"],
                         (n - marks["n"]) / (t1 - marks["t"]),
                         (n - marks["n"]) / max(k._span_rounds - marks["slots"], 1), rh, float(k.step_size.mean()),
                         float(k._mean_accept_prob.mean()), (n - marks["n"]) / (a.samples * a.chains)))
