"""Developer probe: NUTS on BASELINE configs[1]'s model at N = 1e6 from the reference's default starting points
(init_to_uniform) -- finds the chain whose step size collapsed, prints its distance from the other chains, its
step / acceptance / mass, and the float32 potential against float64 at its position and at 1e-6 perturbations
(the position update eps * v is below the position's ulp there: the chain is frozen).  python tools/nuts_stuck_chain.py"""
import sys, torch
sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples, kernels
from pyro_amd.infer.mcmc import MCMC, NUTS
dev = torch.device("cuda", 0)
N = 1_000_000
X, y = examples.synthetic_logreg_data(N, 32, dev, seed=0)
pyro.set_rng_seed(11)
k = NUTS(examples.logreg_model, max_tree_depth=10)
m = MCMC(k, num_samples=20, warmup_steps=150, num_chains=256, shard_chains=False)
m.run(X, y)
step = k.step_size
z = k._z
med = z.median(0)[0]
dist = (z - med).abs().max(1)[0]
worst = int(step.argmin())
far = int(dist.argmax())
print("step: min %.3e (chain %d) median %.3e max %.3e" % (float(step.min()), worst, float(step.median()), float(step.max())))
print("farthest chain %d: max|z - median| %.3e; its step %.3e; its mean accept %.3f; pe %.6e (median pe %.6e)" % (
    far, float(dist[far]), float(step[far]), float(k._mean_accept_prob[far]), float(k._pe[far]), float(k._pe.median())))
print("chain of smallest step: max|z - median| %.3e accept %.3f pe %.6e" % (float(dist[worst]), float(k._mean_accept_prob[worst]), float(k._pe[worst])))
im = k._mm_eff.inverse_mass_matrix
print("inverse mass of the stuck chain: min %.3e max %.3e; median chain: min %.3e max %.3e" % (float(im[far].min()), float(im[far].max()), float(im.median(0)[0].min()), float(im.median(0)[0].max())))
# the potential's noise at the stuck chain's position: kernel (f32) against float64
zz = z[far:far + 1].repeat(8, 1)
zz[1:] += torch.randn_like(zz[1:]) * 1e-6
pe, gr = k._potential(zz)
Xd = X.double(); b_, w_ = zz[:, 0].double(), zz[:, 1:].double()
l = Xd @ w_.t() + b_
U = -(y.double()[:, None] * l - torch.nn.functional.softplus(l)).sum(0) + 0.5 * (zz.double() ** 2).sum(1)
print("potential at the stuck point (+ 1e-6 perturbations): f32 kernel - f64: ", ["%.3e" % float(v) for v in (pe.double() - U - (pe[0].double() - U[0]))], " f64 spread: ", ["%.3e" % float(v) for v in (U - U[0])])
print("abs offset f32 - f64 at the point: %.4e of U = %.6e" % (float(pe[0].double() - U[0] + 0.5 * 33 * 1.8378770664093453), float(U[0])))
