"""Warm timings of the small-site kernels in isolation (developer tool)."""
import sys
import torch
sys.path.insert(0, ".")
from pyro_amd import kernels as k
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
P, D = 64, 32
f = lambda *s: torch.randn(s, device=dev)
zw, zb = f(P, D), f(P, 1)
loc, sc = f(1, D), torch.rand((1, D), device=dev) + 0.5
zero, one = torch.zeros((1, 1), device=dev), torch.ones((1, 1), device=dev)
ll = f(1, P)
ents = [dict(dist=0, rows=P, cols=D, value=zw, p0=zero, p1=one, mask=None, coef=1.0, need=(True, False, False)),
        dict(dist=0, rows=P, cols=1, value=zb, p0=zero, p1=one, mask=None, coef=1.0, need=(True, False, False)),
        dict(dist=100, rows=1, cols=P, value=ll, p0=None, p1=None, mask=None, coef=1.0, need=(False, False, False)),
        dict(dist=0, rows=P, cols=D, value=zw, p0=loc, p1=sc, mask=None, coef=-1.0, need=(True, True, True)),
        dict(dist=0, rows=P, cols=1, value=zb, p0=zero.clone(), p1=one.clone(), mask=None, coef=-1.0, need=(True, True, True))]
g = torch.ones((), device=dev)
print("multi_sum   %.1f us" % timeit(lambda: k.multi_log_prob_sum(ents, -1 / 64, torch.float32, dev), n=200, warm=20))
print("multi_grad  %.1f us" % timeit(lambda: k.multi_log_prob_grad(g, ents, -1 / 64, torch.float32, dev), n=200, warm=20))
locs, rhos = [f(D), f(1)], [f(D), f(1)]
print("mf_sample   %.1f us" % timeit(lambda: k.meanfield_normal_sample(locs, rhos, P, 1, [0, 512]), n=200, warm=20))
zs, scs, los, eps = k.meanfield_normal_sample(locs, rhos, P, 1, [0, 512])
dz = [f(P, D), f(P, 1)]
ds = [f(D), f(1)]
print("mf_bwd      %.1f us" % timeit(lambda: k.meanfield_normal_sample_bwd(rhos, eps, dz, ds, [None, None], P), n=200, warm=20))
x = f(66)
print("torch add (66 elems) %.1f us" % timeit(lambda: x + x, n=200, warm=20))
print("empty launch-ish: counter_add %.1f us" % timeit(lambda: k.counter_add(torch.zeros(1, dtype=torch.int64, device=dev), 1), n=200, warm=20))

# steady-state cost inside a hipGraph (GPU kept busy, clocks up): 50 launches per replay
def graphed(fn, reps=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    return timeit(gr.replay, n=20, warm=3) / reps


print("--- per launch inside a graph of 50 back-to-back launches")
print("torch add           %.2f us" % graphed(lambda: x + x))
print("mf_sample           %.2f us" % graphed(lambda: k.meanfield_normal_sample(locs, rhos, P, 1, [0, 512])))
print("mf_bwd              %.2f us" % graphed(lambda: k.meanfield_normal_sample_bwd(rhos, eps, dz, ds, [None, None], P)))
print("multi_sum           %.2f us" % graphed(lambda: k.multi_log_prob_sum(ents, -1 / 64, torch.float32, dev)))
print("multi_grad          %.2f us" % graphed(lambda: k.multi_log_prob_grad(g, ents, -1 / 64, torch.float32, dev)))
ents1 = ents[:1]
print("multi_grad 1 entry  %.2f us" % graphed(lambda: k.multi_log_prob_grad(g, ents1, -1 / 64, torch.float32, dev)))
print("multi_sum 1 entry   %.2f us" % graphed(lambda: k.multi_log_prob_sum(ents1, -1 / 64, torch.float32, dev)))
