"""Where the wall time of a graphed SVI.step goes (developer tool)."""
import sys, time
import torch
sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal

dev = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(1_000_000, 32, dev, seed=0)
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
guide = AutoNormal(examples.logreg_model, init_scale=0.1)
svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.01}),
          Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), hip_graph=True, graph_warmup=2)
for _ in range(6):
    svi.step(X, y)
entry = list(svi._graphs.values())[0]


def t(fn, n=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("svi.step                         %.1f us" % t(lambda: svi.step(X, y)))
print("graph.replay + loss.item()       %.1f us" % t(lambda: (entry.graph.replay(), entry.loss.item())))
print("graph.replay + stream.sync       %.1f us" % t(lambda: (entry.graph.replay(), torch.cuda.current_stream().synchronize())))
print("graph.replay back-to-back (no sync) %.1f us" % t(lambda: entry.graph.replay()))
pin = torch.empty((), dtype=torch.float32).pin_memory()
ev = torch.cuda.Event()
def f():
    entry.graph.replay(); pin.copy_(entry.loss, non_blocking=True); ev.record(); ev.synchronize(); return float(pin)
print("replay + async D2H to pinned + event sync %.1f us" % t(f))
from pyro_amd.infer.svi import _arg_key
print("_arg_key                         %.1f us" % t(lambda: (_arg_key((X, y)), _arg_key(()))))

# ---- anatomy of one step() on the host (perf_counter stamps inside the fast path) -----------------
import numpy as np
def anatomy(n=400):
    rows = []
    torch.cuda.synchronize()
    for _ in range(n):
        t0 = time.perf_counter()
        key = (_arg_key((X, y)), _arg_key(()))
        e = svi._graphs[key]
        from pyro_amd import kernels
        kernels.glm_planes_revalidate(); kernels.lda_index_revalidate(); kernels.bow_revalidate()
        e.cap.before_replay()
        t1 = time.perf_counter()
        e.graph.replay()
        t2 = time.perf_counter()
        e.cap.after_replay()
        loss = e.read_loss()
        t3 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2))
    a = np.array(rows[50:]) * 1e6
    print("host anatomy (median us): before replay %.1f | graph.replay() call %.1f | poll until loss %.1f | total %.1f"
          % (np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2]), np.median(a.sum(1))))
anatomy()
print("back-to-back replays, 300 per sync: %.1f us per replay (GPU-side cadence without the host in the loop)"
      % t(lambda: entry.graph.replay(), n=300))
