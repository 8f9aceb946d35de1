import sys, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal
dev = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(1_000_000, 32, dev, seed=0)
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
g = AutoNormal(examples.logreg_model, init_scale=0.1)
svi = SVI(examples.logreg_model, g, pyro.optim.Adam({"lr": 0.01}), Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1))
for _ in range(5): svi.step(X, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    svi.step(X, y); torch.cuda.synchronize()
for e in sorted(prof.key_averages(group_by_input_shape=True, group_by_stack_n=14), key=lambda e: -e.device_time_total)[:30]:
    if e.device_time_total <= 0: continue
    frames = [f for f in e.stack if "pyro_amd" in f or "examples" in f][:4]
    print("%8.1f us x%-2d %-26s %-40s %s" % (e.device_time_total, e.count, e.key[:26], str(e.input_shapes)[:40], " <- ".join(f.split("/")[-1][:44] for f in frames)))
