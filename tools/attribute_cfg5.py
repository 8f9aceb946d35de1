"""Which host-side source lines issue the torch operators of a config-5 step (developer tool): one eager
step under a TorchDispatchMode, non-view operators grouped by the innermost pyro_amd / examples frame."""
import collections
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal

dev = torch.device("cuda:0")
N, D, G, P = 1_000_000, 32, 1000, 64
X, y, g = examples.synthetic_hier_logreg_data_unsorted(N, D, G, dev, seed=1)
model = lambda X_, y_, g_: examples.hier_logreg_model_reference(X_, y_, g_, G)  # noqa: E731
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
svi = SVI(model, AutoNormal(model, init_scale=0.1), pyro.optim.Adam({"lr": 0.01}),
          Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1), hip_graph=False)
for _ in range(4):
    svi.step(X, y, g)
torch.cuda.synchronize()
SKIP = ("aten::view", "aten::expand", "aten::reshape", "aten::_unsafe_view", "aten::t", "aten::transpose",
        "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::as_strided", "aten::permute",
        "aten::select", "aten::slice", "aten::empty", "aten::empty_like", "aten::empty_strided")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name()
        out = func(*args, **(kwargs or {}))
        if not name.startswith(SKIP):
            frame = "(autograd thread / no python frame)"
            for fr in reversed(traceback.extract_stack()[:-1]):
                fn = fr.filename
                if ("pyro_amd/" in fn or fn.endswith("examples.py")) and "poutine/runtime" not in fn \
                        and "poutine/handlers" not in fn and "tools/" not in fn:
                    frame = "%s:%d %s" % (fn.split("pyro_amd/")[-1], fr.lineno, fr.name)
                    break
            shp = tuple(out.shape) if isinstance(out, torch.Tensor) else None
            self.rows.append((name, shp, frame))
        return out


with Count() as cnt:
    svi.step(X, y, g)
    torch.cuda.synchronize()
print("non-view operators dispatched by one eager step:", len(cnt.rows))
for r in cnt.rows:
    print("%-28s %-18s %s" % r)
