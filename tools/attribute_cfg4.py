"""Which host-side source lines issue the small launches of a config-4 step (developer tool): one eager
step under torch.profiler with stacks, device kernels grouped by the innermost pyro_amd / examples frame."""
import collections
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, TraceEnum_ELBO

dev = torch.device("cuda:0")
args = examples.LdaArgs(num_docs=100_000)
data = examples.synthetic_lda_data(args, dev)
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
predictor = examples.lda_make_predictor(args, dev)
guide = lambda data, args: examples.lda_guide(predictor, data, args, None)  # noqa: E731
svi = SVI(examples.lda_model, guide, pyro.optim.ClippedAdam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=2),
          hip_graph=False)
for _ in range(4):
    svi.step(data, args)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

SKIP = ("aten::view", "aten::expand", "aten::reshape", "aten::_unsafe_view", "aten::t", "aten::transpose",
        "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::as_strided", "aten::permute",
        "aten::select", "aten::slice", "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::_to_copy")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name().rsplit(".", 1)[0] if func.name().count(".") else func.name()
        if not name.startswith(SKIP):
            frame = "(autograd thread / no python frame)"
            for fr in reversed(traceback.extract_stack()[:-1]):
                fn = fr.filename
                if ("pyro_amd/" in fn or fn.endswith("examples.py")) and "poutine/runtime" not in fn \
                        and "poutine/handlers" not in fn and "tools/" not in fn and "ops/lazy.py" not in fn:
                    frame = "%s:%d %s" % (fn.split("pyro_amd/")[-1], fr.lineno, fr.name)
                    break
            self.by[frame][name] += 1
        return func(*args, **(kwargs or {}))


with Count() as cnt:
    svi.step(data, args)
    torch.cuda.synchronize()
tot = sum(sum(c.values()) for c in cnt.by.values())
print("non-view operators dispatched by one eager step:", tot)
for fr, c in sorted(cnt.by.items(), key=lambda x: -sum(x[1].values()))[:45]:
    print("%4d  %-58s %s" % (sum(c.values()), fr[-58:], dict(c.most_common(5))))
