"""Where does a config-4 (LDA) step spend its GPU time, by the Python frame that launched the op?
(developer tool)"""
import collections
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, TraceEnum_ELBO

dev = torch.device("cuda:0")
args = examples.LdaArgs(num_docs=100000)
pyro.set_rng_seed(0)
pyro.clear_param_store()
pyro.enable_validation("--validate" in sys.argv)
data = examples.synthetic_lda_data(args, dev)
predictor = examples.lda_make_predictor(args, dev)
guide = lambda data, args: examples.lda_guide(predictor, data, args)  # noqa: E731
svi = SVI(examples.lda_model, guide, pyro.optim.TorchAdam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=2))
for _ in range(3):
    svi.step(data, args)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    svi.step(data, args)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=32,
                                                          max_name_column_width=46, max_shapes_column_width=60))

print("---- by Python stack (device time of ops launched under each frame) ----")
rows = sorted(prof.key_averages(group_by_stack_n=12), key=lambda e: -e.self_device_time_total)[:22]
for e in rows:
    frames = [f for f in e.stack if "pyro_amd" in f or "examples" in f or "tools/" in f][:3]
    print("%9.1f us  x%-3d %-28s %s" % (e.self_device_time_total, e.count, e.key[:28], " <- ".join(f.split("/")[-1][:60] for f in frames)))

print("---- sums / reductions with shapes ----")
rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=14)
        if e.key in ("aten::sum", "aten::copy_", "aten::mul", "aten::add", "aten::div", "aten::exp", "aten::log", "aten::fill_", "aten::zeros", "aten::gather", "aten::index", "aten::index_select") and e.device_time_total > 20]
for e in sorted(rows, key=lambda e: -e.device_time_total)[:24]:
    frames = [f for f in e.stack if "pyro_amd" in f or "examples" in f or "torch/distributions" in f][:4]
    print("%9.1f us  x%-3d %-14s %-50s %s" % (e.device_time_total, e.count, e.key, str(e.input_shapes)[:50], " <- ".join(f.split("/")[-1][:48] for f in frames)))
