"""Which ATen kernels does one eager step of each config-2 model variant launch? (developer tool)"""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal

dev = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(200_000, 32, dev)
pyro.enable_validation(False)
for name, model in (("verbatim", examples.logreg_model), ("explicit", examples.logreg_model_explicit)):
    pyro.clear_param_store()
    guide = AutoNormal(model, init_scale=0.1)
    svi = SVI(model, guide, pyro.optim.Adam({"lr": 0.01}),
              Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1))
    for _ in range(4):
        svi.step(X, y)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 record_shapes=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        svi.step(X, y)
        torch.cuda.synchronize()
    print("=====", name)
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name in ("aten::add", "aten::add_", "aten::fill_", "aten::zeros", "aten::copy_"):
            st = [s for s in (ev.stack or [])][:12]
            print(ev.name, [tuple(i) for i in (ev.input_shapes or [])][:3])
            for s_ in st:
                print("      ", s_[-110:])
