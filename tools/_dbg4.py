import sys, traceback, collections
sys.path.insert(0, ".")
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, TraceEnum_ELBO
dev = torch.device("cuda:0")
args = examples.LdaArgs(num_docs=100000)
data = examples.synthetic_lda_data(args, dev)
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
predictor = examples.lda_make_predictor(args, dev)
guide = lambda data, args: examples.lda_guide(predictor, data, args)
svi = SVI(examples.lda_model, guide, pyro.optim.TorchAdam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=2))
for _ in range(2): svi.step(data, args)
big = collections.Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), kw=None):
        out = func(*a, **(kw or {}))
        n = max([t.numel() for t in list(a) + [out] if isinstance(t, torch.Tensor)] + [0])
        isl = any(isinstance(t, torch.Tensor) and t.dtype in (torch.int64, torch.bool) for t in list(a) + [out])
        if n >= 1_000_000 and isl:
            st = [f for f in traceback.extract_stack() if "pyro_amd" in f.filename or "tools/" in f.filename]
            where = "; ".join("%s:%d" % (f.filename.split("/")[-1], f.lineno) for f in st[-5:]) if st else "autograd"
            big[(str(func), where, n)] += 1
        return out
with M():
    svi.step(data, args)
for k, v in sorted(big.items(), key=lambda x: -x[0][2])[:16]: print(v, k)
