#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYRO_AMD_DEBUG_GRAPH=1
python - <<'PY' 2>&1 | tail -30
import sys, time, torch
sys.path.insert(0, "tools")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, TraceEnum_ELBO
dev = torch.device("cuda:0")
args = examples.LdaArgs(num_docs=100000)
data = examples.synthetic_lda_data(args, dev)
for graph in (False, True):
    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    predictor = examples.lda_make_predictor(args, dev)
    guide = lambda data, args: examples.lda_guide(predictor, data, args)
    svi = SVI(examples.lda_model, guide, pyro.optim.ClippedAdam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=2), hip_graph=graph, graph_warmup=3)
    losses = [svi.step(data, args) for _ in range(8)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): l = svi.step(data, args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("graph", graph, "ms/step %.3f" % (dt * 1e3), "graphs", len(svi._graphs), [round(x, 1) for x in losses])
PY
