#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -8
import sys, torch
sys.path.insert(0, "tools")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, TraceEnum_ELBO
dev = torch.device("cuda:0")
args = examples.LdaArgs(num_docs=100000)
data = examples.synthetic_lda_data(args, dev)
for graph in (False, True):
    pyro.clear_param_store(); pyro.set_rng_seed(0); torch.manual_seed(0); pyro.enable_validation(False)
    predictor = examples.lda_make_predictor(args, dev)
    seen = []
    def guide(data, args):
        return examples.lda_guide(predictor, data, args, 32)
    svi = SVI(examples.lda_model, guide, pyro.optim.ClippedAdam({"lr": 0.0}), TraceEnum_ELBO(max_plate_nesting=2), hip_graph=graph, graph_warmup=3)
    losses = [round(svi.step(data, args)) for _ in range(12)]
    print("graph", graph, "distinct losses", len(set(losses)), losses[:12])
PY
