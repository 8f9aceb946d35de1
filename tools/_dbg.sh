#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -12
import sys, torch
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, TraceEnum_ELBO, traceenum_elbo as te
dev = torch.device("cuda:0")
args = examples.LdaArgs(num_docs=1000)
data = examples.synthetic_lda_data(args, dev)
predictor = examples.lda_make_predictor(args, dev)
guide = lambda data, args: examples.lda_guide(predictor, data, args)
real = te._enum_log_prob
def dbg(site):
    fn, value = site["fn"], site["value"]
    base = getattr(fn, "_base_logits", None)
    print("enum site", site["name"], type(fn), None if base is None else tuple(base.shape), tuple(value.shape), site["infer"].get("_enumerate_dim"), type(fn).log_prob is torch.distributions.Categorical.log_prob)
    r = real(site)
    print(" ->", tuple(r.shape), r.stride())
    return r
te._enum_log_prob = dbg
svi = SVI(examples.lda_model, guide, pyro.optim.ClippedAdam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=2))
svi.step(data, args)
PY
