#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, torch
sys.path.insert(0, "tools")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import TraceMeanField_ELBO, trace_mean_field_elbo as tmf, SVI
from pyro_amd.infer.autoguide import AutoNormal
dev = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(10000, 32, dev, seed=0)
real = tmf._add_normal_kl
def dbg(batch, gsite, msite):
    r = real(batch, gsite, msite)
    print("site", type(gsite["fn"]), type(msite["fn"]), tmf._normal_operands(gsite["fn"]) is not None, tmf._normal_operands(msite["fn"]) is not None, r)
    q, p = tmf._normal_operands(gsite["fn"]), tmf._normal_operands(msite["fn"])
    if q and p:
        print([tuple(t.shape) for t in q[:2]], q[2:], [tuple(t.shape) for t in p[:2]], p[2:], gsite["mask"], gsite["scale"])
    return r
tmf._add_normal_kl = dbg
g = AutoNormal(examples.logreg_model, init_scale=0.1)
svi = SVI(examples.logreg_model, g, pyro.optim.Adam({"lr": 0.01}), TraceMeanField_ELBO(num_particles=4, vectorize_particles=True, max_plate_nesting=1), hip_graph=False)
svi.step(X, y)
PY
