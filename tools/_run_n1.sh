#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "test_dist_log_prob_sum_grad or test_dist_log_prob_nd or test_dist_empty" 2>&1 | tail -15
python -m pytest tests/test_svi_gpu.py -x -q -m gpu -k "gamma_function or scale_mask or eight_schools" 2>&1 | tail -15
python - <<'PY'
import numpy as np, torch
import pyro_amd.distributions as d
g = np.load("tests/golden/dists.npz")
dev = "cuda:0"
for fam, mk in [("gamma", lambda a, b: d.Gamma(a, b)), ("beta", lambda a, b: d.Beta(a, b)),
                ("poisson", lambda a, b: d.Poisson(a)), ("binomial_logits", lambda a, b: d.Binomial(b, logits=a))]:
    for dt, tol in [(torch.float64, 1e-11), (torch.float32, 1e-4)]:
        v = torch.tensor(g[fam + "/v"], dtype=dt, device=dev, requires_grad=fam in ("gamma", "beta"))
        a = torch.tensor(g[fam + "/a"], dtype=dt, device=dev, requires_grad=True)
        b = torch.tensor(g[fam + "/b"], dtype=dt, device=dev, requires_grad=fam in ("gamma", "beta")) if fam + "/b" in g.files else None
        dd = mk(a, b)
        lp = dd.log_prob(v)
        np.testing.assert_allclose(lp.detach().cpu().numpy(), g[fam + "/lp"], rtol=tol, atol=tol)
        ins = [t for t in (v, a, b) if t is not None and t.requires_grad]
        names = [n for n, t in (("dv", v), ("da", a), ("db", b)) if t is not None and t.requires_grad]
        for route in ("lp", "sum"):
            out = lp.sum() if route == "lp" else dd.fused_log_prob_sum(v)
            gs = torch.autograd.grad(out, ins, retain_graph=True)
            for n, gg in zip(names, gs):
                ref = g[fam + "/" + n]
                np.testing.assert_allclose(gg.cpu().numpy(), ref, rtol=tol * 10, atol=tol * 10 * np.abs(ref).max())
        print(fam, dt, "ok", float(lp.sum()))
PY
