#!/usr/bin/env python
"""The UNMODIFIED reference (pyro 1.9.1 under /root/reference, with oracle/refshim's opt_einsum
stand-in on the path) timed on BASELINE configs[1] on this container's host cores:

    SVI(model, AutoNormal(model), Adam, Trace_ELBO(num_particles=64, vectorize_particles=True)).step(X, y)

(pyro/infer/svi.py:134-162), N=1e6, D=32, f32, validation on and off.  /root/reference does not exist on
the GPU box, so this runs HERE (the build container: 8 cores) and its result is committed under
profiles/ as a fixture; bench.py prints it in `cpu_baseline.reference` beside the port it times live.

    python tools/time_reference_cpu.py [--steps 6] [--out profiles/r06_reference_cpu.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle", "refshim"))
sys.path.insert(0, "/root/reference")

import pyro  # noqa: E402
import pyro.distributions as dist  # noqa: E402
from pyro.infer import SVI, Trace_ELBO  # noqa: E402
from pyro.infer.autoguide import AutoNormal  # noqa: E402

assert pyro.__version__ == "1.9.1" and pyro.__file__.startswith("/root/reference")


def model(X, y):                  # SURVEY 8(d)'s model text
    N, D = X.shape
    w = pyro.sample("w", dist.Normal(X.new_zeros(D), 1.0).to_event(1))
    b = pyro.sample("b", dist.Normal(X.new_zeros(()), 1.0))
    with pyro.plate("data", N):
        logits = w @ X.t()
        logits = logits.squeeze(-2) if logits.dim() > 1 else logits
        pyro.sample("obs", dist.Bernoulli(logits=logits + b), obs=y)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--plate", type=int, default=1_000_000)
    ap.add_argument("--features", type=int, default=32)
    ap.add_argument("--particles", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_reference_cpu.json"))
    a = ap.parse_args()
    N, D, P = a.plate, a.features, a.particles
    g = torch.Generator().manual_seed(0)
    X = torch.randn((N, D), generator=g)
    w_true = torch.randn((D,), generator=g)
    y = (torch.rand((N,), generator=g) < torch.sigmoid(X @ w_true)).float()
    threads = torch.get_num_threads()
    res = {}
    for validate in (False, True, False, True):       # alternated: the best of the two passes each
        pyro.clear_param_store()
        pyro.set_rng_seed(0)
        pyro.enable_validation(validate)
        svi = SVI(model, AutoNormal(model, init_scale=0.1), pyro.optim.Adam({"lr": 0.01}),
                  Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1))
        svi.step(X, y)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            loss = svi.step(X, y)
        dt = time.perf_counter() - t0
        k = "validation_on" if validate else "validation_off"
        if k not in res or a.steps / dt > res[k]["steps_per_s"]:
            res[k] = {"steps_per_s": a.steps / dt, "s_per_step": dt / a.steps, "steps": a.steps, "last_loss": loss,
                      "passes": 2}
    out = {"what": "unmodified reference pyro %s SVI.step, BASELINE configs[1] (N=%d, D=%d, P=%d, f32, AutoNormal, "
                   "Adam, Trace_ELBO vectorize_particles)" % (pyro.__version__, N, D, P),
           "where": "build container (no GPU)", "cores": os.cpu_count(), "torch_threads": threads,
           "cpu": next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?"),
           "torch": torch.__version__, "how": "tools/time_reference_cpu.py", **res}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
