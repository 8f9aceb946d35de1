"""INTEGRATION.md's stubs executed under the UNMODIFIED reference (pyro 1.9.1 at /root/reference):

* seam 1: a ``TorchDistribution`` subclass for the plated GLM site whose ``fused_log_prob_sum``
  goes through the C-ABI contract of ``pa_glm_bernoulli_fwd_bwd`` (one call: ll, gw, gb; the
  autograd dual is ``g * gw``, ``g * gb``);
* seam 2: ``HipTrace_ELBO(pyro.infer.Trace_ELBO)`` pre-filling ``site["log_prob_sum"]``
  (pyro/poutine/trace_struct.py:221-222,262) before the reference's own estimator runs;
* the reference's ``pyro.infer.SVI`` / ``pyro.optim.Adam`` drive it (pyro/infer/svi.py:76-90,144-156).

There is no GPU in the build container, so the kernel behind the binding is answered by the numpy
oracle (``oracle/glm.py``) with exactly the entry point's signature and outputs -- the point of
this script is the SEAMS: that unmodified Pyro accepts the plugin objects, that the loss and every
parameter gradient equal the golden vectors the unmodified reference produced on its own path
(tests/golden/logreg_f64.npz), and that an SVI step runs end to end.

Run as a script (tests/test_reference_seam.py does, in a subprocess with
PYTHONDONTWRITEBYTECODE=1 so that nothing is written under /root/reference):

    python tests/reference_seam.py            -> prints "SEAM OK ..." and exits 0
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "refshim"))  # after `import torch` on purpose
sys.path.insert(0, "/root/reference")

import pyro  # noqa: E402
import pyro.distributions as dist  # noqa: E402
from pyro.distributions.torch_distribution import TorchDistribution  # noqa: E402
from pyro.infer import SVI, Trace_ELBO  # noqa: E402
from pyro.infer.autoguide import AutoNormal  # noqa: E402
from pyro.infer.enum import get_importance_trace  # noqa: E402
from torch.distributions import constraints  # noqa: E402

from oracle import glm as o_glm  # noqa: E402  (stands in for libpyro_amd.so: no GPU here)

assert pyro.__version__ == "1.9.1"
CALLS = {"glm": 0}


def pa_glm_bernoulli_fwd_bwd(X, y, w, b, mask, scale):
    """The binding a maintainer adds (pyro_amd/_lib.py, INTEGRATION.md seam 2): contiguous tensors
    in, (ll[P], gw[P,D], gb[P]) out.  Here answered by the numpy oracle."""
    CALLS["glm"] += 1
    ll, gw, gb = o_glm.glm_bernoulli_fwd_bwd(X.numpy(), y.numpy(), w.numpy(),
                                             None if b is None else b.numpy(),
                                             None if mask is None else mask.numpy(), scale)
    t = lambda a: torch.as_tensor(a, dtype=X.dtype)  # noqa: E731
    return t(ll), t(gw), t(gb)


class _GlmLogLik(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, y, w, b, mask, scale):
        ll, gw, gb = pa_glm_bernoulli_fwd_bwd(X, y, w.detach().contiguous(), b.detach().contiguous(),
                                              mask, scale)
        ctx.save_for_backward(gw, gb)
        return ll

    @staticmethod
    def backward(ctx, g):
        gw, gb = ctx.saved_tensors
        return None, None, g[:, None] * gw, g * gb, None, None


class FusedBernoulliLinear(TorchDistribution):
    """Seam 1: Bernoulli(logits = (w @ X^T).squeeze(-2) + b) as a distribution object Pyro's
    handlers accept (batch_shape, expand, log_prob) with the fused protocol on top."""

    arg_constraints = {}
    support = constraints.boolean
    has_rsample = False

    def __init__(self, X, w, b):
        self.X, self.w, self.b = X, w, b
        lead = w.shape[:-2] if w.dim() > 1 else torch.Size()
        super().__init__(torch.Size(lead) + (X.shape[0],), torch.Size(), validate_args=False)

    def _logits(self):
        out = self.w @ self.X.t()
        return (out.squeeze(-2) if self.w.dim() > 1 else out) + self.b

    def expand(self, batch_shape, _instance=None):
        assert torch.Size(batch_shape) == self.batch_shape
        return self

    def sample(self, sample_shape=torch.Size()):
        return dist.Bernoulli(logits=self._logits()).sample(sample_shape)

    def log_prob(self, value):                    # what unmodified Pyro calls (trace_struct.py:264)
        return dist.Bernoulli(logits=self._logits()).log_prob(value)

    def fused_log_prob_sum(self, value, scale, mask):
        lead = self.batch_shape[:-1]
        P = int(np.prod(lead)) if lead else 1
        D = self.X.shape[1]
        w2 = self.w.expand(lead + (1, D)).reshape(P, D) if self.w.dim() > 1 else self.w.reshape(1, D)
        b1 = self.b.expand(lead + (1,)).reshape(P) if self.b.dim() > 0 else self.b.reshape(1).expand(P)
        s = 1.0 if scale is None else float(scale)
        return _GlmLogLik.apply(self.X, value.contiguous(), w2, b1, mask, s).sum()


class HipTrace_ELBO(Trace_ELBO):
    """Seam 2, as INTEGRATION.md writes it."""

    def _get_trace(self, model, guide, args, kwargs):
        model_trace, guide_trace = get_importance_trace("flat", self.max_plate_nesting, model, guide,
                                                        args, kwargs)
        for trace in (model_trace, guide_trace):
            for site in trace.nodes.values():
                if site["type"] == "sample" and hasattr(site["fn"], "fused_log_prob_sum"):
                    site["log_prob_sum"] = site["fn"].fused_log_prob_sum(site["value"], site["scale"],
                                                                         site["mask"])
        return model_trace, guide_trace


def logreg_model(X, y):
    N, D = X.shape
    w = pyro.sample("w", dist.Normal(torch.zeros(D, dtype=X.dtype), 1.0).to_event(1))
    b = pyro.sample("b", dist.Normal(torch.zeros((), dtype=X.dtype), 1.0))
    with pyro.plate("data", N):
        pyro.sample("obs", FusedBernoulliLinear(X, w, b), obs=y)


class _Replay:
    """Feeds recorded standard-normal draws to the reference (as tests/golden/make_golden.py did)."""

    def __init__(self, eps):
        self.eps, self.i = list(eps), 0

    def __call__(self, shape, dtype, device):
        e = self.eps[self.i]
        self.i += 1
        assert tuple(e.shape) == tuple(shape), (e.shape, shape)
        return torch.as_tensor(e, dtype=dtype, device=device)

    def __enter__(self):
        import torch.distributions.normal as tn
        self._old = tn._standard_normal
        tn._standard_normal = self
        return self

    def __exit__(self, *a):
        import torch.distributions.normal as tn
        tn._standard_normal = self._old


def main():
    torch.set_default_dtype(torch.float64)
    g = np.load(os.path.join(HERE, "golden", "logreg_f64.npz"))
    X, y, P = torch.tensor(g["X"]), torch.tensor(g["y"]), int(g["P"])
    pyro.clear_param_store()
    guide = AutoNormal(logreg_model, init_scale=0.1)
    elbo = HipTrace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1)
    eps = [g[k] for k in sorted(k for k in g.files if k.startswith("eps/"))]
    with _Replay(eps):
        loss = elbo.loss_and_grads(logreg_model, guide, X, y)
    assert CALLS["glm"] == 1, CALLS
    assert abs(loss - float(g["loss"])) < 1e-9 * abs(float(g["loss"])), (loss, float(g["loss"]))
    worst = 0.0
    for name, p in pyro.get_param_store().named_parameters():
        ref = g["grads/" + name]
        got = p.grad.detach().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-9, err_msg=name)
        worst = max(worst, float(np.abs(got - ref).max()))
    # second golden point (moved parameters)
    with torch.no_grad():
        for name, p in pyro.get_param_store().named_parameters():
            p.grad = None
        store = pyro.get_param_store()
        for name in list(store.keys()):
            store[name] = torch.tensor(g["params2/" + name])
    eps2 = [g[k] for k in sorted(k for k in g.files if k.startswith("eps2/"))]
    with _Replay(eps2):
        loss2 = elbo.loss_and_grads(logreg_model, guide, X, y)
    assert abs(loss2 - float(g["loss2"])) < 1e-9 * abs(float(g["loss2"])), (loss2, float(g["loss2"]))
    for name, p in pyro.get_param_store().named_parameters():
        np.testing.assert_allclose(p.grad.detach().numpy(), g["grads2/" + name], rtol=1e-8, atol=1e-9,
                                   err_msg=name)
    # the reference's own SVI drives the plugin (svi.py:144-156): a few optimizer steps move the loss
    pyro.clear_param_store()
    pyro.set_rng_seed(0)
    guide = AutoNormal(logreg_model, init_scale=0.1)
    svi = SVI(logreg_model, guide, pyro.optim.Adam({"lr": 0.05}),
              HipTrace_ELBO(num_particles=8, vectorize_particles=True, max_plate_nesting=1))
    n0 = CALLS["glm"]
    losses = [svi.step(X, y) for _ in range(15)]
    assert CALLS["glm"] - n0 == 15
    assert losses[-1] < losses[0], losses
    print("SEAM OK: loss %.6f (golden %.6f), worst |grad - golden| %.2e, %d SVI steps %.1f -> %.1f"
          % (loss, float(g["loss"]), worst, len(losses), losses[0], losses[-1]))


if __name__ == "__main__":
    main()
