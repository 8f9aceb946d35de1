"""tests/optim/test_optim.py of the reference restated against the drop-in API (SURVEY 8a row a17,
8f rank 1), on the CPU host logic (fused kernels replaced by the oracle backend where needed):
per-parameter arguments through a callable, save / load with step counts, gradient clipping through
clip_args, ClippedAdam's clamp / pass-through / learning-rate decay, and the interchange of
checkpoints between the flat fused Adam and the per-parameter torch optimizers."""
import io

import numpy as np
import pytest
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd import optim
from pyro_amd.infer import SVI, TraceGraph_ELBO
from pyro_amd.optim.clipped_adam import ClippedAdam as TorchClippedAdam


@pytest.mark.parametrize("fixed_param,free_param", [("loc_q", "log_sig_q"), ("log_sig_q", "loc_q")])
def test_per_param_optim(monkeypatch, fixed_param, free_param):
    """lr = 0 reaches exactly the parameter the callable names; get_state counts steps per parameter;
    a checkpoint written after one step resumes at step two (test_optim.py:32-84)."""
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    lam0, loc0, lam, data = (torch.tensor([v]) for v in (0.1, 0.5, 6.0, 1.0))
    pyro.clear_param_store()

    def model():
        loc_latent = pyro.sample("loc_latent", dist.Normal(loc0, torch.pow(lam0, -0.5)))
        pyro.sample("obs", dist.Normal(loc_latent, torch.pow(lam, -0.5)), obs=data)
        return loc_latent

    def guide():
        loc_q = pyro.param("loc_q", torch.zeros(1, requires_grad=True))
        log_sig_q = pyro.param("log_sig_q", torch.zeros(1, requires_grad=True))
        pyro.sample("loc_latent", dist.Normal(loc_q, torch.exp(log_sig_q)))

    def optim_params(param_name):
        return {"lr": 0.00} if param_name == fixed_param else {"lr": 0.01}

    def get_steps(adam):
        state = adam.get_state()["loc_q"]["state"]
        return int(list(state.values())[0]["step"])

    adam, adam2 = optim.Adam(optim_params), optim.Adam(optim_params)
    svi = SVI(model, guide, adam, loss=TraceGraph_ELBO())
    svi2 = SVI(model, guide, adam2, loss=TraceGraph_ELBO())
    svi.step()
    assert get_steps(adam) == 1
    buf = io.BytesIO()
    torch.save(adam.get_state(), buf)
    svi.step()
    assert get_steps(adam) == 2
    buf.seek(0)
    adam2.set_state(torch.load(buf, weights_only=False))
    svi2.step()
    assert get_steps(adam2) == 2
    assert torch.equal(pyro.param(fixed_param).data, torch.zeros(1))
    assert not torch.equal(pyro.param(free_param).data, torch.zeros(1))


@pytest.mark.parametrize("pyro_optim", [optim.Adam, optim.SGD])
@pytest.mark.parametrize("clip", ["clip_norm", "clip_value"])
@pytest.mark.parametrize("value", [1.0, 3.0, 5.0])
def test_clip_args(pyro_optim, clip, value):
    """A gradient above the threshold, clipped, moves a parameter exactly as the threshold itself
    does un-clipped (test_optim.py:176-194)."""
    x1 = torch.tensor(0.0, requires_grad=True)
    x2 = torch.tensor(0.0, requires_grad=True)
    opt_c = pyro_optim({"lr": 1.0}, {clip: value})
    # the un-clipped comparator is the per-parameter optimizer too: the flat fused Adam zeroes the
    # gradient in its own launch, so ``x.grad`` could not be inspected after the step
    opt = optim.TorchAdam({"lr": 1.0}) if pyro_optim is optim.Adam else optim.SGD({"lr": 1.0})
    assert isinstance(opt_c, optim.PyroOptim)
    for step in range(3):
        x1.backward(torch.empty(()).uniform_(value, value + 3.0))
        x2.backward(torch.tensor(value))
        opt_c([x1])
        opt([x2])
        assert abs(x1.grad.item() - value) < 1e-5 and x2.grad.item() == value   # (norm clip: v/(v+1e-6))
        torch.testing.assert_close(x1, x2, rtol=1e-5, atol=1e-5)
        opt_c.optim_objs[x1].zero_grad()
        opt.optim_objs[x2].zero_grad()


def test_callable_clip_args():
    """clip_args as a callable of (module name, parameter name) -> dict, per parameter, as
    pyro/optim/optim.py:238-255: one parameter clipped, the other not."""
    pyro.clear_param_store()
    a = pyro.param("mod$$$a", torch.tensor(0.0))
    b = pyro.param("b", torch.tensor(0.0))
    seen = []

    def clip(module_name, param_name):
        seen.append((module_name, param_name))
        return {"clip_value": 1.0} if param_name == "a" else {}

    opt = optim.SGD({"lr": 1.0}, clip)
    ua, ub = pyro.get_param_store()._params["mod$$$a"], pyro.get_param_store()._params["b"]
    ua.grad = torch.tensor(5.0)
    ub.grad = torch.tensor(5.0)
    opt([ua, ub])
    assert ("mod", "a") in seen and ("b", "b") in seen      # (a name without a module: the reference passes it twice)
    assert float(ua) == -1.0 and float(ub) == -5.0


@pytest.mark.parametrize("clip_norm", [1.0, 3.0, 5.0])
def test_clippedadam_clip(clip_norm):
    x1 = torch.tensor(0.0, requires_grad=True)
    x2 = torch.tensor(0.0, requires_grad=True)
    opt_ca = TorchClippedAdam(params=[x1], lr=1.0, lrd=1.0, clip_norm=clip_norm)
    opt_a = torch.optim.Adam(params=[x2], lr=1.0)
    for step in range(3):
        opt_ca.zero_grad()
        opt_a.zero_grad()
        x1.backward(torch.empty(()).uniform_(clip_norm, clip_norm + 3.0))
        x2.backward(torch.tensor(clip_norm))
        opt_ca.step()
        opt_a.step()
        torch.testing.assert_close(x1, x2)


@pytest.mark.parametrize("clip_norm", [1.0, 3.0, 5.0])
def test_clippedadam_pass(clip_norm):
    torch.manual_seed(2)          # (a tiny draw makes the two Adams differ visibly in the eps term)
    x1 = torch.tensor(0.0, requires_grad=True)
    x2 = torch.tensor(0.0, requires_grad=True)
    opt_ca = TorchClippedAdam(params=[x1], lr=1.0, lrd=1.0, clip_norm=clip_norm)
    opt_a = torch.optim.Adam(params=[x2], lr=1.0)
    for step in range(3):
        g = torch.empty(()).uniform_(0.05 * clip_norm, clip_norm) * (1 - 2 * (step % 2))
        opt_ca.zero_grad()
        opt_a.zero_grad()
        x1.backward(g)
        x2.backward(g)
        opt_ca.step()
        opt_a.step()
        torch.testing.assert_close(x1, x2)


@pytest.mark.parametrize("lrd", [1.0, 3.0, 5.0])
def test_clippedadam_lrd(lrd):
    x1 = torch.tensor(0.0, requires_grad=True)
    opt_ca = TorchClippedAdam(params=[x1], lr=1.0, lrd=lrd)
    for step in range(3):
        x1.backward(torch.empty(()).uniform_(-5.0, 5.0))
        opt_ca.step()
        assert opt_ca.param_groups[0]["lr"] == 1.0 * lrd ** (step + 1)


@pytest.mark.parametrize("clipped", [False, True])
def test_flat_and_per_parameter_optimizers_agree_and_exchange_checkpoints(monkeypatch, clipped):
    """The flat fused (Clipped)Adam and the per-parameter torch route: same trajectory from the
    same gradients, and a state saved by either resumes in the other."""
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    torch.set_default_dtype(torch.float64)
    try:
        rng = np.random.default_rng(0)
        shapes = {"a": (3,), "b": (2, 2), "c": ()}
        grads = [{n: torch.tensor(3.0 * rng.standard_normal(sh)) for n, sh in shapes.items()}
                 for _ in range(6)]
        args = {"lr": 0.05, "betas": (0.8, 0.95)}
        if clipped:
            args.update({"clip_norm": 2.0, "lrd": 0.9})
        make_flat = lambda: (optim.ClippedAdam if clipped else optim.Adam)(dict(args))   # noqa: E731
        make_per = lambda: optim.PyroOptim(TorchClippedAdam if clipped else torch.optim.Adam, dict(args))  # noqa: E731

        def run(first, second, switch_at):
            pyro.clear_param_store()
            ps = {n: pyro.param(n, torch.zeros(sh)) for n, sh in shapes.items()}
            leaves = [p.unconstrained() if hasattr(p, "unconstrained") else p for p in ps.values()]
            opt = first()
            for k, g in enumerate(grads):
                if k == switch_at:
                    nxt = second()
                    nxt.set_state(opt.get_state())
                    opt = nxt
                for leaf, n in zip(leaves, shapes):
                    leaf.grad = g[n].clone()
                opt(leaves)
            return {n: pyro.param(n).detach().clone() for n in shapes}

        ref = run(make_per, make_per, None)
        for first, second, at in ((make_flat, make_flat, None), (make_flat, make_per, 3),
                                  (make_per, make_flat, 3), (make_flat, make_flat, 2)):
            got = run(first, second, at)
            for n in shapes:
                np.testing.assert_allclose(got[n].numpy(), ref[n].numpy(), rtol=1e-10, atol=1e-12)
    finally:
        torch.set_default_dtype(torch.float32)


def test_lr_scheduler_wrappers_and_their_checkpoints():
    """pyro.optim.<Scheduler>({"optimizer": ..., "optim_args": ..., **scheduler_args}) (tests/optim/
    test_optim.py:332-440): svi.step steps the optimizers, scheduler.step() the schedules; the state holds
    both; every optimizer / scheduler class of torch has a wrapper."""
    import pyro_amd as pyro
    from pyro_amd import optim
    for name in ("Adagrad", "Adamax", "RMSprop", "SGD", "ExponentialLR", "StepLR", "LambdaLR",
                 "ReduceLROnPlateau", "MultiStepLR", "CosineAnnealingLR"):
        assert hasattr(optim, name), name
    pyro.clear_param_store()
    pyro.param("x", torch.tensor(1.0))

    def run(sched, epochs):
        for _ in range(epochs):
            u = pyro.param("x").unconstrained()
            u.grad = torch.tensor(1.0)
            sched([u])
            sched.step()
        return pyro.param("x").item()

    cfg = {"optimizer": torch.optim.SGD, "optim_args": {"lr": 0.1}, "gamma": 0.5}
    sched = optim.ExponentialLR(dict(cfg))
    assert abs(run(sched, 3) - (1.0 - 0.1 - 0.05 - 0.025)) < 1e-6
    state = sched.get_state()
    assert set(state["x"]) == {"scheduler", "optimizer"}
    again = optim.ExponentialLR(dict(cfg))
    again.set_state(state)
    assert abs(run(again, 1) - (0.825 - 0.0125)) < 1e-6        # continues at the decayed rate
    assert not optim.optim.is_scheduler(torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=0.1))


def test_per_parameter_arguments_receive_module_style_names():
    """tests/optim/test_optim.py (dynamic lr, name_preserved_by_to_pyro_module): a callable gets
    ``"net.bias"`` for a pyro.module parameter, or (module, parameter) in the deprecated two-argument form."""
    import pyro_amd as pyro
    from pyro_amd import optim
    pyro.clear_param_store()
    net = torch.nn.Linear(2, 1)
    pyro.module("net", net)
    pyro.param("free", torch.zeros(3))
    leaves = [pyro.param(n).unconstrained() for n in sorted(pyro.get_param_store().keys())]
    for leaf in leaves:
        leaf.grad = torch.ones_like(leaf)
    seen, seen2 = set(), set()

    def one_arg(name):
        seen.add(name)
        return {"lr": 0.1 if name == "free" else 0.0}

    def two_args(module_name, param_name):
        seen2.add((module_name, param_name))
        return {"lr": 0.0}

    before = net.bias.detach().clone()
    optim.SGD(one_arg)(leaves)
    assert seen == {"free", "net.weight", "net.bias"}
    assert torch.equal(net.bias.detach(), before)                      # lr 0 for the module's parameters
    assert torch.allclose(pyro.param("free"), torch.full((3,), -0.1))
    optim.SGD(two_args)(leaves)
    assert seen2 == {("free", "free"), ("net", "weight"), ("net", "bias")}


def test_no_update_optimizer_is_a_flat_optimizer_with_a_zero_rate():
    """pyro_amd.optim.NoUpdate: the flat buffers of the package's Adam (so that a captured step keeps its fused
    tail) with lr = 0 -- the device code returns before touching parameter or moments; no arguments."""
    import pytest

    import pyro_amd as pyro
    from pyro_amd.infer.svi import _FlatOptimOK

    o = pyro.optim.NoUpdate()
    assert o.lr == 0.0 and o.weight_decay == 0.0 and o.zeroes_grads and _FlatOptimOK(o)
    with pytest.raises(ValueError):
        pyro.optim.NoUpdate({"lr": 0.1})
