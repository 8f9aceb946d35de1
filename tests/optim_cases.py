"""Flat fused Adam against the reference's execution model -- one torch.optim.Adam per parameter,
stepped only when the parameter is passed (pyro/optim/optim.py:117-155) -- shared by the CPU
host-logic tests and the MI355X tests."""
import io

import numpy as np
import torch


def _make_params(device, dtype):
    import pyro_amd as pyro
    pyro.clear_param_store()
    g = torch.Generator().manual_seed(0)
    shapes = {"a": (3,), "b": (2, 2), "c": (5,), "late": (4,)}
    vals = {k: torch.randn(s, generator=g, dtype=torch.float64) for k, s in shapes.items()}
    for k, v in vals.items():
        if k != "late":
            pyro.param(k, v.to(device=device, dtype=dtype))
    store = pyro.get_param_store()
    leaves = {k: store._params[k] for k in vals if k != "late"}
    return vals, leaves


def _grads(step, names, dtype=torch.float64):
    g = torch.Generator().manual_seed(100 + step)
    shapes = {"a": (3,), "b": (2, 2), "c": (5,), "late": (4,)}
    return {k: torch.randn(shapes[k], generator=g, dtype=torch.float64) for k in names}


def run_semantics(device, dtype=torch.float64, tol=1e-12):
    """Partial steps, a parameter that appears at step 4, then a checkpoint round trip."""
    import pyro_amd as pyro
    vals, leaves = _make_params(device, dtype)
    opt = pyro.optim.Adam({"lr": 0.05, "betas": (0.9, 0.99)})
    # reference model: independent torch.optim.Adam objects on float64 CPU copies
    ref_p = {k: v.clone().requires_grad_(True) for k, v in vals.items() if k != "late"}
    ref_o = {k: torch.optim.Adam([p], lr=0.05, betas=(0.9, 0.99)) for k, p in ref_p.items()}
    schedule = [["a", "b", "c"], ["a", "b", "c"], ["a", "c"], ["b"], ["a", "b", "c", "late"],
                ["late", "a"], ["a", "b", "c", "late"]]
    saved = None
    for step, names in enumerate(schedule):
        if "late" in names and "late" not in leaves:
            pyro.param("late", vals["late"].to(device=device, dtype=dtype))
            leaves["late"] = pyro.get_param_store()._params["late"]
            ref_p["late"] = vals["late"].clone().requires_grad_(True)
            ref_o["late"] = torch.optim.Adam([ref_p["late"]], lr=0.05, betas=(0.9, 0.99))
        gr = _grads(step, names)
        for k in names:
            leaves[k].grad = gr[k].to(device=device, dtype=dtype) if leaves[k].grad is None else \
                leaves[k].grad.copy_(gr[k].to(device=device, dtype=dtype))
            ref_p[k].grad = gr[k].clone()
            ref_o[k].step()
        opt([leaves[k] for k in names])
        for k in leaves:
            np.testing.assert_allclose(leaves[k].detach().cpu().double().numpy(),
                                       ref_p[k].detach().numpy(), rtol=tol, atol=tol,
                                       err_msg="step %d param %s" % (step, k))
            assert float(leaves[k].grad.abs().sum()) == 0.0          # zeroed by the update launch
        if step == 4:
            buf = io.BytesIO()
            torch.save(opt.get_state(), buf)
            saved = (buf.getvalue(), {k: v.detach().clone() for k, v in leaves.items()})
    # ---- resume from the checkpoint taken after step 4 with a NEW optimizer object: the last two
    #      steps must land exactly where the uninterrupted run did
    final = {k: v.detach().clone() for k, v in leaves.items()}
    pyro.clear_param_store()
    blob, params_at_4 = saved
    leaves = {}
    for k, v in params_at_4.items():
        pyro.param(k, v.clone())
        leaves[k] = pyro.get_param_store()._params[k]
    opt2 = pyro.optim.Adam({"lr": 0.05, "betas": (0.9, 0.99)})
    opt2.set_state(torch.load(io.BytesIO(blob), weights_only=False))
    for step in (5, 6):
        names = schedule[step]
        gr = _grads(step, names)
        for k in names:
            leaves[k].grad = gr[k].to(device=device, dtype=dtype)
        opt2([leaves[k] for k in names])
    for k in final:
        np.testing.assert_allclose(leaves[k].detach().cpu().double().numpy(),
                                   final[k].cpu().double().numpy(), rtol=tol, atol=tol,
                                   err_msg="resumed run, param %s" % k)
    pyro.clear_param_store()
