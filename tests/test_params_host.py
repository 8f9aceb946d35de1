"""The parameter store (CPU): mapping interface, constraints, persistence together with ``pyro.module``,
nested scopes.  The behaviours are the ones tests/params/test_param.py of the reference pins."""
import pytest
import torch
from torch.distributions import constraints

import pyro_amd as pyro


@pytest.fixture
def store():
    s = pyro.get_param_store()
    s.clear()
    yield s
    s.clear()


def _same(a, b):
    torch.testing.assert_close(torch.as_tensor(a).detach(), torch.as_tensor(b).detach(), rtol=1e-6, atol=1e-6)


# ---- the mapping interface --------------------------------------------------------------------------------
def test_an_empty_store_is_falsy_and_empty(store):
    assert not store and len(store) == 0
    assert [list(view) for view in (store.keys(), store.values(), store.items())] == [[], [], []]
    assert "anything" not in store


def test_assignment_setdefault_and_deletion(store):
    store["w"] = torch.zeros(1, 2, 3)
    store.setdefault("s", torch.ones(4, 5), constraint=constraints.positive)
    assert store and len(store) == 2 and set(store.keys()) == {"w", "s"} == {k for k, _ in store.items()}
    assert store["w"].shape == (1, 2, 3) and store["s"].shape == (4, 5)
    # setdefault never overwrites
    _same(store.setdefault("w", torch.ones(1, 2, 3)), torch.zeros(1, 2, 3))
    _same(store.setdefault("s", torch.zeros(4, 5)), torch.ones(4, 5))
    # a real-valued parameter IS its unconstrained leaf; a positive one lives in log space
    assert store["w"].unconstrained() is store["w"]
    _same(store["s"].unconstrained(), torch.zeros(4, 5))
    del store["w"]
    assert "w" not in store and list(store.keys()) == ["s"]
    _same(store["s"].unconstrained(), torch.zeros(4, 5))
    del store["s"]
    assert not store


# ---- persistence and pyro.module ---------------------------------------------------------------------------
def test_a_saved_store_feeds_a_new_module_object(store, tmp_path):
    trained, bystander, newcomer = (torch.nn.Linear(3, 2) for _ in range(3))
    net = pyro.module("net", trained)
    pyro.module("other", bystander)
    gain = pyro.param("gain", torch.full((1,), 1.234))
    (net(torch.randn(1, 3)).pow(2).sum() * gain.pow(4)).backward()
    before = pyro.param("gain").detach().clone()
    torch.optim.Adam(list(trained.parameters()) + [gain.unconstrained()], lr=0.01).step()
    after = pyro.param("gain").detach().clone()
    assert not torch.equal(before, after)
    names = sorted(store.keys())
    assert names == ["gain", "net$$$bias", "net$$$weight", "other$$$bias", "other$$$weight"]

    path = str(tmp_path / "store.pt")
    store.save(path)
    store.clear()
    assert len(store) == 0
    store.load(path)
    assert sorted(store.keys()) == names and torch.equal(pyro.param("gain").detach(), after)

    def adopted():
        return torch.equal(newcomer.weight, trained.weight) and torch.equal(newcomer.bias, trained.bias)

    # registering a different module object under the same name: the store wins, and only
    # update_module_params=True puts its tensors INTO the module
    pyro.module("net", newcomer, update_module_params=False)
    assert newcomer.weight is not pyro.param("net$$$weight") and not adopted()
    pyro.module("net", newcomer, update_module_params=True)
    assert newcomer.weight is pyro.param("net$$$weight") and adopted()


# ---- scopes ---------------------------------------------------------------------------------------------------
def _declare(spec):
    for name, (value, constraint) in spec.items():
        pyro.param(name, value, constraint=constraint)


def _holds(store, spec):
    assert set(store) == set(spec)
    for name, (value, constraint) in spec.items():
        _same(pyro.param(name), value)
        assert store._constraints[name] == constraint


def test_scopes_are_separate_stores_that_can_be_reentered(store):
    simplex_row = torch.randn(1, 4).exp()
    simplex_row = simplex_row / simplex_row.sum()
    outer = {"x": (torch.randn(()), constraints.real), "z": (torch.randn(5).exp(), constraints.positive)}
    first = {"x": (torch.randn(3), constraints.real), "y": (torch.randn(2, 1).exp(), constraints.positive)}
    second = {"y": (torch.randn(2, 1).exp(), constraints.positive), "z": (simplex_row, constraints.simplex)}
    _declare(outer)
    _holds(store, outer)
    states = []
    for spec in (first, second):
        with store.scope() as state:
            assert not store                       # a new scope starts empty
            _declare(spec)
            _holds(store, spec)
        states.append(state)
        _holds(store, outer)                       # and leaves the enclosing store as it was
    for spec, state in zip((first, second), states):
        with store.scope(state) as again:
            assert again is state
            _holds(store, spec)
        _holds(store, outer)


@pytest.mark.parametrize("constraint,lower", [("positive", 0.0), ("greater_than", 0.5), ("unit_interval", None)])
def test_exp_constrained_parameters_read_through_one_kernel_each_way(monkeypatch, constraint, lower):
    """A parameter under a positive / greater-than constraint: the store hands out lower + exp(u) from
    fused.exp_lower (pa_exp_site_fwd without the Jacobian term; reference: param_store.py:99-119 applies
    transform_to(constraint) operator by operator) -- same value, same gradient, differentiable twice;
    other constraints keep the transform."""
    from torch.distributions import constraints as C, transform_to

    from pyro_amd.distributions import fused
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    calls = []
    real = fused.exp_lower
    monkeypatch.setattr(fused, "exp_lower", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    con = {"positive": C.positive, "greater_than": C.greater_than(0.5), "unit_interval": C.unit_interval}[constraint]
    pyro.clear_param_store()
    init = torch.rand(3, 4, dtype=torch.float64) * 0.4 + 0.55
    value = pyro.param("p", init, constraint=con)
    assert calls == ([1] if lower is not None else [])
    torch.testing.assert_close(value.detach(), init, rtol=1e-12, atol=0)
    u = pyro.get_param_store()._params["p"]
    ref = transform_to(con)(u)
    w = torch.randn(3, 4, dtype=torch.float64)
    g, = torch.autograd.grad((w * value * value).sum(), u, create_graph=True)
    rg, = torch.autograd.grad((w * ref * ref).sum(), u, create_graph=True)
    torch.testing.assert_close(g, rg, rtol=1e-12, atol=1e-14)
    gg, = torch.autograd.grad(g.sum(), u)
    rgg, = torch.autograd.grad(rg.sum(), u)
    torch.testing.assert_close(gg, rgg, rtol=1e-12, atol=1e-14)
    assert value.unconstrained() is u
    pyro.clear_param_store()
