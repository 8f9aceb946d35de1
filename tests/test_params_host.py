"""tests/params/test_param.py of the reference restated (CPU): the dict interface of the parameter
store, save / load with pyro.module, scopes."""
import os
import tempfile

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import constraints

import pyro_amd as pyro


def _eq(a, b):
    np.testing.assert_allclose(torch.as_tensor(a).detach().numpy(), torch.as_tensor(b).detach().numpy(),
                               rtol=1e-6, atol=1e-6)


def test_save_and_load():
    pyro.clear_param_store()
    lin1, lin2, lin3 = nn.Linear(3, 2), nn.Linear(3, 2), nn.Linear(3, 2)
    lin = pyro.module("mymodule", lin1)
    pyro.module("mymodule2", lin2)
    x = torch.randn(1, 3)
    myparam = pyro.param("myparam", 1.234 * torch.ones(1))
    cost = torch.sum(torch.pow(lin(x), 2.0)) * torch.pow(myparam, 4.0)
    cost.backward()
    leaf = myparam.unconstrained() if hasattr(myparam, "unconstrained") else myparam
    optim = torch.optim.Adam(list(lin1.parameters()) + [leaf], lr=0.01)
    stale = pyro.param("myparam").detach().numpy().copy()
    optim.step()
    fresh = pyro.param("myparam").detach().numpy().copy()
    store = pyro.get_param_store()
    names = sorted(store.keys())
    assert len(names) == 5
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "paramstore.unittest.out")
        store.save(f)
        pyro.clear_param_store()
        assert len(list(store.keys())) == 0
        store.load(f)

    def modules_are_equal():
        return bool((lin3.weight == lin1.weight).all() and (lin3.bias == lin1.bias).all())

    assert not modules_are_equal()
    pyro.module("mymodule", lin3, update_module_params=False)
    assert id(lin3.weight) != id(pyro.param("mymodule$$$weight"))
    assert not modules_are_equal()
    pyro.module("mymodule", lin3, update_module_params=True)
    assert id(lin3.weight) == id(pyro.param("mymodule$$$weight"))
    assert modules_are_equal()
    now = pyro.param("myparam").detach().numpy()
    assert stale != now and fresh == now
    assert sorted(store.keys()) == names


def test_dict_interface():
    ps = pyro.get_param_store()
    ps.clear()
    assert not ps and len(ps) == 0 and "x" not in ps
    assert list(ps.items()) == [] and list(ps.keys()) == [] and list(ps.values()) == []
    ps["x"] = torch.zeros(1, 2, 3)
    assert ps and len(ps) == 1 and "x" in ps and "y" not in ps
    assert list(ps.keys()) == ["x"] and [k for k, v in ps.items()] == ["x"]
    assert len(list(ps.values())) == 1 and ps["x"].shape == (1, 2, 3)
    _eq(ps.setdefault("x", torch.ones(1, 2, 3)), torch.zeros(1, 2, 3))
    assert ps["x"].unconstrained() is ps["x"]
    ps.setdefault("y", torch.ones(4, 5), constraint=constraints.positive)
    assert len(ps) == 2 and sorted(ps.keys()) == ["x", "y"]
    assert ps["y"].shape == (4, 5)
    _eq(ps.setdefault("y", torch.zeros(4, 5)), torch.ones(4, 5))
    _eq(ps["y"].unconstrained(), torch.zeros(4, 5))
    del ps["x"]
    assert len(ps) == 1 and "x" not in ps and "y" in ps and list(ps.keys()) == ["y"]
    _eq(ps["y"].unconstrained(), torch.zeros(4, 5))
    del ps["y"]
    assert not ps and len(ps) == 0 and list(ps.keys()) == []


def test_scope():
    x0, z0 = torch.randn(()), torch.randn(5).exp()
    x1, y1 = torch.randn(3), torch.randn(2, 1).exp()
    y2, z2 = torch.randn(2, 1).exp(), torch.randn(1, 4).exp()
    z2 /= z2.sum()
    table = {"z0": constraints.positive, "y1": constraints.positive, "y2": constraints.positive,
             "z2": constraints.simplex}
    ps = pyro.get_param_store()
    ps.clear()

    def check(name):
        assert ps._constraints[name[:1]] == table.get(name, constraints.real)

    def base():
        assert set(ps) == {"x", "z"}
        _eq(pyro.param("x"), x0); _eq(pyro.param("z"), z0)
        check("x0"); check("z0")

    assert not ps
    pyro.param("x", x0)
    pyro.param("z", z0, constraint=constraints.positive)
    base()
    with ps.scope() as scope1:
        assert not ps
        pyro.param("x", x1)
        pyro.param("y", y1, constraint=constraints.positive)
        assert set(ps) == {"x", "y"}
        _eq(pyro.param("x"), x1); _eq(pyro.param("y"), y1)
        check("x1"); check("y1")
    base()
    with ps.scope() as scope2:
        assert not ps
        pyro.param("y", y2, constraint=constraints.positive)
        pyro.param("z", z2, constraint=constraints.simplex)
        assert set(ps) == {"y", "z"}
        _eq(pyro.param("y"), y2); _eq(pyro.param("z"), z2)
        check("y2"); check("z2")
    base()
    with ps.scope(scope1) as s:
        assert s is scope1 and set(ps) == {"x", "y"}
        _eq(pyro.param("x"), x1); _eq(pyro.param("y"), y1)
    base()
    with ps.scope(scope2) as s:
        assert s is scope2 and set(ps) == {"y", "z"}
        _eq(pyro.param("y"), y2); _eq(pyro.param("z"), z2)
    base()
