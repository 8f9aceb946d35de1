"""tests/poutine/test_mapdata.py, test_runtime.py and test_trace_struct.py of the reference restated:
subsampling plates (vectorised and sequential) under trace / replay, custom subsamples, the
model/guide subsample-size mismatch error, get_mask / get_plates, and the trace's DAG helpers."""
import itertools

import pytest
import torch

import pyro_amd as pyro
from pyro_amd import poutine
from pyro_amd.distributions import Normal


@pytest.fixture(autouse=True)
def _host(monkeypatch):
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    pyro.clear_param_store()


def test_nested_sequential_plates_scale():
    means = [torch.randn(2) for _ in range(8)]
    stds = [torch.randn(2).abs() for _ in range(6)]

    def model(means, stds):
        a_plate = pyro.plate("a", len(means), 2)
        b_plate = pyro.plate("b", len(stds), 3)
        return [[pyro.sample("x_{}{}".format(i, j), Normal(means[i], stds[j])) for j in b_plate]
                for i in a_plate]

    xs = model(means, stds)
    assert len(xs) == 2 and len(xs[0]) == 3
    tr = poutine.trace(model).get_trace(means, stds)
    for name, node in tr.nodes.items():
        if node["type"] == "sample" and name.startswith("x_"):
            assert node["scale"] == 4.0 * 2.0


def plate_model(subsample_size):
    loc, scale = torch.zeros(20), torch.ones(20)
    with pyro.plate("plate", 20, subsample_size) as batch:
        pyro.sample("x", Normal(loc[batch], scale[batch]))
        return [int(i) for i in batch]


def iplate_model(subsample_size):
    loc, scale = torch.zeros(20), torch.ones(20)
    result = []
    for i in pyro.plate("plate", 20, subsample_size):
        pyro.sample("x_{}".format(i), Normal(loc[i], scale[i]))
        result.append(int(i))
    return result


def nested_iplate_model(subsample_size):
    loc, scale = torch.zeros(20), torch.ones(20)
    result = []
    inner = pyro.plate("inner", 20, 5)
    for i in pyro.plate("outer", 20, subsample_size):
        result.append([])
        for j in inner:
            pyro.sample("x_{}_{}".format(i, j), Normal(loc[i] + loc[j], scale[i] + scale[j]))
            result[-1].append(int(j))
    return result


MODELS = [plate_model, iplate_model, nested_iplate_model]


@pytest.mark.parametrize("subsample_size", [5, 20])
@pytest.mark.parametrize("model", MODELS)
def test_cond_indep_stack(model, subsample_size):
    tr = poutine.trace(model).get_trace(subsample_size)
    for name, node in tr.nodes.items():
        if name.startswith("x"):
            assert node["cond_indep_stack"], name


@pytest.mark.parametrize("subsample_size", [5, 20])
@pytest.mark.parametrize("model", MODELS)
def test_replay_reuses_the_subsample(model, subsample_size):
    pyro.set_rng_seed(0)
    traced = poutine.trace(model)
    original = traced(subsample_size)
    assert poutine.replay(model, trace=traced.trace)(subsample_size) == original
    if subsample_size < 20:
        assert traced(subsample_size) != original


@pytest.mark.parametrize("sequential", [False, True])
def test_custom_subsample(sequential):
    def model(subsample):
        if sequential:
            return [int(i) for i in pyro.plate("plate", 20, subsample=subsample)]
        with pyro.plate("plate", 20, subsample=subsample) as batch:
            return [int(i) for i in batch]

    subsample = [1, 3, 5, 7]
    assert model(subsample) == subsample
    assert poutine.trace(model)(subsample) == subsample


@pytest.mark.parametrize("model", [plate_model, iplate_model])
@pytest.mark.parametrize("behavior,model_size,guide_size", [
    ("error", 20, 5), ("error", 5, 20), ("error", 5, None), ("ok", 20, 20), ("ok", 20, None),
    ("ok", 5, 5), ("ok", None, 20), ("ok", None, 5), ("ok", None, None)])
def test_model_guide_subsample_size_mismatch(behavior, model_size, guide_size, model):
    traced = poutine.trace(model)
    expected = traced(guide_size)
    if behavior == "ok":
        assert poutine.replay(model, trace=traced.trace)(model_size) == expected
    else:
        with pytest.raises(ValueError):
            poutine.replay(model, trace=traced.trace)(model_size)


# ---- runtime queries --------------------------------------------------------------------------------------
def test_get_mask():
    from pyro_amd.poutine.runtime import get_mask
    assert get_mask() is None
    with poutine.mask(mask=True):
        assert get_mask() is True
    with poutine.mask(mask=False):
        assert get_mask() is False
    with pyro.plate("i", 2, dim=-1):
        mask1 = torch.tensor([False, True, True])
        mask2 = torch.tensor([True, True, False])
        with poutine.mask(mask=mask1):
            assert torch.equal(get_mask(), mask1)
            with poutine.mask(mask=mask2):
                assert torch.equal(get_mask(), mask1 & mask2)


def test_get_plates():
    from pyro_amd.poutine.runtime import get_plates

    def names():
        plates = get_plates()
        assert isinstance(plates, tuple)
        return {f.name for f in plates}

    assert names() == set()
    with pyro.plate("foo", 5):
        assert names() == {"foo"}
        with pyro.plate("bar", 3):
            assert names() == {"foo", "bar"}


# ---- the trace as a DAG -----------------------------------------------------------------------------------
EDGE_SETS = [[(1, 2), (1, 3), (3, 4), (3, 5), (4, 6), (4, 7)],
             [(1, 2), (3, 5), (1, 4), (1, 3), (5, 6), (6, 7)]]
PERMS = [perm for edges in EDGE_SETS for perm in itertools.islice(itertools.permutations(edges), 0, 720, 37)]


@pytest.mark.parametrize("edges", PERMS)
def test_topological_sort_and_removal(edges):
    tr = poutine.Trace()
    for a, b in edges:
        tr.add_edge(a, b)
    order = tr.topological_sort()
    expected = set().union(*edges)
    assert len(order) == len(expected) and set(order) == expected
    rank = {n: r for r, n in enumerate(order)}
    assert all(rank[a] < rank[b] for a, b in edges)
    # removing in reverse topological order keeps the rest reachable from the root
    while order:
        assert len(list(tr._dfs(1, set()))) == len(order)
        tr.remove_node(order.pop())
