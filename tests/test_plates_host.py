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


LOC, SCALE = torch.zeros(20), torch.ones(20)


def _vectorised(subsample_size):
    with pyro.plate("plate", 20, subsample_size) as batch:
        pyro.sample("x", Normal(LOC[batch], SCALE[batch]))
        return batch.tolist()


def _sequential(subsample_size):
    picked = []
    for i in pyro.plate("plate", 20, subsample_size):
        pyro.sample("x_{}".format(i), Normal(LOC[i], SCALE[i]))
        picked.append(int(i))
    return picked


def _nested_sequential(subsample_size):
    inner = pyro.plate("inner", 20, 5)
    picked = []
    for i in pyro.plate("outer", 20, subsample_size):
        row = []
        for j in inner:
            pyro.sample("x_{}_{}".format(i, j), Normal(LOC[i] + LOC[j], SCALE[i] + SCALE[j]))
            row.append(int(j))
        picked.append(row)
    return picked


PROGRAMS = {"vectorised": _vectorised, "sequential": _sequential, "nested": _nested_sequential}
plate_model, iplate_model = _vectorised, _sequential


@pytest.mark.parametrize("subsample_size", [5, 20])
@pytest.mark.parametrize("kind", sorted(PROGRAMS))
def test_every_site_in_a_plate_knows_its_frames(kind, subsample_size):
    trace = poutine.trace(PROGRAMS[kind]).get_trace(subsample_size)
    inside = [node for name, node in trace.nodes.items() if name.startswith("x")]
    assert inside and all(node["cond_indep_stack"] for node in inside)


@pytest.mark.parametrize("subsample_size", [5, 20])
@pytest.mark.parametrize("kind", sorted(PROGRAMS))
def test_replay_reuses_the_subsample(kind, subsample_size):
    program = PROGRAMS[kind]
    pyro.set_rng_seed(0)
    traced = poutine.trace(program)
    first = traced(subsample_size)
    assert poutine.replay(program, trace=traced.trace)(subsample_size) == first
    if subsample_size < 20:                       # a real subsample: a fresh run draws another one
        assert traced(subsample_size) != first


@pytest.mark.parametrize("sequential", [False, True])
def test_a_given_subsample_is_used_as_is(sequential):
    wanted = [1, 3, 5, 7]

    def program(subsample):
        if sequential:
            return [int(i) for i in pyro.plate("plate", 20, subsample=subsample)]
        with pyro.plate("plate", 20, subsample=subsample) as batch:
            return [int(i) for i in batch]

    assert program(wanted) == wanted == poutine.trace(program)(wanted)


# sizes the guide ran with -> sizes the model may run with when replayed against it
_COMPATIBLE = [(20, 20), (None, 20), (5, 5), (20, None), (5, None), (None, None)]
_CLASHING = [(5, 20), (20, 5), (None, 5)]


@pytest.mark.parametrize("kind", ["vectorised", "sequential"])
@pytest.mark.parametrize("guide_size,model_size", _COMPATIBLE + _CLASHING)
def test_model_guide_subsample_size_mismatch(kind, guide_size, model_size):
    program = PROGRAMS[kind]
    traced = poutine.trace(program)
    drawn = traced(guide_size)
    replayed = poutine.replay(program, trace=traced.trace)
    if (guide_size, model_size) in _CLASHING:
        with pytest.raises(ValueError):
            replayed(model_size)
    else:
        assert replayed(model_size) == drawn


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
