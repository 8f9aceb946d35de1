"""tests/poutine/test_poutines.py of the reference restated against the drop-in API (CPU host
logic; the fused families go through the oracle backend): trace / replay / block / substitute /
condition / uncondition / infer_config / scale factors under nested sequential plates / plate
bookkeeping / decorator forms / error messages.  (queue, lift, escape, equalize belong to the
reference's search-based inference and are not part of the scoped path.)"""
import pytest
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd import poutine
from pyro_amd.distributions import Bernoulli, Normal


@pytest.fixture(autouse=True)
def _host(monkeypatch):
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    pyro.clear_param_store()


def model():
    latent1 = pyro.sample("latent1", Normal(torch.zeros(2), torch.ones(2)))
    latent2 = pyro.sample("latent2", Normal(latent1, 5 * torch.ones(2)))
    pyro.sample("obs", Normal(latent2, torch.ones(2)), obs=torch.ones(2))
    return latent1


def guide():
    loc1 = pyro.param("loc1", torch.randn(2, requires_grad=True))
    scale1 = pyro.param("scale1", torch.ones(2, requires_grad=True))
    pyro.sample("latent1", Normal(loc1, scale1))
    loc2 = pyro.param("loc2", torch.randn(2, requires_grad=True))
    scale2 = pyro.param("scale2", torch.ones(2, requires_grad=True))
    return pyro.sample("latent2", Normal(loc2, scale2))


MODEL_SITES = ["latent1", "latent2", "obs", "_INPUT", "_RETURN"]
GUIDE_SITES = ["latent1", "latent2", "loc1", "scale1", "loc2", "scale2", "_INPUT", "_RETURN"]
FULL = ["latent1", "latent2"]
PARTIAL = ["latent1"]


# ---- trace ------------------------------------------------------------------------------------------
def test_trace_full():
    guide_trace = poutine.trace(guide).get_trace()
    model_trace = poutine.trace(model).get_trace()
    assert all(name in MODEL_SITES for name in model_trace.nodes)
    for name, node in guide_trace.nodes.items():
        assert name in GUIDE_SITES and node["type"] in ("args", "return", "sample", "param")
        if node["type"] == "sample":
            assert not node["is_observed"]


def test_trace_return():
    tr = poutine.trace(model).get_trace()
    assert torch.equal(torch.as_tensor(tr.nodes["latent1"]["value"]), torch.as_tensor(tr.nodes["_RETURN"]["value"]))


def test_trace_param_only():
    tr = poutine.trace(model, param_only=True).get_trace()
    assert all(site["type"] == "param" for site in tr.nodes.values())


# ---- replay -----------------------------------------------------------------------------------------
def test_replay_full():
    guide_trace = poutine.trace(guide).get_trace()
    model_trace = poutine.trace(poutine.replay(model, trace=guide_trace)).get_trace()
    for name in FULL:
        assert torch.equal(model_trace.nodes[name]["value"], guide_trace.nodes[name]["value"])


def test_replay_full_repeat():
    model_trace = poutine.trace(model).get_trace()
    ftr = poutine.trace(poutine.replay(model, trace=model_trace))
    tr11, tr12 = ftr.get_trace(), ftr.get_trace()
    tr2 = poutine.trace(poutine.replay(model, trace=model_trace)).get_trace()
    for name in FULL:
        for a, b in ((tr11, tr12), (tr11, tr2), (model_trace, tr11), (model_trace, tr2)):
            assert torch.equal(a.nodes[name]["value"], b.nodes[name]["value"])


# ---- block ------------------------------------------------------------------------------------------
def test_block_hide_fn():
    tr = poutine.trace(poutine.block(model, hide_fn=lambda msg: "latent" in msg["name"],
                                     expose=["latent1"])).get_trace()
    assert "latent1" not in tr and "latent2" not in tr and "obs" in tr


def test_block_expose_fn():
    tr = poutine.trace(poutine.block(model, expose_fn=lambda msg: "latent" in msg["name"],
                                     hide=["latent1"])).get_trace()
    assert "latent1" in tr and "latent2" in tr and "obs" not in tr


def test_block_full():
    for fn in (model, guide):
        tr = poutine.trace(poutine.block(fn)).get_trace()
        assert all(node["type"] in ("args", "return") for node in tr.nodes.values())


def test_block_full_hide():
    for fn, sites in ((model, MODEL_SITES), (guide, GUIDE_SITES)):
        tr = poutine.trace(poutine.block(fn, hide=sites)).get_trace()
        assert all(node["type"] in ("args", "return") for node in tr.nodes.values())


def test_block_full_expose():
    for fn, sites in ((model, MODEL_SITES), (guide, GUIDE_SITES)):
        tr = poutine.trace(poutine.block(fn, expose=sites)).get_trace()
        assert all(name in tr for name in sites)


def test_block_full_hide_expose():
    with pytest.raises(AssertionError):
        poutine.block(model, hide=PARTIAL, expose=PARTIAL)()


def test_block_partial_hide_and_expose():
    for fn in (model, guide):
        tr = poutine.trace(poutine.block(fn, hide=PARTIAL)).get_trace()
        assert "latent1" not in tr and "latent2" in tr
        tr = poutine.trace(poutine.block(fn, expose=PARTIAL)).get_trace()
        assert "latent1" in tr and "latent2" not in tr


def test_block_tutorial_case():
    model_trace = poutine.trace(model).get_trace()
    guide_trace = poutine.trace(poutine.block(guide, hide_types=["observe"])).get_trace()
    assert "latent1" in model_trace and "latent1" in guide_trace
    assert "obs" in model_trace and "obs" not in guide_trace


# ---- substitute / condition / uncondition --------------------------------------------------------------
def test_substitute():
    data = {"loc1": torch.randn(2)}
    tr = poutine.trace(poutine.substitute(guide, data=data)).get_trace()
    assert tr.nodes["loc1"]["type"] == "param" and tr.nodes["loc1"]["value"] is data["loc1"]
    data1, data2 = {"loc1": torch.randn(2)}, {"loc1": torch.randn(2)}
    with poutine.trace() as tr:
        poutine.substitute(poutine.substitute(guide, data=data1), data=data2)()
    assert tr.trace.nodes["loc1"]["value"] is data2["loc1"]
    data2 = {"loc2": torch.randn(2)}
    tr = poutine.trace(poutine.substitute(poutine.substitute(guide, data=data1), data=data2)).get_trace()
    assert tr.nodes["loc1"]["value"] is data1["loc1"] and tr.nodes["loc2"]["value"] is data2["loc2"]


def test_condition():
    data = {"latent2": torch.randn(2)}
    tr = poutine.trace(poutine.condition(model, data=data)).get_trace()
    assert tr.nodes["latent2"]["type"] == "sample" and tr.nodes["latent2"]["is_observed"]
    assert tr.nodes["latent2"]["value"] is data["latent2"]


def test_condition_on_trace_data():
    tr1 = poutine.trace(poutine.block(model, expose_types=["sample"])).get_trace()
    tr2 = poutine.trace(poutine.condition(model, data=tr1)).get_trace()
    assert tr2.nodes["latent2"]["is_observed"]
    assert tr2.nodes["latent2"]["value"] is tr1.nodes["latent2"]["value"]


def test_condition_stack():
    data1, data2 = {"latent2": torch.randn(2)}, {"latent2": torch.randn(2)}
    with poutine.trace() as tr:
        poutine.condition(poutine.condition(model, data=data1), data=data2)()
    assert tr.trace.nodes["latent2"]["value"] is data2["latent2"]
    data1 = {"latent1": torch.randn(2)}
    tr = poutine.trace(poutine.condition(poutine.condition(model, data=data1), data=data2)).get_trace()
    for name, d in (("latent1", data1), ("latent2", data2)):
        assert tr.nodes[name]["is_observed"] and tr.nodes[name]["value"] is d[name]


def test_uncondition():
    unconditioned = poutine.uncondition(model)
    assert not torch.equal(poutine.trace(unconditioned).get_trace().nodes["obs"]["value"], torch.ones(2))
    assert torch.equal(poutine.trace(model).get_trace().nodes["obs"]["value"], torch.ones(2))
    recond = pyro.condition(unconditioned, {"obs": torch.ones(2)})
    assert torch.equal(poutine.trace(recond).get_trace().nodes["obs"]["value"], torch.ones(2))


# ---- infer_config, scale factors ----------------------------------------------------------------------
def test_infer_config_sample():
    def m():
        pyro.param("p", torch.zeros(1, requires_grad=True))
        pyro.sample("a", Bernoulli(torch.tensor([0.5])), infer={"enumerate": "parallel"})
        pyro.sample("b", Bernoulli(torch.tensor([0.5])))

    cfg = poutine.infer_config(m, config_fn=lambda site: {"blah": True} if site["type"] == "sample" else {})
    tr = poutine.trace(cfg).get_trace()
    assert tr.nodes["a"]["infer"] == {"enumerate": "parallel", "blah": True}
    assert tr.nodes["b"]["infer"] == {"blah": True}
    assert tr.nodes["p"]["infer"] == {}


def test_scale_factors_of_nested_sequential_plates():
    def m(batch_size_outer=2, batch_size_inner=2):
        data = [[torch.ones(1)] * 2] * 2
        loc_latent = pyro.sample("loc_latent", Normal(torch.zeros(1), torch.ones(1)))
        for i in pyro.plate("plate_outer", 2, batch_size_outer):
            for j in pyro.plate("plate_inner_%d" % i, 2, batch_size_inner):
                pyro.sample("z_%d_%d" % (i, j), Normal(loc_latent + data[i][j], torch.ones(1)))

    def factors(bo, bi):
        tr = poutine.trace(m).get_trace(batch_size_outer=bo, batch_size_inner=bi)
        return [tr.nodes[n]["scale"] for n in ["z_0_0", "z_0_1", "z_1_0", "z_1_1"] if n in tr]

    assert factors(1, 1) == [4.0]
    assert factors(2, 2) == [1.0] * 4
    assert factors(1, 2) == [2.0] * 2
    assert factors(2, 1) == [2.0] * 2


# ---- plates -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("has_rsample", [False, True])
@pytest.mark.parametrize("depth", [0, 1, 2])
def test_plate_preserves_has_rsample(has_rsample, depth):
    def g():
        loc = pyro.param("loc", torch.tensor(0.0))
        with pyro.plate_stack("plates", (2,) * depth):
            return pyro.sample("x", Normal(loc, 1.0).has_rsample_(has_rsample))

    x = g()
    assert x.dim() == depth and x.requires_grad == has_rsample


def test_plate_error_on_enter():
    from pyro_amd.poutine.runtime import _DIM_ALLOCATOR

    def m():
        with pyro.plate("foo", 0):
            pass

    assert len(_DIM_ALLOCATOR._stack) == 0
    with pytest.raises(ZeroDivisionError):
        poutine.trace(m)()
    assert len(_DIM_ALLOCATOR._stack) == 0, "stack was not cleaned on error"


# ---- decorator forms ------------------------------------------------------------------------------------
def test_decorator_interface_primitives():
    @poutine.trace
    def m():
        pyro.param("p", torch.zeros(1, requires_grad=True))
        pyro.sample("a", Bernoulli(torch.tensor([0.5])), infer={"enumerate": "parallel"})
        pyro.sample("b", Bernoulli(torch.tensor([0.5])))

    tr = m.get_trace()
    assert isinstance(tr, poutine.Trace) and tr.graph_type == "flat"

    @poutine.trace(graph_type="dense")
    def m2():
        pyro.param("p", torch.zeros(1, requires_grad=True))
        pyro.sample("a", Bernoulli(torch.tensor([0.5])), infer={"enumerate": "parallel"})
        pyro.sample("b", Bernoulli(torch.tensor([0.5])))

    tr = m2.get_trace()
    assert isinstance(tr, poutine.Trace) and tr.graph_type == "dense"
    tr2 = poutine.trace(poutine.replay(m2, trace=tr)).get_trace()
    assert torch.equal(tr2.nodes["a"]["value"], tr.nodes["a"]["value"])


def test_method_decorator_interface_condition():
    class cls_model:
        @poutine.condition(data={"b": torch.tensor(1.0)})
        def model(self, p):
            self._model(p)

        def _model(self, p):
            pyro.sample("a", Bernoulli(p))
            pyro.sample("b", Bernoulli(torch.tensor([0.5])))

    tr = poutine.trace(cls_model().model).get_trace(torch.tensor(0.5))
    assert isinstance(tr, poutine.Trace) and tr.graph_type == "flat"
    assert tr.nodes["b"]["is_observed"] and tr.nodes["b"]["value"].item() == 1.0


# ---- error messages -------------------------------------------------------------------------------------
def test_trace_error_messages_name_the_site():
    pyro.enable_validation(True)
    try:
        def m(v):
            pyro.sample("test_site", dist.Laplace(torch.tensor(0.0), 1.0).mask(True), obs=v)

        def beta(v):
            pyro.sample("test_site", dist.Uniform(torch.tensor(0.0), torch.tensor(1.0)), obs=v)

        tr = poutine.trace(beta).get_trace(torch.tensor(2.0))
        with pytest.raises(ValueError, match=r"Error while computing log_prob at site 'test_site':.*"):
            tr.compute_log_prob()
        tr = poutine.trace(beta).get_trace(torch.tensor(2.0))
        with pytest.raises(ValueError, match=r"Error while computing log_prob_sum at site 'test_site':.*"):
            tr.log_prob_sum()
        tr = poutine.trace(beta).get_trace(torch.tensor(2.0))
        with pytest.raises(ValueError, match=r"Error while computing score_parts at site 'test_site':.*"):
            tr.compute_score_parts()
    finally:
        pyro.enable_validation(False)


# ---- do (tests/poutine/test_counterfactual.py) ---------------------------------------------------------
def _item(x):
    return x.item() if isinstance(x, torch.Tensor) else x


@pytest.mark.parametrize("intervene,observe,flip", [(True, False, False), (False, True, False),
                                                    (True, True, False), (True, True, True)])
def test_counterfactual_query(intervene, observe, flip):
    sites = ["x", "y", "z", "w"]
    observations = {"x": 1.0, "y": None, "z": 1.0, "w": 1.0}
    interventions = {"x": None, "y": 0.0, "z": 2.0, "w": 1.0}

    def m():
        x = _item(pyro.sample("x", Normal(0.0, 1.0)))
        y = _item(pyro.sample("y", Normal(x, 1.0)))
        z = _item(pyro.sample("z", Normal(y, 1.0)))
        w = _item(pyro.sample("w", Normal(z, 1.0)))
        return dict(x=x, y=y, z=z, w=w)

    fn = m
    if not flip:
        if intervene:
            fn = poutine.do(fn, data=interventions)
        if observe:
            fn = poutine.condition(fn, data=observations)
    else:
        fn = poutine.do(poutine.condition(fn, data=observations), data=interventions)
    tr = poutine.trace(fn).get_trace()
    actual = tr.nodes["_RETURN"]["value"]
    for name in sites:
        node = tr.nodes[name]
        if not intervene and observe:
            if observations[name] is not None:
                assert node["is_observed"]
                assert observations[name] == actual[name] == _item(node["value"])
            if interventions[name] != observations[name]:
                assert interventions[name] != actual[name]
        elif intervene and not observe:
            assert not node["is_observed"]
            if interventions[name] is not None:
                assert interventions[name] == actual[name]
            assert observations[name] != _item(node["value"])
            assert interventions[name] != _item(node["value"])
        else:
            if observations[name] is not None:
                assert node["is_observed"] and observations[name] == _item(node["value"])
            if interventions[name] is not None:
                assert interventions[name] == actual[name]
            if interventions[name] != observations[name]:
                assert interventions[name] != _item(node["value"])


def test_do_under_a_plate_does_not_enter_the_plate_twice():
    def m(n):
        with pyro.plate("x_plate", n):
            z1 = pyro.sample("z1", Normal(torch.zeros(2), 1.0).to_event(1))
            z2 = pyro.sample("z2", Normal(torch.zeros(2), 1.0).to_event(1))
            return pyro.sample("x", Normal(z1 + z2, 1.0).to_event(1))

    fix_z1 = torch.tensor([[-6.1258, -6.1524], [-4.1513, -4.3080]])
    obs_x = torch.tensor([[-6.1258, -6.1524], [-4.1513, -4.3080]])
    fn = poutine.condition(poutine.do(m, data={"z1": fix_z1}), data={"x": obs_x})
    tr = poutine.trace(fn).get_trace(2)
    assert tr.nodes["z1"]["value"].shape == (2, 2) and not tr.nodes["z1"]["is_observed"]
    assert len(tr.nodes["z1"]["cond_indep_stack"]) == 1
    assert "z1__CF" not in tr.nodes                       # hidden from the handlers outside
    assert torch.equal(tr.nodes["x"]["fn"].base_dist.loc, fix_z1 + tr.nodes["z2"]["value"])
    tr.compute_log_prob()


# ---- lift, escape / queue, equalize (tests/poutine/test_poutines.py Lift / Queue / Equalize handler tests) ----------
def test_lift_turns_params_into_sample_sites():
    def prior_for(tensor, *args, **kwargs):
        # (positive: the lifted scale parameters are DRAWN from it and validated by Normal(loc, scale))
        return Normal(torch.zeros(tensor.shape), 1.0).sample().abs() + 0.1

    tr = poutine.trace(guide).get_trace()
    params = {"loc1", "scale1", "loc2", "scale2"}
    # a function-valued prior given directly hides the lifted sites from the handlers outside
    lifted = poutine.trace(poutine.lift(guide, prior=prior_for)).get_trace()
    assert all((name in lifted) == (name not in params) for name in tr.nodes)
    # a dict names the params to lift; the others stay params
    pyro.clear_param_store()
    lifted = poutine.trace(poutine.lift(guide, prior={"loc1": prior_for, "scale1": Normal(1.0, 0.1)})).get_trace()
    assert lifted.nodes["loc1"]["type"] == "sample" and not lifted.nodes["loc1"]["is_observed"]
    assert lifted.nodes["loc1"]["fn"] is prior_for
    assert lifted.nodes["scale1"]["type"] == "sample" and lifted.nodes["loc2"]["type"] == "param"

    def twice():
        a = pyro.param("loc")
        b = pyro.param("loc")
        assert a == b                      # the second statement sees the first draw

    poutine.trace(poutine.lift(twice, prior=Normal(0.0, 1.0)))()


def test_random_module_draws_a_copy_of_the_network():
    net = torch.nn.Linear(2, 1)
    with pytest.warns(FutureWarning):
        lifted = pyro.random_module("net", net, prior=Normal(0.0, 1.0))
    tr = poutine.trace(lifted).get_trace()
    sites = [n for n, node in tr.nodes.items() if node["type"] == "sample"]
    assert sorted(sites) == ["net$$$bias", "net$$$weight"]
    drawn = tr.nodes["_RETURN"]["value"]
    assert drawn is not net and torch.equal(drawn.weight, tr.nodes["net$$$weight"]["value"])


def test_queue_visits_every_assignment_of_the_discrete_sites():
    from queue import Queue

    def hmm():
        probs = torch.tensor([[0.8], [0.3]])
        state = torch.ones(1)
        path = []
        for t in range(3):
            state = pyro.sample("latent_{}".format(t), Bernoulli(probs[state[0].long()]))
            pyro.sample("observe_{}".format(t), Normal(state, 1.0), obs=torch.ones(1))
            path.append(int(state.item()))
        return tuple(path)

    todo = Queue()
    todo.put(poutine.Trace())
    runner = poutine.trace(poutine.queue(hmm, queue=todo))
    seen = []
    while not todo.empty():
        tr = runner.get_trace()
        assert {"_INPUT", "_RETURN", "latent_2", "observe_2"} <= set(tr.nodes)
        seen.append(tr.nodes["_RETURN"]["value"])
    assert len(seen) == 8 == len(set(seen))
    with pytest.raises(poutine.NonlocalExit):
        poutine.escape(hmm, escape_fn=lambda msg: msg["name"] == "latent_1")()


def test_equalize_ties_sites_together():
    def per_category(category):
        shift = pyro.param("{}_shift".format(category), torch.randn(1))
        std = pyro.sample("{}_std".format(category), dist.LogNormal(0.0, 1.0))
        return pyro.sample("{}_values".format(category), Normal(shift, std))

    def program():
        return {c: per_category(c) for c in ("dog", "cat")}

    tr = poutine.trace(poutine.equalize(program, ".+_std")).get_trace()
    assert torch.equal(tr.nodes["dog_std"]["value"], tr.nodes["cat_std"]["value"])
    assert tr.nodes["cat_std"]["is_observed"] and tr.nodes["cat_std"]["infer"] == {"_deterministic": True}
    assert not tr.nodes["dog_std"]["is_observed"]
    tied = poutine.equalize(poutine.equalize(program, ".+_std"), ".+_shift", "param")
    tr = poutine.trace(tied).get_trace()
    assert torch.equal(tr.nodes["dog_shift"]["value"], tr.nodes["cat_shift"]["value"])
    # keep_dist=True: conditioning on equality, the later site keeps scoring under its own prior
    def two():
        pyro.sample("x", Normal(0.0, 1.0))
        pyro.sample("y", Normal(5.0, 3.0))

    tr = poutine.trace(poutine.equalize(two, ["x", "y"], keep_dist=True)).get_trace()
    x = tr.nodes["x"]["value"]
    assert torch.equal(tr.nodes["y"]["value"], x) and tr.nodes["y"]["is_observed"]
    expected = Normal(0.0, 1.0).log_prob(x) + Normal(5.0, 3.0).log_prob(x)
    assert abs(float(tr.log_prob_sum()) - float(expected)) < 1e-5


def test_broadcast_is_the_identity_and_block_messengers_mutes():
    from pyro_amd.poutine.runtime import block_messengers

    @poutine.broadcast
    def program():
        with pyro.plate("p", 3):
            return pyro.sample("x", Normal(0.0, 1.0))

    assert program().shape == (3,)
    with poutine.trace() as outer:
        with poutine.scale(scale=2.0) as scaling:
            with block_messengers(lambda m: m is scaling) as muted:
                assert muted == [scaling]
                pyro.sample("a", Normal(0.0, 1.0))
            pyro.sample("b", Normal(0.0, 1.0))
    assert outer.trace.nodes["a"]["scale"] == 1.0 and outer.trace.nodes["b"]["scale"] == 2.0
