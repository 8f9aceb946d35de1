"""Error / warning behaviour of the estimators under ``pyro.enable_validation`` -- a restatement of the
parts of tests/infer/test_valid_models.py (and tests/test_settings.py, tests/test_util.py,
tests/ops/test_provenance.py) where running the reference's own files against this package
(tests/refsuite) found differences.  CPU host logic; kernels answered by the oracle backend."""
import warnings

import pytest
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd import poutine
from pyro_amd.infer import (SVI, Trace_ELBO, TraceEnum_ELBO, TraceGraph_ELBO, TraceMeanField_ELBO,
                            config_enumerate)
from pyro_amd.optim import Adam

ELBOS = [Trace_ELBO, TraceGraph_ELBO, TraceEnum_ELBO]


@pytest.fixture(autouse=True)
def _host(monkeypatch):
    from tests import oracle_backend
    oracle_backend.install(monkeypatch)
    pyro.clear_param_store()
    pyro.enable_validation(True)
    yield
    pyro.enable_validation(False)


def assert_ok(model, guide, elbo, **kwargs):
    """Inference runs without warnings or errors, and the three entry points agree."""
    pyro.clear_param_store()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        SVI(model, guide, Adam({"lr": 1e-6}), elbo).step(**kwargs)
        pyro.set_rng_seed(0)
        loss = elbo.loss(model, guide, **kwargs)
        pyro.set_rng_seed(0)
        diff = elbo.differentiable_loss(model, guide, **kwargs)
        assert abs(torch.as_tensor(diff).detach().item() - loss) < 0.01
        pyro.set_rng_seed(0)
        assert abs(elbo.loss_and_grads(model, guide, **kwargs) - loss) < 0.01


def assert_error(model, guide, elbo, match=None):
    pyro.clear_param_store()
    with pytest.raises((NotImplementedError, UserWarning, KeyError, ValueError, RuntimeError), match=match):
        SVI(model, guide, Adam({"lr": 1e-6}), elbo).step()


def assert_warning(model, guide, elbo):
    pyro.clear_param_store()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        SVI(model, guide, Adam({"lr": 1e-6}), elbo).step()
    assert len(w), "No warnings were raised"


# ---- model / guide structure -------------------------------------------------------------------------------
@pytest.mark.parametrize("Elbo", ELBOS)
@pytest.mark.parametrize("strict", [True, False])
def test_nonempty_model_empty_guide(Elbo, strict):
    def model():
        pyro.sample("x", dist.Normal(torch.zeros(2), 1.0).to_event(1), obs=torch.zeros(2))

    elbo = Elbo(strict_enumeration_warning=strict)
    if strict and Elbo is TraceEnum_ELBO:
        assert_warning(model, lambda: None, elbo)        # nothing is enumerated: say so
    else:
        assert_ok(model, lambda: None, elbo)


@pytest.mark.parametrize("Elbo", ELBOS)
def test_empty_model_empty_guide(Elbo):
    assert_ok(lambda: None, lambda: None, Elbo(strict_enumeration_warning=False))


@pytest.mark.parametrize("Elbo", ELBOS)
def test_variable_clash_in_model(Elbo):
    def model():
        p = torch.tensor(0.5)
        pyro.sample("x", dist.Bernoulli(p))
        pyro.sample("x", dist.Bernoulli(p))

    def guide():
        pyro.sample("x", dist.Bernoulli(pyro.param("p", torch.tensor(0.5))))

    assert_error(model, guide, Elbo(), match="Multiple sample sites named")


@pytest.mark.parametrize("Elbo", ELBOS)
def test_model_guide_dim_mismatch(Elbo):
    def model():
        pyro.sample("x", dist.Normal(torch.zeros(2), torch.ones(2)).to_event(1))

    def guide():
        pyro.sample("x", dist.Normal(pyro.param("loc", torch.zeros(2, 1)), torch.ones(2, 1)).to_event(2))

    assert_error(model, guide, Elbo(strict_enumeration_warning=False),
                 match="invalid log_prob shape|Model and guide event_dims disagree")


@pytest.mark.parametrize("Elbo", ELBOS)
def test_model_guide_shape_mismatch(Elbo):
    def model():
        pyro.sample("x", dist.Normal(torch.zeros(1), torch.ones(1)).to_event(1))

    def guide():
        pyro.sample("x", dist.Normal(pyro.param("loc", torch.zeros(2)), torch.ones(2)).to_event(1))

    assert_error(model, guide, Elbo(strict_enumeration_warning=False),
                 match="Model and guide shapes disagree")


@pytest.mark.parametrize("Elbo", ELBOS)
def test_variable_in_guide_not_model_warns_unless_auxiliary(Elbo):
    def model():
        pyro.sample("x", dist.Bernoulli(torch.tensor(0.5)))

    def guide(aux):
        p = pyro.param("p", torch.tensor(0.5))
        pyro.sample("x", dist.Bernoulli(p))
        pyro.sample("y", dist.Normal(p, 1.0), infer={"is_auxiliary": True} if aux else {})

    assert_warning(model, lambda: guide(False), Elbo(strict_enumeration_warning=False))


@pytest.mark.parametrize("Elbo", ELBOS)
def test_variable_in_model_not_guide_warns(Elbo):
    def model():
        loc = torch.zeros(2)
        pyro.sample("x", dist.Normal(loc, 1.0).to_event(1))
        pyro.sample("y", dist.Normal(loc, 1.0).to_event(1))

    def guide():
        pyro.sample("x", dist.Normal(pyro.param("loc", torch.zeros(2)), 1.0).to_event(1))

    assert_warning(model, guide, Elbo(strict_enumeration_warning=False))


# ---- plates ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Elbo", ELBOS)
def test_batched_site_outside_of_a_plate_is_an_error(Elbo):
    def model():
        pyro.sample("x", dist.Normal(torch.zeros(3), 1.0))          # no plate, no to_event

    def guide():
        pyro.sample("x", dist.Normal(pyro.param("loc", torch.zeros(3)), 1.0))

    assert_error(model, guide, Elbo(strict_enumeration_warning=False), match="invalid log_prob shape")


@pytest.mark.parametrize("Elbo", ELBOS)
def test_plate_stack_overflow(Elbo):
    def model():
        with pyro.plate("a", 2), pyro.plate("b", 3):
            pyro.sample("x", dist.Normal(0.0, 1.0))

    def guide():
        with pyro.plate("a", 2), pyro.plate("b", 3):
            pyro.sample("x", dist.Normal(pyro.param("loc", torch.tensor(0.0)), 1.0))

    assert_error(model, guide, Elbo(max_plate_nesting=1, strict_enumeration_warning=False),
                 match="plate stack overflow")


def test_nested_plates_on_one_dim_collide():
    def model():
        with pyro.plate("a", 2, dim=-1):
            with pyro.plate("b", 2, dim=-1):
                pyro.sample("x", dist.Normal(0.0, 1.0))

    with pytest.raises(ValueError, match="collide at dim=-1"):
        model()


@pytest.mark.parametrize("Elbo", ELBOS)
def test_plate_without_size(Elbo):
    def model():
        with pyro.plate("plate") as ind:
            assert ind is None
            pyro.sample("x", dist.Normal(torch.zeros(5), 1.0))

    def guide():
        with pyro.plate("plate"):
            pyro.sample("x", dist.Normal(pyro.param("loc", torch.zeros(5)), 1.0))

    assert_ok(model, guide, Elbo(strict_enumeration_warning=False))


@pytest.mark.parametrize("Elbo", ELBOS)
@pytest.mark.parametrize("subsample_size", [None, 5], ids=["full", "subsample"])
def test_subsample_primitive(Elbo, subsample_size):
    data = torch.randn(10, 3)

    def model():
        with pyro.plate("plate", 10, subsample_size):
            batch = pyro.subsample(data, event_dim=1)
            assert batch.shape == (10 if subsample_size is None else 5, 3)
            z = pyro.sample("z", dist.Normal(0.0, 1.0))
            pyro.sample("x", dist.Normal(z.unsqueeze(-1), 1.0).to_event(1), obs=batch)

    def guide():
        loc = pyro.param("loc", torch.zeros(10), event_dim=0)
        with pyro.plate("plate", 10, subsample_size):
            loc = pyro.subsample(loc, event_dim=0)
            pyro.sample("z", dist.Normal(loc, 1.0))

    assert_ok(model, guide, Elbo(strict_enumeration_warning=False))
    assert pyro.subsample(data, event_dim=1) is data             # outside of plates: untouched


@pytest.mark.parametrize("Elbo", ELBOS)
@pytest.mark.parametrize("subsample_size", [None, 5], ids=["full", "subsample"])
@pytest.mark.parametrize("shape", [(10, 2), (9,)], ids=["wrong_dim", "wrong_size"])
def test_param_with_the_wrong_plate_size_is_an_error(Elbo, subsample_size, shape):
    def model():
        with pyro.plate("plate", 10, subsample_size):
            pyro.sample("x", dist.Normal(0.0, 1.0))

    def guide():
        with pyro.plate("plate", 10, subsample_size):
            loc = pyro.param("loc", torch.zeros(shape), event_dim=0)
            pyro.sample("x", dist.Normal(loc, 1.0))

    assert_error(model, guide, Elbo(strict_enumeration_warning=False),
                 match="invalid shape of pyro.param|invalid log_prob shape")


def test_subsampled_param_records_its_rows():
    pyro.set_rng_seed(0)
    with pyro.plate("plate", 10, 4) as ind:
        loc = pyro.param("loc", torch.arange(10.0), event_dim=0)
    assert torch.equal(loc, torch.arange(10.0)[ind])
    leaf = pyro.param("loc").unconstrained()
    assert torch.equal(leaf._pyro_subsample[-1], ind)


def test_block_plate():
    from pyro_amd.poutine.plate_messenger import block_plate

    def model():
        with pyro.plate("plate", 3):
            with block_plate("plate"):
                loc = pyro.sample("loc", dist.Normal(0.0, 1.0))
            assert loc.shape == ()
            x = pyro.sample("x", dist.Normal(loc, 1.0))
            assert x.shape == (3,)
            with block_plate(dim=-1):
                assert pyro.sample("y", dist.Normal(0.0, 1.0)).shape == ()

    model()
    with pytest.raises(ValueError, match="Exactly one of name,dim"):
        with block_plate():
            pass
    with pytest.raises(ValueError, match="block_plate matched 0 messengers"):
        with block_plate("nope"):
            pass
    with block_plate("nope", strict=False):
        pass


# ---- enumeration -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("Elbo", [Trace_ELBO, TraceGraph_ELBO])
def test_enumeration_asked_of_an_estimator_that_cannot_warns(Elbo):
    def model():
        pyro.sample("x", dist.Bernoulli(torch.tensor(0.5)))

    @config_enumerate
    def guide():
        pyro.sample("x", dist.Bernoulli(pyro.param("p", torch.tensor(0.5))))

    assert_warning(model, guide, Elbo())


def test_enum_sequential_in_model_error():
    def model():
        pyro.sample("a", dist.Bernoulli(pyro.param("p", torch.tensor(0.25))),
                    infer={"enumerate": "sequential"})

    assert_error(model, lambda: None, TraceEnum_ELBO(max_plate_nesting=0),
                 match="At site .*, model-side sequential enumeration is not implemented")


def test_enumerated_site_after_a_site_of_a_smaller_plate_context_warns():
    # "z" (outside the plate) comes after "y" (enumerated inside it): possibly invalid dependency
    def model():
        with pyro.plate("plate", 2):
            pyro.sample("y", dist.Bernoulli(torch.tensor(0.5)))
        pyro.sample("z", dist.Bernoulli(torch.tensor(0.5)))

    @config_enumerate
    def guide():
        p = pyro.param("p", torch.tensor(0.5))
        with pyro.plate("plate", 2):
            pyro.sample("y", dist.Bernoulli(p))
        pyro.sample("z", dist.Bernoulli(p))

    pyro.clear_param_store()
    with pytest.warns(RuntimeWarning, match='Expected site "z" to precede sites "y"'):
        TraceEnum_ELBO(max_plate_nesting=1).loss(model, guide)


def test_local_sampling_warnings():
    def model():
        pyro.sample("x", dist.Bernoulli(torch.tensor(0.5)), infer={"enumerate": "parallel",
                                                                   "num_samples": 5})

    with pytest.warns(RuntimeWarning, match="multiply sampled in model"):
        TraceEnum_ELBO(max_plate_nesting=0).loss(model, lambda: None)


# ---- pyro.factor in the guide --------------------------------------------------------------------------------
@pytest.mark.parametrize("Elbo", ELBOS + [TraceMeanField_ELBO])
@pytest.mark.parametrize("has_rsample", [False, True])
def test_factor_in_guide(Elbo, has_rsample):
    def guide():
        pyro.factor("f", torch.tensor(0.0), has_rsample=has_rsample)

    assert_ok(lambda: None, guide, Elbo(strict_enumeration_warning=False))


@pytest.mark.parametrize("Elbo", ELBOS)
def test_factor_in_guide_must_say_has_rsample(Elbo):
    def guide():
        pyro.factor("f", torch.tensor(0.0))

    assert_error(lambda: None, guide, Elbo(strict_enumeration_warning=False),
                 match="missing specification of has_rsample")


# ---- settings, warn_if_nan / warn_if_inf, provenance containers ---------------------------------------------------
_TEST_SETTING = 0.1


def test_settings_registry():
    from pyro_amd import settings
    all_ = settings.get()
    assert all(isinstance(k, str) for k in all_)
    for alias in ("validate_distributions_pyro", "validate_distributions_torch", "validate_poutine",
                  "validate_infer"):
        assert settings.get(alias) is True
    with pytest.raises(KeyError):
        settings.get("test_setting")

    @settings.register("test_setting", __name__, "_TEST_SETTING")
    def _validate(value):
        assert isinstance(value, float) and 0 < value

    assert settings.get("test_setting") == 0.1
    settings.set(test_setting=0.2)
    assert settings.get("test_setting") == 0.2
    with pytest.raises(AssertionError):
        settings.set(test_setting=-0.1)
    with settings.context(test_setting=0.3):
        assert settings.get("test_setting") == 0.3
    assert settings.get("test_setting") == 0.2

    @settings.context(test_setting=0.4)
    def fn():
        assert settings.get("test_setting") == 0.4

    fn()
    assert settings.get("test_setting") == 0.2
    with settings.context(validate_infer=False):
        from pyro_amd.infer.util import is_validation_enabled
        assert not is_validation_enabled()


def test_warn_if_nan_and_inf_also_watch_the_gradient():
    from pyro_amd import util
    msg = "example message"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        x = float("inf")
        assert util.warn_if_nan(x, msg) is x and len(w) == 0
        util.warn_if_nan(float("nan"), msg)
        assert len(w) == 1 and msg in str(w[-1].message)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        x = torch.ones(2, requires_grad=True)
        util.warn_if_nan(x, msg)
        x.sum().backward(torch.tensor(float("nan")))
        assert len(w) == 1 and "backward " + msg in str(w[-1].message)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        x = torch.ones(2, requires_grad=True)
        util.warn_if_inf(x, msg, allow_posinf=True)
        y = x.sum()
        y.backward(torch.tensor(float("inf")), retain_graph=True)
        assert len(w) == 0
        y.backward(torch.tensor(-float("inf")))
        assert len(w) == 1 and msg in str(w[-1].message)


@pytest.mark.parametrize("make", [
    lambda: torch.tensor([1, 2, 3]),
    lambda: frozenset([torch.tensor([0, 1]), torch.tensor([2, 3])]),
    lambda: [torch.tensor([0, 1]), torch.tensor([2, 3])],
    lambda: (torch.tensor([0, 1]), torch.tensor([2, 3])),
    lambda: {"a": torch.tensor([0, 1]), "b": [torch.tensor([2, 3]), torch.tensor([4, 5])]},
])
def test_track_provenance_of_containers(make):
    from pyro_amd.ops.provenance import ProvenanceTensor, get_provenance, track_provenance
    x = make()
    assert get_provenance(x) == frozenset()
    tagged = track_provenance(x, frozenset("x"))
    assert type(tagged) is type(x) or isinstance(tagged, torch.Tensor)
    assert get_provenance(tagged) == frozenset("x")
    y = ProvenanceTensor(torch.tensor([1.0, 2.0]), frozenset(["y"]))
    assert get_provenance(y + 1) == frozenset(["y"])
