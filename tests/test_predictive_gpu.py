"""tests/infer/test_predictive.py of the reference on the MI355X (tests/predictive_kat_cases.py)."""
import pytest

from tests import predictive_kat_cases as pk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("parallel", [False, True])
def test_posterior_predictive_svi_manual_guide(gpu, parallel):
    pk.run_manual_guide(gpu, parallel)


@pytest.mark.parametrize("parallel", [False, True])
def test_posterior_predictive_svi_auto_delta_guide(gpu, parallel):
    pk.run_auto_delta(gpu, parallel)


@pytest.mark.parametrize("return_trace", [False, True])
def test_posterior_predictive_svi_auto_diag_normal_guide(gpu, return_trace):
    pk.run_auto_diag_normal(gpu, return_trace)


def test_posterior_predictive_svi_one_hot(gpu):
    pk.run_one_hot(gpu)


@pytest.mark.parametrize("parallel", [False, True])
def test_shapes(gpu, parallel):
    pk.run_shapes(gpu, parallel)


@pytest.mark.parametrize("with_plate", [True, False])
@pytest.mark.parametrize("event_shape", [(), (2,)])
def test_deterministic(gpu, with_plate, event_shape):
    pk.run_deterministic(gpu, with_plate, event_shape)


def test_get_mask_optimization(gpu):
    pk.run_get_mask_optimization(gpu)
