"""Reference distribution KATs (tests/distributions/test_delta.py, test_mask.py, test_categorical.py)
on the MI355X: Delta, .mask over the FUSED Bernoulli / Normal families, Categorical."""
import pytest
import torch

import pyro_amd.distributions as dist
from tests import dist_kat_cases as dk

pytestmark = pytest.mark.gpu


def test_delta(gpu):
    dk.run_delta(gpu)


@pytest.mark.parametrize("batch_dim,event_dim", [(b, e) for b in range(4) for e in range(1 + b)])
@pytest.mark.parametrize("has_log_density", [False, True])
def test_delta_shapes(gpu, batch_dim, event_dim, has_log_density):
    dk.run_delta_shapes(gpu, batch_dim, event_dim, has_log_density)


@pytest.mark.parametrize("batch_dim,mask_dim", [(b, m) for b in range(3) for m in range(1 + b)])
@pytest.mark.parametrize("event_dim", [0, 1, 2])
def test_mask_over_fused_bernoulli(gpu, batch_dim, event_dim, mask_dim):
    dk.run_mask(gpu, lambda shape: dist.Bernoulli(torch.tensor(0.1, device=gpu)).expand_by(shape),
                batch_dim, event_dim, mask_dim)


@pytest.mark.parametrize("mask", [False, True, torch.tensor(False), torch.tensor(True)])
def test_mask_type_fused_normal(gpu, mask):
    dk.run_mask_type(gpu, dist.Normal, mask)


@pytest.mark.parametrize("event_shape", [(), (4,)])
@pytest.mark.parametrize("dist_shape", [(), (3,), (2, 1), (2, 3)])
@pytest.mark.parametrize("mask_shape", [(), (3,), (2, 1), (2, 3)])
def test_mask_broadcast(gpu, event_shape, dist_shape, mask_shape):
    dk.run_mask_broadcast(gpu, dist.Normal, event_shape, dist_shape, mask_shape)


def test_mask_kl_divergence(gpu):
    dk.run_mask_kl(gpu, dist.Normal)


@pytest.mark.parametrize("p_mask", [False, True, torch.tensor(False), torch.tensor(True)])
@pytest.mark.parametrize("q_mask", [False, True, torch.tensor(False), torch.tensor(True)])
def test_mask_kl_divergence_type(gpu, p_mask, q_mask):
    dk.run_mask_kl_type(gpu, dist.Normal, p_mask, q_mask)


@pytest.mark.parametrize("shape", [None, (), (4,), (3, 2)], ids=str)
def test_mask_noop(gpu, shape):
    dk.run_mask_noop(gpu, dist.Normal, shape)


def test_categorical(gpu):
    dk.run_categorical(gpu)


def test_second_order_gradients_through_the_fused_families(gpu):
    dk.run_second_order_gradients(gpu)
