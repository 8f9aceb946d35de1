"""Reference distribution KATs on the torch-wrapped classes (CPU): Delta, .mask, Categorical."""
import pytest
import torch

import pyro_amd.distributions as dist
from tests import dist_kat_cases as dk

CPU = torch.device("cpu")


def test_delta():
    dk.run_delta(CPU)


@pytest.mark.parametrize("batch_dim,event_dim", [(b, e) for b in range(4) for e in range(1 + b)])
@pytest.mark.parametrize("has_log_density", [False, True])
def test_delta_shapes(batch_dim, event_dim, has_log_density):
    dk.run_delta_shapes(CPU, batch_dim, event_dim, has_log_density)


@pytest.mark.parametrize("batch_shape", [(), [], (2,), [2], torch.Size([2]), [2, 3]])
def test_delta_expand(batch_shape):
    dk.run_delta_expand(CPU, batch_shape)


@pytest.mark.parametrize("batch_dim,mask_dim", [(b, m) for b in range(3) for m in range(1 + b)])
@pytest.mark.parametrize("event_dim", [0, 1, 2])
def test_mask(batch_dim, event_dim, mask_dim):
    # a torch-wrapped (un-fused) discrete family: the fused ones refuse CPU tensors by design
    dk.run_mask(CPU, lambda shape: dist.Geometric(torch.tensor(0.1)).expand_by(shape),
                batch_dim, event_dim, mask_dim)


@pytest.mark.parametrize("mask", [False, True, torch.tensor(False), torch.tensor(True)])
def test_mask_type(mask):
    dk.run_mask_type(CPU, dist.Laplace, mask)


@pytest.mark.parametrize("event_shape", [(), (4,)])
@pytest.mark.parametrize("dist_shape", [(), (3,), (2, 1), (2, 3)])
@pytest.mark.parametrize("mask_shape", [(), (3,), (2, 1), (2, 3)])
def test_mask_broadcast(event_shape, dist_shape, mask_shape):
    dk.run_mask_broadcast(CPU, dist.Laplace, event_shape, dist_shape, mask_shape)


def test_mask_kl_divergence():
    dk.run_mask_kl(CPU, dist.Laplace)


@pytest.mark.parametrize("p_mask", [False, True, torch.tensor(False), torch.tensor(True)])
@pytest.mark.parametrize("q_mask", [False, True, torch.tensor(False), torch.tensor(True)])
def test_mask_kl_divergence_type(p_mask, q_mask):
    dk.run_mask_kl_type(CPU, dist.Laplace, p_mask, q_mask)


@pytest.mark.parametrize("shape", [None, (), (4,), (3, 2)], ids=str)
def test_mask_noop(shape):
    dk.run_mask_noop(CPU, dist.Laplace, shape)


def test_categorical():
    dk.run_categorical(CPU)


def test_families_built_from_python_numbers_expand(oracle_backend):
    """dist.Beta(1.0, 1.0) inside a plate (tests/infer/mcmc/test_mcmc_util.py beta_bernoulli)."""
    import pyro_amd.distributions as dist
    for d in (dist.Beta(1.0, 1.0), dist.Gamma(2.0, 1.0), dist.Normal(0.0, 1.0), dist.Poisson(3.0),
              dist.Exponential(1.0), dist.LogNormal(0.0, 1.0), dist.HalfCauchy(1.0), dist.HalfNormal(1.0)):
        e = d.expand((3,))
        assert e.batch_shape == (3,)
        x = e.sample()
        assert x.shape == (3,) and e.log_prob(x).shape == (3,)


def test_second_order_gradients_through_the_fused_families(oracle_backend):
    dk.run_second_order_gradients(torch.device("cpu"))


def test_a_user_distribution_without_expand_works_inside_plates(oracle_backend):
    """tests/distributions/test_distributions.py (test_expand_new_dim & co. with the default expand): a
    TorchDistribution subclass that does not write ``expand`` gets the generic ExpandedDistribution."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd import poutine
    from torch.distributions import constraints

    class Shifted(dist.TorchDistribution):
        arg_constraints = {"loc": constraints.real}
        support = constraints.real
        has_rsample = True

        def __init__(self, loc):
            self.loc = loc
            super().__init__(loc.shape, validate_args=False)

        def rsample(self, sample_shape=torch.Size()):
            shape = torch.Size(sample_shape) + self.loc.shape
            return self.loc + torch.randn(shape)

        def log_prob(self, value):
            return -0.5 * (value - self.loc) ** 2

    d = Shifted(torch.tensor([[0.0], [10.0]]))              # batch (2, 1)
    big = d.expand((3, 2, 4))
    assert isinstance(big, dist.ExpandedDistribution) and big.batch_shape == (3, 2, 4)
    x = big.rsample((5,))
    assert x.shape == (5, 3, 2, 4)
    assert (x[:, :, 0].abs() < 6).all() and ((x[:, :, 1] - 10).abs() < 6).all()
    assert x[0, 0, 0].unique().numel() == 4                  # independent along the stretched dim
    assert big.log_prob(x).shape == (5, 3, 2, 4)
    assert big.expand((7, 3, 2, 4)).batch_shape == (7, 3, 2, 4)
    for bad in ((2, 4), (3, 3, 4)):
        with pytest.raises(ValueError, match="Cannot broadcast"):
            Shifted(torch.zeros(2, 1)).expand((3, 2, 4)).expand(bad)

    def model():
        with pyro.plate("a", 4, dim=-1), pyro.plate("b", 3, dim=-3):
            return pyro.sample("x", Shifted(torch.tensor([[0.0], [10.0]])))

    tr = poutine.trace(model).get_trace()
    assert tr.nodes["x"]["value"].shape == (3, 2, 4)
    tr.compute_log_prob()
    assert tr.nodes["x"]["log_prob"].shape == (3, 2, 4)


def test_sum_plan_lists_one_pass_per_run_of_reduced_dims():
    """fused._sum_plan: the pa_sum_to_nd passes [(A, R, B)] that bring a gradient of the frame's shape down
    to a broadcast operand's shape -- adjacent reduced dims share a pass, size-1 dims are skipped, a pass
    whose leading extent does not fit the kernel's grid gives None (torch's sum_to_size then)."""
    from pyro_amd.distributions.fused import _sum_plan
    assert _sum_plan((64, 1000, 32), (64, 1, 32)) == [(64, 1000, 32)]
    assert _sum_plan((64, 1000, 32), (32,)) == [(1, 64000, 32)]
    assert _sum_plan((64, 1000, 32), (64, 1000, 1)) == [(64000, 32, 1)]
    assert _sum_plan((64, 1000, 32), (1, 1000, 1)) == [(64000, 32, 1), (1, 64, 1000)]
    assert _sum_plan((5, 1, 7), (7,)) == [(1, 5, 7)]
    assert _sum_plan((4, 6), (4, 6)) == []
    assert _sum_plan((70000, 3, 5), (70000, 1, 5)) is None          # A >= 65536: outside the grid
    # every plan reproduces sum_to_size
    import itertools
    g = torch.arange(2 * 3 * 4 * 5, dtype=torch.float64).reshape(2, 3, 4, 5)
    for like in itertools.product((1, 2), (1, 3), (1, 4), (1, 5)):
        x = g
        for A, R, B in _sum_plan(g.shape, like):
            x = x.reshape(A, R, B).sum(1)
        torch.testing.assert_close(x.reshape(like), g.sum_to_size(like))


def test_shape_and_scaling_helpers():
    """distributions/util.py: block sums at either end of a tensor, the un-fused scale_and_mask (SURVEY 8a
    row a5 for log-densities that come from torch's own classes), numpy-style broadcast of shape tuples."""
    from pyro_amd.distributions import util as u
    t = torch.arange(120.0).reshape(2, 3, 4, 5)
    assert torch.equal(u.sum_rightmost(t, 2), t.sum((-1, -2))) and torch.equal(u.sum_leftmost(t, 2), t.sum((0, 1)))
    assert torch.equal(u.sum_rightmost(t, -1), t.sum((1, 2, 3)))          # keep one leftmost dim
    assert torch.equal(u.sum_leftmost(t, -1), t.sum((0, 1, 2)))           # keep one rightmost dim
    assert u.sum_rightmost(t, 0) is t and u.sum_rightmost(t, 9).shape == () and u.sum_rightmost(2.5, 3) == 2.5
    m = t.remainder(2) == 0
    assert u.scale_and_mask(t, 1.0, None) is t and u.scale_and_mask(t, 1.0, True) is t
    assert torch.equal(u.scale_and_mask(t, 3.0), 3.0 * t)
    assert torch.equal(u.scale_and_mask(t, 3.0, m), torch.where(m, 3.0 * t, torch.zeros(())))
    assert torch.equal(u.scale_and_mask(t, 1.0, m), torch.where(m, t, torch.zeros(())))
    assert not u.scale_and_mask(t, 3.0, False).any() and u.scale_and_mask(0.0, 3.0, m) == 0.0
    assert u.broadcast_shape((2, 1), (3,), ()) == (2, 3) and u.broadcast_shape() == ()
    assert u.broadcast_shape((4, 1, 1), (2, 3)) == (4, 2, 3)
    assert u.broadcast_shape((2, 3), (3,), strict=True) == (2, 3)
    for a, b, strict in (((2,), (3,), False), ((2, 1), (2, 3), True), ((1,), (3,), True)):
        with pytest.raises(ValueError, match="shape mismatch"):
            u.broadcast_shape(a, b, strict=strict)
