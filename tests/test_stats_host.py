"""pyro_amd.ops.stats against the reference's known answers (tests/ops/test_stats.py)."""
import pytest
import torch

from tests import stats_kat_cases as sk

CPU = torch.device("cpu")


@pytest.mark.parametrize("case", [sk.run_quantile_pi_hpdi, sk.run_interval_statistics_batch,
                                  sk.run_autocorrelation, sk.run_chain_diagnostics],
                         ids=lambda f: f.__name__[4:])
def test_stats_kats(case):
    case(CPU)


def test_cummin_and_resample():
    import torch
    from pyro_amd.ops.stats import _cummin, resample
    x = torch.tensor([[3.0, 1.0], [2.0, 5.0], [4.0, 0.5], [1.0, 2.0]])
    assert torch.equal(_cummin(x), torch.tensor([[3.0, 1.0], [2.0, 1.0], [2.0, 0.5], [1.0, 0.5]]))
    torch.manual_seed(0)
    y = resample(torch.arange(100.0), 30)
    assert y.shape == (30,) and len(set(y.tolist())) == 30            # without replacement
    z = resample(torch.arange(4.0).reshape(2, 2), 5, dim=1, replacement=True)
    assert z.shape == (2, 5)
