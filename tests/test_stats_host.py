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
