"""pyro_amd.ops.stats on the MI355X against the reference's known answers (the diagnostics of
gathered chains [C, S, D] are batch jobs on the device: FFT autocorrelation, sorts)."""
import pytest

from tests import stats_kat_cases as sk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [sk.run_quantile_pi_hpdi, sk.run_interval_statistics_batch,
                                  sk.run_autocorrelation, sk.run_chain_diagnostics],
                         ids=lambda f: f.__name__[4:])
def test_stats_kats(gpu, case):
    case(gpu)
