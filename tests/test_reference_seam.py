"""The reference-side binding story, executed: INTEGRATION.md's ``HipTrace_ELBO`` /
fused-distribution stubs run under the unmodified reference's ``pyro.infer.SVI`` (build container
only: /root/reference does not travel to the GPU box)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/pyro"),
                    reason="the unmodified reference is only present in the build container")
def test_integration_stubs_run_under_the_real_reference_svi():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, os.path.join(HERE, "reference_seam.py")], env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "SEAM OK" in res.stdout
