"""The recorder of pyro_amd/ops/fuser.py on a machine without a GPU: a program that exercises every recorded
operator family is recorded with host tensors standing in for device tensors, scheduled into launch levels, and
every generated HIP source is compiled for gfx950 with hiprtc (tools/fuser_dry.py).  Nothing is launched: the
numbers are the GPU tests' business (tests/test_fuser_gpu.py); this pins the code generator's syntax and types
for both dtypes, and what the recorder takes."""
import os

import pytest
import torch


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/libhiprtc.so"), reason="needs hiprtc")
def test_every_generated_source_compiles_for_gfx950():
    from pyro_amd.ops import fuser
    from tools import fuser_dry

    fuser.UNFUSED.clear()
    del fuser_dry.SOURCES[:]
    with fuser_dry.dry():
        for dt in (torch.float32, torch.float64):
            before = dict(fuser.STATS)
            with fuser.Fuser():
                res = fuser_dry.program(dt)
            del res
            d = {k: fuser.STATS[k] - before[k] for k in fuser.STATS}
            assert d["recorded"] >= 150 and d["dead"] > 0, d
    # one launch per level, not per operator; the families' expressions, gathers, softmax, joins and long sums in
    assert 4 <= len(fuser_dry.SOURCES) <= 40
    text = "\n".join(fuser_dry.SOURCES)
    for needle in ("pa::Fam<", "fam_g<", "__shfl_xor", "long long x_", "__builtin_bit_cast(double"):
        assert needle in text, needle
    # only what the program draws / creates with operators outside the recorder's reach is left to ATen
    assert set(fuser.UNFUSED) <= {"aten::randn", "aten::rand", "aten::sgn", "aten::arange", "aten::randint"}, \
        fuser.UNFUSED
    assert fuser._dev(torch.empty(1)) is False                 # (the stand-ins are gone)
