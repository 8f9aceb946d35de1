"""Gap finder (build container only): runs the REFERENCE's own test files against this package.

``import pyro`` is answered by ``pyro_amd`` (every submodule registered under the ``pyro.`` name), the
HIP kernels are answered by the numpy/torch oracle backend (no GPU here), and pytest collects the test
files where they lie under /root/reference -- nothing is copied, nothing is written there:

    cd /tmp/refsuite && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/tests/refsuite:/root/repo:/root/reference \
      python -m pytest -p pyro_alias_plugin -p no:cacheprovider --rootdir=/tmp/refsuite -q \
      /root/reference/tests/infer/test_valid_models.py

A module of the reference that this package does not have makes the importing test file fail at
collection, which is the information wanted.  What it found and what was restated as committed tests is
recorded in tests/refsuite/RESULTS.md.  This is test infrastructure kept next to the tests (it uses tests/oracle_backend.py): nothing in the
product, in the collected test files, in bench.py or in smoke() imports it, and it cannot run on the GPU box (no /root/reference there).
"""
import importlib
import importlib.util
import os
import pkgutil
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import pyro_amd  # noqa: E402


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _Setter:
    @staticmethod
    def setattr(obj, name, value):
        setattr(obj, name, value)


# the oracle answers the kernels (tests/oracle_backend.py of THIS repo; the name ``tests`` itself must
# stay free for the reference's ``tests.common``)
sys.path.insert(0, ROOT)
_pkg = type(sys)("_pa_tests")
_pkg.__path__ = [os.path.join(ROOT, "tests")]
sys.modules["_pa_tests"] = _pkg
_backend = _load_by_path("_pa_tests.oracle_backend", os.path.join(ROOT, "tests", "oracle_backend.py"))
_backend.install(_Setter)

for info in pkgutil.walk_packages(pyro_amd.__path__, "pyro_amd."):
    if ".csrc" in info.name:
        continue
    try:
        importlib.import_module(info.name)
    except Exception as e:                                         # noqa: BLE001
        print("pyro_alias_plugin: could not import %s: %r" % (info.name, e), file=sys.stderr)
for name, mod in list(sys.modules.items()):
    if name == "pyro_amd" or name.startswith("pyro_amd."):
        sys.modules["pyro" + name[len("pyro_amd"):]] = mod

# The reference keeps one module per handler (pyro/poutine/<name>_messenger.py, trace_struct.py,
# distributions/torch_distribution.py, distributions/distribution.py); this package keeps them in
# handlers.py / trace.py / distributions/base.py.  The reference's TEST FILES import the per-handler
# paths, so the harness (not the product) provides them: synthetic modules holding the same names.
def _module_paths():
    import types
    import torch
    from pyro_amd.distributions import base
    import importlib
    from pyro_amd.poutine import handlers
    trace = importlib.import_module("pyro_amd.poutine.trace")      # (the package exports a FUNCTION of that name)
    H = handlers
    table = {
        "poutine.block_messenger": {"BlockMessenger": H.BlockMessenger},
        "poutine.broadcast_messenger": {"BroadcastMessenger": H.PlateMessenger},
        "poutine.condition_messenger": {"ConditionMessenger": H.ConditionMessenger},
        "poutine.do_messenger": {"DoMessenger": H.DoMessenger},
        "poutine.enum_messenger": {"EnumMessenger": H.EnumMessenger},
        "poutine.equalize_messenger": {"EqualizeMessenger": H.EqualizeMessenger},
        "poutine.escape_messenger": {"EscapeMessenger": H.EscapeMessenger},
        "poutine.indep_messenger": {"CondIndepStackFrame": H.CondIndepStackFrame, "IndepMessenger": H.PlateMessenger},
        "poutine.infer_config_messenger": {"InferConfigMessenger": H.InferConfigMessenger},
        "poutine.lift_messenger": {"LiftMessenger": H.LiftMessenger},
        "poutine.markov_messenger": {"MarkovMessenger": H.MarkovMessenger},
        "poutine.mask_messenger": {"MaskMessenger": H.MaskMessenger},
        "poutine.reparam_messenger": {"ReparamMessenger": H.ReparamMessenger},
        "poutine.replay_messenger": {"ReplayMessenger": H.ReplayMessenger},
        "poutine.scale_messenger": {"ScaleMessenger": H.ScaleMessenger},
        "poutine.seed_messenger": {"SeedMessenger": H.SeedMessenger},
        "poutine.subsample_messenger": {"SubsampleMessenger": H.PlateMessenger, "_Subsample": H._Subsample},
        "poutine.substitute_messenger": {"SubstituteMessenger": H.SubstituteMessenger},
        "poutine.trace_messenger": {"TraceMessenger": H.TraceMessenger, "TraceHandler": H._TraceHandler},
        "poutine.uncondition_messenger": {"UnconditionMessenger": H.UnconditionMessenger},
        "poutine.trace_struct": {"Trace": trace.Trace},
        "distributions.torch_distribution": {k: getattr(base, k) for k in
                                             ("ExpandedDistribution", "MaskedDistribution", "TorchDistribution",
                                              "TorchDistributionMixin") if hasattr(base, k)},
        "distributions.distribution": {"Distribution": torch.distributions.Distribution},
    }
    for name, attrs in table.items():
        mod = types.ModuleType("pyro." + name)
        mod.__dict__.update(attrs)
        sys.modules["pyro." + name] = mod
        parent = sys.modules["pyro." + name.split(".")[0]]
        setattr(parent, name.split(".")[1], mod)


_module_paths()

# small stand-ins for test-support modules of the reference (pyro/distributions/testing/fakes.py)
from pyro_amd_fakes import install as _install_fakes, install_out_of_scope  # noqa: E402
_install_fakes()
install_out_of_scope()
