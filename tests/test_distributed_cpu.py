"""Multi-process (world size 2, gloo, 127.0.0.1) tests of the N>1 paths: the
particle-sharded SVI step with ONE flat gradient all-reduce, and chain-sharded MCMC."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest
import torch

from tests import dist_worker as dw


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(target, world, args, tmp_path, tag):
    ctx = mp.get_context("spawn")
    port = _free_port()
    out = str(tmp_path / (tag + "_%d.pt"))
    procs = [ctx.Process(target=target, args=(r, world, port, out) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "worker failed (exit code %s)" % p.exitcode
    return [torch.load(out % r, weights_only=False) for r in range(world)]


@pytest.mark.timeout(600)
def test_particle_sharded_svi_equals_single_process(tmp_path):
    """2 ranks x P particles with a flat gradient all-reduce (mean) == 1 process with the 2P
    particles: identical parameters after every step, on every rank."""
    g = np.random.default_rng(0)
    N, D, P, steps = 200, 5, 4, 3
    X = g.standard_normal((N, D))
    y = (g.uniform(size=N) < 0.5).astype(np.float64)

    def bank(seed):
        r = np.random.default_rng(seed)
        out = []
        for _ in range(steps):
            out += [r.standard_normal((P, 1, D)), r.standard_normal((P, 1))]   # sites w, b
        return out

    b0, b1 = bank(1), bank(2)
    both = [np.concatenate([a, b], axis=0) for a, b in zip(b0, b1)]
    two = _run(dw.svi_worker, 2, ([b0, b1], X, y, steps), tmp_path, "svi2")
    one = _run(dw.svi_worker, 1, ([both], X, y, steps), tmp_path, "svi1")
    for k in two[0]["params"]:
        torch.testing.assert_close(two[0]["params"][k], two[1]["params"][k], rtol=0, atol=0)
        torch.testing.assert_close(two[0]["params"][k], one[0]["params"][k], rtol=1e-10, atol=1e-12)
    # each rank reports the loss of ITS particles; their mean is the 2P-particle loss
    np.testing.assert_allclose((np.array(two[0]["losses"]) + np.array(two[1]["losses"])) / 2,
                               one[0]["losses"], rtol=1e-10)


@pytest.mark.timeout(600)
def test_chain_sharded_mcmc_reproduces_single_process_chains(tmp_path):
    from tests import mcmc_cases as mc
    D, C, S = 6, 4, 3
    Lam = mc.make_precision(D, 2)
    z0 = np.random.default_rng(3).standard_normal((C, D)) * 0.3
    two = _run(dw.mcmc_worker, 2, (Lam, z0, C, S), tmp_path, "mcmc2")
    one = _run(dw.mcmc_worker, 1, (Lam, z0, C, S), tmp_path, "mcmc1")
    assert [r["local_chains"] for r in two] == [2, 2] and [r["offset"] for r in two] == [0, 2]
    assert two[0]["x"].shape == (C, S, D)
    torch.testing.assert_close(two[0]["x"], two[1]["x"], rtol=0, atol=0)      # all_gather
    torch.testing.assert_close(two[0]["x"], one[0]["x"], rtol=1e-12, atol=1e-12)


@pytest.mark.timeout(600)
def test_data_sharded_svi_equals_single_process(tmp_path):
    """2 ranks x half of the plate's rows (scaled to the full plate), same particles, generic
    per-parameter optimizer behind RcclOptimizer == 1 process with all rows."""
    g = np.random.default_rng(3)
    N, D, P, steps = 120, 4, 3, 3
    X = g.standard_normal((N, D))
    y = (g.uniform(size=N) < 0.5).astype(np.float64)
    bank = []
    for _ in range(steps):
        bank += [g.standard_normal((P, 1, D)), g.standard_normal((P, 1))]
    two = _run(dw.data_sharded_worker, 2, (bank, X, y, steps), tmp_path, "ds2")
    one = _run(dw.data_sharded_worker, 1, (bank, X, y, steps), tmp_path, "ds1")
    for k in two[0]["params"]:
        torch.testing.assert_close(two[0]["params"][k], two[1]["params"][k], rtol=0, atol=0)
        torch.testing.assert_close(two[0]["params"][k], one[0]["params"][k], rtol=1e-9, atol=1e-11)


@pytest.mark.timeout(600)
def test_bench_torchrun_path_prints_one_line_with_two_ranks(tmp_path):
    """`bench.py --gpus 2 --steps 3` end to end under the driver's launch contract (two processes,
    RANK / WORLD_SIZE / MASTER_* from the environment), host tensors + gloo + oracle kernels: rank 0
    prints exactly one JSON line that says two ranks took part in the gradient all-reduce; rank 1
    prints nothing."""
    import json
    outs = _run(dw.bench_worker, 2, (3,), tmp_path, "bench2")
    lines = [ln for ln in outs[0]["stdout"].splitlines() if ln.strip()]
    assert len(lines) == 1, outs[0]["stdout"]
    assert len(lines[0]) < 4096                   # (the driver keeps a bounded tail of stdout)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["steps"] == 3
    assert {"value", "ms_per_step", "roofline", "config", "dtype", "unit"} <= set(rec)
    assert rec["metric"].startswith("ELBO-grad steps/sec") and rec["value"] > 0
    assert rec["scaling"] == "weak" and rec["higher_is_better"] is True
    assert outs[1]["stdout"].strip() == ""
    sh = rec["other_configs"]["config5_plate_sharded"]
    assert sh["ranks"] == 2 and sh["rows_total"] == 600 and sh["steps_per_s"] > 0



@pytest.mark.timeout(900)
def test_bench_with_eight_ranks_including_both_nuts_blocks(tmp_path):
    """`bench.py --gpus 8 --steps 3` as the driver launches it on an 8-GPU node, on host tensors over gloo
    with the oracle's kernels: eight ranks in the gradient all-reduce, config 5 sharded over them, the
    chain-sharded NUTS blocks summed over ranks -- so that a first 8-GPU run does not trip on rank > 1
    bookkeeping.  Rank 0 prints the one line; the others nothing."""
    import json
    outs = _run(dw.bench_worker, 8, (3, True), tmp_path, "bench8")
    lines = [ln for ln in outs[0]["stdout"].splitlines() if ln.strip()]
    assert len(lines) == 1, outs[0]["stdout"]
    assert len(lines[0]) < 4096
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["rccl_ranks"] == 8 and rec["steps"] == 3
    assert rec["value"] > 0 and rec["scaling"] == "weak"
    sh = rec["other_configs"]["config5_plate_sharded"]
    assert sh["ranks"] == 8 and sh["rows_total"] == 600 and sh["steps_per_s"] > 0
    nuts = rec["secondary"]
    assert nuts["n_gpus"] == 8 and nuts["leapfrogs"] >= 8 * 2 * (6 + 4)      # >= one leapfrog per transition
    mn = rec["secondary_model_nuts"]
    assert "error" not in mn, mn
    run = next(iter(mn["runs"].values()))
    assert mn["n_gpus"] == 8 and run["leapfrogs"] >= 8 * 2 * (5 + 3) and run["value"] > 0
    assert rec["full_record"] == "bench_full.json"
    for r in range(1, 8):
        assert outs[r]["stdout"].strip() == ""


def test_captured_collective_falls_back_to_the_split_form(monkeypatch):
    """SVI._capture with several ranks: the step is first captured as ONE graph holding the gradient
    all-reduce; when that capture fails the split form (graph 1 -> eager collective -> graph 2) is
    captured instead, with a warning, and the captured-step mode stays on.  The capture itself needs a
    GPU; its CONTROL FLOW does not (the one-graph form has never met a second rank on hardware, so the
    fall-back is what a first multi-GPU run may take)."""
    import warnings

    import pyro_amd as pyro
    from pyro_amd.infer import SVI, Trace_ELBO

    class _TwoRankOptim:                 # what RcclOptimizer looks like to SVI at world size 2
        multi_rank = True
        zeroes_grads = True

        def reduce_gradients(self, params):
            pass

        def __call__(self, params, *a, **k):
            pass

    svi = SVI(lambda: None, lambda: None, _TwoRankOptim(), Trace_ELBO(), hip_graph=True)
    forms = []

    def fake_capture_once(key, args, kwargs, rec, force_split=None, quiet=False, with_gate=False):
        assert with_gate is False            # (several ranks: a collective cannot be given up)
        forms.append((force_split, quiet))
        if force_split is False:         # the one-graph form fails (as a capture error would)
            svi.hip_graph = False
            return None
        return "split-entry"

    monkeypatch.setattr(svi, "_capture_once", fake_capture_once)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        entry = svi._capture(("k",), (), {})
    assert entry == "split-entry" and forms == [(False, True), (True, False)]
    assert svi.hip_graph is True and any("two graphs" in str(x.message) for x in w)
    # PYRO_AMD_GRAPH_COLLECTIVE=0 goes straight to the default (split) form
    forms.clear()
    monkeypatch.setenv("PYRO_AMD_GRAPH_COLLECTIVE", "0")
    assert svi._capture(("k",), (), {}) == "split-entry" and forms == [(None, False)]
    # the one-graph form succeeding is taken as is
    monkeypatch.delenv("PYRO_AMD_GRAPH_COLLECTIVE")
    forms.clear()
    monkeypatch.setattr(svi, "_capture_once",
                        lambda *a, force_split=None, quiet=False, with_gate=False: "one-graph")
    assert svi._capture(("k",), (), {}) == "one-graph"
    pyro.clear_param_store()


def test_prearmed_capture_tries_the_late_gate_then_the_first_node_gate_then_none(monkeypatch):
    """SVI._capture with prearm: the step is captured with the gate in front of its chained tail; if that
    capture is not armable (something other than the GLM kernel ran in front of the gate) with the gate as
    first node; if that is not armable either (a torch kernel in the step) without a gate.  Control flow
    only (the captures need a GPU: tests/test_chain_gpu.py)."""
    import pyro_amd as pyro
    from pyro_amd.infer import SVI, Trace_ELBO

    class _Optim:
        zeroes_grads = True            # (a flat optimizer: its update sits in the graph; others update eagerly, un-gated)

        def __call__(self, params, *a, **k):
            pass

    class _Gate:
        def __init__(self, armable):
            self.armable = armable

    class _Entry:
        def __init__(self, gate):
            self.gate = gate

    def run(armable_by_form, speculate=True):
        svi = SVI(lambda: None, lambda: None, _Optim(), Trace_ELBO(), hip_graph=True, prearm=True,
                  speculate=speculate)
        forms = []

        def fake_capture_once(key, args, kwargs, rec, force_split=None, quiet=False, with_gate=False):
            forms.append(with_gate)
            e = _Entry(None if with_gate is False else _Gate(armable_by_form[with_gate]))
            svi._graphs[key] = e
            return e

        monkeypatch.setattr(svi, "_capture_once", fake_capture_once)
        entry = svi._capture(("k",), (), {})
        return forms, entry

    forms, entry = run({"late": True})
    assert forms == ["late"] and entry.gate.armable
    forms, entry = run({"late": False, True: True})
    assert forms == ["late", True] and entry.gate.armable
    forms, entry = run({"late": False, True: False})
    assert forms == ["late", True, False] and entry.gate is None
    forms, entry = run({True: False}, speculate=False)
    assert forms == [True, False] and entry.gate is None
    pyro.clear_param_store()


def test_step_gate_armable_truth_table():
    """kernels.StepGate.armable: every launch behind the gate polls it, torch launched nothing, the node was
    emitted, and -- for a late gate -- nothing but the plane-image GLM kernel ran in front of it."""
    from types import SimpleNamespace as NS

    from pyro_amd.kernels import StepGate

    def armable(**kw):
        base = dict(emitted=True, total=3, aware=3, torch_ops=0, late=False, pre=0, pre_other=0)
        base.update(kw)
        return StepGate.armable.fget(NS(**base))

    assert armable()
    assert not armable(aware=2)                       # a launch of ours that does not poll the gate
    assert not armable(torch_ops=1)                   # a torch kernel in the step
    assert not armable(total=0, aware=0)              # nothing captured
    assert armable(late=True, total=2, aware=2, pre=1, pre_other=0)
    assert not armable(late=True, total=2, aware=2, pre=2, pre_other=1)     # e.g. a separate guide-draw launch
    assert not armable(late=True, emitted=False, total=0, aware=0, pre=3)   # no chained tail to carry the gate
