"""Host logic (handlers, Trace_ELBO assembly, autoguide, SVI, flat Adam) against the golden
vectors of the unmodified reference, with the kernels answered by the numpy oracle
(tests/oracle_backend.py).  The same test bodies run on the GPU through the real HIP kernels in
tests/test_svi_gpu.py."""
import os

import numpy as np
import pytest
import torch

from tests import models

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


@pytest.fixture(autouse=True)
def _cpu_backend(oracle_backend):
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


def test_eight_schools_loss_grads_and_trajectory(monkeypatch):
    models.run_eight_schools(load("eight_schools"), torch.device("cpu"), monkeypatch, rtol=1e-9)


@pytest.mark.parametrize("tag,fused", [("f64", False), ("p1", False), ("f64", True), ("p1", True)])
def test_logreg_loss_and_grads(monkeypatch, tag, fused):
    # fused=True exercises the LinearLogits -> glm kernel route (oracle GLM here, f64 arithmetic)
    models.run_logreg(load("logreg_" + tag), torch.device("cpu"), monkeypatch, fused=fused,
                      dtype=torch.float64, rtol=1e-9)


def test_scale_mask_subsample(monkeypatch):
    models.run_scale_mask(load("scale_mask"), torch.device("cpu"), monkeypatch, rtol=1e-9)


def test_score_function_guide(monkeypatch):
    models.run_score_function(load("score_function"), torch.device("cpu"), rtol=1e-9)


@pytest.mark.parametrize("fused", [False, True])
def test_hierarchical_logreg_loss_and_grads(monkeypatch, fused):
    # config 5 at toy size: per-group weights under plate("groups"), ragged groups (one empty)
    models.run_hier(load("hier"), torch.device("cpu"), monkeypatch, fused=fused, rtol=1e-9)


@pytest.mark.parametrize("tag", ["p1", "p5"])
def test_trace_mean_field_elbo(monkeypatch, tag):
    models.run_meanfield(load("meanfield"), torch.device("cpu"), monkeypatch, tag, rtol=1e-9)


def test_predictive(monkeypatch):
    models.run_predictive(load("predictive"), torch.device("cpu"), monkeypatch, rtol=1e-10)


@pytest.mark.parametrize("which", ["diag", "mvn"])
@pytest.mark.parametrize("tag", ["p1", "p4"])
def test_autocontinuous_guides(monkeypatch, which, tag):
    models.run_autocont(load("autocont"), torch.device("cpu"), monkeypatch, which, tag, rtol=1e-9)
