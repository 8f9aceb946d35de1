"""Known-answer tests of the reference's own suite for the ELBO-gradient path, restated against the
drop-in API (tests/infer/test_gradient.py:38-127 test_particle_gradient, :130-215
test_subsample_gradient, :218-275 test_plate, :277-330 test_plate_elbo_vectorized_particles;
tests/infer/test_inference.py:943-1007 plate-sum semantics).  The bodies follow the reference tests
line by line -- same data, seeds where they matter, closed-form expectations and precisions; only
the import changes.  Shared by the CPU suite (kernels answered by the numpy oracle: host logic) and
the GPU suite (HIP kernels)."""
import math

import numpy as np
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
import pyro_amd.poutine as poutine
from pyro_amd.distributions import constraints
from pyro_amd.infer import SVI, Trace_ELBO, TraceEnum_ELBO, TraceGraph_ELBO, TraceMeanField_ELBO
from pyro_amd.optim import Adam


class NonreparameterizedNormal(dist.Normal):       # tests/common.py fakes
    has_rsample = False


def DiffTrace_ELBO(*args, **kwargs):
    return Trace_ELBO(*args, **kwargs).differentiable_loss


ELBOS = {"Trace_ELBO": Trace_ELBO, "TraceEnum_ELBO": TraceEnum_ELBO, "TraceGraph_ELBO": TraceGraph_ELBO,
         "TraceMeanField_ELBO": TraceMeanField_ELBO, "DiffTrace_ELBO": DiffTrace_ELBO}


def _t(x, device):
    return torch.tensor(x, device=device)


def run_particle_gradient(device, elbo_name, reparameterized, has_rsample):
    """test_gradient.py:38-127: one particle, gradients equal the closed-form pathwise /
    score-function estimator evaluated at the very draw the ELBO used."""
    Elbo = ELBOS[elbo_name]
    pyro.clear_param_store()
    data = _t([-0.5, 2.0], device)
    Normal = dist.Normal if reparameterized else NonreparameterizedNormal
    zero, one = _t(0.0, device), _t(1.0, device)

    def model():
        with pyro.plate("data", len(data)) as ind:
            x = data[ind]
            z = pyro.sample("z", Normal(zero, one))
            pyro.sample("x", Normal(z, one), obs=x)

    def guide():
        scale = pyro.param("scale", lambda: _t([1.0], device))
        with pyro.plate("data", len(data)):
            loc = pyro.param("loc", lambda: torch.zeros(len(data), device=device), event_dim=0)
            z_dist = Normal(loc, scale)
            if has_rsample is not None:
                z_dist.has_rsample_(has_rsample)
            pyro.sample("z", z_dist)

    elbo = Elbo(max_plate_nesting=1, num_particles=1, strict_enumeration_warning=False)
    pyro.set_rng_seed(0)
    elbo.loss_and_grads(model, guide)
    params = dict(pyro.get_param_store().named_parameters())
    actual_grads = {name: param.grad.detach().cpu() for name, param in params.items()}

    pyro.set_rng_seed(0)
    guide_tr = poutine.trace(guide).get_trace()
    model_tr = poutine.trace(poutine.replay(model, guide_tr)).get_trace()
    guide_tr.compute_log_prob()
    model_tr.compute_log_prob()
    x = data.cpu()
    z = guide_tr.nodes["z"]["value"].detach().cpu()
    loc = pyro.param("loc").detach().cpu()
    scale = pyro.param("scale").detach().cpu()

    if reparameterized and has_rsample is not False:
        expected_grads = {
            "scale": -(-z * (z - loc) + (x - z) * (z - loc) + 1).sum(0, keepdim=True) / scale,
            "loc": -(-z + (x - z)),
        }
    else:
        elbo_v = (model_tr.nodes["x"]["log_prob"].detach().cpu()
                  + model_tr.nodes["z"]["log_prob"].detach().cpu()
                  - guide_tr.nodes["z"]["log_prob"].detach().cpu())
        dlogq_dloc = (z - loc) / scale ** 2
        dlogq_dscale = (z - loc) ** 2 / scale ** 3 - 1 / scale
        if Elbo is TraceEnum_ELBO:
            expected_grads = {"scale": -(dlogq_dscale * elbo_v - dlogq_dscale).sum(0, keepdim=True),
                              "loc": -(dlogq_dloc * elbo_v - dlogq_dloc)}
        else:
            expected_grads = {"scale": -(dlogq_dscale * elbo_v).sum(0, keepdim=True),
                              "loc": -(dlogq_dloc * elbo_v)}
    for name in sorted(params):
        np.testing.assert_allclose(actual_grads[name].numpy(), expected_grads[name].numpy(),
                                   rtol=1e-4, atol=1e-4, err_msg=name)


def run_subsample_gradient(device, elbo_name, reparameterized, has_rsample, subsample, scale,
                           num_particles=50000):
    """test_gradient.py:130-215: E[grad] = scale * ([0.5, -2.0], [2.0]) with and without
    subsampling and poutine.scale, 50 000 vectorised particles, precision 0.06 * scale."""
    Elbo = ELBOS[elbo_name]
    pyro.clear_param_store()
    data = _t([-0.5, 2.0], device)
    subsample_size = 1 if subsample else len(data)
    precision = 0.06 * scale
    if not (reparameterized and has_rsample is not False):
        # 0.06 is ~1.3 sigma of the score-function estimator at 50 000 particles: the reference
        # passes on its torch RNG stream at seed 0, the Philox stream draws other numbers, so
        # these cases take 8x the particles (same precision)
        num_particles *= 8
    Normal = dist.Normal if reparameterized else NonreparameterizedNormal
    zero, one = _t(0.0, device), _t(1.0, device)

    def model(subsample):
        with pyro.plate("data", len(data), subsample_size, subsample) as ind:
            x = data[ind]
            z = pyro.sample("z", Normal(zero, one))
            pyro.sample("x", Normal(z, one), obs=x)

    def guide(subsample):
        scale_ = pyro.param("scale", lambda: _t([1.0], device))
        with pyro.plate("data", len(data), subsample_size, subsample):
            loc = pyro.param("loc", lambda: torch.zeros(len(data), device=device), event_dim=0)
            z_dist = Normal(loc, scale_)
            if has_rsample is not None:
                z_dist.has_rsample_(has_rsample)
            pyro.sample("z", z_dist)

    if scale != 1.0:
        model = poutine.scale(model, scale=scale)
        guide = poutine.scale(guide, scale=scale)

    elbo = Elbo(max_plate_nesting=1, num_particles=num_particles, vectorize_particles=True,
                strict_enumeration_warning=False)
    inference = SVI(model, guide, Adam({"lr": 0.1}), loss=elbo)
    pyro.set_rng_seed(0)        # the reference's conftest seeds every test with 0
    if subsample_size == 1:
        inference.loss_and_grads(model, guide, subsample=_t([0], device))
        inference.loss_and_grads(model, guide, subsample=_t([1], device))
    else:
        inference.loss_and_grads(model, guide, subsample=_t([0, 1], device))
    params = dict(pyro.get_param_store().named_parameters())
    normalizer = 2 if subsample else 1
    actual_grads = {name: param.grad.detach().cpu().numpy() / normalizer
                    for name, param in params.items()}
    expected_grads = {"loc": scale * np.array([0.5, -2.0]), "scale": scale * np.array([2.0])}
    for name in sorted(params):
        np.testing.assert_allclose(actual_grads[name], expected_grads[name], atol=precision,
                                   err_msg=name)


def run_plate(device, elbo_name, reparameterized, num_particles=200000, vectorized_elbo=False):
    """test_gradient.py:218-275 (particles as an explicit outer plate, nuisance sites in between)
    and :277-330 (the ELBO's own vectorised particles around nested plates)."""
    Elbo = ELBOS[elbo_name]
    pyro.clear_param_store()
    data = _t([-0.5, 2.0], device)
    precision = 0.06
    Normal = dist.Normal if reparameterized else NonreparameterizedNormal

    def c(v):
        return _t(float(v), device)

    if not vectorized_elbo:
        def model():
            particles_plate = pyro.plate("particles", num_particles, dim=-2)
            data_plate = pyro.plate("data", len(data), dim=-1)
            pyro.sample("nuisance_a", Normal(c(0), c(1)))
            with particles_plate, data_plate:
                z = pyro.sample("z", Normal(c(0), c(1)))
            pyro.sample("nuisance_b", Normal(c(2), c(3)))
            with data_plate, particles_plate:
                pyro.sample("x", Normal(z, c(1)), obs=data)
            pyro.sample("nuisance_c", Normal(c(4), c(5)))

        def guide():
            loc = pyro.param("loc", torch.zeros(len(data), device=device))
            scale = pyro.param("scale", _t([1.0], device))
            pyro.sample("nuisance_c", Normal(c(4), c(5)))
            with pyro.plate("particles", num_particles, dim=-2):
                with pyro.plate("data", len(data), dim=-1):
                    pyro.sample("z", Normal(loc, scale))
            pyro.sample("nuisance_b", Normal(c(2), c(3)))
            pyro.sample("nuisance_a", Normal(c(0), c(1)))

        elbo = Elbo(strict_enumeration_warning=False)
        norm = num_particles
    else:
        def model():
            data_plate = pyro.plate("data", len(data))
            pyro.sample("nuisance_a", Normal(c(0), c(1)))
            with data_plate:
                z = pyro.sample("z", Normal(c(0), c(1)))
            pyro.sample("nuisance_b", Normal(c(2), c(3)))
            with data_plate:
                pyro.sample("x", Normal(z, c(1)), obs=data)
            pyro.sample("nuisance_c", Normal(c(4), c(5)))

        def guide():
            loc = pyro.param("loc", torch.zeros(len(data), device=device))
            scale = pyro.param("scale", _t([1.0], device))
            pyro.sample("nuisance_c", Normal(c(4), c(5)))
            with pyro.plate("data", len(data)):
                pyro.sample("z", Normal(loc, scale))
            pyro.sample("nuisance_b", Normal(c(2), c(3)))
            pyro.sample("nuisance_a", Normal(c(0), c(1)))

        elbo = Elbo(num_particles=num_particles, vectorize_particles=True, max_plate_nesting=1,
                    strict_enumeration_warning=False)
        norm = 1
    inference = SVI(model, guide, Adam({"lr": 0.1}), loss=elbo)
    pyro.set_rng_seed(0)
    inference.loss_and_grads(model, guide)
    params = dict(pyro.get_param_store().named_parameters())
    actual_grads = {name: param.grad.detach().cpu().numpy() / norm for name, param in params.items()}
    expected_grads = {"loc": np.array([0.5, -2.0]), "scale": np.array([2.0])}
    for name in sorted(params):
        np.testing.assert_allclose(actual_grads[name], expected_grads[name], atol=precision,
                                   err_msg=name)


def run_plating_sums(device):
    """tests/infer/test_inference.py:943-1007: the ELBO of nested / sequential / non-nested plates
    counts every site exactly once per plate index."""
    def c(v):
        return _t(float(v), device)

    def nested_model(data):
        with pyro.plate("a", 3, dim=-2):
            with pyro.plate("b", 4, dim=-1):
                pyro.sample("x", dist.Normal(c(0), c(1)), obs=data)

    def sequential_model(data):
        for i in pyro.plate("a", 3):
            with pyro.plate("b_{}".format(i), 4):
                pyro.sample("x_{}".format(i), dist.Normal(c(0), c(1)), obs=data[i])

    def non_nested_model(data):
        pa = pyro.plate("a", 3, dim=-2)
        pb = pyro.plate("b", 4, dim=-1)
        with pa:
            pyro.sample("u", dist.Normal(c(0), c(1)).expand([3, 1]), obs=data[:, :1])
        with pb:
            pyro.sample("v", dist.Normal(c(0), c(1)).expand([4]), obs=data[0])
        with pa, pb:
            pyro.sample("x", dist.Normal(c(0), c(1)), obs=data)

    def guide(data):
        pass

    g = torch.Generator().manual_seed(0)
    data = torch.randn(3, 4, generator=g).to(device)
    lp = torch.distributions.Normal(0.0, 1.0).log_prob(data.cpu())
    for model, expected in ((nested_model, lp.sum()), (sequential_model, lp.sum()),
                            (non_nested_model, lp.sum() + lp[:, :1].sum() + lp[0].sum())):
        pyro.clear_param_store()
        loss = Trace_ELBO(max_plate_nesting=2).loss(model, guide, data)
        np.testing.assert_allclose(loss, -expected.item(), rtol=1e-5)


def run_normal_normal(device, elbo_name, reparameterized, n_steps, hip_graph=True):
    """tests/infer/test_inference.py:55-173 NormalNormalTests.do_elbo_test: conjugate
    Normal-Normal with known precisions; SVI (Adam lr 0.001) from a perturbed start must reach the
    analytic posterior (MSE of loc and log sigma < 0.05).  The step runs as a hipGraph replay."""
    Elbo = ELBOS[elbo_name]
    lam0, loc0 = _t([0.1, 0.1], device), _t([0.0, 0.5], device)
    lam = _t([6.0, 4.0], device)
    data = _t([[-0.1, 0.3], [0.00, 0.4], [0.20, 0.5], [0.10, 0.7]], device)
    n_data = float(len(data))
    analytic_lam_n = lam0 + n_data * lam
    analytic_log_sig_n = -0.5 * torch.log(analytic_lam_n)
    analytic_loc_n = data.sum(0) * (lam / analytic_lam_n) + loc0 * (lam0 / analytic_lam_n)
    prior_scale, obs_scale = torch.pow(lam0, -0.5), torch.pow(lam, -0.5)
    Normal = dist.Normal if reparameterized else NonreparameterizedNormal
    pyro.clear_param_store()

    def model():
        loc_latent = pyro.sample("loc_latent", dist.Normal(loc0, prior_scale).to_event(1))
        with pyro.plate("data", 4):
            pyro.sample("obs", dist.Normal(loc_latent, obs_scale).to_event(1), obs=data)
        return loc_latent

    def guide():
        loc_q = pyro.param("loc_q", lambda: analytic_loc_n.detach() + 0.134)
        log_sig_q = pyro.param("log_sig_q", lambda: analytic_log_sig_n.detach() - 0.14)
        pyro.sample("loc_latent", Normal(loc_q, torch.exp(log_sig_q)).to_event(1))

    pyro.set_rng_seed(0)
    svi = SVI(model, guide, Adam({"lr": 0.001}), loss=Elbo(), hip_graph=hip_graph)
    for _ in range(n_steps):
        svi.step()
    if hip_graph:
        assert svi.hip_graph and len(svi._graphs) == 1
    loc_error = float(((pyro.param("loc_q").detach() - analytic_loc_n) ** 2).mean())
    log_sig_error = float(((pyro.param("log_sig_q").detach() - analytic_log_sig_n) ** 2).mean())
    assert loc_error < 0.05 and log_sig_error < 0.05, (loc_error, log_sig_error)
    return loc_error, log_sig_error


def run_tracegraph_normal_normal(device, reparameterized, n_steps, prec, baseline=None, lr=0.0015):
    """tests/integration_tests/test_tracegraph_elbo.py:25-105 NormalNormalTests.do_elbo_test:
    TraceGraph_ELBO on the conjugate Normal-Normal model with the latent inside a plate and the
    observations in a Python loop; Adam(lr 0.0015, betas (0.97, 0.999)); the guide must reach the
    analytic posterior.  ``baseline``: infer["baseline"] options for the guide site (the reference
    exercises them on NormalNormalNormalTests, :107-260)."""
    lam0, loc0 = _t([0.1, 0.1], device), _t([0.0, 0.5], device)
    lam = _t([6.0, 4.0], device)
    data = [_t(v, device) for v in ([-0.1, 0.3], [0.00, 0.4], [0.20, 0.5], [0.10, 0.7])]
    n_data = float(len(data))
    sum_data = data[0] + data[1] + data[2] + data[3]
    analytic_lam_n = lam0 + n_data * lam
    analytic_log_sig_n = -0.5 * torch.log(analytic_lam_n)
    analytic_loc_n = sum_data * (lam / analytic_lam_n) + loc0 * (lam0 / analytic_lam_n)
    prior_scale, obs_scale = torch.pow(lam0, -0.5), torch.pow(lam, -0.5)
    Normal = dist.Normal if reparameterized else NonreparameterizedNormal
    pyro.clear_param_store()

    def model():
        with pyro.plate("plate", 2):
            loc_latent = pyro.sample("loc_latent", Normal(loc0, prior_scale))
            for i, x in enumerate(data):
                pyro.sample("obs_%d" % i, dist.Normal(loc_latent, obs_scale), obs=x)
        return loc_latent

    def guide():
        loc_q = pyro.param("loc_q", lambda: analytic_loc_n.detach() + 0.334)
        log_sig_q = pyro.param("log_sig_q", lambda: analytic_log_sig_n.detach() - 0.29)
        with pyro.plate("plate", 2):
            pyro.sample("loc_latent", Normal(loc_q, torch.exp(log_sig_q)),
                        infer={} if baseline is None else {"baseline": dict(baseline)})

    pyro.set_rng_seed(0)
    svi = SVI(model, guide, Adam({"lr": lr, "betas": (0.97, 0.999)}), loss=TraceGraph_ELBO())
    for _ in range(n_steps):
        svi.step()
    loc_error = float(((pyro.param("loc_q").detach() - analytic_loc_n) ** 2).mean())
    log_sig_error = float(((pyro.param("log_sig_q").detach() - analytic_log_sig_n) ** 2).mean())
    assert loc_error < prec and log_sig_error < prec, (loc_error, log_sig_error)
    if baseline is not None and baseline.get("use_decaying_avg_baseline"):
        avg = pyro.get_param_store()["__baseline_avg_downstream_cost_loc_latent"].detach()
        assert avg.shape == (2,) and bool(torch.isfinite(avg).all()) and float(avg.abs().sum()) > 0


class NonreparameterizedGamma(dist.Gamma):
    has_rsample = False


class NonreparameterizedBeta(dist.Beta):
    has_rsample = False


def run_poisson_gamma(device, reparameterized, n_steps, hip_graph=False):
    """tests/infer/test_inference.py:307-402 PoissonGammaTests.do_elbo_test: Gamma(1,1) prior,
    Poisson data [1,2,3]; SVI (Adam lr 2e-4, betas (0.97, 0.999)) from a perturbed start reaches the
    conjugate posterior Gamma(alpha0 + sum, beta0 + n) within 0.2 / 0.15."""
    alpha0, beta0 = _t(1.0, device), _t(1.0, device)
    data = _t([1.0, 2.0, 3.0], device)
    alpha_n, beta_n = alpha0 + data.sum(), beta0 + 3.0
    Gamma = dist.Gamma if reparameterized else NonreparameterizedGamma
    pyro.clear_param_store()

    def model():
        lam = pyro.sample("lambda_latent", Gamma(alpha0, beta0))
        with pyro.plate("data", 3):
            pyro.sample("obs", dist.Poisson(lam), obs=data)

    def guide():
        alpha_q = pyro.param("alpha_q", lambda: alpha_n.detach() + math.exp(0.17),
                             constraint=constraints.positive)
        beta_q = pyro.param("beta_q", lambda: beta_n.detach() / math.exp(0.143),
                            constraint=constraints.positive)
        pyro.sample("lambda_latent", Gamma(alpha_q, beta_q))

    pyro.set_rng_seed(0)
    svi = SVI(model, guide, Adam({"lr": 0.0002, "betas": (0.97, 0.999)}), loss=Trace_ELBO(),
              hip_graph=hip_graph)
    for _ in range(n_steps):
        svi.step()
    a_err = abs(float(pyro.param("alpha_q").detach()) - float(alpha_n))
    b_err = abs(float(pyro.param("beta_q").detach()) - float(beta_n))
    assert a_err < 0.2 and b_err < 0.15, (a_err, b_err)


def run_bernoulli_beta(device, reparameterized, n_steps, vectorized, hip_graph=False):
    """tests/infer/test_inference.py:588-716 BernoulliBetaTests.do_elbo_test (incl. the vectorised
    two-particle variant): Beta(1,1) prior, data [0,1,1,1]; log alpha_q, log beta_q reach the
    conjugate posterior within 0.08."""
    alpha0, beta0 = _t(1.0, device), _t(1.0, device)
    data = _t([0.0, 1.0, 1.0, 1.0], device)
    log_alpha_n = torch.log(alpha0 + data.sum())
    log_beta_n = torch.log(beta0 - data.sum() + 4.0)
    Beta = dist.Beta if reparameterized else NonreparameterizedBeta
    pyro.clear_param_store()

    def model():
        p = pyro.sample("p_latent", Beta(alpha0, beta0))
        with pyro.plate("data", 4):
            pyro.sample("obs", dist.Bernoulli(p), obs=data)

    def guide():
        a = pyro.param("alpha_q_log", lambda: log_alpha_n.detach() + 0.17)
        b = pyro.param("beta_q_log", lambda: log_beta_n.detach() - 0.143)
        pyro.sample("p_latent", Beta(torch.exp(a), torch.exp(b)))

    loss = Trace_ELBO(num_particles=2, vectorize_particles=True, max_plate_nesting=1) \
        if vectorized else Trace_ELBO()
    pyro.set_rng_seed(0)
    svi = SVI(model, guide, Adam({"lr": 0.001, "betas": (0.97, 0.999)}), loss=loss,
              hip_graph=hip_graph)
    for _ in range(n_steps):
        svi.step()
    a_err = abs(float(pyro.param("alpha_q_log").detach() - log_alpha_n))
    b_err = abs(float(pyro.param("beta_q_log").detach() - log_beta_n))
    assert a_err < 0.08 and b_err < 0.08, (a_err, b_err)


def run_elbo_mapdata(device, map_type, batch_size, n_steps, lr):
    """tests/infer/test_elbo_mapdata.py:20-139: conjugate Normal-Normal with the data mapped through a
    sequential plate / a vectorised plate / a plain Python loop, optionally subsampled, the guide
    holding a dummy plate of the same name so that model and guide share the subsample;
    TraceGraph_ELBO + Adam reach the analytic posterior (sum of squared errors 0.05 / 0.06)."""
    lam0, loc0 = _t([0.1, 0.1], device), _t([0.0, 0.5], device)
    lam = _t([6.0, 4.0], device)
    data = _t([[0.1, 0.21], [0.16, 0.11], [0.06, 0.31], [-0.01, 0.07], [0.23, 0.25], [0.19, 0.18],
               [0.09, 0.41], [-0.04, 0.17]], device)
    n = len(data)
    analytic_lam_n = lam0 + float(n) * lam
    analytic_log_sig_n = -0.5 * torch.log(analytic_lam_n)
    analytic_loc_n = data.sum(0) * (lam / analytic_lam_n) + loc0 * (lam0 / analytic_lam_n)
    off = _t([-0.18, 0.23], device)
    pyro.clear_param_store()
    seen = []

    def model():
        loc_latent = pyro.sample("loc_latent", dist.Normal(loc0, torch.pow(lam0, -0.5)).to_event(1))
        obs_scale = torch.pow(lam, -0.5)
        if map_type == "iplate":
            for i in pyro.plate("aaa", n, batch_size):
                pyro.sample("obs_%d" % i, dist.Normal(loc_latent, obs_scale).to_event(1), obs=data[i])
        elif map_type == "plate":
            with pyro.plate("aaa", n, batch_size) as ind:
                seen.append(("model", tuple(int(v) for v in ind)))
                pyro.sample("obs", dist.Normal(loc_latent, obs_scale).to_event(1), obs=data[ind])
        else:
            for i, x in enumerate(data):
                pyro.sample("obs_%d" % i, dist.Normal(loc_latent, obs_scale).to_event(1), obs=x)

    def guide():
        loc_q = pyro.param("loc_q", lambda: analytic_loc_n.detach().clone() + off)
        log_sig_q = pyro.param("log_sig_q", lambda: analytic_log_sig_n.detach().clone() - off)
        pyro.sample("loc_latent", dist.Normal(loc_q, torch.exp(log_sig_q)).to_event(1))
        if map_type == "iplate":
            for i in pyro.plate("aaa", n, batch_size):
                pass
        elif map_type == "plate":
            with pyro.plate("aaa", n, batch_size) as ind:
                seen.append(("guide", tuple(int(v) for v in ind)))

    pyro.set_rng_seed(161)
    svi = SVI(model, guide, Adam({"lr": lr}), loss=TraceGraph_ELBO())
    for _ in range(n_steps):
        svi.step()
    if map_type == "plate":          # the model replays the guide's subsample, step by step
        assert len(seen) == 2 * n_steps
        assert all(seen[2 * k][1] == seen[2 * k + 1][1] for k in range(n_steps))
        if batch_size is not None and batch_size < n:
            assert len({s[1] for s in seen}) > 1
    loc_error = float(((analytic_loc_n - pyro.param("loc_q").detach()) ** 2).sum())
    log_sig_error = float(((analytic_log_sig_n - pyro.param("log_sig_q").detach()) ** 2).sum())
    assert loc_error < 0.05 and log_sig_error < 0.06, (loc_error, log_sig_error)
