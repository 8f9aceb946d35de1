"""TEST INFRASTRUCTURE -- numpy restatement of one Trace_ELBO loss-and-gradient evaluation for
the BASELINE config-2 model (Bayesian logistic regression, AutoNormal guide, P vectorised
particles), following the reference's chain of operations site by site:

  guide  (pyro/infer/autoguide/guides.py:520-560):
      u_w = loc_w + softplus(rho_w) * eps_w            Normal(...).to_event(1).rsample
      u_b = loc_b + softplus(rho_b) * eps_b
      log q = Normal.log_prob(u).sum()                 (identity transform: Delta log_density 0)
  model  (SURVEY.md 8d config 2):
      log p(w) = Normal(0,1).log_prob(w).sum(-1), log p(b), obs: Bernoulli GLM under plate N
  estimator (pyro/infer/trace_elbo.py:82-159, fully reparameterised => no score-function term):
      loss = -(sum_model log_prob_sum - sum_guide log_prob_sum) / P ; grads = d loss / d params

Parameters are the UNCONSTRAINED store values: loc (real) and rho with scale = softplus(rho)
(AutoNormal.scale_constraint = softplus_positive, guides.py:446).
"""
import numpy as np

from .dists import _sigmoid, _softplus, normal_grad, normal_log_prob
from .glm import glm_bernoulli_fwd_bwd


def logreg_autonormal_loss_and_grads(X, y, loc_w, rho_w, loc_b, rho_b, eps_w, eps_b,
                                     mask=None, scale=1.0):
    """eps_w: [P, D], eps_b: [P]. Returns loss (float) and grads dict for the 4 parameters."""
    X = np.asarray(X, np.float64)
    P = eps_w.shape[0]
    s_w, s_b = _softplus(rho_w), _softplus(rho_b)
    w = loc_w[None, :] + s_w[None, :] * eps_w          # [P, D]
    b = loc_b + s_b * eps_b                            # [P]
    # guide terms
    logq = normal_log_prob(w, loc_w[None], s_w[None]).sum() + normal_log_prob(b, loc_b, s_b).sum()
    # model terms
    logp_prior = normal_log_prob(w, 0.0, 1.0).sum() + normal_log_prob(b, 0.0, 1.0).sum()
    ll, gw, gb = glm_bernoulli_fwd_bwd(X, y, w, b, mask, scale)
    elbo = (logp_prior + ll.sum() - logq) / P
    # d elbo / d w, d b (total derivative through the sample) ------------------------------
    d_w = -w + gw                                       # prior + likelihood
    d_b = -b + gb
    # guide log q depends on (u, loc, scale): d(-logq)/d... ; with u = loc + s*eps:
    dv_w, dl_w, ds_w = normal_grad(w, loc_w[None], s_w[None])
    dv_b, dl_b, ds_b = normal_grad(b, loc_b, s_b)
    # total derivatives of (model - guide) w.r.t. loc and scale
    g_loc_w = (d_w - dv_w - dl_w).sum(0)
    g_s_w = ((d_w - dv_w) * eps_w - ds_w).sum(0)
    g_loc_b = (d_b - dv_b - dl_b).sum()
    g_s_b = ((d_b - dv_b) * eps_b - ds_b).sum()
    dsig_w, dsig_b = _sigmoid(rho_w), _sigmoid(rho_b)   # d softplus / d rho
    grads = {"loc_w": -g_loc_w / P, "rho_w": -g_s_w * dsig_w / P,
             "loc_b": -g_loc_b / P, "rho_b": -g_s_b * dsig_b / P}
    return -elbo, grads
