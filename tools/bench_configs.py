"""BASELINE configs 4 (LDA, TraceEnum_ELBO) and 5 (hierarchical logistic regression, one GPU's share)
on one MI355X (developer tool; bench.py reports the same numbers under `other_configs`)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples, kernels
from pyro_amd.infer import SVI, Trace_ELBO, TraceEnum_ELBO
from pyro_amd.infer.autoguide import AutoMultivariateNormal, AutoNormal


ROUND = 6       # profiles of THIS round only: a counter file of another round measured other kernels


def _committed_traffic(cfg, kernel):
    """HBM bytes per launch of ``kernel`` from the PMC passes committed THIS round (tools/pmc_cfg.sh ->
    profiles/r%02d_traffic_cfg<cfg>.json), or None: counters cannot be collected from inside the run, and a
    file of an earlier round is refused (its kernels were other kernels)."""
    import json
    import os
    path = _traffic_path(cfg)
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path)).get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:  # noqa: BLE001
        return None


def _traffic_path(cfg):
    import os
    pdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    return os.path.join(pdir, "r%02d_traffic_cfg%s.json" % (ROUND, cfg))


def _traffic_source(cfg):
    import os
    path = _traffic_path(cfg)
    if os.path.exists(path):
        return "profiles/%s (rocprofv3 --pmc, tools/pmc_cfg.sh)" % os.path.basename(path)
    return "none: no profiles/%s committed this round (files of earlier rounds are refused)" % os.path.basename(path)


def timed(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def config5(dev, N=10_000_000, D=32, G=1000, P=64, steps=20, graph=True, reference_text=True):
    """reference_text=True (what bench.py reports): SURVEY 8(d)'s formulation verbatim -- UNSORTED int64
    group ids, logits = (w[..., g, :] * X).sum(-1) + b -- recognised lazily; False: the backend-specific
    spelling dist.grouped_linear_logits on rows pre-sorted by group (rounds 1-3)."""
    if reference_text:
        X, y, gid = examples.synthetic_hier_logreg_data_unsorted(N, D, G, dev, seed=0)
        model, margs = examples.hier_logreg_model_reference, (X, y, gid, G)
    else:
        X, y, off = examples.synthetic_hier_logreg_data(N, D, G, dev, seed=0)
        model, margs = examples.hier_logreg_model, (X, y, kernels.GroupSegments(off, dev))
    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    guide = AutoNormal(model, init_scale=0.1)
    svi = SVI(model, guide, pyro.optim.Adam({"lr": 0.01}),
              Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1),
              hip_graph=graph, graph_warmup=2)
    clock = kernels.GlmDeviceClock(dev)          # before the capture: the pointer is a launch argument
    dt = timed(lambda: svi.step(*margs), steps, 5)
    kms = []
    for _ in range(5):                           # the grouped plane-image kernel inside the captured step
        clock.arm()
        svi.step(*margs)
        torch.cuda.synchronize()
        kms.append(clock.read_ms())
    clock.close()
    kms = [v for v in kms if v == v]
    out = {"steps_per_s": 1 / dt, "ms_per_step": dt * 1e3, "graphed": bool(graph and svi.hip_graph and len(svi._graphs) == 1),
           "model_text": "examples.hier_logreg_model_reference: (w[..., g, :] * X).sum(-1) + b, unsorted int64 g"
           if reference_text else "examples.hier_logreg_model: dist.grouped_linear_logits, rows sorted by group",
           "algorithmic_TBps": N * (4 * D + 4) / dt / 1e12}
    if kms:
        k_ms = sum(kms) / len(kms)
        alg = N * (4 * D + 4)
        f16 = kernels.glm_planes_format() == kernels.GLM_PLANES_F16X2
        out["roofline"] = {"bound": "hbm", "kernel": "glm_planes_f16_kernel<grouped>" if f16 else "glm_planes_kernel<grouped>",
                           "kernel_ms": k_ms, "kernel_ms_source": "device wall-clock stamps inside the captured step "
                                                                  "(pa_glm_planes_stamps), mean of %d replays" % len(kms),
                           "algorithmic_bytes_per_launch": alg, "achieved": alg / (k_ms * 1e-3) / 1e9,
                           "peak": 8000.0, "unit": "GB/s", "frac": alg / (k_ms * 1e-3) / 8e12,
                           "traffic": _committed_traffic(5, "glm_planes_f16_kernel" if f16 else "glm_planes_kernel"),
                           "traffic_source": _traffic_source(5),
                           "share_of_step": k_ms / (dt * 1e3)}
    return out


# An optimizer that leaves the parameters alone: SVI.step is then loss_and_grads (+ the gradient zeroing every
# step needs) -- SURVEY 8(d)'s "loss_and_grads only" figure, captured like the full step and ending in the same
# fused tail kernel (pyro_amd.optim.NoUpdate keeps the flat buffers; until round 6 a host-side no-op object stood
# here, which left the step an unfused tail and a separate zeroing launch: 92 us against the full step's 74).
_NoUpdate = pyro.optim.NoUpdate


def config2_variant(dev, guide="mvn", P=64, N=1_000_000, D=32, steps=50, graph=True, model=None,
                    lazy_matmul=True, elbo=Trace_ELBO, planes_format=None, no_update=False):
    """BASELINE configs[1] with the other guide SURVEY 8(d) names (AutoMultivariateNormal), with
    the reference's default num_particles = 1 (few-particle GLM kernel), with the explicit
    dist.linear_logits model, or with the lazy recognition of w @ X.t() switched off (materialised
    logits: rocBLAS products around the fused site kernels)."""
    from pyro_amd.ops import lazy
    X, y = examples.synthetic_logreg_data(N, D, dev, seed=0)
    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    model = examples.logreg_model if model is None else model
    g = AutoMultivariateNormal(model, init_scale=0.1) if guide == "mvn" else \
        AutoNormal(model, init_scale=0.1)
    svi = SVI(model, g, _NoUpdate() if no_update else pyro.optim.Adam({"lr": 0.01}),
              elbo(num_particles=P, vectorize_particles=True, max_plate_nesting=1),
              hip_graph=graph, graph_warmup=2)
    prev = lazy.ENABLED["on"]
    lazy.ENABLED["on"] = bool(lazy_matmul)
    prev_fmt = kernels.glm_planes_format()
    if planes_format is not None:
        kernels.glm_set_planes_format(planes_format)
    try:
        # (no_update: timed like the headline it is compared with -- past the ~250 slower replays after a capture)
        dt = timed(lambda: svi.step(X, y), max(steps, 300) if no_update else steps, 300 if no_update else 8)
    finally:
        lazy.ENABLED["on"] = prev
        if planes_format is not None:
            kernels.glm_set_planes_format(prev_fmt)
    return {"steps_per_s": 1 / dt, "ms_per_step": dt * 1e3, "num_particles": P, "D": D,
            "graphed": bool(graph and svi.hip_graph and len(svi._graphs) == 1),
            "algorithmic_TBps": N * (4 * D + 4) / dt / 1e12}


def config1(dev, steps=300, graph=True):
    """BASELINE configs[0]: eight schools (examples/eight_schools/svi.py:20-76), Trace_ELBO with one
    particle, Adam lr 0.01 -- every site is a handful of elements: pure launch / host overhead."""
    from pyro_amd import distributions as dist
    from pyro_amd.distributions import constraints
    J = 8
    y = torch.tensor([28.0, 8, -3, 7, -1, 1, 18, 12], device=dev)
    sigma = torch.tensor([15.0, 10, 16, 11, 9, 11, 10, 18], device=dev)
    zJ, oJ = torch.zeros(J, device=dev), torch.ones(J, device=dev)
    z1, o10, o25 = torch.zeros(1, device=dev), 10 * torch.ones(1, device=dev), 25 * torch.ones(1, device=dev)

    def model(y, sigma):
        with pyro.plate("data", J):
            eta = pyro.sample("eta", dist.Normal(zJ, oJ))
            mu = pyro.sample("mu", dist.Normal(z1, o10))
            tau = pyro.sample("tau", dist.HalfCauchy(scale=o25))
            pyro.sample("obs", dist.Normal(mu + tau * eta, sigma), obs=y)

    def guide(y, sigma):
        m_eta = pyro.param("loc_eta", lambda: torch.zeros(J, device=dev))
        s_eta = pyro.param("scale_eta", lambda: 0.1 * torch.ones(J, device=dev), constraint=constraints.positive)
        m_mu = pyro.param("loc_mu", lambda: torch.zeros(1, device=dev))
        s_mu = pyro.param("scale_mu", lambda: 0.1 * torch.ones(1, device=dev), constraint=constraints.positive)
        m_lt = pyro.param("loc_logtau", lambda: torch.zeros(1, device=dev))
        s_lt = pyro.param("scale_logtau", lambda: 0.1 * torch.ones(1, device=dev), constraint=constraints.positive)
        with pyro.plate("data", J):
            pyro.sample("eta", dist.Normal(m_eta, s_eta))
            pyro.sample("mu", dist.Normal(m_mu, s_mu))
            pyro.sample("tau", dist.LogNormal(m_lt, s_lt))

    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    svi = SVI(model, guide, pyro.optim.Adam({"lr": 0.01}), Trace_ELBO(), hip_graph=graph, graph_warmup=2)
    dt = timed(lambda: svi.step(y, sigma), steps, 10)
    return {"steps_per_s": 1 / dt, "us_per_step": dt * 1e6, "last_loss": svi.step(y, sigma),
            "graphed": bool(graph and svi.hip_graph and len(svi._graphs) == 1)}


def config_hmm_vectorised(dev, S=229, L=129, K=16, D=88, steps=50, graph=True):
    """The same HMM likelihood with time vectorised in one DiscreteHMM site (examples/hmm.py
    model_7's construction): observation log-probabilities for all steps at once, the chain summed
    out by one pa_logchain_fwd_bwd launch."""
    seqs, lengths = examples.synthetic_hmm_data(S, L, D, dev)
    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    model = lambda s, l: examples.hmm_model_vectorised(s, l, K)      # noqa: E731
    svi = SVI(model, lambda s, l: None, pyro.optim.Adam({"lr": 0.05}),
              TraceEnum_ELBO(max_plate_nesting=1), hip_graph=graph, graph_warmup=2)
    l0 = svi.step(seqs, lengths)
    dt = timed(lambda: svi.step(seqs, lengths), steps, 3)
    l1 = svi.step(seqs, lengths)
    return {"steps_per_s": 1 / dt, "ms_per_step": dt * 1e3, "loss_first": l0, "loss_last": l1,
            "graphed": bool(graph and svi.hip_graph and len(svi._graphs) == 1)}


def config_hmm(dev, S=229, L=129, K=16, D=88, steps=5, fused=True, graph=False):
    """examples/hmm.py model_1 at the size of its JSB-chorales default (229 sequences, up to 129
    steps, 16 hidden states, 88 tones), TraceEnum_ELBO, Adam: the chain elimination is one
    pa_logchain_fwd_bwd launch (fused) or T pairwise log-space contractions (generic)."""
    import pyro_amd.ops.contract as contract
    seqs, lengths = examples.synthetic_hmm_data(S, L, D, dev)
    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    model = lambda s, l: examples.hmm_model_1(s, l, K)      # noqa: E731
    svi = SVI(model, lambda s, l: None, pyro.optim.Adam({"lr": 0.05}),
              TraceEnum_ELBO(max_plate_nesting=2), hip_graph=graph, graph_warmup=2)
    contract.FUSED_CHAIN = fused
    try:
        l0 = svi.step(seqs, lengths)
        dt = timed(lambda: svi.step(seqs, lengths), steps, 2)
        l1 = svi.step(seqs, lengths)
    finally:
        contract.FUSED_CHAIN = True
    return {"steps_per_s": 1 / dt, "ms_per_step": dt * 1e3, "loss_first": l0, "loss_last": l1,
            "fused_chain": fused, "graphed": bool(graph and svi.hip_graph and len(svi._graphs) == 1)}


def config_gmm(dev, N=1_000_000, K=16, steps=200, leaf=True):
    """A plated Gaussian mixture with the assignment enumerated (TraceEnum_ELBO's other plated pattern beside
    LDA): z_n ~ Categorical(w) summed out, x_n ~ Normal(loc[z_n], 1).  leaf=True: the likelihood never exists as a
    [K, N] tensor (csrc/mixture.hip); False: materialised once and eliminated by pa_logsumexp_terms (round 3)."""
    import pyro_amd.distributions as dist
    import pyro_amd.ops.contract as contract
    from pyro_amd.infer import config_enumerate
    from torch.distributions import constraints
    g = torch.Generator().manual_seed(0)
    data = (torch.randn(N, generator=g) + 3 * torch.randint(0, K, (N,), generator=g).float()).to(dev)

    @config_enumerate
    def model(data):
        w = pyro.sample("w", dist.Dirichlet(torch.ones(K, device=dev)))
        with pyro.plate("k", K):
            loc = pyro.sample("loc", dist.Normal(torch.zeros((), device=dev), 20.0))
        with pyro.plate("n", N):
            z = pyro.sample("z", dist.Categorical(w))
            pyro.sample("x", dist.Normal(loc[z], 1.0), obs=data)

    def guide(data):
        wq = pyro.param("wq", torch.ones(K, device=dev), constraint=constraints.positive)
        lq = pyro.param("lq", 3.0 * torch.arange(K, device=dev, dtype=torch.float32))
        pyro.sample("w", dist.Dirichlet(wq))
        with pyro.plate("k", K):
            pyro.sample("loc", dist.Normal(lq, 0.5))

    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    prev = contract.FUSED_MIXTURE
    contract.FUSED_MIXTURE = bool(leaf)
    try:
        svi = SVI(model, guide, pyro.optim.Adam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=1),
                  hip_graph=True, graph_warmup=3)
        dt = timed(lambda: svi.step(data), steps, 20)
        graphed = bool(svi.hip_graph and len(svi._graphs) == 1)
        last = svi.step(data)
        svi.release()
    finally:
        contract.FUSED_MIXTURE = prev
    return {"steps_per_s": 1 / dt, "ms_per_step": dt * 1e3, "graphed": graphed, "last_loss": last,
            "mixture_leaf_kernel": bool(leaf), "rows_per_s": N / dt}


def config4(dev, docs=100_000, steps=10, batch_size=None):
    """batch_size=None: every document in every step (BASELINE configs[3] as quoted); 32 / 4096: the
    mini-batch variants SURVEY 8(d) lists (examples/lda.py's own default is 32): the sub-sampled
    word matrix is a fresh tensor every step, so the factor takes the LDS-atomic kernel."""
    args = examples.LdaArgs(num_docs=docs)
    data = examples.synthetic_lda_data(args, dev)
    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    predictor = examples.lda_make_predictor(args, dev)
    guide = lambda data, args: examples.lda_guide(predictor, data, args, batch_size)  # noqa: E731
    # examples/lda.py:131 uses ClippedAdam: here the flat fused one (one launch for all parameters,
    # the predictor's weights included)
    svi = SVI(examples.lda_model, guide, pyro.optim.ClippedAdam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=2),
              hip_graph=True, graph_warmup=3)
    dt = timed(lambda: svi.step(data, args), steps, 6)
    pairs = (docs if batch_size is None else batch_size) * args.num_words_per_doc
    out = {"batch_size": batch_size, "steps_per_s": 1 / dt, "ms_per_step": dt * 1e3, "word_doc_pairs_per_s": pairs / dt,
           "graphed": bool(svi.hip_graph and len(svi._graphs) == 1), "last_loss": svi.step(data, args),
           "algorithmic_TBps": pairs * 8.5 / dt / 1e12}
    import os
    if batch_size is None and not os.environ.get("PA_NO_ROOFLINE"):
        try:
            out["roofline"] = _config4_roofline(data, args, predictor, dt)
        except Exception as e:  # noqa: BLE001
            out["roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def _step_composition(cfg):
    """Launch count and the largest kernels of one step from the latest committed one-step trace
    (profiles/rNN_cfg<cfg>_step_trace.txt, tools/trace_cfg.sh): what the step is made of when no single
    kernel dominates it."""
    import glob
    import os
    import re
    from collections import defaultdict
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                          "r*_cfg%d_step_trace.txt" % cfg)))
    if not paths:
        return None
    agg, n, tot, span = defaultdict(lambda: [0, 0.0]), 0, 0.0, None
    for ln in open(paths[-1]):
        m = re.match(r"\s*([\d.]+) us\s+([\d.]+) us\s+(.*)", ln)
        if m:
            name = re.sub(r"^void ", "", m.group(3).strip()).split("(")[0].split("<")[0]
            agg[name][0] += 1
            agg[name][1] += float(m.group(2))
            n += 1
            tot += float(m.group(2))
        m = re.match(r"step span ([\d.]+) us", ln)
        if m:
            span = float(m.group(1))
    top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]
    return {"source": "profiles/" + os.path.basename(paths[-1]) + " (rocprofv3 --kernel-trace of one captured step)",
            "launches": n, "kernel_time_us": round(tot, 1), "step_span_us_under_the_tracer": span,
            "largest": [{"kernel": k, "launches": c, "us": round(t, 1), "share_of_kernel_time": round(t / tot, 3)}
                        for k, (c, t) in top]}


def _config4_roofline(data, args, predictor, dt):
    """No kernel dominates this step (the largest is 7 % of it: ``step_composition``); the block prices
    the first layer of the amortised guide on the corpus's cached bag-of-words image (pa_bow_linear_fwd),
    timed stand-alone on the same operands with HIP events."""
    V = args.num_words
    imgs = kernels.bow_images_of(data, V)
    if imgs is None:
        return None
    lin = [m for m in predictor.modules() if isinstance(m, torch.nn.Linear)][0]
    B = data.shape[1]
    W, bias = lin.weight.detach().contiguous(), lin.bias.detach().contiguous()
    for _ in range(3):
        kernels.bow_linear_fwd(imgs[0], W, bias, B)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        kernels.bow_linear_fwd(imgs[0], W, bias, B)
    e.record()
    torch.cuda.synchronize()
    k_ms = s.elapsed_time(e) / 20
    alg = imgs[0].numel() * 2 + B * W.shape[0] * 4 + W.numel() * 4     # image once, out once, W once
    return {"bound": "hbm", "kernel": "bow_linear_fwd_kernel (+ its split / reduce launches)", "kernel_ms": k_ms,
            "kernel_ms_source": "HIP events around 20 stand-alone calls on the step's operands",
            "algorithmic_bytes_per_launch": alg, "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": 8000.0,
            "unit": "GB/s", "frac": alg / (k_ms * 1e-3) / 8e12,
            "traffic": _committed_traffic(4, "bow_linear_fwd_kernel"),
            "traffic_source": _traffic_source(4),
            "share_of_step": k_ms / (dt * 1e3),
            # the WHOLE step against HBM: what one step must move at least -- the word ids once (int64), the
            # two bag-of-words images (forward and backward pass of the guide's first layer), the guide's
            # activations written and read back once per layer (forward + backward)
            "whole_step": _config4_whole_step(data, args, predictor, imgs, dt),
            "note": "a flat step: no kernel above a tenth of it -- see step_composition",
            "step_composition": _step_composition(4)}


def _config4_whole_step(data, args, predictor, imgs, dt):
    B = data.shape[1]
    widths = [m.out_features for m in predictor.modules() if isinstance(m, torch.nn.Linear)]
    act = sum(widths) * B * 4
    alg = data.numel() * 8 + 2 * sum(int(i.numel()) * i.element_size() for i in imgs) + 4 * act
    return {"algorithmic_bytes_per_step": alg, "achieved": alg / dt / 1e9, "unit": "GB/s", "peak": 8000.0,
            "frac": alg / dt / 8e12,
            "bytes": "word ids %d + bag-of-words images 2 x %d + activations 4 x %d" % (
                data.numel() * 8, sum(int(i.numel()) * i.element_size() for i in imgs), act)}



if __name__ == "__main__":
    dev = torch.device("cuda:0")
    print("config 5 (N=1e7, P=64, G=1000), reference text:", config5(dev))
    print("config 5, grouped_linear_logits on sorted rows:", config5(dev, reference_text=False))
    print("config 5 eager:", config5(dev, steps=5, graph=False))
    print("config 4 (1e5 docs):", config4(dev))
    print("hmm (229 x 129 x 16 states), fused chain:", config_hmm(dev))
    print("hmm, generic elimination:", config_hmm(dev, steps=2, fused=False))
    print("hmm, fused chain, graphed step:", config_hmm(dev, steps=10, graph=True))
    print("hmm, time vectorised (DiscreteHMM), graphed:", config_hmm_vectorised(dev))
    print("hmm, time vectorised (DiscreteHMM), eager:", config_hmm_vectorised(dev, steps=20, graph=False))
    print("config 1 (eight schools):", config1(dev))
    print("config 1 eager:", config1(dev, steps=100, graph=False))
    print("config 2, AutoMultivariateNormal:", config2_variant(dev, "mvn"))
    print("config 2, AutoNormal, num_particles=1:", config2_variant(dev, "normal", P=1))
    print("config 2, AutoNormal, num_particles=4:", config2_variant(dev, "normal", P=4))
