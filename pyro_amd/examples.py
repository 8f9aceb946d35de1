"""Benchmark / example programs written against the drop-in API (BASELINE.json configs)."""
import torch

from . import distributions as dist
from .primitives import plate, sample


_ZEROS = {}


def _zeros(shape, like):
    """Prior-location constants hoisted out of the model body (what a user does with
    ``loc = X.new_zeros(D)`` above the model): allocated and filled once, not once per step."""
    key = (tuple(shape), like.dtype, like.device)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(shape, dtype=like.dtype, device=like.device)
    return z


def logreg_model(X, y):
    """BASELINE config 2, the model text of SURVEY 8(d) as a Pyro user writes it (broadcast-safe
    under the vectorised-particle plate).  Nothing in it is specific to this backend: the matmul of
    the latent ``w`` with the constant design matrix is recognised lazily (ops/lazy.py) and the
    observed site runs the fused one-pass GLM kernel."""
    N, D = X.shape
    w = sample("w", dist.Normal(X.new_zeros(D), 1.0).to_event(1))
    b = sample("b", dist.Normal(X.new_zeros(()), 1.0))
    with plate("data", N):
        logits = w @ X.t()
        logits = logits.squeeze(-2) if logits.dim() > 1 else logits
        sample("obs", dist.Bernoulli(logits=logits + b), obs=y)


def logreg_model_explicit(X, y):
    """The same model with the lazy logits spelled out (``dist.linear_logits``, an API the
    reference does not have) and the prior constants hoisted: what the recognition above saves
    the user from writing.  Same kernels, two fill launches fewer per step."""
    N, D = X.shape
    w = sample("w", dist.Normal(_zeros((D,), X), 1.0).to_event(1))
    b = sample("b", dist.Normal(_zeros((), X), 1.0))
    with plate("data", N):
        sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, b)), obs=y)


logreg_model_unfused = logreg_model      # materialised logits when ops.lazy.ENABLED["on"] is False


def synthetic_logreg_data(N, D, device, seed=0, dtype=torch.float32):
    g = torch.Generator(device=device).manual_seed(seed)
    X = torch.randn((N, D), device=device, dtype=dtype, generator=g)
    w_true = torch.randn((D,), device=device, dtype=dtype, generator=g)
    y = (torch.rand((N,), device=device, dtype=dtype, generator=g) < torch.sigmoid(X @ w_true)).to(dtype)
    return X, y


def correlated_gaussian_precision(D, seed=0, dtype=torch.float64):
    """BASELINE config 3: Sigma = A A^T / D + 0.1 I, Lambda = Sigma^-1 (built in float64)."""
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((D, D), dtype=torch.float64, generator=g)
    Sigma = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64)
    Lam = torch.linalg.inv(Sigma)
    return Sigma.to(dtype), (0.5 * (Lam + Lam.T)).to(dtype)


# ---- BASELINE config 4: examples/lda.py:42-122 restated against the drop-in API ----------------
class LdaArgs:
    def __init__(self, num_topics=8, num_words=1024, num_docs=1000, num_words_per_doc=64,
                 layer_sizes="100-100"):
        self.num_topics, self.num_words, self.num_docs = num_topics, num_words, num_docs
        self.num_words_per_doc, self.layer_sizes = num_words_per_doc, layer_sizes


def lda_model(data=None, args=None, batch_size=None, device=None):
    """examples/lda.py:42-70: word_topics is enumerated in parallel and summed out exactly."""
    dev = data.device if data is not None else device
    with plate("topics", args.num_topics):
        topic_weights = sample("topic_weights", dist.Gamma(
            torch.full((), 1.0 / args.num_topics, device=dev), torch.ones((), device=dev)))
        topic_words = sample("topic_words", dist.Dirichlet(
            torch.ones(args.num_words, device=dev) / args.num_words))
    with plate("documents", args.num_docs, device=dev) as ind:
        if data is not None:
            assert data.shape == (args.num_words_per_doc, args.num_docs)
            if ind.shape[0] != args.num_docs:
                data = data[:, ind]
        doc_topics = sample("doc_topics", dist.Dirichlet(topic_weights))
        with plate("words", args.num_words_per_doc):
            word_topics = sample("word_topics", dist.Categorical(doc_topics),
                                 infer={"enumerate": "parallel"})
            data = sample("doc_words", dist.Categorical(topic_words[word_topics]), obs=data)
    return topic_weights, topic_words, data


def lda_make_predictor(args, device):
    """examples/lda.py:75-90: MLP from word histograms to topic proportions."""
    import torch.nn as nn
    sizes = [args.num_words] + [int(s) for s in args.layer_sizes.split("-")] + [args.num_topics]
    layers = []
    for a, b in zip(sizes, sizes[1:]):
        layer = nn.Linear(a, b)
        layer.weight.data.normal_(0, 0.001)
        layer.bias.data.normal_(0, 0.001)
        layers += [layer, nn.Sigmoid()]
    layers.append(nn.Softmax(dim=-1))
    return nn.Sequential(*layers).to(device)


def lda_guide(predictor, data, args, batch_size=None):
    """examples/lda.py:93-122: conjugate global factors + amortised Delta for doc_topics."""
    from .distributions import constraints
    from .primitives import module, param
    dev = data.device
    topic_weights_posterior = param("topic_weights_posterior",
                                    lambda: torch.ones(args.num_topics, device=dev),
                                    constraint=constraints.positive)
    topic_words_posterior = param("topic_words_posterior",
                                  lambda: torch.ones(args.num_topics, args.num_words, device=dev),
                                  constraint=constraints.greater_than(0.5))
    with plate("topics", args.num_topics):
        sample("topic_weights", dist.Gamma(topic_weights_posterior, torch.ones((), device=dev)))
        sample("topic_words", dist.Dirichlet(topic_words_posterior))
    module("predictor", predictor)
    with plate("documents", args.num_docs, batch_size, device=dev) as ind:
        if ind.shape[0] != args.num_docs:
            data = data[:, ind]
        counts = torch.zeros(args.num_words, data.shape[1], device=dev).scatter_add(
            0, data, torch.ones(data.shape, device=dev))
        doc_topics = predictor(counts.transpose(0, 1))
        sample("doc_topics", dist.Delta(doc_topics, event_dim=1))


def synthetic_lda_data(args, device, seed=0):
    """Documents drawn from a random topic model (int64 word ids [words_per_doc, num_docs])."""
    g = torch.Generator(device=device).manual_seed(seed)
    T, V, D, W = args.num_topics, args.num_words, args.num_docs, args.num_words_per_doc
    phi = torch.softmax(3.0 * torch.randn((T, V), device=device, generator=g), -1)
    theta = torch.softmax(2.0 * torch.randn((D, T), device=device, generator=g), -1)
    z = torch.multinomial(theta, W, replacement=True, generator=g)          # [D, W]
    cdf = phi.cumsum(-1)
    u = torch.rand((D, W), device=device, generator=g)
    words = (u.unsqueeze(-1) > cdf[z]).sum(-1).clamp(max=V - 1)            # inverse-CDF draw
    return words.t().contiguous()


# ---- BASELINE config 5: hierarchical logistic regression ---------------------------
def hier_logreg_model(X, y, segments, plate_scale=1.0):
    """mu ~ N(0,1)^D, tau ~ HalfNormal(1)^D, w_g ~ N(mu, tau) for g in plate(groups),
    obs_n ~ Bernoulli(logits = x_n . w_{g(n)} + b); rows of X sorted by group.
    ``plate_scale``: this process holds 1 / plate_scale of the plate's rows (a data-sharded run,
    SURVEY 8e variant 2): its likelihood is scaled to the full plate, as plate(size, subsample_size)
    does in examples/svi_horovod.py:52-59, and the mean of the ranks' gradients is the full-data one."""
    N, D = X.shape
    G = segments.G
    z = torch.zeros(D, dtype=X.dtype, device=X.device)
    mu = sample("mu", dist.Normal(z, 1.0).to_event(1))
    tau = sample("tau", dist.HalfNormal(torch.ones(D, dtype=X.dtype, device=X.device)).to_event(1))
    b = sample("b", dist.Normal(torch.zeros((), dtype=X.dtype, device=X.device), 1.0))
    with plate("groups", G):
        w = sample("w", dist.Normal(mu, tau).to_event(1))
    from . import poutine
    with poutine.scale(scale=float(plate_scale)), plate("data", N):
        sample("obs", dist.Bernoulli(logits=dist.grouped_linear_logits(X, w, b, segments)), obs=y)


def hier_prior_logreg_model(X, y):
    """Logistic regression with a LEARNED prior scale: tau ~ HalfNormal(1), w ~ N(0, tau)^D, b ~ N(0, 1),
    obs_n ~ Bernoulli(logits = x_n . w + b).  The smallest well-identified model whose site parameter is another
    latent site's value (BASELINE config 5's w_g ~ N(mu, tau) has the same structure over G groups): under NUTS
    the direct potential holds it (infer/mcmc/direct.py, parent-valued parameters).  ``tau.unsqueeze(-1)``: the
    broadcast-safe text vectorised chains / particles need (a scalar site against an event dimension)."""
    N, D = X.shape
    tau = sample("tau", dist.HalfNormal(X.new_ones(())))
    b = sample("b", dist.Normal(X.new_zeros(()), 1.0))
    w = sample("w", dist.Normal(X.new_zeros(D), tau.unsqueeze(-1)).to_event(1))
    with plate("data", N):
        sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, b)), obs=y)


def hier_logreg_model_reference(X, y, g, G, plate_scale=1.0):
    """SURVEY 8(d) config 5 exactly as the reference would write it: g = int64 [N] group ids in ANY
    order, logit_n = x_n . w_{g_n} + b through an advanced-index gather.  Nothing here names the
    backend: the gather of a latent is recognised lazily (ops/lazy.py::DeferredGroupDot) and the
    observed site runs the grouped plane-image kernel; with the recognition switched off the same
    text runs operator by operator."""
    N, D = X.shape
    z = torch.zeros(D, dtype=X.dtype, device=X.device)
    mu = sample("mu", dist.Normal(z, 1.0).to_event(1))
    tau = sample("tau", dist.HalfNormal(torch.ones(D, dtype=X.dtype, device=X.device)).to_event(1))
    b = sample("b", dist.Normal(torch.zeros((), dtype=X.dtype, device=X.device), 1.0))
    with plate("groups", G):
        w = sample("w", dist.Normal(mu, tau).to_event(1))
    from . import poutine
    with poutine.scale(scale=float(plate_scale)), plate("data", N):
        logits = (w[..., g, :] * X).sum(-1) + b
        sample("obs", dist.Bernoulli(logits=logits), obs=y)


def hier_logreg_model_unfused(X, y, segments):
    """Same model in the reference's formulation: gather the per-row weights and reduce."""
    N, D = X.shape
    G = segments.G
    z = torch.zeros(D, dtype=X.dtype, device=X.device)
    mu = sample("mu", dist.Normal(z, 1.0).to_event(1))
    tau = sample("tau", dist.HalfNormal(torch.ones(D, dtype=X.dtype, device=X.device)).to_event(1))
    b = sample("b", dist.Normal(torch.zeros((), dtype=X.dtype, device=X.device), 1.0))
    with plate("groups", G):
        w = sample("w", dist.Normal(mu, tau).to_event(1))
    import numpy as np
    off = segments.group_offsets
    g_of = torch.as_tensor(np.repeat(np.arange(G), np.diff(off)), device=X.device)
    with plate("data", N):
        logits = (w[..., g_of, :] * X).sum(-1) + b
        sample("obs", dist.Bernoulli(logits=logits), obs=y)


def synthetic_hier_logreg_data(N, D, G, device, seed=0, dtype=torch.float32):
    """Rows sorted by group; returns X, y, group offsets (host int64 [G+1])."""
    import numpy as np
    g = torch.Generator(device=device).manual_seed(seed)
    X = torch.randn((N, D), device=device, dtype=dtype, generator=g)
    grp = torch.randint(0, G, (N,), device=device, generator=g).sort()[0]
    mu = torch.randn((D,), device=device, dtype=dtype, generator=g)
    wg = mu + 0.5 * torch.randn((G, D), device=device, dtype=dtype, generator=g)
    logits = (wg[grp] * X).sum(-1)
    y = (torch.rand((N,), device=device, dtype=dtype, generator=g) < torch.sigmoid(logits)).to(dtype)
    counts = torch.bincount(grp, minlength=G).cpu().numpy().astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)])
    return X, y, offsets


def synthetic_hier_logreg_data_unsorted(N, D, G, device, seed=0, dtype=torch.float32):
    """SURVEY 8(d) config 5's data: g = randint(0, G, (N,)) left in its random order; returns X, y, g."""
    g = torch.Generator(device=device).manual_seed(seed)
    X = torch.randn((N, D), device=device, dtype=dtype, generator=g)
    grp = torch.randint(0, G, (N,), device=device, generator=g)
    mu = torch.randn((D,), device=device, dtype=dtype, generator=g)
    wg = mu + 0.5 * torch.randn((G, D), device=device, dtype=dtype, generator=g)
    y = torch.empty((N,), device=device, dtype=dtype)
    step = 1 << 20                                   # (the gathered weights of 1e7 rows are 1.3 GB)
    for lo in range(0, N, step):
        hi = min(N, lo + step)
        logits = (wg[grp[lo:hi]] * X[lo:hi]).sum(-1)
        y[lo:hi] = (torch.rand((hi - lo,), device=device, dtype=dtype, generator=g)
                    < torch.sigmoid(logits)).to(dtype)
    return X, y, grp


# ---- examples/hmm.py:97-137 (model_1) restated against the drop-in API ---------------------------
def hmm_model_1(sequences, lengths, hidden_dim=16):
    """One hidden chain per sequence under pyro.markov, emissions in a nested plate, ragged
    sequences masked; transition / emission probabilities are learnable parameters (maximum
    likelihood; the example's priors + AutoDelta guide add two small sites)."""
    from . import markov, poutine
    from .distributions import constraints
    from .primitives import param
    S, L, D = sequences.shape
    dev = sequences.device
    probs_x = param("probs_x", lambda: torch.softmax(torch.randn(hidden_dim, hidden_dim, device=dev), -1),
                    constraint=constraints.simplex)
    probs_y = param("probs_y", lambda: torch.rand(hidden_dim, D, device=dev) * 0.8 + 0.1,
                    constraint=constraints.unit_interval)
    tones_plate = plate("tones", D, dim=-1)
    with plate("sequences", S, dim=-2):
        x = 0
        for t in markov(range(L)):
            with poutine.mask(mask=(t < lengths).unsqueeze(-1)):
                x = sample("x_{}".format(t), dist.Categorical(probs_x[x]),
                           infer={"enumerate": "parallel"})
                with tones_plate:
                    sample("y_{}".format(t), dist.Bernoulli(probs_y[x.squeeze(-1)]),
                           obs=sequences[:, t])


def hmm_model_vectorised(sequences, lengths, hidden_dim=16):
    """The same likelihood with time vectorised (examples/hmm.py:580-611 model_7's construction):
    one DiscreteHMM site per sequence batch, ragged lengths through a masked observation
    distribution -- the whole marginal likelihood is one forward-backward launch."""
    from .distributions import constraints
    from .primitives import param
    S, L, D = sequences.shape
    dev = sequences.device
    probs_x = param("probs_x", lambda: torch.softmax(torch.randn(hidden_dim, hidden_dim, device=dev), -1),
                    constraint=constraints.simplex)
    probs_y = param("probs_y", lambda: torch.rand(hidden_dim, D, device=dev) * 0.8 + 0.1,
                    constraint=constraints.unit_interval)
    with plate("sequences", S, dim=-1):
        t = torch.arange(L, device=dev)
        # the chain starts in state 0 (built on the device: a host-side element assignment would
        # be a host-to-device copy, which a captured step cannot contain)
        init_logits = torch.zeros(hidden_dim, device=dev, dtype=probs_x.dtype).masked_fill(
            torch.arange(hidden_dim, device=dev) > 0, -float("inf"))
        obs_dist = dist.Bernoulli(probs_y).to_event(1)                    # batch [K]
        obs_dist = obs_dist.mask((t < lengths.unsqueeze(-1)).unsqueeze(-1))   # batch [S, L, K]
        hmm = dist.DiscreteHMM(init_logits, probs_x.log(), obs_dist)
        sample("y", hmm, obs=sequences)


def synthetic_hmm_data(S, L, D, device, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    seqs = (torch.rand((S, L, D), device=device, generator=g) < 0.3).float()
    lengths = torch.randint(L // 2, L + 1, (S,), device=device, generator=g)
    lengths[0] = L
    return seqs, lengths
