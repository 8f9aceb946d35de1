"""TEST INFRASTRUCTURE -- the enumerated mixture factor of examples/lda.py:53-71 restated in
numpy, following the reference's own chain of operations:
  word_topics enumerated as arange(T) on a new leftmost dim   enum_messenger.py:114-231
  Categorical(topic_words[word_topics]).log_prob(data)        -> b[t, w, d]   (gather)
  Categorical(doc_topics).log_prob(word_topics)               -> a[t, 1, d]
  sum-product over t in log space (max-shift / exp / sum / log)  ops/einsum/torch_log.py:14-55
  plate products = sums over words, documents                 ops/contract.py:79-160
"""
import numpy as np


def lda_factor(words, log_theta, log_phi):
    """words int64 [Wd,B]; log_theta [B,T]; log_phi [T,V].
    Returns out_doc[B], g_theta[B,T], g_phi[T,V] (float64)."""
    words = np.asarray(words)
    log_theta = np.asarray(log_theta, dtype=np.float64)
    log_phi = np.asarray(log_phi, dtype=np.float64)
    T, V = log_phi.shape
    Wd, B = words.shape
    # enumerated gather indices are integers: exact by construction
    a = log_theta.T[:, None, :]                      # [T,1,B]
    b = log_phi[:, words]                            # [T,Wd,B]
    s = a + b
    shift = s.max(0, keepdims=True)
    shift = np.where(np.isfinite(shift), shift, 0.0)  # torch_log.py: clamp(min=finfo.min)
    e = np.exp(s - shift)
    tot = e.sum(0, keepdims=True)
    with np.errstate(divide="ignore"):
        lse = np.log(tot) + shift                    # [1,Wd,B]
    out_doc = lse[0].sum(0)
    with np.errstate(invalid="ignore", divide="ignore"):
        post = np.where(tot > 0, e / tot, 0.0)       # [T,Wd,B]
    g_theta = post.sum(1).T                          # [B,T]
    g_phi = np.zeros((T, V))
    for t in range(T):
        np.add.at(g_phi[t], words.reshape(-1), post[t].reshape(-1))
    return out_doc, g_theta, g_phi


LDA_SEG = 2048      # pairs per word-major task (pyro_amd/csrc/lda.hip)


def lda_word_index(words, V):
    """The inverted index of a corpus (include/pyro_amd.h, pa_lda_build_index): for every word v the
    documents of the pairs (w, d) that hold it, in ascending order of w * B + d, and the cut of the
    lists into tasks of <= LDA_SEG pairs.  Returns (off[V+1], first_task[V+1], task_v, task_start,
    task_len, docs) as int32 arrays.  Ids outside [0, V) are filed under 0."""
    words = np.asarray(words)
    Wd, B = words.shape
    flat = words.reshape(-1).astype(np.int64)
    flat = np.where((flat < 0) | (flat >= V), 0, flat)
    order = np.argsort(flat, kind="stable")
    docs = (order % max(B, 1)).astype(np.int32)
    counts = np.bincount(flat, minlength=V)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    ntask = (counts + LDA_SEG - 1) // LDA_SEG
    first_task = np.concatenate([[0], np.cumsum(ntask)]).astype(np.int32)
    task_v, task_start, task_len = [], [], []
    for v in range(V):
        for st in range(off[v], off[v + 1], LDA_SEG):
            task_v.append(v)
            task_start.append(st)
            task_len.append(min(LDA_SEG, off[v + 1] - st))
    i32 = lambda x: np.asarray(x, dtype=np.int32)   # noqa: E731
    return off, first_task, i32(task_v), i32(task_start), i32(task_len), docs


def bow_counts(words, V):
    """The dense histogram counts[v, b] of examples/lda.py:116-119 (zeros(V, B).scatter_add(0, data,
    ones)), float64."""
    words = np.asarray(words)
    Wd, B = words.shape
    c = np.zeros((V, B))
    np.add.at(c, (words, np.broadcast_to(np.arange(B), words.shape)), 1.0)
    return c


def bow_images(words, V):
    """(image_a, image_b) of pyro_amd/kernels.py::bow_images as float arrays [blocks.., 64 lanes, 8]:
    image_a block (mt, kt), lane l -> document 32 mt + (l & 31), words 16 kt + 8 (l >> 5) + 0..7;
    image_b block (vt, kt), lane l -> word 32 vt + (l & 31), documents 16 kt + 8 (l >> 5) + 0..7."""
    c = bow_counts(words, V)
    B = c.shape[1]
    Bp = -(-B // 32) * 32
    cp = np.zeros((V, Bp))
    cp[:, :B] = c
    a = np.zeros((Bp // 32, V // 16, 64, 8))
    b = np.zeros((V // 32, Bp // 16, 64, 8))
    lane = np.arange(64)
    for q in range(8):
        # image_a[mt, kt, l, q] = cp[16 kt + 8 (l >> 5) + q, 32 mt + (l & 31)]
        a[:, :, :, q] = cp[(16 * np.arange(V // 16)[None, :, None] + 8 * (lane >> 5)[None, None, :] + q),
                           (32 * np.arange(Bp // 32)[:, None, None] + (lane & 31)[None, None, :])]
        b[:, :, :, q] = cp[(32 * np.arange(V // 32)[:, None, None] + (lane & 31)[None, None, :]),
                           (16 * np.arange(Bp // 16)[None, :, None] + 8 * (lane >> 5)[None, None, :] + q)]
    return a, b


def bow_linear(words, V, W, bias):
    """h = counts^T W^T + bias, float64 (the first nn.Linear of examples/lda.py:76-92 on the
    transposed histogram)."""
    return bow_counts(words, V).T @ np.asarray(W, dtype=np.float64).T + (0.0 if bias is None else bias)


def bow_linear_grad(words, V, d_out):
    """(dW, dbias) of bow_linear for the upstream gradient d_out [B, H]."""
    d = np.asarray(d_out, dtype=np.float64)
    return d.T @ bow_counts(words, V).T, d.sum(0)



# ---- csrc/tall.hip: a Linear layer over a tall batch (the inner layers of examples/lda.py:76-92's predictor),
# restated in float64: torch.nn.functional.linear and autograd's products for it
def tall_linear(x, weight, bias=None):
    """F.linear(x, weight, bias) = x weight^T + bias (torch/nn/functional.py linear)."""
    out = np.asarray(x, np.float64) @ np.asarray(weight, np.float64).T
    return out if bias is None else out + np.asarray(bias, np.float64)


def tall_linear_grads(x, weight, g):
    """(dx, dW, db) of F.linear for the upstream gradient g: g weight, g^T x, g.sum(0)."""
    x, weight, g = (np.asarray(a, np.float64) for a in (x, weight, g))
    return g @ weight, g.T @ x, g.sum(0)
