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
