"""TEST INFRASTRUCTURE -- Adam / ClippedAdam single step restated from torch.optim.Adam
(torch: torch/optim/adam.py _single_tensor_adam) and pyro/optim/clipped_adam.py:52-100."""
import math

import numpy as np


def adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
              clip_norm=0.0, lrd=1.0, clipped=False):
    """`step` is the 1-based index of this step. Returns new (p, m, v)."""
    b1, b2 = betas
    g = np.array(g, dtype=np.float64)
    p = np.array(p, dtype=np.float64)
    if clipped and clip_norm > 0:
        g = np.clip(g, -clip_norm, clip_norm)
    if weight_decay != 0:
        g = g + weight_decay * p
    m = b1 * np.asarray(m, dtype=np.float64) + (1 - b1) * g
    v = b2 * np.asarray(v, dtype=np.float64) + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    if clipped:
        lr_t = lr * lrd ** step
        p = p - lr_t * math.sqrt(bc2) / bc1 * m / (np.sqrt(v) + eps)
    else:
        p = p - lr / bc1 * m / (np.sqrt(v) / math.sqrt(bc2) + eps)
    return p, m, v
