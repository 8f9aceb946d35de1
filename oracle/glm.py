"""TEST INFRASTRUCTURE -- numpy restatement of the plated Bernoulli-logits GLM site and of the
Trace_ELBO surrogate for the BASELINE config-2 model (Bayesian logistic regression).

Reference chain of operations restated here (P vectorised particles, plate size N):
  logits = w @ X^T + b                     user model
  log p = Bernoulli(logits).log_prob(y)    torch: torch/distributions/bernoulli.py:121-125
  scale_and_mask, .sum()                   pyro/poutine/trace_struct.py:264-278,
                                           pyro/distributions/util.py:311-328
  backward                                 pyro/infer/trace_elbo.py:153-157
"""
import numpy as np

from .dists import _sigmoid, _softplus


def glm_bernoulli_fwd_bwd(X, y, w, b=None, mask=None, scale=1.0):
    """Returns ll[P], gw[P,D], gb[P] in float64."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    logits = w @ X.T
    if b is not None:
        logits = logits + np.asarray(b, dtype=np.float64)[:, None]
    lp = y[None, :] * logits - _softplus(logits)
    g = y[None, :] - _sigmoid(logits)
    if mask is not None:
        m = np.asarray(mask, dtype=bool)[None, :]
        lp = np.where(m, lp, 0.0)
        g = np.where(m, g, 0.0)
    return scale * lp.sum(1), scale * (g @ X), scale * g.sum(1)


def glm_bernoulli_grouped_fwd_bwd(X, y, w, group_of_row, b=None, mask=None, scale=1.0):
    """Hierarchical variant: logit[p, n] = w[p, g(n), :] . x_n + b[p].
    w: [P, G, D].  Returns ll[P], gw[P, G, D], gb[P] in float64."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    g_of = np.asarray(group_of_row)
    logits = np.einsum("pnd,nd->pn", w[:, g_of, :], X)
    if b is not None:
        logits = logits + np.asarray(b, dtype=np.float64)[:, None]
    lp = y[None, :] * logits - _softplus(logits)
    g = y[None, :] - _sigmoid(logits)
    if mask is not None:
        m = np.asarray(mask, dtype=bool)[None, :]
        lp = np.where(m, lp, 0.0)
        g = np.where(m, g, 0.0)
    gw = np.zeros_like(w)
    for grp in range(w.shape[1]):
        sel = g_of == grp
        gw[:, grp, :] = g[:, sel] @ X[sel]
    return scale * lp.sum(1), scale * gw, scale * g.sum(1)


# ---- the plane image of pa_glm_pack_planes (include/pyro_amd.h), restated in numpy ---------------
def _bf16_round(x):
    """f32 array -> f32 array holding the nearest bf16 (round to nearest even), as
    v_cvt_pk_bf16_f32 does for finite inputs."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_split3(x):
    """x (f32) -> (x1, x2, x3) bf16-valued f32 arrays with x1 + x2 + x3 == x exactly (finite x
    away from the bf16 overflow / f32 underflow ranges): the residuals are computed in f32,
    where they are exact."""
    x = np.asarray(x, dtype=np.float32)
    x1 = _bf16_round(x)
    r = (x - x1).astype(np.float32)
    x2 = _bf16_round(r)
    x3 = _bf16_round((r - x2).astype(np.float32))
    return x1, x2, x3


def glm_plane_image(X):
    """uint16 image [tiles, 3 planes, 1024] of an [N, D <= 32] f32 design matrix: per 32-row tile
    three [32 rows][32 cols] bf16 planes, the four 16-byte slots (8 columns) of row r stored at slot
    s ^ ((r >> 2) & 3); rows >= N and columns >= D are zero; whole 128-row groups."""
    X = np.asarray(X, dtype=np.float32)
    N, D = X.shape
    assert D <= 32
    tiles = -(-max(N, 0) // 32)
    tiles = -(-tiles // 4) * 4
    Xp = np.zeros((tiles * 32, 32), dtype=np.float32)
    Xp[:N, :D] = X
    img = np.zeros((tiles, 3, 32, 4, 8), dtype=np.uint16)
    r = np.arange(32)
    for pl, piece in enumerate(bf16_split3(Xp)):
        bits = (piece.view(np.uint32) >> 16).astype(np.uint16).reshape(tiles, 32, 4, 8)
        for s in range(4):
            img[:, pl, r, s ^ ((r >> 2) & 3), :] = bits[:, r, s, :]
    return img.reshape(tiles, 3, 1024)


# ---- the two-plane scaled f16 image (format PA_GLM_PLANES_F16X2, csrc/glm_planes16.h) -----------
def label_moments(X, y):
    """float64[33] = {sum_n (y_n - 1/2) X[n, d] (0 beyond D), sum_n (y_n - 1/2)}: the data moments of
    the label-linear part of the Bernoulli-logits log-likelihood, sum_n (y_n - 1/2) (x_n . w + b) =
    c . w + c0 b (pa_glm_label_moments)."""
    X = np.asarray(X, dtype=np.float64)
    yh = np.asarray(y, dtype=np.float64) - 0.5
    out = np.zeros(33)
    out[:X.shape[1]] = yh @ X
    out[32] = yh.sum()
    return out


def _f16_exponent_of_bits(bits):
    e = (bits >> 23) & 0xFF
    if bits == 0 or e == 0xFF:
        return 0
    return 14 - (-127 if e == 0 else e - 127)


def f16_image_exponent(X):
    """kx[32] of the image, one exponent per COLUMN: max_n |X[n,d]| * 2^kx[d] in [2^14, 2^15); 0 for
    an all-zero or non-finite column and for the padding columns d >= D (glmh_exponent_of on
    glm_absmax_kernel's column maxima: the unsigned maximum of the magnitudes' bit patterns, NaN
    above inf)."""
    X = np.asarray(X, dtype=np.float32)
    # (images of more than 32 columns -- feature tiles, csrc/glm_planes16d.h -- carry 128 exponents)
    kx = np.zeros(32 if X.ndim < 2 or X.shape[1] <= 32 else 128, dtype=np.int32)
    if X.size == 0:
        return kx
    bits = (np.ascontiguousarray(X).view(np.uint32) & np.uint32(0x7FFFFFFF)).max(axis=0)
    for d, bb in enumerate(bits):
        kx[d] = _f16_exponent_of_bits(int(bb))
    return kx


def f16_split2(x):
    """x (f32) -> (x1, x2) f16-valued f32 arrays, x1 = RN_f16(x), x2 = RN_f16(x - x1) (the residual
    is exact in f32): x1 + x2 = x to 2^-22 |x| while x2 is a normal f16."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        x1 = x.astype(np.float16)
        r = (x - x1.astype(np.float32)).astype(np.float32)
        x2 = r.astype(np.float16)
    return x1, x2


def glm_plane_image_f16(X, kx=None):
    """(uint16 image [tiles * DT, 2 planes, 1024], kx[32 or 128]) of an [N, D <= 128] f32 design matrix:
    glm_plane_image's tile geometry with the two f16 pieces of X[:, d] * 2^kx[d], DT = 1 / 2 / 4 feature
    tiles of 32 columns per 32-row tile."""
    X = np.asarray(X, dtype=np.float32)
    N, D = X.shape
    assert D <= 128
    DT = 1 if D <= 32 else (2 if D <= 64 else 4)        # feature tiles of 32 columns
    if kx is None:
        kx = f16_image_exponent(X)
    tiles = -(-max(N, 0) // 32)
    tiles = -(-tiles // 4) * 4
    Xp = np.zeros((tiles * 32, 32 * DT), dtype=np.float32)
    Xp[:N, :D] = np.ldexp(X, np.asarray(kx)[None, :D]).astype(np.float32)
    # tile T's sub-tile dt (columns 32 dt .. 32 dt + 31) is image block T * DT + dt
    img = np.zeros((tiles, DT, 2, 32, 4, 8), dtype=np.uint16)
    r = np.arange(32)
    for pl, piece in enumerate(f16_split2(Xp)):
        bits = piece.view(np.uint16).reshape(tiles, 32, DT, 4, 8)
        for dt in range(DT):
            for s in range(4):
                img[:, dt, pl, r, s ^ ((r >> 2) & 3), :] = bits[:, r, dt, s, :]
    return img.reshape(tiles * DT, 2, 1024), kx


def glm_grouped_plane_image_f16(X, y, seg):
    """(uint16 tile image, float32 padded observations, kx): glm_grouped_plane_image in the f16 format;
    the column exponents come from the whole of X."""
    X = np.asarray(X, dtype=np.float32)
    y = np.asarray(y, dtype=np.float32)
    D = X.shape[1]
    kx = f16_image_exponent(X)
    blocks, yb = [], []
    for a, e, _ in np.asarray(seg).reshape(-1, 3):
        a, e = int(a), int(e)
        rows = e - a
        pad = -(-rows // 64) * 64
        blk = np.zeros((pad, D), dtype=np.float32)
        blk[:rows] = X[a:e]
        yv = np.zeros(pad, dtype=np.float32)
        # as the kernel consumes them: 2^14 (y - 1/2) (one rounding, an fma), 0 in the padding
        yv[:rows] = (y[a:e].astype(np.float64) * 16384.0 - 8192.0).astype(np.float32)
        blocks.append(blk)
        yb.append(yv)
    if not blocks:
        return np.zeros((0, 2, 1024), dtype=np.uint16), np.zeros(0, dtype=np.float32), kx
    Xp = np.concatenate(blocks)
    img = glm_plane_image_f16(Xp, kx)[0][: Xp.shape[0] // 32]
    return img, np.concatenate(yb), kx


def glm_grouped_plane_image(X, y, seg):
    """(uint16 tile image, float32 padded observations) of pa_glm_pack_planes_grouped: ``seg`` =
    rows [a, e) of one group each (kernels.GroupSegments.seg); every segment starts on a 64-row
    super-tile boundary, its last super-tile is padded with zero rows; the tile format is
    glm_plane_image's.  Restates the layout rule of the reference-side gather w[..., g_n, :] for
    rows sorted by group: nothing of the reference is numeric here, the image
    is a bit-exact re-arrangement of X's exact bf16 split."""
    X = np.asarray(X, dtype=np.float32)
    y = np.asarray(y, dtype=np.float32)
    D = X.shape[1]
    blocks, yb = [], []
    for a, e, _ in np.asarray(seg).reshape(-1, 3):
        a, e = int(a), int(e)
        rows = e - a
        pad = -(-rows // 64) * 64
        blk = np.zeros((pad, D), dtype=np.float32)
        blk[:rows] = X[a:e]
        yv = np.zeros(pad, dtype=np.float32)
        yv[:rows] = y[a:e]
        blocks.append(blk)
        yb.append(yv)
    if not blocks:
        return np.zeros((0, 3, 1024), dtype=np.uint16), np.zeros(0, dtype=np.float32)
    Xp = np.concatenate(blocks)
    img = glm_plane_image(Xp)[: Xp.shape[0] // 32]       # (glm_plane_image pads to 128-row groups)
    return img, np.concatenate(yb)



# ---- rows of a plate ordered by an unsorted group id (pa_group_rows_build) ------------------------
def group_rows(g, G):
    """(offsets int64 [G+1], rows int64 [N]): rows[offsets[k]:offsets[k+1]] = the n with g[n] == k in
    ascending n.  Restates what the reference's advanced index ``w[..., g, :]`` (SURVEY 8d config 5;
    torch: aten index, scored by pyro/poutine/trace_struct.py:264-278) implies for a kernel that
    visits rows group by group: integer work, the kernel's output is compared bit for bit.  Ids in [-G, 0)
    count from the end, as the reference's index does; anything else is its IndexError."""
    g = np.asarray(g, dtype=np.int64)
    g = np.where(g < 0, g + G, g)
    if g.size and (g.min() < 0 or g.max() >= G):
        raise IndexError("group id outside [-%d, %d)" % (G, G))
    rows = np.argsort(g, kind="stable").astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(np.bincount(g, minlength=G))]).astype(np.int64)
    return offsets, rows
