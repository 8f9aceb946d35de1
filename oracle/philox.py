"""TEST INFRASTRUCTURE -- Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) restated in numpy,
bit-exact with pyro_amd/csrc/common.h.  The reference draws through torch's global generator
(torch: torch/distributions/normal.py:83-86, mt19937 on CPU / Philox on GPU with a different
counter layout), so sample-for-sample parity with the reference is impossible by construction
(SURVEY.md section 7 "RNG parity"); the integer stream below is the contract between the
oracle and the HIP kernels, and the reference is pinned by injecting these draws.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(seed, ctr_lo, ctr_hi):
    """seed: python int (64 bit); ctr_lo, ctr_hi: uint64 arrays (broadcastable).
    Returns 4 uint32 arrays (x, y, z, w)."""
    ctr_lo = np.asarray(ctr_lo, dtype=np.uint64)
    ctr_hi = np.asarray(ctr_hi, dtype=np.uint64)
    ctr_lo, ctr_hi = np.broadcast_arrays(ctr_lo, ctr_hi)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    c0 = (ctr_lo & MASK32).astype(np.uint32)
    c1 = (ctr_lo >> np.uint64(32)).astype(np.uint32)
    c2 = (ctr_hi & MASK32).astype(np.uint32)
    c3 = (ctr_hi >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u32_to_unit_f32(x):
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
            + np.float32(2.0 ** -25))


def u32x2_to_unit_f64(a, b):
    return ((a >> np.uint32(5)).astype(np.float64) * 67108864.0
            + (b >> np.uint32(6)).astype(np.float64)) * 2.0 ** -53 + 2.0 ** -54


def _box_muller(u1, u2, dtype):
    r = np.sqrt(dtype(-2.0) * np.log(u1))
    th = dtype(6.283185307179586) * u2
    return (r * np.cos(th)).astype(dtype), (r * np.sin(th)).astype(dtype)


def uniform_bits(n, dtype, seed, offset, stream_id=0):
    """The raw uniforms behind element i = 0..n-1 (f32: 4 per block, f64: 2 per block)."""
    dtype = np.dtype(dtype).type
    per = 4 if dtype is np.float32 else 2
    nblk = (n + per - 1) // per
    ctr = np.uint64(offset) + np.arange(nblk, dtype=np.uint64)
    x, y, z, w = philox4x32_10(seed, ctr, np.uint64(stream_id))
    if dtype is np.float32:
        out = np.stack([u32_to_unit_f32(x), u32_to_unit_f32(y), u32_to_unit_f32(z),
                        u32_to_unit_f32(w)], axis=1).reshape(-1)
    else:
        out = np.stack([u32x2_to_unit_f64(x, y), u32x2_to_unit_f64(z, w)], axis=1).reshape(-1)
    return out[:n]


def normal(n, dtype, seed, offset, stream_id=0):
    """Standard normals 0..n-1 of the (seed, offset, stream_id) stream -- same mapping as
    pa_philox_normal / philox_normal_f32/_f64 in common.h."""
    dtype = np.dtype(dtype).type
    per = 4 if dtype is np.float32 else 2
    nblk = (n + per - 1) // per
    ctr = np.uint64(offset) + np.arange(nblk, dtype=np.uint64)
    x, y, z, w = philox4x32_10(seed, ctr, np.uint64(stream_id))
    if dtype is np.float32:
        a0, a1 = _box_muller(u32_to_unit_f32(x), u32_to_unit_f32(y), np.float32)
        b0, b1 = _box_muller(u32_to_unit_f32(z), u32_to_unit_f32(w), np.float32)
        out = np.stack([a0, a1, b0, b1], axis=1).reshape(-1)
    else:
        a0, a1 = _box_muller(u32x2_to_unit_f64(x, y), u32x2_to_unit_f64(z, w), np.float64)
        out = np.stack([a0, a1], axis=1).reshape(-1)
    return out[:n]


def uniform(n, dtype, seed, offset, stream_id=0):
    return uniform_bits(n, dtype, seed, offset, stream_id)


def block(seed, ctr_lo, stream_id=0):
    """One Philox block as python ints (x, y, z, w)."""
    x, y, z, w = philox4x32_10(seed, np.uint64(ctr_lo), np.uint64(stream_id))
    return int(x), int(y), int(z), int(w)


def unit_from_block(b, dtype, second=False):
    dtype = np.dtype(dtype).type
    if dtype is np.float32:
        return u32_to_unit_f32(np.uint32(b[2] if second else b[0]))
    if second:
        return u32x2_to_unit_f64(np.uint32(b[2]), np.uint32(b[3]))
    return u32x2_to_unit_f64(np.uint32(b[0]), np.uint32(b[1]))
