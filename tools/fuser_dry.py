"""Dry run of pyro_amd/ops/fuser.py on a machine WITHOUT a GPU (developer tool): host tensors are treated as
device tensors, every generated kernel is compiled with hiprtc for gfx950 (syntax / type check) and NOT
launched -- recorded outputs stay uninitialised, so nothing here checks numbers (tests/test_fuser_gpu.py does).
    python tools/fuser_dry.py
"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from pyro_amd.ops import fuser  # noqa: E402

rtc = ctypes.CDLL("/opt/rocm/lib/libhiprtc.so")
SOURCES = []
_COMPILED = set()


def compile_only(src, grid, block, tensors):
    SOURCES.append(src)
    if src in _COMPILED:
        return
    prog = ctypes.c_void_p()
    assert rtc.hiprtcCreateProgram(ctypes.byref(prog), src.encode(), b"k.hip", 0, None, None) == 0
    opts = (ctypes.c_char_p * 4)(b"--offload-arch=gfx950", b"-O3", b"-ffp-contract=off", b"-std=c++17")
    r = rtc.hiprtcCompileProgram(prog, 4, opts)
    if r != 0:
        n = ctypes.c_size_t()
        rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
        log = ctypes.create_string_buffer(n.value + 1)
        rtc.hiprtcGetProgramLog(prog, log)
        print(src)
        raise SystemExit(log.value.decode()[:3000])
    _COMPILED.add(src)          # (its own record: fuser._CACHE holds kernel handles)
    fuser.STATS["compiled"] += 1
    assert len(tensors) <= fuser.MAX_POINTERS


import contextlib


@contextlib.contextmanager
def dry():
    """Host tensors stand in for device tensors, every generated source is compiled and nothing is launched."""
    saved = (fuser._dev, fuser._launch, fuser.Fuser._factory, fuser.Fuser._const)
    fuser._dev = lambda t: True
    fuser._launch = compile_only
    fuser.Fuser._factory = _factory_cuda
    fuser.Fuser._const = _const
    try:
        yield
    finally:
        fuser._dev, fuser._launch, fuser.Fuser._factory, fuser.Fuser._const = saved


def _factory_cuda(self, func, args, kwargs, value):       # (host tensors stand in: accept device=cpu)
    return self._fill(value, self._meta(func, args, kwargs), device=torch.device("cpu"))


def _const(self, literal, meta, out, device):
    fresh = out is None
    if out is None:
        out = torch.empty(tuple(meta.shape), dtype=meta.dtype)
    return self._new_node("const", literal, [], meta, out=out, fresh=fresh)


def program(dtype):
    x = torch.randn(7, 5, dtype=dtype, requires_grad=True)
    w = torch.randn(5, dtype=dtype, requires_grad=True)
    m = torch.rand(7, 5) > 0.5
    y = (x * w + 2.0).exp().clamp(min=1e-3, max=50.0)
    p = y / y.sum(-1, keepdim=True)
    z = torch.where(m, p.log(), torch.zeros((), dtype=dtype)) * 3.0 - torch.sigmoid(x) ** 2
    q = z.sum(0) + w.abs().sqrt().sum()
    loss = (q * torch.ones(5, dtype=dtype)).sum() + (x.t().contiguous() ** 3).sum()
    loss.backward()
    acc = torch.zeros(7, 5, dtype=dtype)
    acc.add_(x.detach(), alpha=0.5).mul_(2.0).clamp_(min=-1.0)
    acc[2:4].zero_()
    acc[:, 1].fill_(3.0)
    b = (acc > 0) & (acc < 2.0) | torch.isnan(acc)
    # the enumeration idiom table[values] and its backward, (log_)softmax over a short dim
    table = torch.randn(4, 6, dtype=dtype, requires_grad=True)
    idx = torch.arange(4).reshape(4, 1, 1)
    sm = torch.softmax(table[idx] * 2.0, -1) + torch.log_softmax(table, 0)
    (sm * sm).sum().backward()
    sel = torch.randn(3, 7, 5, dtype=dtype, requires_grad=True)
    ((sel[:, 2] * 2.0).sum() + sel[1].sum() + sel[..., 4].sum()).sum().backward()
    lengths = torch.randint(1, 9, (7,))
    parts = [torch.where((t < lengths).unsqueeze(-1), x.detach() * float(t), x.new_zeros(())) for t in range(5)]
    st = torch.stack(parts).permute(1, 0, 2).contiguous() + torch.cat(parts, -1).sum()
    d = torch.dot(w.detach(), w.detach() * 2.0) + (x.detach().t() / acc.t()).sum()
    f = fuser.active()
    v = torch.rand(7, 5, dtype=dtype) + 0.5
    fam = [f.family_log_prob(k, v, w.detach().abs() + 0.5, x.detach().abs() + 0.5, (7, 5)) for k in range(10)]
    assert all(t is not None for t in fam)
    fam += list(f.family_grads(6, fam[0], v, w.detach().abs() + 0.5, x.detach().abs() + 0.5, (7, 5), (True, True, True)))
    fam += list(f.family_grads(1, fam[1].sum(), v, w.detach(), None, (7, 5), (False, True, False)))
    return loss, b.to(dtype).sum(), st, d, fam


def main():
    with dry():
        for dt in (torch.float32, torch.float64):
            before = dict(fuser.STATS)
            with fuser.Fuser():
                res = program(dt)
            del res
            print(dt, {k: fuser.STATS[k] - before[k] for k in fuser.STATS})
    print("kernels generated:", len(SOURCES), "distinct:", len(set(SOURCES)))
    print("not taken:", fuser.UNFUSED)


if __name__ == "__main__":
    main()
