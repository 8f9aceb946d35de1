"""Dry run of pyro_amd/ops/fuser.py on a machine WITHOUT a GPU (developer tool): host tensors are treated as
device tensors, every generated kernel is compiled with hiprtc for gfx950 (syntax / type check) and NOT
launched -- recorded outputs stay uninitialised, so nothing here checks numbers (tests/test_fuser_gpu.py does).
    python tools/fuser_dry.py
"""
import ctypes
import os
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
    fuser.STATS["loaded"] += 1
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


# ---- the schedule, checked on the host -----------------------------------------------------------------
# ``replaying()``: instead of generating code, every recorded operator is kept on its node and RE-RUN (the ATen
# operator itself, on the host tensors) when and where the schedule puts it -- level by level, kernel by kernel,
# node by node; a store the recorder declares dead is poisoned with NaN after its kernel.  A program gives the
# numbers of its eager run, bit for bit, exactly if the schedule respects every dependence of the program and
# drops no store that something still reads: launch levels, kernel merges, partial flushes, sums that take their
# operand's kernel along, dead-store elimination by reference counts.
def _resolve(x):
    if isinstance(x, fuser._Ref):
        t = x.ref()
        assert t is not None, "a recorded operand died before the operator that reads it ran"
        return t.detach()
    if isinstance(x, torch.Tensor):
        return x.detach()               # (the same memory, no autograd history: operators re-run below autograd)
    if isinstance(x, (list, tuple)):
        return type(x)(_resolve(v) for v in x)
    return x


def _rerun(n):
    assert n.replay is not None, "node %s carries no operator" % n.op
    func, args, kwargs = n.replay
    with torch.no_grad():
        res = func(*_resolve(args), **{k: _resolve(v) for k, v in kwargs.items()})
        out = n.out.detach()
        if not (res.data_ptr() == out.data_ptr() and res.shape == out.shape and res.stride() == out.stride()):
            out.copy_(res)


def _replay_level(ks):
    for k in ks:
        if k.absorbed is not None:
            continue
        nodes = list(k.absorbs.nodes) + list(k.nodes) if k.kind == "red" and k.absorbs is not None else list(k.nodes)
        for n in nodes:
            _rerun(n)
        for n in nodes:
            if not n.live and n.out.is_floating_point():
                with torch.no_grad():
                    n.out.detach().fill_(float("nan"))


@contextlib.contextmanager
def replaying():
    saved = (fuser._launch_level, fuser.REPLAY["on"])
    fuser._launch_level = _replay_level
    fuser.REPLAY["on"] = True
    try:
        with dry():
            yield
    finally:
        fuser._launch_level, fuser.REPLAY["on"] = saved


# ---- the generated code, RUN on the host ---------------------------------------------------------------
# ``hosting()``: the generated HIP source itself is compiled by g++ behind a page of shims (the kernel becomes a
# function, blockIdx / threadIdx thread-local variables set by a serial launcher, the lane-group sum a small
# buffer filled by the lanes in turn, the three amdgcn builtins of csrc/dist_fam.h their libm equivalents) and
# executed on the host tensors: index arithmetic, broadcasting, strides, gathers / scatters, softmax, sums with
# their operand's kernel inside, scalars in the argument table -- against the eager run, without a GPU.
_HOST_SHIM = r"""
#include <cmath>
#include <cstdint>
#include <cstring>
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(x)
struct PaDim3 { unsigned x, y, z; };
static thread_local PaDim3 blockIdx, threadIdx;
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static thread_local double pa_group_buf_[64];
#define PA_GROUP_SUM(s, W) { pa_group_buf_[lane] = (double)s; if (lane == (W) - 1) { double t_ = 0; \
    for (int q_ = 0; q_ < (W); ++q_) t_ += pa_group_buf_[q_]; s = t_; } }
#define PA_GROUP_LEADER(lane, W) ((lane) == (W) - 1)
"""
_HOST_LAUNCHER = r"""
extern "C" void pa_host_launch(const void* const* table, int n, long grid) {
  Ptrs a;
  std::memset(&a, 0, sizeof a);
  for (int j = 0; j < n; ++j) a.p[j] = const_cast<void*>(table[j]);
  for (long b = 0; b < grid; ++b)
    for (unsigned t = 0; t < 256; ++t) { blockIdx.x = (unsigned)b; threadIdx.x = t; k(a); }
}
"""
_HOST_LIBS = {}
_HOST_DIR = []


def _host_dir():
    """One scratch directory per process for the host builds, removed at exit."""
    if not _HOST_DIR:
        import atexit
        import shutil
        import tempfile
        _HOST_DIR.append(tempfile.mkdtemp(prefix="pa_fuser_host_"))
        atexit.register(shutil.rmtree, _HOST_DIR[0], ignore_errors=True)
    return _HOST_DIR[0]


def _host_launch(src, grid, block, tensors):
    import hashlib
    import subprocess
    assert block == 256
    SOURCES.append(src)
    lib = _HOST_LIBS.get(src)
    if lib is None:
        name = os.path.join(_host_dir(), hashlib.sha1(src.encode()).hexdigest()[:16])
        with open(name + ".cpp", "w") as fh:
            fh.write(_HOST_SHIM + src + _HOST_LAUNCHER)
        r = subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-w", "-ffp-contract=off",
                            name + ".cpp", "-o", name + ".so"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host build of a generated kernel failed:\n" + r.stderr[:3000] + "\n" + src)
        lib = _HOST_LIBS[src] = ctypes.CDLL(name + ".so")
        lib.pa_host_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long]
        fuser.STATS["compiled"] += 1
    fuser.STATS["loaded"] += 1
    keep = [t for t in tensors if isinstance(t, torch.Tensor)]
    table = (ctypes.c_void_p * len(tensors))(*[t[1] if isinstance(t, tuple) else t.data_ptr() for t in tensors])
    lib.pa_host_launch(table, len(tensors), int(grid))
    del keep
    fuser.STATS["kernels"] += 1


@contextlib.contextmanager
def hosting():
    saved = (fuser._dev, fuser._launch, fuser.Fuser._factory, fuser.Fuser._const)
    fuser._dev = lambda t: True
    fuser._launch = _host_launch
    fuser.Fuser._factory = _factory_cuda
    fuser.Fuser._const = _const
    try:
        yield
    finally:
        fuser._dev, fuser._launch, fuser.Fuser._factory, fuser.Fuser._const = saved


def random_program(seed, n_ops=60, grad=False, long_sums=False):
    """A list of steps over a pool of tensors: element-wise chains with broadcasting, in-place writes through
    views, reductions short and long, where / comparisons, stack / cat, gathers and their accumulate duals,
    softmax, operators the recorder does not know (partial flushes), dropped references (dead stores)."""
    import random
    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    shapes = [(4, 3), (3,), (4, 1), (2, 4, 3), (), (1, 3), (70, 3), (5, 33), (5, 32), (2, 1, 2, 1, 3, 2)]
    if long_sums and seed % 3 == 0:
        shapes.append((17000,))                 # (a sum beyond one lane group's reach: two recorded stages)
    pool0 = [torch.randn(sh, generator=g) for sh in shapes for _ in range(2)]
    steps = []
    unary = [torch.exp, torch.neg, torch.abs, torch.sigmoid, torch.tanh, lambda t: t * 0.5, lambda t: t + 1.5,
             lambda t: t.clamp(min=-0.5, max=0.75), lambda t: t ** 2, lambda t: torch.log1p(t.abs()),
             lambda t: t.clone(), lambda t: 1.0 - t, lambda t: torch.where(t > 0.1, t, t * 0.0 - 1.0)]
    binary = [torch.add, torch.mul, torch.sub, lambda a, b: a / (b.abs() + 1.0), torch.maximum,
              lambda a, b: torch.where(a > b, a, b * 2.0)]
    for _ in range(n_ops):
        steps.append((rnd.choice(["unary", "unary", "binary", "binary", "inplace", "view", "sum", "sum", "join",
                                  "index", "softmax", "unknown", "drop", "scatter", "cast", "logic", "fills",
                                  "expand", "copyview", "pow", "dot", "intcmp"]),
                      rnd.random(), rnd.random(), rnd.random(), rnd.randrange(len(unary)), rnd.randrange(len(binary))))

    def run():
        pool = [t.clone().requires_grad_(grad) for t in pool0]
        leaves = list(pool)

        def pick(r, pred=lambda t: True):
            c = [t for t in pool if pred(t)]
            return c[int(r * len(c)) % len(c)] if c else None
        for kind, r0, r1, r2, iu, ib in steps:
            out = None
            if kind == "unary":
                out = unary[iu](pick(r0))
            elif kind == "binary":
                a = pick(r0)
                b = pick(r1, lambda t: fuser._bcast(tuple(t.shape), tuple(a.shape)) is not None)
                out = binary[ib](a, b)
            elif kind == "inplace":
                a = pick(r0, lambda t: t.dim() >= 1 and t.shape[0] > 1 and not t._is_view() or t.dim() == 0)
                if a is not None and a.dim() >= 1 and not grad:
                    tgt = a[1:] if r1 < 0.5 else a
                    src = pick(r2, lambda t: fuser._bcast(tuple(t.shape), tuple(tgt.shape)) == tuple(tgt.shape))
                    if r1 < 0.25:
                        tgt.mul_(0.5)
                    elif r1 < 0.5:
                        tgt.zero_()
                    elif src is not None and src.data_ptr() != tgt.data_ptr():
                        tgt.add_(src, alpha=0.25)
                    else:
                        tgt.clamp_(min=-1.0)
            elif kind == "view":
                a = pick(r0, lambda t: t.dim() >= 2)
                out = a.transpose(0, 1) if r1 < 0.4 else (a[0] if r1 < 0.7 else a.unsqueeze(0))
            elif kind == "sum":
                a = pick(r0, lambda t: t.dim() >= 1)
                out = a.sum() if r1 < 0.3 else a.sum(int(r2 * a.dim()) % a.dim(), keepdim=r1 < 0.6)
            elif kind == "join":
                a = pick(r0, lambda t: t.dim() >= 1)
                b = pick(r1, lambda t: t.shape == a.shape)
                out = torch.stack([a, b * 2.0]) if r2 < 0.5 else torch.cat([a, b, a], -1)
            elif kind == "index":
                a = pick(r0, lambda t: t.dim() >= 2)
                idx = torch.tensor([[a.shape[0] - 1], [0], [-1]])
                out = a[idx] * 1.5
            elif kind == "scatter":
                a = pick(r0, lambda t: t.dim() == 2)
                v = pick(r1, lambda t: t.shape == a.shape[1:] and t.dtype == a.dtype)
                if v is not None:
                    out = torch.index_put(a, (torch.tensor([0, a.shape[0] - 1, 0]),), v, accumulate=True)
            elif kind == "softmax":
                a = pick(r0, lambda t: t.dim() >= 1 and t.shape[-1] <= 32)
                out = torch.softmax(a, -1) if r1 < 0.5 else torch.log_softmax(a, 0)
            elif kind == "unknown":
                a = pick(r0, lambda t: t.dim() >= 1)
                out = torch.cumsum(a, 0)                       # (not recorded: whatever it reads is flushed first)
            elif kind == "cast":
                a = pick(r0)
                out = a.double().float() * 1.5 if r1 < 0.3 else ((a > 0.2).to(a.dtype) + a if r1 < 0.6 else
                                                                 a.double() + 0.25)
            elif kind == "logic":
                a = pick(r0)
                b = pick(r1, lambda t: t.shape == a.shape)
                m = ((a > 0.1) & (b < 0.5)) | (a != a) | ~(b >= -1.0)
                out = torch.where(m, a, b * 2.0) if r2 < 0.7 else m.to(a.dtype)
            elif kind == "fills":
                a = pick(r0)
                out = torch.zeros_like(a) + a if r1 < 0.3 else (torch.full_like(a, 2.5) * a if r1 < 0.6 else
                                                                a.new_ones(tuple(a.shape)) - a)
            elif kind == "expand":
                a = pick(r0, lambda t: 1 <= t.dim() <= 4)
                e = a.unsqueeze(0).expand(3, *a.shape)
                out = e * 2.0 if r1 < 0.5 else (e + 1.0).sum(0)
            elif kind == "copyview":
                a = pick(r0, lambda t: t.dim() >= 2 and t.shape[-1] >= 2)
                if a is not None and not grad:
                    dst = a.clone()
                    dst[..., 0].copy_(dst[..., 1] * 2.0)
                    dst[..., 1].fill_(0.5)
                    out = dst
            elif kind == "pow":
                a = pick(r0)
                out = (a.abs() + 0.1) ** 1.5 if r1 < 0.3 else (a ** 3 if r1 < 0.6 else (a.abs() + 0.5) ** -1.0)
            elif kind == "dot":
                a = pick(r0, lambda t: t.dim() == 1 and t.shape[0] <= 32)
                b = pick(r1, lambda t: t.shape == a.shape and t.dtype == a.dtype) if a is not None else None
                if b is not None:
                    out = torch.dot(a, b * 0.5)
            elif kind == "intcmp":
                a = pick(r0, lambda t: t.dim() >= 1)
                ids = torch.arange(a.shape[-1])
                out = torch.where((ids < int(r1 * a.shape[-1]) + 1) & (ids >= 0), a, a * 0.0 + 3.0)
            elif kind == "drop" and len(pool) > 8:
                del pool[int(r0 * len(pool)) % len(pool)]
            if out is not None and out.numel() > 0:
                pool.append(out)
        if grad:        # the autograd duals run on the autograd thread, inside the same scope: recorded too
            terms = [(t * (0.5 + 0.25 * j)).sum().float() for j, t in enumerate(pool[-12:]) if t.requires_grad]
            torch.stack(terms).sum().backward()
            return [t.detach() for t in pool] + [t.grad for t in leaves if t.grad is not None]
        return pool
    return run


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
    # sums that walk their range per thread (many outputs, innermost dim kept), that take their operand's
    # kernel along, and that are longer than a lane group takes (two recorded stages)
    big = torch.randn(40, 1100, dtype=dtype)
    lng = torch.randn(20000, 2, dtype=dtype)
    sums = [(big * 2.0).tanh().sum(0), big.sum(0), (big.t() + 1.0).sum(1), (lng * 0.5).sum(0), lng.abs().sum(),
            lng.view(4, 5000, 2).sum(1), (big[:, :70] * big[:, 70:140]).sum(-1)]
    # the element-wise families: recorded with csrc/dist_fam.h's expressions under a scope, torch's own eagerly
    from pyro_amd.distributions import fused
    f = fuser.active()
    v = torch.rand(7, 5, dtype=dtype) * 0.8 + 0.1
    a0, b0 = w.detach().abs() + 0.5, x.detach().abs() + 0.5
    if f is not None:
        fam = [f.family_log_prob(k, v, a0, b0, (7, 5)) for k in range(10)]
        assert all(t is not None for t in fam)
        for k in range(10):
            fam += list(f.family_grads(k, fam[0], v, a0, b0, (7, 5), (True, True, True)))
        fam.append(f.family_grads(1, fam[1].sum(), v, w.detach(), None, (7, 5), (False, True, False))[1])
    else:
        fam = [fused._differentiable_log_prob(k, v, a0, b0) for k in range(10)]
        for k in range(10):
            leaves = [t.expand(7, 5).clone().requires_grad_(True) for t in (v, a0, b0)]
            got = torch.autograd.grad(fused._differentiable_log_prob(k, *leaves), leaves, fam[0], allow_unused=True)
            fam += [torch.zeros(7, 5, dtype=dtype) if t is None else t for t in got]      # (one-parameter families)
        la = w.detach().expand(7, 5).clone().requires_grad_(True)
        fam.append(torch.autograd.grad(fused._differentiable_log_prob(1, v, la, None), [la],
                                       fam[1].sum() * torch.ones(7, 5, dtype=dtype))[0])
    return loss, b.to(dtype).sum(), st, d, fam, sums


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
