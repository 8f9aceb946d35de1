"""The small element-wise torch operators of a captured step, recorded instead of launched.

A model / guide text and its autograd duals consist largely of operators on tensors of a few thousand
elements -- constraint transforms of parameters (pyro/params/param_store.py:186-206 in the reference),
normalisations, clamps, scalings.  Inside a captured hipGraph each is a node that costs ~4.8 us of
dependent dispatch whatever it computes (60 of the 153 nodes of a config-4 step,
``profiles/r03_cfg4_operator_attribution.txt``).  ``SmallOps`` is a ``TorchDispatchMode`` that SVI
enters while it captures a step: an eligible operator -- float32 operands on the device, at most 64 K
elements, at most four dims -- is not launched; its output is allocated and an instruction is appended to
the pending program of ``csrc/smallops.hip`` (``pa_smallops_record``), which runs as ONE workgroup that
interprets the instructions in order.  The program is emitted before anything else runs: any other
torch operator (this mode flushes), any launch of the library, a recorded chain phase (the library
flushes), or when it is full.  The interpreter uses torch's arithmetic (IEEE + - * /, the same libm
exp / log, ATen's NaN rules for clamp / where), so the captured step computes what the eager steps
computed; ``tests/test_smallops_gpu.py`` holds every instruction to bitwise equality with torch.

MEASURED NEGATIVE RESULT -- OFF by default (``PYRO_AMD_SMALLOPS=1`` opts in).  On config 4 the mode
records 62 operators into 18 interpreter launches and the step gets SLOWER, 1.24 -> 1.36-1.41 ms (config
5: 14 operators in 5 launches, no change): an interpreted instruction is a load -> compute -> store ->
barrier round trip of ~3.5 us on one workgroup -- what a separate graph node costs -- whether a thread's
elements are taken one by one or eight loads at a time.  Folding these operators pays only if
intermediates stay in registers across instructions, i.e. with generated fused kernels, not with an
interpreter.  The instruction set itself is validated against torch (``tests/test_smallops_gpu.py``);
inside a capture it has only been run on configs 4 and 5, and a small hand-written model faulted during
capture with the mode on (unresolved) -- do not switch it on for real work.
"""
import ctypes
import os

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from .. import _lib, kernels

MAX_NUMEL = 1 << 16
(ADD, SUB, MUL, DIV, ADD_IMM, MUL_IMM, DIV_IMM, RSUB_IMM, RDIV_IMM, NEG, EXP, LOG, RECIP, SQRT, CLAMP, COPY,
 FILL, WHERE, GE_IMM, LE_IMM, GT_IMM, LT_IMM, AND_U8) = range(1, 24)


class pa_smallop(ctypes.Structure):          # include/pyro_amd.h
    _fields_ = [("op", ctypes.c_uint32), ("ndim", ctypes.c_uint32), ("numel", ctypes.c_uint32),
                ("barrier", ctypes.c_uint32), ("shape", ctypes.c_uint32 * 4),
                ("dst_stride", ctypes.c_int32 * 4), ("src0_stride", ctypes.c_int32 * 4),
                ("src1_stride", ctypes.c_int32 * 4), ("src2_stride", ctypes.c_int32 * 4),
                ("imm", ctypes.c_float), ("imm2", ctypes.c_float), ("dst", ctypes.c_void_p),
                ("src0", ctypes.c_void_p), ("src1", ctypes.c_void_p), ("src2", ctypes.c_void_p)]


class _No(Exception):
    """The operator is not eligible: it takes the ordinary route."""


def enabled():
    return os.environ.get("PYRO_AMD_SMALLOPS", "0") == "1"


def _is_num(x):
    return isinstance(x, (int, float)) and not isinstance(x, bool)


def _scalar_of(x):
    """A python number, or the value of a 0-dim CPU tensor (a wrapped number: no device sync)."""
    if _is_num(x):
        return float(x)
    if isinstance(x, torch.Tensor) and x.dim() == 0 and x.device.type == "cpu" and x.dtype in (
            torch.float32, torch.float64, torch.int64, torch.int32):
        return float(x.item())
    return None


class SmallOps(TorchDispatchMode):
    def __init__(self, protected=()):
        super().__init__()
        self.protected = set(protected)      # storages nothing may write (hoisted constants)
        self.keep = []                       # tensors the pending program touches
        self.read, self.written = set(), set()
        self.recorded = self.launches = 0
        self._lib = _lib.load()

    # ---- life cycle
    def __enter__(self):
        kernels.check(self._lib.pa_smallops_begin(kernels._stream()))
        return super().__enter__()

    def __exit__(self, *exc):
        try:
            return super().__exit__(*exc)
        finally:
            a, b = ctypes.c_int(0), ctypes.c_int(0)
            rc = self._lib.pa_smallops_end(ctypes.byref(a), ctypes.byref(b))
            self.launches, self.recorded = a.value, b.value
            self.keep = []
            if exc[0] is None:
                kernels.check(rc)

    def flush(self):
        kernels.check(self._lib.pa_smallops_flush())
        self.keep = []
        self.read, self.written = set(), set()

    # ---- eligibility and emission
    def _f32(self, t):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.layout == torch.strided
                and 0 < t.numel() <= MAX_NUMEL):
            raise _No
        return t

    def _u8(self, t):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bool and 0 < t.numel() <= MAX_NUMEL):
            raise _No
        return t

    def _emit(self, op, dst, srcs, imm=0.0, imm2=0.0):
        frame = tuple(dst.shape)
        if len(frame) > 4 or dst.numel() == 0 or dst.numel() > MAX_NUMEL:
            raise _No
        ins = pa_smallop()
        ins.op, ins.ndim, ins.numel = op, len(frame), dst.numel()
        views = []
        for s in srcs:
            if s is None:
                views.append(None)
                continue
            try:
                views.append(s.expand(frame))
            except RuntimeError:
                raise _No
        for t in [dst] + [v for v in views if v is not None]:
            if any(abs(st) >= 2 ** 31 for st in t.stride()):
                raise _No
        dkey = dst.untyped_storage().data_ptr()
        if dkey in self.protected:
            raise _No                        # (the constant replayer below refuses the write)
        skeys = [v.untyped_storage().data_ptr() for v in views if v is not None]
        for v in views:
            # an in-place operator whose source aliases the destination through another element mapping
            # would race inside the instruction
            if v is not None and v.untyped_storage().data_ptr() == dkey and (
                    v.data_ptr() != dst.data_ptr() or tuple(v.stride()) != tuple(dst.stride())):
                raise _No
        hazard = dkey in self.written or dkey in self.read or any(k in self.written for k in skeys)
        if hazard:
            self.read, self.written = set(), set()
        ins.barrier = 1 if hazard else 0
        # bit 1: every operand dense in the frame's row-major order, or a scalar (stride 0 everywhere):
        # the kernel then needs no index decode; the per-operand factor 0 / 1 rides in stride slot 3
        def lin(t):
            if all(st == 0 or sz == 1 for st, sz in zip(t.stride(), t.shape)):
                return 0
            return 1 if t.is_contiguous() else None
        lins = [lin(v) if v is not None else 0 for v in views]
        if dst.is_contiguous() and all(x is not None for x in lins):
            ins.barrier |= 2
            for f, x in zip((ins.src0_stride, ins.src1_stride, ins.src2_stride), lins + [0] * (3 - len(lins))):
                f[3] = x
            ins.ndim = 0                       # (strides below are not filled)
        for d in range(len(frame) if not (ins.barrier & 2) else 0):
            ins.shape[d] = frame[d]
            ins.dst_stride[d] = dst.stride(d)
        fields = (ins.src0_stride, ins.src1_stride, ins.src2_stride)
        ptrs = []
        for v, f in zip(views, fields):
            if v is None:
                ptrs.append(None)
                continue
            if not (ins.barrier & 2):
                for d in range(len(frame)):
                    f[d] = v.stride(d)
            ptrs.append(v.data_ptr())
        ptrs += [None] * (3 - len(ptrs))
        ins.imm, ins.imm2 = imm, imm2
        ins.dst, ins.src0, ins.src1, ins.src2 = dst.data_ptr(), ptrs[0], ptrs[1], ptrs[2]
        kernels.check(self._lib.pa_smallops_record(ctypes.byref(ins)))
        self.keep.append(dst)
        self.keep.extend(v for v in views if v is not None)
        self.written.add(dkey)
        self.read.update(skeys)
        return dst

    def _new(self, shape, like, dtype=torch.float32):
        return torch.empty(tuple(shape), dtype=dtype, device=like.device)

    # ---- operators
    def _binary(self, op_tt, op_imm, rimm, a, b, out=None):
        """a (op) b for tensors / numbers; ``rimm``: the instruction for number (op) tensor or None."""
        sa, sb = _scalar_of(a), _scalar_of(b)
        if sa is not None and sb is None:
            if rimm is None:
                raise _No
            t = self._f32(b)
            return self._emit(rimm, self._new(t.shape, t) if out is None else out, [t], imm=sa)
        t = self._f32(a)
        if sb is not None:
            return self._emit(op_imm, self._new(t.shape, t) if out is None else out, [t], imm=sb)
        u = self._f32(b)
        shape = torch.broadcast_shapes(t.shape, u.shape)
        if out is not None and tuple(out.shape) != tuple(shape):
            raise _No
        return self._emit(op_tt, self._new(shape, t) if out is None else out, [t, u])

    def _unary(self, op, a, imm=0.0, imm2=0.0, out=None):
        t = self._f32(a)
        return self._emit(op, self._new(t.shape, t) if out is None else out, [t], imm=imm, imm2=imm2)

    def _compare(self, op, a, s):
        t, v = self._f32(a), _scalar_of(s)
        if v is None:
            raise _No
        return self._emit(op, self._new(t.shape, t, torch.bool), [t], imm=v)

    def handle(self, name, args, kwargs):
        a = args
        if name in ("add.Tensor", "sub.Tensor", "add_.Tensor", "sub_.Tensor"):
            if kwargs.get("alpha", 1) != 1 or len(a) != 2:
                raise _No
            sub = name.startswith("sub")
            out = self._f32(a[0]) if name[3] == "_" else None
            if sub and _scalar_of(a[1]) is not None:
                return self._unary(ADD_IMM, a[0], imm=-_scalar_of(a[1]), out=out)
            return self._binary(SUB if sub else ADD, ADD_IMM, RSUB_IMM if sub else ADD_IMM, a[0], a[1], out)
        if name in ("mul.Tensor", "mul_.Tensor"):
            return self._binary(MUL, MUL_IMM, MUL_IMM, a[0], a[1], self._f32(a[0]) if name[3] == "_" else None)
        if name in ("div.Tensor", "div_.Tensor"):
            return self._binary(DIV, DIV_IMM, RDIV_IMM, a[0], a[1], self._f32(a[0]) if name[3] == "_" else None)
        if name == "rsub.Scalar" and kwargs.get("alpha", 1) == 1 and len(a) == 2:
            return self._unary(RSUB_IMM, a[0], imm=float(a[1]))
        if name in ("neg.default", "exp.default", "log.default", "reciprocal.default", "sqrt.default"):
            return self._unary({"neg": NEG, "exp": EXP, "log": LOG, "reciprocal": RECIP, "sqrt": SQRT}[name[:-8]], a[0])
        if name in ("clamp.default", "clamp_.default"):
            lo = a[1] if len(a) > 1 else kwargs.get("min")
            hi = a[2] if len(a) > 2 else kwargs.get("max")
            if (lo is not None and not _is_num(lo)) or (hi is not None and not _is_num(hi)):
                raise _No
            return self._unary(CLAMP, a[0], imm=float("-inf") if lo is None else float(lo),
                               imm2=float("inf") if hi is None else float(hi),
                               out=self._f32(a[0]) if name == "clamp_.default" else None)
        if name == "clamp_min.default" and _is_num(a[1]):
            return self._unary(CLAMP, a[0], imm=float(a[1]), imm2=float("inf"))
        if name == "clamp_max.default" and _is_num(a[1]):
            return self._unary(CLAMP, a[0], imm=float("-inf"), imm2=float(a[1]))
        if name == "clone.default":
            if kwargs.get("memory_format") not in (None, torch.contiguous_format, torch.preserve_format):
                raise _No
            return self._unary(COPY, a[0])
        if name == "scalar_tensor.default":
            dev = kwargs.get("device")
            if kwargs.get("dtype", torch.float32) != torch.float32 or dev is None or torch.device(dev).type != "cuda" \
                    or not _is_num(a[0]):
                raise _No
            out = torch.empty((), dtype=torch.float32, device=dev)
            return self._emit(FILL, out, [], imm=float(a[0]))
        if name == "where.self":
            c, x, y = self._u8(a[0]), self._f32(a[1]), self._f32(a[2])
            shape = torch.broadcast_shapes(c.shape, x.shape, y.shape)
            return self._emit(WHERE, self._new(shape, x), [x, y, c])
        if name in ("ge.Scalar", "le.Scalar", "gt.Scalar", "lt.Scalar"):
            return self._compare({"ge": GE_IMM, "le": LE_IMM, "gt": GT_IMM, "lt": LT_IMM}[name[:2]], a[0], a[1])
        if name in ("logical_and.default", "logical_and_.default"):
            p, q = self._u8(a[0]), self._u8(a[1])
            shape = torch.broadcast_shapes(p.shape, q.shape)
            if name == "logical_and_.default":
                if tuple(shape) != tuple(p.shape):
                    raise _No
                return self._emit(AND_U8, p, [p, q])
            return self._emit(AND_U8, self._new(shape, p, torch.bool), [p, q])
        raise _No

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = _NAMES.get(func)
        if name is not None:
            try:
                return self.handle(name, args, kwargs)
            except _No:
                pass
        if not kernels._launches_nothing(func) and self._depends(func, args, kwargs):
            self.flush()                    # it reads what the pending program writes (or writes what it touches)
        return func(*args, **kwargs)

    def _depends(self, func, args, kwargs):
        """Does an operator that is not recorded touch memory of the pending program?  Reads of pending
        writes, and -- for mutating operators -- any overlap; operators whose arguments cannot be
        inspected count as dependent."""
        if not self.written and not self.read:
            return False
        try:
            mutable = func._schema.is_mutable
        except AttributeError:
            return True
        touched = self.written | self.read if mutable else self.written

        def walk(x):
            if isinstance(x, torch.Tensor):
                return x.is_cuda and x.untyped_storage().data_ptr() in touched
            if isinstance(x, (list, tuple)):
                return any(walk(v) for v in x)
            return False
        return walk(args) or walk(list(kwargs.values()))


def _names():
    a = torch.ops.aten
    out = {}
    for n in ("add.Tensor", "sub.Tensor", "add_.Tensor", "sub_.Tensor", "mul.Tensor", "mul_.Tensor", "div.Tensor",
              "div_.Tensor", "rsub.Scalar", "neg.default", "exp.default", "log.default", "reciprocal.default",
              "sqrt.default", "clamp.default", "clamp_.default", "clamp_min.default", "clamp_max.default",
              "clone.default", "scalar_tensor.default", "where.self", "ge.Scalar", "le.Scalar", "gt.Scalar",
              "lt.Scalar", "logical_and.default", "logical_and_.default"):
        pkt, ov = n.split(".")
        try:
            out[getattr(getattr(a, pkt), ov)] = n
        except AttributeError:
            pass
    return out


_NAMES = _names()
