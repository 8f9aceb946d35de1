"""Torch-tensor wrappers over the C-ABI of libpyro_amd.so.

Every function here hands raw device pointers of torch tensors to a HIP kernel on torch's
current stream.  Tensors must live on a HIP device: there is no CPU implementation and no
fallback (``_require_gpu`` raises).  PyTorch is used for device memory and streams only.
"""
import ctypes
import os

import torch
import torch.utils._python_dispatch

from . import _lib
from ._lib import NULL_VIEW, Unsupported, View2D, check  # noqa: F401

_DTYPES = {torch.float32: _lib.PA_F32, torch.float64: _lib.PA_F64}


def on_device(t):
    """Does ``t`` live where the HIP kernels run?  The fused routes are taken for such tensors only;
    everything else goes through the torch operators."""
    return t.is_cuda


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "pyro_amd: kernel called with a tensor on %s; the HIP backend only runs on an "
                "MI355X device and has no CPU fallback" % t.device)


def _dtype(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise ValueError("pyro_amd: unsupported dtype %s (float32/float64 only)" % t.dtype)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class DirectReplay:
    """A captured hipGraph that is one short chain of kernel nodes, replayed as plain kernel launches
    (csrc/replay.hip).  Opt-in (DIRECT_REPLAY below: measured slower than hipGraphLaunch for a step that waits
    for its loss, faster when replays are queued ahead).  Holds the graph (the plan points into its nodes'
    argument blocks)."""

    def __init__(self, handle, graph, n_nodes):
        self._handle = ctypes.c_void_p(handle)
        self.graph, self.n_nodes = graph, n_nodes
        self._launch = _lib.load().pa_graph_direct_launch

    def launch(self):
        rc = self._launch(self._handle, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            check(rc)

    def free(self):
        h, self._handle = self._handle, None
        if h is not None:
            _lib.load().pa_graph_direct_free(h)

    def __del__(self):
        try:
            self.free()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass


# Measured on the config-2 step (two kernel nodes; tools/graph_launch_host_cost.py, ROCm 7.2): both forms cost the
# host ~10 us per enqueue.  UN-ARMED (step() launches its replay and waits for the loss) the pre-encoded graph
# reaches the device sooner: 72.2 us per step against 75.7 as two launches -- so "on" is opt-in
# (PYRO_AMD_DIRECT_REPLAY=1 / DIRECT_REPLAY["on"] = True).  When replays are enqueued AHEAD of the host the device's
# cadence is what counts and plain launches follow each other more closely than graph launches (62.5 against
# 66.5 us per step with the queue kept full).
DIRECT_REPLAY = {"on": os.environ.get("PYRO_AMD_DIRECT_REPLAY", "0") == "1", "max_nodes": 4}


def new_graph():
    """A torch CUDAGraph that keeps its hipGraph_t after the capture (so that graph_direct_plan can read its
    nodes) when direct replay is switched on and this torch can do that; a plain one otherwise."""
    if DIRECT_REPLAY["on"]:
        try:
            return torch.cuda.CUDAGraph(keep_graph=True)
        except TypeError:      # (an older torch: no keep_graph)
            pass
    return torch.cuda.CUDAGraph()


def graph_direct_plan(graph):
    """DirectReplay for a captured torch CUDAGraph made by new_graph(), or None (more than
    DIRECT_REPLAY["max_nodes"] nodes, a node that is not a kernel, a fork, a module-launched kernel, or a torch
    that cannot hand out the hipGraph_t): the caller then replays the graph the ordinary way."""
    if not DIRECT_REPLAY["on"] or not hasattr(graph, "raw_cuda_graph"):
        return None
    try:
        raw = graph.raw_cuda_graph()
    except RuntimeError:       # (keep_graph was not set)
        return None
    plan, n = ctypes.c_void_p(), ctypes.c_int()
    check(_lib.load().pa_graph_direct_plan(ctypes.c_void_p(raw), DIRECT_REPLAY["max_nodes"], ctypes.byref(plan),
                                           ctypes.byref(n)))
    return DirectReplay(plan.value, graph, n.value) if plan.value else None


# Who wants to know that a launch of ours is about to touch tensor t: pyro_amd/ops/fuser.py (it defers
# eligible torch operators; whatever is recorded and shares memory with t has to be materialised first),
# SVI's capture (it notes every tensor a captured step reads that was not made inside the step)
_PTR_HOOKS = []


def _ptr(t):
    if t is None:
        return None
    for hook in _PTR_HOOKS:
        hook(t)
    if _CHAIN["keep"] is not None:
        _CHAIN["keep"].append(t)       # a recorded launch reads / writes it at the flush
    return ctypes.c_void_p(t.data_ptr())


def _view(t, rows, cols):
    """Describe tensor ``t`` (already broadcast-compatible with a logical [rows, cols]) as a
    2-D strided view without materialising broadcasts."""
    if t is None:
        return NULL_VIEW
    for hook in _PTR_HOOKS:
        hook(t)
    if _CHAIN["keep"] is not None:
        _CHAIN["keep"].append(t)
    assert t.dim() == 2
    sr = 0 if t.shape[0] == 1 and rows != 1 else t.stride(0)
    sc = 0 if t.shape[1] == 1 and cols != 1 else t.stride(1)
    if t.shape[0] == 1:
        sr = 0
    if t.shape[1] == 1:
        sc = 0
    return View2D(t.data_ptr(), sr, sc)


# ------------------------------------------------------------------------------------------
# the chained tail of a captured SVI step (include/pyro_amd.h "the chained tail", csrc/chain.hip)
# ------------------------------------------------------------------------------------------
# While ``chain_recording`` is active the dependent small launches that end an ELBO-gradient step
# (GLM finalize, ELBO assembly, guide backward, Adam) are recorded by the library and launched as
# the phases of ONE kernel.  Everything the library launches itself flushes the pending phases
# first (csrc: as_stream); what torch launches is watched by a TorchDispatchMode: any operator
# that is not a pure view / allocation flushes before it runs, so a torch kernel can never read
# (or overwrite) a buffer a pending phase has yet to write (or read).  Tensors whose pointers were
# handed to the library while recording are kept alive until the flush.
_CHAIN = {"keep": None, "sync": {}, "stats": None}

_FREE_OPS = None


def _free_ops():
    global _FREE_OPS
    if _FREE_OPS is None:
        a = torch.ops.aten
        names = ["empty.memory_format", "empty_strided.default", "empty_like.default",
                 "new_empty.default", "new_empty_strided.default", "detach.default", "alias.default",
                 "lift_fresh.default", "_unsafe_view.default", "_reshape_alias.default",
                 "is_same_size.default", "sym_size.int", "sym_stride.int", "sym_numel.default",
                 "sym_storage_offset.default", "is_pinned.default"]
        ops = set()
        for n in names:
            pkt, ov = n.split(".")
            try:
                ops.add(getattr(getattr(a, pkt), ov))
            except AttributeError:
                pass
        _FREE_OPS = ops
    return _FREE_OPS


def _launches_nothing(func):
    """True for operators that only make views / metadata / uninitialised memory."""
    if func in _free_ops():
        return True
    try:
        return bool(func.is_view)
    except AttributeError:
        return False


class _ChainGuard(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if not _launches_nothing(func):
            if "torch_ops" in _CHAIN:          # (a StepGate scope counts what torch launches)
                _CHAIN["torch_ops"] += 1
            chain_flush()
        return func(*args, **(kwargs or {}))


def chain_sync_buffer(device):
    """The zeroed device words the chain kernel's phase barriers count in (allocate BEFORE a graph
    capture; persistent: captured chain launches keep using it)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    buf = _CHAIN["sync"].get(key)
    if buf is None:
        buf = torch.zeros((_lib.CHAIN_SYNC_BYTES // 4,), dtype=torch.int32, device=device)
        _CHAIN["sync"][key] = buf
    return buf


def chain_debug_stamps(buf):
    """Developer hook: ``buf`` = int64[32] device tensor that later chain launches fill with
    wall-clock stamps at their phase boundaries (None: off).  pa_chain_debug_stamps."""
    check(_lib.load().pa_chain_debug_stamps(None if buf is None else ctypes.c_void_p(buf.data_ptr())))
    _CHAIN["stamps"] = buf


def chain_tune(fuse_tail=True, fast_sites=True, jitter_seed=0, fuse_draw=True):
    """pa_chain_tune: allow (default) / forbid the per-site fused form of the chained tail, and
    within it the one-pass code for AutoNormal-shaped sites.  ``jitter_seed`` != 0: race hunting --
    pseudo-random per-workgroup delays around every device-wide arrival / wait of the chain kernels.
    ``fuse_draw``: inside a recording the mean-field guide draw is made by the plane-image GLM kernel
    that consumes it (default) or stays its own launch."""
    check(_lib.load().pa_chain_tune((1 if fuse_tail else 0) | (0 if fast_sites else 2)
                                    | (0 if fuse_draw else 4) | ((int(jitter_seed) & 0x7fffff) << 8)))


def chain_flush():
    """Launch the recorded phases now (no-op when nothing is pending)."""
    if _CHAIN["keep"] is not None:
        check(_lib.load().pa_chain_flush())


class chain_recording:
    """Context manager: record the chainable launches made inside on torch's current stream and run
    them as phases of one kernel (see above).  ``stats`` after exit: (chain launches, phases)."""

    def __init__(self, device):
        self.sync = chain_sync_buffer(device)
        self.stats = (0, 0)
        self.fused = 0
        self._guard = None

    def __enter__(self):
        assert _CHAIN["keep"] is None, "nested chain recordings"
        check(_lib.load().pa_chain_begin(_stream(), ctypes.c_void_p(self.sync.data_ptr()),
                                         self.sync.numel() * 4))
        _CHAIN["keep"] = []
        self._guard = _ChainGuard()
        self._guard.__enter__()
        return self

    def __exit__(self, *exc):
        try:
            self._guard.__exit__(*exc)
        finally:
            a, b = ctypes.c_int(0), ctypes.c_int(0)
            rc = _lib.load().pa_chain_end(ctypes.byref(a), ctypes.byref(b))
            _CHAIN["keep"] = None
            self.stats = (a.value, b.value)
            self.fused = _lib.load().pa_chain_fused_launches()
        if exc[0] is None:
            check(rc)
        return False


# ------------------------------------------------------------------------------------------
# measurement hook
# ------------------------------------------------------------------------------------------

class KernelTimer:
    """Collects HIP-event brackets around launches of one dominant kernel (bench.py).
    ``arm()`` before the launching call; ``mean_ms()`` after a device synchronise."""

    def __init__(self, kernel_tag):
        self.tag = kernel_tag
        self.pairs = []

    def arm(self):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        # torch creates the underlying hipEvent lazily on first record; force creation
        s.record()
        e.record()
        check(_lib.load().pa_profile_bracket_next(self.tag, ctypes.c_void_p(s.cuda_event),
                                                  ctypes.c_void_p(e.cuda_event)))
        self.pairs.append((s, e))

    def read_last(self):
        """Elapsed ms of the most recently armed bracket (after a device synchronise).  When the
        bracket was armed for a launch that got captured into a hipGraph, the two event records
        are graph nodes and every replay re-records them: this is the last replay's kernel time."""
        s, e = self.pairs[-1]
        return s.elapsed_time(e)

    def times_ms(self):
        return [s.elapsed_time(e) for s, e in self.pairs]

    def mean_ms(self):
        t = self.times_ms()
        return sum(t) / len(t) if t else float("nan")


# ------------------------------------------------------------------------------------------
# RNG
# ------------------------------------------------------------------------------------------

def philox_normal(shape, dtype, device, seed, offset, offset_dev=None):
    """offset_dev: optional int64[1] device tensor added to ``offset`` by the kernel."""
    out = torch.empty(shape, dtype=dtype, device=device)
    _require_gpu(out, offset_dev)
    check(_lib.load().pa_philox_normal(_ptr(out), out.numel(), _dtype(out), seed, offset,
                                       _ptr(offset_dev), _stream()))
    return out


def philox_uniform(shape, dtype, device, seed, offset, offset_dev=None):
    out = torch.empty(shape, dtype=dtype, device=device)
    _require_gpu(out, offset_dev)
    check(_lib.load().pa_philox_uniform(_ptr(out), out.numel(), _dtype(out), seed, offset,
                                        _ptr(offset_dev), _stream()))
    return out


def gamma_rsample(alpha, rows, cols, seed, offset, offset_dev=None, want_grad=True):
    """alpha: 2-D tensor broadcastable to [rows, cols] -> (standard-Gamma draws [rows, cols],
    d draw / d alpha [rows, cols] or None).  pa_gamma_rsample."""
    _require_gpu(alpha, offset_dev)
    out = torch.empty((rows, cols), dtype=alpha.dtype, device=alpha.device)
    dal = torch.empty_like(out) if want_grad else None
    check(_lib.load().pa_gamma_rsample(_dtype(alpha), _ptr(out), _ptr(dal), _view(alpha, rows, cols),
                                       rows, cols, seed, offset, _ptr(offset_dev), _stream()))
    return out, dal


def gamma_implicit_grad(alpha, value):
    """d value / d alpha of reparameterised standard-Gamma draws at given (alpha, value), same shapes."""
    _require_gpu(alpha, value)
    a2, v2 = alpha.reshape(1, -1), value.reshape(1, -1)
    out = torch.empty_like(v2)
    check(_lib.load().pa_gamma_implicit_grad(_dtype(alpha), _ptr(out), _view(a2, 1, a2.shape[1]),
                                             _view(v2, 1, v2.shape[1]), 1, v2.shape[1], _stream()))
    return out.reshape(value.shape)


def publish_scalar(src, host_value, host_seq, counter=None, inc=0):
    """Graph epilogue: counter += inc (optional), then the device scalar ``src`` is written to the
    pinned host tensors ``host_value`` (float64[1]) / ``host_seq`` (int64[1], incremented)."""
    _require_gpu(src, counter)
    assert host_value.is_pinned() and host_seq.is_pinned()
    assert host_value.dtype == torch.float64 and host_seq.dtype == torch.int64
    assert counter is None or (counter.dtype == torch.int64 and counter.numel() >= 2)
    check(_lib.load().pa_publish_scalar(_dtype(src), _ptr(src), _ptr(host_value), _ptr(host_seq),
                                        _ptr(counter), int(inc), _stream()))


def exp_site_fwd(u, cols, lower=0.0, want_ld=True):
    """u contiguous with prod(trailing dims) == cols per row -> (value = lower + exp(u) [u.shape],
    log_density = -sum over each row [rows], or None without ``want_ld``).  pa_exp_site_fwd."""
    _require_gpu(u)
    assert u.is_contiguous() and u.dtype in (torch.float32, torch.float64) and cols >= 1
    rows = u.numel() // cols
    value = torch.empty_like(u)
    ld = torch.empty((rows,), dtype=u.dtype, device=u.device) if want_ld else None
    check(_lib.load().pa_exp_site_fwd(_dtype(u), _ptr(u), rows, cols, float(lower), _ptr(value), _ptr(ld), _stream()))
    return value, ld


def exp_site_bwd(value, g_value, g_ld, cols, lower=0.0):
    """d u = g_value * exp(u) - g_ld (per row); either gradient may be None."""
    _require_gpu(value, g_value, g_ld)
    assert value.is_contiguous()
    rows = value.numel() // cols
    g_u = torch.empty_like(value)
    check(_lib.load().pa_exp_site_bwd(_dtype(value), _ptr(value), _ptr(g_value), _ptr(g_ld), rows, cols,
                                      float(lower), _ptr(g_u), _stream()))
    return g_u


def meanfield_score(z, loc, scale, P, coef):
    """(partial sums of coef * sum log Normal(z; loc, scale) [blocks], -coef * P / scale [n]) for a draw
    z [P, n] of the guide's own (pa_meanfield_score)."""
    _require_gpu(z, loc, scale)
    n = loc.numel()
    assert z.is_contiguous() and loc.is_contiguous() and scale.is_contiguous() and z.numel() == P * n
    assert z.dtype == loc.dtype == scale.dtype
    lib = _lib.load()
    nb = lib.pa_meanfield_score_blocks(P, n)
    partial = torch.empty((nb,), dtype=z.dtype, device=z.device)
    gscale = torch.empty((n,), dtype=z.dtype, device=z.device)
    check(lib.pa_meanfield_score(_dtype(z), _ptr(z), _ptr(loc), _ptr(scale), P, n, float(coef), _ptr(partial),
                                 _ptr(gscale), _stream()))
    return partial, gscale


class StepGate:
    """The device / pinned-host words of one captured step's gate (include/pyro_amd.h "the step gate"):
    a replay enqueued ahead of time waits in its first node until the host writes its number into
    ``go`` -- or gives itself up after ``timeout_us`` and changes nothing.  40 us: several times what a
    host loop needs to come back (3-8 us), short enough that a stream synchronisation right after a
    step -- which has to sit out the armed replay's patience -- costs little."""

    def __init__(self, device, timeout_us=40, late=False):
        self.gate = torch.zeros((2,), dtype=torch.int64, device=device)     # {last step run, abort}
        self.go = torch.zeros((1,), dtype=torch.int64).pin_memory()
        self.ack = torch.zeros((1,), dtype=torch.int64).pin_memory()
        self.go_np, self.ack_np = self.go.numpy(), self.ack.numpy()
        self.timeout_us = int(timeout_us)
        self.next = 1                  # number of the next replay that will RUN
        self.total = self.aware = -1   # launches of the capture / gate-aware ones among them
        self.torch_ops = -1            # torch operators that launched something during the capture
        # late = the gate node sits in front of the step's chained tail, not first: the forward pass of a
        # replay enqueued ahead of time runs while the host is still between two step() calls
        # (include/pyro_amd.h, pa_gate_defer)
        self.late = bool(late)
        self.pre = self.pre_other = -1  # launches before a late gate / of them not the plane-image GLM
        self.emitted = not self.late

    def launch(self):
        """The gate node: first node of the capture, or (late) registered now and emitted in front of
        the chained tail."""
        lib = _lib.load()
        if self.late:
            check(lib.pa_gate_defer(_ptr(self.go), _ptr(self.gate), _ptr(self.ack), self.timeout_us))
        else:
            check(lib.pa_gate(_ptr(self.go), _ptr(self.gate), _ptr(self.ack), self.timeout_us, _stream()))

    def __enter__(self):
        # (late: pa_gate_defer in launch() resets the counters; the scope opens when the node is emitted)
        check(_lib.load().pa_gate_scope(None if self.late else _ptr(self.gate)))
        _CHAIN["torch_ops"] = 0
        return self

    def __exit__(self, *exc):
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        lib = _lib.load()
        lib.pa_gate_stats(ctypes.byref(a), ctypes.byref(b))
        if self.late:
            p, q, e = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
            lib.pa_gate_defer_stats(ctypes.byref(p), ctypes.byref(q), ctypes.byref(e))
            self.pre, self.pre_other, self.emitted = p.value, q.value, bool(e.value)
        lib.pa_gate_scope(None)
        self.total, self.aware = a.value, b.value
        self.torch_ops = _CHAIN.pop("torch_ops", -1)
        return False

    @property
    def armable(self):
        """Every node of the captured step behind the gate returns at once when the gate gives a replay
        up -- and (late gate) nothing but the plane-image GLM kernel runs in front of it."""
        return (self.emitted and self.total > 0 and self.total == self.aware and self.torch_ops == 0
                and (not self.late or self.pre_other == 0))


def counter_add(counter, inc):
    """*counter += inc on the stream (device-resident Philox base offset; graph-replay safe)."""
    _require_gpu(counter)
    assert counter.dtype == torch.int64 and counter.numel() >= 1
    check(_lib.load().pa_counter_add(_ptr(counter), int(inc), _stream()))


# ------------------------------------------------------------------------------------------
# element-wise site kernels on a [rows, cols] broadcast frame
# ------------------------------------------------------------------------------------------

def dist_log_prob(dist_id, value, p0, p1, rows, cols):
    """value/p0/p1: 2-D tensors broadcastable to [rows, cols]. Returns [rows, cols]."""
    _require_gpu(value, p0, p1)
    out = torch.empty((rows, cols), dtype=value.dtype, device=value.device)
    check(_lib.load().pa_dist_log_prob(dist_id, _dtype(value), _ptr(out), _view(value, rows, cols),
                                       _view(p0, rows, cols), _view(p1, rows, cols), rows, cols,
                                       _stream()))
    return out


def dist_log_prob_sum(dist_id, value, p0, p1, mask, scale, rows, cols, want_total=False):
    """Fused log_prob -> scale_and_mask -> sum over cols. Returns rowsum[rows], or
    (rowsum[rows], total 0-dim) with ``want_total`` (the total comes from the same launch)."""
    _require_gpu(value, p0, p1, mask)
    lib = _lib.load()
    buf = torch.empty((rows + 1,), dtype=value.dtype, device=value.device)
    out, total = buf[:rows], buf[rows]
    nbytes = lib.pa_dist_log_prob_sum_workspace(rows, cols)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=value.device) if nbytes else None
    check(lib.pa_dist_log_prob_sum(dist_id, _dtype(value), _ptr(out),
                                   _ptr(total) if want_total else None,
                                   _view(value, rows, cols), _view(p0, rows, cols),
                                   _view(p1, rows, cols), _view(mask, rows, cols), float(scale),
                                   rows, cols, _ptr(ws), nbytes, _stream()))
    return (out, total) if want_total else out


def dist_log_prob_grad(dist_id, g, value, p0, p1, mask, scale, rows, cols, need):
    """Gradients w.r.t. (value, p0, p1) as contiguous [rows, cols] (None where not needed)."""
    _require_gpu(g, value, p0, p1, mask)
    outs = [torch.empty((rows, cols), dtype=value.dtype, device=value.device) if n else None
            for n in need]
    check(_lib.load().pa_dist_log_prob_grad(
        dist_id, _dtype(value), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _view(g, rows, cols),
        _view(value, rows, cols), _view(p0, rows, cols), _view(p1, rows, cols),
        _view(mask, rows, cols), float(scale), rows, cols, _stream()))
    return outs


def _nd_args(shape, t):
    """(pointer, int64[ndim] element strides of ``t`` expanded to ``shape``) for the N-D kernels."""
    import ctypes
    if t is None:
        return None, None, None
    e = t.expand(shape)
    arr = (ctypes.c_int64 * len(shape))(*[0 if n == 1 else int(st)
                                          for n, st in zip(shape, e.stride())])
    return _ptr(e), arr, e


def dist_log_prob_sum_nd(dist_id, shape, value, p0, p1, mask, scale):
    """sum(scale_and_mask(log_prob)) over the broadcast frame ``shape`` (<= 4 dims) of operands
    that need not be 2-D collapsible (pa_dist_log_prob_sum_nd).  Returns a 0-dim tensor."""
    import ctypes
    _require_gpu(value, p0, p1, mask)
    lib = _lib.load()
    shape = tuple(int(s) for s in shape)
    sizes = (ctypes.c_int64 * len(shape))(*shape)
    total = torch.empty((), dtype=value.dtype, device=value.device)
    nbytes = lib.pa_dist_log_prob_sum_nd_workspace()
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=value.device)
    pv, sv, kv = _nd_args(shape, value)
    pa_, sa, ka = _nd_args(shape, p0)
    pb, sb, kb = _nd_args(shape, p1)
    pm, sm, km = _nd_args(shape, mask)
    check(lib.pa_dist_log_prob_sum_nd(dist_id, _dtype(value), _ptr(total), len(shape), sizes, pv,
                                      sv, pa_, sa, pb, sb, pm, sm, float(scale), _ptr(ws), nbytes,
                                      _stream()))
    return total


def dist_log_prob_grad_nd(dist_id, shape, g, value, p0, p1, mask, scale, need):
    """Un-reduced gradients w.r.t. (value, p0, p1), each contiguous of ``shape`` (None where not
    needed); ``g`` = upstream gradient of the site sum (one element)."""
    import ctypes
    _require_gpu(g, value, p0, p1, mask)
    shape = tuple(int(s) for s in shape)
    sizes = (ctypes.c_int64 * len(shape))(*shape)
    outs = [torch.empty(shape, dtype=value.dtype, device=value.device) if n else None for n in need]
    pv, sv, kv = _nd_args(shape, value)
    pa_, sa, ka = _nd_args(shape, p0)
    pb, sb, kb = _nd_args(shape, p1)
    pm, sm, km = _nd_args(shape, mask)
    g = g.reshape(1).contiguous()
    check(_lib.load().pa_dist_log_prob_grad_nd(
        dist_id, _dtype(value), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(g), len(shape),
        sizes, pv, sv, pa_, sa, pb, sb, pm, sm, float(scale), _stream()))
    return outs


def sum_to_nd(x, A, R, B):
    """x contiguous, viewed as [A, R, B] -> [A, B] summed over R (pa_sum_to_nd)."""
    _require_gpu(x)
    assert x.is_contiguous() and x.numel() == A * R * B
    lib = _lib.load()
    out = torch.empty((A, B), dtype=x.dtype, device=x.device)
    nbytes = lib.pa_sum_to_nd_workspace(A, R, B)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device) if nbytes else None
    check(lib.pa_sum_to_nd(_dtype(x), _ptr(x), _ptr(out), A, R, B, _ptr(ws), nbytes, _stream()))
    return out


def sum_to_nd_pair(x0, x1, A, R, B):
    """Two contiguous tensors viewed as [A, R, B] -> two [A, B] by the same launches (pa_sum_to_nd_pair)."""
    _require_gpu(x0, x1)
    assert x0.is_contiguous() and x1.is_contiguous() and x0.numel() == x1.numel() == A * R * B
    assert x0.dtype == x1.dtype
    lib = _lib.load()
    out0 = torch.empty((A, B), dtype=x0.dtype, device=x0.device)
    out1 = torch.empty_like(out0)
    nbytes = 2 * lib.pa_sum_to_nd_workspace(A, R, B)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x0.device) if nbytes else None
    check(lib.pa_sum_to_nd_pair(_dtype(x0), _ptr(x0), _ptr(out0), _ptr(x1), _ptr(out1), A, R, B, _ptr(ws),
                                nbytes, _stream()))
    return out0, out1


def normal_rsample(loc, scale, rows, cols, seed, offset, want_eps=True, offset_dev=None):
    _require_gpu(loc, scale, offset_dev)
    out = torch.empty((rows, cols), dtype=loc.dtype, device=loc.device)
    eps = torch.empty_like(out) if want_eps else None
    check(_lib.load().pa_normal_rsample(_dtype(loc), _ptr(out), _ptr(eps), _view(loc, rows, cols),
                                        _view(scale, rows, cols), rows, cols, seed, offset,
                                        _ptr(offset_dev), _stream()))
    return out, eps


# ------------------------------------------------------------------------------------------
# many small sites in one launch (ELBO assembly, mean-field Normal guide)
# ------------------------------------------------------------------------------------------

def _reduced_shape(view, rows, cols):
    """Shape [rows or 1, cols or 1] of the gradient of an operand described by ``view``."""
    return (1 if (view.stride_row == 0 and rows > 1) else rows,
            1 if (view.stride_col == 0 and cols > 1) else cols)


def multi_log_prob_sum(entries, coef_all, dtype, device):
    """entries: list of dicts {dist, rows, cols, value, p0, p1, mask (2-D strided tensors or
    None), coef}.  Returns the 0-dim total  coef_all * sum_e coef_e * sum(masked log_prob_e)."""
    lib = _lib.load()
    out = torch.empty((), dtype=dtype, device=device)
    n = len(entries)
    for lo in range(0, max(n, 1), _lib.MULTI_MAX_ENTRIES):
        chunk = entries[lo:lo + _lib.MULTI_MAX_ENTRIES]
        arr = (_lib.SiteEntry * max(len(chunk), 1))()
        for k, e in enumerate(chunk):
            rows, cols = e["rows"], e["cols"]
            _require_gpu(e["value"], e["p0"], e["p1"], e["mask"])
            arr[k] = _lib.SiteEntry(e["dist"], 0, rows, cols, _view(e["value"], rows, cols),
                                    _view(e["p0"], rows, cols), _view(e["p1"], rows, cols),
                                    _view(e["mask"], rows, cols), float(e["coef"]), None, None, None,
                                    -1, 0, None, 0.0)
        check(lib.pa_multi_log_prob_sum(_DTYPES[dtype], _ptr(out), arr, len(chunk), float(coef_all),
                                        1 if lo > 0 else 0, _stream()))
    return out


def multi_log_prob_grad(g, entries, coef_all, dtype, device):
    """Backward of multi_log_prob_sum: ``entries`` as above plus ``need`` = (bool, bool, bool) and
    optionally ``chain_next`` (index of the next entry scoring the SAME value tensor: the head's
    d_value receives the sum), ``by_chain`` (this entry's value gradient comes from a chain head),
    ``extra_grad`` / ``extra_coef`` (a known gradient added to d_value).  Chains must not straddle
    a PA_MULTI_MAX_ENTRIES boundary.  Returns per entry a tuple (d_value, d_p0, d_p1) of tensors
    already reduced to the operand's own 2-D broadcast shape (None where not produced)."""
    lib = _lib.load()
    _require_gpu(g)
    assert g.numel() == 1
    g = g.reshape(()).contiguous()
    outs = []
    for lo in range(0, len(entries), _lib.MULTI_MAX_ENTRIES):
        chunk = entries[lo:lo + _lib.MULTI_MAX_ENTRIES]
        arr, grads = _grad_table(chunk, lo, dtype, device)
        outs += grads
        check(lib.pa_multi_log_prob_grad(_DTYPES[dtype], _ptr(g), arr, len(chunk), float(coef_all),
                                         _stream()))
    return outs


def _grad_table(chunk, lo, dtype, device):
    """ctypes entry table (with freshly allocated gradient outputs) of one launch's entries."""
    arr = (_lib.SiteEntry * len(chunk))()
    outs = []
    for k, e in enumerate(chunk):
        rows, cols = e["rows"], e["cols"]
        _require_gpu(e["value"], e["p0"], e["p1"], e["mask"])
        views = [_view(e["value"], rows, cols), _view(e["p0"], rows, cols),
                 _view(e["p1"], rows, cols)]
        grads, need_bits = [], 0
        by_chain = bool(e.get("by_chain", False))
        for j, (need, src) in enumerate(zip(e["need"], (e["value"], e["p0"], e["p1"]))):
            if need and src is not None and not (j == 0 and by_chain):
                grads.append(torch.empty(_reduced_shape(views[j], rows, cols), dtype=dtype,
                                         device=device))
                need_bits |= 1 << j
            else:
                grads.append(None)
        if by_chain:
            need_bits |= _lib.NEED_VALUE | _lib.VALUE_BY_CHAIN
        nxt = e.get("chain_next", -1)
        if nxt >= 0:
            assert lo <= nxt < lo + len(chunk), "a gradient chain straddles two launches"
            nxt -= lo
        xg = e.get("extra_grad")
        if xg is not None:
            _require_gpu(xg)
            assert xg.is_contiguous() and xg.numel() == rows * cols and xg.dtype == dtype
        arr[k] = _lib.SiteEntry(e["dist"], need_bits, rows, cols, views[0], views[1], views[2],
                                _view(e["mask"], rows, cols), float(e["coef"]),
                                _ptr(grads[0]), _ptr(grads[1]), _ptr(grads[2]), nxt, 0,
                                _ptr(xg), float(e.get("extra_coef", 0.0)))
        outs.append(tuple(grads))
    return arr, outs


def multi_log_prob_sum_grad(entries, coef_all, dtype, device):
    """multi_log_prob_sum AND multi_log_prob_grad (upstream gradient 1) of one table of at most
    PA_MULTI_MAX_ENTRIES entries in ONE launch: returns (total 0-dim, per-entry gradient tuples)."""
    assert 0 < len(entries) <= _lib.MULTI_MAX_ENTRIES
    out = torch.empty((), dtype=dtype, device=device)
    arr, grads = _grad_table(entries, 0, dtype, device)
    check(_lib.load().pa_multi_log_prob_sum_grad(_DTYPES[dtype], _ptr(out), None, arr, len(entries),
                                                 float(coef_all), 0, _stream()))
    return out, grads


def meanfield_normal_sample(locs, rhos, P, seed, offsets, offset_dev=None):
    """All mean-field Normal sites of a guide in one launch.  locs/rhos: lists of contiguous 1-D
    (flattened) parameter tensors; offsets: Philox block offset per site.  Returns lists
    (z [P, n], scale [n], loc_out [n], eps [P, n])."""
    lib = _lib.load()
    _require_gpu(*locs, *rhos, offset_dev)
    dtype, device = locs[0].dtype, locs[0].device
    zs, scales, louts, epss = [], [], [], []
    for lo in range(0, len(locs), _lib.MF_MAX_SITES):
        hi = min(lo + _lib.MF_MAX_SITES, len(locs))
        arr = (_lib.MfSite * (hi - lo))()
        for k in range(lo, hi):
            n = locs[k].numel()
            assert locs[k].is_contiguous() and rhos[k].is_contiguous() and rhos[k].numel() == n
            z = torch.empty((P, n), dtype=dtype, device=device)
            eps = torch.empty((P, n), dtype=dtype, device=device)
            sc = torch.empty((n,), dtype=dtype, device=device)
            lout = torch.empty((n,), dtype=dtype, device=device)
            zs.append(z); scales.append(sc); louts.append(lout); epss.append(eps)
            arr[k - lo] = _lib.MfSite(_ptr(locs[k]), _ptr(rhos[k]), _ptr(z), _ptr(sc), _ptr(lout),
                                      _ptr(eps), n, int(offsets[k]), 0, 0, None, None, None, None,
                                      None)
        check(lib.pa_meanfield_normal_sample(_DTYPES[dtype], arr, hi - lo, int(P), int(seed),
                                             _ptr(offset_dev), _stream()))
    return zs, scales, louts, epss


def meanfield_normal_sample_bwd(rhos, epss, d_zs, d_scales, d_louts, P, sinks=None):
    """Backward of meanfield_normal_sample: d_zs[k] [P, n], d_scales[k] [n], d_louts[k] [n] (each
    may be None = zero) -> lists (d_loc [n], d_rho [n]).  ``sinks[k]`` = (grad_loc, grad_rho)
    contiguous tensors the results are ADDED to instead (the optimizer's flat gradient views);
    the returned entries are None for those sites."""
    lib = _lib.load()
    dtype, device = rhos[0].dtype, rhos[0].device
    d_locs, d_rhos = [], []
    for lo in range(0, len(rhos), _lib.MF_MAX_SITES):
        hi = min(lo + _lib.MF_MAX_SITES, len(rhos))
        arr = (_lib.MfSite * (hi - lo))()
        keep = []
        for k in range(lo, hi):
            n = rhos[k].numel()
            dz, ds, dl_in = (None if t is None else t.contiguous()
                             for t in (d_zs[k], d_scales[k], d_louts[k]))
            _require_gpu(rhos[k], epss[k], dz, ds, dl_in)
            sink = None if sinks is None else sinks[k]
            if sink is not None:
                dl, dr = sink
                _require_gpu(dl, dr)
                assert dl.is_contiguous() and dr.is_contiguous() and dl.numel() == n == dr.numel()
                d_locs.append(None); d_rhos.append(None)
            else:
                dl = torch.empty((n,), dtype=dtype, device=device)
                dr = torch.empty((n,), dtype=dtype, device=device)
                d_locs.append(dl); d_rhos.append(dr)
            keep += [dz, ds, dl_in]
            arr[k - lo] = _lib.MfSite(None, _ptr(rhos[k]), None, None, None, _ptr(epss[k]), n, 0,
                                      1 if sink is not None else 0, 0, _ptr(dz), _ptr(ds),
                                      _ptr(dl_in), _ptr(dl), _ptr(dr))
        check(lib.pa_meanfield_normal_sample_bwd(_DTYPES[dtype], arr, hi - lo, int(P), _stream()))
        del keep
    return d_locs, d_rhos


def logchain_fwd_bwd(unary, pairwise):
    """Chain of enumerated variables summed out (pa_logchain_fwd_bwd): unary [B, T, K], pairwise
    [B, T-1, K, K] (contiguous; or [T-1, K, K] / [K, K] shared) ->
    (log_z [B], grad_unary [B, T, K], grad_pairwise [B, T-1, K, K])."""
    _require_gpu(unary, pairwise)
    B, T, K = unary.shape
    assert unary.is_contiguous() and (T == 1 or pairwise.is_contiguous())
    lib = _lib.load()
    spb = spt = 0
    if T > 1:
        assert pairwise.dtype == unary.dtype and pairwise.shape[-2:] == (K, K)
        if pairwise.dim() == 4:
            assert pairwise.shape[:2] == (B, T - 1)
            spb, spt = (T - 1) * K * K, K * K
        elif pairwise.dim() == 3:
            assert pairwise.shape[0] == T - 1
            spt = K * K
        else:
            assert pairwise.dim() == 2
    log_z = torch.empty((B,), dtype=unary.dtype, device=unary.device)
    g_u = torch.empty_like(unary)
    g_p = torch.empty((B, max(T - 1, 0), K, K), dtype=unary.dtype, device=unary.device)
    nbytes = lib.pa_logchain_workspace(_dtype(unary), B, T, K)
    ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=unary.device)
    check(lib.pa_logchain_fwd_bwd(_dtype(unary), _ptr(unary), _ptr(pairwise) if T > 1 else None,
                                  spb, spt, B, T, K, _ptr(log_z), _ptr(g_u), _ptr(g_p), _ptr(ws),
                                  nbytes, _stream()))
    return log_z, g_u, g_p


def mvn_tril_sample(loc, rho, A, P, seed=0, offset=0, offset_dev=None, eps=None):
    """Full-covariance Normal guide draw for P particles (pa_mvn_tril_sample): loc, rho [n],
    A [n, n] unconstrained; ``eps`` [P, n] given = use it instead of the Philox stream.
    Returns (z [P, n], logq [P], eps [P, n])."""
    _require_gpu(loc, rho, A, eps, offset_dev)
    n = loc.numel()
    assert loc.is_contiguous() and rho.is_contiguous() and A.is_contiguous()
    assert rho.numel() == n and A.shape == (n, n) and rho.dtype == loc.dtype == A.dtype
    given = eps is not None
    if given:
        assert eps.shape == (P, n) and eps.is_contiguous() and eps.dtype == loc.dtype
    else:
        eps = torch.empty((P, n), dtype=loc.dtype, device=loc.device)
    z = torch.empty((P, n), dtype=loc.dtype, device=loc.device)
    logq = torch.empty((P,), dtype=loc.dtype, device=loc.device)
    check(_lib.load().pa_mvn_tril_sample(_dtype(loc), _ptr(loc), _ptr(rho), _ptr(A), n, int(P),
                                         int(seed), int(offset), _ptr(offset_dev), int(given),
                                         _ptr(eps), _ptr(z), _ptr(logq), _stream()))
    return z, logq, eps


def mvn_tril_sample_bwd(loc, rho, eps, z, d_z, d_logq, sinks=None):
    """Backward of mvn_tril_sample -> (d_loc [n], d_rho [n], d_A [n, n]); ``sinks`` = three
    contiguous tensors the gradients are ADDED to instead (then None is returned for each)."""
    P, n = eps.shape
    d_z = None if d_z is None else d_z.contiguous()
    d_logq = None if d_logq is None else d_logq.contiguous()
    _require_gpu(loc, rho, eps, z, d_z, d_logq)
    if sinks is not None:
        d_loc, d_rho, d_A = sinks
        _require_gpu(d_loc, d_rho, d_A)
        assert all(t.is_contiguous() and t.dtype == loc.dtype for t in sinks)
        assert d_loc.numel() == n and d_rho.numel() == n and d_A.numel() == n * n
    else:
        d_loc = torch.empty((n,), dtype=loc.dtype, device=loc.device)
        d_rho = torch.empty((n,), dtype=loc.dtype, device=loc.device)
        d_A = torch.empty((n, n), dtype=loc.dtype, device=loc.device)
    check(_lib.load().pa_mvn_tril_sample_bwd(_dtype(loc), _ptr(loc), _ptr(rho), _ptr(eps), _ptr(z),
                                             _ptr(d_z), _ptr(d_logq), n, int(P), _ptr(d_loc),
                                             _ptr(d_rho), _ptr(d_A), int(sinks is not None),
                                             _stream()))
    return (None, None, None) if sinks is not None else (d_loc, d_rho, d_A)


# ------------------------------------------------------------------------------------------
# fused Bernoulli-logits GLM
# ------------------------------------------------------------------------------------------

GLM_AUTO, GLM_EXACT_F32, GLM_BF16X3 = 0, 1, 2


_glm_variant = GLM_AUTO


def glm_set_variant(variant):
    """Process-wide kernel choice of the fused GLM site: GLM_AUTO (default: vector-ALU streaming
    kernel for P <= 4, otherwise the bf16 matrix cores with 3-way split operands --
    f32-roundoff-class error), GLM_EXACT_F32 (f32 MFMA, bit-for-bit an fmaf chain) or GLM_BF16X3
    (the split-precision matrix-core kernel at every P)."""
    global _glm_variant
    check(_lib.load().pa_glm_set_variant(int(variant)))
    _glm_variant = int(variant)


# ---- the design matrix as its bf16 planes (pa_glm_pack_planes), cached per tensor OBJECT ---------
# X does not change between ELBO-gradient steps, so its exact 3-way bf16 split is computed once and
# kept beside it (1.5x the bytes of X).  An entry belongs to one tensor object (weak reference: it
# goes when the tensor goes) and remembers the tensor's version counter: an in-place update of X
# re-packs INTO THE SAME BUFFER, so a captured hipGraph that reads the image stays valid --
# glm_planes_revalidate() is called by SVI before every replay.
GLM_PLANES_OFF, GLM_PLANES_AUTO, GLM_PLANES_ALWAYS = 0, 1, 2
_planes_mode = GLM_PLANES_AUTO
# image formats (include/pyro_amd.h): three exact bf16 planes, or two scaled f16 planes (default:
# 4 B per element and 13 instead of 25 matrix instructions per tile; f32-class, see glm_planes16.h)
GLM_PLANES_BF16X3, GLM_PLANES_F16X2 = 0, 1
_planes_format = {"bf16x3": GLM_PLANES_BF16X3, "f16x2": GLM_PLANES_F16X2}[
    os.environ.get("PYRO_AMD_GLM_PLANES", "f16x2")]
_planes_cache = {}          # id(X) -> [weakref, version, planes or None, sightings]
_PLANES_MAX_D, _PLANES_MIN_P = 32, 33
_PLANES_MAX_D_F16 = 128            # the f16 image has feature tiles (csrc/glm_planes16d.h): D <= 128


def planes_max_d():
    """Largest feature count the plane image of the current format holds."""
    return _PLANES_MAX_D_F16 if _planes_format == GLM_PLANES_F16X2 else _PLANES_MAX_D


def glm_set_planes_mode(mode):
    """GLM_PLANES_AUTO (default): a design matrix seen for the second time is packed and the
    plane-image kernel used from then on; GLM_PLANES_ALWAYS: packed at first sight;
    GLM_PLANES_OFF: the design matrix is split on the fly every step."""
    global _planes_mode
    _planes_mode = int(mode)
    if _planes_mode == GLM_PLANES_OFF:
        _planes_cache.clear()


def glm_set_planes_format(fmt):
    """GLM_PLANES_F16X2 (default) or GLM_PLANES_BF16X3 for images packed from now on; cached images
    of the other format are dropped."""
    global _planes_format
    assert fmt in (GLM_PLANES_BF16X3, GLM_PLANES_F16X2)
    if fmt != _planes_format:
        _planes_cache.clear()
        for segs in list(_grouped_with_image):
            segs._planes = None
    _planes_format = int(fmt)


def glm_planes_format():
    return _planes_format


def _new_image(nbytes, device, fmt):
    out = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=device)
    out.pa_format = fmt           # the image buffer remembers its format
    return out


def _format_of(planes):
    return getattr(planes, "pa_format", GLM_PLANES_BF16X3)


def glm_planes_tune(ring_depth=0, blocks_per_cu=0):
    check(_lib.load().pa_glm_planes_tune(int(ring_depth), int(blocks_per_cu)))


def glm_planes_finalize_mode(in_kernel=False):
    """pa_glm_planes_finalize_mode: reduce the plane-image kernel's partial records with the separate
    finalize launch (default; a phase of the chained tail in a captured step) or inside the kernel
    (bit-identical, measured slower at large plates)."""
    check(_lib.load().pa_glm_planes_finalize_mode(int(bool(in_kernel))))


class GlmDeviceClock:
    """Duration of the plane-image GLM kernel from the device's own wall clock (pa_glm_planes_stamps):
    works inside a captured hipGraph.  ``arm()`` before a launch / replay, ``read_ms()`` after a
    device synchronise.  Must be created BEFORE a step is captured (the pointer is a launch
    argument)."""

    def __init__(self, device):
        self.buf = torch.zeros(2, dtype=torch.int64, device=device)
        check(_lib.load().pa_glm_planes_stamps(ctypes.c_void_p(self.buf.data_ptr())))
        self._init = torch.tensor([-1, 0], dtype=torch.int64, device=device)   # {UINT64_MAX, 0}

    def arm(self):
        self.buf.copy_(self._init)

    def read_ms(self):
        lo, hi = self.buf.tolist()
        if lo == -1 or hi == 0:
            return float("nan")
        return (hi - lo) / 100e6 * 1e3

    def close(self):
        check(_lib.load().pa_glm_planes_stamps(None))


def glm_pack_planes(X, out=None, fmt=None):
    """X[N,D] f32 (D <= 32) -> the uint8 tile image pa_glm_bernoulli_planes_fwd_bwd reads (``fmt``:
    GLM_PLANES_F16X2 / GLM_PLANES_BF16X3, default the process-wide format; ``out``: re-pack into an
    existing image, in its format)."""
    _require_gpu(X)
    N, D = X.shape
    lib = _lib.load()
    if out is not None:
        fmt = _format_of(out)
    elif fmt is None:
        fmt = _planes_format
    nbytes = lib.pa_glm_planes_bytes(fmt, N, D)
    if nbytes == 0 and N > 0:
        raise Unsupported("pyro_amd: no plane image for N=%d D=%d" % (N, D))
    if out is None:
        out = _new_image(nbytes, X.device, fmt)
    check(lib.pa_glm_pack_planes(fmt, _ptr(X), N, D, _ptr(out), nbytes, _stream()))
    return out


def _planes_entry_of(X):
    # X may be a fresh view object every step (``w @ X.t()`` hands over ``X.t().t()``): the entry
    # belongs to the view's BASE tensor (the user's long-lived object; all views share its
    # version counter) and the view's geometry
    base = X._base if X._base is not None else X
    key = (id(base), X.storage_offset(), tuple(X.shape), tuple(X.stride()))
    ent = _planes_cache.get(key)
    if ent is not None and ent[0]() is not base:     # the id was recycled
        ent = None
    if ent is None:
        import weakref
        ent = [weakref.ref(base, lambda _r, k=key: _planes_cache.pop(k, None)), X._version, None, 0,
               (X.storage_offset(), tuple(X.shape), tuple(X.stride())), {}]
        _planes_cache[key] = ent
    return ent


# ---- label moments (include/pyro_amd.h pa_glm_label_moments): cached per (image of X, y) -------------
LABEL_MOMENTS = {"on": os.environ.get("PYRO_AMD_GLM_LABEL_MOMENTS", "1") != "0"}


def glm_label_moments(X, y, out=None):
    """float64[33] = {sum_n (y_n - 1/2) X[n, d] for d < 32 (0 beyond D), sum_n (y_n - 1/2)}."""
    _require_gpu(X, y)
    N, D = X.shape
    assert X.is_contiguous() and y.is_contiguous() and X.dtype == torch.float32 and y.dtype == torch.float32
    lib = _lib.load()
    nbytes = lib.pa_glm_label_moments_workspace(N)
    ws = torch.empty((max(nbytes, 8),), dtype=torch.uint8, device=X.device)
    if out is None:
        out = torch.empty((33,), dtype=torch.float64, device=X.device)
    check(lib.pa_glm_label_moments(_ptr(X), _ptr(y), N, D, _ptr(out), _ptr(ws), nbytes, _stream()))
    return out


def glm_label_moments_of(X, y):
    """The cached moments of (X, y) for the plane-image kernel, or None (no f16 image of X yet, inside
    a capture without a cached entry, labels that change from call to call, switched off)."""
    if not LABEL_MOMENTS["on"] or _planes_format != GLM_PLANES_F16X2 or _planes_mode == GLM_PLANES_OFF:
        return None
    if not (X.is_contiguous() and y.is_contiguous() and y.dtype == torch.float32) or X.shape[1] > 32:
        return None
    ent = _planes_entry_of(X)
    if ent[2] is None or _format_of(ent[2]) != GLM_PLANES_F16X2:
        return None
    cache = ent[5]
    ybase = y._base if y._base is not None else y
    key = (id(ybase), y.storage_offset(), tuple(y.shape), tuple(y.stride()))
    m = cache.get(key)
    if m is not None and m[0]() is not ybase:
        m = None
    capturing = torch.cuda.is_current_stream_capturing()
    if m is None:
        if capturing or cache.get("misses", 0) > 8:
            return None                       # (labels that keep changing: the kernel sums the term itself)
        import weakref
        # (no eviction: a captured step holds the ADDRESS of the moments it was recorded with and SVI keeps
        # that graph as long as it keeps (X, y); the miss budget above bounds the entries at nine of 264 bytes)
        cache["misses"] = cache.get("misses", 0) + 1
        m = [weakref.ref(ybase), y._version, X._version, glm_label_moments(X, y),
             (y.storage_offset(), tuple(y.shape), tuple(y.stride()))]
        cache[key] = m
        return m[3]
    if m[1] != y._version or m[2] != X._version:
        if capturing:
            raise RuntimeError("pyro_amd: the observations of a GLM site changed in place during a graph capture")
        glm_label_moments(X, y, out=m[3])
        m[1], m[2] = y._version, X._version
    return m[3]


def _label_moments_stale(ent):
    base = ent[0]()
    for k, m in ent[5].items():
        if k == "misses":
            continue
        yb = m[0]()
        if base is not None and yb is not None and (m[1] != yb._version or m[2] != base._version):
            return True
    return False


def _label_moments_refresh(ent):
    base = ent[0]()
    if base is None:
        return
    off, shape, stride = ent[4]
    for k, m in list(ent[5].items()):
        if k == "misses":
            continue
        yb = m[0]()
        if yb is not None and (m[1] != yb._version or m[2] != base._version):
            yo, ys, yst = m[4]
            glm_label_moments(base.as_strided(shape, stride, off), yb.as_strided(ys, yst, yo), out=m[3])
            m[1], m[2] = yb._version, base._version


def glm_planes_of(X):
    """The cached plane image of X or None (not yet worth packing / mode off)."""
    if _planes_mode == GLM_PLANES_OFF:
        return None
    ent = _planes_entry_of(X)
    ent[3] += 1
    if ent[2] is None:
        if _planes_mode == GLM_PLANES_AUTO and ent[3] < 2:
            return None
        if torch.cuda.is_current_stream_capturing():
            return None                  # never allocate + pack inside a capture
        ent[2] = glm_pack_planes(X)
        ent[1] = X._version
    elif ent[1] != X._version:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("pyro_amd: a design matrix changed in place during a graph capture")
        glm_pack_planes(X, out=ent[2])
        ent[1] = X._version
    return ent[2]


def glm_planes_invalidate(X=None):
    """Forget the cached plane image of ``X`` (of every tensor when None).  The caches follow a tensor's
    version counter only: a write that does not bump it (``X.data.copy_()``, a numpy / DLPack alias, a
    kernel of another library) leaves a stale image behind -- call this after such a write.  A captured
    step that was reading the image must be captured again (``SVI`` does so when the argument
    signature changes; otherwise create a new ``SVI``)."""
    if X is None:
        _planes_cache.clear()
        for segs in list(_grouped_with_image):
            segs._planes = None
        return
    base = X._base if X._base is not None else X
    for key in [k for k, ent in _planes_cache.items() if ent[0]() is base]:
        _planes_cache.pop(key, None)
    for segs in list(_grouped_with_image):
        ent = segs._planes
        if ent is not None and (ent[0]() is X or ent[2]() is X):
            segs._planes = None


def glm_planes_revalidate():
    """Re-pack every cached image whose tensor was modified in place (called before a captured
    step is replayed: the graph reads the image, not X)."""
    for ent in list(_planes_cache.values()):
        base = ent[0]()
        if base is not None and ent[2] is not None and ent[1] != base._version:
            off, shape, stride = ent[4]
            glm_pack_planes(base.as_strided(shape, stride, off), out=ent[2])
            ent[1] = base._version
        if len(ent[5]) > 0 and _label_moments_stale(ent):
            _label_moments_refresh(ent)           # (the captured step reads the moments, not y)
    if not len(_grouped_with_image):              # (a WeakSet: iterating an empty one still costs a microsecond)
        return
    for segs in list(_grouped_with_image):
        ent = segs._planes
        if ent is None or ent[4] is None:
            continue
        X, y = ent[0](), ent[2]()
        if segs.ids is not None and segs.ids._version != segs.ids_version:
            # a captured step reads the segment table of the OLD ids: nothing can be patched in place
            raise RuntimeError("pyro_amd: the group-id tensor of a hierarchical GLM site was written in "
                               "place under a captured step; pass a new tensor instead")
        if X is not None and y is not None and (ent[1] != X._version or ent[3] != y._version):
            glm_pack_planes_grouped(X, y, segs, out=ent[4])
            ent[1], ent[3] = X._version, y._version


def revalidate_pending():
    """True when one of the revalidate hooks (plane images, LDA index, bag-of-words images) would
    re-pack something: a tensor behind a cached image was written in place since the image was made."""
    for ent in list(_planes_cache.values()):
        base = ent[0]()
        if base is not None and ent[2] is not None and ent[1] != base._version:
            return True
        if len(ent[5]) > 0 and _label_moments_stale(ent):
            return True
    for segs in list(_grouped_with_image):
        ent = segs._planes
        if ent is None or ent[4] is None:
            continue
        X, y = ent[0](), ent[2]()
        if segs.ids is not None and segs.ids._version != segs.ids_version:
            return True
        if X is not None and y is not None and (ent[1] != X._version or ent[3] != y._version):
            return True
    for cache in (_lda_index_cache, _bow_cache):
        for ent in list(cache.values()):
            base = ent[0]()
            if base is not None and ent[2] is not None and ent[1] != base._version:
                return True
    return False


def glm_bernoulli_planes_fwd_bwd(planes, y, w, b, scale, N, D, moments=None):
    """planes = glm_pack_planes(X[N,D]), y[N], w[P,D], b[P] or None -> (ll[P], gw[P,D], gb[P]).
    ``moments``: glm_label_moments(X, y) (f16 image, default tuning) or None."""
    _require_gpu(planes, y, w, b, moments)
    P = w.shape[0]
    assert y.is_contiguous() and w.is_contiguous() and y.shape == (N,) and w.shape == (P, D)
    assert y.dtype == torch.float32 and w.dtype == torch.float32
    if b is not None:
        assert b.is_contiguous() and b.shape == (P,)
    lib = _lib.load()
    nbytes = lib.pa_glm_bernoulli_planes_workspace(N, D, P)
    if nbytes == 0:
        raise Unsupported("pyro_amd: plane-image GLM kernel does not support N=%d D=%d P=%d" % (N, D, P))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=y.device)
    ll = torch.empty((P,), dtype=w.dtype, device=y.device)
    gw = torch.empty((P, D), dtype=w.dtype, device=y.device)
    gb = torch.empty((P,), dtype=w.dtype, device=y.device)
    check(lib.pa_glm_bernoulli_planes_fwd_bwd(_format_of(planes), _ptr(planes), _ptr(y), _ptr(w), _ptr(b), float(scale),
                                              N, D, P, _ptr(ll), _ptr(gw), _ptr(gb), _ptr(ws),
                                              nbytes, _ptr(moments), _stream()))
    return ll, gw, gb


def glm_bernoulli_fwd_bwd(X, y, w, b, mask, scale):
    """X[N,D], y[N], w[P,D], b[P] or None, mask[N] bool or None -> (ll[P], gw[P,D], gb[P])."""
    _require_gpu(X, y, w, b, mask)
    if X.dtype != torch.float32:
        raise Unsupported("pyro_amd: fused GLM kernel is float32 only")
    N, D = X.shape
    P = w.shape[0]
    assert X.is_contiguous() and y.is_contiguous() and w.is_contiguous()
    assert y.shape == (N,) and w.shape == (P, D)
    if b is not None:
        assert b.is_contiguous() and b.shape == (P,)
    if mask is not None:
        assert mask.is_contiguous() and mask.shape == (N,) and mask.dtype in (torch.bool,
                                                                             torch.uint8)
    if (mask is None and D <= planes_max_d() and P >= _PLANES_MIN_P and N > 0
            and _glm_variant == GLM_AUTO):
        planes = glm_planes_of(X)
        if planes is not None:
            return glm_bernoulli_planes_fwd_bwd(planes, y, w, b, scale, N, D,
                                                moments=glm_label_moments_of(X, y))
    lib = _lib.load()
    nbytes = lib.pa_glm_bernoulli_workspace(N, D, P)
    if nbytes == 0:
        raise Unsupported("pyro_amd: fused GLM kernel does not support N=%d D=%d P=%d" % (N, D, P))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=X.device)
    ll = torch.empty((P,), dtype=X.dtype, device=X.device)
    gw = torch.empty((P, D), dtype=X.dtype, device=X.device)
    gb = torch.empty((P,), dtype=X.dtype, device=X.device)
    check(lib.pa_glm_bernoulli_fwd_bwd(_ptr(X), _ptr(y), _ptr(w), _ptr(b), _ptr(mask),
                                       float(scale), N, D, P, _ptr(ll), _ptr(gw), _ptr(gb),
                                       _ptr(ws), nbytes, _stream()))
    return ll, gw, gb


def glm_chain(g, gw, gb, need_w=True, need_b=True):
    """(g[:, None...] * gw, g * gb) in one launch: the backward of the fused GLM site."""
    _require_gpu(g, gw, gb)
    P = g.numel()
    g = g.reshape(P).contiguous()
    assert gw.is_contiguous() and gw.shape[0] == P and gw.dtype == torch.float32
    W = gw.numel() // max(P, 1)
    dw = torch.empty_like(gw) if need_w else None
    db = torch.empty_like(gb) if (need_b and gb is not None) else None
    check(_lib.load().pa_glm_chain(_ptr(g), _ptr(gw), _ptr(gb), P, W, _ptr(dw), _ptr(db), _stream()))
    return dw, db


class GroupSegments:
    """Work description of the grouped GLM kernel for rows sorted by group: built once per data
    set from ``group_offsets`` (host int64 [G+1], rows of group g are offsets[g]:offsets[g+1])."""

    def __init__(self, group_offsets, device, target_segments=4096):
        import numpy as np
        off = np.asarray(group_offsets, dtype=np.int64)
        assert off.ndim == 1 and off.size >= 2 and (np.diff(off) >= 0).all() and off[0] == 0
        self.G = off.size - 1
        self.N = int(off[-1])
        # rows per segment: a multiple of 128 (= 4 waves x 32-row tiles) giving ~target_segments
        rows = max(128, -(-self.N // max(target_segments, 1)))
        rows = -(-rows // 128) * 128
        seg, gso = [], [0]
        for g in range(self.G):
            a, b = int(off[g]), int(off[g + 1])
            while a < b:
                e = min(a + rows, b)
                seg.append((a, e, g))
                a = e
            gso.append(len(seg))
        self.nseg = len(seg)
        self.max_seg_rows = rows if seg else 0
        self.seg = torch.tensor(seg if seg else [(0, 0, 0)], dtype=torch.int64, device=device)
        self.group_seg_off = torch.tensor(gso, dtype=torch.int64, device=device)
        self.group_offsets = off
        # the plane image of this row partition (pa_glm_pack_planes_grouped): every segment starts on
        # a 64-row super-tile boundary
        st = [0]
        for a, e, _ in seg:
            st.append(st[-1] + (e - a + 63) // 64)
        self.nst_total = st[-1]
        self.st_off = torch.tensor(st, dtype=torch.int64, device=device)
        self._planes = None          # [weakref X, X version, weakref y, y version, image, sightings]
        # rows NOT sorted by group (grouped_rows_of): image row i is the data's row rows[i], and
        # ``ids`` is the unsorted id vector g itself (what the un-fused formulation gathers with)
        self.rows = None
        self.ids = None
        self.ids_version = 0


# ---- the plane image of a grouped design matrix (belongs to the GroupSegments object) ------------
import weakref as _weakref  # noqa: E402

_grouped_with_image = _weakref.WeakSet()


def glm_pack_planes_grouped(X, y, segs, out=None, fmt=None):
    """X[N,D] f32 (D <= 32), y[N], segs -> the uint8 image pa_glm_bernoulli_grouped_planes_fwd_bwd
    reads (tile planes, then y in the image's padded row order)."""
    _require_gpu(X, y)
    N, D = X.shape
    lib = _lib.load()
    if out is not None:
        fmt = _format_of(out)
    elif fmt is None:
        fmt = _planes_format
    nbytes = lib.pa_glm_grouped_planes_bytes(fmt, segs.nst_total, D)
    if nbytes == 0 and N > 0:
        raise Unsupported("pyro_amd: no grouped plane image for N=%d D=%d" % (N, D))
    assert X.is_contiguous() and y.is_contiguous() and X.dtype == torch.float32 == y.dtype
    if out is None:
        out = _new_image(nbytes, X.device, fmt)
    check(lib.pa_glm_pack_planes_grouped_rows(fmt, _ptr(X), _ptr(y), _ptr(segs.rows), N, D, _ptr(segs.seg),
                                              _ptr(segs.st_off), segs.nseg, segs.nst_total, _ptr(out),
                                              nbytes, _stream()))
    return out


def group_rows_build(g, G):
    """g: int64 [N] device, unsorted group ids -> (offsets int64 [G+1], rows int64 [N]) on the device:
    rows[offsets[k]:offsets[k+1]] = the n with g[n] == k, ascending (a stable counting sort;
    bit-exact against oracle/glm.py::group_rows).  Ids in [-G, 0) count from the end and anything outside
    [-G, G) raises IndexError, as torch's advanced indexing does; Unsupported beyond the kernel's limits."""
    _require_gpu(g)
    assert g.dtype == torch.int64 and g.dim() == 1 and g.is_contiguous()
    N, G = g.shape[0], int(G)
    if N:
        g = torch.where(g < 0, g + G, g)       # (built once per id tensor: the partition is cached)
    lib = _lib.load()
    nbytes = lib.pa_group_rows_workspace(N, G)
    if nbytes == 0:
        raise Unsupported("pyro_amd: no group-row index for N=%d G=%d" % (N, G))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=g.device)
    offsets = torch.empty((G + 1,), dtype=torch.int64, device=g.device)
    rows = torch.empty((N,), dtype=torch.int64, device=g.device)
    bad = torch.empty((1,), dtype=torch.int64, device=g.device)
    check(lib.pa_group_rows_build(_ptr(g), N, G, _ptr(offsets), _ptr(rows), _ptr(bad), _ptr(ws), nbytes,
                                  _stream()))
    nbad = int(bad.item())
    if nbad:
        raise IndexError("pyro_amd: %d group ids are outside [-%d, %d)" % (nbad, G, G))
    return offsets, rows


_group_rows_cache = {}      # id(g) -> [weakref g, version, G, GroupSegments]


def glm_grouped_rows_servable(X, y, mask, segs):
    """Can the fused grouped kernel serve this call?  Always for rows sorted by group; rows in their
    original order (segs.rows) only through the plane image."""
    if segs.rows is None:
        return True
    if (mask is not None or X.shape[1] > _PLANES_MAX_D or _planes_mode == GLM_PLANES_OFF
            or X.dtype != torch.float32 or y.dtype != torch.float32 or not X.is_contiguous()
            or X.shape[0] == 0):
        return False
    if segs.ids._version != segs.ids_version:
        return False            # stale partition (ids written in place): the caller re-derives it
    ent = segs._planes
    have = ent is not None and ent[4] is not None and ent[0]() is X and ent[2]() is y
    return have or not torch.cuda.is_current_stream_capturing()


def grouped_rows_of(g, G):
    """The GroupSegments of an UNSORTED id vector (built once per tensor object and version: the sort
    permutation, the segment table of the sorted order); None inside a graph capture when it does
    not exist yet (the sort reads its offsets back to the host)."""
    ent = _group_rows_cache.get(id(g))
    if ent is not None and (ent[0]() is not g or ent[2] != G):
        ent = None
    if ent is not None and ent[1] == g._version:
        return ent[3]
    if torch.cuda.is_current_stream_capturing():
        if ent is not None:
            raise RuntimeError("pyro_amd: group ids changed in place during a graph capture")
        return None
    offsets, rows = group_rows_build(g, G)
    if ent is not None:
        # same tensor, new contents: the partition (and with it every shape downstream) changes
        _grouped_with_image.discard(ent[3])
    segs = GroupSegments(offsets.cpu().numpy(), g.device)
    segs.rows, segs.ids, segs.ids_version = rows, g, g._version
    for k in [k for k, e in _group_rows_cache.items() if e[0]() is None]:
        del _group_rows_cache[k]
    _group_rows_cache[id(g)] = [_weakref.ref(g), g._version, G, segs]
    return segs


def glm_grouped_planes_of(X, y, segs):
    """The cached image of (X, y) under ``segs`` or None; same policy as glm_planes_of (second
    sighting, never packed inside a capture, re-packed into the same buffer when X or y changed in
    place)."""
    if _planes_mode == GLM_PLANES_OFF:
        return None
    ent = segs._planes
    if ent is not None and (ent[0]() is not X or ent[2]() is not y):
        ent = None                                     # another data set under the same partition
    if ent is None:
        ent = [_weakref.ref(X), X._version, _weakref.ref(y), y._version, None, 0]
        segs._planes = ent
    ent[5] += 1
    capturing = torch.cuda.is_current_stream_capturing()
    if ent[4] is None:
        # (rows in their original order have no other fused kernel: packed at first sight)
        if (_planes_mode == GLM_PLANES_AUTO and ent[5] < 2 and segs.rows is None) or capturing:
            return None
        ent[4] = glm_pack_planes_grouped(X, y, segs)
        ent[1], ent[3] = X._version, y._version
        _grouped_with_image.add(segs)
    elif ent[1] != X._version or ent[3] != y._version:
        if capturing:
            raise RuntimeError("pyro_amd: a design matrix changed in place during a graph capture")
        glm_pack_planes_grouped(X, y, segs, out=ent[4])
        ent[1], ent[3] = X._version, y._version
    return ent[4]


def glm_bernoulli_grouped_planes_fwd_bwd(planes, w, b, scale, N, D, segs):
    """planes = glm_pack_planes_grouped(X, y, segs), w[P,G,D], b[P] or None ->
    (ll[P], gw[P,G,D], gb[P])."""
    _require_gpu(planes, w, b)
    P, G = w.shape[0], w.shape[1]
    assert w.is_contiguous() and w.shape == (P, G, D) and G == segs.G and w.dtype == torch.float32
    if b is not None:
        assert b.is_contiguous() and b.shape == (P,)
    lib = _lib.load()
    nbytes = lib.pa_glm_bernoulli_grouped_planes_workspace(segs.nseg, P)
    ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=w.device)
    ll = torch.empty((P,), dtype=w.dtype, device=w.device)
    gw = torch.empty((P, G, D), dtype=w.dtype, device=w.device)
    gb = torch.empty((P,), dtype=w.dtype, device=w.device)
    check(lib.pa_glm_bernoulli_grouped_planes_fwd_bwd(
        _format_of(planes), _ptr(planes), _ptr(w), _ptr(b), float(scale), N, D, P, G, _ptr(segs.seg), _ptr(segs.st_off),
        segs.nseg, _ptr(segs.group_seg_off), segs.nst_total, _ptr(ll), _ptr(gw), _ptr(gb), _ptr(ws),
        nbytes, _stream()))
    return ll, gw, gb


def glm_bernoulli_grouped_fwd_bwd(X, y, w, b, mask, scale, segs):
    """X[N,D] (rows sorted by group), y[N], w[P,G,D], b[P] or None, mask[N] or None, segs:
    GroupSegments -> (ll[P], gw[P,G,D], gb[P])."""
    _require_gpu(X, y, w, b, mask)
    if X.dtype != torch.float32:
        raise Unsupported("pyro_amd: fused GLM kernel is float32 only")
    if (mask is None and X.shape[1] <= _PLANES_MAX_D
            and (w.shape[0] >= _PLANES_MIN_P or segs.rows is not None)
            and X.shape[0] > 0 and (_glm_variant == GLM_AUTO or segs.rows is not None)
            and X.is_contiguous() and y.is_contiguous() and y.dtype == torch.float32):
        planes = glm_grouped_planes_of(X, y, segs)
        if planes is not None:
            return glm_bernoulli_grouped_planes_fwd_bwd(planes, w, b, scale, X.shape[0], X.shape[1],
                                                        segs)
    if segs.rows is not None:
        raise Unsupported("pyro_amd: rows in their original order are only served by the plane image "
                          "(no mask, D <= 32, float32, image packed outside a capture)")
    N, D = X.shape
    P, G = w.shape[0], w.shape[1]
    assert X.is_contiguous() and y.is_contiguous() and w.is_contiguous()
    assert w.shape == (P, G, D) and G == segs.G and N == segs.N and y.shape == (N,)
    if b is not None:
        assert b.is_contiguous() and b.shape == (P,)
    if mask is not None:
        assert mask.is_contiguous() and mask.shape == (N,)
    lib = _lib.load()
    nbytes = lib.pa_glm_bernoulli_grouped_workspace(segs.nseg, D, P)
    if nbytes == 0 and segs.nseg > 0:
        raise Unsupported("pyro_amd: grouped GLM kernel does not support D=%d P=%d" % (D, P))
    ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=X.device)
    ll = torch.empty((P,), dtype=X.dtype, device=X.device)
    gw = torch.empty((P, G, D), dtype=X.dtype, device=X.device)
    gb = torch.empty((P,), dtype=X.dtype, device=X.device)
    check(lib.pa_glm_bernoulli_grouped_fwd_bwd(
        _ptr(X), _ptr(y), _ptr(w), _ptr(b), _ptr(mask), float(scale), N, D, P, G, _ptr(segs.seg),
        segs.nseg, _ptr(segs.group_seg_off), segs.max_seg_rows, _ptr(ll), _ptr(gw), _ptr(gb),
        _ptr(ws), nbytes, _stream()))
    return ll, gw, gb


# ------------------------------------------------------------------------------------------
# HMC / NUTS
# ------------------------------------------------------------------------------------------

def leapfrog_kick_drift(z, r, grad, inv_mass, step):
    """In place: r -= 0.5*eps*grad; z += eps*inv_mass*r. z,r,grad: [C,D]; inv_mass [D] or [C,D];
    step: [C] or 0-dim tensor."""
    _require_gpu(z, r, grad, inv_mass, step)
    C, D = z.shape
    assert z.is_contiguous() and r.is_contiguous() and grad.is_contiguous()
    assert inv_mass.is_contiguous() and step.is_contiguous()
    im_stride = D if inv_mass.dim() == 2 else 0
    st_stride = 1 if step.numel() == C and step.dim() == 1 and C > 1 else (1 if step.dim() == 1 and step.numel() == C else 0)
    check(_lib.load().pa_leapfrog_kick_drift(_dtype(z), _ptr(z), _ptr(r), _ptr(grad),
                                             _ptr(inv_mass), im_stride, _ptr(step), st_stride, C,
                                             D, _stream()))


def chain_matvec(M, x, transpose=False):
    """y[c] = M[c] @ x[c] (or M[c]^T @ x[c]): M [C, D, D] (or [D, D] shared by all chains),
    x [C, D]; one launch for all chains (pa_chain_matvec)."""
    _require_gpu(M, x)
    C, D = x.shape
    assert M.shape[-2:] == (D, D) and M.dim() in (2, 3) and M.dtype == x.dtype
    assert M.is_contiguous() and x.is_contiguous()
    assert M.dim() == 2 or M.shape[0] == C
    y = torch.empty_like(x)
    check(_lib.load().pa_chain_matvec(_dtype(x), _ptr(M), D * D if M.dim() == 3 else 0, _ptr(x),
                                      _ptr(y), C, D, int(bool(transpose)), _stream()))
    return y


def leapfrog_kick(r, grad, step):
    _require_gpu(r, grad, step)
    C, D = r.shape
    assert r.is_contiguous() and grad.is_contiguous() and step.is_contiguous()
    st_stride = 1 if step.dim() == 1 and step.numel() == C else 0
    check(_lib.load().pa_leapfrog_kick(_dtype(r), _ptr(r), _ptr(grad), _ptr(step), st_stride, C,
                                       D, _stream()))


def nuts_gaussian_transition(z, pe, grad, Lambda, inv_mass, step, max_tree_depth, use_multinomial,
                             seed, t, chain_offset=0):
    """One NUTS transition for all chains, in place on (z, pe, grad).
    Returns dict(accept_prob[C], n_leapfrog, depth, diverging, accepted: int32[C])."""
    _require_gpu(z, pe, grad, Lambda, inv_mass, step)
    C, D = z.shape
    for x in (z, pe, grad, Lambda, inv_mass, step):
        assert x.is_contiguous() and x.dtype == z.dtype
    assert Lambda.shape == (D, D) and inv_mass.shape == (C, D) and step.shape == (C,)
    ap = torch.empty((C,), dtype=z.dtype, device=z.device)
    ints = torch.empty((4, C), dtype=torch.int32, device=z.device)
    check(_lib.load().pa_nuts_gaussian_transition(
        _dtype(z), _ptr(z), _ptr(pe), _ptr(grad), _ptr(Lambda), _ptr(inv_mass), _ptr(step), C, D,
        int(max_tree_depth), int(bool(use_multinomial)), int(seed), int(t), int(chain_offset),
        _ptr(ap),
        _ptr(ints[0]), _ptr(ints[1]), _ptr(ints[2]), _ptr(ints[3]), _stream()))
    return {"accept_prob": ap, "n_leapfrog": ints[0], "depth": ints[1], "diverging": ints[2],
            "accepted": ints[3]}


def nuts_gaussian_find_step(z, pe, grad, Lambda, inv_mass, step, seed, key, chain_offset,
                            min_step, max_step, direction_threshold):
    """Reasonable step size per chain (hmc.py:170-229) for the Gaussian potential, one launch:
    returns a new [C] tensor; ``step`` is the starting point."""
    _require_gpu(z, pe, grad, Lambda, inv_mass, step)
    C, D = z.shape
    out = step.contiguous().clone()
    im = inv_mass if inv_mass.dim() == 2 else inv_mass.expand(C, D)
    check(_lib.load().pa_nuts_gaussian_find_step(
        _dtype(z), _ptr(z.contiguous()), _ptr(pe.contiguous()), _ptr(grad.contiguous()),
        _ptr(Lambda), _ptr(im.contiguous()), _ptr(out), C, D, int(seed), int(key),
        int(chain_offset), float(min_step), float(max_step), float(direction_threshold), _stream()))
    return out


def nuts_gaussian_run(z, pe, grad, Lambda, inv_mass, step, max_tree_depth, use_multinomial, seed,
                      t0, num_transitions, chain_offset=0, da_state=None, target_accept=0.8,
                      welford=None, welford_n0=0, samples=None, mean_accept=None, mean_n0=0,
                      counters=None, count_accepts=False, div_flags=None):
    """``num_transitions`` NUTS transitions per chain in ONE launch, in place on (z, pe, grad) and,
    when adapting, on (step, da_state [C,5], welford [C,2,D]).  Returns the statistics of the last
    transition like nuts_gaussian_transition."""
    _require_gpu(z, pe, grad, Lambda, inv_mass, step, da_state, welford, samples, mean_accept,
                 counters, div_flags)
    C, D = z.shape
    K = int(num_transitions)
    for x in (z, pe, grad, Lambda, inv_mass, step, da_state, welford, samples, mean_accept):
        assert x is None or (x.is_contiguous() and x.dtype == z.dtype)
    assert Lambda.shape == (D, D) and inv_mass.shape == (C, D) and step.shape == (C,)
    assert da_state is None or da_state.shape == (C, 5)
    assert welford is None or welford.shape == (C, 2, D)
    assert samples is None or samples.shape == (K, C, D)
    assert counters is None or (counters.shape == (3, C) and counters.dtype == torch.int64
                                and counters.is_contiguous())
    assert div_flags is None or (div_flags.shape == (K, C) and div_flags.dtype == torch.int8
                                 and div_flags.is_contiguous())
    ap = torch.empty((C,), dtype=z.dtype, device=z.device)
    ints = torch.empty((4, C), dtype=torch.int32, device=z.device)
    check(_lib.load().pa_nuts_gaussian_run(
        _dtype(z), _ptr(z), _ptr(pe), _ptr(grad), _ptr(Lambda), _ptr(inv_mass), _ptr(step), C, D,
        int(max_tree_depth), int(bool(use_multinomial)), int(seed), int(t0), K, int(chain_offset),
        _ptr(da_state), float(target_accept), _ptr(welford), int(welford_n0), _ptr(samples),
        _ptr(mean_accept), int(mean_n0), _ptr(counters), int(bool(count_accepts)), _ptr(div_flags),
        _ptr(ap), _ptr(ints[0]), _ptr(ints[1]), _ptr(ints[2]), _ptr(ints[3]), _stream()))
    return {"accept_prob": ap, "n_leapfrog": ints[0], "depth": ints[1], "diverging": ints[2],
            "accepted": ints[3]}


class NutsTree:
    """Device-resident NUTS tree state for C chains of dimension D (pa_nuts_tree_*).

    Protocol per transition:  ``begin(t)``; then until ``n_active() == 0``: the caller writes
    the potential energy / gradient at ``zq`` into ``peq`` / ``gq`` and calls ``advance()``.
    ``z, pe, grad`` are the chains' current state (updated in place when a proposal is accepted).
    """

    def __init__(self, z, pe, grad, inv_mass, step, max_tree_depth=10, use_multinomial=True,
                 seed=0, chain_offset=0):
        _require_gpu(z, pe, grad, inv_mass, step)
        C, D = z.shape
        for x in (z, pe, grad, inv_mass, step):
            assert x.is_contiguous() and x.dtype == z.dtype
        assert pe.shape == (C,) and grad.shape == (C, D) and step.shape == (C,)
        assert inv_mass.shape in ((D,), (C, D))
        self.z, self.pe, self.grad, self.inv_mass, self.step = z, pe, grad, inv_mass, step
        self.C, self.D = C, D
        self.im_stride = D if inv_mass.dim() == 2 else 0
        self.max_tree_depth, self.multinomial = int(max_tree_depth), int(bool(use_multinomial))
        self.seed, self.chain_offset = int(seed), int(chain_offset)
        self.dt = _dtype(z)
        lib = _lib.load()
        nbytes = lib.pa_nuts_tree_workspace(self.dt, C, D, self.max_tree_depth)
        if nbytes == 0 and C > 0:
            raise Unsupported("pyro_amd: NUTS tree kernel does not support C=%d D=%d depth=%d"
                              % (C, D, max_tree_depth))
        self.nbytes = nbytes
        self.ws = torch.zeros((max(nbytes, 16),), dtype=torch.uint8, device=z.device)
        self.zq = torch.empty_like(z)
        self.rq = torch.empty_like(z)
        self.accept_prob = torch.zeros((C,), dtype=z.dtype, device=z.device)
        self.ints = torch.zeros((4, C), dtype=torch.int32, device=z.device)
        self._n_active = torch.zeros((1,), dtype=torch.int32, device=z.device)
        self.t = 0
        self.t_dev = torch.zeros((1,), dtype=torch.int64, device=z.device)   # for advance_replayable

    def begin(self, t):
        self.t = int(t)
        self.t_dev.fill_(self.t)
        check(_lib.load().pa_nuts_tree_begin(
            self.dt, _ptr(self.z), _ptr(self.pe), _ptr(self.grad), _ptr(self.zq), _ptr(self.rq),
            _ptr(self.inv_mass), self.im_stride, _ptr(self.step), self.C, self.D,
            self.max_tree_depth, self.multinomial, self.seed, self.t, self.chain_offset,
            _ptr(self.ws), self.nbytes, _stream()))

    def advance(self, peq, gq):
        """peq[C], gq[C,D]: potential energy and gradient at the cursor ``zq``."""
        _require_gpu(peq, gq)
        assert peq.is_contiguous() and gq.is_contiguous() and gq.dtype == self.z.dtype
        assert peq.dtype == self.z.dtype and gq.shape == self.z.shape and peq.numel() == self.C
        check(_lib.load().pa_nuts_tree_advance(
            self.dt, _ptr(self.z), _ptr(self.pe), _ptr(self.grad), _ptr(self.zq), _ptr(self.rq),
            _ptr(gq), _ptr(peq), _ptr(self.inv_mass), self.im_stride, _ptr(self.step),
            self.C, self.D, self.max_tree_depth, self.multinomial, self.seed, self.t,
            self.chain_offset, _ptr(self.accept_prob), _ptr(self.ints[0]), _ptr(self.ints[1]),
            _ptr(self.ints[2]), _ptr(self.ints[3]), _ptr(self._n_active), _ptr(self.ws),
            self.nbytes, _stream()))

    def advance_replayable(self, peq, gq):
        """advance() whose transition index comes from device memory (set by begin()): the launch
        may be captured into a hipGraph once and replayed for every leapfrog of every transition
        (pa_nuts_tree_advance_tdev).  inv_mass / step must be persistent buffers."""
        _require_gpu(peq, gq)
        assert peq.is_contiguous() and gq.is_contiguous() and gq.dtype == self.z.dtype
        assert peq.dtype == self.z.dtype and gq.shape == self.z.shape and peq.numel() == self.C
        check(_lib.load().pa_nuts_tree_advance_tdev(
            self.dt, _ptr(self.z), _ptr(self.pe), _ptr(self.grad), _ptr(self.zq), _ptr(self.rq),
            _ptr(gq), _ptr(peq), _ptr(self.inv_mass), self.im_stride, _ptr(self.step),
            self.C, self.D, self.max_tree_depth, self.multinomial, self.seed, _ptr(self.t_dev),
            self.chain_offset, _ptr(self.accept_prob), _ptr(self.ints[0]), _ptr(self.ints[1]),
            _ptr(self.ints[2]), _ptr(self.ints[3]), _ptr(self._n_active), _ptr(self.ws),
            self.nbytes, _stream()))

    def n_active(self):
        """Number of chains still building their tree (host synchronisation)."""
        return int(self._n_active.item())

    # ---- asynchronous chains: spans of K transitions per chain (pa_nuts_tree_run_*) ------------
    RUN_ADAPT_STEP, RUN_WELFORD, RUN_COUNT_ACCEPTS = 1, 2, 4

    def run_buffers(self):
        """Span control words and per-chain progress (allocated once: a captured graph of rounds
        holds their addresses)."""
        if getattr(self, "ctl", None) is None:
            dev = self.z.device
            self.ctl = torch.zeros((8,), dtype=torch.int64, device=dev)
            self.tc = torch.zeros((self.C,), dtype=torch.int32, device=dev)
            self.n_done = torch.zeros((1,), dtype=torch.int32, device=dev)
            self.gate = torch.zeros((2,), dtype=torch.int64, device=dev)   # [1]: the abort word
        return self.ctl

    def set_span(self, t0, K, mean_n0=0, welford_n0=0, flags=0, samples=None, div_flags=None, row0=0):
        """Transitions t0 .. t0+K-1 of every chain; ``samples`` [rows, C, D] / ``div_flags``
        [rows, C] int8 receive transition t0 + k at row row0 + k."""
        self.run_buffers()
        if samples is not None:
            _require_gpu(samples)
            assert samples.is_contiguous() and samples.dtype == self.z.dtype
            assert samples.shape[1:] == self.z.shape and samples.shape[0] >= row0 + K
        if div_flags is not None:
            _require_gpu(div_flags)
            assert div_flags.is_contiguous() and div_flags.dtype == torch.int8
            assert div_flags.shape[1] == self.C and div_flags.shape[0] >= row0 + K
        self._span_keep = (samples, div_flags)
        words = [int(t0), int(K), int(mean_n0), int(welford_n0), int(flags),
                 0 if samples is None else samples.data_ptr(),
                 0 if div_flags is None else div_flags.data_ptr(), int(row0)]
        self.ctl.copy_(torch.tensor(words, dtype=torch.int64))

    def run_begin(self):
        check(_lib.load().pa_nuts_tree_run_begin(
            self.dt, _ptr(self.z), _ptr(self.pe), _ptr(self.grad), _ptr(self.zq), _ptr(self.rq),
            _ptr(self.inv_mass), self.im_stride, _ptr(self.step), self.C, self.D,
            self.max_tree_depth, self.multinomial, self.seed, self.chain_offset, _ptr(self.ctl),
            _ptr(self.tc), _ptr(self.n_done), _ptr(self.gate[1:]), _ptr(self.ws), self.nbytes,
            _stream()))

    def compact(self, n_slots, program=None):
        """Slots of a COMPACTED round: the chains still building a tree (ascending, -1 pads) and their
        cursor rows.  Returns (slot2chain int32[n_slots], zq_slot [n_slots, D]) -- persistent per size (a
        captured graph of rounds of that size holds their addresses) -- after refreshing them; the caller
        makes sure n_slots >= the number of active chains.  With ``program`` (infer/mcmc/direct.py) the
        cursor buffer is SITE-MAJOR (the sites' blocks [n_slots, len] one after the other) and
        ``n_slots == C`` is the full round (slot2chain None: the identity)."""
        bufs = getattr(self, "_slots", None)
        if bufs is None:
            bufs = self._slots = {}
            self._n_placed = torch.zeros((1,), dtype=torch.int32, device=self.z.device)
        key = (n_slots, program is not None)
        if key not in bufs:
            full = program is not None and n_slots == self.C
            bufs[key] = (None if full else torch.full((n_slots,), -1, dtype=torch.int32, device=self.z.device),
                         torch.zeros((n_slots * self.D,), dtype=self.z.dtype, device=self.z.device)
                         if program is not None else
                         torch.zeros((n_slots, self.D), dtype=self.z.dtype, device=self.z.device))
        s2c, zqs = bufs[key]
        check(_lib.load().pa_nuts_tree_compact(
            self.dt, _ptr(self.zq), self.C, self.D, self.max_tree_depth, _ptr(s2c), _ptr(zqs), n_slots,
            _ptr(self._n_placed), 0 if program is None else program.n,
            None if program is None else program.off, None if program is None else program.len,
            _ptr(self.ws), self.nbytes, _stream()))
        return s2c, zqs

    def run_advance_direct(self, program, ll_ext, g_ext, da_state, target_accept, welford, mean_accept,
                           counters, slots):
        """run_advance for a flat model (infer/mcmc/direct.DirectProgram): the latent sites' own
        log-densities and gradients are computed by the tree kernel, ``ll_ext`` [n_slots] / ``g_ext``
        {site: [n_slots, len]} come from the observed site's kernel; ``slots`` = compact(n, program)."""
        s2c, pack = slots
        n_slots = self.C if s2c is None else s2c.numel()
        _require_gpu(ll_ext, pack, da_state, welford, mean_accept, counters)
        assert self.z.dtype == torch.float32 and ll_ext.is_contiguous() and ll_ext.numel() == n_slots
        for t in g_ext.values():
            assert t.is_contiguous() and t.dtype == torch.float32
        p0, p1, ge = program.pointers(g_ext)
        check(_lib.load().pa_nuts_tree_run_advance_direct(
            _ptr(self.z), _ptr(self.pe), _ptr(self.grad), _ptr(self.zq), _ptr(self.rq), _ptr(self.inv_mass),
            self.im_stride, _ptr(self.step), self.C, self.D, self.max_tree_depth, self.multinomial, self.seed,
            self.chain_offset, _ptr(self.ctl), _ptr(da_state), float(target_accept), _ptr(welford),
            _ptr(mean_accept), _ptr(counters), _ptr(self.tc), _ptr(self.n_done), _ptr(self.gate[1:]),
            _ptr(s2c), _ptr(pack), n_slots, program.n, program.off, program.len, program.dist,
            program.transform, program.lower, p0, program.s0, p1, program.s1, ge, _ptr(ll_ext),
            _ptr(self.accept_prob), _ptr(self.ints[0]), _ptr(self.ints[1]), _ptr(self.ints[2]),
            _ptr(self.ints[3]), _ptr(self.ws), self.nbytes, _stream()))

    def run_advance(self, peq, gq, da_state, target_accept, welford, mean_accept, counters, slots=None):
        """One round of a span: every live chain consumes (peq, gq) at its cursor; a chain whose
        tree finishes adapts, stores its draw and begins its next transition in the same launch.
        ``slots`` = (slot2chain, zq_slot) of ``compact``: a compacted round, (peq, gq) per slot."""
        _require_gpu(peq, gq, da_state, welford, mean_accept, counters)
        n_slots = self.C if slots is None else slots[0].numel()
        assert peq.is_contiguous() and gq.is_contiguous() and gq.dtype == self.z.dtype
        assert peq.dtype == self.z.dtype and gq.shape == (n_slots, self.D) and peq.numel() == n_slots
        assert da_state.shape == (self.C, 5) and welford.shape == (self.C, 2, self.D)
        assert da_state.dtype == welford.dtype == mean_accept.dtype == self.z.dtype
        assert counters.shape == (3, self.C) and counters.dtype == torch.int64
        for x in (da_state, welford, mean_accept, counters):
            assert x.is_contiguous()
        check(_lib.load().pa_nuts_tree_run_advance(
            self.dt, _ptr(self.z), _ptr(self.pe), _ptr(self.grad), _ptr(self.zq), _ptr(self.rq),
            _ptr(gq), _ptr(peq), _ptr(self.inv_mass), self.im_stride, _ptr(self.step),
            self.C, self.D, self.max_tree_depth, self.multinomial, self.seed, self.chain_offset,
            _ptr(self.ctl), _ptr(da_state), float(target_accept), _ptr(welford), _ptr(mean_accept),
            _ptr(counters), _ptr(self.tc), _ptr(self.n_done), _ptr(self.gate[1:]),
            None if slots is None else _ptr(slots[0]), None if slots is None else _ptr(slots[1]), n_slots,
            _ptr(self.accept_prob), _ptr(self.ints[0]), _ptr(self.ints[1]), _ptr(self.ints[2]),
            _ptr(self.ints[3]), _ptr(self.ws), self.nbytes, _stream()))

    def chains_done(self):
        """How many chains have completed their span (host synchronisation)."""
        return int(self.n_done.item())

    def span_done(self):
        """True when every chain has completed its span (host synchronisation)."""
        return self.chains_done() >= self.C

    def stats(self):
        return {"accept_prob": self.accept_prob, "n_leapfrog": self.ints[0], "depth": self.ints[1],
                "diverging": self.ints[2], "accepted": self.ints[3]}


# ------------------------------------------------------------------------------------------
# enumerated LDA factor
# ------------------------------------------------------------------------------------------

# ------------------------------------------------------------------------------------------
# Dirichlet log-density rows
# ------------------------------------------------------------------------------------------

def dirichlet_log_prob(value, concentration):
    """value [..., K] on the simplex, concentration [..., K] or [K] -> log-density [...]."""
    _require_gpu(value, concentration)
    K = value.shape[-1]
    batch = torch.broadcast_shapes(value.shape[:-1], concentration.shape[:-1])
    rows = 1
    for n in batch:
        rows *= int(n)
    keep = []
    views = []
    for t in (value, concentration):
        if t.numel() != K:
            t = t.expand(batch + (K,)).reshape(rows, K)     # (a copy only for a partial broadcast)
            keep.append(t)
            views.append(_lib.View2D(_ptr(t), t.stride(0), t.stride(1)))
        else:
            t = t.reshape(K)
            keep.append(t)
            views.append(_lib.View2D(_ptr(t), 0, t.stride(0)))
    out = torch.empty(batch, dtype=value.dtype, device=value.device)
    check(_lib.load().pa_dirichlet_log_prob(_dtype(value), _ptr(out), views[0], views[1], rows, K,
                                            _stream()))
    return out


def dirichlet_log_prob_grad(g, value, concentration, need_value, need_conc):
    """Gradients of sum(g * log_prob): (d_value, d_concentration) as [rows..., K] tensors of the
    BROADCAST batch shape (the caller reduces a broadcast operand's gradient), None where unwanted."""
    _require_gpu(g, value, concentration)
    K = value.shape[-1]
    batch = torch.broadcast_shapes(value.shape[:-1], concentration.shape[:-1])
    rows = 1
    for n in batch:
        rows *= int(n)
    keep, views = [], []
    for t in (value, concentration):
        if t.numel() != K:
            t = t.expand(batch + (K,)).reshape(rows, K)
            views.append(_lib.View2D(_ptr(t), t.stride(0), t.stride(1)))
        else:
            t = t.reshape(K)
            views.append(_lib.View2D(_ptr(t), 0, t.stride(0)))
        keep.append(t)
    g = g.expand(batch).contiguous()
    dv = torch.empty(batch + (K,), dtype=value.dtype, device=value.device) if need_value else None
    dc = torch.empty(batch + (K,), dtype=value.dtype, device=value.device) if need_conc else None
    check(_lib.load().pa_dirichlet_log_prob_grad(_dtype(value), _ptr(g), views[0], views[1], rows, K,
                                                 _ptr(dv), _ptr(dc), _stream()))
    return dv, dc


# ------------------------------------------------------------------------------------------
# one elimination step of the plated sum-product (logsumexp of a sum of broadcast terms)
# ------------------------------------------------------------------------------------------

from ctypes import c_int64 as _c_int64  # noqa: E402


def _lse_table(terms, frame):
    arr = (_lib.LseTerm * len(terms))()
    for k, t in enumerate(terms):
        e = t.expand(frame)                       # stride 0 where the term does not depend on a dim
        st = (_c_int64 * _lib.LSE_MAX_DIMS)(*([int(x) for x in e.stride()] +
                                              [0] * (_lib.LSE_MAX_DIMS - len(frame))))
        arr[k] = _lib.LseTerm(_ptr(t), st)
    sizes = (_c_int64 * len(frame))(*[int(x) for x in frame])
    return arr, sizes


def logsumexp_terms(terms, frame, rdim):
    """out = logsumexp over dim ``rdim`` of the sum of ``terms`` (tensors broadcastable to the
    shape ``frame``; nothing of the frame's size is materialised).  Returns a contiguous tensor of
    the frame's shape without ``rdim``."""
    _require_gpu(*terms)
    frame = tuple(int(x) for x in frame)
    assert 1 <= len(terms) <= _lib.LSE_MAX_TERMS and 1 <= len(frame) <= _lib.LSE_MAX_DIMS
    dtype = terms[0].dtype
    assert all(t.dtype == dtype for t in terms)
    kept = frame[:rdim] + frame[rdim + 1:]
    out = torch.empty(kept, dtype=dtype, device=terms[0].device)
    arr, sizes = _lse_table(terms, frame)
    check(_lib.load().pa_logsumexp_terms(_DTYPES[dtype], _ptr(out), len(terms), arr, len(frame), sizes,
                                         rdim, _stream()))
    return out


def logsumexp_terms_grad(terms, frame, rdim, out, g_out):
    """G[frame] = g_out[kept] * exp(sum of terms - out[kept]): every term's gradient is G summed over
    the dims the term does not have."""
    _require_gpu(out, g_out, *terms)
    frame = tuple(int(x) for x in frame)
    dtype = terms[0].dtype
    G = torch.empty(frame, dtype=dtype, device=terms[0].device)
    g_out = g_out.contiguous()
    arr, sizes = _lse_table(terms, frame)
    check(_lib.load().pa_logsumexp_terms_grad(_DTYPES[dtype], _ptr(G), _ptr(g_out), _ptr(out),
                                              len(terms), arr, len(frame), sizes, rdim, _stream()))
    return G


# The corpus does not change between ELBO-gradient steps: its inverted index (pa_lda_build_index) is
# built once per tensor object and kept beside it, under the same policy as the GLM plane image
# above (second sighting; never inside a capture; re-built into the same buffer when the tensor was
# modified in place, so a captured graph that reads the index stays valid).
LDA_INDEX_OFF, LDA_INDEX_AUTO, LDA_INDEX_ALWAYS = 0, 1, 2
_lda_index_mode = LDA_INDEX_AUTO
_lda_index_cache = {}       # (id(base), offset, shape, stride) -> [weakref, version, index, sightings, geometry, V]


def lda_set_index_mode(mode):
    """LDA_INDEX_AUTO (default): a word-id tensor seen for the second time is indexed and the
    atomic-free kernels used from then on; LDA_INDEX_ALWAYS: at first sight; LDA_INDEX_OFF: the
    LDS-atomic kernel every step (a mini-batch ``data[:, ind]`` is a fresh tensor every step and
    takes that route under every mode but ALWAYS)."""
    global _lda_index_mode
    _lda_index_mode = int(mode)
    if _lda_index_mode == LDA_INDEX_OFF:
        _lda_index_cache.clear()


def lda_build_index(words, V, out=None):
    """words int64 [Wd,B] -> the int32 index image pa_lda_factor_indexed_fwd_bwd reads (or None
    when the shape has none)."""
    _require_gpu(words)
    Wd, B = words.shape
    lib = _lib.load()
    nbytes = lib.pa_lda_index_bytes(Wd, B, V)
    if nbytes == 0:
        return None
    assert words.dtype == torch.int64 and words.is_contiguous()
    if out is None:
        out = torch.empty((nbytes // 4,), dtype=torch.int32, device=words.device)
    wbytes = lib.pa_lda_index_workspace(Wd, B, V)
    ws = torch.empty((max(wbytes, 256),), dtype=torch.uint8, device=words.device)
    check(lib.pa_lda_build_index(_ptr(words), Wd, B, V, _ptr(out), out.numel() * 4, _ptr(ws),
                                 ws.numel(), _stream()))
    return out


def _lda_index_of(words, V):
    if _lda_index_mode == LDA_INDEX_OFF:
        return None
    base = words._base if words._base is not None else words
    geom = (words.storage_offset(), tuple(words.shape), tuple(words.stride()))
    key = (id(base),) + geom + (int(V),)
    ent = _lda_index_cache.get(key)
    if ent is not None and ent[0]() is not base:
        ent = None
    if ent is None:
        import weakref
        ent = [weakref.ref(base, lambda _r, k=key: _lda_index_cache.pop(k, None)), words._version,
               None, 0, geom, int(V)]
        _lda_index_cache[key] = ent
    ent[3] += 1
    if ent[2] is None:
        if _lda_index_mode == LDA_INDEX_AUTO and ent[3] < 2:
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        ent[2] = lda_build_index(words, V)
        if ent[2] is None:
            ent[2] = False               # no index for this shape: do not ask again
        ent[1] = words._version
    elif ent[2] is not False and ent[1] != words._version:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("pyro_amd: a corpus changed in place during a graph capture")
        lda_build_index(words, V, out=ent[2])
        ent[1] = words._version
    return ent[2] if ent[2] is not False else None


def lda_index_revalidate():
    """Re-build every cached index whose tensor was modified in place (before a graph replay)."""
    for ent in list(_lda_index_cache.values()):
        base = ent[0]()
        if base is not None and isinstance(ent[2], torch.Tensor) and ent[1] != base._version:
            off, shape, stride = ent[4]
            lda_build_index(base.as_strided(shape, stride, off), ent[5], out=ent[2])
            ent[1] = base._version


MIXTURE_FAMILIES = (_lib.DIST_NORMAL, _lib.DIST_LOG_NORMAL, _lib.DIST_EXPONENTIAL, _lib.DIST_BERNOULLI_LOGITS,
                    _lib.DIST_POISSON, _lib.DIST_GAMMA)
MIXTURE_MAX_K = 64


def mixture_fwd_bwd(dist_id, x, a, p0, s0, p1, s1, p0_bs=0, p1_bs=0):
    """sum_n logsumexp_k(a[b, k] + log p(x[n] | p0[b, k], p1[b, k])) and its gradients from ONE pass over x per
    parameter set (pa_mixture_fwd_bwd): x [N]; a [K] or [B, K]; p0 / p1 flat, addressed b * bs + k * s (stride 0:
    shared); p1 None for one-parameter families.  -> float64 [1 + 3 K] (a [K]) or [B, 1 + 3 K] on the device:
    S, dS/da, dS/dp0 (per k), dS/dp1 (per k)."""
    _require_gpu(x, a, p0, p1)
    batched = a.dim() == 2
    B, K, N = (a.shape[0] if batched else 1), a.shape[-1], x.numel()
    assert x.is_contiguous() and a.is_contiguous() and p0.is_contiguous() and (p1 is None or p1.is_contiguous())
    assert a.dtype == x.dtype and p0.dtype == x.dtype and (p1 is None or p1.dtype == x.dtype)
    lib = _lib.load()
    nbytes = lib.pa_mixture_workspace(K, B)
    if nbytes == 0:
        raise Unsupported("pyro_amd: mixture_fwd_bwd needs 1 <= K <= %d (K=%d)" % (MIXTURE_MAX_K, K))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    out = torch.empty((B, 1 + 3 * K) if batched else (1 + 3 * K,), dtype=torch.float64, device=x.device)
    check(lib.pa_mixture_fwd_bwd(_dtype(x), int(dist_id), _ptr(x), N, K, B, _ptr(a), K, _ptr(p0), int(s0),
                                 int(p0_bs), _ptr(p1), int(s1), int(p1_bs), _ptr(ws), ws.numel(), _ptr(out),
                                 _stream()))
    return out


MIXTURE_MAX_D = 8


def mixture_diag_normal_fwd_bwd(x, a, loc, scale):
    """The mixture leaf for a diagonal Normal over D <= 8 features (pa_mixture_diag_normal_fwd_bwd): x [N, D];
    a [B, K]; loc, scale [B | 1, K | 1, D | 1] (contiguous).  -> (S [B], dS/da [B, K], dS/dloc [B, K, D],
    dS/dscale [B, K, D]) in float64 on the device, per (b, k, d): a shared parameter takes the sum."""
    _require_gpu(x, a, loc, scale)
    N, D = x.shape
    B, K = a.shape
    assert x.is_contiguous() and a.is_contiguous() and loc.is_contiguous() and scale.is_contiguous()
    assert loc.dim() == 3 and scale.dim() == 3 and a.dtype == x.dtype == loc.dtype == scale.dtype
    lib = _lib.load()
    kp, dd = ctypes.c_int(), ctypes.c_int()
    J = lib.pa_mixture_diag_normal_layout(K, D, ctypes.byref(kp), ctypes.byref(dd))
    nbytes = lib.pa_mixture_diag_normal_workspace(K, D, B)
    if J == 0 or nbytes == 0:
        raise Unsupported("pyro_amd: mixture_diag_normal_fwd_bwd needs K <= %d, D <= %d (K=%d, D=%d)" % (
            MIXTURE_MAX_K, MIXTURE_MAX_D, K, D))
    KP, DD = kp.value, dd.value

    def strides(p):
        Bp, Kp, Dp = p.shape
        assert Bp in (1, B) and Kp in (1, K) and Dp in (1, D), (tuple(p.shape), (B, K, D))
        return (Dp if Kp > 1 else 0), (1 if Dp > 1 else 0), (Kp * Dp if Bp > 1 else 0)

    lk, ld, lb = strides(loc)
    sk, sd, sb = strides(scale)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    out = torch.empty((B, J), dtype=torch.float64, device=x.device)
    check(lib.pa_mixture_diag_normal_fwd_bwd(_dtype(x), _ptr(x), N, D, K, B, _ptr(a), K, _ptr(loc), lk, ld, lb,
                                             _ptr(scale), sk, sd, sb, _ptr(ws), ws.numel(), _ptr(out), _stream()))
    S = out[:, 0]
    da = out[:, 1:1 + K]
    dl = out[:, 1 + KP:1 + KP + KP * DD].reshape(B, KP, DD)[:, :K, :D]
    dc = out[:, 1 + KP + KP * DD:].reshape(B, KP, DD)[:, :K, :D]
    return S, da, dl, dc


def lda_factor_fwd_bwd(words, log_theta, log_phi, index=None):
    """words int64 [Wd,B]; log_theta [B,T]; log_phi [T,V] -> (out_doc[B], g_theta[B,T], g_phi[T,V]).
    ``index``: an image from lda_build_index (default: the cached one of ``words``, if any)."""
    _require_gpu(words, log_theta, log_phi)
    Wd, B = words.shape
    T, V = log_phi.shape
    assert words.dtype == torch.int64 and words.is_contiguous()
    assert log_theta.shape == (B, T) and log_theta.is_contiguous() and log_phi.is_contiguous()
    lib = _lib.load()
    dt = _dtype(log_theta)
    if index is None:
        index = _lda_index_of(words, V)
    out = torch.empty((B,), dtype=log_theta.dtype, device=words.device)
    g_theta = torch.empty_like(log_theta)
    g_phi = torch.empty_like(log_phi)
    if index is not None:
        nbytes = lib.pa_lda_factor_indexed_workspace(dt, Wd, B, T, V)
        ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=words.device)
        check(lib.pa_lda_factor_indexed_fwd_bwd(dt, _ptr(words), _ptr(index), index.numel() * 4,
                                                _ptr(log_theta), _ptr(log_phi), Wd, B, T, V,
                                                _ptr(out), _ptr(g_theta), _ptr(g_phi), _ptr(ws),
                                                ws.numel(), _stream()))
        return out, g_theta, g_phi
    nbytes = lib.pa_lda_factor_workspace(dt, B, T, V)
    ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=words.device)
    check(lib.pa_lda_factor_fwd_bwd(dt, _ptr(words), _ptr(log_theta), _ptr(log_phi), Wd, B, T, V,
                                    _ptr(out), _ptr(g_theta), _ptr(g_phi), _ptr(ws), ws.numel(),
                                    _stream()))
    return out, g_theta, g_phi


# ------------------------------------------------------------------------------------------
# bag-of-words first layer (the amortised guide of examples/lda.py)
# ------------------------------------------------------------------------------------------
# The word histogram of a corpus as two bf16 operand images (include/pyro_amd.h "first layer of an
# amortised guide"), built once per corpus tensor and kept beside it under the policy of the GLM
# plane image / LDA index (second sighting, never inside a capture, re-built into the same buffers
# when the tensor was modified in place).
_bow_cache = {}          # (id(base), geometry, V) -> [weakref, version, (image_a, image_b), sightings, geometry, V]
BOW_MAX_COUNT = 256      # counts are stored in bf16: exact up to 256 words per document


def bow_images(words, V):
    """words int64 [Wd, B] (word ids < V) -> (image_a, image_b) bf16 tensors.  Integer work, exact;
    one-off per corpus, so it is written with torch operators: the dense histogram, then the two
    operand orders by reshape / permute."""
    _require_gpu(words)
    Wd, B = words.shape
    assert words.dtype == torch.int64 and V % 128 == 0 and Wd <= BOW_MAX_COUNT
    Bp = (B + 31) // 32 * 32
    counts = torch.zeros((V, Bp), dtype=torch.float32, device=words.device)
    counts[:, :B].scatter_add_(0, words, torch.ones(words.shape, dtype=torch.float32, device=words.device))
    img_b = counts.reshape(V // 32, 32, Bp // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().to(torch.bfloat16)
    img_a = counts.t().reshape(Bp // 32, 32, V // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().to(torch.bfloat16)
    return img_a, img_b


def bow_images_of(words, V):
    """The cached images of ``words`` or None (not yet worth building / shape not covered)."""
    if words.dtype != torch.int64 or words.dim() != 2 or V % 128 != 0 or words.shape[0] > BOW_MAX_COUNT \
            or not words.is_cuda:
        return None
    import weakref
    base = words._base if words._base is not None else words
    geom = (words.storage_offset(), tuple(words.shape), tuple(words.stride()))
    key = (id(base),) + geom + (int(V),)
    ent = _bow_cache.get(key)
    if ent is not None and ent[0]() is not base:
        ent = None
    if ent is None:
        ent = [weakref.ref(base, lambda _r, k=key: _bow_cache.pop(k, None)), words._version, None, 0, geom,
               int(V)]
        _bow_cache[key] = ent
    ent[3] += 1
    capturing = torch.cuda.is_current_stream_capturing()
    if ent[2] is None:
        if ent[3] < 2 or capturing:
            return None
        ent[2] = bow_images(words, V)
        ent[1] = words._version
    elif ent[1] != words._version:
        if capturing:
            raise RuntimeError("pyro_amd: a corpus changed in place during a graph capture")
        a, b = bow_images(words, V)
        ent[2][0].copy_(a)
        ent[2][1].copy_(b)
        ent[1] = words._version
    return ent[2]


def bow_revalidate():
    """Re-build the cached images of corpora that were modified in place (before a graph replay)."""
    for ent in list(_bow_cache.values()):
        base = ent[0]()
        if base is not None and ent[2] is not None and ent[1] != base._version:
            off, shape, stride = ent[4]
            a, b = bow_images(base.as_strided(shape, stride, off), ent[5])
            ent[2][0].copy_(a)
            ent[2][1].copy_(b)
            ent[1] = base._version


def bow_linear_fwd(image_a, W, bias, B, sigmoid=False):
    """out[B, H] = bias + counts @ W.T, through a sigmoid when asked (pa_bow_linear_fwd_act); W [H, V] f32."""
    _require_gpu(image_a, W, bias)
    H, V = W.shape
    assert W.is_contiguous() and W.dtype == torch.float32 and (bias is None or bias.is_contiguous())
    lib = _lib.load()
    nbytes = lib.pa_bow_workspace(B, V, H)
    if nbytes == 0:
        raise Unsupported("pyro_amd: bag-of-words layer does not cover B=%d V=%d H=%d" % (B, V, H))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=W.device)
    out = torch.empty((B, H), dtype=torch.float32, device=W.device)
    check(lib.pa_bow_linear_fwd_act(_ptr(image_a), _ptr(W), _ptr(bias), B, V, H, int(bool(sigmoid)), _ptr(out),
                                    _ptr(ws), nbytes, _stream()))
    return out


def bow_linear_bwd(image_b, d_out, V, y_mul=None, want_bias=False):
    """dW[H, V] = d.T @ counts with d = d_out [B, H] f32, or d = d_out * (1 - y_mul) * y_mul (the gradient
    through the layer's sigmoid output ``y_mul``) -- pa_bow_linear_bwd_act.  ``want_bias``: returns
    (dW, partial sums [4, ceil(B / 32) * 2, 32] of d over blocks of 16 documents: .sum(1) is the bias gradient
    of hidden units 32 t + c)."""
    _require_gpu(image_b, d_out, y_mul)
    B, H = d_out.shape
    d_out = d_out.contiguous()
    assert y_mul is None or (y_mul.shape == d_out.shape and y_mul.is_contiguous() and y_mul.dtype == torch.float32)
    lib = _lib.load()
    nbytes = lib.pa_bow_workspace(B, V, H)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=d_out.device)
    dW = torch.empty((H, V), dtype=torch.float32, device=d_out.device)
    nkt = (B + 31) // 32 * 2
    part = torch.empty((4, nkt, 32), dtype=torch.float32, device=d_out.device) if want_bias else None
    check(lib.pa_bow_linear_bwd_act(_ptr(image_b), _ptr(d_out), _ptr(y_mul), B, V, H, _ptr(dW), _ptr(part),
                                    _ptr(ws), nbytes, _stream()))
    return (dW, part) if want_bias else dW


def tall_linear(g, W, w_row_stride, w_col_stride, C, bias=None, y_mul=None, sigmoid=False):
    """g[B, R] @ Wm[R, C] (+ bias): pa_tall_linear_act; Wm[r][c] = W.flat[r * w_row_stride + c * w_col_stride].
    ``y_mul`` [B, R]: the first operand is g * (1 - y_mul) * y_mul; ``sigmoid``: the result goes through one."""
    _require_gpu(g, W, bias, y_mul)
    B, R = g.shape
    assert g.dtype == torch.float32 == W.dtype and g.is_contiguous() and W.is_contiguous()
    assert y_mul is None or (y_mul.shape == g.shape and y_mul.is_contiguous() and y_mul.dtype == torch.float32)
    if R > 128 or C > 128:
        raise Unsupported("pyro_amd: tall_linear covers at most 128 features (R=%d C=%d)" % (R, C))
    out = torch.empty((B, C), dtype=torch.float32, device=g.device)
    check(_lib.load().pa_tall_linear_act(_ptr(g), B, R, _ptr(W), int(w_row_stride), int(w_col_stride), int(C),
                                         _ptr(bias), _ptr(y_mul), int(bool(sigmoid)), _ptr(out), _stream()))
    return out


def tall_wgrad(g, x, want_bias=True, y_mul=None):
    """(d[B, R].T @ x[B, K], d.sum(0)) in one pass over the batch with d = g, or g * (1 - y_mul) * y_mul:
    pa_tall_wgrad_act."""
    _require_gpu(g, x, y_mul)
    B, R = g.shape
    K = x.shape[1]
    assert x.shape[0] == B and g.dtype == torch.float32 == x.dtype and g.is_contiguous() and x.is_contiguous()
    assert y_mul is None or (y_mul.shape == g.shape and y_mul.is_contiguous() and y_mul.dtype == torch.float32)
    lib = _lib.load()
    nbytes = lib.pa_tall_wgrad_workspace(B, R, K)
    if nbytes == 0:
        raise Unsupported("pyro_amd: tall_wgrad covers at most 128 features (R=%d K=%d)" % (R, K))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=g.device)
    dW = torch.empty((R, K), dtype=torch.float32, device=g.device)
    db = torch.empty((R,), dtype=torch.float32, device=g.device) if want_bias else None
    check(lib.pa_tall_wgrad_act(_ptr(g), _ptr(x), _ptr(y_mul), B, R, K, _ptr(dW), _ptr(db), _ptr(ws), nbytes,
                                _stream()))
    return dW, db


def tsgemm_tn(a, x):
    """a[B, M].T @ x[B, N] -> [M, N] for tall f32 operands (M, N <= 128): pa_tsgemm_tn."""
    _require_gpu(a, x)
    B, M = a.shape
    N = x.shape[1]
    assert x.shape[0] == B and a.dtype == torch.float32 == x.dtype
    a, x = a.contiguous(), x.contiguous()
    lib = _lib.load()
    nbytes = lib.pa_tsgemm_tn_workspace(B, M, N)
    if nbytes == 0:
        raise Unsupported("pyro_amd: tsgemm_tn does not cover B=%d M=%d N=%d" % (B, M, N))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=a.device)
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    check(lib.pa_tsgemm_tn(_ptr(a), _ptr(x), B, M, N, _ptr(out), _ptr(ws), nbytes, _stream()))
    return out


# ------------------------------------------------------------------------------------------
# flat Adam
# ------------------------------------------------------------------------------------------

def adam_step(param, grad, exp_avg, exp_avg_sq, step_dev, lr, betas=(0.9, 0.999), eps=1e-8,
              weight_decay=0.0, clip_norm=0.0, lrd=1.0, clipped=False, zero_grad=True,
              publish=None):
    """``publish`` = (device scalar, pinned float64[1], pinned int64[1], counter or None, inc):
    the launch's last workgroup also does what publish_scalar does (pa_adam_step_publish)."""
    _require_gpu(param, grad, exp_avg, exp_avg_sq, step_dev)
    assert step_dev.dtype == torch.int64 and step_dev.numel() == 2   # [step, ticket]
    for x in (param, grad, exp_avg, exp_avg_sq):
        assert x.is_contiguous() and x.dtype == param.dtype and x.numel() == param.numel()
    if publish is not None:
        src, host_value, host_seq, counter, inc = publish
        _require_gpu(src)
        assert src.numel() == 1 and host_value.dtype == torch.float64 and host_value.is_pinned()
        assert host_seq.dtype == torch.int64 and host_seq.is_pinned()
        assert counter is None or (counter.dtype == torch.int64 and counter.numel() >= 2)
        check(_lib.load().pa_adam_step_publish(
            _dtype(param), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(),
            float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
            float(clip_norm), float(lrd), int(bool(clipped)), _ptr(step_dev), int(bool(zero_grad)),
            _dtype(src), _ptr(src), _ptr(host_value), _ptr(host_seq),
            _ptr(counter) if counter is not None else None, int(inc), _stream()))
        return
    check(_lib.load().pa_adam_step(_dtype(param), _ptr(param), _ptr(grad), _ptr(exp_avg),
                                   _ptr(exp_avg_sq), param.numel(), float(lr), float(betas[0]),
                                   float(betas[1]), float(eps), float(weight_decay),
                                   float(clip_norm), float(lrd), int(bool(clipped)),
                                   _ptr(step_dev), int(bool(zero_grad)), _stream()))
