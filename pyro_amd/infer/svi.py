"""SVI: the drop-in training-step API (reference: pyro/infer/svi.py:38-162).

``SVI(..., hip_graph=True)`` is the MI355X-native fast path: after a few eager steps (which create
the parameters, the flat optimizer state and run all validation) the WHOLE step -- guide sampling
from the Philox stream, model replay, fused ELBO-gradient kernels, autograd backward, the flat
Adam update and the gradient zeroing -- is captured once into a hipGraph and every later
``step()`` with the same argument tensors is a single graph launch plus the one host read of
the loss.  No Python handler code and no per-kernel launch latency remain in the step; the random
stream advances on the device, so a graph-replayed run draws exactly the numbers the eager run
would (tests/test_svi_gpu.py checks the trajectories are identical).
"""
import os as _os
import warnings

import torch
import torch.utils._python_dispatch

from .. import kernels, poutine
from ..params import _PARAM_STORE
from ..util import capture_scope, torch_isnan, zero_grads
from .elbo import ELBO


def _arg_key(x):
    if isinstance(x, torch.Tensor):
        # identity and geometry only -- NOT the version counter: a replay reads the live memory, so
        # a tensor that is updated in place between steps (the way to feed new data to a captured
        # step) keeps its capture; images derived from it (GLM planes, LDA index) are re-packed by
        # the revalidate hooks before the replay
        return ("t", x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, x.device)
    if isinstance(x, (list, tuple)):
        return tuple(_arg_key(v) for v in x)
    try:
        hash(x)
        return ("p", x)
    except TypeError:
        return ("id", id(x))


def _fast_geometry(args):
    """Per positional argument: (data_ptr, shape, stride) of a tensor, None otherwise -- what the fast path of
    SVI.step compares (with the arguments' identity) instead of building the signature key again."""
    return tuple((a.data_ptr(), a.shape, a.stride()) if isinstance(a, torch.Tensor) else None for a in args)


class _ReadSet(torch.utils._python_dispatch.TorchDispatchMode):
    """Every tensor a step READS (or writes) that was not made inside the step: the arguments of the
    operators torch dispatches and of the package's own launches (kernels._ptr) whose storage was not
    allocated while this scope was open.  A replay enqueued ahead of the host (prearm) is only let run when
    the version counters of all of them are what they were when it was enqueued -- not just those of
    step()'s arguments and of the parameters: a model that closes over a tensor the user updates between
    steps is noticed the same way."""

    def __init__(self):
        super().__init__()
        self.internal = set()
        self.external = {}

    def note(self, t):
        if t.is_cuda and t.untyped_storage().data_ptr() not in self.internal:
            self.external[id(t)] = t

    def __enter__(self):
        kernels._PTR_HOOKS.append(self.note)
        return super().__enter__()

    def __exit__(self, *exc):
        kernels._PTR_HOOKS.remove(self.note)
        return super().__exit__(*exc)

    def _walk(self, xs, fn):
        for x in xs:
            if isinstance(x, torch.Tensor):
                fn(x)
            elif isinstance(x, (list, tuple)):
                self._walk(x, fn)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        self._walk(args, self.note)
        self._walk(tuple(kwargs.values()), self.note)
        out = func(*args, **kwargs)
        self._walk(out if isinstance(out, (list, tuple)) else (out,),
                   lambda t: self.internal.add(t.untyped_storage().data_ptr()) if t.is_cuda else None)
        return out


class _CapturedStep:
    eager_params = None    # parameters of a step whose optimizer update runs eagerly behind the replay
    alive = True           # False once the entry left SVI._graphs (evicted, stale, released)
    rtc_blocks = None
    reads = ()             # tensors the step reads that it did not make (see _ReadSet)
    direct = None          # kernels.DirectReplay when the graph is a short chain of kernels (launched as such)
    gate = None            # kernels.StepGate when the step's first node is a gate (SVI(prearm=True))
    armed = False          # the NEXT replay is already enqueued behind its gate
    armed_state = None     # what the host looked like when it was enqueued
    armed_nargs = 0
    arm_backoff = 0        # steps to run un-armed after the gate gave an armed replay up
    arm_penalty = 1

    def __init__(self, graph, cap, loss, graph2=None, between=None, mailbox=None):
        self.graph, self.cap, self.loss = graph, cap, loss
        self.graph2, self.between = graph2, between     # split capture around a collective
        # pinned host memory the graph's last node writes the loss into (value, sequence number):
        # polled by the host instead of a stream synchronisation + device-to-host copy
        self.mailbox = mailbox
        if mailbox is not None:
            self._value_np, self._seq_np = mailbox[0].numpy(), mailbox[1].numpy()
            self._seq = int(self._seq_np[0])

    def replay(self):
        """Enqueue the captured step: one hipGraphLaunch, or -- opt-in, kernels.DIRECT_REPLAY -- its two or three
        kernels one by one (measured slower for a step that waits for its loss, closer together when queued ahead)."""
        d = self.direct
        if d is not None:
            d.launch()
        else:
            self.graph.replay()

    # ---- the step gate (kernels.StepGate): replays enqueued ahead of the host ------------------
    def launch(self):
        """Run the step now: release the replay waiting in its gate, or enqueue one that passes."""
        g = self.gate
        if g is None:
            self.replay()
            return False
        g.go_np[0] = g.next
        if self.armed:
            self.armed = False
            return True                 # (it may have given itself up meanwhile: read_loss checks)
        self.replay()
        return False

    def arm(self, state):
        """Enqueue the NEXT step behind its gate (call right after launch(): the host's launch
        latency then overlaps the device's execution of the current step)."""
        self.replay()
        self.armed, self.armed_state = True, state

    def cancel(self):
        """The armed replay must not run (the host changed something it would read)."""
        g = self.gate
        n = g.next
        g.go_np[0] = -n
        spins = 0
        while int(g.ack_np[0]) != n:
            spins += 1
            if spins > 2_000_000:
                torch.cuda.current_stream().synchronize()
                break
        self.armed = False

    def read_loss(self, released_armed=False):
        if self.mailbox is None:
            return self.loss.item()
        # one publish per replay, each with a new sequence number (a device-side count: it need not
        # be the previous number of THIS mailbox plus one)
        last = self._seq
        seq, spins = self._seq_np, 0
        g = self.gate
        ack = g.ack_np if (g is not None and released_armed) else None
        gave_up = False
        while int(seq[0]) == last:
            spins += 1
            if ack is not None and int(ack[0]) == g.next:
                # the gate had given the armed replay up before the release arrived (the host was
                # away longer than the gate's patience): nothing ran.  A replay already enqueued
                # for the NEXT step numbers itself by what has run, finds its number released and
                # runs as this step; otherwise the step is launched the ordinary way
                ack = None
                gave_up = True
                if self.armed:
                    self.armed = False
                else:
                    self.replay()
                self.arm_backoff = self.arm_penalty
                self.arm_penalty = min(self.arm_penalty * 2, 1024)
            if spins > 5_000_000:           # ~seconds: something is wrong, fall back to a real sync
                torch.cuda.current_stream().synchronize()
                if int(seq[0]) == last:
                    raise RuntimeError("pyro_amd: the captured step did not publish its loss")
        self._seq = int(seq[0])
        if released_armed and not gave_up and self.arm_penalty > 1:
            self.arm_penalty >>= 1          # an armed replay that ran: the back-off decays again
        if g is not None:
            g.next += 1
        return float(self._value_np[0])


# ``SVI(model, guide, optim, loss)`` -- the reference's constructor, nothing else said -- captures its step
# into a hipGraph by itself ("auto") when the step can be one: the ELBO assembles its loss on the device,
# every tensor argument of step() lives on the GPU and there is at least one (a model that closes over
# its data gives no argument whose identity says "same step again").  False: eager unless asked;
# True: as if every SVI were built with hip_graph=True.  (pyro.settings alias: svi_capture_steps)
CAPTURE_STEPS = "auto"


def _FlatOptimOK(optim):
    """An optimizer whose update a captured step may hold: the package's own (their step counters live on
    the device); a wrapped torch optimizer keeps host-side state per call."""
    return bool(getattr(optim, "zeroes_grads", False) or getattr(optim, "capturable", False))


def _capturable_arguments(args, kwargs):
    """Every tensor argument on the GPU, at least one of them, and nothing else but hashable constants."""
    seen = False
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor):
            if not a.is_cuda:
                return False
            seen = True
        elif isinstance(a, (list, tuple)):
            for t in a:
                if isinstance(t, torch.Tensor):
                    if not t.is_cuda:
                        return False
                    seen = True
        elif not isinstance(a, (int, float, bool, str, type(None))):
            return False
    return seen


class CapturedStepWarning(UserWarning):
    """Emitted once per process when an SVI built WITHOUT hip_graph=... captures its step by itself."""


_CAPTURE_NOTE = (
    "pyro_amd: SVI captured its step into a hipGraph after {n} eager steps (device tensors as arguments and a "
    "device-side ELBO; with an optimizer other than this package's flat ones the update runs eagerly behind "
    "the replay: nobody asked, so this is said once).  A captured step does not "
    "re-run the Python of the model and guide: what they read from the host at capture time is frozen into "
    "the graph.  Guarded -- the capture is dropped and re-made when it changes: step() arguments, Python "
    "scalars / flags the model or guide reach through closures, globals, functools.partial or attributes "
    "(an annealing factor `self.beta`, nn.Module.training), the seed, the set of parameters "
    "(pyro.clear_param_store()).  NOT guarded: Python control flow on device values, host-side randomness "
    "(random / numpy / CPU torch generators), state mutated through containers (lists, dicts, numpy "
    "arrays).  Models that need those: SVI(..., hip_graph=False), PYRO_AMD_HIP_GRAPH=0 or "
    "pyro.settings.set(svi_capture_steps=False).")
_WARNED_CAPTURE = [False]
_SCALAR_TYPES = (bool, int, float, str, type(None))
_MISSING = object()


def _host_scalar_watch(*fns, depth=3):
    """Every Python scalar (bool / int / float / str / None) a callable can read from outside its arguments
    without the version counter of any tensor moving: closure cells, the globals its code names, the
    arguments of a functools.partial, the attributes of the object a bound method / callable object belongs
    to and -- for torch.nn.Module trees -- of every sub-module (`training` among them).  Returns
    ([(mapping, key, value)], [(cell, value)]): live references, compared by SVI.step before every replay
    (the reference re-runs the model each step, pyro/infer/svi.py:134-162, so it sees such changes)."""
    import functools
    import types
    maps, cells, seen = [], [], set()

    def scan_mapping(d, keys=None):
        for k in (list(d) if keys is None else keys):
            v = d.get(k, _MISSING)
            if isinstance(v, _SCALAR_TYPES) and isinstance(k, str) and not k.startswith("__"):
                maps.append((d, k, v))

    def scan_object(o, level):
        if id(o) in seen:
            return
        seen.add(id(o))
        if isinstance(o, torch.nn.Module):
            for m in o.modules():
                if id(m) not in seen or m is o:
                    seen.add(id(m))
                    scan_mapping(vars(m))
        elif hasattr(o, "__dict__") and not isinstance(o, (type, types.ModuleType)):
            scan_mapping(vars(o))
        if level > 0 and hasattr(o, "__dict__") and not isinstance(o, (type, types.ModuleType)):
            for v in list(vars(o).values()):
                if callable(v) and not isinstance(v, (type, torch.Tensor)):
                    scan(v, level - 1)

    def scan(fn, level):
        if fn is None or id(fn) in seen or level < 0:
            return
        if isinstance(fn, functools.partial):
            seen.add(id(fn))
            scan(fn.func, level)
            for a in list(fn.args) + list((fn.keywords or {}).values()):
                if callable(a) or hasattr(a, "__dict__"):
                    scan(a, level - 1)
            return
        if isinstance(fn, types.MethodType):
            scan_object(fn.__self__, level - 1)
            fn = fn.__func__
        if isinstance(fn, types.FunctionType):
            seen.add(id(fn))
            code = fn.__code__
            scan_mapping(fn.__globals__, [n for n in code.co_names if n in fn.__globals__])
            for c in fn.__closure__ or ():
                try:
                    v = c.cell_contents
                except ValueError:
                    continue
                if isinstance(v, _SCALAR_TYPES):
                    cells.append((c, v))
                elif isinstance(v, torch.Tensor):
                    continue
                elif callable(v) or hasattr(v, "__dict__"):
                    scan(v, level - 1)
            return
        if isinstance(fn, (torch.Tensor, type, types.ModuleType, types.BuiltinFunctionType)):
            return
        scan_object(fn, level - 1)              # a callable object (autoguide, nn.Module, class instance)

    for f in fns:
        scan(f, depth)
    return maps, cells


def _host_scalars_changed(watch):
    maps, cells = watch
    for d, k, v in maps:
        w = d.get(k, _MISSING)
        if w is not v and w != v:
            return k
    for c, v in cells:
        w = c.cell_contents
        if w is not v and w != v:
            return "<closure>"
    return None


class SVI:
    def __init__(self, model, guide, optim, loss, loss_and_grads=None, num_samples=0, num_steps=0,
                 hip_graph=None, graph_warmup=3, prearm=None, speculate=True, **kwargs):
        if num_steps or num_samples:
            warnings.warn("num_steps / num_samples are ignored (TracePosterior is not part of "
                          "this backend)")
        self.model, self.guide, self.optim = model, guide, optim
        self._loss_device = None
        if isinstance(loss, ELBO):
            self.loss = loss.loss
            self.loss_and_grads = loss.loss_and_grads
            self._loss_device = getattr(loss, "loss_and_grads_device", None)
        else:
            if loss_and_grads is None:
                def _loss_and_grads(model, guide, *args, **kwargs):
                    loss_val = loss(model, guide, *args, **kwargs)
                    if getattr(loss_val, "requires_grad", False):
                        loss_val.backward(retain_graph=True)
                    return loss_val

                loss_and_grads = _loss_and_grads
            self.loss, self.loss_and_grads = loss, loss_and_grads
        # hip_graph=None (the default): decided at the first step() from its arguments, see CAPTURE_STEPS;
        # a capture that fails leaves such an SVI eager without a warning (nobody asked for a graph)
        self._auto_graph = hip_graph is None
        # an optimizer whose update cannot sit in a graph (a wrapped torch optimizer keeps host-side state per
        # call; a plain callable): the step's LOSS AND GRADIENTS are captured, the update and the gradient
        # zeroing run eagerly behind every replay -- the reference's step order (pyro/infer/svi.py:144-156)
        # with its first half as one graph launch
        self._eager_update = not _FlatOptimOK(optim)
        if hip_graph is None:
            want = CAPTURE_STEPS if _os.environ.get("PYRO_AMD_HIP_GRAPH", "1") != "0" else False
            hip_graph = (want is True or want == "auto") and self._loss_device is not None \
                and callable(optim)
        self.hip_graph = bool(hip_graph)
        # prearm (OPT-IN, default off): right after launching step k the replay of step k+1 is enqueued
        # behind a gate node and released by the next step() call with one store to pinned memory (the
        # launch latency of a step overlaps the execution of the one before).  THE CALLER'S PROMISE: between
        # two step() calls nothing is enqueued on the step's stream that READS what a step writes
        # (parameters, optimizer state) -- a `pyro.param("w").detach().clone()`, an EMA update or a
        # device-side metric enqueued after step k returns would sit BEHIND step k+1's replay on the stream
        # and see the parameters after step k+1.  (Call SVI.pause() before such work: it gives the waiting
        # replay up at once.)  What the package can check itself it does: the replay is only released when
        # the version counter of every tensor the captured step reads that it did not make itself
        # (arguments, parameters, optimizer state, tensors the model closes over: _ReadSet), the position
        # and seed of the random stream and the set of parameters are what they were when it was enqueued;
        # otherwise it is given up and the step runs the ordinary way.  A read-only use of a parameter
        # moves no version counter -- hence the promise, and hence opt-in.  Only steps whose every node can
        # be given up are armed; the gate gives a replay up by itself after 40 us without a release.
        if prearm is None:
            prearm = False
        self.prearm = bool(prearm) and _os.environ.get("PYRO_AMD_PREARM", "1") != "0"
        # speculate (with prearm): the gate sits in front of the step's chained tail instead of first, so
        # the forward pass of the replay enqueued ahead (the plane-image GLM kernel, which also makes the
        # guide draw) runs while the host is still reading the previous loss and calling step() again.
        # It reads what that step left behind and writes scratch only; everything that changes persistent
        # state waits behind the gate and is given up with it.  Same promise by the caller as prearm.
        # Taken only when nothing but that kernel precedes the tail; otherwise the gate goes first.
        self.speculate = bool(speculate) and _os.environ.get("PYRO_AMD_SPECULATE", "1") != "0"
        self._armed_fast = None     # (entry, argument objects, their key) of the armed replay
        self._last_fast = None      # (argument objects, their geometry, entry) of the last un-gated replay
        if self.hip_graph and self._loss_device is None:
            raise ValueError("hip_graph=True needs an ELBO that provides loss_and_grads_device")
        self.graph_warmup = int(graph_warmup)
        # captured steps by argument signature, least recently used first; bounded: fresh
        # mini-batch tensors or a python scalar that changes every step would otherwise grow the
        # table (and the graph memory pool) without ever reaching a replay
        self._graphs = {}
        self._eager_seen = {}
        self.max_graphs = int(kwargs.pop("max_graphs", 8))
        self._warned_keys = False
        self._const_rec = {}       # signature -> ConstantRecorder of its last eager step
        # host state a captured step froze (see _CAPTURE_NOTE): Python scalars the model / guide can reach,
        # the parameter store's generation
        self._watch = None
        self._watch_first = None   # the same scalars after the FIRST eager step of a signature
        self._store_generation = None
        self._self_mutating = False

    def evaluate_loss(self, *args, **kwargs):
        with torch.no_grad():
            loss = self.loss(self.model, self.guide, *args, **kwargs)
            return loss.item() if isinstance(loss, torch.Tensor) else loss

    def _params_of(self, param_capture):
        return set(site["value"].unconstrained() if hasattr(site["value"], "unconstrained")
                   else getattr(site["value"], "_pyro_unconstrained_param", site["value"])
                   for site in param_capture.trace.nodes.values())

    def _eager_step(self, *args, **kwargs):
        with poutine.trace(param_only=True) as param_capture:
            loss = self.loss_and_grads(self.model, self.guide, *args, **kwargs)
        params = self._params_of(param_capture)
        self.optim(params)
        if not getattr(self.optim, "zeroes_grads", False):
            grads = [p.grad for p in params if p.grad is not None]
            if grads:
                torch._foreach_zero_(grads)           # one launch for all of them, in place (zero_grads' contract)
        if isinstance(loss, tuple):
            return type(loss)(map(lambda x: x.item() if isinstance(x, torch.Tensor) else x, loss))
        return loss.item() if isinstance(loss, torch.Tensor) else loss

    def step(self, *args, **kwargs):
        """One gradient step: loss_and_grads, optimizer update on every touched param, zero grads."""
        if not self.hip_graph:
            return self._eager_step(*args, **kwargs)
        if self._auto_graph and not _capturable_arguments(args, kwargs):
            return self._eager_step(*args, **kwargs)
        if self._graphs:
            changed = _host_scalars_changed(self._watch) if self._watch is not None else None
            if changed is not None or _PARAM_STORE.generation != self._store_generation:
                # something the captured steps froze has changed on the host (an annealing factor, a
                # train()/eval() flag, pyro.clear_param_store()): drop them; this step and the next
                # graph_warmup - 1 run eagerly (as the reference's every step does), then it is captured anew
                self.release()
                self._eager_seen.clear()
        last = self._last_fast
        if last is not None and not kwargs and len(args) == len(last[0]):
            # the same argument OBJECTS as the last replay, still where and what they were: straight to the
            # replay (the signature key -- nested tuples of every tensor's geometry -- is what the rest of
            # this function's ~6 us of Python is spent on)
            entry = last[2]
            same = entry.alive
            if same:
                for a, b, g in zip(args, last[0], last[1]):
                    if a is not b or (g is not None and (a.data_ptr() != g[0] or a.shape != g[1]
                                                         or a.stride() != g[2])):
                        same = False
                        break
            if same and not entry.cap.stale():
                kernels.glm_planes_revalidate()     # data the graph reads through a cached image
                kernels.lda_index_revalidate()
                kernels.bow_revalidate()
                entry.cap.before_replay()
                entry.replay()
                if entry.graph2 is not None:
                    entry.between()
                    entry.graph2.replay()
                entry.cap.after_replay()
                if entry.eager_params is not None:
                    self._update_eagerly(entry.eager_params)
                return entry.read_loss()
            self._last_fast = None
        fast = self._armed_fast
        if fast is not None:
            # the step after an armed one, called with the very same argument objects: release the
            # waiting replay before anything else (the device idles while the host is in here)
            entry, fargs, fkey = fast
            if (not kwargs and len(args) == len(fargs) and all(a is b for a, b in zip(args, fargs))
                    and entry.armed and _arg_key(args) == fkey
                    and self._host_state_unchanged(*entry.armed_state)
                    and not kernels.revalidate_pending()):
                return self._gated_step(entry, args, kwargs, checked=True)
        key = (_arg_key(args), _arg_key(tuple(sorted(kwargs.items()))))
        entry = self._graphs.get(key)
        if entry is not None and entry.cap.stale():
            # torch.manual_seed since the capture: the recorded draws hold the old seed as a launch constant
            if entry.armed:
                entry.cancel()
            self._armed_fast = None
            self._drop(key)
            entry = None
        if entry is None:
            n = self._eager_seen.get(key, 0)
            if n < self.graph_warmup:
                if n == 0 and len(self._eager_seen) >= 8 * self.max_graphs:
                    # signatures that never came back: forget the oldest half
                    for k in list(self._eager_seen)[:len(self._eager_seen) // 2]:
                        del self._eager_seen[k]
                        self._const_rec.pop(k, None)
                    if not self._warned_keys and not self._auto_graph:
                        self._warned_keys = True
                        warnings.warn("pyro_amd: SVI(hip_graph=True) keeps seeing new argument "
                                      "signatures (fresh tensors or changing scalars every step); "
                                      "such steps run eagerly. Pass the same tensors (update them "
                                      "in place) to reach the captured step.")
                self._eager_seen[key] = n + 1
                first = n == 0 and not self._graphs
                if n == self.graph_warmup - 1 and _os.environ.get("PYRO_AMD_HOIST", "1") != "0":
                    # the last eager step before the capture: note the constant tensors the model
                    # and guide create, so that the captured step need not fill them again
                    from .constants import ConstantRecorder
                    rec = ConstantRecorder()
                    from ..ops import fuser
                    # (under the fuser too: the kernels the captured step will launch are generated and
                    #  compiled here, before the capture starts)
                    with fuser.scope(), rec:
                        out = self._eager_step(*args, **kwargs)
                    self._const_rec[key] = rec
                else:
                    out = self._eager_step(*args, **kwargs)
                if first and self.graph_warmup > 1:
                    # (what the model and guide can read from the host, after their first run: compared at
                    #  capture time -- a scalar that moves while they run cannot be frozen)
                    self._watch_first = _host_scalar_watch(self.model, self.guide)
                return out
            if self._watch_first is not None and not self._graphs:
                moved = _host_scalars_changed(self._watch_first)
                self._watch_first = None
                if moved is not None:
                    # the model / guide change a host scalar of their own on every call (a step counter,
                    # an annealing schedule advanced inside the model): a replay would not
                    self._self_mutating = True
                    if self._auto_graph:
                        self.hip_graph = False
                        return self._eager_step(*args, **kwargs)
                    warnings.warn("pyro_amd: SVI(hip_graph=True): the model or guide changed the host-side "
                                  "value %r while it ran; the captured step freezes it" % (moved,))
            entry = self._capture(key, args, kwargs)
            if entry is None:                      # capture failed: stay eager
                return self._eager_step(*args, **kwargs)
            self._watch = _host_scalar_watch(self.model, self.guide)
            self._store_generation = _PARAM_STORE.generation
            if self._auto_graph and not _WARNED_CAPTURE[0]:
                _WARNED_CAPTURE[0] = True
                warnings.warn(_CAPTURE_NOTE.format(n=self.graph_warmup), CapturedStepWarning, stacklevel=2)
            self._eager_seen.pop(key, None)
            while len(self._graphs) > self.max_graphs:     # evict the least recently used capture
                self._drop(next(iter(self._graphs)))
        else:
            self._graphs[key] = self._graphs.pop(key)      # most recently used last
        if entry.gate is None:
            kernels.glm_planes_revalidate()     # data the graph reads through a cached image
            kernels.lda_index_revalidate()
            kernels.bow_revalidate()
            entry.cap.before_replay()
            entry.replay()
            if entry.graph2 is not None:
                entry.between()            # eager RCCL all-reduce of the flat gradient
                entry.graph2.replay()
            entry.cap.after_replay()
            if entry.eager_params is not None:
                self._update_eagerly(entry.eager_params)
            if not kwargs:
                self._last_fast = (args, _fast_geometry(args), entry)
            return entry.read_loss()
        return self._gated_step(entry, args, kwargs)

    def _update_eagerly(self, params):
        """The second half of the reference's step (pyro/infer/svi.py:153-156) behind a captured first half:
        the optimizer's own launches and the gradient zeroing, enqueued while the host has not read the loss
        yet (the replay's last node published it; these kernels queue behind it)."""
        self.optim(params)
        if not getattr(self.optim, "zeroes_grads", False):
            grads = [p.grad for p in params if p.grad is not None]
            if grads:
                torch._foreach_zero_(grads)           # one launch for all of them, in place (zero_grads' contract)

    def _drop(self, key):
        e = self._graphs.pop(key, None)
        if e is not None:
            e.alive = False
        return e

    def release(self):
        """Drop every captured step (hipGraph executables, their private memory pools, pinned mailboxes
        and gate words) now instead of when the cyclic collector gets to this object.  A process that
        builds hundreds of SVI objects with captured steps (a test suite, a hyper-parameter sweep) should
        call this, or ``gc.collect()``, between them: the GPU suite of this repository aborted inside
        the HIP runtime after ~300 un-collected captures (tests/conftest.py collects per module)."""
        for e in self._graphs.values():
            if e.armed:
                e.cancel()
            e.alive = False
        self._armed_fast = self._last_fast = None
        self._graphs.clear()
        self._const_rec.clear()

    def pause(self):
        """The caller is about to wait for the device (``synchronize()``, a device-to-host read) or to
        enqueue work of its own: a replay waiting behind its gate is given up NOW instead of after the
        gate's patience.  Pre-arming stays on: the next ``step()`` launches its replay itself and arms the
        one after it."""
        self._armed_fast = None
        for e in self._graphs.values():
            if e.armed:
                e.cancel()

    def disarm(self):
        """Stop enqueuing replays ahead of time (prearm=True): cancels a waiting one; later steps of
        the existing captures run as ordinary replays through their (already released) gates."""
        self.prearm = False
        self._armed_fast = None
        for e in self._graphs.values():
            if e.armed:
                e.cancel()
            e.arm_backoff = 1 << 60

    def _host_state(self, args, kwargs, entry=None):
        """What an armed replay depends on besides the device's own state: (tensor, version) of every
        argument tensor, every parameter and every other tensor the captured step reads from outside
        itself, and the host-side position of the Philox stream."""
        from .. import rng
        refs = [(t, t._version) for t in entry.reads] if entry is not None else []
        for a in list(args) + [v for _, v in sorted(kwargs.items())]:
            if isinstance(a, torch.Tensor):
                refs.append((a, a._version))
            elif isinstance(a, (list, tuple)):
                refs.extend((t, t._version) for t in a if isinstance(t, torch.Tensor))
        params = _PARAM_STORE._params
        refs.extend((p, p._version) for p in params.values())
        return refs, len(params), rng._STATE

    @staticmethod
    def _host_state_unchanged(state, offset):
        from .. import rng
        rng._follow_torch()                # (a re-seeded default generator restarts the stream: offset 0)
        refs, nparams, rng_state = state
        if rng_state["offset"] != offset or len(_PARAM_STORE._params) != nparams:
            return False
        for t, v in refs:
            if t._version != v:
                return False
        return True

    def _gated_step(self, entry, args, kwargs, checked=False):
        """The fast path of a captured step whose first node is a gate (prearm=True)."""
        if checked:
            entry.launch()
            entry.cap.after_replay()
            from .. import rng
            entry.arm((self._host_state(args, kwargs, entry), rng._STATE["offset"]))
            return entry.read_loss(released_armed=True)
        if entry.armed:
            # the armed replay reads the device as the stream will have left it BEFORE anything the
            # host enqueued since: only sound if the host enqueued nothing it depends on
            if not (self._host_state_unchanged(*entry.armed_state)
                    and len(args) == entry.armed_nargs) or kernels.revalidate_pending():
                entry.cancel()
        for other in self._graphs.values():          # (a replay armed for another signature)
            if other is not entry and other.armed:
                other.cancel()
        if not entry.armed:
            kernels.glm_planes_revalidate()
            kernels.lda_index_revalidate()
            kernels.bow_revalidate()
            entry.cap.before_replay()
        released = entry.launch()
        entry.cap.after_replay()
        if entry.arm_backoff > 0:
            entry.arm_backoff -= 1
        else:
            from .. import rng
            entry.armed_nargs = len(args)
            entry.arm((self._host_state(args, kwargs, entry), rng._STATE["offset"]))
            self._armed_fast = (entry, args, _arg_key(args)) if not kwargs else None
        loss = entry.read_loss(released_armed=released)
        if released and entry.arm_backoff == 0:
            entry.arm_penalty = 1
        return loss

    def _capture(self, key, args, kwargs):
        from .constants import HoistedConstantWritten
        rec = self._const_rec.pop(key, None)
        multi = hasattr(self.optim, "reduce_gradients") and getattr(self.optim, "multi_rank", False)
        # with several ranks the step is first captured as ONE graph with the RCCL all-reduce of the
        # flat gradient inside it (RCCL collectives are capturable like NCCL's); if that capture
        # fails the step is captured in the split form [loss + backward] -> eager all-reduce ->
        # [update], which has run on RCCL since round 2
        forms = [False, True] if multi and not getattr(self, "_force_split", False) \
            and _os.environ.get("PYRO_AMD_GRAPH_COLLECTIVE", "1") != "0" else [None]
        for form in forms:
            # with prearm: first with the gate in front of the chained tail (the forward pass of a replay
            # enqueued ahead runs while the host is between two calls), then with the gate as the first
            # node, then without one
            gated = ("late" if self.speculate else True) if (self.prearm and not multi
                                                             and not self._eager_update) else False
            try:
                entry = self._capture_once(key, args, kwargs, rec, force_split=form,
                                           quiet=form is False, with_gate=gated)
            except HoistedConstantWritten:
                # the step writes into a tensor it created with zeros()/ones()/full(): such a
                # tensor has to be filled on every replay -- capture again with the fills inside
                rec = None
                entry = self._capture_once(key, args, kwargs, None, force_split=form,
                                           quiet=form is False, with_gate=gated)
            while getattr(entry, "gate", None) is not None and not entry.gate.armable:
                if _os.environ.get("PYRO_AMD_DEBUG_GATE"):
                    g = entry.gate
                    print("pyro_amd: gate form %r not armable: launches %s gate-aware %s torch operators %s "
                          "emitted %s in front of a late gate %s (not the GLM kernel: %s)"
                          % (gated, g.total, g.aware, g.torch_ops, getattr(g, "emitted", None),
                             getattr(g, "pre", None), getattr(g, "pre_other", None)), flush=True)
                # some node of this step would still run after the gate gave a replay up (a torch
                # kernel, a launch of ours that does not poll the gate), or something other than the
                # GLM kernel runs in front of a late gate: the next weaker form
                gated = True if gated == "late" else False
                self._drop(key)
                entry = self._capture_once(key, args, kwargs, rec, force_split=form,
                                           quiet=form is False, with_gate=gated)
            if entry is not None:
                return entry
            if form is False:
                warnings.warn("pyro_amd: capturing the gradient all-reduce inside the step's graph "
                              "failed; capturing the step in two graphs around an eager collective")
                self.hip_graph = True           # (the failed attempt switched it off)
        return None

    def _capture_once(self, key, args, kwargs, const_rec, force_split=None, quiet=False,
                      with_gate=False, _retried=False):
        from .. import rng
        from ..ops import fuser
        from ..primitives import validation_enabled

        compiled_before = fuser.STATS["loaded"]

        device = None
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                device = a.device
                break
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        cap = rng.GraphCapture(device)
        mailbox = (torch.zeros(1, dtype=torch.float64).pin_memory(),
                   torch.zeros(1, dtype=torch.int64).pin_memory())
        graph = kernels.new_graph()
        graph2 = between = None
        split = hasattr(self.optim, "reduce_gradients") and \
            (getattr(self.optim, "multi_rank", False) or getattr(self, "_force_split", False))
        # split = False with several ranks: the RCCL all-reduce of the flat gradient is captured
        # INSIDE the step's graph -- one replay per step at any world size, no eager collective and
        # no stream hand-over between two graphs (the default, see _capture; PYRO_AMD_GRAPH_COLLECTIVE=0
        # keeps the two-graph form)
        if force_split is not None:
            split = bool(force_split)
        # the dependent small launches that end the step (GLM finalize, ELBO assembly, guide
        # backward, Adam + loss hand-over) become phases of ONE kernel (kernels.chain_recording);
        # PYRO_AMD_CHAIN=0 keeps them as separate graph nodes
        import contextlib
        chained = _os.environ.get("PYRO_AMD_CHAIN", "1") != "0"
        if chained:
            kernels.chain_sync_buffer(device)          # allocated and zeroed outside the capture
        chain = lambda: kernels.chain_recording(device) if chained else contextlib.nullcontext()  # noqa: E731
        self.chain_stats = []
        if const_rec is not None and const_rec.calls:
            from .constants import ConstantReplayer
            consts = ConstantReplayer(const_rec)       # pre-filled copies, made outside the capture
        else:
            consts = None
        hoist = (lambda: consts) if consts is not None else contextlib.nullcontext
        gate = kernels.StepGate(device, late=(with_gate == "late")) \
            if (with_gate and chained and not split) else None
        gated = (lambda: gate) if gate is not None else contextlib.nullcontext
        try:
            with validation_enabled(False):   # validation ran in the eager warm-up steps
                # with a process group alive its watchdog thread polls events while we capture:
                # only THIS thread's calls may invalidate the capture
                multi = getattr(self.optim, "multi_rank", False)
                mode = {"capture_error_mode": "thread_local"} if (split or multi) else {}
                reads = _ReadSet()
                # (parameter blocks of generated-kernel launches: owned by this capture, freed with it)
                blocks = fuser.RtcBlocks()
                with capture_scope(), blocks, torch.cuda.graph(graph, **mode):
                    # (the fuser outermost: it sees what the inner modes let through, last; the read-set
                    #  recorder innermost: it sees every operator first)
                    with fuser.scope(), gated(), cap, chain() as rec, hoist(), reads:
                        if gate is not None:
                            gate.launch()          # first node: holds a replay enqueued ahead of time
                        with poutine.trace(param_only=True) as param_capture:
                            loss = self._loss_device(self.model, self.guide, *args, **kwargs)
                        params = self._params_of(param_capture)
                        loss = loss.detach() if isinstance(loss, torch.Tensor) else \
                            torch.full((), float(loss), device=device)
                        if not split and getattr(self.optim, "fused_publish", False):
                            # the update launch also advances the Philox counter and hands the
                            # loss to the host: the step ends in it
                            self.optim(params, publish=cap.finish_args((loss,) + mailbox))
                        else:
                            if not split and not self._eager_update:
                                self.optim(params)
                                if not getattr(self.optim, "zeroes_grads", False):
                                    zero_grads(params)
                            cap.finish(publish=(loss,) + mailbox)
                    self.chain_stats.append(getattr(rec, "stats", None))
                    self.chain_fused = getattr(rec, "fused", 0)
                if split:
                    # the gradient all-reduce is NOT captured: graph 1 = loss + backward, then an
                    # eager collective, then graph 2 = optimizer update + gradient zeroing
                    optim = self.optim
                    optim.reduce_gradients(params)
                    graph2 = torch.cuda.CUDAGraph()
                    blocks2 = fuser.RtcBlocks()
                    with capture_scope(), blocks2, torch.cuda.graph(graph2, pool=graph.pool(), **mode):
                        with chain():
                            optim.apply(params)
                            if not getattr(optim, "zeroes_grads", False):
                                zero_grads(params)
                    between = lambda: optim.reduce_gradients(params)  # noqa: E731
        except Exception as e:  # noqa: BLE001  (anything that synchronises inside the capture)
            import os
            from .constants import HoistedConstantWritten
            for b_ in (locals().get("blocks"), locals().get("blocks2")):
                if b_ is not None:
                    b_.free()
            if isinstance(e, HoistedConstantWritten):
                raise
            if not _retried and fuser.STATS["loaded"] != compiled_before:
                # a kernel the fuser generated DURING the capture loaded its module there, which a capture
                # does not allow; it is cached now: the second attempt finds it
                return self._capture_once(key, args, kwargs, const_rec, force_split=force_split, quiet=quiet,
                                          with_gate=with_gate, _retried=True)
            if os.environ.get("PYRO_AMD_DEBUG_GRAPH"):
                raise
            if quiet or self._auto_graph:
                self.hip_graph = False
                return None
            warnings.warn("pyro_amd: hipGraph capture of SVI.step failed ({}: {}); continuing "
                          "with eager steps".format(type(e).__name__, e))
            self.hip_graph = False
            return None
        entry = _CapturedStep(graph, cap, loss, graph2, between, mailbox)
        if not split:
            entry.direct = kernels.graph_direct_plan(graph)
        if entry.direct is None and hasattr(graph, "instantiate"):
            for g_ in (graph, graph2):          # (kept hipGraph_t: the executable is made here, not in a timed step)
                if g_ is not None:
                    try:
                        g_.instantiate()
                    except RuntimeError:
                        pass
        if self._eager_update and not split:
            # (the gradients the captured backward writes are the .grad tensors of these leaves: the eager
            #  update reads them, zero_grads clears them in place -- their addresses are part of the graph)
            entry.eager_params = list(params)
        entry.rtc_blocks = (blocks, blocks2 if split else None)     # (die with the entry: RtcBlocks.__del__)
        entry.gate = gate
        entry.reads = tuple(reads.external.values())
        # the graph reads the hoisted constants on every replay: they live as long as the entry
        entry.constants = consts.tensors if consts is not None else []
        entry.constants_served = consts.served if consts is not None else 0
        self._graphs[key] = entry
        return entry
