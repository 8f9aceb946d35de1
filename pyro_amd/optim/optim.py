"""Optimizers (reference: pyro/optim/optim.py:72-200 PyroOptim, clipped_adam.py).

``PyroOptim`` keeps the reference behaviour: one torch optimizer object per parameter, created
lazily the first time the parameter is seen.

``Adam`` / ``ClippedAdam`` are the MI355X-native replacement: all
unconstrained parameters are views into ONE flat device buffer (values, grads, moments), and a
step is a single HIP kernel launch (pa_adam_step) that also zeroes the gradient, instead of a
Python loop over per-parameter optimizers plus a zeros_like re-allocation per parameter.  The
flat gradient buffer is also what the RCCL all-reduce wrapper reduces in one collective.
"""
import torch

from .. import kernels
from ..params import _PARAM_STORE


def is_scheduler(optimizer):
    """Is this a torch learning-rate scheduler (it holds its optimizer) rather than an optimizer?"""
    return hasattr(optimizer, "optimizer")


def _state_of(obj):
    # checkpoint of one parameter's optimizer, or of its scheduler AND optimizer (optim.py:45-68)
    if is_scheduler(obj):
        return {"scheduler": obj.state_dict(), "optimizer": obj.optimizer.state_dict()}
    return obj.state_dict()


def _restore(obj, state):
    if is_scheduler(obj):
        obj.load_state_dict(state["scheduler"])
        obj.optimizer.load_state_dict(state["optimizer"])
    else:
        obj.load_state_dict(state)


class PyroOptim:
    """Wrap a torch optimizer class; one instance per parameter (reference-compatible)."""

    def __init__(self, optim_constructor, optim_args, clip_args=None):
        self.pt_optim_constructor = optim_constructor
        assert callable(optim_args) or isinstance(optim_args, dict)
        self.pt_optim_args = optim_args
        self.pt_clip_args = clip_args
        self.optim_objs = {}
        self._state_waiting_to_be_consumed = {}

    def _args_for(self, param):
        """The optimizer arguments of one parameter.  A callable receives the parameter's name the way
        ``nn.Module.named_parameters()`` spells it (``"net.linear.bias"``; a ``pyro.module`` name's ``$$$`` is a
        dot) -- or, for the deprecated two-argument form, (module name, parameter name)."""
        if not callable(self.pt_optim_args):
            return self.pt_optim_args
        from ..params import (module_from_param_with_module_name, normalize_param_name, user_param_name)
        import inspect
        name = _PARAM_STORE.param_name(param)
        try:
            two = len(inspect.signature(self.pt_optim_args).parameters) == 2
        except (TypeError, ValueError):
            two = False
        if two and name is not None:
            chosen = self.pt_optim_args(module_from_param_with_module_name(name), user_param_name(name))
        else:
            chosen = self.pt_optim_args(name if name is None else normalize_param_name(name))
        assert isinstance(chosen, dict), "per-param optim arg must return defaults dictionary"
        return chosen

    def _get_optim(self, param):
        return self.pt_optim_constructor([param], **self._args_for(param))

    @staticmethod
    def _optimizer_of(obj):
        # a learning-rate scheduler (PyroLRScheduler) holds its optimizer
        return getattr(obj, "optimizer", obj)

    def __call__(self, params, *args, **kwargs):
        for p in params:
            if p not in self.optim_objs:
                self.optim_objs[p] = self._get_optim(p)
                name = _PARAM_STORE.param_name(p)
                state = self._state_waiting_to_be_consumed.pop(name, None)
                if state is not None:
                    _restore(self.optim_objs[p], state)
            if self.pt_clip_args is not None:
                clip = self.pt_clip_args
                if callable(clip):
                    # per-parameter clipping: called with (module name, parameter name) as the
                    # reference does (optim.py:238-255)
                    from ..params import module_from_param_with_module_name, user_param_name
                    pname = _PARAM_STORE.param_name(p)
                    clip = clip(module_from_param_with_module_name(pname), user_param_name(pname))
                    assert isinstance(clip, dict), "per-param clip arg must return defaults dictionary"
                if "clip_norm" in clip:
                    torch.nn.utils.clip_grad_norm_([p], clip["clip_norm"])
                if "clip_value" in clip:
                    torch.nn.utils.clip_grad_value_([p], clip["clip_value"])
            self._optimizer_of(self.optim_objs[p]).step(*args, **kwargs)

    def get_state(self):
        return {_PARAM_STORE.param_name(p): _state_of(o) for p, o in self.optim_objs.items()}

    def set_state(self, state_dict):
        self._state_waiting_to_be_consumed.update(state_dict)

    def save(self, filename):
        torch.save(self.get_state(), filename)

    def load(self, filename, map_location=None):
        self.set_state(torch.load(filename, map_location=map_location, weights_only=False))


class PyroLRScheduler(PyroOptim):
    """A torch learning-rate scheduler per dynamically created parameter (reference: lr_scheduler.py).
    ``optim_args`` holds ``"optimizer"`` (a torch optimizer class), ``"optim_args"`` (its arguments) and the
    scheduler's own arguments; ``svi.step`` steps the optimizers, ``scheduler.step()`` the schedules::

        scheduler = pyro.optim.ExponentialLR({"optimizer": torch.optim.SGD, "optim_args": {"lr": 0.01},
                                              "gamma": 0.1})
    """

    def __init__(self, scheduler_constructor, optim_args, clip_args=None):
        optim_args = dict(optim_args)
        self.pt_scheduler_constructor = scheduler_constructor
        pt_optim_constructor = optim_args.pop("optimizer")
        optim_kwargs = optim_args.pop("optim_args")
        self.kwargs = optim_args
        super().__init__(pt_optim_constructor, optim_kwargs, clip_args)

    def _get_optim(self, param):
        return self.pt_scheduler_constructor(super()._get_optim(param), **self.kwargs)

    def step(self, *args, **kwargs):
        """Advance every parameter's schedule (same arguments as the torch scheduler's ``step``)."""
        for scheduler in self.optim_objs.values():
            scheduler.step(*args, **kwargs)


def _torch_wrappers():
    """``pyro.optim.<Name>`` for every optimizer class of torch.optim and every scheduler class of
    torch.optim.lr_scheduler (reference: pytorch_optimizers.py builds the same names in a loop)."""
    import functools
    out = {}
    for name, cls in vars(torch.optim).items():
        if isinstance(cls, type) and issubclass(cls, torch.optim.Optimizer) \
                and cls not in (torch.optim.Optimizer, torch.optim.LBFGS):
            out[name] = functools.partial(PyroOptim, cls)
    base = torch.optim.lr_scheduler.LRScheduler
    for name, cls in vars(torch.optim.lr_scheduler).items():
        if isinstance(cls, type) and (issubclass(cls, base) or name == "ReduceLROnPlateau") \
                and cls is not base and not name.startswith("_"):
            out[name] = functools.partial(PyroLRScheduler, cls)
    return out


def TorchAdam(optim_args, clip_args=None):
    return PyroOptim(torch.optim.Adam, optim_args, clip_args)


def SGD(optim_args, clip_args=None):
    return PyroOptim(torch.optim.SGD, optim_args, clip_args)


def AdamW(optim_args, clip_args=None):
    return PyroOptim(torch.optim.AdamW, optim_args, clip_args)


def RMSprop(optim_args, clip_args=None):
    return PyroOptim(torch.optim.RMSprop, optim_args, clip_args)


class _FlatGroup:
    """One generation of parameters: every tensor created in the same optimizer call.  Their
    values, gradients and Adam moments are views into four flat buffers that are never
    re-allocated (a captured hipGraph keeps pointing at live memory), and they share one device
    step counter for as long as every step touches all of them."""

    def __init__(self, params):
        proto = params[0]
        kernels._require_gpu(proto)  # HIP-only flat fused kernel (TorchAdam is the generic one)
        self.params = list(params)
        self.index = {}
        total = sum(p.numel() for p in params)
        self.flat, self.grad, self.exp_avg, self.exp_avg_sq = (
            torch.zeros(total, dtype=proto.dtype, device=proto.device) for _ in range(4))
        off = 0
        for p in params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.detach().reshape(-1))
            if p.grad is not None:
                self.grad[off:off + n].copy_(p.grad.reshape(-1))
            self.index[p] = (off, n)
            # the parameter (and its .grad) becomes a view into the flat buffers
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            off += n
        self.step_dev = torch.zeros(2, dtype=torch.int64, device=proto.device)   # [step, ticket]
        self.own_steps = None      # param -> own [step, ticket] once the group is stepped raggedly

    def adopt_grads(self, params):
        """autograd may have re-created .grad for a parameter whose grad was None: fold it back."""
        for p in params:
            o, n = self.index[p]
            if p.grad is None:
                p.grad = self.grad[o:o + n].view(p.shape)
            elif p.grad.data_ptr() != self.grad.data_ptr() + o * self.grad.element_size():
                self.grad[o:o + n].add_(p.grad.reshape(-1))
                p.grad = self.grad[o:o + n].view(p.shape)


class _FlatAdam:
    """Adam over flat buffers.  Semantics of the reference's per-parameter optimizers
    (pyro/optim/optim.py:117-155) are kept: a step moves exactly the parameters passed to it, and a
    parameter's bias correction / learning-rate decay count ITS OWN steps.  Parameters created in
    one call form a group that is updated by ONE launch while every step touches the whole group
    (the normal case: all parameters exist after the first step and are used by every step);
    parameters created later form a new group with its own step counter, and a group that is ever
    stepped partially falls back to one launch and one counter per parameter."""

    _clipped = False
    zeroes_grads = True  # the kernel zeroes the flat gradient in the same pass

    def __init__(self, optim_args, clip_args=None):
        if callable(optim_args):
            raise ValueError("the flat fused Adam takes one dict of arguments for all parameters; "
                             "use pyro_amd.optim.TorchAdam for per-parameter arguments")
        a = dict(optim_args)
        self.lr = float(a.pop("lr", 1e-3))
        self.betas = tuple(a.pop("betas", (0.9, 0.999)))
        self.eps = float(a.pop("eps", 1e-8))
        self.weight_decay = float(a.pop("weight_decay", 0.0))
        self.clip_norm = float(a.pop("clip_norm", 10.0 if self._clipped else 0.0))
        self.lrd = float(a.pop("lrd", 1.0))
        if a:
            raise ValueError("unsupported optimizer arguments: {}".format(sorted(a)))
        self._groups = []
        self._group_of = {}        # leaf -> its _FlatGroup
        self._pending_state = None
        self.grad_hook = None      # e.g. the RCCL all-reduce of a flat gradient buffer

    # ---- views used by the distributed wrapper and the tests ------------------------------------
    def grad_buffers(self):
        return [g.grad for g in self._groups]

    @property
    def grad(self):
        """The flat gradient buffer (the RCCL message) when all parameters live in one group."""
        if not self._groups:
            return None
        if len(self._groups) > 1:
            raise RuntimeError("parameters were created in several optimizer calls: use "
                               "grad_buffers()")
        return self._groups[0].grad

    @property
    def flat(self):
        return self._groups[0].flat if len(self._groups) == 1 else None

    @property
    def step_dev(self):
        return self._groups[0].step_dev if len(self._groups) == 1 else None

    fused_publish = True     # __call__(publish=...) folds the loss hand-over into the update launch

    def _launch(self, flat, grad, m, v, step_dev, publish):
        kernels.adam_step(flat, grad, m, v, step_dev, lr=self.lr, betas=self.betas, eps=self.eps,
                          weight_decay=self.weight_decay, clip_norm=self.clip_norm, lrd=self.lrd,
                          clipped=self._clipped, zero_grad=True, publish=publish)

    def __call__(self, params, *args, skip_grad_hook=False, publish=None, **kwargs):
        params = list(params)
        new = [p for p in params if p not in self._group_of]
        if new:
            # deterministic order (same on every rank): by param-store name
            new.sort(key=lambda p: _PARAM_STORE.param_name(p) or "")
            group = _FlatGroup(new)
            self._groups.append(group)
            for p in new:
                self._group_of[p] = group
            self._restore_into(group)
        wanted, fresh = set(params), set(new)
        work = []
        for g in self._groups:
            touched = [p for p in g.params if p in wanted]
            if not touched:
                continue            # the reference only steps parameters seen in this step
            g.adopt_grads([p for p in touched if p not in fresh])
            work.append((g, touched))
        if self.grad_hook is not None and not skip_grad_hook:
            for g, _ in work:
                self.grad_hook(g.grad)
        launches = []
        for g, touched in work:
            if g.own_steps is None and len(touched) == len(g.params):
                launches.append((g.flat, g.grad, g.exp_avg, g.exp_avg_sq, g.step_dev))
                continue
            if g.own_steps is None:          # first partial step: every parameter gets its own
                g.own_steps = {p: g.step_dev.clone() for p in g.params}     # counter from here on
            for p in touched:
                o, n = g.index[p]
                launches.append((g.flat[o:o + n], g.grad[o:o + n], g.exp_avg[o:o + n],
                                 g.exp_avg_sq[o:o + n], g.own_steps[p]))
        launches = [l for l in launches if l[0].numel()]
        for i, l in enumerate(launches):
            self._launch(*l, publish=publish if i == len(launches) - 1 else None)
        if publish is not None and not launches:
            kernels.publish_scalar(*publish)

    # ---- checkpointing (reference: PyroOptim.get_state / set_state, optim.py:157-200) -----------
    def _group_args(self):
        g = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}
        if self._clipped:
            g.update({"clip_norm": self.clip_norm, "lrd": self.lrd})
        return g

    def get_state(self):
        """{parameter name: state_dict of a one-parameter torch optimizer} -- the layout of the
        reference's ``PyroOptim.get_state`` (pyro/optim/optim.py:157-166: ``{"state": {0: {"step",
        "exp_avg", "exp_avg_sq"}}, "param_groups": [...]}`` per parameter), so that a checkpoint
        depends neither on how parameters were grouped here nor on which of the two
        implementations wrote it."""
        state = {}
        for g in self._groups:
            shared = None if g.own_steps is not None else int(g.step_dev[0].item())
            for p in g.params:
                o, n = g.index[p]
                step = shared if shared is not None else int(g.own_steps[p][0].item())
                group = dict(self._group_args(), params=[0])
                if self._clipped:
                    group["lr"] = self.lr * self.lrd ** step
                state[_PARAM_STORE.param_name(p)] = {
                    "state": {0: {"step": torch.tensor(float(step)),
                                  "exp_avg": g.exp_avg[o:o + n].clone().view(p.shape),
                                  "exp_avg_sq": g.exp_avg_sq[o:o + n].clone().view(p.shape)}},
                    "param_groups": [group]}
        return state

    def set_state(self, state):
        """Restores moments and step counts; parameters that do not exist yet pick their entry up
        when they are first seen (as the reference's _state_waiting_to_be_consumed does)."""
        self._pending_state = dict(state)
        for g in self._groups:
            self._restore_into(g)

    def _restore_into(self, g):
        if not self._pending_state:
            return
        steps = {}
        for p in g.params:
            name = _PARAM_STORE.param_name(p)
            entry = self._pending_state.pop(name, None)
            if entry is None:
                continue
            if "state" in entry:          # torch optimizer layout (this class's own, the reference's)
                inner = entry["state"]
                if not inner:
                    continue              # the parameter had not been stepped when it was saved
                entry = next(iter(inner.values()))
            o, n = g.index[p]
            for key, buf in (("exp_avg", g.exp_avg), ("exp_avg_sq", g.exp_avg_sq)):
                t = entry[key]
                if t.numel() != n:
                    raise ValueError("optimizer state of {!r}: {} has {} elements, the parameter {}"
                                     .format(name, key, t.numel(), n))
                buf[o:o + n].copy_(t.reshape(-1).to(buf.device, buf.dtype))
            steps[p] = int(entry["step"])
        if not steps:
            return
        if len(steps) == len(g.params) and len(set(steps.values())) == 1 and g.own_steps is None:
            g.step_dev[0] = next(iter(steps.values()))
        else:
            if g.own_steps is None:
                g.own_steps = {p: g.step_dev.clone() for p in g.params}
            for p, st in steps.items():
                g.own_steps[p][0] = st

    def save(self, filename):
        torch.save(self.get_state(), filename)

    def load(self, filename, map_location=None):
        self.set_state(torch.load(filename, map_location=map_location, weights_only=False))


def _per_parameter_route(optim_args, clip_args):
    """Arguments the flat buffers cannot express -- a callable giving every parameter its own
    settings, gradient clipping by ``clip_args`` (pyro/optim/optim.py:96-114, 140-153) -- take the
    reference's per-parameter route."""
    return callable(optim_args) or clip_args is not None


class Adam(_FlatAdam):
    """torch.optim.Adam semantics, one fused launch for all parameters.  ``Adam(callable)`` or
    ``Adam(args, clip_args)`` returns the per-parameter ``PyroOptim(torch.optim.Adam, ...)``."""

    def __new__(cls, optim_args, clip_args=None):
        if _per_parameter_route(optim_args, clip_args):
            return PyroOptim(torch.optim.Adam, optim_args, clip_args)
        return super().__new__(cls)


class NoUpdate(_FlatAdam):
    """The optimizer that does nothing -- ``SVI(model, guide, NoUpdate(), loss).step()`` is the reference's step
    (pyro/infer/svi.py:134-162) without its middle line: the loss and the gradients are computed, the loss is
    handed to the host, the gradients are zeroed; parameters (and the moments this object never uses) stay
    bit for bit.  It keeps the flat buffers of the package's Adam, so a captured step ends in the same fused
    tail kernel as the full step (the tail's update arithmetic returns at once for a zero learning rate): what
    a step costs WITHOUT the update, on the same launches -- bench.py's `config2_loss_and_grads_only`.  To read
    gradients use ``loss.loss_and_grads`` (they accumulate in ``.grad`` as the reference's do)."""

    def __init__(self, optim_args=None, clip_args=None):
        if optim_args or clip_args:
            raise ValueError("NoUpdate takes no arguments")
        super().__init__({"lr": 0.0})


class ClippedAdam(_FlatAdam):
    """pyro.optim.ClippedAdam semantics (element-wise gradient clamp + lr decay)."""

    _clipped = True

    def __new__(cls, optim_args, clip_args=None):
        if _per_parameter_route(optim_args, clip_args):
            from .clipped_adam import ClippedAdam as TorchClippedAdam
            return PyroOptim(TorchClippedAdam, optim_args, clip_args)
        return super().__new__(cls)
