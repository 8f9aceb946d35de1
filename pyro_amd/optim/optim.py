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
        if callable(self.pt_optim_args):
            name = _PARAM_STORE.param_name(param)
            return self.pt_optim_args(name)
        return self.pt_optim_args

    def __call__(self, params, *args, **kwargs):
        for p in params:
            if p not in self.optim_objs:
                self.optim_objs[p] = self.pt_optim_constructor([p], **self._args_for(p))
                name = _PARAM_STORE.param_name(p)
                state = self._state_waiting_to_be_consumed.pop(name, None)
                if state is not None:
                    self.optim_objs[p].load_state_dict(state)
            if self.pt_clip_args is not None:
                clip = self.pt_clip_args
                if "clip_norm" in clip:
                    torch.nn.utils.clip_grad_norm_([p], clip["clip_norm"])
                if "clip_value" in clip:
                    torch.nn.utils.clip_grad_value_([p], clip["clip_value"])
            self.optim_objs[p].step(*args, **kwargs)

    def get_state(self):
        return {_PARAM_STORE.param_name(p): o.state_dict() for p, o in self.optim_objs.items()}

    def set_state(self, state_dict):
        self._state_waiting_to_be_consumed.update(state_dict)

    def save(self, filename):
        torch.save(self.get_state(), filename)

    def load(self, filename, map_location=None):
        self.set_state(torch.load(filename, map_location=map_location, weights_only=False))


def TorchAdam(optim_args, clip_args=None):
    return PyroOptim(torch.optim.Adam, optim_args, clip_args)


def SGD(optim_args, clip_args=None):
    return PyroOptim(torch.optim.SGD, optim_args, clip_args)


def AdamW(optim_args, clip_args=None):
    return PyroOptim(torch.optim.AdamW, optim_args, clip_args)


def RMSprop(optim_args, clip_args=None):
    return PyroOptim(torch.optim.RMSprop, optim_args, clip_args)


class _FlatAdam:
    """Adam over one flat buffer holding every parameter seen so far."""

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
        self._params = []          # list of leaf tensors, in flat order
        self._index = {}           # leaf -> (offset, numel)
        self.flat = self.grad = self.exp_avg = self.exp_avg_sq = self.step_dev = None
        self.grad_hook = None      # e.g. the RCCL all-reduce of the flat gradient

    # -- flat buffer management ----------------------------------------------------------------
    def _rebuild(self, new_params):
        olds = (self.flat, self.grad, self.exp_avg, self.exp_avg_sq)
        old_n = 0 if self.flat is None else self.flat.numel()
        params = self._params + new_params
        proto = params[0]
        kernels._require_gpu(proto)  # HIP-only flat fused kernel (TorchAdam is the generic one)
        total = sum(p.numel() for p in params)
        bufs = [torch.zeros(total, dtype=proto.dtype, device=proto.device) for _ in range(4)]
        if old_n:
            for b, o in zip(bufs, olds):
                b[:old_n].copy_(o)
        off = old_n
        for p in new_params:
            n = p.numel()
            bufs[0][off:off + n].copy_(p.detach().reshape(-1))
            if p.grad is not None:
                bufs[1][off:off + n].copy_(p.grad.reshape(-1))
            self._index[p] = (off, n)
            off += n
        self._params = params
        self.flat, self.grad, self.exp_avg, self.exp_avg_sq = bufs
        # every parameter (and its .grad) becomes a view into the flat buffers
        for p in params:
            o, n = self._index[p]
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
        if self.step_dev is None:
            self.step_dev = torch.zeros(2, dtype=torch.int64, device=proto.device)  # [step, ticket]

    fused_publish = True     # __call__(publish=...) folds the loss hand-over into the update launch

    def __call__(self, params, *args, skip_grad_hook=False, publish=None, **kwargs):
        new = [p for p in params if p not in self._index]
        if new:
            # deterministic order (same on every rank): by param-store name
            new.sort(key=lambda p: _PARAM_STORE.param_name(p) or "")
            self._rebuild(new)
        else:
            for p in params:  # autograd may have re-created .grad for a param whose grad was None
                o, n = self._index[p]
                if p.grad is None:
                    p.grad = self.grad[o:o + n].view(p.shape)
                elif p.grad.data_ptr() != self.grad.data_ptr() + o * self.grad.element_size():
                    self.grad[o:o + n].add_(p.grad.reshape(-1))
                    p.grad = self.grad[o:o + n].view(p.shape)
        if self.grad_hook is not None and not skip_grad_hook:
            self.grad_hook(self.grad)
        kernels.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_dev,
                          lr=self.lr, betas=self.betas, eps=self.eps,
                          weight_decay=self.weight_decay, clip_norm=self.clip_norm, lrd=self.lrd,
                          clipped=self._clipped, zero_grad=True,
                          publish=publish if self.flat.numel() else None)
        if publish is not None and not self.flat.numel():
            kernels.publish_scalar(*publish)

    def get_state(self):
        return {"names": [_PARAM_STORE.param_name(p) for p in self._params],
                "exp_avg": None if self.exp_avg is None else self.exp_avg.clone(),
                "exp_avg_sq": None if self.exp_avg_sq is None else self.exp_avg_sq.clone(),
                "step": None if self.step_dev is None else int(self.step_dev[0].item())}

    def set_state(self, state):
        self._pending_state = state

    def save(self, filename):
        torch.save(self.get_state(), filename)

    def load(self, filename, map_location=None):
        self.set_state(torch.load(filename, map_location=map_location, weights_only=False))


class Adam(_FlatAdam):
    """torch.optim.Adam semantics, one fused launch for all parameters."""


class ClippedAdam(_FlatAdam):
    """pyro.optim.ClippedAdam semantics (element-wise gradient clamp + lr decay)."""

    _clipped = True
