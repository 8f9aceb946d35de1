"""The effect handlers the two hot paths need (reference: pyro/poutine/*_messenger.py).

trace / replay / block / condition / scale / mask / plate (independence + subsampling +
broadcasting) / enum (parallel enumeration) / seed.  Same names, arguments and message
semantics as the reference so existing models run unchanged; implementation is independent.
"""
import numbers
import warnings
from collections import namedtuple

import torch

from .. import rng as _rng
from . import settings
from .runtime import (_DIM_ALLOCATOR, _ENUM_ALLOCATOR, Messenger, NonlocalExit, _BlockLike, _BoundHandler,
                      apply_stack,
                      new_message)
from .trace import Trace


# ---------------------------------------------------------------------------------------------
def _dense_edges(trace):
    """Edges site -> later site unless the two sit in different iterations of a common sequential
    plate (the dependency structure "dense" traces carry; trace_messenger.py:20-46)."""
    from .util import site_is_subsample
    sites = [(name, node) for name, node in trace.nodes.items()
             if node["type"] == "sample" and not site_is_subsample(node)]
    for k, (name, node) in enumerate(sites):
        for past_name, past in sites[:k]:
            independent = any(a.name == b.name and a.counter != b.counter
                              for a, b in zip(node["cond_indep_stack"], past["cond_indep_stack"]))
            if not independent:
                trace.add_edge(past_name, name)


class _TraceHandler(_BoundHandler):
    """``poutine.trace(fn)``: calling it runs ``fn`` under the messenger and records the call's
    arguments and return value as the ``_INPUT`` / ``_RETURN`` nodes; ``.trace`` is the live trace of
    the last call, ``.get_trace(*args)`` runs and returns a copy (trace_messenger.py:158-217)."""

    def __call__(self, *args, **kwargs):
        msngr = self.handler
        with msngr:
            msngr.trace.add_node("_INPUT", name="_INPUT", type="args", args=args, kwargs=kwargs)
            try:
                ret = self.fn(*args, **kwargs)
            except (ValueError, RuntimeError) as e:
                shapes = msngr.trace.format_shapes()
                raise type(e)("{}\n{}".format(e, shapes)).with_traceback(e.__traceback__) from e
            msngr.trace.add_node("_RETURN", name="_RETURN", type="return", value=ret)
        return ret

    @property
    def trace(self):
        return self.handler.trace

    def get_trace(self, *args, **kwargs):
        self(*args, **kwargs)
        return self.handler.get_trace()


class TraceMessenger(Messenger):
    """Record every message into a Trace (reference: trace_messenger.py:49-156)."""

    def __init__(self, graph_type=None, param_only=None):
        super().__init__()
        self.graph_type = "flat" if graph_type is None else graph_type
        assert self.graph_type in ("flat", "dense")
        self.param_only = bool(param_only)
        self.trace = Trace(self.graph_type)

    def __enter__(self):
        self.trace = Trace(self.graph_type)
        return super().__enter__()

    def __exit__(self, *args):
        if self.param_only:
            for name, node in list(self.trace.nodes.items()):
                if node["type"] != "param":
                    self.trace.remove_node(name)
        if self.graph_type == "dense":
            _dense_edges(self.trace)
        return super().__exit__(*args)

    def __call__(self, fn):
        if not callable(fn):
            raise ValueError("{} is not callable, did you mean to pass it as a keyword arg?"
                             .format(fn))
        return _TraceHandler(self, fn)

    def get_trace(self):
        return self.trace.copy()

    def _reset(self):
        fresh = Trace(self.graph_type)
        if "_INPUT" in self.trace.nodes:        # a restarted run is still the same call
            fresh.add_node("_INPUT", **self.trace.nodes["_INPUT"])
        self.trace = fresh

    def _pyro_post_sample(self, msg):
        if self.param_only:
            return
        if msg["infer"].get("_do_not_trace"):
            return
        self.trace.add_node(msg["name"], **msg.copy())

    def _pyro_post_param(self, msg):
        self.trace.add_node(msg["name"], **msg.copy())


# ---------------------------------------------------------------------------------------------
class ReplayMessenger(Messenger):
    """Replay sample sites from a guide trace (reference: replay_messenger.py:60-90)."""

    def __init__(self, trace=None, params=None):
        super().__init__()
        if trace is None and params is None:
            raise ValueError("must provide trace or params to replay against")
        self.trace, self.params = trace, params

    def _pyro_sample(self, msg):
        name = msg["name"]
        if self.trace is not None and name in self.trace:
            guide_msg = self.trace.nodes[name]
            if msg["is_observed"]:
                return
            if guide_msg["type"] != "sample" or guide_msg["is_observed"]:
                raise RuntimeError("site {} must be sampled in trace".format(name))
            msg["done"] = True
            msg["value"] = guide_msg["value"]
            msg["infer"] = guide_msg["infer"]
            msg["_replayed"] = True       # primitives.sample: the value reaches MODEL code

    def _pyro_param(self, msg):
        if self.params is not None and msg["name"] in self.params:
            p = self.params[msg["name"]]
            msg["done"] = True
            msg["value"] = p["value"] if isinstance(p, dict) else p


# ---------------------------------------------------------------------------------------------
class _HideRule:
    """block's decision rule as a picklable object (wrapped models are torch.save'd by users)."""

    def __init__(self, hide_all, hide, expose, hide_types, expose_types):
        self.hide_all, self.hide, self.expose = hide_all, hide, expose
        self.hide_types, self.expose_types = hide_types, expose_types

    def __call__(self, msg):
        kind = "observe" if msg["type"] == "sample" and msg["is_observed"] else msg["type"]
        if msg["name"] in self.hide or kind in self.hide_types:
            return True
        return self.hide_all and msg["name"] not in self.expose and kind not in self.expose_types


class _Negated:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, msg):
        return not self.fn(msg)


class BlockMessenger(Messenger, _BlockLike):
    """Hide sites from handlers outside (reference: block_messenger.py:19-84 decision rule, :101-168).

    With no arguments everything is hidden.  ``hide=`` / ``hide_types=`` hide only what they name;
    ``expose=`` / ``expose_types=`` hide everything except what they name; an observed sample site has
    type ``"observe"`` for the two ``*_types`` lists.  ``hide_fn`` / ``expose_fn`` replace the rule."""

    def __init__(self, hide_fn=None, expose_fn=None, hide_all=True, expose_all=False, hide=None,
                 expose=None, hide_types=None, expose_types=None):
        super().__init__()
        if not (hide_fn is None or expose_fn is None):
            raise ValueError("Only specify one of hide_fn or expose_fn")
        if hide_fn is not None:
            self.hide_fn = hide_fn
        elif expose_fn is not None:
            self.hide_fn = _Negated(expose_fn)
        else:
            self.hide_fn = self._default_rule(hide_all, expose_all, hide, expose, hide_types,
                                              expose_types)

    @staticmethod
    def _default_rule(hide_all, expose_all, hide, expose, hide_types, expose_types):
        assert (hide_all is False and expose_all is False) or hide_all != expose_all, \
            "cannot hide and expose a site"
        # naming things to hide turns "hide all" off, naming things to expose turns it on; in the
        # order the reference applies them (hide, expose, hide_types, expose_types)
        for given, turns_on in ((hide, False), (expose, True), (hide_types, False), (expose_types, True)):
            if given is not None:
                hide_all = turns_on
        hide, expose = frozenset(hide or ()), frozenset(expose or ())
        hide_types, expose_types = frozenset(hide_types or ()), frozenset(expose_types or ())
        assert hide.isdisjoint(expose), "cannot hide and expose a site"
        assert hide_types.isdisjoint(expose_types), "cannot hide and expose a site type"
        return _HideRule(hide_all, hide, expose, hide_types, expose_types)

    def _process_message(self, msg):
        msg["stop"] = bool(self.hide_fn(msg))


# ---------------------------------------------------------------------------------------------
class ConditionMessenger(Messenger):
    """Fix sample sites to observed values (reference: condition_messenger.py)."""

    def __init__(self, data):
        super().__init__()
        self.data = data

    def _pyro_sample(self, msg):
        name = msg["name"]
        if isinstance(self.data, Trace):
            if name in self.data.nodes:
                msg["value"] = self.data.nodes[name]["value"]
                msg["is_observed"] = msg["value"] is not None
        elif name in self.data:
            msg["value"] = self.data[name]
            msg["is_observed"] = msg["value"] is not None


class SubstituteMessenger(Messenger):
    """Fix ``pyro.param`` sites to given values (reference: substitute_messenger.py:18-107): the
    value of a param whose user-facing name is in ``data`` is replaced; the rest are untouched."""

    def __init__(self, data):
        super().__init__()
        self.data = data
        self._seen = {}

    def __enter__(self):
        self._seen = {}
        if settings.validation_enabled() and isinstance(self.data, dict):
            self._hits, self._misses = set(), set()
        return super().__enter__()

    def __exit__(self, *args):
        self._seen = {}
        if settings.validation_enabled() and isinstance(self.data, dict):
            extra = set(self.data) - self._hits
            if extra:
                warnings.warn("pyro.module data did not find params ['{}']. Did you instead mean one "
                              "of ['{}']?".format("', '".join(extra), "', '".join(self._misses)))
        return super().__exit__(*args)

    def _pyro_param(self, msg):
        from ..params import user_param_name
        name = msg["name"]
        key = user_param_name(name)
        validating = settings.validation_enabled() and isinstance(self.data, dict)
        if key not in self.data.keys():
            if validating:
                self._misses.add(key)
            return
        msg["value"] = self.data[key]
        if validating:
            self._hits.add(key)
        if name in self._seen:          # the same param statement again: same value
            msg["value"] = self._seen[name]["value"]
        else:
            self._seen[name] = msg


class InferConfigMessenger(Messenger):
    """``msg["infer"].update(config_fn(msg))`` at every sample and param site
    (reference: infer_config_messenger.py:14-58)."""

    def __init__(self, config_fn):
        super().__init__()
        self.config_fn = config_fn

    def _pyro_sample(self, msg):
        msg["infer"].update(self.config_fn(msg))

    _pyro_param = _pyro_sample


class DoMessenger(Messenger):
    """Intervention (reference: do_messenger.py:14-97): the original statement still runs as its own,
    un-intervened site; the site the rest of the program sees carries the given value, is renamed
    ``name + "__CF"``, marked observed and hidden from the handlers outside."""

    def __init__(self, data):
        super().__init__()
        self.data = data
        self._token = str(id(self))

    def _pyro_sample(self, msg):
        if msg.get("_intervener_id") == self._token or self.data.get(msg["name"]) is None:
            return
        if msg.get("_intervener_id") is not None:
            warnings.warn("Attempting to intervene on variable {} multiple times,this is almost "
                          "certainly incorrect behavior".format(msg["name"]), RuntimeWarning)
        msg["_intervener_id"] = self._token
        twin = msg.copy()                       # the un-intervened statement, sent down the stack
        twin["cond_indep_stack"] = ()           # the plates add themselves again on the way
        apply_stack(twin)
        intervention = self.data[msg["name"]]
        msg["name"] = msg["name"] + "__CF"
        if isinstance(intervention, numbers.Number):
            intervention = torch.tensor(intervention)
        elif not isinstance(intervention, torch.Tensor):
            raise NotImplementedError("Interventions of type {} not implemented (yet)".format(
                type(intervention)))
        msg["value"], msg["is_observed"], msg["stop"] = intervention, True, True


class EscapeMessenger(Messenger):
    """Leave the program at the first sample site ``escape_fn`` accepts, by raising
    :class:`NonlocalExit` carrying that site once the handlers below have seen it
    (reference: escape_messenger.py:9-44)."""

    def __init__(self, escape_fn):
        super().__init__()
        self.escape_fn = escape_fn

    def _pyro_sample(self, msg):
        if self.escape_fn(msg):
            msg["done"] = True
            msg["stop"] = True

            def leave(m):
                raise NonlocalExit(m)

            msg["continuation"] = leave


class LiftMessenger(Messenger):
    """Turn ``pyro.param`` statements into ``pyro.sample`` statements drawn from ``prior``: a
    distribution, a stochastic function of the param's initial tensor, or a dict of those by param
    name -- unnamed params stay params (reference: lift_messenger.py:21-140)."""

    def __init__(self, prior):
        super().__init__()
        self.prior = prior
        self._drawn = {}

    def __enter__(self):
        self._drawn = {}
        if settings.validation_enabled() and isinstance(self.prior, dict):
            self._hits, self._misses = set(), set()
        return super().__enter__()

    def __exit__(self, *args):
        self._drawn = {}
        if settings.validation_enabled() and isinstance(self.prior, dict):
            extra = set(self.prior) - self._hits
            if extra:
                warnings.warn("pyro.module prior did not find params ['{}']. Did you instead mean one "
                              "of ['{}']?".format("', '".join(extra), "', '".join(self._misses)))
        return super().__exit__(*args)

    def _pyro_param(self, msg):
        from ..params import user_param_name
        from torch.distributions import Distribution
        name = msg["name"]
        key = user_param_name(name)
        prior = self.prior
        if isinstance(prior, dict):
            validating = settings.validation_enabled()
            if key not in prior:
                if validating:
                    self._misses.add(key)
                return
            if validating:
                self._hits.add(key)
            prior = prior[key]
        if isinstance(prior, Distribution):
            msg["fn"], msg["args"], msg["kwargs"], msg["infer"] = prior, (), {}, {}
        elif callable(prior):
            # called with what followed the name in the param statement (its initial tensor);
            # a function-valued prior given directly (not through a dict) also hides the site
            if not isinstance(self.prior, dict):
                msg["stop"] = True
            msg["fn"], msg["args"] = prior, msg["args"][1:]
        else:
            raise TypeError("prior must be a distribution, a callable or a dict of those")
        msg["type"] = "sample"
        if name in self._drawn:                 # the same param statement again: the same draw
            msg["value"], msg["is_observed"], msg["stop"] = self._drawn[name]["value"], True, True
        else:
            self._drawn[name] = msg
            msg["is_observed"] = False


class EqualizeMessenger(Messenger):
    """Force the primitive statements whose names match ``sites`` (full-match regular expressions)
    to take the value of the first of them.  Later matches become observed ``Delta`` sites that
    score nothing, or -- ``keep_dist=True`` -- keep their distribution, which conditions the model
    on the values being equal (reference: equalize_messenger.py:14-103)."""

    def __init__(self, sites, type="sample", keep_dist=False):
        super().__init__()
        self.sites = [sites] if isinstance(sites, str) else sites
        self.type, self.keep_dist = type, keep_dist
        self.value = None

    def __enter__(self):
        self.value = None
        return super().__enter__()

    def _matches(self, msg):
        import re
        return msg["type"] == self.type and any(re.fullmatch(pattern, msg["name"]) is not None
                                                for pattern in self.sites)

    def _postprocess_message(self, msg):
        if self.value is None and self._matches(msg):
            assert msg["value"] is not None
            self.value = msg["value"]

    def _process_message(self, msg):
        if self.value is None or not self._matches(msg):
            return
        msg["value"] = self.value
        if msg["type"] == "sample":
            msg["is_observed"] = True
            if not self.keep_dist:
                from ..distributions import Delta
                msg["infer"] = {"_deterministic": True}
                msg["fn"] = Delta(self.value, event_dim=msg["fn"].event_dim).mask(False)


class ReparamMessenger(Messenger):
    """Replace the sample sites named by ``config`` (a dict site name -> reparameteriser, or a function of
    the site returning one or None) by the reparameteriser's auxiliary sites + a deterministic map
    (reference: reparam_messenger.py:35-122).  As a decorator the reparameterisers also see the call's
    ``args, kwargs``."""

    def __init__(self, config):
        super().__init__()
        assert isinstance(config, dict) or callable(config)
        self.config = config
        self._args_kwargs = None

    def __call__(self, fn):
        if not callable(fn):
            raise ValueError("{} is not callable, did you mean to pass it as a keyword arg?".format(fn))
        return _ReparamHandler(self, fn)

    def _pyro_sample(self, msg):
        if type(msg["fn"]).__name__ == "_Subsample":
            return
        reparam = self.config.get(msg["name"]) if isinstance(self.config, dict) else self.config(msg)
        if reparam is None:
            return
        reparam.args_kwargs = self._args_kwargs
        try:
            new = reparam.apply({"name": msg["name"], "fn": msg["fn"], "value": msg["value"],
                                 "is_observed": msg["is_observed"]})
        finally:
            reparam.args_kwargs = None
        if new["value"] is not None and msg["value"] is not None and msg["value"] is not new["value"]:
            assert new["value"].shape == msg["value"].shape
        msg["fn"], msg["value"], msg["is_observed"] = new["fn"], new["value"], new["is_observed"]


class _ReparamHandler(_BoundHandler):
    def __call__(self, *args, **kwargs):
        self.handler._args_kwargs = args, kwargs
        try:
            with self.handler:
                return self.fn(*args, **kwargs)
        finally:
            self.handler._args_kwargs = None


class UnconditionMessenger(Messenger):
    def _pyro_sample(self, msg):
        if msg["is_observed"]:
            msg["is_observed"] = False
            msg["infer"]["was_observed"] = True
            msg["infer"]["obs"] = msg["value"]
            msg["value"] = None
            msg["done"] = False


# ---------------------------------------------------------------------------------------------
class ScaleMessenger(Messenger):
    """Multiply the log-probability of enclosed sites (reference: scale_messenger.py:52)."""

    def __init__(self, scale):
        super().__init__()
        if isinstance(scale, torch.Tensor):
            if settings.validation_enabled() and not bool((scale > 0).all()):
                raise ValueError("Expected scale > 0 but got {}".format(scale))
        elif not (scale > 0):
            raise ValueError("Expected scale > 0 but got {}".format(scale))
        self.scale = scale

    def _process_message(self, msg):
        msg["scale"] = self.scale * msg["scale"]


class MaskMessenger(Messenger):
    """Mask the log-probability of enclosed sites (reference: mask_messenger.py:39)."""

    def __init__(self, mask):
        super().__init__()
        if isinstance(mask, torch.Tensor):
            if mask.dtype != torch.bool:
                raise ValueError("Expected mask to be a BoolTensor but got {}".format(type(mask)))
        elif mask not in (True, False):
            raise ValueError("Expected mask to be a boolean but got {}".format(type(mask)))
        self.mask = mask

    def _process_message(self, msg):
        msg["mask"] = self.mask if msg["mask"] is None else msg["mask"] & self.mask


from .runtime import get_mask, get_plates  # noqa: E402,F401


# ---------------------------------------------------------------------------------------------
class CondIndepStackFrame(namedtuple("CondIndepStackFrame",
                                     ["name", "dim", "size", "counter", "full_size"])):
    @property
    def vectorized(self):
        return self.dim is not None

    def __eq__(self, other):
        return (type(self) is type(other) and self[:-1] == other[:-1]
                and (self.full_size == other.full_size if not isinstance(self.full_size, torch.Tensor)
                     else True))

    def __hash__(self):
        return hash((self.name, self.dim, self.size, self.counter))

    def __str__(self):
        return self.name


_ARANGE_CACHE = {}


def _cached_arange(size, device):
    """Index vector of a full (not subsampled) plate.  The reference re-creates
    torch.arange(size) at every plate entry (subsample_messenger.py:60); for a 1e6..1e7 plate
    that alone costs milliseconds per step, so it is built once per (size, device)."""
    key = (int(size), str(device))
    t = _ARANGE_CACHE.get(key)
    if t is None:
        if len(_ARANGE_CACHE) > 64:
            _ARANGE_CACHE.clear()
        t = _ARANGE_CACHE[key] = torch.arange(size, device=device)
    return t


class _Subsample:
    """The distribution-like object behind a plate's subsample site."""

    def __init__(self, size, subsample_size, device):
        self.size, self.subsample_size, self.device = size, subsample_size, device
        self.has_rsample = False
        self.batch_shape = torch.Size()
        self.event_shape = torch.Size()

    def __call__(self, sample_shape=torch.Size()):
        ss = self.subsample_size
        if ss is None or ss >= self.size:
            return _cached_arange(self.size, self.device)
        return torch.randperm(self.size, device=self.device)[:ss].clone()

    def log_prob(self, x):
        return torch.tensor(0.0, device=self.device)


class PlateMessenger(Messenger):
    """pyro.plate: conditional independence + subsampling scale + automatic broadcasting
    (reference: indep_messenger.py:47-147, subsample_messenger.py:74-217,
    broadcast_messenger.py:46-93, plate_messenger.py:17)."""

    def __init__(self, name, size=None, subsample_size=None, subsample=None, dim=None,
                 use_cuda=None, device=None):
        super().__init__()
        if size is None:
            assert subsample_size is None and subsample is None
            size = -1
            subsample_size = -1
        else:
            if size == 0:
                raise ZeroDivisionError("size cannot be zero")
            if not isinstance(size, numbers.Number) or size <= 0:
                if not isinstance(size, torch.Tensor):
                    raise ValueError("size must be a positive number, got {}".format(size))
            if use_cuda is not None:
                device = "cuda" if use_cuda else "cpu"
            msg = new_message("sample", name, _Subsample(size, subsample_size, device),
                              value=subsample)
            apply_stack(msg)
            subsample = msg["value"]
            if subsample_size is None:
                subsample_size = len(subsample)
            elif subsample is not None and subsample_size != len(subsample):
                raise ValueError("subsample_size does not match len(subsample), {} vs {}. Did you "
                                 "accidentally use different subsample_size in the model and "
                                 "guide?".format(subsample_size, len(subsample)))
        if size == 0:
            raise ZeroDivisionError("size cannot be zero")
        self.name, self.size, self.subsample_size = name, size, subsample_size
        self.dim, self.device = dim, device
        self._indices = subsample
        self._vectorized = None
        self.counter = 0
        self._installed = False

    @property
    def indices(self):
        if self._indices is None:
            self._indices = _cached_arange(self.size, self.device)
        return self._indices

    def __enter__(self):
        if self._vectorized is not False:
            self._vectorized = True
        if self._vectorized is True:
            self.dim = _DIM_ALLOCATOR.allocate(self.name, self.dim)
        if settings.validation_enabled():
            self._check_no_conflict()
        super().__enter__()
        # a plate of unknown size (``pyro.plate("name")``) has no indices to hand out
        return self.indices if self._indices is not None or self.size >= 0 else None

    def __exit__(self, *args):
        if self._vectorized is True:
            _DIM_ALLOCATOR.free(self.name, self.dim)
        return super().__exit__(*args)

    def _check_no_conflict(self):
        pass

    def __iter__(self):
        # sequential plate
        if self._vectorized is True or self.dim is not None:
            raise ValueError("cannot use plate {} as both vectorized and non-vectorized "
                             "independence context".format(self.name))
        self._vectorized = False
        self.dim = None
        with_ = self
        for i in self.indices:
            self.counter += 1
            with with_:
                yield i if isinstance(i, numbers.Number) else i.item()

    def _reset(self):
        if self._vectorized:
            try:
                _DIM_ALLOCATOR.free(self.name, self.dim)
            except Exception:
                pass
        self._vectorized = None
        self.counter = 0

    def _process_message(self, msg):
        frame = CondIndepStackFrame(self.name, self.dim if self._vectorized else None,
                                    self.subsample_size, self.counter, self.size)
        msg["cond_indep_stack"] = (frame,) + msg["cond_indep_stack"]
        if self.size != self.subsample_size:
            msg["scale"] = msg["scale"] * self.size / self.subsample_size
        # broadcasting: expand the distribution's batch shape to the plate sizes
        if msg["type"] == "sample" and self._vectorized and not msg["done"] \
                and not isinstance(msg["fn"], _Subsample):
            self._broadcast(msg, frame)

    @staticmethod
    def _broadcast(msg, frame):
        dist = msg["fn"]
        if not hasattr(dist, "batch_shape") or msg["infer"].get("_enumerate_dim") is not None \
                and False:
            return
        actual = list(dist.batch_shape)
        target = list(actual)
        if frame.dim is None:
            return
        k = -frame.dim
        if len(target) < k:
            target = [1] * (k - len(target)) + target
        if target[-k] == 1 and frame.size != 1 and frame.size != -1:
            target[-k] = frame.size
        elif target[-k] != frame.size and frame.size != -1:
            raise ValueError(
                "Shape mismatch inside plate('{}') at site {} dim {}, {} vs {}".format(
                    frame.name, msg["name"], frame.dim, frame.size, target[-k]))
        if target != actual:
            msg["fn"] = dist.expand(torch.Size(target))

    def _postprocess_message(self, msg):
        """``pyro.param(..., event_dim=e)`` and ``pyro.subsample(data, event_dim=e)`` inside a
        vectorised plate: the tensor's dim at this plate must have the plate's full size (or 1, or be
        absent) and is cut down to the current subsample (subsample_messenger.py:176-217)."""
        if msg["type"] not in ("param", "subsample") or self.dim is None:
            return
        event_dim = msg["kwargs"].get("event_dim")
        if event_dim is None:
            return
        assert event_dim >= 0
        dim = self.dim - event_dim
        value = msg["value"]
        shape = value.shape
        if len(shape) < -dim or shape[dim] == 1:
            return
        if settings.validation_enabled() and shape[dim] != self.size:
            statement = "pyro.param({}, ..., event_dim={})".format(msg["name"], event_dim) \
                if msg["type"] == "param" else "pyro.subsample(..., event_dim={})".format(event_dim)
            raise ValueError("Inside pyro.plate({}, {}, dim={}) invalid shape of {}: {}".format(
                self.name, self.size, self.dim, statement, shape))
        if self.subsample_size < self.size:
            new_value = value.index_select(dim, self._indices.to(value.device))
            if msg["type"] == "param":
                param = getattr(value, "_pyro_unconstrained_param", None)
                if param is None:
                    param = value.unconstrained()
                # which rows of the parameter this step touches, for optimisers that care
                if not hasattr(param, "_pyro_subsample"):
                    param._pyro_subsample = {}
                param._pyro_subsample[dim] = self._indices
                new_value._pyro_unconstrained_param = param
            msg["value"] = new_value


# ---------------------------------------------------------------------------------------------
class EnumMessenger(Messenger):
    """Parallel enumeration of discrete sample sites marked infer={"enumerate": "parallel"}
    (reference: enum_messenger.py:114-254): the site's value becomes its support on a tensor dim
    to the left of every plate; the support indices are int64 and exact.

    Every sample site that passes through also gets ``infer["_dim_to_id"]``: which enumerated
    VARIABLE (unique id) each enumeration dim of its tensors refers to at this point of the
    program.  Outside pyro.markov a dim belongs to one variable for good; inside, dims are recycled
    once their variable has left the Markov scope, and the id is what tells x_{t-2} from x_t."""

    def __init__(self, first_available_dim=None):
        super().__init__()
        assert first_available_dim is None or first_available_dim < 0
        self.first_available_dim = first_available_dim

    def __enter__(self):
        if self.first_available_dim is not None:
            _ENUM_ALLOCATOR.set_first_available_dim(self.first_available_dim)
        self._markov_depths = {}      # site name -> markov depth at which it was sampled
        self._param_dims = {}         # site name -> {enum dim: id} visible to its parameters
        self._value_dims = {}         # site name -> {enum dim: id} its value really depends on
        return super().__enter__()

    def _pyro_sample(self, msg):
        if msg["done"] or not isinstance(msg["fn"], torch.distributions.Distribution):
            return
        scope = msg["infer"].get("_markov_scope")          # site name -> depth (MarkovMessenger)
        param_dims = dict(_ENUM_ALLOCATOR.dim_to_id)
        if scope is not None:
            for name, depth in scope.items():
                # a site whose markov context has exited since is no longer visible
                if self._markov_depths.get(name) == depth:
                    param_dims.update(self._value_dims.get(name, {}))
            self._markov_depths[msg["name"]] = msg["infer"]["_markov_depth"]
        self._param_dims[msg["name"]] = param_dims
        if msg["is_observed"]:
            return
        strategy = msg["infer"].get("enumerate")
        if strategy != "parallel":
            return          # sequential: branched on by SequentialEnumMessenger (guide sites)
        if msg["infer"].get("num_samples") is not None:
            return self._sample_locally(msg, scope, param_dims)
        dist = msg["fn"]
        if not getattr(dist, "has_enumerate_support", False):
            raise NotImplementedError("{} does not support enumeration".format(type(dist)))
        value = dist.enumerate_support(expand=False)
        dim, id_ = _ENUM_ALLOCATOR.allocate(None if scope is None else set(param_dims))
        event_dim = len(dist.event_shape)
        # enumerate_support(expand=False) is [K, 1, ..., 1, *event_shape]: put the support axis at
        # tensor dim `dim` (counted left of the event dims), whatever rank the batch shape has
        shape = value.shape
        ev = tuple(shape[len(shape) - event_dim:]) if event_dim else ()
        assert all(n == 1 for n in shape[1:len(shape) - event_dim]), \
            "enumerate_support(expand=False) must not expand batch dims"
        tag = getattr(value, "_pyro_categorical_support", None)
        value = value.reshape(shape[:1] + (1,) * (-1 - dim) + ev)
        if msg["infer"].get("expand", False):
            # enumerate_support(expand=True): the support axis, then the full batch shape
            batch = tuple(dist.batch_shape)
            value = value.expand(shape[:1] + (1,) * (-1 - dim - len(batch)) + batch + ev)
        if tag is not None:
            value._pyro_categorical_support = tag
        value_dims = {d: param_dims[d] for d in range(event_dim - value.dim(), 0)
                      if d in param_dims and value.size(d - event_dim) > 1}
        value_dims[dim] = id_
        msg["infer"]["_enumerate_dim"] = dim
        msg["infer"]["_dim_to_id"] = value_dims
        msg["value"] = value
        msg["done"] = True

    def _sample_locally(self, msg, scope, param_dims):
        """``infer={"enumerate": "parallel", "num_samples": n}``: n draws on a fresh dim instead of the
        support (local Monte-Carlo "enumeration"; enum_messenger.py:17-134).  Without ``expand`` the
        draws keep only the plate dims of the batch shape: along every other batch dim of size > 1 (a
        population of upstream particles) draw s keeps ONE ancestor -- its own index ("diagonal", the
        default) or a uniformly chosen one ("mixture")."""
        dist, n = msg["fn"], msg["infer"]["num_samples"]
        event_dim = len(dist.event_shape)
        if n > 1 and msg["infer"].get("expand", False):
            value = dist(sample_shape=torch.Size([n]))
        elif n > 1:
            strategy = msg["infer"].get("tmc", "diagonal")
            if strategy not in ("diagonal", "mixture"):
                raise ValueError("{} not a valid TMC strategy".format(strategy))
            value = dist(sample_shape=torch.Size([n]))
            keep = [1] * len(dist.batch_shape)
            for f in msg["cond_indep_stack"]:
                if f.vectorized:
                    keep[f.dim] = f.size if f.size > 0 else dist.batch_shape[f.dim]
            for k, want in enumerate(keep):
                d = 1 + k                                   # dim of `value` (0 is the draw axis)
                have = value.shape[d]
                if have > 1 and want == 1:
                    if strategy == "diagonal":
                        ancestor = torch.arange(have, device=value.device)
                        if have != n:
                            raise ValueError("diagonal TMC needs as many draws ({}) as upstream "
                                             "particles ({}) at site '{}'".format(n, have, msg["name"]))
                    else:
                        ancestor = torch.randint(have, (n,), device=value.device)
                    index_shape = list(value.shape)
                    index_shape[d] = 1
                    index = ancestor.reshape((n,) + (1,) * (value.dim() - 1)).expand(index_shape)
                    value = value.gather(d, index)
            assert tuple(value.shape) == (n,) + tuple(keep) + tuple(dist.event_shape)
        else:
            raise ValueError("num_samples must be greater than 1 at site '{}'".format(msg["name"]))
        dim, id_ = _ENUM_ALLOCATOR.allocate(None if scope is None else set(param_dims))
        actual_dim = -1 - len(dist.batch_shape)             # where the draw axis sits now
        if dim < actual_dim:
            value = value.reshape(value.shape[:1] + (1,) * (actual_dim - dim) + value.shape[1:])
        elif actual_dim < dim:
            assert value.size(dim - event_dim) == 1, \
                "pyro.markov dim conflict at dim {}".format(actual_dim)
            value = value.transpose(dim - event_dim, actual_dim - event_dim)
            while value.dim() and value.size(0) == 1:
                value = value.squeeze(0)
        value_dims = {d: param_dims[d] for d in range(event_dim - value.dim(), 0)
                      if d in param_dims and value.size(d - event_dim) > 1}
        value_dims[dim] = id_
        msg["infer"]["_enumerate_dim"] = dim
        msg["infer"]["_dim_to_id"] = value_dims
        msg["value"] = value
        msg["done"] = True

    def _pyro_post_sample(self, msg):
        if not isinstance(msg["fn"], torch.distributions.Distribution) or msg["value"] is None:
            return
        if not msg["is_observed"] and msg["infer"].get("enumerate") == "sequential" \
                and msg["infer"].get("_enum_total") is None:
            # neither branched on in the guide nor replayed from a guide site that was
            raise NotImplementedError(
                "At site {!r}, model-side sequential enumeration is not implemented. Try parallel "
                "enumeration or guide-side enumeration.".format(msg["name"]))
        value = msg["value"]
        event_dim = len(msg["fn"].event_shape)
        shape = value.shape[:value.dim() - event_dim]
        dim_to_id = msg["infer"].setdefault("_dim_to_id", {})
        for d, i in self._param_dims.get(msg["name"], {}).items():
            dim_to_id.setdefault(d, i)
        self._value_dims[msg["name"]] = {d: i for d, i in dim_to_id.items()
                                         if len(shape) >= -d and shape[d] > 1}


class SequentialEnumMessenger(Messenger):
    """One branch of the sequential enumeration of a program's discrete sites marked
    infer={"enumerate": "sequential"} (reference: pyro/infer/enum.py:88-135 iter_discrete_traces
    with poutine.queue / iter_discrete_escape / iter_discrete_extend).  ``assignment`` fixes the
    support index of the sites already branched on; the first time a further sequential site is
    met the run continues with its first value and the alternatives are pushed on ``queue`` --
    every element of the queue is one complete assignment prefix, every run one trace."""

    def __init__(self, assignment, queue):
        super().__init__()
        self.assignment, self.queue = dict(assignment), queue

    def _pyro_sample(self, msg):
        if msg["done"] or msg["is_observed"] or msg["infer"].get("enumerate") != "sequential":
            return
        dist = msg["fn"]
        if not getattr(dist, "has_enumerate_support", False):
            raise NotImplementedError("{} does not support enumeration".format(type(dist)))
        support = dist.enumerate_support(expand=bool(msg["infer"].get("expand", False)))
        name = msg["name"]
        if name not in self.assignment:
            for k in range(1, support.shape[0]):
                alt = dict(self.assignment)
                alt[name] = k
                self.queue.append(alt)
            self.assignment[name] = 0
        msg["infer"]["_enum_total"] = support.shape[0]
        msg["value"] = support[self.assignment[name]]
        msg["done"] = True


class MarkovMessenger(Messenger):
    """Markov dependency declaration (reference: markov_messenger.py:17-130): sample sites in the
    body of ``for t in pyro.markov(range(T), history=h)`` may depend only on sites of the current
    and the previous ``h`` iterations, so enumeration dims are recycled after h + 1 steps.  Marks
    every sample site with its Markov scope (names of the sites visible from it and their depth)
    for EnumMessenger.  Re-entrant: the same instance is entered once per iteration."""

    def __init__(self, history=1, keep=False, dim=None, name=None):
        super().__init__()
        assert history >= 0
        if dim is not None or name is not None:
            raise NotImplementedError("vectorized markov is not implemented (neither in the "
                                      "reference): leave dim and name unset")
        self.history, self.keep = history, keep
        self._iterable = None
        self._pos = -1
        self._stack = []
        self._ref_count = 0

    def generator(self, iterable):
        self._iterable = iterable
        return self

    def __iter__(self):
        from contextlib import ExitStack
        with ExitStack() as stack:
            for value in self._iterable:
                stack.enter_context(self)
                yield value

    def __enter__(self):
        self._pos += 1
        if len(self._stack) <= self._pos:
            self._stack.append(set())
        self._ref_count += 1                      # re-entrant: on the handler stack only once
        if self._ref_count == 1:
            super().__enter__()
        return self

    def __exit__(self, *args):
        if not self.keep:
            self._stack.pop()
        self._pos -= 1
        self._ref_count -= 1
        if self._ref_count == 0:
            return super().__exit__(*args)

    def _pyro_sample(self, msg):
        from collections import Counter
        if msg["done"] or type(msg["fn"]).__name__ == "_Subsample":
            return
        infer = msg["infer"]
        scope = infer.setdefault("_markov_scope", Counter())     # site name -> markov depth
        for pos in range(max(0, self._pos - self.history), self._pos + 1):
            scope.update(self._stack[pos])
        infer["_markov_depth"] = 1 + infer.get("_markov_depth", 0)
        self._stack[self._pos].add(msg["name"])


# ---------------------------------------------------------------------------------------------
class SeedMessenger(Messenger):
    """Run the wrapped program under a fixed RNG seed, restoring the state afterwards."""

    def __init__(self, rng_seed):
        super().__init__()
        assert isinstance(rng_seed, int)
        self.rng_seed = rng_seed

    def __enter__(self):
        self.old_state = _rng.get_rng_state()
        _rng.set_rng_seed(self.rng_seed)
        return super().__enter__()

    def __exit__(self, *args):
        _rng.set_rng_state(self.old_state)
        return super().__exit__(*args)


# ---------------------------------------------------------------------------------------------
def _make_handler(cls):
    def handler(fn=None, *args, **kwargs):
        if fn is not None and not (callable(fn) or isinstance(fn, (list, tuple))):
            raise ValueError("{} is not callable, did you mean to pass it as a keyword arg?"
                             .format(fn))
        m = cls(*args, **kwargs)
        return m(fn) if fn is not None else m

    handler.__name__ = cls.__name__.replace("Messenger", "").lower()
    handler.__doc__ = cls.__doc__
    return handler


trace = _make_handler(TraceMessenger)
replay = _make_handler(ReplayMessenger)
block = _make_handler(BlockMessenger)
condition = _make_handler(ConditionMessenger)
uncondition = _make_handler(UnconditionMessenger)
substitute = _make_handler(SubstituteMessenger)
infer_config = _make_handler(InferConfigMessenger)
do = _make_handler(DoMessenger)
escape = _make_handler(EscapeMessenger)
reparam = _make_handler(ReparamMessenger)


def broadcast(fn=None):
    """Deprecated in the reference (plates broadcast by themselves since 0.3): the identity."""
    return Messenger()(fn) if fn is not None else Messenger()


equalize = _make_handler(EqualizeMessenger)
lift = _make_handler(LiftMessenger)


def queue(fn=None, queue=None, max_tries=None, extend_fn=None, escape_fn=None, num_samples=None):
    """Sequential search: take a partial trace from ``queue``, replay ``fn`` against it until the
    first site ``escape_fn(partial_trace, msg)`` accepts, push ``extend_fn``'s extensions of the
    trace-so-far, repeat; the first complete run's return value is the result
    (reference: pyro/poutine/handlers.py:542-606)."""
    import functools
    from . import util
    max_tries = int(1e6) if max_tries is None else max_tries
    extend_fn = util.enum_extend if extend_fn is None else extend_fn
    escape_fn = util.discrete_escape if escape_fn is None else escape_fn
    num_samples = -1 if num_samples is None else num_samples

    def wrapper(wrapped):
        def _fn(*args, **kwargs):
            for _ in range(max_tries):
                assert not queue.empty(), "trying to get() from an empty queue will deadlock"
                partial_trace = queue.get()
                traced = trace(escape(replay(wrapped, trace=partial_trace),
                                      escape_fn=functools.partial(escape_fn, partial_trace)))
                try:
                    return traced(*args, **kwargs)
                except NonlocalExit as exit_:
                    exit_.reset_stack()
                    for longer in extend_fn(traced.trace.copy(), exit_.site, num_samples=num_samples):
                        queue.put(longer)
            raise ValueError("max tries ({}) exceeded".format(max_tries))

        return _fn

    return wrapper(fn) if fn is not None else wrapper
scale = _make_handler(ScaleMessenger)
mask = _make_handler(MaskMessenger)
enum = _make_handler(EnumMessenger)
seed = _make_handler(SeedMessenger)


def markov(fn=None, history=1, keep=False, dim=None, name=None):
    """Markov dependency declaration: as a context manager, a decorator, or around an iterable
    (``for t in markov(range(T))``) (reference: pyro/poutine/handlers.py markov)."""
    if fn is None:
        return MarkovMessenger(history=history, keep=keep, dim=dim, name=name)
    if not callable(fn):
        # an iterable
        return MarkovMessenger(history=history, keep=keep, dim=dim, name=name).generator(iterable=fn)
    return MarkovMessenger(history=history, keep=keep, dim=dim, name=name)(fn)
