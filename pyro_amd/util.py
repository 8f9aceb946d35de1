"""Small shared utilities (reference: pyro/util.py:107-146 warn_if_nan/inf, pyro/infer/util.py
torch_item / zero_grads)."""
import contextlib
import itertools
import math
import numbers
import sys
import warnings

import torch


def torch_item(x):
    return x if isinstance(x, numbers.Number) else x.item()


def torch_isnan(x):
    """Does a number or tensor hold ANY NaN (a 0-dim bool tensor for tensors; pyro/util.py:66-75)."""
    if isinstance(x, numbers.Number):
        return x != x
    return torch.isnan(x).any()


def torch_isinf(x):
    """Does a number or tensor hold any +inf or -inf."""
    if isinstance(x, numbers.Number):
        return x in (math.inf, -math.inf)
    return torch.isinf(x).any()


def _caller(filename, lineno):
    # warnings are attributed to whoever asked for the check (and de-duplicated per call site)
    if filename is not None:
        return filename, lineno
    try:
        frame = sys._getframe(2)
    except ValueError:
        return "sys", 1
    return frame.f_code.co_filename, frame.f_lineno


def _any(flag):
    return bool(flag.any()) if torch.is_tensor(flag) else bool(flag)


def warn_if_nan(value, msg="", *, filename=None, lineno=None):
    """Warn if a number or tensor holds a NaN -- and, for a tensor that requires grad, if the
    gradient that later flows into it does ("backward " + msg).  Returns ``value`` itself."""
    filename, lineno = _caller(filename, lineno)
    if torch.is_tensor(value) and value.requires_grad:
        def _check_grad(g):
            warn_if_nan(g, "backward " + msg, filename=filename, lineno=lineno)

        value.register_hook(_check_grad)
    if _any(torch_isnan(value)):
        warnings.warn_explicit("Encountered NaN{}".format(": " + msg if msg else "."), UserWarning,
                               filename, lineno)
    return value


def warn_if_inf(value, msg="", allow_posinf=False, allow_neginf=False, *, filename=None,
                lineno=None):
    """As :func:`warn_if_nan` for +inf / -inf, each of which can be allowed."""
    filename, lineno = _caller(filename, lineno)
    if torch.is_tensor(value) and value.requires_grad:
        def _check_grad(g):
            warn_if_inf(g, "backward " + msg, allow_posinf, allow_neginf, filename=filename,
                        lineno=lineno)

        value.register_hook(_check_grad)
    for allowed, bad, sign in ((allow_posinf, math.inf, "+"), (allow_neginf, -math.inf, "-")):
        if not allowed and _any(value == bad):
            warnings.warn_explicit("Encountered {}inf{}".format(sign, ": " + msg if msg else "."),
                                   UserWarning, filename, lineno)
    return value


def zero_grads(tensors):
    """Zero the .grad of each tensor in place (the reference re-allocates zeros_like every step,
    pyro/infer/util.py:85-91; in-place keeps the allocator out of the step)."""
    with torch.no_grad():
        for p in tensors:
            if p.grad is not None:
                if p.grad.grad_fn is not None:
                    p.grad = p.grad.detach()
                p.grad.zero_()     # works for gradients that are views of a flat buffer too


class capture_scope:
    """Around a hipGraph capture: the cyclic collector must not run inside it.  Captured steps of earlier SVI / NUTS
    objects sit in reference cycles; when the collector gets to them in the middle of ANOTHER capture their
    hipGraph executables and private memory pools are destroyed under the open capture and the HIP runtime aborts
    (seen as "Fatal Python error: Aborted ... Garbage-collecting" in the GPU suite, depending on where the
    collector's thresholds fall).  So: keep the collector off until the capture has ended, and collect BEFORE it
    starts -- garbage graphs die outside it -- but a full collection costs ~70 ms in a process of this size and a
    NUTS run makes several captures within a second (span sizes, compacted rounds: 0.3 s of a 1.2-s run went there):
    at most one full collection every few seconds, the young generations otherwise."""

    _last_full = [0.0]
    FULL_EVERY_S = 5.0

    def __enter__(self):
        import gc
        import time
        self._was = gc.isenabled()
        now = time.monotonic()
        if now - self._last_full[0] >= self.FULL_EVERY_S:
            gc.collect()
            self._last_full[0] = time.monotonic()
        else:
            gc.collect(1)
        gc.disable()
        return self

    def __exit__(self, *exc):
        if self._was:
            import gc
            gc.enable()
        return False


def scalar_like(prototype, fill_value):
    return torch.tensor(fill_value, dtype=prototype.dtype, device=prototype.device)


from .rng import get_rng_state, set_rng_seed, set_rng_state  # noqa: E402,F401  (pyro/util.py:37-63)


# ---- small conveniences user code imports from pyro.util (pyro/util.py:639-725) -----------------------------
class optional:
    """``with optional(ctx, condition):`` enters ``ctx`` only if ``condition`` holds."""

    def __init__(self, context_manager, condition):
        self.context_manager, self.condition = context_manager, condition

    def __enter__(self):
        if self.condition:
            return self.context_manager.__enter__()

    def __exit__(self, exc_type, exc_val, exc_tb):
        if self.condition:
            return self.context_manager.__exit__(exc_type, exc_val, exc_tb)


class ExperimentalWarning(UserWarning):
    pass


@contextlib.contextmanager
def ignore_experimental_warning():
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", category=ExperimentalWarning)
        yield


@contextlib.contextmanager
def ignore_jit_warnings(filter=None):
    """Nothing is traced by a compiler here (HIP graphs capture launches, not Python), so there are no
    tracer warnings to silence; kept so that models written for the reference run unchanged."""
    yield


def jit_iter(tensor):
    return list(tensor)


class timed:
    def __enter__(self):
        import timeit
        self._timer = timeit.default_timer
        self.start = self._timer()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.end = self._timer()
        self.elapsed = self.end - self.start


def torch_float(x):
    return x.float() if isinstance(x, torch.Tensor) else float(x)



# ---- model / guide structure checks (run under pyro.enable_validation; pyro/util.py:284-636) --------------
def _is_subsample_site(site):
    return type(site["fn"]).__name__ == "_Subsample"


def _sample_names(trace, keep):
    return {name for name, site in trace.nodes.items() if site["type"] == "sample" and keep(site)}


def check_traces_match(trace1, trace2):
    """Two runs of the same program: same sample sites (else a warning), same shapes (else ValueError)."""
    names1 = _sample_names(trace1, lambda s: True)
    names2 = _sample_names(trace2, lambda s: True)
    if names1 != names2:
        warnings.warn("Model vars changed: {} vs {}".format(names1, names2))
    for name in names1:
        a, b = trace1.nodes[name], trace2.nodes[name]
        if hasattr(a["fn"], "shape") and hasattr(b["fn"], "shape"):
            shape_a = a["fn"].shape(*a["args"], **a["kwargs"])
            shape_b = b["fn"].shape(*b["args"], **b["kwargs"])
            if shape_a != shape_b:
                raise ValueError("Site dims disagree at site '{}': {} vs {}".format(name, shape_a,
                                                                                  shape_b))


def _inside_plates(shape, event_dim, max_plate_nesting):
    # what is left of the max_plate_nesting plate dims may broadcast freely (enumeration dims)
    if len(shape) > max_plate_nesting:
        return shape[len(shape) - max_plate_nesting - event_dim:]
    return shape


def check_model_guide_match(model_trace, guide_trace, max_plate_nesting=math.inf):
    """What an ELBO assumes of a (model, guide) pair:

    1. every latent of the model is in the guide (or is enumerated in the model) -- else a warning;
       a model-only site asking for SEQUENTIAL enumeration is NotImplementedError;
    2. every guide site is in the model or marked ``infer={"is_auxiliary": True}`` -- else a warning;
    3. every plate of the guide is in the model -- else a warning;
    4. at common sites event_dim and shape agree (up to broadcasting) -- else ValueError;
    5. an auxiliary ``pyro.factor`` in the guide says whether it ``has_rsample`` -- else ValueError."""
    def latent(site):
        return not _is_subsample_site(site)

    guide_vars = _sample_names(guide_trace, latent)
    aux_vars = _sample_names(guide_trace, lambda s: s["infer"].get("is_auxiliary"))
    model_vars = _sample_names(model_trace, lambda s: not s["is_observed"] and latent(s))
    enum_vars = {name for name in model_vars - guide_vars
                 if model_trace.nodes[name]["infer"].get("_enumerate_dim") is not None}
    if aux_vars & model_vars:
        warnings.warn("Found auxiliary vars in the model: {}".format(aux_vars & model_vars))
    if not guide_vars <= model_vars | aux_vars:
        warnings.warn("Found non-auxiliary vars in guide but not model, consider marking these "
                      "infer={{'is_auxiliary': True}}:\n{}".format(guide_vars - aux_vars - model_vars))
    missing = model_vars - guide_vars - enum_vars
    if missing:
        for name in missing:
            if model_trace.nodes[name]["infer"].get("enumerate") == "sequential":
                raise NotImplementedError(
                    "At site {!r}, model-side sequential enumeration is not implemented. Try parallel "
                    "enumeration or guide-side enumeration.".format(name))
        warnings.warn("Found vars in model but not guide: {}".format(missing))

    for name in model_vars & guide_vars:
        m, g = model_trace.nodes[name], guide_trace.nodes[name]
        if hasattr(m["fn"], "event_dim") and hasattr(g["fn"], "event_dim") \
                and m["fn"].event_dim != g["fn"].event_dim:
            raise ValueError("Model and guide event_dims disagree at site '{}': {} vs {}".format(
                name, m["fn"].event_dim, g["fn"].event_dim))
        if not (hasattr(m["fn"], "shape") and hasattr(g["fn"], "shape")):
            continue
        m_shape = m["fn"].shape(*m["args"], **m["kwargs"])
        g_shape = g["fn"].shape(*g["args"], **g["kwargs"])
        if m_shape == g_shape:
            continue
        m_shape = _inside_plates(m_shape, m["fn"].event_dim, max_plate_nesting)
        g_shape = _inside_plates(g_shape, g["fn"].event_dim, max_plate_nesting)
        if m_shape == g_shape:
            continue
        for m_size, g_size in itertools.zip_longest(reversed(m_shape), reversed(g_shape), fillvalue=1):
            if m_size != g_size:
                raise ValueError("Model and guide shapes disagree at site '{}': {} vs {}".format(
                    name, m_shape, g_shape))

    model_plates = _sample_names(model_trace, lambda s: not s["is_observed"] and _is_subsample_site(s))
    guide_plates = _sample_names(guide_trace, _is_subsample_site)
    if not guide_plates <= model_plates:
        warnings.warn("Found plate statements in guide but not model: {}".format(
            guide_plates - model_plates))

    for name, site in guide_trace.nodes.items():
        if site["type"] == "sample" and site["infer"].get("is_auxiliary") \
                and type(site["fn"]).__name__ == "Unit" and "has_rsample" not in site["fn"].__dict__:
            raise ValueError(
                'At guide site pyro.factor("{}",...), missing specification of has_rsample. Please '
                "either set has_rsample=True if the factor statement arises from reparametrized "
                "sampling or has_rsample=False otherwise.".format(name))


def _log_prob_shape(site):
    # the fused path never materialises the un-reduced log_prob; its shape is known without it
    if "log_prob" in site:
        return list(site["log_prob"].shape)
    fn, value = site["fn"], site["value"]
    value_batch = tuple(value.shape[:value.dim() - fn.event_dim]) if torch.is_tensor(value) else ()
    return list(torch.broadcast_shapes(tuple(fn.batch_shape), value_batch))


def check_site_shape(site, max_plate_nesting):
    """The log_prob of a site must have, on the ``max_plate_nesting`` rightmost dims, exactly the sizes
    of the vectorised plates it sits in (at their dims) and may be anything to their left; ValueError
    with the advice the reference gives otherwise."""
    actual = _log_prob_shape(site)
    expected = []
    for frame in site["cond_indep_stack"]:
        if frame.dim is None:
            continue
        assert frame.dim < 0
        expected = [None] * (-frame.dim - len(expected)) + expected
        if expected[frame.dim] is not None:
            raise ValueError("\n  ".join([
                'at site "{}" within plate("{}", dim={}), dim collision'.format(
                    site["name"], frame.name, frame.dim),
                "Try setting dim arg in other plates."]))
        expected[frame.dim] = frame.size
    expected = [-1 if size is None else size for size in expected]
    if len(expected) > max_plate_nesting:
        raise ValueError("\n  ".join([
            'at site "{}", plate stack overflow'.format(site["name"]),
            "Try increasing max_plate_nesting to at least {}".format(len(expected))]))
    if max_plate_nesting < len(actual):
        actual = actual[len(actual) - max_plate_nesting:]
    for actual_size, expected_size in itertools.zip_longest(reversed(actual), reversed(expected),
                                                            fillvalue=1):
        if expected_size != -1 and expected_size != actual_size:
            raise ValueError("\n  ".join([
                'at site "{}", invalid log_prob shape'.format(site["name"]),
                "Expected {}, actual {}".format(expected, actual),
                "Try one of the following fixes:",
                "- enclose the batched tensor in a with pyro.plate(...): context",
                "- .to_event(...) the distribution being sampled",
                "- .permute() data dimensions"]))
    enum_dim = site["infer"].get("_enumerate_dim")
    if enum_dim is not None:
        batch_shape = site["fn"].batch_shape
        if len(batch_shape) >= -enum_dim and batch_shape[enum_dim] != 1:
            raise ValueError("\n  ".join([
                'Enumeration dim conflict at site "{}"'.format(site["name"]),
                "Try increasing pyro.markov history size"]))


def _sequentially_independent(counters1, counters2):
    # two sites in different iterations of one sequential plate
    return any(name in counters2 and counters2[name] != count for name, count in counters1.items())


def check_traceenum_requirements(model_trace, guide_trace):
    """TraceEnum_ELBO sums enumerated variables plate by plate, so nothing outside a plate may depend on
    a variable enumerated inside it.  Dependencies cannot be seen, order can: warn (RuntimeWarning) when
    a site of a strictly smaller plate context comes AFTER a guide-enumerated site of a larger one."""
    enumerated = _sample_names(guide_trace, lambda s: s["infer"].get("enumerate"))
    for role, trace in (("model", model_trace), ("guide", guide_trace)):
        counters = {}                       # site -> {sequential plate: iteration}
        contexts = {}                       # frozenset of vectorised frames -> enumerated sites in it
        for name, site in trace.nodes.items():
            if site["type"] != "sample":
                continue
            counter = {f.name: f.counter for f in site["cond_indep_stack"] if not f.vectorized}
            context = frozenset(f for f in site["cond_indep_stack"] if f.vectorized)
            for inner, names in contexts.items():
                if not context < inner:
                    continue
                late = sorted(n for n in names if not _sequentially_independent(counter, counters[n]))
                if not late:
                    continue
                broken = sorted(f.name for f in inner - context)
                warnings.warn("\n  ".join([
                    'at {} site "{}", possibly invalid dependency.'.format(role, name),
                    'Expected site "{}" to precede sites "{}"'.format(name, '", "'.join(late)),
                    'to avoid breaking independence of plates "{}"'.format('", "'.join(broken))]),
                    RuntimeWarning)
            counters[name] = counter
            if name in enumerated:
                contexts.setdefault(context, set()).add(name)


def check_if_enumerated(guide_trace):
    """The estimators that do not enumerate warn when a guide site asks for it."""
    enumerated = [name for name, site in guide_trace.nodes.items()
                  if site["type"] == "sample" and site["infer"].get("enumerate")]
    if enumerated:
        warnings.warn("\n".join([
            "Found sample sites configured for enumeration:" + ", ".join(enumerated),
            "If you want to enumerate sites, you need to use TraceEnum_ELBO instead."]))
