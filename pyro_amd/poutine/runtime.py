"""Effect-handler runtime: the host-side control plane that drives the HIP kernels.

Mirrors the interface of the reference's ``pyro.poutine.runtime`` (message dict fields,
bottom-up ``_process_message`` / top-down ``_postprocess_message`` traversal of a global
handler stack; reference: pyro/poutine/runtime.py:108-181,334-390) so that models and guides
written for the reference run unchanged.  It is deliberately small: host code is plumbing, the
numerics live in libpyro_amd.so.
"""
import contextlib
import functools

_PYRO_STACK = []


class NonlocalExit(Exception):
    """Raised by a handler to abort the traced program and resume elsewhere (escape/queue)."""

    def __init__(self, site, *args):
        super().__init__(*args)
        self.site = site

    def reset_stack(self):
        # unwind handlers installed inside the aborted program
        for frame in reversed(_PYRO_STACK):
            frame._reset()
            if isinstance(frame, _BlockLike) and frame.hide_fn(self.site):
                break


class _BlockLike:
    """marker mixin, see handlers.BlockMessenger"""


class _DimAllocator:
    """Allocates negative tensor dims to vectorised plates (reference: runtime.py:184-243)."""

    def __init__(self):
        self._stack = []  # index i <-> dim -(i+1); None = free

    def allocate(self, name, dim):
        if name in self._stack:
            raise ValueError('duplicate plate "{}"'.format(name))
        if dim is None:
            dim = -1
            while -dim <= len(self._stack) and self._stack[-1 - dim] is not None:
                dim -= 1
        elif dim >= 0:
            raise ValueError("Expected dim < 0 to index from the right, actual {}".format(dim))
        while -dim > len(self._stack):
            self._stack.append(None)
        if self._stack[-1 - dim] is not None:
            raise ValueError('at plates "{}" and "{}", collide at dim={}\nTry moving the dim of '
                             "one plate to the left, e.g. dim={}".format(
                                 name, self._stack[-1 - dim], dim, dim - 1))
        self._stack[-1 - dim] = name
        return dim

    def free(self, name, dim):
        free_idx = -1 - dim
        assert self._stack[free_idx] == name
        self._stack[free_idx] = None
        while self._stack and self._stack[-1] is None:
            self._stack.pop()


_DIM_ALLOCATOR = _DimAllocator()


class _EnumAllocator:
    """Allocates tensor dims (left of all plates) and unique integer ids to enumerated sample
    sites (reference: runtime.py:246-303).  A site outside any pyro.markov context gets a global
    dim that is never reused; inside one it gets the first dim not occupied by a variable that is
    still visible from it (``scope_dims``), so a chain of T variables needs history + 1 dims."""

    def __init__(self):
        self.set_first_available_dim(-1)

    def set_first_available_dim(self, first_available_dim):
        assert first_available_dim < 0
        self.next_available_dim = first_available_dim
        self.next_available_id = 0
        self.dim_to_id = {}                     # global dims only

    def allocate(self, scope_dims=None):
        id_ = self.next_available_id
        self.next_available_id += 1
        dim = self.next_available_dim
        if dim == -float("inf"):
            raise ValueError("max_plate_nesting must be set to a finite value for parallel "
                             "enumeration")
        if scope_dims is None:
            self.next_available_dim -= 1
            self.dim_to_id[dim] = id_
        else:
            while dim in scope_dims:
                dim -= 1
        return dim, id_


_ENUM_ALLOCATOR = _EnumAllocator()


def new_message(type_, name, fn, args=(), kwargs=None, value=None, is_observed=False,
                infer=None):
    return {
        "type": type_, "name": name, "fn": fn, "is_observed": is_observed, "args": args,
        "kwargs": kwargs or {}, "value": value, "scale": 1.0, "mask": None,
        "cond_indep_stack": (), "done": False, "stop": False, "continuation": None,
        "infer": {} if infer is None else infer,
    }


def default_process_message(msg):
    if msg["done"] or msg["is_observed"] or msg["value"] is not None:
        msg["done"] = True
        return msg
    msg["value"] = msg["fn"](*msg["args"], **msg["kwargs"])
    msg["done"] = True
    return msg


def apply_stack(msg):
    """Send ``msg`` down the handler stack (innermost first), run the default behaviour, then
    let the visited handlers post-process it (outermost visited first)."""
    pointer = 0
    for pointer, frame in enumerate(reversed(_PYRO_STACK)):
        frame._process_message(msg)
        if msg["stop"]:
            break
    default_process_message(msg)
    for frame in _PYRO_STACK[len(_PYRO_STACK) - pointer - 1:]:
        frame._postprocess_message(msg)
    cont = msg["continuation"]
    if cont is not None:
        cont(msg)
    return msg


def am_i_wrapped():
    return len(_PYRO_STACK) > 0


def effectful(fn=None, type=None):
    """Wrap a callable so that calling it sends a message of type ``type`` through the stack."""
    if fn is None:
        return functools.partial(effectful, type=type)
    assert type is not None and type != "sample"

    @functools.wraps(fn)
    def _fn(*args, **kwargs):
        name = kwargs.pop("name", None)
        infer = kwargs.pop("infer", {})
        value = kwargs.pop("obs", None)
        if not am_i_wrapped():
            return fn(*args, **kwargs)
        msg = new_message(type, name, fn, args, kwargs, value, infer=infer)
        apply_stack(msg)
        return msg["value"]

    _fn._is_effectful = True
    return _fn


class Messenger:
    """Base effect handler: a context manager that sits on the global stack and may rewrite
    messages.  Also usable as a decorator / function wrapper (``handler(fn)``)."""

    def __init__(self):
        pass

    def __call__(self, fn):
        if not callable(fn):
            raise ValueError("{} is not callable, did you mean to pass it as a keyword arg?"
                             .format(fn))
        return _BoundHandler(self, fn)

    def __enter__(self):
        if self not in _PYRO_STACK:
            _PYRO_STACK.append(self)
            return self
        raise ValueError("cannot install a Messenger instance twice")

    def __exit__(self, exc_type, exc_value, traceback):
        if exc_type is None:
            if _PYRO_STACK and _PYRO_STACK[-1] is self:
                _PYRO_STACK.pop()
            else:
                raise ValueError("This Messenger is not on the top of the stack")
        else:
            # an exception unwinds through us: drop ourselves and everything installed above
            if self in _PYRO_STACK:
                loc = _PYRO_STACK.index(self)
                for _ in range(loc, len(_PYRO_STACK)):
                    _PYRO_STACK.pop()

    def _reset(self):
        pass

    def _process_message(self, msg):
        method = getattr(self, "_pyro_" + msg["type"], None)
        if method is not None:
            method(msg)

    def _postprocess_message(self, msg):
        method = getattr(self, "_pyro_post_" + msg["type"], None)
        if method is not None:
            method(msg)


class _BoundHandler:
    def __init__(self, handler, fn):
        self.handler = handler
        self.fn = fn
        functools.update_wrapper(self, fn, updated=[])

    def __call__(self, *args, **kwargs):
        with self.handler:
            return self.fn(*args, **kwargs)

    def __get__(self, instance, owner=None):
        # a handler applied to a method in a class body: bind like a function would
        # (the role of _bound_partial, pyro/poutine/messenger.py:35-56)
        if instance is None:
            return self
        return functools.partial(self, instance)

    def __getattr__(self, name):
        # what the wrapped callable offers (e.g. .get_trace of a traced function further in)
        return getattr(object.__getattribute__(self, "fn"), name)


def get_mask():
    """The mask the enclosing ``poutine.mask`` handlers apply right now: None, a bool, or a tensor
    (reference: pyro/poutine/runtime.py get_mask)."""
    return _query("get_mask")["mask"]


def get_plates():
    """The frames of the ``pyro.plate`` contexts the caller is inside of, as a tuple
    (reference: pyro/poutine/runtime.py get_plates)."""
    return tuple(_query("get_plates")["cond_indep_stack"])


def _query(kind):
    msg = new_message(kind, kind, None)
    msg["done"] = True
    msg["value"] = None
    apply_stack(msg)
    return msg


@contextlib.contextmanager
def block_messengers(predicate):
    """EXPERIMENTAL in the reference too (messenger.py:264-287): the handlers on the stack for which
    ``predicate`` holds are muted (replaced by do-nothing handlers, without exit/enter) for the
    duration of the context; yields the list of muted handlers."""
    muted = {}
    try:
        for i, handler in enumerate(_PYRO_STACK):
            if predicate(handler):
                muted[i] = handler
                _PYRO_STACK[i] = Messenger()
        yield list(muted.values())
    finally:
        for i, handler in muted.items():
            _PYRO_STACK[i] = handler
