"""Global parameter store (reference: pyro/params/param_store.py:138-336): named, optionally
constrained parameters backed by unconstrained leaf tensors that the optimizers update."""
import re
import weakref

import torch
from torch.distributions import constraints, transform_to


class ParamStoreDict:
    def __init__(self):
        self._params = {}        # name -> unconstrained leaf tensor
        self._param_to_name = {}  # unconstrained leaf -> name
        self._constraints = {}
        # bumped whenever the SET of leaf tensors changes (a parameter created, replaced, deleted, the store
        # cleared, a scope entered or left): a captured SVI step holds the leaves it was captured with by
        # address, so it is dropped when this moves (infer/svi.py)
        self.generation = 0

    def clear(self):
        self._params, self._param_to_name, self._constraints = {}, {}, {}
        self.generation += 1

    def items(self):
        for name in self._params:
            yield name, self[name]

    def keys(self):
        return self._params.keys()

    def values(self):
        for name in self._params:
            yield self[name]

    def __bool__(self):
        return bool(self._params)

    def __len__(self):
        return len(self._params)

    def __contains__(self, name):
        return name in self._params

    def __iter__(self):
        return iter(self.keys())

    def __delitem__(self, name):
        unconstrained = self._params.pop(name)
        self._param_to_name.pop(unconstrained)
        self._constraints.pop(name)
        self.generation += 1

    def __getitem__(self, name):
        unconstrained = self._params[name]
        constraint = self._constraints[name]
        constrained = None
        if constraint is not constraints.real and unconstrained.dtype in (torch.float32, torch.float64):
            from . import kernels
            if kernels.on_device(unconstrained) and unconstrained.is_contiguous():
                from .distributions import fused
                lower = fused.exp_lower_bound_of(transform_to(constraint))
                if lower is not None:
                    # positive / greater_than: value = lower + exp(u) in one launch each way instead of
                    # exp, mul, add and their duals per access
                    constrained = fused.exp_lower(unconstrained, lower)
        if constrained is None:
            constrained = transform_to(constraint)(unconstrained)
        if constrained is unconstrained:
            constrained = unconstrained.view_as(unconstrained) if False else unconstrained
        try:
            constrained.unconstrained = weakref.ref(unconstrained)
        except AttributeError:
            pass
        return constrained

    def __setitem__(self, name, new_constrained_value):
        constraint = self._constraints.get(name, constraints.real)
        self.setdefault_or_replace(name, new_constrained_value, constraint)

    def setdefault_or_replace(self, name, value, constraint):
        if constraint is constraints.real and isinstance(value, torch.nn.Parameter):
            unconstrained = value  # nn.Module parameters are stored by identity (pyro.module)
        else:
            with torch.no_grad():
                unconstrained = transform_to(constraint).inv(value.detach()).clone().contiguous()
            unconstrained.requires_grad_(True)
        if name in self._params:
            self._param_to_name.pop(self._params[name])
        self._params[name] = unconstrained
        self._param_to_name[unconstrained] = name
        self._constraints[name] = constraint
        self.generation += 1

    def setdefault(self, name, init_constrained_value, constraint=constraints.real):
        if name not in self._params:
            if callable(init_constrained_value):
                init_constrained_value = init_constrained_value()
            self.setdefault_or_replace(name, init_constrained_value, constraint)
        return self[name]

    def get_param(self, name, init_tensor=None, constraint=constraints.real, event_dim=None):
        if init_tensor is None:
            if name not in self._params:
                raise KeyError("param '{}' is not in the param store and no init was given".format(
                    name))
            return self[name]
        return self.setdefault(name, init_tensor, constraint)

    def param_name(self, p):
        return self._param_to_name.get(p)

    def named_parameters(self):
        return self._params.items()

    def get_all_param_names(self):
        return set(self._params.keys())

    def match(self, name):
        pattern = re.compile(name)
        return {n: self[n] for n in self._params if pattern.match(n)}

    def get_state(self):
        return {"params": {n: p.detach().clone() for n, p in self._params.items()},
                "constraints": dict(self._constraints)}

    def set_state(self, state):
        self.clear()
        for name, unconstrained in state["params"].items():
            constraint = state["constraints"][name]
            u = unconstrained.detach().clone().requires_grad_(True)
            self._params[name] = u
            self._param_to_name[u] = name
            self._constraints[name] = constraint
        self.generation += 1

    def scope(self, state=None):
        """Context manager for several parameter stores in one process (param_store.py:337-372):
        inside, the store holds ``state`` (empty by default); on exit the previous contents come
        back and the scope's own are written into the yielded state dict, which can be entered
        again."""
        import contextlib

        @contextlib.contextmanager
        def _scope():
            inner = {"params": {}, "constraints": {}} if state is None else state
            outer = self._snapshot()
            try:
                self.clear()
                self._restore(inner)
                yield inner
                inner.update(self._snapshot())
            finally:
                self.clear()
                self._restore(outer)
        return _scope()

    def _snapshot(self):
        """The store's tensors themselves (no copies: a scope hands the same leaves back)."""
        return {"params": dict(self._params), "constraints": dict(self._constraints)}

    def _restore(self, snap):
        for name, u in snap["params"].items():
            self._params[name] = u
            self._param_to_name[u] = name
            self._constraints[name] = snap["constraints"][name]
        self.generation += 1

    def save(self, filename):
        torch.save(self.get_state(), filename)

    def load(self, filename, map_location=None):
        self.set_state(torch.load(filename, map_location=map_location, weights_only=False))


_MODULE_NAMESPACE_DIVIDER = "$$$"


def param_with_module_name(pyro_name, param_name):
    return _MODULE_NAMESPACE_DIVIDER.join([pyro_name, param_name])


def module_from_param_with_module_name(param_name):
    return param_name.split(_MODULE_NAMESPACE_DIVIDER)[0]


def user_param_name(param_name):
    """The name the user gave: what follows ``<module>$$$`` (param_store.py:388-391)."""
    if _MODULE_NAMESPACE_DIVIDER in param_name:
        return param_name.split(_MODULE_NAMESPACE_DIVIDER)[1]
    return param_name


def normalize_param_name(name):
    return name.replace(_MODULE_NAMESPACE_DIVIDER, ".")


_PARAM_STORE = ParamStoreDict()
