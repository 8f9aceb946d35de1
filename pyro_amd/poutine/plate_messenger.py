"""pyro.poutine.plate_messenger: the plate handler under its reference module name, and
``block_plate`` (reference: plate_messenger.py:27-90)."""
import contextlib

from .handlers import PlateMessenger  # noqa: F401
from .runtime import block_messengers


@contextlib.contextmanager
def block_plate(name=None, dim=None, *, strict=True):
    """Temporarily leave ONE enclosing plate, named by ``name`` or by ``dim`` (exactly one of the two):
    sites sampled inside are neither broadcast along that plate nor scaled by it -- for a global
    variable that has to be drawn from inside a plate.  ``strict`` (default) raises ValueError unless
    exactly one enclosing plate matches."""
    if (name is not None) == (dim is not None):
        raise ValueError("Exactly one of name,dim must be specified")
    assert name is None or isinstance(name, str)
    assert dim is None or (isinstance(dim, int) and dim < 0)

    def matches(handler):
        if not isinstance(handler, PlateMessenger):
            return False
        return handler.name == name if name is not None else handler.dim == dim

    with block_messengers(matches) as found:
        if strict and len(found) != 1:
            raise ValueError("block_plate matched {} messengers. Try either removing the block_plate "
                             "or setting strict=False.".format(len(found)))
        yield
