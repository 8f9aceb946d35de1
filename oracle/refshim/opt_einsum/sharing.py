"""TEST INFRASTRUCTURE ONLY -- see __init__.py."""
import contextlib

_STACK = []


def current_cache():
    return _STACK[-1] if _STACK else None


@contextlib.contextmanager
def shared_intermediates(cache=None):
    if cache is None:
        cache = {}
    _STACK.append(cache)
    try:
        yield cache
    finally:
        _STACK.pop()


def count_cached_ops(cache):
    out = {}
    for key in cache:
        if isinstance(key, tuple) and key:
            out[key[0]] = out.get(key[0], 0) + 1
    return out
