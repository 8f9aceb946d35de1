"""Validation switches of the handler layer."""
_VALIDATE = __debug__


def enable_validation(is_validate=True):
    global _VALIDATE
    _VALIDATE = bool(is_validate)


def validation_enabled():
    return _VALIDATE
