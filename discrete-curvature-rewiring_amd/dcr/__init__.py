"""Host-side binding of libdcr_hip.so (see include/dcr.h) and small helpers.

Kept import-light on purpose: importing ``dcr`` does not load the HIP library;
the first ``DcrGraph`` (or ``dcr._lib.lib()``) does, and raises if it is missing.
"""
from .data import Data, Dataset  # noqa: F401


def __getattr__(name):
    if name == 'DcrGraph':
        from .graph import DcrGraph
        return DcrGraph
    raise AttributeError(name)
