"""Importable alias of the package directory ``clearwater-riverine_amd/`` (a hyphen cannot appear in
a Python module name).  All code lives there; this file only extends ``__path__``."""
import os as _os

_impl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'clearwater-riverine_amd')
if not _os.path.isdir(_impl):
    raise ImportError(f'{_impl} is missing')
__path__.append(_impl)

from .engine import TransportEngine, SolverNotConverged, load_library, ABI_SYMBOLS, LIB_PATH  # noqa: E402,F401
from .model import ClearwaterRiverine, Mesh, Constituent  # noqa: E402,F401
from . import synthetic  # noqa: E402,F401

__version__ = '0.1.0'
