"""Shim: the dolfinx names used by the reference's hot-path demos, backed by beat.grid (see ../README.md)."""
from types import SimpleNamespace

import numpy as _np

from beat import grid as _g

__version__ = "0.10.0"
default_scalar_type = _g.default_scalar_type
default_real_type = _np.float64

from . import fem, io, mesh  # noqa: E402,F401

cpp = SimpleNamespace(mesh=SimpleNamespace(CellType=_g.CellType))
