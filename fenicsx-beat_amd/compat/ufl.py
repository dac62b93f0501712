"""Shim: the ufl names used to describe stimuli, measures and exact solutions (beat.grid's expression language)."""
import numpy as _np

from beat.grid import (  # noqa: F401
    And, Measure, Or, SpatialCoordinate, conditional, cos, ds, dx, exp, ge, gt, le, lt, sin, sqrt, variable,
)

pi = _np.pi
