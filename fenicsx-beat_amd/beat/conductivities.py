"""Conductivity helpers (interface of src/beat/conductivities.py:29-118).  With a constant fibre
direction the tensor ``M = s_l f(x)f + s_t (I - f(x)f)`` is a plain (dim, dim) NumPy matrix; with a
fibre field given per cell (grid.CellField) it is a CellField of (dim, dim) tensors."""

from __future__ import annotations

import logging
from typing import NamedTuple

import numpy as np

from .units import to_quantity, ureg

logger = logging.getLogger(__name__)


def default_conductivities(name="Niederer") -> dict:
    if name == "Niederer":
        return {"g_il": 0.17 * ureg("S/m"), "g_it": 0.019 * ureg("S/m"), "g_el": 0.62 * ureg("S/m"),
                "g_et": 0.24 * ureg("S/m"), "chi": 1400.0 * ureg("cm**-1")}
    elif name == "Bishop":
        return {"g_il": 0.34 * ureg("S/m"), "g_it": 0.060 * ureg("S/m"), "g_el": 0.12 * ureg("S/m"),
                "g_et": 0.08 * ureg("S/m"), "chi": 1400.0 * ureg("cm**-1")}
    elif name == "Potse":
        return {"g_il": 3.0 * ureg("mS/cm"), "g_it": 0.3 * ureg("mS/cm"), "g_el": 3.0 * ureg("mS/cm"),
                "g_et": 1.2 * ureg("mS/cm"), "chi": 800.0 * ureg("cm**-1")}
    raise ValueError(f"Unknown conductivity tensor {name}")


class Conductivities(NamedTuple):
    s_l: float
    s_t: float


def get_harmonic_mean_conductivity(chi, g_il=0.17, g_it=0.019, g_el=0.62, g_et=0.24) -> Conductivities:
    sigma_il, sigma_it = to_quantity(g_il, "S/m"), to_quantity(g_it, "S/m")
    sigma_el, sigma_et = to_quantity(g_el, "S/m"), to_quantity(g_et, "S/m")

    def harmonic_mean(a, b):
        return a * b / (a + b)

    sigma_l = harmonic_mean(sigma_il, sigma_el)
    sigma_t = harmonic_mean(sigma_it, sigma_et)
    if not isinstance(chi, ureg.Quantity):
        chi = chi * ureg("cm**-1")
    s_l = (sigma_l / chi).to("uA/mV").magnitude
    s_t = (sigma_t / chi).to("uA/mV").magnitude
    return Conductivities(s_l, s_t)


def conductivity_tensor(s_l: float, s_t: float, f0) -> np.ndarray:
    from .grid import CellField, Constant, Function

    if isinstance(f0, Function):
        raise NotImplementedError("nodal fibre fields are not implemented: pass the fibres per cell (grid.CellField)")
    if isinstance(f0, CellField):
        f = f0.values
        dim = f.shape[1]
        ff = f[:, :, None] * f[:, None, :]
        return CellField(f0.mesh, s_t * np.eye(dim)[None] + (s_l - s_t) * ff)
    f = np.asarray(f0.value if isinstance(f0, Constant) else f0, dtype=np.float64)
    dim = len(f)
    return s_l * np.outer(f, f) + s_t * (np.eye(dim) - np.outer(f, f))


def define_conductivity_tensor(chi, f0, g_il=0.17, g_it=0.019, g_el=0.62, g_et=0.24) -> np.ndarray:
    if f0 is None:
        raise ValueError("f0 must be provided")
    s_l, s_t = get_harmonic_mean_conductivity(chi, g_il, g_it, g_el, g_et)
    return conductivity_tensor(s_l, s_t, f0)
