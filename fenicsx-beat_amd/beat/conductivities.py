"""Conductivity helpers: the public functions of the reference's module (src/beat/conductivities.py:29-118) on the
structured-grid data model.

The monodomain tensor is ``M = s_l f f^T + s_t (I - f f^T)`` with ``s = (g_i g_e / (g_i + g_e)) / chi`` in uA/mV
(:63-104).  A constant fibre direction gives a plain (dim, dim) matrix, a per-cell fibre field (``grid.CellField``)
a per-cell tensor field that ``MonodomainModel`` turns into per-node stencil rows.
"""

from __future__ import annotations

from typing import NamedTuple

import numpy as np

from .units import to_quantity, ureg

# published parameter sets: (unit of the conductivities, g_il, g_it, g_el, g_et, chi in 1/cm)
_PRESETS = {
    "Niederer": ("S/m", 0.17, 0.019, 0.62, 0.24, 1400.0),
    "Bishop": ("S/m", 0.34, 0.060, 0.12, 0.08, 1400.0),
    "Potse": ("mS/cm", 3.0, 0.3, 3.0, 1.2, 800.0),
}
_KEYS = ("g_il", "g_it", "g_el", "g_et")


def default_conductivities(name="Niederer") -> dict:
    """Intra-/extracellular conductivities along (l) and across (t) the fibres and the surface-to-volume ratio of
    a named parameter set, as quantities with units."""
    try:
        unit, *values, chi = _PRESETS[name]
    except KeyError:
        raise ValueError(f"Unknown conductivity tensor {name}") from None
    out = {key: value * ureg(unit) for key, value in zip(_KEYS, values)}
    out["chi"] = chi * ureg("cm**-1")
    return out


class Conductivities(NamedTuple):
    s_l: float
    s_t: float


def _series(g_i, g_e):
    """two conductors in series (harmonic mean of intra- and extracellular conductivity), in S/m"""
    a, b = to_quantity(g_i, "S/m"), to_quantity(g_e, "S/m")
    return a * b / (a + b)


def get_harmonic_mean_conductivity(chi, g_il=0.17, g_it=0.019, g_el=0.62, g_et=0.24) -> Conductivities:
    """(s_l, s_t) in uA/mV: the monodomain conductivities divided by the surface-to-volume ratio ``chi``
    (a bare number is taken as 1/cm)."""
    chi_q = chi if isinstance(chi, ureg.Quantity) else chi * ureg("cm**-1")
    return Conductivities(*((_series(gi, ge) / chi_q).to("uA/mV").magnitude for gi, ge in ((g_il, g_el), (g_it, g_et))))


def conductivity_tensor(s_l: float, s_t: float, f0):
    """``s_l f f^T + s_t (I - f f^T)`` for a constant direction (vector / ``grid.Constant``) or a per-cell field."""
    from . import grid

    if isinstance(f0, grid.VectorFunction):
        # fibre direction as a vector P1 function (geo.f0 of the reference's ventricular geometries): inside a
        # simplex f = sum_a lambda_a f_a, so M(x) is quadratic and the stiffness integrals (constant gradients) see
        # exactly its cell average   s_t I + (s_l - s_t) / ((d+1)(d+2)) sum_ab (1 + delta_ab) f_a f_b^T
        mesh = f0.function_space.mesh
        d = mesh.dim
        if f0.function_space.value_size != d:
            raise ValueError(f"fibre field has {f0.function_space.value_size} components on a {d}-D mesh")
        verts = mesh.cell_vertices(np.arange(mesh.num_cells_global, dtype=np.int64))  # (ncells, d+1) global node ids
        F = f0.values[verts]                                       # (ncells, d+1, d)
        tot = F.sum(axis=1)
        ff = (np.einsum("ci,cj->cij", tot, tot) + np.einsum("cai,caj->cij", F, F)) / ((d + 1) * (d + 2))
        return grid.CellField(mesh, s_t * np.eye(d)[None] + (s_l - s_t) * ff)
    if isinstance(f0, grid.Function):
        raise NotImplementedError("scalar functions are not fibre fields: use a vector P1 function or a grid.CellField")
    if isinstance(f0, grid.CellField):
        fibres = f0.values
        eye = np.eye(fibres.shape[1])
        return grid.CellField(f0.mesh, s_t * eye[None] + (s_l - s_t) * np.einsum("ci,cj->cij", fibres, fibres))
    direction = np.asarray(f0.value if isinstance(f0, grid.Constant) else f0, dtype=np.float64)
    ff = np.outer(direction, direction)
    return s_t * np.eye(len(direction)) + (s_l - s_t) * ff


def define_conductivity_tensor(chi, f0, g_il=0.17, g_it=0.019, g_el=0.62, g_et=0.24):
    if f0 is None:
        raise ValueError("f0 must be provided")
    return conductivity_tensor(*get_harmonic_mean_conductivity(chi, g_il, g_it, g_el, g_et), f0)
