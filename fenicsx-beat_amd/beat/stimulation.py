"""Stimulus definition (interface of src/beat/stimulation.py:14-272).

A stimulus contributes ``dt * I_s(t) * int_{dz(marker)} phi_i`` to the right-hand side
(src/beat/base_model.py:247-248).  ``expr`` is an expression of the time ``Constant`` (and
possibly of the coordinate); the model splits it into an amplitude evaluated on the host each step
and a nodal weight field that lives on the device."""

from __future__ import annotations

import logging
from typing import NamedTuple

import numpy as np

from . import grid
from .units import ureg

logger = logging.getLogger(__name__)


class Stimulus(NamedTuple):
    expr: grid.Expr
    dZ: grid.Measure
    marker: int | None = None

    @property
    def dz(self):
        return self.dZ(self.marker)

    def assign(self, amp: float):
        self.expr.amplitude = amp


def compute_effective_dim(mesh, subdomain_data) -> int:
    dim = subdomain_data.dim
    if mesh.topology.dim == 3:
        return dim
    elif mesh.topology.dim == 2:
        return dim + 1
    elif mesh.topology.dim == 1:
        return dim + 2
    raise ValueError("Invalid mesh topology dimension")


def get_dZ(mesh, subdomain_data) -> grid.Measure:
    dim = subdomain_data.dim
    if dim == mesh.topology.dim - 1:
        if mesh.topology.dim <= 1:
            raise ValueError("Invalid mesh topology dimension")
        return grid.Measure("ds", domain=mesh, subdomain_data=subdomain_data)
    elif dim == mesh.topology.dim:
        return grid.Measure("dx", domain=mesh, subdomain_data=subdomain_data)
    raise ValueError("Invalid subdomain data dimension")


def convert_amplitude(effective_dim: int, amplitude):
    if isinstance(amplitude, ureg.Quantity):
        return amplitude
    if effective_dim <= 1:
        unit = ureg("uA / cm")
    elif effective_dim == 2:
        unit = ureg("uA / cm**2")
    elif effective_dim == 3:
        unit = ureg("uA / cm**3")
    else:
        raise ValueError(f"Invalid effective dimension {effective_dim}. Must be 0, 1, 2 or 3.")
    return amplitude * unit


def compute_stimulus_unit(effective_dim: int, mesh_unit: str):
    if effective_dim < 0:
        raise ValueError("Effective dimension must be non-negative")
    if effective_dim > 3:
        raise ValueError("Effective dimension must be less than or equal to 3")
    if effective_dim == 0:
        return ureg("uA")
    return ureg(f"uA/{mesh_unit}**{effective_dim - 1}")


def convert_chi(chi, mesh_unit: str):
    if isinstance(chi, ureg.Quantity):
        return chi
    return chi * ureg(f"{mesh_unit}**-1")


def define_stimulus(mesh, chi, time, subdomain_data, marker: int, mesh_unit: str = "cm", duration: float = 2.0,
                    amplitude: float = 500.0, start: float = 0.0) -> Stimulus:
    """``I_s(t) = (amplitude/chi)`` in ``uA/mesh_unit^(d-1)`` for ``start <= t <= start+duration``,
    else 0, on the cells of ``subdomain_data`` tagged ``marker`` (stimulation.py:210-272)."""
    effective_dim = compute_effective_dim(mesh, subdomain_data)
    chi = convert_chi(chi, mesh_unit)
    A = convert_amplitude(effective_dim, amplitude)
    dZ = get_dZ(mesh, subdomain_data)
    unit = compute_stimulus_unit(effective_dim, mesh_unit)
    amp = (A / chi).to(unit).magnitude
    I_s = grid.conditional(grid.And(grid.ge(time, start), grid.le(time, start + duration)), amp, 0.0)
    return Stimulus(dZ=dZ, marker=marker, expr=I_s)


# ---- host-side assembly of the nodal weights  w_i = int_{cells} f(x) phi_i -----------------------
def _gauss_simplex(d: int, m: int = 5):
    gx, gw = np.polynomial.legendre.leggauss(m)
    gx, gw = 0.5 * (gx + 1.0), 0.5 * gw
    if d == 1:
        lam, w = np.stack([1 - gx, gx], axis=1), gw
    elif d == 2:
        a, b = np.meshgrid(gx, gx, indexing="ij")
        wa, wb = np.meshgrid(gw, gw, indexing="ij")
        l1, l2 = a, (1 - a) * b
        lam, w = np.stack([1 - l1 - l2, l1, l2], axis=-1).reshape(-1, 3), (wa * wb * (1 - a)).ravel()
    else:
        a, b, c = np.meshgrid(gx, gx, gx, indexing="ij")
        wa, wb, wc = np.meshgrid(gw, gw, gw, indexing="ij")
        l1, l2, l3 = a, (1 - a) * b, (1 - a) * (1 - b) * c
        lam = np.stack([1 - l1 - l2 - l3, l1, l2, l3], axis=-1).reshape(-1, 4)
        w = (wa * wb * wc * (1 - a) ** 2 * (1 - b)).ravel()
    return lam, w / w.sum()


def assemble_facet_weights(mesh, facets) -> np.ndarray:
    """Nodal weights  w_i = int_{facets} phi_i dS  on the LOCAL slab for exterior facets (stimulation.py:63-111,
    the ``ds`` branch): an edge gives L/2 to both ends; a box-cell face is two triangles split along the diagonal
    from its lowest to its highest corner, which therefore get A/3 each and the other two corners A/6."""
    facets = np.asarray(facets, dtype=np.int64)
    verts = mesh.facet_vertices(facets)
    area = mesh.facet_area(facets)
    if mesh.dim == 2:
        share = np.array([0.5, 0.5])
    elif mesh.dim == 3:
        share = np.array([1.0 / 3.0, 1.0 / 6.0, 1.0 / 6.0, 1.0 / 3.0])
    else:
        raise ValueError("facet integrals need a 2-D or 3-D mesh")
    w = np.zeros(mesh.num_nodes)
    lo, hi = mesh.slab.z0 * mesh.plane, mesh.slab.z1 * mesh.plane
    v = verts.ravel()
    c = (area[:, None] * share[None, :]).ravel()
    sel = (v >= lo) & (v < hi)
    np.add.at(w, v[sel] - lo, c[sel])
    return w


def assemble_weights(mesh, cells, spatial: grid.Expr | None, chunk: int = 1 << 18) -> np.ndarray:
    """Nodal weights on the LOCAL slab (length mesh.num_nodes).  ``cells`` = global cell ids or None
    (whole mesh); ``spatial`` = coordinate-dependent factor or None (= 1)."""
    d = mesh.dim
    vol = float(np.prod(mesh.h)) / {1: 1, 2: 2, 3: 6}[d]
    nx, ny, nzg = mesh.shape_global
    w_global_slab = np.zeros(mesh.num_nodes)
    lo, hi = mesh.slab.z0 * mesh.plane, mesh.slab.z1 * mesh.plane
    if cells is None:
        cells = mesh.all_cells()
    cells = np.asarray(cells, dtype=np.int64)
    lam = wq = None
    if spatial is not None:
        lam, wq = _gauss_simplex(d)
    for s in range(0, len(cells), chunk):
        verts = mesh.cell_vertices(cells[s : s + chunk])  # global ids
        if spatial is None:
            contrib = np.full(verts.shape, vol / (d + 1))
        else:
            X = grid._node_xyz(mesh, verts.ravel()).reshape(verts.shape + (3,))
            xq = np.einsum("qa,cad->dcq", lam, X)  # (3, nc, nq)
            fq = np.broadcast_to(np.asarray(spatial.evaluate(xq), dtype=np.float64), xq.shape[1:])
            contrib = vol * np.einsum("cq,q,qa->ca", fq, wq, lam)
        v, c = verts.ravel(), contrib.ravel()
        sel = (v >= lo) & (v < hi)
        np.add.at(w_global_slab, v[sel] - lo, c[sel])
    return w_global_slab
