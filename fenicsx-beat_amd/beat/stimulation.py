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


def assemble_facet_weights(mesh, facets, spatial: grid.Expr | None = None) -> np.ndarray:
    """Nodal weights  w_i = int_{facets} f phi_i dS  on the LOCAL slab for exterior facets (stimulation.py:63-111,
    the ``ds`` branch; ``f`` = the coordinate-dependent factor ``spatial`` of the stimulus expression or 1): an edge
    gives L/2 to both ends; a box-cell face is two triangles split along the diagonal from its lowest to its highest
    corner, which therefore get A/3 each and the other two corners A/6.  With a spatial factor the integrals are
    taken by Gauss quadrature on every edge / triangle (as UFL does for a non-constant integrand)."""
    facets = np.asarray(facets, dtype=np.int64)
    verts = mesh.facet_vertices(facets)
    area = mesh.facet_area(facets)
    if mesh.dim not in (2, 3):
        raise ValueError("facet integrals need a 2-D or 3-D mesh")
    w = np.zeros(mesh.num_nodes)
    lo, hi = mesh.slab.z0 * mesh.plane, mesh.slab.z1 * mesh.plane
    if spatial is None:
        share = np.array([0.5, 0.5]) if mesh.dim == 2 else np.array([1.0 / 3.0, 1.0 / 6.0, 1.0 / 6.0, 1.0 / 3.0])
        v = verts.ravel()
        c = (area[:, None] * share[None, :]).ravel()
        sel = (v >= lo) & (v < hi)
        np.add.at(w, v[sel] - lo, c[sel])
        return w
    # simplices of the facets: the edge itself, or the two triangles of a face (each half its area)
    if mesh.dim == 2:
        simplices, meas = [verts], [area]
    else:
        simplices, meas = [verts[:, [0, 1, 3]], verts[:, [0, 2, 3]]], [0.5 * area, 0.5 * area]
    lam, wq = _gauss_simplex(mesh.dim - 1)
    for sv, m in zip(simplices, meas):
        X = grid._node_xyz(mesh, sv.ravel()).reshape(sv.shape + (3,))
        xq = np.einsum("qa,cad->dcq", lam, X)  # (3, nfacets, nq)
        fq = np.broadcast_to(np.asarray(spatial.evaluate(xq), dtype=np.float64), xq.shape[1:])
        contrib = m[:, None] * np.einsum("cq,q,qa->ca", fq, wq, lam)
        v, c = sv.ravel(), contrib.ravel()
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


# ---- spatio-temporal activation patterns (stimulation.py:275-363) -----------------------------------------------
def near(a, b, tol: float = 1e-12):
    return grid.And(grid.ge(a, b - tol), grid.le(a, b + tol))


class LocalActivation(grid.Expr):
    """sum_i amplitude * [ |x - p_i|_inf <= tol ] * [ start + delay_i <= t <= start + duration + delay_i ]:
    the expression generate_random_activation builds from conditionals (stimulation.py:331-349), kept in closed
    form so that it can be evaluated on the cells near the currently active points only."""

    def __init__(self, mesh, time, points, delays, stim_start, stim_duration, stim_amplitude, tol):
        pts = np.atleast_2d(np.asarray(points, dtype=np.float64))
        self.points = np.zeros((len(pts), 3))
        self.points[:, : pts.shape[1]] = pts[:, :3]
        self.delays = np.asarray(delays, dtype=np.float64)
        self.time, self.start, self.duration = time, float(stim_start), float(stim_duration)
        self.amplitude, self.tol, self.mesh = float(stim_amplitude), float(tol), mesh

    def depends_on_x(self):
        return True

    def depends_on_constants(self):
        return True

    def active_points(self) -> np.ndarray:
        t = float(self.time)
        return np.nonzero((t >= self.start + self.delays) & (t <= self.start + self.duration + self.delays))[0]

    def evaluate(self, x=None):
        x = np.asarray(x, dtype=np.float64)
        out = np.zeros(x.shape[1:])
        d = self.mesh.dim
        for i in self.active_points():
            hit = np.ones(x.shape[1:], dtype=bool)
            for a in range(d):
                hit &= np.abs(x[a] - self.points[i, a]) <= self.tol
            out += self.amplitude * hit
        return out

    def evaluate_cells(self, mesh) -> np.ndarray:
        """Values at the centroids of all simplices (0 on cells away from the active points and outside the
        tissue); only box cells within ``tol`` of an active point are visited."""
        spc = mesh.simplices_per_cell
        out = np.zeros(mesh.num_box_cells * spc)
        d = mesh.dim
        lower, h, n = np.array(mesh.lower), np.array(mesh.h), np.array(mesh.n)
        for i in self.active_points():
            p = self.points[i, :d]
            lo = np.clip(np.floor((p - self.tol - lower) / h).astype(np.int64), 0, n - 1)
            hi = np.clip(np.floor((p + self.tol - lower) / h).astype(np.int64), 0, n - 1)
            rng = [np.arange(lo[a], hi[a] + 1) for a in range(d)]
            gridc = np.meshgrid(*reversed(rng), indexing="ij")
            idx = [g.ravel() for g in reversed(gridc)]  # ix, iy, iz
            box = idx[0] + (n[0] * (idx[1] + (n[1] * idx[2] if d == 3 else 0)) if d >= 2 else 0)
            cells = (box[:, None] * spc + np.arange(spc)[None, :]).ravel()
            if mesh.active is not None:
                cells = cells[mesh.active[cells]]
            mid = grid.cell_midpoints(mesh, cells)
            hit = (np.abs(mid[:, :d] - p[None, :]) <= self.tol).all(axis=1)
            np.add.at(out, cells[hit], self.amplitude)
        return out


def generate_random_activation(mesh, time, points, delays, stim_start: float = 0.0, stim_duration: float = 2.0,
                               stim_amplitude: float = 1.0, tol: float = 1e-12) -> grid.Expr:
    """Spatio-temporal activation pattern: every point fires ``stim_amplitude`` inside the cube of half-width
    ``tol`` around it for ``stim_duration`` after ``stim_start + delay`` (stimulation.py:279-363)."""
    assert len(points) == len(delays), "Points and delays must have the same length"
    if len(points) == 0:
        return grid.zero()
    return LocalActivation(mesh, time, points, delays, stim_start, stim_duration, stim_amplitude, tol)
