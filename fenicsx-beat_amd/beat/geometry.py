"""Slab geometries on the structured box mesh -- the public functions of src/beat/geometry.py:9-218
(``get_{2D,3D}_slab_{mesh,microstructure,geometry}``) with constant fibre / sheet / normal directions.

``dx`` is the target edge length; the number of cells per axis is ``rint(L / dx)`` as in the reference
(geometry.py:130-139)."""

from __future__ import annotations

from typing import NamedTuple

import numpy as np

from . import grid


class Geometry(NamedTuple):
    mesh: grid.Mesh
    ffun: object | None = None
    markers: dict | None = None
    f0: object | None = None
    s0: object | None = None
    n0: object | None = None


def _slab_mesh(comm, lengths, dx, cell_type):
    cells = [int(np.rint(length / dx)) for length in lengths]
    corners = [np.zeros(len(lengths)), np.asarray(lengths, dtype=np.float64)]
    make = grid.create_rectangle if len(lengths) == 2 else grid.create_box
    return make(comm, corners, cells, cell_type)


def _axes(mesh, order):
    """unit vectors e_k for k in ``order`` as mesh constants"""
    eye = np.eye(len(order))
    return tuple(grid.Constant(mesh, tuple(eye[k])) for k in order)


def get_2D_slab_mesh(comm, dx, Lx, Ly, cell_type=grid.CellType.triangle, dtype=np.float64):
    return _slab_mesh(comm, (Lx, Ly), dx, cell_type)


def get_3D_slab_mesh(comm, dx, Lx, Ly, Lz, cell_type=grid.CellType.tetrahedron, dtype=np.float64):
    return _slab_mesh(comm, (Lx, Ly, Lz), dx, cell_type)


def get_2D_slab_microstructure(mesh, transverse: bool = False):
    """(f0, s0): fibres along x (or along y when ``transverse``)"""
    return _axes(mesh, (1, 0) if transverse else (0, 1))


def get_3D_slab_microstructure(mesh, transverse: bool = False):
    """(f0, s0, n0): fibres along x, sheets along y, normal z (``transverse``: z, x, y)"""
    return _axes(mesh, (2, 0, 1) if transverse else (0, 1, 2))


def get_2D_slab_geometry(comm, dx, Lx, Ly, cell_type=grid.CellType.triangle, dtype=np.float64,
                         transverse: bool = False) -> Geometry:
    mesh = get_2D_slab_mesh(comm, dx, Lx, Ly, cell_type, dtype)
    f0, s0 = get_2D_slab_microstructure(mesh, transverse)
    return Geometry(mesh=mesh, f0=f0, s0=s0)


def get_3D_slab_geometry(comm, dx, Lx, Ly, Lz, cell_type=grid.CellType.tetrahedron, dtype=np.float64,
                         transverse: bool = False) -> Geometry:
    mesh = get_3D_slab_mesh(comm, dx, Lx, Ly, Lz, cell_type, dtype)
    f0, s0, n0 = get_3D_slab_microstructure(mesh, transverse)
    return Geometry(mesh=mesh, f0=f0, s0=s0, n0=n0)


# ---- voxelisation of unstructured geometries -------------------------------------------------------
class VoxelGeometry(NamedTuple):
    mesh: grid.Mesh
    mask: np.ndarray                 # bool (cz, cy, cx): voxel centre inside the tetrahedral mesh
    cell_data: dict                  # name -> per-voxel array (nvoxels, ...), values of the containing tetrahedron
    tet_index: np.ndarray            # (nvoxels,) index of the containing tetrahedron, -1 outside


def voxelize_tetrahedra(comm, points, tets, h: float, cell_data: dict | None = None, padding: int = 1) -> VoxelGeometry:
    """Rasterise a tetrahedral mesh (``points`` (np, 3), ``tets`` (nt, 4) vertex ids -- e.g. a ventricular geometry
    with its per-cell fibre vectors) onto a box of cubic voxels of edge ``h``: a voxel is tissue when its centre lies
    in some tetrahedron, and inherits that tetrahedron's ``cell_data`` (fibres, markers, ...).  The box is the
    bounding box of the points plus ``padding`` voxels on every side.  Returns the voxel mesh
    (``grid.create_voxel_mesh``) together with the per-voxel data, ready for ``grid.CellField`` /
    ``define_conductivity_tensor``."""
    points = np.asarray(points, dtype=np.float64)
    tets = np.asarray(tets, dtype=np.int64)
    lo = points.min(axis=0) - padding * h
    n = np.maximum(1, np.ceil((points.max(axis=0) + padding * h - lo) / h).astype(np.int64))
    owner = np.full(tuple(n[::-1]), -1, dtype=np.int64)  # (cz, cy, cx)
    X = points[tets]                                       # (nt, 4, 3)
    # barycentric coordinates: lambda = T^-1 (x - x3), T columns x0-x3, x1-x3, x2-x3
    T = np.transpose(X[:, :3, :] - X[:, 3:4, :], (0, 2, 1))
    Tinv = np.linalg.inv(T)
    first = np.maximum(0, np.floor((X.min(axis=1) - lo) / h - 0.5).astype(np.int64))
    last = np.minimum(n - 1, np.ceil((X.max(axis=1) - lo) / h - 0.5).astype(np.int64))
    eps = 1e-12
    for t in range(len(tets)):
        a, b = first[t], last[t]
        if (b < a).any():
            continue
        ax = [lo[k] + (np.arange(a[k], b[k] + 1) + 0.5) * h for k in range(3)]
        Z, Y, Xc = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        d = np.stack([Xc, Y, Z], axis=-1) - X[t, 3]
        lam = d @ Tinv[t].T
        inside = (lam >= -eps).all(axis=-1) & (lam.sum(axis=-1) <= 1.0 + eps)
        block = owner[a[2] : b[2] + 1, a[1] : b[1] + 1, a[0] : b[0] + 1]
        block[inside & (block < 0)] = t
    mask = owner >= 0
    mesh = grid.create_voxel_mesh(comm, mask, h, origin=tuple(lo))
    flat = owner.ravel()
    data = {}
    for name, values in (cell_data or {}).items():
        values = np.asarray(values)
        out = np.zeros((flat.size,) + values.shape[1:], dtype=values.dtype)
        out[flat >= 0] = values[flat[flat >= 0]]
        data[name] = out
    return VoxelGeometry(mesh=mesh, mask=mask, cell_data=data, tet_index=flat)
