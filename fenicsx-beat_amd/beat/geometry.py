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
