"""Slab geometries (interface of src/beat/geometry.py:9-218) on the structured box mesh."""

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


def get_2D_slab_microstructure(mesh, transverse: bool = False):
    if transverse:
        return grid.Constant(mesh, (0.0, 1.0)), grid.Constant(mesh, (1.0, 0.0))
    return grid.Constant(mesh, (1.0, 0.0)), grid.Constant(mesh, (0.0, 1.0))


def get_3D_slab_microstructure(mesh, transverse: bool = False):
    if transverse:
        return (grid.Constant(mesh, (0.0, 0.0, 1.0)), grid.Constant(mesh, (1.0, 0.0, 0.0)),
                grid.Constant(mesh, (0.0, 1.0, 0.0)))
    return (grid.Constant(mesh, (1.0, 0.0, 0.0)), grid.Constant(mesh, (0.0, 1.0, 0.0)),
            grid.Constant(mesh, (0.0, 0.0, 1.0)))


def get_2D_slab_mesh(comm, dx, Lx, Ly, cell_type=grid.CellType.triangle, dtype=np.float64):
    nx, ny = int(np.rint(Lx / dx)), int(np.rint(Ly / dx))
    return grid.create_rectangle(comm, [np.array([0.0, 0.0]), np.array([Lx, Ly])], [nx, ny], cell_type)


def get_3D_slab_mesh(comm, dx, Lx, Ly, Lz, cell_type=grid.CellType.tetrahedron, dtype=np.float64):
    nx, ny, nz = int(np.rint(Lx / dx)), int(np.rint(Ly / dx)), int(np.rint(Lz / dx))
    return grid.create_box(comm, [np.array([0.0, 0.0, 0.0]), np.array([Lx, Ly, Lz])], [nx, ny, nz], cell_type)


def get_3D_slab_geometry(comm, dx, Lx, Ly, Lz, cell_type=grid.CellType.tetrahedron, dtype=np.float64,
                         transverse: bool = False) -> Geometry:
    mesh = get_3D_slab_mesh(comm, dx, Lx, Ly, Lz, cell_type, dtype)
    f0, s0, n0 = get_3D_slab_microstructure(mesh, transverse)
    return Geometry(mesh=mesh, f0=f0, s0=s0, n0=n0)


def get_2D_slab_geometry(comm, dx, Lx, Ly, cell_type=grid.CellType.triangle, dtype=np.float64,
                         transverse: bool = False) -> Geometry:
    mesh = get_2D_slab_mesh(comm, dx, Lx, Ly, cell_type, dtype)
    f0, s0 = get_2D_slab_microstructure(mesh, transverse)
    return Geometry(mesh=mesh, f0=f0, s0=s0)
