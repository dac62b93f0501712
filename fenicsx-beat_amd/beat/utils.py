"""Function-space helpers on the hot path (interface of src/beat/utils.py:26-112)."""

from __future__ import annotations

from . import grid


def local_project(v: grid.Function, V: grid.FunctionSpace, u: grid.Function | None = None):
    """ODE-space -> PDE-space transfer.  Both are P1 on the same mesh here, i.e. the identity-copy
    branch of utils.py:52-54 (a device-to-device copy)."""
    U = grid.Function(V) if u is None else u
    if v.x.array.size != U.x.array.size:
        raise NotImplementedError("projection between different spaces is not implemented on the HIP backend")
    U.x.array[:] = v.x.array
    return U


def space_from_string(space_string: str, mesh: grid.Mesh, dim: int = 1) -> grid.FunctionSpace:
    family, degree = space_string.split("_")
    if dim != 1:
        raise NotImplementedError("vector spaces are not implemented")
    return grid.FunctionSpace(mesh, family, int(degree))


def interpolation_points(V):
    return None
