"""Function-space helpers on the hot path (interface of src/beat/utils.py:26-112) and the transmural
layer markers of utils.py:115-355 (Laplace solves on the device, reusing the diffusion PCG)."""

from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _hip, grid

logger = logging.getLogger(__name__)


def _interp_maps(V):
    """device copies of the interpolation maps of a P2 / DG1 space (cached on the space)"""
    from ._device import Context

    if getattr(V, "_dev_maps", None) is None:
        ctx = Context.default()
        (idx, w), to_p1 = V.layout()
        pick = np.stack([to_p1, to_p1], axis=1)
        one = np.tile([1.0, 0.0], (len(to_p1), 1))
        V._dev_maps = {"from_p1": (ctx.from_numpy(idx), ctx.from_numpy(w)),
                       "to_p1": (ctx.from_numpy(np.ascontiguousarray(pick)), ctx.from_numpy(np.ascontiguousarray(one)))}
    return V._dev_maps


def local_project(v: grid.Function, V: grid.FunctionSpace, u: grid.Function | None = None):
    """ODE-space <-> PDE-space transfer (utils.py:26-58): a copy when both have the same degrees of freedom (the hot
    path: P1 on both sides), otherwise the interpolation of ``v`` at the points of ``V`` -- between P1 and a P2 / DG1
    ODE space that is a fixed two-term combination per dof, applied on the device (beat_interp2)."""
    U = grid.Function(V) if u is None else u
    src_space = v.function_space
    if v.x.array.size == U.x.array.size:
        U.x.array[:] = v.x.array
        return U
    if src_space.is_p1 and not V.is_p1:
        idx, w = _interp_maps(V)["from_p1"]
        mesh = src_space.mesh
        if mesh.comm.size > 1:  # dofs on the slab's faces interpolate between this rank's and its neighbour's vertices
            from ._engine import exchange_ghost_planes

            exchange_ghost_planes(v.field, mesh.slab, mesh.comm.group)
    elif V.is_p1 and not src_space.is_p1:
        idx, w = _interp_maps(src_space)["to_p1"]
    else:
        raise NotImplementedError("projection between two non-P1 spaces is not implemented")
    dst = U.writable_field()
    ctx = U._ctx
    _hip.check(ctx.lib.beat_interp2(ctx.handle, dst.ptr, v.field.ptr, C.c_void_p(idx.data_ptr()), C.c_void_p(w.data_ptr()), dst.n))
    U._touch()
    return U


def space_from_string(space_string: str, mesh: grid.Mesh, dim: int = 1) -> grid.FunctionSpace:
    """'{family}_{degree}' -> function space (utils.py:86-112): P / CG / Lagrange 1 and 2, DG / dP 0 and 1."""
    family, degree = space_string.split("_")
    known = ("Lagrange", "P", "CG", "Discontinuous Lagrange", "DG", "dP", "Quadrature", "Q", "Quad")
    if family not in known:  # same error as the reference's parse_element (utils.py:80-83)
        raise ValueError(f"Unknown element family: {family}, available families: {list(known)}")
    if dim != 1:
        raise NotImplementedError("vector spaces are not implemented")
    return grid.FunctionSpace(mesh, family, int(degree))


def interpolation_points(V):
    return None


def laplace_dirichlet(V: grid.FunctionSpace, bcs, rtol: float = 1e-10, atol: float = 1e-15, max_it: int = 10_000):
    """Solve  div grad u = 0  on the (active) mesh with Dirichlet values on the vertices of exterior facets:
    ``bcs`` = [(facet ids, value), ...], later entries win where they overlap (the reference hands PETSc the
    list [endo, epi] -- utils.py:176-182).  P1 stiffness rows with M = I are assembled on the device, the
    conditions are imposed by symmetric elimination (beat_rows_apply_dirichlet) and the system is solved with the
    same Jacobi-PCG as the diffusion step (C_m = 0, theta dt = 1).  Returns a Function in V."""
    from ._device import Context
    from ._engine import HipOps

    mesh = V.mesh
    # A one-off set-up solve: on a slab-decomposed mesh every rank solves the WHOLE problem (facet ids and their
    # vertices are global, so every rank knows all the conditions) and keeps its slab of the solution -- replicas, no
    # communication; the full-box rows are freed afterwards (51 GB for a 141 M-node box: fits beside the slab data).
    ctx = Context.default()
    shape, n, plane = mesh.shape_global, mesh.num_nodes_global, mesh.plane
    ops0 = HipOps.from_voxels(ctx, mesh.dim, mesh.n, mesh.h, np.eye(mesh.dim),
                              None if mesh.active is None else mesh._box_active().ravel(), shape, 0, True, True)
    flag = np.zeros(n, dtype=np.uint8)
    g = np.zeros(n)
    for facets, value in bcs:
        nodes = np.unique(mesh.facet_vertices(np.asarray(facets, dtype=np.int64)).ravel())
        flag[nodes] = 1
        g[nodes] = float(value)
    if not flag.any():
        raise ValueError("no Dirichlet facets given")
    flag_dev = ctx.from_numpy(flag)
    gfld, ffld, xfld = ops0.new_field(), ops0.new_field(), ops0.new_field()
    gfld.set(g)
    stiff = ops0._stiff_dev  # modified in place: ops0 is not used for anything else
    _hip.check(ctx.lib.beat_rows_apply_dirichlet(ctx.handle, (C.c_int64 * 3)(*shape), C.c_void_p(stiff.data_ptr()), n,
                                                 C.c_void_p(flag_dev.data_ptr()), gfld.ptr, ffld.ptr))
    ops = HipOps(ctx, shape, True, True, ops0._mass_dev, stiff, per_node=True)
    ops.set_timestep(0.0, 1.0, 1.0)  # A = K
    res = ops.solve_single(gfld, [ffld], [1.0], xfld, rtol, atol, max_it)
    logger.info("Laplace solve: %d PCG iterations, |r| = %.3e", res.iterations, res.residual_norm)
    u = grid.Function(V, name="laplace")
    dst = u.writable_field()
    if mesh.comm.size > 1:
        dst.data.copy_(xfld.data[mesh.slab.z0 * plane : mesh.slab.z1 * plane])
    else:
        dst.copy_from(xfld)
    u._touch()
    return u


def _layers(V, arr, endo_size, epi_size, mid, endo, epi) -> grid.Function:
    out = np.full(arr.shape, float(mid))
    out[arr <= endo_size] = endo
    out[arr >= 1 - epi_size] = epi
    f = grid.Function(V, name="endo_epi")
    f.x.array[:] = out
    return f


def expand_layer(V: grid.FunctionSpace, ft: grid.MeshTags, endo_marker: int, epi_marker: int, endo_size: float,
                 epi_size: float, output_mid_marker: int = 0, output_endo_marker: int = 1,
                 output_epi_marker: int = 2) -> grid.Function:
    """Transmural layer markers (utils.py:115-222): u = 0 on the endocardial facets, 1 on the epicardial ones,
    harmonic in between; nodes with u <= endo_size are endo, u >= 1 - epi_size epi, the rest mid-wall.  Nodes
    outside the tissue keep the value the Laplace solve leaves there (0 -> endo marker); mask them with
    ``mesh.node_active()`` when building per-cell-type ODE solvers."""
    u = laplace_dirichlet(V, [(ft.find(endo_marker), 0.0), (ft.find(epi_marker), 1.0)])
    return _layers(V, np.asarray(u.x.array), endo_size, epi_size, output_mid_marker, output_endo_marker, output_epi_marker)


def expand_layer_biv(V: grid.FunctionSpace, ft: grid.MeshTags, endo_lv_marker: int, endo_rv_marker: int,
                     epi_marker: int, endo_size: float, epi_size: float, output_mid_marker: int = 0,
                     output_endo_marker: int = 1, output_epi_marker: int = 2) -> grid.Function:
    """Bi-ventricular variant (utils.py:225-355): one solve per endocardium, pointwise minimum."""
    u_lv = laplace_dirichlet(V, [(ft.find(endo_lv_marker), 0.0), (ft.find(epi_marker), 1.0)])
    u_rv = laplace_dirichlet(V, [(ft.find(endo_rv_marker), 0.0), (ft.find(epi_marker), 1.0)])
    arr = np.minimum(np.asarray(u_rv.x.array), np.asarray(u_lv.x.array))
    return _layers(V, arr, endo_size, epi_size, output_mid_marker, output_endo_marker, output_epi_marker)
