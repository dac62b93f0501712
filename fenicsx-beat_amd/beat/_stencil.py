"""Host-side set-up of the matrix-free P1 operators (runs once per model, NumPy).

The reference assembles ``C_m*Mass + theta*dt*K`` with DOLFINx on the simplicial subdivision
of a box (``src/beat/geometry.py:78-139``: 2 triangles per quad sharing the v0-v3 diagonal,
6 tetrahedra per hexahedron sharing the v0-v7 diagonal) from the form of
``src/beat/monodomain_model.py:68-98``.  On a uniform box with constant conductivity tensor
that matrix is a 15-point stencil whose coefficients depend only on which faces of the box a
node lies on (27 node types).  This module builds the element matrices of ONE cell (sum of
its simplices) and accumulates, for each node type, the rows contributed by the cells that
exist around such a node.  The HIP kernels consume the resulting (27, 15) tables.
"""

from __future__ import annotations

import itertools

import numpy as np

# (dx, dy, dz) of the 15 stencil points; must match beat_stencil_offsets() in libbeat_hip
OFFSETS = (
    (0, 0, 0),
    (1, 0, 0), (-1, 0, 0),
    (0, 1, 0), (0, -1, 0),
    (0, 0, 1), (0, 0, -1),
    (1, 1, 0), (-1, -1, 0),
    (0, 1, 1), (0, -1, -1),
    (1, 0, 1), (-1, 0, -1),
    (1, 1, 1), (-1, -1, -1),
)
_OFFSET_INDEX = {o: k for k, o in enumerate(OFFSETS)}

# simplices of one cell, as corner ids with corner k at offsets (k&1, (k>>1)&1, (k>>2)&1)
_SIMPLICES = {
    1: ((0, 1),),
    2: ((0, 1, 3), (0, 2, 3)),
    3: ((0, 1, 3, 7), (0, 1, 7, 5), (0, 5, 7, 4), (0, 3, 2, 7), (0, 6, 4, 7), (0, 2, 6, 7)),
}


def conductivity_matrix(M, dim: int) -> np.ndarray:
    """Accepts a scalar or a (dim, dim) array-like; returns a (dim, dim) float array."""
    M = np.asarray(M, dtype=np.float64)
    if M.ndim == 0:
        return float(M) * np.eye(dim)
    if M.shape == (3, 3) and dim < 3:
        M = M[:dim, :dim]
    if M.shape != (dim, dim):
        raise ValueError(f"conductivity tensor has shape {M.shape}, expected ({dim}, {dim})")
    return M


def cell_matrices(h, M) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Element mass and stiffness matrices (2^d x 2^d) of one box cell of size ``h`` split into
    simplices, plus the per-simplex matrices: returns (mass, stiff, simplex_data) where
    simplex_data[s] = (corner ids, mass_s, stiff_s, volume)."""
    h = np.atleast_1d(np.asarray(h, dtype=np.float64))
    d = len(h)
    M = conductivity_matrix(M, d)
    nc = 2**d
    corners = np.array([[(k >> a) & 1 for a in range(d)] for k in range(nc)], dtype=np.float64) * h
    mass = np.zeros((nc, nc))
    stiff = np.zeros((nc, nc))
    per_simplex = []
    fact = float(np.prod(np.arange(1, d + 1)))
    for simplex in _SIMPLICES[d]:
        X = corners[list(simplex)]  # (d+1, d)
        A = np.hstack([np.ones((d + 1, 1)), X])
        vol = abs(np.linalg.det(A)) / fact
        grads = np.linalg.inv(A)[1:, :].T  # (d+1, d): gradient of each barycentric function
        ke = vol * grads @ M @ grads.T
        me = vol / ((d + 1) * (d + 2)) * (np.ones((d + 1, d + 1)) + np.eye(d + 1))
        idx = np.ix_(simplex, simplex)
        mass[idx] += me
        stiff[idx] += ke
        per_simplex.append((simplex, me, ke, vol))
    return mass, stiff, per_simplex


def stencil_tables(dim: int, h, M) -> tuple[np.ndarray, np.ndarray]:
    """(mass_tab, stiff_tab), each (27, 15); type = tx + 3*ty + 9*tz with t = 0 (low face),
    1 (interior), 2 (high face).  Axes >= dim are 'interior' with zero coupling."""
    mass, stiff, _ = cell_matrices(h, M)
    mass_tab = np.zeros((27, 15))
    stiff_tab = np.zeros((27, 15))
    for types in itertools.product(range(3), repeat=3):  # (tx, ty, tz)
        if any(types[a] != 1 for a in range(dim, 3)):
            continue
        typ = types[0] + 3 * types[1] + 9 * types[2]
        # cells around the node: per axis the cell starts at node-1 (c = -1) or node (c = 0)
        choices = []
        for a in range(dim):
            choices.append({0: (0,), 1: (-1, 0), 2: (-1,)}[types[a]])
        for cell in itertools.product(*choices):
            local = sum((-cell[a]) << a for a in range(dim))  # corner id of the node in that cell
            for other in range(2**dim):
                off = tuple(cell[a] + ((other >> a) & 1) for a in range(dim)) + (0,) * (3 - dim)
                k = _OFFSET_INDEX.get(off)
                if k is None:
                    assert mass[local, other] == 0.0 and stiff[local, other] == 0.0, off
                    continue
                mass_tab[typ, k] += mass[local, other]
                stiff_tab[typ, k] += stiff[local, other]
    return mass_tab, stiff_tab


def tables_y_as_z(tab: np.ndarray) -> np.ndarray:
    """A 2-D (27, 15) table re-expressed for a grid stored as (nx, 1, ny): the kernels decompose along their slowest
    axis, so a 2-D mesh that is cut into slabs of rows is handed to them with y in the place of z -- node numbering
    unchanged (x fastest, then the rows).  Type tx + 3 ty + 9 (interior z) -> tx + 3 (collapsed y) + 9 ty, offset
    (dx, dy, 0) -> (dx, 0, dy); every 2-D offset has its slot in the 15-point set."""
    out = np.zeros_like(tab)
    for tx in range(3):
        for ty in range(3):
            for k, (dx, dy, dz) in enumerate(OFFSETS):
                if dz != 0 or tab[tx + 3 * ty + 9, k] == 0.0:
                    continue
                out[tx + 3 + 9 * ty, _OFFSET_INDEX[(dx, 0, dy)]] = tab[tx + 3 * ty + 9, k]
    return out


def fields_y_as_z(rows: np.ndarray) -> np.ndarray:
    """Per-node rows (15, n) of a 2-D operator re-expressed for a grid stored as (nx, 1, ny) -- tables_y_as_z for
    stencil_fields: the coefficient of offset (dx, dy, 0) moves to the slot of (dx, 0, dy); node numbering unchanged."""
    out = np.zeros_like(rows)
    for k, (dx, dy, dz) in enumerate(OFFSETS):
        if dz == 0:
            out[_OFFSET_INDEX[(dx, 0, dy)]] = rows[k]
        elif rows[k].any():
            raise ValueError("fields_y_as_z: a 2-D operator has no coefficients out of its plane")
    return out


def stencil_fields(dim: int, cells: tuple[int, ...], h, M, active=None, z_range=None) -> tuple[np.ndarray, np.ndarray]:
    """Per-node rows of the same operators for voxel-masked domains and spatially varying conductivity:
    returns (mass, stiff), each (15, n_local) with n_local = nx*ny*(z1-z0) nodes, x fastest.

    cells   : box cells per axis (length dim)
    M       : scalar, (dim, dim), per box cell (nbox, dim, dim) or per simplex (nbox*spc, dim, dim)
    active  : None, bool per box cell (nbox,) or per simplex (nbox*spc,); box cells numbered x fastest,
              simplex id = box*spc + k as in grid.Mesh
    z_range : (z0, z1) node planes owned by this rank (3-D only); default: all

    Element matrices are those of cell_matrices(); every (simplex, corner pair) contributes to one stencil
    slot of one node per cell, so the accumulation is 96 shifted array additions (no scatter)."""
    h = tuple(float(v) for v in np.atleast_1d(h))[:dim]
    d = dim
    c = [int(v) for v in cells] + [1] * (3 - d)           # box cells per axis (1 for unused axes)
    nn = [c[a] + 1 if a < d else 1 for a in range(3)]     # nodes per axis
    simp = _SIMPLICES[d]
    spc = len(simp)
    nbox = c[0] * c[1] * c[2]
    z0, z1 = (0, nn[2]) if z_range is None else (int(z_range[0]), int(z_range[1]))
    # cell layers touching the owned planes, node planes they cover
    if d == 3:
        cz_lo, cz_hi = max(z0 - 1, 0), min(z1, c[2])
    else:
        cz_lo, cz_hi = 0, 1
    ncz = cz_hi - cz_lo
    pz = ncz + 1 if d == 3 else 1  # node planes covered by those layers
    mass = np.zeros((15, pz, nn[1], nn[0]))
    stiff = np.zeros((15, pz, nn[1], nn[0]))

    def per_cell(arr, tail):
        """-> array (ncz, cy, cx, spc) + tail, restricted to the needed cell layers"""
        arr = np.asarray(arr)
        if arr.shape[0] == nbox:
            arr = arr.reshape((c[2], c[1], c[0]) + tail)[cz_lo:cz_hi]
            return np.broadcast_to(arr[:, :, :, None], arr.shape[:3] + (spc,) + tail)
        if arr.shape[0] == nbox * spc:
            return arr.reshape((c[2], c[1], c[0], spc) + tail)[cz_lo:cz_hi]
        raise ValueError(f"per-cell data has leading size {arr.shape[0]}, expected {nbox} or {nbox * spc}")

    act = None if active is None else per_cell(np.asarray(active, dtype=bool), ())
    Marr = np.asarray(M, dtype=np.float64)
    Mc = None
    if Marr.ndim == 3:
        Mc = per_cell(Marr, (d, d))
    else:
        Mconst = conductivity_matrix(Marr, d)
    _, _, per_simplex = cell_matrices(h, np.eye(d))
    corners = np.array([[(k >> a) & 1 for a in range(d)] for k in range(2**d)], dtype=np.float64) * np.asarray(h)
    fact = float(np.prod(np.arange(1, d + 1)))
    for k, s in enumerate(simp):
        X = corners[list(s)]
        A = np.hstack([np.ones((d + 1, 1)), X])
        vol = abs(np.linalg.det(A)) / fact
        G = np.linalg.inv(A)[1:, :].T  # (d+1, d)
        me = per_simplex[k][1]
        if Mc is None:
            ke = vol * G @ Mconst @ G.T  # (d+1, d+1)
        else:
            ke = vol * np.einsum("ai,zyxij,bj->zyxab", G, Mc[:, :, :, k], G)
        a_k = None if act is None else act[:, :, :, k]
        for ia, ca in enumerate(s):
            oa = (ca & 1, (ca >> 1) & 1, (ca >> 2) & 1)
            sl = (slice(oa[2], oa[2] + ncz) if d == 3 else slice(0, 1), slice(oa[1], oa[1] + c[1]) if d >= 2 else slice(0, 1),
                  slice(oa[0], oa[0] + c[0]))
            for ib, cb in enumerate(s):
                ob = (cb & 1, (cb >> 1) & 1, (cb >> 2) & 1)
                slot = _OFFSET_INDEX[(ob[0] - oa[0], ob[1] - oa[1], ob[2] - oa[2])]
                kv = ke[..., ia, ib] if Mc is not None else ke[ia, ib]
                mv = me[ia, ib]
                if a_k is not None:
                    kv = kv * a_k
                    mv = mv * a_k
                mass[(slot,) + sl] += mv
                stiff[(slot,) + sl] += kv
    if d == 3:
        lo = z0 - cz_lo
        mass = mass[:, lo : lo + (z1 - z0)]
        stiff = stiff[:, lo : lo + (z1 - z0)]
    return (np.ascontiguousarray(mass.reshape(15, -1)), np.ascontiguousarray(stiff.reshape(15, -1)))


def voxel_element_tensors(dim: int, h) -> tuple[np.ndarray, np.ndarray]:
    """(T, Me) for the device-side row assembly (beat_pde_assemble_rows): T[a, b, 3*i + j] such that a voxel's
    element stiffness matrix is K_e[a, b] = sum_ij T[a, b, 3i+j] M_ij, and the voxel's element mass matrix Me;
    corners beyond 2^dim and tensor components beyond dim are zero-padded."""
    h = tuple(float(v) for v in np.atleast_1d(h))[:dim]
    d = dim
    T = np.zeros((8, 8, 9))
    Me = np.zeros((8, 8))
    corners = np.array([[(k >> a) & 1 for a in range(d)] for k in range(2**d)], dtype=np.float64) * np.asarray(h)
    fact = float(np.prod(np.arange(1, d + 1)))
    for s in _SIMPLICES[d]:
        X = corners[list(s)]
        A = np.hstack([np.ones((d + 1, 1)), X])
        vol = abs(np.linalg.det(A)) / fact
        G = np.linalg.inv(A)[1:, :].T  # (d+1, d)
        me = vol / ((d + 1) * (d + 2)) * (np.ones((d + 1, d + 1)) + np.eye(d + 1))
        for ia, ca in enumerate(s):
            for ib, cb in enumerate(s):
                Me[ca, cb] += me[ia, ib]
                for i in range(d):
                    for j in range(d):
                        T[ca, cb, 3 * i + j] += vol * G[ia, i] * G[ib, j]
    return T, Me
