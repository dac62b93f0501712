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
