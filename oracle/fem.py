"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in NumPy/SciPy, of the arithmetic the reference delegates to
DOLFINx/FFCx/PETSc for the monodomain diffusion step:

* P1 (Lagrange degree 1) finite elements on DOLFINx's simplicial subdivision of a
  box: 2 triangles per quad sharing the v0-v3 diagonal (``DiagonalType.right``),
  6 tetrahedra per hexahedron all sharing the v0-v7 diagonal (Kuhn / Freudenthal
  subdivision).  Mesh sizes follow ``src/beat/geometry.py:78-139``
  (``n = rint(L/dx)`` cells per axis).
* the theta-rule weak form of ``src/beat/monodomain_model.py:68-98``::

      (C_m Mass + theta dt K) v = (C_m Mass - (1-theta) dt K) v_ + dt b_stim(t0 + theta dt)

  with *consistent* mass matrix, K_ij = int (M grad phi_j) . grad phi_i, and
  b_stim,i = sum_k I_k(t) int_{dz_k} phi_i  (``src/beat/base_model.py:247-248``).
* step / solve control flow of ``src/beat/base_model.py:208-297`` including the
  "no assign_previous() after the final step" behaviour of ``solve``.

The element matrices are assembled literally cell by cell (no stencil is assumed);
``stencil_table`` then *derives* the constant-coefficient 15-point stencil (and its 27
boundary variants) from a literal assembly on a 2x2x2-cell mini mesh, and
``tests/test_oracle_fem.py`` checks that table against the assembled sparse matrix.

Pinned by: tests/test_oracle_fem.py (reference thresholds tests/test_monodomain.py:11-147,
tests/test_stimulation.py:12-107 and the Niederer table demos/niederer_benchmark.py:315-319).
The subdivision pattern itself is stated from memory of dolfinx's mesh generator (the
dolfinx sources are not in /root/reference).
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# corner k of a hex has offsets (k&1, (k>>1)&1, (k>>2)&1)  -> v0..v7 as in dolfinx build_tet
KUHN_TETS = ((0, 1, 3, 7), (0, 1, 7, 5), (0, 5, 7, 4), (0, 3, 2, 7), (0, 6, 4, 7), (0, 2, 6, 7))
# corner k of a quad has offsets (k&1, (k>>1)&1); DiagonalType.right
RIGHT_TRIS = ((0, 1, 3), (0, 2, 3))

# The 15 stencil offsets (dx, dy, dz) of the Kuhn subdivision: centre, the 3 axes,
# the 3 face diagonals (1,1,0),(0,1,1),(1,0,1) and the body diagonal (1,1,1), +/-.
STENCIL_OFFSETS = (
    (0, 0, 0),
    (1, 0, 0), (-1, 0, 0),
    (0, 1, 0), (0, -1, 0),
    (0, 0, 1), (0, 0, -1),
    (1, 1, 0), (-1, -1, 0),
    (0, 1, 1), (0, -1, -1),
    (1, 0, 1), (-1, 0, -1),
    (1, 1, 1), (-1, -1, -1),
)


@dataclass
class BoxMesh:
    """Structured simplicial mesh of [0,L] (1-D), [0,Lx]x[0,Ly] or a 3-D box.

    ``n`` = cells per axis (length dim); nodes are numbered lexicographically with x
    fastest: ``id = ix + (nx+1) * (iy + (ny+1) * iz)``.  (DOLFINx renumbers dofs; nothing
    on the hot path depends on the numbering.)
    """

    n: tuple[int, ...]
    L: tuple[float, ...]
    origin: tuple[float, ...] | None = None
    cells: np.ndarray = field(init=False, repr=False)  # (ncells, dim+1) node ids
    x: np.ndarray = field(init=False, repr=False)  # (nnodes, dim)

    def __post_init__(self):
        self.n = tuple(int(v) for v in self.n)
        self.L = tuple(float(v) for v in self.L)
        if self.origin is None:
            self.origin = (0.0,) * len(self.n)
        dim = len(self.n)
        assert dim in (1, 2, 3)
        axes = [o + np.linspace(0.0, L, m + 1) for o, L, m in zip(self.origin, self.L, self.n)]
        # x fastest: build index grids with the slowest axis first
        idx = np.meshgrid(*[np.arange(m + 1) for m in reversed(self.n)], indexing="ij")
        idx = list(reversed(idx))  # idx[a] = index along axis a for every node
        self.x = np.stack([axes[a][idx[a].ravel()] for a in range(dim)], axis=1)
        npts = [m + 1 for m in self.n]
        if dim == 1:
            i = np.arange(self.n[0])
            self.cells = np.stack([i, i + 1], axis=1)
        elif dim == 2:
            nx, ny = self.n
            iy, ix = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
            v0 = (ix + npts[0] * iy).ravel()
            corner = np.stack([v0, v0 + 1, v0 + npts[0], v0 + npts[0] + 1], axis=1)
            self.cells = np.concatenate(
                [corner[:, list(t)][:, None, :] for t in RIGHT_TRIS], axis=1
            ).reshape(-1, 3)
        else:
            nx, ny, nz = self.n
            iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
            v0 = (ix + npts[0] * (iy + npts[1] * iz)).ravel()
            sx, sy, sz = 1, npts[0], npts[0] * npts[1]
            corner = np.stack(
                [v0 + (k & 1) * sx + ((k >> 1) & 1) * sy + ((k >> 2) & 1) * sz for k in range(8)],
                axis=1,
            )
            self.cells = np.concatenate(
                [corner[:, list(t)][:, None, :] for t in KUHN_TETS], axis=1
            ).reshape(-1, 4)

    @property
    def dim(self) -> int:
        return len(self.n)

    @property
    def num_nodes(self) -> int:
        return self.x.shape[0]

    @property
    def shape_nodes(self) -> tuple[int, ...]:
        return tuple(m + 1 for m in self.n)

    def locate_cells(self, predicate) -> np.ndarray:
        """dolfinx.mesh.locate_entities(mesh, tdim, predicate): cells whose vertices ALL
        satisfy ``predicate(x)`` with x of shape (3, npoints) (padded with zeros)."""
        xp = np.zeros((3, self.num_nodes))
        xp[: self.dim] = self.x.T
        ok = np.asarray(predicate(xp), dtype=bool)
        return np.nonzero(ok[self.cells].all(axis=1))[0]


def _cell_geometry(mesh: BoxMesh):
    """volumes (ncells,) and barycentric gradients (ncells, dim+1, dim)."""
    X = mesh.x[mesh.cells]  # (nc, d+1, d)
    d = mesh.dim
    nc = X.shape[0]
    A = np.concatenate([np.ones((nc, d + 1, 1)), X], axis=2)  # rows [1 x y z]
    Ainv = np.linalg.inv(A)  # columns are coefficients of each barycentric function
    grads = Ainv[:, 1:, :].transpose(0, 2, 1)  # (nc, d+1, d)
    fact = {1: 1.0, 2: 2.0, 3: 6.0}[d]
    vol = np.abs(np.linalg.det(A)) / fact
    return vol, grads


def _as_tensor(M, dim: int) -> np.ndarray:
    M = np.asarray(M, dtype=float)
    if M.ndim == 0:
        return float(M) * np.eye(dim)
    assert M.shape == (dim, dim)
    return M


def assemble_mass(mesh: BoxMesh, cells: np.ndarray | None = None) -> sp.csr_matrix:
    vol, _ = _cell_geometry(mesh)
    d = mesh.dim
    c = mesh.cells if cells is None else mesh.cells[cells]
    v = vol if cells is None else vol[cells]
    # exact P1 mass matrix on a simplex: |T| / ((d+1)(d+2)) * (1 + delta_ab)
    loc = (np.ones((d + 1, d + 1)) + np.eye(d + 1)) / ((d + 1) * (d + 2))
    data = v[:, None, None] * loc[None]
    rows = np.repeat(c[:, :, None], d + 1, axis=2)
    cols = np.repeat(c[:, None, :], d + 1, axis=1)
    N = mesh.num_nodes
    return sp.coo_matrix((data.ravel(), (rows.ravel(), cols.ravel())), shape=(N, N)).tocsr()


def assemble_stiffness(mesh: BoxMesh, M) -> sp.csr_matrix:
    """K_ij = int (M grad phi_j) . grad phi_i ; M scalar, (dim,dim) constant, or per-cell
    (ncells, dim, dim)."""
    vol, g = _cell_geometry(mesh)
    d = mesh.dim
    M = np.asarray(M, dtype=float)
    if M.ndim == 3:
        Mg = np.einsum("cij,cbj->cbi", M, g)
    else:
        Mg = np.einsum("ij,cbj->cbi", _as_tensor(M, d), g)
    data = vol[:, None, None] * np.einsum("cai,cbi->cab", g, Mg)
    c = mesh.cells
    rows = np.repeat(c[:, :, None], d + 1, axis=2)
    cols = np.repeat(c[:, None, :], d + 1, axis=1)
    N = mesh.num_nodes
    return sp.coo_matrix((data.ravel(), (rows.ravel(), cols.ravel())), shape=(N, N)).tocsr()


def stimulus_weights(mesh: BoxMesh, cells: np.ndarray | None = None) -> np.ndarray:
    """w_i = int_{marked cells} phi_i  (|T|/(d+1) per vertex).  cells=None: whole domain."""
    vol, _ = _cell_geometry(mesh)
    d = mesh.dim
    c = mesh.cells if cells is None else mesh.cells[cells]
    v = vol if cells is None else vol[cells]
    w = np.zeros(mesh.num_nodes)
    np.add.at(w, c.ravel(), np.repeat(v / (d + 1), d + 1))
    return w


def exterior_facet_weights(mesh: BoxMesh, active_cells=None, node_ok=None, facet_filter=None) -> np.ndarray:
    """w_i = int_{exterior facets} phi_i dS (the ``ds`` measure of stimulation.py:63-111) from the simplicial
    mesh itself: the facets of the (active) simplices that belong to exactly one of them; ``node_ok`` (bool per
    node) keeps the facets whose vertices ALL satisfy it (dolfinx locate_entities_boundary)."""
    d = mesh.dim
    cells = mesh.cells if active_cells is None else mesh.cells[np.asarray(active_cells, dtype=bool)]
    faces = np.concatenate([np.delete(cells, k, axis=1) for k in range(d + 1)], axis=0)
    faces = np.sort(faces, axis=1)
    uniq, counts = np.unique(faces, axis=0, return_counts=True)
    ext = uniq[counts == 1]
    if node_ok is not None:
        ext = ext[np.asarray(node_ok, dtype=bool)[ext].all(axis=1)]
    if facet_filter is not None:  # callable on the facets' vertex coordinates (nf, d, dim) -> bool (nf,)
        ext = ext[np.asarray(facet_filter(mesh.x[ext]), dtype=bool)]
    X = mesh.x[ext]
    if d == 2:
        meas = np.linalg.norm(X[:, 1] - X[:, 0], axis=1)
    else:
        meas = 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)
    w = np.zeros(mesh.num_nodes)
    np.add.at(w, ext.ravel(), np.repeat(meas / d, d))
    return w


def exterior_facet_load(mesh: BoxMesh, f, active_cells=None, node_ok=None, facet_filter=None, m: int = 6) -> np.ndarray:
    """b_i = int_{exterior facets} f(x) phi_i dS (a UFL expression times the test function on a ``ds`` measure:
    stimulation.py:14-24, base_model.py:247-248), f taking x of shape (dim, npts); same facet selection as
    ``exterior_facet_weights``; Gauss rule of 2m-(d-1) exactness on every exterior edge / triangle."""
    d = mesh.dim
    cells = mesh.cells if active_cells is None else mesh.cells[np.asarray(active_cells, dtype=bool)]
    faces = np.concatenate([np.delete(cells, k, axis=1) for k in range(d + 1)], axis=0)
    faces = np.sort(faces, axis=1)
    uniq, counts = np.unique(faces, axis=0, return_counts=True)
    ext = uniq[counts == 1]
    if node_ok is not None:
        ext = ext[np.asarray(node_ok, dtype=bool)[ext].all(axis=1)]
    if facet_filter is not None:
        ext = ext[np.asarray(facet_filter(mesh.x[ext]), dtype=bool)]
    X = mesh.x[ext]  # (nf, d, dim)
    if d == 2:
        meas = np.linalg.norm(X[:, 1] - X[:, 0], axis=1)
    else:
        meas = 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)
    lam, w = _simplex_quadrature(d - 1, m)
    w = w / w.sum()
    xq = np.einsum("qa,fad->fqd", lam, X)
    fq = np.asarray(f(xq.reshape(-1, d).T)).reshape(xq.shape[0], -1)
    loc = np.einsum("fq,q,qa->fa", fq, w, lam) * meas[:, None]
    b = np.zeros(mesh.num_nodes)
    np.add.at(b, ext.ravel(), loc.ravel())
    return b


def assemble_stiffness_nodal_fibres(mesh: BoxMesh, f_nodal: np.ndarray, s_l: float, s_t: float, m: int = 4) -> sp.csr_matrix:
    """K_ij = int (s_l f f^T + s_t (I - f f^T)) grad phi_j . grad phi_i dx with the fibre direction f a vector P1
    FUNCTION (nodal values ``f_nodal`` (num_nodes, dim), interpolated linearly inside every simplex -- the UFL
    expression conductivities.py:101-104 builds from ``geo.f0``), integrated by quadrature at the points of every
    simplex (not by averaging the tensor in closed form, which is what the product does)."""
    d = mesh.dim
    lam, w = _simplex_quadrature(d, m)
    w = w / w.sum()
    F = np.asarray(f_nodal, dtype=np.float64)[mesh.cells]  # (nc, d+1, d)
    fq = np.einsum("qa,cad->cqd", lam, F)                  # fibre at the quadrature points
    ff = np.einsum("cqi,cqj,q->cij", fq, fq, w)            # cell average of f f^T
    Mbar = s_t * np.eye(d)[None] + (s_l - s_t) * ff
    return assemble_stiffness(mesh, Mbar)


# Degree-precision quadrature used only to integrate smooth manufactured sources and
# L2 errors (the reference lets UFL pick a degree / uses quadrature_degree=8).
def _simplex_quadrature(d: int, m: int = 6):
    """Collapsed Gauss-Jacobi-free tensor rule via Duffy transform; exact to degree >= 2m-d."""
    gx, gw = np.polynomial.legendre.leggauss(m)
    gx = 0.5 * (gx + 1.0)
    gw = 0.5 * gw
    if d == 1:
        return np.stack([1 - gx, gx], axis=1), gw
    if d == 2:
        pts, wts = [], []
        for (a, wa), (b, wb) in itertools.product(zip(gx, gw), repeat=2):
            l1 = a
            l2 = (1 - a) * b
            pts.append((1 - l1 - l2, l1, l2))
            wts.append(wa * wb * (1 - a))
        return np.array(pts), np.array(wts)
    pts, wts = [], []
    for (a, wa), (b, wb), (c, wc) in itertools.product(zip(gx, gw), repeat=3):
        l1 = a
        l2 = (1 - a) * b
        l3 = (1 - a) * (1 - b) * c
        pts.append((1 - l1 - l2 - l3, l1, l2, l3))
        wts.append(wa * wb * wc * (1 - a) ** 2 * (1 - b))
    return np.array(pts), np.array(wts)


def load_vector(mesh: BoxMesh, f, cells: np.ndarray | None = None, m: int = 6) -> np.ndarray:
    """b_i = int f(x) phi_i dx with f(x) taking x of shape (dim, npts)."""
    vol, _ = _cell_geometry(mesh)
    d = mesh.dim
    lam, w = _simplex_quadrature(d, m)  # (nq, d+1), weights sum to 1/d!
    w = w / w.sum()
    c = mesh.cells if cells is None else mesh.cells[cells]
    v = vol if cells is None else vol[cells]
    X = mesh.x[c]  # (nc, d+1, d)
    xq = np.einsum("qa,cad->cqd", lam, X)  # (nc, nq, d)
    fq = np.asarray(f(xq.reshape(-1, d).T)).reshape(xq.shape[0], -1)
    loc = np.einsum("cq,q,qa->ca", fq, w, lam) * v[:, None]
    b = np.zeros(mesh.num_nodes)
    np.add.at(b, c.ravel(), loc.ravel())
    return b


def l2_error(mesh: BoxMesh, vh: np.ndarray, exact, m: int = 6) -> float:
    """sqrt(int (v_h - exact)^2) with v_h the P1 function with nodal values vh."""
    vol, _ = _cell_geometry(mesh)
    d = mesh.dim
    lam, w = _simplex_quadrature(d, m)
    w = w / w.sum()
    X = mesh.x[mesh.cells]
    xq = np.einsum("qa,cad->cqd", lam, X)
    vq = np.einsum("qa,ca->cq", lam, vh[mesh.cells])
    eq = np.asarray(exact(xq.reshape(-1, d).T)).reshape(vq.shape)
    return float(np.sqrt(np.sum(((vq - eq) ** 2) * w[None, :] * vol[:, None])))


def evaluate_p1(mesh: BoxMesh, vh: np.ndarray, points: np.ndarray) -> np.ndarray:
    """Point evaluation of a P1 function (scifem.evaluate_function stand-in)."""
    points = np.atleast_2d(np.asarray(points, dtype=float))[:, : mesh.dim]
    d = mesh.dim
    out = np.empty(len(points))
    h = np.array(mesh.L) / np.array(mesh.n)
    npts = mesh.shape_nodes
    for k, p in enumerate(points):
        rel = (p - np.array(mesh.origin)) / h
        idx = np.minimum(np.maximum(np.floor(rel).astype(int), 0), np.array(mesh.n) - 1)
        hexid = idx[0]
        if d >= 2:
            hexid += mesh.n[0] * idx[1]
        if d == 3:
            hexid += mesh.n[0] * mesh.n[1] * idx[2]
        per = {1: 1, 2: 2, 3: 6}[d]
        best = None
        for c in range(hexid * per, hexid * per + per):
            X = mesh.x[mesh.cells[c]]
            A = np.concatenate([np.ones((d + 1, 1)), X], axis=1)
            lam = np.linalg.solve(A.T, np.concatenate([[1.0], p]))
            if best is None or lam.min() > best[0]:
                best = (lam.min(), lam, c)
        out[k] = best[1] @ vh[mesh.cells[best[2]]]
    return out


# ----------------------------------------------------------------------------------------------
# stencil derivation (what the HIP kernels apply matrix-free)
# ----------------------------------------------------------------------------------------------


def stencil_table(dim: int, h, M, C_m: float, theta_dt: float) -> tuple[np.ndarray, np.ndarray]:
    """Derive, by literal element assembly on a 2-cells-per-axis mini mesh, the rows of
    ``mass_scale*Mass`` and ``K`` for each of the 27 node types (lo/mid/hi per axis).

    Returns (mass_tab, stiff_tab) of shape (27, 15): entry [type, k] multiplies the node at
    offset STENCIL_OFFSETS[k]; type = tx + 3*ty + 9*tz with t in {0: low face, 1: interior,
    2: high face}.  Axes beyond ``dim`` use type 1 and zero coefficients.
    (C_m and theta_dt are *not* folded in; they are accepted so callers can form
    ``C_m*mass + theta_dt*stiff`` consistently.)
    """
    h = tuple(float(v) for v in np.atleast_1d(h))
    assert len(h) == dim
    mesh = BoxMesh(n=(2,) * dim, L=tuple(2 * v for v in h))
    Mass = assemble_mass(mesh).toarray()
    K = assemble_stiffness(mesh, M).toarray()
    mass_tab = np.zeros((27, 15))
    stiff_tab = np.zeros((27, 15))
    for t3 in itertools.product(range(3), repeat=3):  # (tz, ty, tx)
        tz, ty, tx = t3
        ttuple = (tx, ty, tz)
        if any(ttuple[a] != 1 for a in range(dim, 3)):
            continue
        node = sum(ttuple[a] * 3**a for a in range(dim))
        typ = tx + 3 * ty + 9 * tz
        for k, off in enumerate(STENCIL_OFFSETS):
            if any(off[a] != 0 for a in range(dim, 3)):
                continue
            nb = [ttuple[a] + off[a] for a in range(dim)]
            if any(v < 0 or v > 2 for v in nb):
                continue
            j = sum(nb[a] * 3**a for a in range(dim))
            mass_tab[typ, k] = Mass[node, j]
            stiff_tab[typ, k] = K[node, j]
    # sanity: every assembled entry is covered by the 15 offsets
    for tab, mat in ((mass_tab, Mass), (stiff_tab, K)):
        for node in range(3**dim):
            tt = [(node // 3**a) % 3 for a in range(dim)] + [1] * (3 - dim)
            typ = tt[0] + 3 * tt[1] + 9 * tt[2]
            assert abs(tab[typ].sum() - mat[node].sum()) < 1e-12 * max(1.0, abs(mat[node]).sum())
    return mass_tab, stiff_tab


def node_types(shape_nodes: tuple[int, ...]) -> np.ndarray:
    """type id (tx + 3 ty + 9 tz) of every node of a grid with the given node counts
    (x fastest).  An axis with a single node is 'interior' (type 1, its coefficients are 0)."""
    s = list(shape_nodes) + [1] * (3 - len(shape_nodes))

    def t(n):
        if n == 1:
            return np.ones(1, dtype=np.int64)
        a = np.ones(n, dtype=np.int64)
        a[0] = 0
        a[-1] = 2
        return a

    tx, ty, tz = t(s[0]), t(s[1]), t(s[2])
    typ = tx[None, None, :] + 3 * ty[None, :, None] + 9 * tz[:, None, None]
    return typ.ravel()


def apply_stencil(tab: np.ndarray, shape_nodes: tuple[int, ...], x: np.ndarray) -> np.ndarray:
    """y = A x with A given by the (27,15) coefficient table (NumPy restatement of the HIP
    stencil kernel; zero padding outside the box)."""
    s = list(shape_nodes) + [1] * (3 - len(shape_nodes))
    nx, ny, nz = s
    X = np.zeros((nz + 2, ny + 2, nx + 2))
    X[1:-1, 1:-1, 1:-1] = x.reshape(nz, ny, nx)
    typ = node_types(shape_nodes).reshape(nz, ny, nx)
    y = np.zeros((nz, ny, nx))
    for k, (ox, oy, oz) in enumerate(STENCIL_OFFSETS):
        c = tab[:, k][typ]
        y += c * X[1 + oz : 1 + oz + nz, 1 + oy : 1 + oy + ny, 1 + ox : 1 + ox + nx]
    return y.ravel()


# ----------------------------------------------------------------------------------------------
# theta-rule PDE model (restates base_model.py / monodomain_model.py)
# ----------------------------------------------------------------------------------------------


def pcg_jacobi(A, b, x0, rtol=1e-10, atol=1e-50, maxit=10000):
    """Jacobi-preconditioned CG, convergence ||r||_2 <= max(rtol*||b||_2, atol) (PETSc's
    default unpreconditioned-norm style test, stated relative to ||b||).  Returns x, its, rnorm.
    This is the algorithm the HIP PCG implements; dinv = 1/diag(A)."""
    if sp.issparse(A):
        dinv = 1.0 / A.diagonal()
        mv = lambda v: A @ v
    else:
        mv, dinv = A
    x = x0.copy()
    r = b - mv(x)
    bnorm = np.linalg.norm(b)
    tol = max(rtol * bnorm, atol)
    z = dinv * r
    p = z.copy()
    rz = r @ z
    rnorm = np.linalg.norm(r)
    its = 0
    while rnorm > tol and its < maxit:
        q = mv(p)
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        z = dinv * r
        rz_new = r @ z
        rnorm = np.linalg.norm(r)
        beta = rz_new / rz
        rz = rz_new
        p = z + beta * p
        its += 1
    return x, its, rnorm


def chebyshev_coefficients(m, lmin, lmax):
    """Polynomial preconditioner coefficients: m steps of the Chebyshev iteration for B z = rhat on
    [lmin, lmax] from a zero start, expanded in powers of B (Saad, Iterative Methods, Alg. 12.1)."""
    import numpy.polynomial.polynomial as P

    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    rho = 1.0 / sigma
    d = np.array([1.0 / theta])
    z = d.copy()
    res = P.polysub([1.0], P.polymulx(z))
    for _ in range(1, m):
        rho_new = 1.0 / (2.0 * sigma - rho)
        d = P.polyadd(rho_new * rho * d, (2.0 * rho_new / delta) * res)
        z = P.polyadd(z, d)
        res = P.polysub([1.0], P.polymulx(z))
        rho = rho_new
    out = np.zeros(m)
    out[: len(z)] = z
    return out


def pcg_polynomial(A, b, x0, coef, rtol=1e-10, atol=1e-50, maxit=10000):
    """PCG with the preconditioner z = sum_k coef[k] (D^-1 A)^k D^-1 r (coef of length 1 = Jacobi)."""
    dinv = 1.0 / A.diagonal()

    def precond(r):
        rh = dinv * r
        s = coef[-1] * rh
        for c in coef[-2::-1]:
            s = c * rh + dinv * (A @ s)
        return s

    x = x0.copy()
    r = b - A @ x
    tol = max(rtol * np.linalg.norm(b), atol)
    z = precond(r)
    p = z.copy()
    rz = r @ z
    its = 0
    while np.linalg.norm(r) > tol and its < maxit:
        q = A @ p
        alpha = rz / (p @ q)
        x += alpha * p
        r -= alpha * q
        z = precond(r)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
        its += 1
    return x, its, np.linalg.norm(r)


class OracleStimulus:
    """I_k(t) * int_{dz_k} f_k(x) phi_i.  ``amp(t)`` scalar function of time, ``weights`` (N,)."""

    def __init__(self, amp, weights):
        self.amp = amp
        self.weights = np.asarray(weights, dtype=float)


def window(start, duration, value):
    """ufl.conditional(And(ge(t,start), le(t,start+duration)), value, 0) -- stimulation.py:270."""
    return lambda t: value if (t >= start and t <= start + duration) else 0.0


class OracleMonodomainModel:
    """Restates BaseModel/MonodomainModel (src/beat/base_model.py:73-297,
    src/beat/monodomain_model.py:42-98) on a BoxMesh with direct (sparse LU) solves, i.e. the
    reference's default ``preonly+lu`` path, or Jacobi-PCG (``solver='pcg'``)."""

    def __init__(self, mesh: BoxMesh, M, stimuli=(), C_m=1.0, theta=0.5, default_timestep=1.0,
                 solver="lu", rtol=1e-10, active_cells=None):
        """``active_cells``: bool per simplex -- the domain is the union of those cells (what the reference
        gets from a mesh of the tissue only); nodes outside it get identity rows and keep their values."""
        self.mesh = mesh
        self.C_m = float(C_m)
        self.theta = float(theta)
        if active_cells is None:
            self.Mass = assemble_mass(mesh)
            self.K = assemble_stiffness(mesh, M)
            self.outside = None
        else:
            act = np.asarray(active_cells, dtype=bool)
            Mc = np.asarray(M, dtype=float)
            if Mc.ndim != 3:
                Mc = np.broadcast_to(_as_tensor(Mc, mesh.dim), (len(act), mesh.dim, mesh.dim))
            self.Mass = assemble_mass(mesh, np.nonzero(act)[0])
            self.K = assemble_stiffness(mesh, Mc * act[:, None, None])
            self.outside = sp.diags(np.where(self.Mass.diagonal() > 0, 0.0, 1.0))
        self.stimuli = list(stimuli)
        self.state = np.zeros(mesh.num_nodes)
        self.v_ = np.zeros(mesh.num_nodes)
        self.time = 0.0
        self._dt = None
        self.solver = solver
        self.rtol = rtol
        self.last_its = 0
        self._set_dt(default_timestep)

    def _set_dt(self, dt):  # base_model.py:188-194
        self._dt = dt
        self.A = (self.C_m * self.Mass + self.theta * dt * self.K).tocsc()
        self.B = (self.C_m * self.Mass - (1.0 - self.theta) * dt * self.K).tocsr()
        if self.outside is not None:
            self.A = (self.A + self.outside).tocsc()
            self.B = (self.B + self.outside).tocsr()
        self._lu = spla.splu(self.A) if self.solver == "lu" else None

    def assign_previous(self):  # monodomain_model.py:59-60
        self.v_[:] = self.state

    def rhs(self, t, dt):
        b = self.B @ self.v_
        for s in self.stimuli:
            a = s.amp(t)
            if a != 0.0:
                b = b + dt * a * s.weights
        return b

    def step(self, interval):  # base_model.py:208-245
        t0, t1 = interval
        dt = t1 - t0
        t = t0 + self.theta * dt
        self.time = t
        if not abs(dt - self._dt) < 1e-12:
            self._set_dt(dt)
        b = self.rhs(t, dt)
        if self.solver == "lu":
            self.state[:] = self._lu.solve(b)
        else:
            x, its, _ = pcg_jacobi(self.A.tocsr(), b, self.v_, rtol=self.rtol)
            self.state[:] = x
            self.last_its = its

    def solve(self, interval, dt=None):  # base_model.py:250-297
        T0, T = interval
        if dt is None:
            dt = T - T0
        t0, t1 = T0, T0 + dt
        while True:
            self.step((t0, t1))
            if (t1 + dt) > (T + 1e-12):
                break
            self.assign_previous()
            t0 = t1
            t1 = t0 + dt
        return self.state
