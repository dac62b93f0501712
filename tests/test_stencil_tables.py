"""CPU: the product's host-side stencil set-up (beat/_stencil.py, per-cell element matrices)
against the oracle's literal sparse assembly (oracle/fem.py)."""

import numpy as np
import pytest

from oracle import fem


def _aniso(dim):
    if dim == 3:
        f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.2])
        f0 /= np.linalg.norm(f0)
        return 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
    if dim == 2:
        return np.array([[2.0, 0.3], [0.3, 1.0]])
    return 1.7


@pytest.mark.parametrize("dim,h", [(1, (0.1,)), (2, (0.25, 0.5)), (3, (0.5, 0.25, 0.125))])
def test_tables_equal_oracle_derivation(dim, h):
    from beat import _stencil

    assert _stencil.OFFSETS == fem.STENCIL_OFFSETS
    mt, kt = _stencil.stencil_tables(dim, h, _aniso(dim))
    mo, ko = fem.stencil_table(dim, h, _aniso(dim), 1.0, 1.0)
    np.testing.assert_allclose(mt, mo, rtol=1e-13, atol=1e-18)
    np.testing.assert_allclose(kt, ko, rtol=1e-12, atol=1e-18)


@pytest.mark.parametrize("cells,L", [((5, 4, 3), (2.5, 2.0, 1.5)), ((4, 1, 2), (2.0, 0.5, 1.0)), ((6, 5), (3.0, 2.5)), ((9,), (1.0,))])
def test_tables_reproduce_assembled_operator(cells, L):
    from beat import _stencil

    dim = len(cells)
    mesh = fem.BoxMesh(cells, L)
    h = tuple(l / c for l, c in zip(L, cells))
    mt, kt = _stencil.stencil_tables(dim, h, _aniso(dim))
    x = np.random.default_rng(0).standard_normal(mesh.num_nodes)
    Mass, K = fem.assemble_mass(mesh), fem.assemble_stiffness(mesh, _aniso(dim))
    np.testing.assert_allclose(fem.apply_stencil(mt, mesh.shape_nodes, x), Mass @ x, rtol=0, atol=1e-13)
    np.testing.assert_allclose(fem.apply_stencil(kt, mesh.shape_nodes, x), K @ x, rtol=0, atol=1e-12 * abs(K).max() * 15)


def test_interior_row_sums():
    """Mass rows sum to the nodal volume, stiffness rows to zero (constants are in the kernel of K)."""
    from beat import _stencil

    mt, kt = _stencil.stencil_tables(3, (0.5, 0.5, 0.5), _aniso(3))
    assert abs(mt[13].sum() - 0.125) < 1e-15
    assert np.abs(kt.sum(axis=1)).max() < 1e-18
    # the corner on the shared v0-v7 diagonal belongs to all 6 tetrahedra of its only cell
    assert abs(mt[0].sum() - 0.125 / 4) < 1e-15
    assert abs(mt[26].sum() - 0.125 / 4) < 1e-15
    # the corner (hi, lo, lo) = v1 belongs to 2 of the 6
    assert abs(mt[2].sum() - 2 * 0.125 / 24) < 1e-15


def _apply_rows(rows, nn, x):
    """y_i = sum_k rows[k, i] x[i + offset_k] with zero padding (NumPy restatement of the per-node kernels)."""
    from beat import _stencil

    X = np.zeros((nn[2] + 2, nn[1] + 2, nn[0] + 2))
    X[1:-1, 1:-1, 1:-1] = x.reshape(nn[2], nn[1], nn[0])
    y = np.zeros((nn[2], nn[1], nn[0]))
    for k, (ox, oy, oz) in enumerate(_stencil.OFFSETS):
        y += rows[k].reshape(nn[2], nn[1], nn[0]) * X[1 + oz : 1 + oz + nn[2], 1 + oy : 1 + oy + nn[1], 1 + ox : 1 + ox + nn[0]]
    return y.ravel()


@pytest.mark.parametrize("cells,h", [((5, 4, 6), (0.1, 0.2, 0.15)), ((7, 5), (0.1, 0.3)), ((6,), (0.2,))])
def test_per_node_rows_equal_assembled_operator_on_masked_cells(cells, h):
    """stencil_fields (voxel mask + per-cell tensors; per box cell and per simplex) vs the oracle's assembly
    over the active cells; a z-range returns exactly the rows of those planes."""
    from beat import _stencil

    dim = len(cells)
    rng = np.random.default_rng(1)
    nbox = int(np.prod(cells))
    spc = {1: 1, 2: 2, 3: 6}[dim]
    mesh = fem.BoxMesh(cells, tuple(c * hh for c, hh in zip(cells, h)))
    nn = [c + 1 for c in cells] + [1] * (3 - dim)
    x = rng.standard_normal(mesh.num_nodes)
    for per_simplex in (False, True):
        ncell = nbox * spc if per_simplex else nbox
        B = rng.standard_normal((ncell, dim, dim))
        Mc = B @ B.transpose(0, 2, 1) + np.eye(dim)
        act = rng.random(ncell) > 0.3
        Ms = Mc if per_simplex else np.repeat(Mc, spc, axis=0)
        acts = act if per_simplex else np.repeat(act, spc)
        K = fem.assemble_stiffness(mesh, Ms * acts[:, None, None])
        Mass = fem.assemble_mass(mesh, np.nonzero(acts)[0])
        mf, kf = _stencil.stencil_fields(dim, cells, h, Mc, act)
        assert np.abs(_apply_rows(mf, nn, x) - Mass @ x).max() < 1e-14
        assert np.abs(_apply_rows(kf, nn, x) - K @ x).max() < 1e-12 * np.abs(K @ x).max()
        if dim == 3:
            pl = nn[0] * nn[1]
            m2, k2 = _stencil.stencil_fields(dim, cells, h, Mc, act, z_range=(2, 5))
            np.testing.assert_array_equal(m2, mf[:, 2 * pl : 5 * pl])
            np.testing.assert_array_equal(k2, kf[:, 2 * pl : 5 * pl])
    # constant tensor, no mask: the rows are the 27-type tables spread over the nodes
    mt, kt = _stencil.stencil_tables(dim, h, _aniso(dim))
    mf, kf = _stencil.stencil_fields(dim, cells, h, _aniso(dim))
    typ = fem.node_types(tuple(c + 1 for c in cells)).ravel()
    np.testing.assert_allclose(mf.T, mt[typ], rtol=1e-13, atol=1e-18)
    np.testing.assert_allclose(kf.T, kt[typ], rtol=1e-12, atol=1e-16)


def test_voxel_mesh_cells_and_nodes():
    """create_voxel_mesh: cell ids, locate_entities, node activity and stimulus weights respect the mask."""
    from beat import grid as g
    from beat.stimulation import assemble_weights

    mask = np.ones((3, 4, 5), dtype=bool)
    mask[:, :, 0] = False
    mask[1, 2, 3] = False
    mesh = g.create_voxel_mesh(g.COMM_WORLD, mask, 0.1)
    assert mesh.n == (5, 4, 3) and mesh.num_cells_global == 6 * int(mask.sum())
    assert np.array_equal(mesh.all_cells(), np.nonzero(np.repeat(mask.ravel(), 6))[0])
    act = mesh.node_active().reshape(4, 5, 6)
    assert not act[:, :, 0].any() and act[:, :, 1:].all()
    cells = g.locate_entities(mesh, 3, lambda x: x[0] <= 0.3 + 1e-12)
    assert len(cells) == 6 * int(mask[:, :, 1:3].sum())
    w = assemble_weights(mesh, None, None)
    omesh = fem.BoxMesh((5, 4, 3), (0.5, 0.4, 0.3))
    np.testing.assert_allclose(w, fem.stimulus_weights(omesh, mesh.all_cells()), rtol=1e-13, atol=1e-18)
    assert np.isclose(w.sum(), mask.sum() * 1e-3)
    centres = g.cell_centers(mesh)
    assert centres.shape == (60, 3) and np.allclose(centres[0], 0.05) and np.allclose(centres[1], [0.15, 0.05, 0.05])


def test_exterior_facet_weights_match_the_simplicial_mesh():
    """ds weights from box-cell faces (two triangles along the lowest-to-highest-corner diagonal) equal the weights
    computed from the exterior triangles / edges of the oracle's simplicial mesh: full box, voxel mask, and a
    vertex predicate (locate_entities_boundary)."""
    from beat import grid as g
    from beat.stimulation import assemble_facet_weights

    rng = np.random.default_rng(3)
    for cells, h in (((4, 3, 5), (0.5, 0.25, 0.2)), ((6, 5), (0.3, 0.2))):
        d = len(cells)
        mask = rng.random(tuple(reversed(cells))) > 0.35
        L = tuple(c * hh for c, hh in zip(cells, h))
        omesh = fem.BoxMesh(cells, L)
        spc = {2: 2, 3: 6}[d]
        for m in (None, mask):
            mesh = g.Mesh(cells, (0.0,) * d, L, active=None if m is None else m.ravel())
            act = None if m is None else np.repeat(m.ravel(), spc)
            w = assemble_facet_weights(mesh, mesh.exterior_facets())
            np.testing.assert_allclose(w, fem.exterior_facet_weights(omesh, act), rtol=1e-13, atol=1e-16)
            pred = lambda x: x[0] <= 0.6 * L[0]  # noqa: E731
            facets = g.locate_entities_boundary(mesh, d - 1, pred)
            xp = np.zeros((3, omesh.num_nodes))
            xp[:d] = omesh.x.T
            np.testing.assert_allclose(assemble_facet_weights(mesh, facets),
                                       fem.exterior_facet_weights(omesh, act, pred(xp)), rtol=1e-13, atol=1e-16)


def test_slabs_balanced_by_tissue_weight():
    """Slab(weights=...): contiguous, non-empty z-ranges covering all planes, cut where the cumulative tissue
    weight is closest to each rank's share (balance masked grids by active voxels per plane)."""
    from beat._engine import Slab

    w = np.array([0, 0, 1, 5, 9, 9, 5, 1, 0, 0, 0, 0], dtype=float)
    for world in (1, 2, 3, 4, 12):
        slabs = [Slab(12, r, world, w) for r in range(world)]
        assert slabs[0].z0 == 0 and slabs[-1].z1 == 12
        assert all(a.z1 == b.z0 for a, b in zip(slabs, slabs[1:])) and all(s.nz >= 1 for s in slabs)
    a, b = (Slab(12, r, 2, w) for r in range(2))
    assert w[a.z0 : a.z1].sum() == w[b.z0 : b.z1].sum() == 15.0
    even = [Slab(10, r, 3) for r in range(3)]
    assert [s.nz for s in even] == [4, 3, 3]
    with pytest.raises(ValueError):
        Slab(2, 0, 3)


def test_probe_weights_split_over_slabs_sum_to_the_global_interpolation():
    """evaluate_function on a decomposed mesh: every rank keeps only the vertices of its own slab; the partial
    sums add up to the global P1 interpolation (checked here without devices, slab by slab)."""
    from types import SimpleNamespace

    from beat import grid as g
    from beat._engine import Slab

    mesh = g.create_box(g.COMM_WORLD, [np.zeros(3), np.array([1.0, 1.0, 2.0])], [4, 4, 8])
    rng = np.random.default_rng(0)
    values = rng.standard_normal(mesh.num_nodes_global)
    idx = rng.integers(0, mesh.num_nodes_global, size=(6, 4))
    wts = rng.random((6, 4))
    total = (wts * values[idx]).sum(axis=1)
    parts = np.zeros(6)
    for r in range(3):
        slab = Slab(mesh.shape_global[2], r, 3)
        local = SimpleNamespace(slab=slab, plane=mesh.plane)
        li, lw = g._local_probe_args(local, idx, wts)
        owned = values[slab.z0 * mesh.plane : slab.z1 * mesh.plane]
        assert li.min() >= 0 and li.max() < len(owned)
        parts += (lw * owned[li]).sum(axis=1)
    np.testing.assert_allclose(parts, total, rtol=1e-14)


def test_voxelize_tetrahedra():
    """geometry.voxelize_tetrahedra: a box tetrahedral mesh rasterised at its own cell size gives a full mask of the
    same cell counts (+ padding ring); a subset of its cells gives exactly the voxels whose centres they contain, and
    per-tetrahedron data arrive in the voxels."""
    import beat
    from beat import grid as g

    om = fem.BoxMesh((4, 3, 2), (2.0, 1.5, 1.0))
    fib = np.zeros((len(om.cells), 3))
    fib[:, 0] = om.x[om.cells].mean(axis=1)[:, 0]  # a per-tet value that varies in x
    vg = beat.geometry.voxelize_tetrahedra(g.COMM_WORLD, om.x, om.cells, 0.5, cell_data={"f0": fib})
    assert vg.mask.shape == (4, 5, 6) and vg.mask[1:-1, 1:-1, 1:-1].all() and vg.mask.sum() == 24
    assert vg.mesh.n == (6, 5, 4) and np.allclose(vg.mesh.lower, -0.5)
    f0 = vg.cell_data["f0"].reshape(4, 5, 6, 3)
    centres_x = (np.arange(6) + 0.5) * 0.5 - 0.5
    inside = vg.mask
    # the containing tetrahedron's centroid lies in the same voxel column, within half a cell
    assert np.abs(f0[..., 0][inside] - np.broadcast_to(centres_x, (4, 5, 6))[inside]).max() <= 0.25 + 1e-12
    assert (vg.tet_index.reshape(4, 5, 6)[~inside] == -1).all()
    # a finer raster of the lower-left part: volume of the mask ~ volume of the selected tetrahedra
    sel = om.x[om.cells].mean(axis=1)[:, 0] < 1.0
    vg2 = beat.geometry.voxelize_tetrahedra(g.COMM_WORLD, om.x, om.cells[sel], 0.125)
    vol, _ = fem._cell_geometry(om)
    assert abs(vg2.mask.sum() * 0.125**3 - vol[sel].sum()) < 0.02 * vol[sel].sum()


def test_p2_and_dg1_ode_space_layouts():
    """Degrees of freedom of the P2 / DG1 ODE spaces: counts against the oracle's simplicial mesh (vertices + unique
    edges; (d+1) per cell), coordinates, and the maps used by utils.local_project (P1 interpolant at the dof points,
    one dof per vertex back)."""
    from beat import grid as g

    for cells in ((3, 2), (2, 2, 2)):
        d = len(cells)
        make = g.create_rectangle if d == 2 else g.create_box
        mesh = make(g.COMM_WORLD, [np.zeros(d), np.ones(d)], list(cells))
        om = fem.BoxMesh(cells, (1.0,) * d)
        edges = {(min(c[a], c[b]), max(c[a], c[b])) for c in om.cells for a in range(d + 1) for b in range(a + 1, d + 1)}
        V2, Vd = g.FunctionSpace(mesh, "CG", 2), g.FunctionSpace(mesh, "DG", 1)
        assert V2.num_dofs == om.num_nodes + len(edges) and Vd.num_dofs == len(om.cells) * (d + 1)
        (idx, w), to_p1 = V2.layout()
        assert {(min(a, b), max(a, b)) for a, b in idx[om.num_nodes:]} == edges and np.allclose(w.sum(axis=1), 1.0)
        xyz = mesh.node_coordinates(pad3=True, local=False)
        X2 = V2.tabulate_dof_coordinates()
        np.testing.assert_allclose(X2[: om.num_nodes], xyz)
        np.testing.assert_allclose(X2[om.num_nodes:], 0.5 * (xyz[idx[om.num_nodes:, 0]] + xyz[idx[om.num_nodes:, 1]]))
        (idx, w), to_p1 = Vd.layout()
        np.testing.assert_allclose(Vd.tabulate_dof_coordinates()[to_p1], xyz)
        # a linear function survives P1 -> space -> P1 exactly
        f = xyz @ np.array([0.3, -1.2, 0.7])
        for V in (V2, Vd):
            (idx, w), to_p1 = V.layout()
            on_space = w[:, 0] * f[idx[:, 0]] + w[:, 1] * f[idx[:, 1]]
            np.testing.assert_allclose(on_space, V.tabulate_dof_coordinates() @ np.array([0.3, -1.2, 0.7]), atol=1e-14)
            np.testing.assert_allclose(on_space[to_p1], f, atol=1e-14)
