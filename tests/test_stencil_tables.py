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
