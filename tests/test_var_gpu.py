"""GPU parity tests of the per-node-coefficient operators (beat_pde_create_var): voxel-masked domains and
spatially varying conductivity, against the oracle's literally assembled sparse FEM matrices
(oracle/fem.py assemble_mass / assemble_stiffness with per-cell tensors and cell subsets)."""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ptr_array(ptrs):
    arr = (C.c_void_p * max(1, len(ptrs)))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


def _dbl_array(vals):
    arr = (C.c_double * max(1, len(vals)))()
    for i, v in enumerate(vals):
        arr[i] = v
    return arr


def _shell_case(cells, L, seed):
    """Ellipsoidal shell of active box cells in a box + a rotating fibre field -> per-cell M."""
    rng = np.random.default_rng(seed)
    d = len(cells)
    cc = np.stack(np.meshgrid(*[(np.arange(c) + 0.5) / c for c in reversed(cells)], indexing="ij"), axis=-1)[..., ::-1]
    cc = cc.reshape(-1, d)  # cell centres in [0,1]^d, x fastest
    r = np.sqrt((((cc - 0.5) / 0.5) ** 2).sum(axis=1))
    active = (r < 0.95) & (r > 0.45)
    ang = np.pi * cc[:, 0] + 0.3 * rng.standard_normal(len(cc))
    f = np.zeros((len(cc), d))
    f[:, 0] = np.cos(ang)
    if d > 1:
        f[:, 1] = np.sin(ang)
    s_l, s_t = 9.5e-4, 1.25e-4
    M = s_t * np.eye(d)[None] + (s_l - s_t) * f[:, :, None] * f[:, None, :]
    return active, M


CASES = [((22, 17, 13), (2.2, 1.7, 1.3)), ((40, 31), (1.0, 1.0)), ((9, 4, 3), (0.9, 0.4, 0.6))]


def _oracle_matrices(cells, L, active, M):
    from oracle import fem

    mesh = fem.BoxMesh(cells, L)
    spc = {1: 1, 2: 2, 3: 6}[len(cells)]
    act_s = np.repeat(active, spc)
    Ms = np.repeat(M, spc, axis=0) * act_s[:, None, None]
    return mesh, fem.assemble_mass(mesh, np.nonzero(act_s)[0]), fem.assemble_stiffness(mesh, Ms)


@pytest.mark.parametrize("cells,L", CASES)
def test_per_node_operators_match_assembled_matrices(hip_ctx, cells, L):
    """A, B, Mass, K applied by the per-node kernels vs scipy CSR matrices assembled cell by cell over the
    active cells with per-cell tensors: <= 1e-13 * ||row||_1 * max|x|.  Rows of untouched nodes are identity
    for A and zero for Mass / K.  Ghost planes are poisoned with NaN."""
    from beat import _hip, _stencil
    from beat._device import Field
    from beat._engine import HipOps

    ctx = hip_ctx
    dim = len(cells)
    h = tuple(l / c for l, c in zip(L, cells))
    active, M = _shell_case(cells, L, 5)
    mesh, Mass, K = _oracle_matrices(cells, L, active, M)
    mf, kf = _stencil.stencil_fields(dim, cells, h, M, active)
    nn = [c + 1 for c in cells] + [1] * (3 - dim)
    ops = HipOps(ctx, nn, True, True, mf, kf, per_node=True)
    C_m, theta, dt = 0.01, 0.5, 0.05
    ops.set_timestep(C_m, theta, dt)
    n = mesh.num_nodes
    touched = Mass.diagonal() > 0
    assert 0 < touched.sum() < n
    ident = np.where(touched, 0.0, 1.0)
    import scipy.sparse as sp

    mats = {0: C_m * Mass + theta * dt * K + sp.diags(ident), 1: C_m * Mass - (1 - theta) * dt * K, 2: Mass, 3: K}
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n)
    fx, fy = ops.new_field(), ops.new_field()
    fx.set(x)
    fx.ghost_lo.fill_(float("nan"))
    fx.ghost_hi.fill_(float("nan"))
    for which, mat in mats.items():
        ops.apply(which, fx, fy)
        ctx.synchronize()
        ref = mat @ x
        scale = max((abs(mat) @ np.ones(n)).max(), 1.0) * np.abs(x).max()
        assert np.abs(fy.numpy() - ref).max() <= 1e-13 * scale, which


def test_per_node_rows_reproduce_the_table_path(hip_ctx):
    """With a constant tensor and no mask the per-node path equals the 27-type table path: operators to
    1e-15 relative, and the theta-step solve gives the same iterates (same iteration count, |dx| <= 1e-12)."""
    from beat import _stencil
    from beat._engine import HipOps

    ctx = hip_ctx
    cells, L = (33, 20, 11), (3.3, 2.0, 1.1)
    h = tuple(l / c for l, c in zip(L, cells))
    f0 = np.array([np.cos(0.5), np.sin(0.5), 0.0])
    M = 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
    nn = [c + 1 for c in cells]
    tab = HipOps(ctx, nn, True, True, *_stencil.stencil_tables(3, h, M))
    var = HipOps(ctx, nn, True, True, *_stencil.stencil_fields(3, cells, h, M), per_node=True)
    rng = np.random.default_rng(0)
    n = int(np.prod(nn))
    v = -85.0 + 100.0 * rng.random(n)
    out = []
    for ops in (tab, var):
        ops.set_timestep(0.01, 0.5, 0.05)
        fv, fx, fy = ops.new_field(), ops.new_field(), ops.new_field()
        fv.set(v)
        ys = []
        for which in range(4):
            ops.apply(which, fv, fy)
            ys.append(fy.numpy().copy())
        res = ops.solve_single(fv, [], [], fx, 1e-12, 1e-50, 500)
        out.append((ys, fx.numpy().copy(), res))
    for a, b in zip(out[0][0], out[1][0]):
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-15 * np.abs(a).max() * 16)
    assert out[0][2].iterations == out[1][2].iterations > 3
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=0, atol=1e-12 * 100)


@pytest.mark.parametrize("cells,L", CASES[:2])
def test_per_node_theta_step_matches_direct_solve(hip_ctx, cells, L):
    """One theta-step on the masked shell with a fibre field and a stimulus: rhs build + Jacobi-PCG
    (rtol 1e-12) vs sparse LU on the active sub-system: <= 1e-9 * max|v|; inactive nodes keep their value;
    same iteration count (+-1) as the oracle's restatement of the PCG."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    from beat import _stencil
    from beat._engine import HipOps
    from oracle import fem

    ctx = hip_ctx
    dim = len(cells)
    h = tuple(l / c for l, c in zip(L, cells))
    active, M = _shell_case(cells, L, 9)
    mesh, Mass, K = _oracle_matrices(cells, L, active, M)
    C_m, theta, dt = 0.01, 0.5, 0.05
    n = mesh.num_nodes
    touched = Mass.diagonal() > 0
    A = (C_m * Mass + theta * dt * K + sp.diags(np.where(touched, 0.0, 1.0))).tocsc()
    B = C_m * Mass - (1 - theta) * dt * K + sp.diags(np.where(touched, 0.0, 1.0))
    rng = np.random.default_rng(2)
    v_prev = np.where(touched, -85.0 + 100.0 * np.exp(-((mesh.x - 0.3 * np.array(L)) ** 2).sum(axis=1) / 0.05), 0.0)
    v_prev += np.where(touched, 0.01 * rng.standard_normal(n), 0.0)
    spc = {2: 2, 3: 6}[dim]
    stim_cells = np.nonzero(np.repeat(active, spc) & (mesh.x[mesh.cells].mean(axis=1)[:, 0] < 0.4 * L[0]))[0]
    w = fem.stimulus_weights(mesh, stim_cells)
    amp = 0.357
    b = B @ v_prev + dt * amp * w
    ref = spla.spsolve(A, b)
    mf, kf = _stencil.stencil_fields(dim, cells, h, M, active)
    nn = [c + 1 for c in cells] + [1] * (3 - dim)
    ops = HipOps(ctx, nn, True, True, mf, kf, per_node=True)
    ops.set_timestep(C_m, theta, dt)
    fv, fx, fw = ops.new_field(), ops.new_field(), ops.new_field()
    fv.set(v_prev)
    fw.set(w)
    res = ops.solve_single(fv, [fw], [amp], fx, 1e-12, 1e-50, 1000)
    out = fx.numpy()
    assert res.converged_reason > 0
    assert np.abs(out - ref).max() <= 1e-9 * np.abs(ref).max()
    np.testing.assert_array_equal(out[~touched], v_prev[~touched])
    _, its, _ = fem.pcg_jacobi(A.tocsr(), b, v_prev, rtol=1e-12)
    assert abs(its - res.iterations) <= 1
    # in place (fused split step): x aliases v_prev
    res2 = ops.solve_single(fv, [fw], [amp], fv, 1e-12, 1e-50, 1000)
    assert res2.iterations == res.iterations
    np.testing.assert_allclose(fv.numpy(), out, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("lo_phys,hi_phys,nzl", [(0, 0, 5), (1, 0, 4), (0, 1, 3), (0, 0, 1), (0, 0, 2)])
def test_per_node_spmv_in_two_parts_equals_whole(hip_ctx, lo_phys, hi_phys, nzl):
    """Slab rows cut out of a larger masked grid: interior part with poisoned ghosts + boundary part equals the
    one-shot SpMV and equals the corresponding rows of the global matrix."""
    from beat import _stencil
    from beat._engine import HipOps

    ctx = hip_ctx
    cells, L = (30, 12, 9), (3.0, 1.2, 0.9)
    h = tuple(l / c for l, c in zip(L, cells))
    active, M = _shell_case(cells, L, 4)
    mesh, Mass, K = _oracle_matrices(cells, L, active, M)
    nx, ny, nz = (c + 1 for c in cells)
    z0 = 0 if lo_phys else 3
    z1 = nz if hi_phys else z0 + nzl
    if hi_phys:
        z0 = nz - nzl
    plane = nx * ny
    mf, kf = _stencil.stencil_fields(3, cells, h, M, active, z_range=(z0, z1))
    ops = HipOps(ctx, (nx, ny, z1 - z0), bool(lo_phys), bool(hi_phys), mf, kf, per_node=True)
    C_m, theta, dt = 0.01, 0.5, 0.05
    ops.set_timestep(C_m, theta, dt)
    import scipy.sparse as sp

    touched = Mass.diagonal() > 0
    A = (C_m * Mass + theta * dt * K + sp.diags(np.where(touched, 0.0, 1.0))).tocsr()
    rng = np.random.default_rng(12)
    xg = rng.standard_normal(mesh.num_nodes)
    ref = (A @ xg)[z0 * plane : z1 * plane]
    p = ops.ring[0]
    p.set(xg[z0 * plane : z1 * plane])
    if z0 > 0:
        p.ghost_lo.copy_(ctx.from_numpy(xg[(z0 - 1) * plane : z0 * plane]))
    if z1 < nz:
        p.ghost_hi.copy_(ctx.from_numpy(xg[z1 * plane : (z1 + 1) * plane]))
    ops.spmv_dot()
    ctx.synchronize()
    q1 = ops.q.numpy().copy()
    pq = float(ops.st[3])
    np.testing.assert_allclose(q1, ref, rtol=0, atol=1e-13 * np.abs(ref).max() * 10)
    glo, ghi = p.ghost_lo.clone(), p.ghost_hi.clone()
    ops.q.fill(float("nan"))
    p.ghost_lo.fill_(float("nan"))
    p.ghost_hi.fill_(float("nan"))
    ops.spmv_interior(p)
    ctx.synchronize()
    p.ghost_lo.copy_(glo)
    p.ghost_hi.copy_(ghi)
    ops.spmv_boundary(p)
    ctx.synchronize()
    np.testing.assert_array_equal(ops.q.numpy(), q1)
    assert np.isclose(float(ops.st[3]), pq, rtol=1e-13)
