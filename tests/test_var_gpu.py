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
    from beat import _stencil
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


def test_row_shifting_rhs_kernel_equals_the_gather_kernel_bit_for_bit():
    """var_rhs_kernel (rows of v_ and of the guess increment loaded once and shifted across the wave) against
    var_stencil_kernel<RHS> (one gather per stencil point, BEAT_VAR_RHS_GATHER=1): identical bits in r, p, the reduced
    scalars and every solution of two four-step sequences (without and with the extrapolated guess) on a masked shell with
    a random fibre field.  Two fresh interpreters, since the switch is read once per process."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    script = Path(__file__).with_name("_var_rhs_script.py")
    digests = []
    for gather in ("1", "0"):
        env = dict(os.environ, BEAT_VAR_RHS_GATHER=gather)
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert digests[0] == digests[1]


def test_tile_ordered_segment_list_gives_the_node_ordered_results(tmp_path):
    """The whole-slab launches of the per-node kernels walk the list of tissue segments in tiles of 8 rows x 8 planes, every
    XCD one contiguous eighth (the default since round 3; BEAT_VAR_TILE=0: node order, dealt round-robin).  Same segments, same
    per-node arithmetic: on a 111 x 75 x 67-node shell (0.36 M tissue nodes, ~9 k segments) q = A p is identical bit for
    bit on every node in both orders, nothing is written outside the tissue segments, p.q agrees to the rounding of its
    summation order, and a theta-step solve takes the same iterations to the same solution.  Two fresh interpreters."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    script = Path(__file__).with_name("_var_tile_script.py")
    res = {}
    for tile in ("0", "8"):
        out = tmp_path / f"tile{tile}.npz"
        run = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, BEAT_VAR_TILE=tile, BEAT_VTL="0"),
                             capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, run.stderr[-2000:]
        res[tile] = np.load(out)
    a, b = res["0"], res["8"]
    assert int(a["nseg_nodes"]) >= 4096 * 64 // 2  # large enough for the tile-ordered list to be the one in use
    written = np.isfinite(a["q"])
    np.testing.assert_array_equal(np.isfinite(b["q"]), written)
    np.testing.assert_array_equal(b["q"][written], a["q"][written])
    assert np.isclose(float(b["pq"]), float(a["pq"]), rtol=1e-12)
    assert int(a["its"]) == int(b["its"]) and int(a["its"]) > 3
    np.testing.assert_allclose(b["x"], a["x"], rtol=0, atol=1e-9 * np.abs(a["x"]).max())


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
    # nodes outside the tissue carry identity rows; wavefront-sized segments without any tissue are never touched
    # by the solver's kernels, so such a node holds either p (identity applied) or the 0 it was initialised with
    tl = touched[z0 * plane : z1 * plane]
    np.testing.assert_allclose(q1[tl], ref[tl], rtol=0, atol=1e-13 * np.abs(ref).max() * 10)
    outside = q1[~tl]
    assert np.all((outside == 0.0) | (outside == ref[~tl]))
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
    q2 = ops.q.numpy()
    written = ~np.isnan(q2)  # segments without tissue keep the NaN fill: nobody writes or reads them
    assert written[tl].all()
    np.testing.assert_array_equal(q2[written], q1[written])
    assert np.isclose(float(ops.st[3]), pq, rtol=1e-13)


@pytest.mark.parametrize("cells,L", CASES + [((70, 9, 40), (7.0, 0.9, 4.0))])
def test_marching_spmv_equals_the_stored_row_kernel_bit_for_bit(hip_ctx, cells, L, monkeypatch):
    """vrr_spmv_kernel (csrc/beat_pde_vrr.hip: marches along z, loads the forward half of each row and takes the backward
    coefficients from the neighbouring lane / row / plane) against var_spmv_kernel (all 15 coefficients gathered, BEAT_VRR=0)
    on masked grids with per-cell tensors, 3-D / 2-D, with 2 and 4 rows per wave: q identical bit for bit on every tissue
    node, nothing written elsewhere, p.q equal to the rounding of its summation order -- and a whole PCG solve agrees."""
    from beat import _stencil
    from beat._engine import HipOps

    ctx = hip_ctx
    d = len(cells)
    h = tuple(l / c for l, c in zip(L, cells))
    active, M = _shell_case(cells, L, 7)
    nn = tuple(c + 1 for c in cells) + (1,) * (3 - d)
    mf, kf = _stencil.stencil_fields(d, cells, h, M, active)
    rng = np.random.default_rng(3)
    n = int(np.prod(nn))
    x = rng.standard_normal(n)
    tissue = mf[0] != 0.0
    out = {}
    monkeypatch.setenv("BEAT_VTL", "0")  # the baseline is the segment-list kernel
    for key, env in (("rows", {"BEAT_VRR": "0"}), ("march2", {"BEAT_VRR": "1", "BEAT_VRR_RY": "2"}), ("march4", {"BEAT_VRR": "1", "BEAT_VRR_RY": "4"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ops = HipOps(ctx, nn, True, True, mf, kf, per_node=True)
        ops.set_timestep(0.01, 0.5, 0.05)
        ops.ring[0].set(np.where(tissue, x, 0.0))
        ops.ring[0].ghost_lo.fill_(float("nan"))
        ops.ring[0].ghost_hi.fill_(float("nan"))
        ops.q.fill(float("nan"))
        ops.st.zero_()
        ops.spmv_dot()
        ctx.synchronize()
        q = ops.q.numpy().copy()
        fv, fx = ops.new_field(), ops.new_field()
        fv.set(np.where(tissue, -80.0 + 20.0 * x, 0.0))
        res = ops.solve_single(fv, [], [], fx, 1e-11, 1e-50, 300)
        out[key] = (q, float(ops.st[3]), fx.numpy().copy(), res.iterations)
    q0, pq0, x0, it0 = out["rows"]
    assert np.isfinite(q0[tissue]).all() and np.isnan(q0[~tissue]).all() and abs(pq0) > 0.0
    for key in ("march2", "march4"):
        q1, pq1, x1, it1 = out[key]
        np.testing.assert_array_equal(q1[tissue], q0[tissue])
        assert np.isnan(q1[~tissue]).all()
        assert np.isclose(pq1, pq0, rtol=1e-12)
        assert abs(it1 - it0) <= 1
        np.testing.assert_allclose(x1, x0, rtol=0, atol=1e-9 * np.abs(x0).max())


TILE_CASES = CASES + [((70, 9, 40), (7.0, 0.9, 4.0)), ((61, 12, 35), (6.1, 1.2, 3.5)), ((130, 21, 9), (13.0, 2.1, 0.9)), ((5, 3, 70), (0.5, 0.3, 7.0))]


@pytest.mark.parametrize("cells,L", TILE_CASES)
def test_workgroup_tile_spmv_equals_the_stored_row_kernel_bit_for_bit(hip_ctx, cells, L, monkeypatch):
    """vtl_spmv_kernel (csrc/beat_pde_vtl.hip: a workgroup owns 62 x-nodes x 4 or 8 rows and marches along z; forward
    coefficients and rows of p loaded once per tile, shared through LDS and registers; raw-buffer loads with out-of-range
    offsets on the lanes outside the tissue) against var_spmv_kernel (all 15 coefficients gathered per 64-node segment,
    BEAT_VTL=0) on masked grids with per-cell tensors: q identical bit for bit on every tissue node, nothing written
    elsewhere (NaN-filled q, NaN ghost planes), p.q equal to the rounding of its summation order, and a whole PCG solve takes
    the same iterations to the same solution.  Grids with one and several x segments of 62 nodes, row blocks that end
    inside the box, runs of planes longer and shorter than a tile run (BEAT_VTL_RUN=5 cuts them every 5 planes); tiles dealt
    round-robin (the default) and taken from the per-XCD counters (BEAT_VTL_DYNAMIC=1: p.q is summed per tile in list order
    either way, so even the sum is the same bits); and the solver's iteration with the direction update fused into the tile
    pass (beat_vtl_pdot, the default) against the three-kernel iteration, bit for bit."""
    from beat import _stencil
    from beat._engine import HipOps

    ctx = hip_ctx
    d = len(cells)
    h = tuple(l / c for l, c in zip(L, cells))
    active, M = _shell_case(cells, L, 11)
    nn = tuple(c + 1 for c in cells) + (1,) * (3 - d)
    mf, kf = _stencil.stencil_fields(d, cells, h, M, active)
    rng = np.random.default_rng(5)
    n = int(np.prod(nn))
    x = rng.standard_normal(n)
    tissue = mf[0] != 0.0
    out = {}
    variants = (("rows", {"BEAT_VTL": "0"}), ("tile4", {"BEAT_VTL": "1", "BEAT_VTL_RY": "4", "BEAT_VTL_RUN": "32", "BEAT_VTL_DYNAMIC": "0"}),
                ("tile8", {"BEAT_VTL": "1", "BEAT_VTL_RY": "8", "BEAT_VTL_RUN": "32", "BEAT_VTL_DYNAMIC": "0"}),
                ("tile4short", {"BEAT_VTL": "1", "BEAT_VTL_RY": "4", "BEAT_VTL_RUN": "5", "BEAT_VTL_DYNAMIC": "0"}),
                ("tile8counter", {"BEAT_VTL": "1", "BEAT_VTL_RY": "8", "BEAT_VTL_RUN": "16", "BEAT_VTL_DYNAMIC": "1"}),
                ("tile8three", {"BEAT_VTL": "1", "BEAT_VTL_RY": "8", "BEAT_VTL_RUN": "32", "BEAT_VTL_DYNAMIC": "0", "BEAT_VTL_PDOT": "0"}),
                ("tile4three", {"BEAT_VTL": "1", "BEAT_VTL_RY": "4", "BEAT_VTL_RUN": "32", "BEAT_VTL_DYNAMIC": "0", "BEAT_VTL_PDOT": "0"}),
                ("tile4counter", {"BEAT_VTL": "1", "BEAT_VTL_RY": "4", "BEAT_VTL_RUN": "7", "BEAT_VTL_DYNAMIC": "1"}))
    monkeypatch.setenv("BEAT_VRR", "0")
    for key, env in variants:
        monkeypatch.setenv("BEAT_VTL_PDOT", "1")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ops = HipOps(ctx, nn, True, True, mf, kf, per_node=True)
        ops.set_timestep(0.01, 0.5, 0.05)
        ops.ring[0].set(np.where(tissue, x, 0.0))
        ops.ring[0].ghost_lo.fill_(float("nan"))
        ops.ring[0].ghost_hi.fill_(float("nan"))
        ops.q.fill(float("nan"))
        ops.st.zero_()
        ops.spmv_dot()
        ctx.synchronize()
        q = ops.q.numpy().copy()
        fv, fx = ops.new_field(), ops.new_field()
        fv.set(np.where(tissue, -80.0 + 20.0 * x, 0.0))
        res = ops.solve_single(fv, [], [], fx, 1e-11, 1e-50, 300)
        out[key] = (q, float(ops.st[3]), fx.numpy().copy(), res.iterations)
    q0, pq0, x0, it0 = out["rows"]
    assert np.isfinite(q0[tissue]).all() and np.isnan(q0[~tissue]).all() and abs(pq0) > 0.0
    for key, _ in variants[1:]:
        q1, pq1, x1, it1 = out[key]
        np.testing.assert_array_equal(q1[tissue], q0[tissue], err_msg=key)
        assert np.isnan(q1[~tissue]).all(), key
        assert np.isclose(pq1, pq0, rtol=1e-12), key
        assert abs(it1 - it0) <= 1, key
        np.testing.assert_allclose(x1, x0, rtol=0, atol=1e-9 * np.abs(x0).max(), err_msg=key)
    # the solver's fused pass -- p = D^-1 r + beta p_old formed while loading, stored, q = A p in the same pass (the default
    # of every tile variant above) -- against the three-kernel iteration over the same tiles (BEAT_VTL_PDOT=0): the same
    # expressions on the same values over the same tiles (same runs of planes: p.q is summed per tile), so the whole solve is
    # the same bits
    for fused, three in (("tile8", "tile8three"), ("tile4", "tile4three")):
        assert out[fused][3] == out[three][3] > 3, (out[fused][3], out[three][3])
        np.testing.assert_array_equal(out[fused][2], out[three][2], err_msg=fused)


@pytest.mark.parametrize("cells,L", [((70, 20, 40), (7.0, 2.0, 4.0)), ((22, 17, 13), (2.2, 1.7, 1.3))])
def test_right_hand_side_on_the_tiles_equals_the_gather_kernel(hip_ctx, cells, L, monkeypatch):
    """The right-hand side of a step in two single-window tile passes (csrc/beat_pde_vtl.hip, round 5, the default on single
    slabs: b = B v_ + dt stim over the rows of B = C_m Mass - (1 - theta) dt K, then r = b - A (v_ + e) with x0 formed while
    loading, z = D^-1 r where the three-kernel iteration wants it) against var_rhs_kernel (BEAT_VTL_RHS=0: r = dt (stim - K v_) - A e)
    on masked grids with per-cell tensors and a stimulus weight field.  The two are the same right-hand side in two
    formulations -- the constant-coefficient kernels use the first, beat_pde_rr.hip --: after a solve cut off before its first
    iteration r agrees to the rounding of b (1e-14 max |A v_|; |r| itself is 1e-3 of that), D^-1 r likewise (three-kernel
    iteration, BEAT_VTL_PDOT=0: the fused pass forms its first direction from r itself and the tile pass leaves z unwritten),
    ||b||^2, r.z and r.r to 1e-9; three consecutive solves -- the second and third start from the extrapolated guess, so
    A e is in the residual -- take the same iterations (+-1) to the same solutions (1e-11 of the scale at rtol 1e-11)."""
    from beat import _stencil
    from beat._engine import HipOps

    ctx = hip_ctx
    h = tuple(l / c for l, c in zip(L, cells))
    active, M = _shell_case(cells, L, 13)
    nn = tuple(c + 1 for c in cells)
    mf, kf = _stencil.stencil_fields(3, cells, h, M, active)
    rng = np.random.default_rng(9)
    n = int(np.prod(nn))
    tissue = mf[0] != 0.0
    v0 = np.where(tissue, -80.0 + 30.0 * rng.random(n), 0.0)
    wst = np.where(tissue, rng.random(n) * 1e-3, 0.0)
    out = {}
    monkeypatch.setenv("BEAT_VRR", "0")
    monkeypatch.setenv("BEAT_VTL", "1")
    av_max = None
    for key, rhs, pdot in (("gather", "0", "1"), ("tiles", "1", "1"), ("tiles-3k", "1", "0")):
        monkeypatch.setenv("BEAT_VTL_RHS", rhs)
        monkeypatch.setenv("BEAT_VTL_PDOT", pdot)
        ops = HipOps(ctx, nn, True, True, mf, kf, per_node=True)
        ops.set_guess_order(2)
        ops.set_timestep(0.01, 0.5, 0.05)
        fv, fx, fw = ops.new_field(), ops.new_field(), ops.new_field()
        fv.set(v0)
        fw.set(wst)
        if av_max is None:
            ops.apply(0, fv, fx)
            av_max = float(np.abs(fx.numpy()[tissue]).max())
        ops.ring[0].fill(0.0)
        try:
            ops.solve_single(fv, [fw], [0.7], fx, 1e-30, 1e-300, 0)  # the right-hand side alone
        except Exception:  # noqa: BLE001 -- "did not converge in 0 iterations"
            pass
        ctx.synchronize()
        st0 = ops.st.cpu().numpy().copy()
        r0, p0 = ops.r.numpy().copy(), ops.ring[0].numpy().copy()
        ops.guess_reset()
        sols, its = [], []
        for k in range(3):
            fv.set(v0 * (1.0 + 0.01 * k))
            res = ops.solve_single(fv, [fw], [0.7 + 0.1 * k], fx, 1e-11, 1e-50, 300)
            assert res.converged_reason > 0
            sols.append(fx.numpy().copy())
            its.append(res.iterations)
        out[key] = (r0, p0, st0, sols, its)
    a = out["gather"]
    assert np.abs(a[0][tissue]).max() > 1e-6 * av_max
    for key in ("tiles", "tiles-3k"):
        b = out[key]
        assert np.abs(b[0] - a[0])[tissue].max() <= 1e-14 * av_max, key
        np.testing.assert_allclose(b[2][:3], a[2][:3], rtol=1e-9)  # BB, RZ, RR
        assert all(abs(i - j) <= 1 for i, j in zip(a[4], b[4])) and b[4][1] < b[4][0]  # (the guess helps: fewer iterations in the second solve)
        for xa, xb in zip(a[3], b[3]):
            np.testing.assert_allclose(xb, xa, rtol=0, atol=1e-11 * np.abs(xa).max())
    # z = D^-1 r: written by the tile pass for the three-kernel iteration only
    dz = np.abs(out["tiles-3k"][1] - a[1])[tissue].max()
    assert np.abs(a[1][tissue]).max() > 0.0 and dz <= 1e-14 * np.abs(a[1][tissue]).max() * av_max / np.abs(a[0][tissue]).max()
    assert not out["tiles"][1][tissue].any()  # (with the fused pass the first direction is formed from r: z is never stored)
    np.testing.assert_array_equal(out["tiles"][0][tissue], out["tiles-3k"][0][tissue])


def test_default_route_of_an_undivided_masked_grid_is_the_tile_kernels_and_equals_the_assembled_matrix(hip_ctx, monkeypatch):
    """What ships BY DEFAULT for a per-node-row operator on an undivided grid -- asserted, not assumed (beat_pde_tile_route: tile
    product, fused pass, right-hand side on the tiles, ring of 12) -- held DIRECTLY against the matrix the oracle assembles cell by
    cell: q = A p of the tile product (beat_pde_spmv_dot) <= 1e-12 max|A p| on every tissue node, p.q with it; then one theta-step
    of the default solve (tile right-hand side + fused passes) against SciPy's sparse LU of the same matrix <= 1e-9 max|v|.
    The variants the other tests in this file compare bit for bit reach the oracle through this one."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    from beat import _stencil
    from beat._engine import HipOps

    for var in ("BEAT_VTL", "BEAT_VTL_RHS", "BEAT_VTL_PDOT", "BEAT_VTL_RY", "BEAT_VTL_RUN", "BEAT_VTL_DYNAMIC", "BEAT_VRR", "BEAT_VAR_RING",
                "BEAT_VAR_TILE"):
        monkeypatch.delenv(var, raising=False)
    ctx = hip_ctx
    cells, L = (70, 24, 19), (7.0, 2.4, 1.9)
    h = tuple(l / c for l, c in zip(L, cells))
    active, M = _shell_case(cells, L, 4)
    mesh, Mass, K = _oracle_matrices(cells, L, active, M)
    nn = [c + 1 for c in cells]
    C_m, theta, dt = 0.01, 0.5, 0.05
    touched = Mass.diagonal() > 0
    A = (C_m * Mass + theta * dt * K + sp.diags(np.where(touched, 0.0, 1.0))).tocsc()
    B = (C_m * Mass - (1 - theta) * dt * K).tocsr()
    mf, kf = _stencil.stencil_fields(3, cells, h, M, active)
    ops = HipOps(ctx, nn, True, True, mf, kf, per_node=True)
    ops.set_timestep(C_m, theta, dt)
    assert ctx.lib.beat_pde_tile_route(ops.handle) == 1 | 2 | 4 | 8
    rng = np.random.default_rng(21)
    x = np.where(touched, rng.standard_normal(mesh.num_nodes), 0.0)
    p = ops.ring[0]
    p.set(x)
    p.ghost_lo.fill_(float("nan"))
    p.ghost_hi.fill_(float("nan"))
    ops.q.fill(float("nan"))
    ops.st.zero_()
    ops.spmv_dot()
    ctx.synchronize()
    ref = A @ x
    q = ops.q.numpy()
    assert np.abs(q[touched] - ref[touched]).max() <= 1e-12 * np.abs(ref).max()
    assert np.isclose(float(ops.st[3]), float(x[touched] @ ref[touched]), rtol=1e-11)
    # one theta-step through the default solve against sparse LU
    v0 = np.where(touched, -80.0 + 30.0 * rng.random(mesh.num_nodes), 0.0)
    v = ops.new_field()
    v.set(v0)
    info = ops.solve_single(v, [], [], v, rtol=1e-12, atol=1e-50, max_it=500)
    ops.flush_pending()
    ctx.synchronize()
    exact = spla.splu(A).solve(B @ v0)
    got = v.numpy()
    assert info.converged_reason > 0 and info.iterations > 1
    assert np.abs(got[touched] - exact[touched]).max() <= 1e-9 * np.abs(exact).max()


def test_workgroup_tile_spmv_on_a_slab_with_live_ghost_planes(hip_ctx, monkeypatch):
    """The tile kernel on slab rows cut out of a larger masked grid, ghost planes holding the neighbouring slabs' p: plane 0
    takes its own stored backward coefficients (the plane below belongs to another rank), the last plane its forward ones;
    q equals the rows of the global matrix and the segment-list kernel bit for bit."""
    import scipy.sparse as sp

    from beat import _stencil
    from beat._engine import HipOps

    ctx = hip_ctx
    cells, L = (70, 12, 19), (7.0, 1.2, 1.9)
    h = tuple(l / c for l, c in zip(L, cells))
    active, M = _shell_case(cells, L, 4)
    mesh, Mass, K = _oracle_matrices(cells, L, active, M)
    nx, ny, nz = (c + 1 for c in cells)
    plane = nx * ny
    C_m, theta, dt = 0.01, 0.5, 0.05
    touched = Mass.diagonal() > 0
    A = (C_m * Mass + theta * dt * K + sp.diags(np.where(touched, 0.0, 1.0))).tocsr()
    rng = np.random.default_rng(12)
    xg = rng.standard_normal(mesh.num_nodes)
    monkeypatch.setenv("BEAT_VRR", "0")
    for z0, z1 in ((4, 13), (0, 7), (11, nz)):
        mf, kf = _stencil.stencil_fields(3, cells, h, M, active, z_range=(z0, z1))
        ref = (A @ xg)[z0 * plane : z1 * plane]
        tl = touched[z0 * plane : z1 * plane]
        qs = {}
        for vtl in ("0", "1"):
            monkeypatch.setenv("BEAT_VTL", vtl)
            ops = HipOps(ctx, (nx, ny, z1 - z0), z0 == 0, z1 == nz, mf, kf, per_node=True)
            ops.set_timestep(C_m, theta, dt)
            p = ops.ring[0]
            p.set(xg[z0 * plane : z1 * plane])
            if z0 > 0:
                p.ghost_lo.copy_(ctx.from_numpy(xg[(z0 - 1) * plane : z0 * plane]))
            else:
                p.ghost_lo.fill_(float("nan"))
            if z1 < nz:
                p.ghost_hi.copy_(ctx.from_numpy(xg[z1 * plane : (z1 + 1) * plane]))
            else:
                p.ghost_hi.fill_(float("nan"))
            ops.q.fill(float("nan"))
            ops.st.zero_()
            ops.spmv_dot()
            ctx.synchronize()
            qs[vtl] = (ops.q.numpy().copy(), float(ops.st[3]))
        np.testing.assert_allclose(qs["1"][0][tl], ref[tl], rtol=0, atol=1e-12 * np.abs(ref).max())
        np.testing.assert_array_equal(qs["1"][0][tl], qs["0"][0][tl])
        assert np.isnan(qs["1"][0][~tl]).all()
        assert np.isclose(qs["1"][1], qs["0"][1], rtol=1e-12)


# ---- API level: voxelised shell, per-cell fibres, transmural cell types (BASELINE configs[4] in small) ---------
def _shell_geometry(n=(26, 22, 18), h=0.5):
    """Truncated ellipsoidal shell voxelised on a box: active voxels, transmural depth in [0, 1] per voxel
    centre (0 = endocardium) and a fibre field rotating from +60 deg (endo) to -60 deg (epi) about the
    radial direction -- synthetic stand-in for demos/biv_endocardial.py's geometry."""
    cx, cy, cz = n
    ax = [(np.arange(c) + 0.5) * h for c in n]
    Z, Y, X = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    c = np.array([cx, cy, cz]) * h / 2.0
    semi_o = 0.48 * np.array(n) * h
    semi_i = 0.62 * semi_o
    P = np.stack([X - c[0], Y - c[1], Z - c[2]], axis=-1)
    ro = np.sqrt(((P / semi_o) ** 2).sum(-1))
    ri = np.sqrt(((P / semi_i) ** 2).sum(-1))
    mask = (ro < 1.0) & (ri > 1.0) & (Z < 0.8 * cz * h)
    depth = np.clip((ri - 1.0) / np.maximum(ri - ro, 1e-12) * 1.0, 0.0, 1.0)  # 0 at the inner, 1 at the outer surface
    rad = P / np.maximum(np.linalg.norm(P, axis=-1, keepdims=True), 1e-12)
    ez = np.array([0.0, 0.0, 1.0])
    circ = np.cross(ez, rad)
    circ /= np.maximum(np.linalg.norm(circ, axis=-1, keepdims=True), 1e-12)
    longi = np.cross(rad, circ)
    ang = np.deg2rad(60.0 - 120.0 * depth)[..., None]
    f0 = np.cos(ang) * circ + np.sin(ang) * longi
    return mask, depth.reshape(-1), f0.reshape(-1, 3)


def _tp06_variants():
    from beat.models import tp06

    base = dict(stim_amplitude=0.0)
    return {1: tp06.init_parameter_values(**base),                                     # endo-like
            0: tp06.init_parameter_values(g_Ks=0.098, **base),                         # mid
            2: tp06.init_parameter_values(g_to=0.073, g_Ks=0.392 * 1.2, **base)}       # epi-like


def test_voxel_shell_multi_celltype_split_step_matches_oracle():
    """Voxelised shell + per-voxel fibre field + three transmural cell types (DolfinMultiODESolver) + endocardial
    layer stimulus, 24 Godunov steps at dt = 0.05 ms: every state of every tissue node against the oracle
    (assembled FEM on the active cells with sparse-LU solves, NumPy TP06 GRL1 per marker) to 1e-7 relative (as
    test_strang_splitting_against_oracle: the stimulated layer is in its upstroke)."""
    import beat
    from beat import grid as g
    from beat.models import tp06
    from oracle import fem, ionic, splitting

    n, h = (26, 22, 18), 0.5
    mask, depth, f0 = _shell_geometry(n, h)
    mesh = g.create_voxel_mesh(g.COMM_WORLD, mask, h)
    assert mesh.active is not None and 0.1 < mask.mean() < 0.6
    cond = beat.conductivities.default_conductivities("Bishop")
    M = beat.conductivities.define_conductivity_tensor(f0=g.CellField(mesh, f0), **cond)
    assert isinstance(M, g.CellField) and M.values.shape == (mesh.num_box_cells, 3, 3)
    time = g.Constant(mesh, 0.0)
    # stimulus: active voxels of the innermost 25 % of the wall in the lower half (volume tags)
    vox_sel = mask.ravel() & (depth < 0.25) & (g.cell_centers(mesh)[:, 2] < 0.45 * n[2] * h)
    stim_cells = np.nonzero(np.repeat(vox_sel, 6))[0]
    tags = g.meshtags(mesh, 3, stim_cells, np.full(len(stim_cells), 7, dtype=np.int32))
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=tags, marker=7,
                                           mesh_unit="mm", amplitude=50_000.0, start=0.0, duration=1.0)
    C_m = 0.01
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=I_s, C_m=C_m, dx=I_s.dZ,
                               params={"petsc_options": {"ksp_rtol": 1e-13}})
    # nodal transmural markers: 1 endo / 0 mid / 2 epi from the mean depth of the active voxels around a node;
    # -1 outside the tissue (no cell model there)
    V = g.functionspace(mesh, ("P", 1))
    nx, ny, nz = mesh.shape_global
    dsum, cnt = np.zeros((nz, ny, nx)), np.zeros((nz, ny, nx))
    D, A = depth.reshape(n[2], n[1], n[0]), mask
    for oz in (0, 1):
        for oy in (0, 1):
            for ox in (0, 1):
                dsum[oz : oz + n[2], oy : oy + n[1], ox : ox + n[0]] += D * A
                cnt[oz : oz + n[2], oy : oy + n[1], ox : ox + n[0]] += A
    node_depth = np.where(cnt > 0, dsum / np.maximum(cnt, 1), -1.0).ravel()
    marker_arr = np.where(node_depth < 0, -1, np.where(node_depth < 0.3, 1, np.where(node_depth < 0.7, 0, 2))).astype(float)
    assert np.array_equal(marker_arr >= 0, mesh.node_active())
    markers = g.Function(V)
    markers.x.array[:] = marker_arr
    params = _tp06_variants()
    ic = tp06.init_state_values()
    keys = (1, 0, 2)
    ode = beat.odesolver.DolfinMultiODESolver(
        v_ode=g.Function(V), v_pde=pde.state, markers=markers, num_states={k: len(ic) for k in keys},
        fun={k: tp06.generalized_rush_larsen for k in keys}, init_states={k: ic for k in keys},
        parameters={k: params[k] for k in keys}, v_index={k: tp06.state_index("V") for k in keys})
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    dt, nsteps = 0.05, 24
    for i in range(nsteps):
        solver.step((i * dt, (i + 1) * dt))
    assert pde.ksp.iterations > 0

    # ---- oracle
    omesh = fem.BoxMesh(n, tuple(c * h for c in n))
    act_s = np.repeat(mask.ravel(), 6)
    w = fem.stimulus_weights(omesh, stim_cells)
    amp = 50000.0 / 1400.0 / 100.0  # uA/cm^3 / cm^-1 -> uA/cm^2 -> uA/mm^2
    model = fem.OracleMonodomainModel(omesh, np.repeat(M.values, 6, axis=0), [fem.OracleStimulus(fem.window(0.0, 1.0, amp), w)],
                                      C_m=C_m, theta=0.5, active_cells=act_s)
    oode = splitting.OracleMultiODE(marker_arr, {k: ionic.tp06_init_state_values() for k in keys},
                                    {k: np.asarray(params[k]) for k in keys},
                                    {k: ionic.tp06_generalized_rush_larsen for k in keys}, {k: 19 for k in keys},
                                    {k: 17 for k in keys})
    for i in range(nsteps):
        t0 = i * dt
        oode.step(t0, dt)
        oode.to_dolfin()
        model.state[:] = oode.v_ode
        model.assign_previous()
        model.step((t0, t0 + dt))
        oode.v_ode[:] = model.state
        oode.from_dolfin()
    for k in keys:
        out, ref = ode.values(k), oode.values[k]
        err = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)
        assert err.max() < 1e-7, (k, err.max())
    # the wave has started: stimulated layer depolarised, far tissue still at rest, outside untouched
    v = np.asarray(pde.state.x.array)
    assert v[marker_arr >= 0].max() > -40.0 and v[marker_arr >= 0].min() < -80.0
    assert np.all(v[marker_arr < 0] == 0.0)


def test_voxel_shell_torord_celltypes_runs():
    """Same geometry with ToR-ORd (45 states) endo / mid / epi through the generated kernel: 40 steps stay finite,
    the stimulated endocardial layer fires and the three cell types keep their own parameter sets."""
    import beat
    from beat import grid as g
    from beat.models import torord

    n, h = (20, 18, 14), 0.5
    mask, depth, f0 = _shell_geometry(n, h)
    mesh = g.create_voxel_mesh(g.COMM_WORLD, mask, h)
    cond = beat.conductivities.default_conductivities("Bishop")
    M = beat.conductivities.define_conductivity_tensor(f0=g.CellField(mesh, f0), **cond)
    time = g.Constant(mesh, 0.0)
    vox_sel = mask.ravel() & (depth < 0.3)
    stim_cells = np.nonzero(np.repeat(vox_sel, 6))[0]
    tags = g.meshtags(mesh, 3, stim_cells, np.full(len(stim_cells), 1, dtype=np.int32))
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=tags, marker=1,
                                           mesh_unit="mm", amplitude=50_000.0, start=0.0, duration=1.0)
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=I_s, C_m=0.01, dx=I_s.dZ)
    V = g.functionspace(mesh, ("P", 1))
    active = mesh.node_active()
    z = mesh.node_coordinates(pad3=True)[:, 2]
    marker_arr = np.where(~active, -1.0, np.where(z < 2.5, 0.0, np.where(z < 4.5, 1.0, 2.0)))
    markers = g.Function(V)
    markers.x.array[:] = marker_arr
    keys = (0, 1, 2)
    ic = torord.init_state_values()
    ode = beat.odesolver.DolfinMultiODESolver(
        v_ode=g.Function(V), v_pde=pde.state, markers=markers, num_states={k: len(ic) for k in keys},
        fun={k: torord.generalized_rush_larsen for k in keys}, init_states={k: ic for k in keys},
        parameters={k: torord.init_parameter_values(i_Stim_Amplitude=0.0, celltype=k) for k in keys},
        v_index={k: torord.state_index("v") for k in keys})
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    dt = 0.05
    for i in range(40):
        solver.step((i * dt, (i + 1) * dt))
    v = np.asarray(pde.state.x.array)
    assert np.isfinite(v).all()
    assert v[active].max() > 0.0 and v[active].min() < -80.0 and np.all(v[~active] == 0.0)
    for k in keys:
        assert np.isfinite(ode.values(k)).all()


def test_voxel_shell_torord_endocardial_pacing_matches_oracle():
    """BASELINE configs[4] in miniature, the workload of demos/biv_endocardial.py:187-282: voxelised shell, fibre
    rotation, endo / mid / epi layers from utils.expand_layer, ToR-ORd-dynCl with one parameter set per layer
    (DolfinMultiODESolver, generated 45-state kernel), endocardial SURFACE stimulus of 2000 uA/cm^2 for 1 ms, 24
    Godunov steps of 0.05 ms (the stimulus ends after step 20).  Every state of every tissue node against the oracle:
    the hand-written NumPy ToR-ORd (oracle/torord.py, independent of the kernel generator's parser) + literally
    assembled P1 FEM on the active simplices with sparse-LU solves + exterior-triangle stimulus weights; 1e-7
    relative as the TP06 twin above.  Also the explanation of the > 100 mV peaks seen at full size: they are the
    stimulus itself at convex staircase corners of the stimulated surface (largest surface-to-volume share), present
    bit for bit in the oracle, and gone a few ms after the stimulus ends."""
    import beat
    from beat import grid as g
    from beat.models import torord
    from oracle import fem, splitting
    from oracle import torord as otor

    n, h = (20, 18, 14), 0.5
    mask, depth, f0 = _shell_geometry(n, h)
    mesh = g.create_voxel_mesh(g.COMM_WORLD, mask, h)
    ft = _shell_facet_tags(mesh, n, h)
    V = g.functionspace(mesh, ("P", 1))
    tissue = mesh.node_active()
    layers = beat.utils.expand_layer(V, ft, 10, 20, endo_size=0.3, epi_size=0.3)  # 1 endo / 0 mid / 2 epi
    marker_arr = np.where(tissue, np.asarray(layers.x.array), -1.0)
    assert all((marker_arr == k).sum() > 50 for k in (0, 1, 2))
    markers = g.Function(V)
    markers.x.array[:] = marker_arr
    cond = beat.conductivities.default_conductivities("Bishop")
    M = beat.conductivities.define_conductivity_tensor(f0=g.CellField(mesh, f0), **cond)
    time = g.Constant(mesh, 0.0)
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=ft, marker=10,
                                           mesh_unit="mm", amplitude=2000.0, start=0.0, duration=1.0)
    assert I_s.dZ.integral_type == "ds"
    C_m = 0.01
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=I_s, C_m=C_m, params={"petsc_options": {"ksp_rtol": 1e-13}})
    keys = (1, 0, 2)
    celltype = {1: 0, 2: 1, 0: 2}  # layer marker -> the model's celltype parameter (0 endo, 1 epi, 2 mid)
    ic = torord.init_state_values()
    vi = torord.state_index("v")
    params = {k: torord.init_parameter_values(i_Stim_Amplitude=0.0, celltype=celltype[k]) for k in keys}
    ode = beat.odesolver.DolfinMultiODESolver(
        v_ode=g.Function(V), v_pde=pde.state, markers=markers, num_states={k: len(ic) for k in keys},
        fun={k: torord.generalized_rush_larsen for k in keys}, init_states={k: ic for k in keys},
        parameters=params, v_index={k: vi for k in keys})
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    dt, nsteps = 0.05, 24
    peak = []
    for i in range(nsteps):
        solver.step((i * dt, (i + 1) * dt))
        peak.append(float(np.asarray(pde.state.x.array)[tissue].max()))

    # ---- oracle ----------------------------------------------------------------------------------------------
    omesh = fem.BoxMesh(n, tuple(c * h for c in n))
    act_s = np.repeat(mask.ravel(), 6)
    endo_nodes = np.zeros(omesh.num_nodes, dtype=bool)
    endo_nodes[np.unique(mesh.facet_vertices(ft.find(10)))] = True
    zb = np.floor(0.8 * n[2]) * h  # triangles in the (untagged) base plane are not part of the endocardium
    w = fem.exterior_facet_weights(omesh, act_s, endo_nodes,
                                   facet_filter=lambda X: ~(np.abs(X[:, :, 2] - zb) < 1e-9).all(axis=1))
    amp = 2000.0 / 1400.0 / 10.0  # uA/cm^2 / cm^-1 = uA/cm -> uA/mm (stimulation.py:153-183)
    model = fem.OracleMonodomainModel(omesh, np.repeat(M.values, 6, axis=0), [fem.OracleStimulus(fem.window(0.0, 1.0, amp), w)],
                                      C_m=C_m, theta=0.5, active_cells=act_s)
    assert tuple(otor.TORORD_STATES) == torord.generalized_rush_larsen.state_names
    oparams = {k: otor.torord_init_parameter_values(i_Stim_Amplitude=0.0, celltype=float(celltype[k])) for k in keys}
    for k in keys:
        np.testing.assert_array_equal(oparams[k], np.asarray(params[k]))
    oode = splitting.OracleMultiODE(marker_arr, {k: otor.torord_init_state_values() for k in keys}, oparams,
                                    {k: otor.torord_generalized_rush_larsen for k in keys}, {k: 45 for k in keys},
                                    {k: vi for k in keys})
    opeak = []
    for i in range(nsteps):
        t0 = i * dt
        oode.step(t0, dt)
        oode.to_dolfin()
        model.state[:] = oode.v_ode
        model.assign_previous()
        model.step((t0, t0 + dt))
        oode.v_ode[:] = model.state
        oode.from_dolfin()
        opeak.append(float(model.state[tissue].max()))
    sdef = np.abs(otor.torord_init_state_values())[:, None]
    for k in keys:
        out, ref = ode.values(k), oode.values[k]
        assert np.isfinite(out).all()
        err = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6 * sdef + 1e-12)
        assert err.max() < 1e-7, (k, err.max(), np.unravel_index(err.argmax(), err.shape))
    v = np.asarray(pde.state.x.array)
    assert np.all(v[~tissue] == 0.0)
    # the stimulated surface fires, the far wall is still at rest; the three layers keep their own parameter sets
    assert v[tissue].max() > 0.0 and v[tissue].min() < -85.0
    # peaks: identical in the oracle at every step, attained on the stimulated surface, largest while the stimulus
    # is on (steps 1..20), and the largest of them sits on a node with (nearly) the largest surface-to-volume share
    np.testing.assert_allclose(peak, opeak, rtol=1e-9, atol=1e-7)
    i_max = int(np.argmax(np.where(tissue, v, -np.inf)))
    assert endo_nodes[i_max]
    assert int(np.argmax(peak)) < 21 and peak[19] > 40.0
    m_lumped = np.asarray(fem.assemble_mass(omesh, np.nonzero(act_s)[0]).sum(axis=1)).ravel()
    share = np.where(w > 0, w / np.maximum(m_lumped, 1e-300), 0.0)  # surface per volume of a node, 1/mm
    v20 = model.state  # (oracle potential after the last step, same as the device's to 1e-7)
    hot = np.argsort(share)[-max(1, int(0.02 * (share > 0).sum())):]
    assert v20[hot].mean() > v20[endo_nodes].mean() + 10.0
    # forcing alone, before any ionic or diffusive response: dv/dt = amp * share / C_m (mV/ms) -- the flat-surface
    # value 2 amp / (C_m h) = 57 mV/ms at h = 0.5 and up to three times that at a convex corner
    assert np.isclose(np.median(share[share > 0]) , 2.0 / h, rtol=0.5) and share.max() > 1.9 * 2.0 / h


@pytest.mark.parametrize("cells,h", [((22, 17, 13), (0.1, 0.12, 0.09)), ((40, 31), (0.05, 0.04)), ((15,), (0.1,))])
def test_device_row_assembly_equals_host_assembly(hip_ctx, cells, h):
    """beat_pde_assemble_rows (per-voxel tensors + mask, on the device) vs _stencil.stencil_fields (NumPy, itself
    checked against the oracle's assembly in test_stencil_tables.py): identical up to summation order; also for a
    z-slab of the grid and for a constant tensor without mask."""
    from beat import _stencil
    from beat._engine import HipOps

    ctx = hip_ctx
    dim = len(cells)
    L = tuple(c * hh for c, hh in zip(cells, h))
    active, M = _shell_case(cells, L, 7)
    nn = [c + 1 for c in cells] + [1] * (3 - dim)
    mf, kf = _stencil.stencil_fields(dim, cells, h, M, active)
    ops = HipOps.from_voxels(ctx, dim, cells, h, M, active, nn, 0, True, True)
    scale_k = np.abs(kf).max()
    np.testing.assert_allclose(ops._mass_dev.cpu().numpy(), mf, rtol=0, atol=1e-15 * np.abs(mf).max())
    np.testing.assert_allclose(ops._stiff_dev.cpu().numpy(), kf, rtol=0, atol=1e-14 * scale_k)
    if dim == 3:
        z0, z1 = 4, 9
        ops = HipOps.from_voxels(ctx, dim, cells, h, M, active, (nn[0], nn[1], z1 - z0), z0, False, False)
        pl = nn[0] * nn[1]
        np.testing.assert_allclose(ops._stiff_dev.cpu().numpy(), kf[:, z0 * pl : z1 * pl], rtol=0, atol=1e-14 * scale_k)
        np.testing.assert_allclose(ops._mass_dev.cpu().numpy(), mf[:, z0 * pl : z1 * pl], rtol=0, atol=1e-15 * np.abs(mf).max())
    Mc = M[len(M) // 2]
    mf, kf = _stencil.stencil_fields(dim, cells, h, Mc)
    ops = HipOps.from_voxels(ctx, dim, cells, h, Mc, None, nn, 0, True, True)
    np.testing.assert_allclose(ops._stiff_dev.cpu().numpy(), kf, rtol=0, atol=1e-14 * np.abs(kf).max())
    np.testing.assert_allclose(ops._mass_dev.cpu().numpy(), mf, rtol=0, atol=1e-15 * np.abs(mf).max())


def _shell_facet_tags(mesh, n, h):
    """Tag the exterior facets of the voxel shell: 10 = endocardial (inner) surface, 20 = epicardial (outer), base
    plane untagged.  Classified by the facet centre."""
    from beat import grid as g

    facets = mesh.exterior_facets()
    xyz = g._node_xyz(mesh, mesh.facet_vertices(facets).ravel()).reshape(len(facets), -1, 3)
    c = xyz.mean(axis=1)
    ctr = np.array(n) * h / 2.0
    semi_o = 0.48 * np.array(n) * h
    semi_i = 0.62 * semi_o
    ro = np.sqrt((((c - ctr) / semi_o) ** 2).sum(axis=1))
    ri = np.sqrt((((c - ctr) / semi_i) ** 2).sum(axis=1))
    base = np.ptp(xyz[:, :, 2], axis=1) < 1e-12
    base &= np.abs(xyz[:, 0, 2] - np.floor(0.8 * n[2]) * h) < 0.51 * h
    base &= (ri > 1.0) & (ro < 1.0)
    values = np.where(base, 0, np.where(np.abs(ri - 1.0) < np.abs(ro - 1.0) * (1.0 / 0.62), 10, 20)).astype(np.int32)
    keep = values > 0
    return g.meshtags(mesh, 2, facets[keep], values[keep])


def test_expand_layer_and_surface_stimulus_on_voxel_shell():
    """utils.expand_layer on the voxel shell (Laplace solve with Dirichlet data on tagged surface facets, on the
    device) against a sparse direct solve of the same P1 problem; a surface (ds) stimulus on the endocardial facets
    assembles the same nodal weights as the oracle's exterior-triangle integration and depolarises that surface."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    import beat
    from beat import grid as g
    from beat.models import tp06
    from oracle import fem

    n, h = (26, 22, 18), 0.5
    mask, depth, f0 = _shell_geometry(n, h)
    mesh = g.create_voxel_mesh(g.COMM_WORLD, mask, h)
    ft = _shell_facet_tags(mesh, n, h)
    assert len(ft.find(10)) > 100 and len(ft.find(20)) > 100
    V = g.functionspace(mesh, ("P", 1))
    u = beat.utils.laplace_dirichlet(V, [(ft.find(10), 0.0), (ft.find(20), 1.0)])
    # oracle: same problem by sparse LU on the free tissue nodes
    omesh = fem.BoxMesh(n, tuple(c * h for c in n))
    act_s = np.repeat(mask.ravel(), 6)
    K = fem.assemble_stiffness(omesh, np.eye(3)[None] * act_s[:, None, None]).tocsr()
    tissue = mesh.node_active()
    gvals = np.full(omesh.num_nodes, np.nan)
    gvals[np.unique(mesh.facet_vertices(ft.find(10)))] = 0.0
    gvals[np.unique(mesh.facet_vertices(ft.find(20)))] = 1.0
    dirichlet = ~np.isnan(gvals)
    free = tissue & ~dirichlet
    ref = np.zeros(omesh.num_nodes)
    ref[dirichlet] = gvals[dirichlet]
    rhs = -(K[free][:, dirichlet] @ gvals[dirichlet])
    ref[free] = spla.spsolve(K[free][:, free].tocsc(), rhs)
    out = np.asarray(u.x.array)
    assert np.abs(out - ref)[tissue].max() < 1e-7
    assert 0.0 <= out[tissue].min() and out[tissue].max() <= 1.0 + 1e-12
    layers = beat.utils.expand_layer(V, ft, 10, 20, endo_size=0.3, epi_size=0.3)
    lay = np.asarray(layers.x.array)
    sure = tissue & (np.abs(ref - 0.3) > 1e-6) & (np.abs(ref - 0.7) > 1e-6)
    expect = np.where(ref <= 0.3, 1, np.where(ref >= 0.7, 2, 0))
    assert np.array_equal(lay[sure], expect[sure])
    assert all((lay[tissue] == k).sum() > 50 for k in (0, 1, 2))

    # surface stimulus on the endocardium (demos/biv_endocardial.py:246-258): effective dimension 2
    cond = beat.conductivities.default_conductivities("Bishop")
    time = g.Constant(mesh, 0.0)
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=ft, marker=10,
                                           mesh_unit="mm", amplitude=2000.0, start=0.0, duration=1.0)
    assert I_s.dZ.integral_type == "ds"
    M = beat.conductivities.define_conductivity_tensor(f0=g.CellField(mesh, f0), **cond)
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=I_s, C_m=0.01)
    xp = np.zeros((3, omesh.num_nodes))
    xp[:] = omesh.x.T
    endo_nodes = np.zeros(omesh.num_nodes, dtype=bool)
    endo_nodes[np.unique(mesh.facet_vertices(ft.find(10)))] = True
    zb = np.floor(0.8 * n[2]) * h  # triangles lying in the (untagged) base plane are not part of the endocardium
    w_ref = fem.exterior_facet_weights(omesh, act_s, endo_nodes,
                                       facet_filter=lambda X: ~(np.abs(X[:, :, 2] - zb) < 1e-9).all(axis=1))
    w = pde._stimuli[0].field.numpy()
    np.testing.assert_allclose(w, w_ref, rtol=1e-12, atol=1e-15)
    time.value = 0.5
    amp = pde._stimuli[0].amplitude()
    assert np.isclose(amp, 2000.0 / 1400.0 / 10.0)  # uA/cm^2 / cm^-1 = uA/cm -> uA/mm (stimulation.py:153-183)
    ic = tp06.init_state_values()
    ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(V), v_pde=pde.state, fun=tp06.generalized_rush_larsen,
                                         init_states=ic, parameters=tp06.init_parameter_values(stim_amplitude=0.0),
                                         num_states=len(ic), v_index=tp06.state_index("V"))
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    for i in range(30):
        solver.step((i * 0.05, (i + 1) * 0.05))
    v = np.asarray(pde.state.x.array)
    assert v[endo_nodes].mean() > v[tissue & ~endo_nodes].mean() + 5.0 and np.isfinite(v).all()


def _square_layers(biv: bool):
    """The set-up of the reference's tests/test_utils.py: unit square, N = 50, facet tags on the left (and, for the
    bi-ventricular variant, bottom-left / top-left) and right edges; endo_size = epi_size = 0.3."""
    import beat
    from beat import grid as g

    N, tol = 50, 1.0e-8
    mesh = g.create_unit_square(g.COMM_WORLD, N, N, g.CellType.triangle)
    fdim = mesh.topology.dim - 1
    if biv:
        groups = [g.locate_entities_boundary(mesh, fdim, lambda x: np.logical_and(x[1] <= tol, x[0] <= 0.5 + tol)),
                  g.locate_entities_boundary(mesh, fdim, lambda x: np.logical_and(x[1] >= 1 - tol, x[0] <= 0.5 + tol)),
                  g.locate_entities_boundary(mesh, fdim, lambda x: x[0] >= 1 - tol)]
    else:
        groups = [g.locate_entities_boundary(mesh, fdim, lambda x: x[0] <= tol),
                  g.locate_entities_boundary(mesh, fdim, lambda x: x[0] >= 1 - tol)]
    facets = np.hstack(groups)
    values = np.hstack([np.full(len(f), k + 1) for k, f in enumerate(groups)])
    order = np.argsort(facets)
    ft = g.meshtags(mesh, fdim, facets[order], values[order])
    V = g.functionspace(mesh, ("Lagrange", 1))
    kw = dict(V=V, ft=ft, endo_size=0.3, epi_size=0.3, output_mid_marker=4, output_endo_marker=3, output_epi_marker=1)
    if biv:
        return beat.utils.expand_layer_biv(endo_lv_marker=1, endo_rv_marker=2, epi_marker=3, **kw)
    return beat.utils.expand_layer(endo_marker=1, epi_marker=2, **kw)


def test_expand_layer_single():
    """tests/test_utils.py:10-69 of the reference."""
    from beat import grid as g

    markers = _square_layers(False)
    points = np.array([(x, y) for x in [0.0, 0.1, 0.2] for y in [0.0, 0.5, 1.0]])
    assert np.allclose(g.evaluate_function(markers, points), 3)
    assert np.allclose(g.evaluate_function(markers, points + np.array([0.4, 0.0])), 4)
    assert np.allclose(g.evaluate_function(markers, points + np.array([0.8, 0.0])), 1)


def test_expand_layer_biv():
    """tests/test_utils.py:72-149 of the reference."""
    from beat import grid as g

    markers = _square_layers(True)
    endo_points = np.array([(0.0, 0.0), (0.0, 1.0), (0.2, 0.2), (0.2, 0.8)])
    mid_points = np.array([(0.5 + i, 0.5 + j) for i in [-0.1, 0.0, 0.1] for j in [-0.1, 0.0, 0.1]] + [(0.0, 0.5)])
    epi_points = np.array([(1.0, 0.0), (1.0, 1.0), (0.8, 0.2), (0.8, 0.8)])
    assert np.allclose(g.evaluate_function(markers, endo_points), 3)
    assert np.allclose(g.evaluate_function(markers, mid_points), 4)
    assert np.allclose(g.evaluate_function(markers, epi_points), 1)


# ---- UFL expressions on surface measures; fibre fields given as vector P1 functions ---------------------------------
@pytest.mark.parametrize("dim", [2, 3])
def test_coordinate_dependent_surface_stimulus_matches_oracle(dim):
    """stimulation.py:14-24 / base_model.py:247-248: a Stimulus is ANY expression times the test function on a measure,
    also a coordinate-dependent one on a surface (``ds``) measure.  Separable expression f(x) g(t): the nodal load
    equals the oracle's Gauss integration of f phi_i over the exterior simplices (independent quadrature code), and
    three theta-steps match the oracle's assembled solve; an expression that mixes x and t inside one factor is
    re-integrated every step."""
    import beat
    from beat import grid as g
    from oracle import fem

    if dim == 2:
        cells, L = (14, 9), (1.4, 0.9)
        mesh = g.create_rectangle(g.COMM_WORLD, [(0.0, 0.0), L], cells)
    else:
        cells, L = (9, 7, 5), (0.9, 0.7, 0.5)
        mesh = g.create_box(g.COMM_WORLD, [(0.0, 0.0, 0.0), L], cells)
    fdim = mesh.topology.dim - 1
    facets = g.locate_entities_boundary(mesh, fdim, lambda x: x[0] <= 1e-10)  # the face x = 0
    ft = g.meshtags(mesh, fdim, np.sort(facets), np.full(len(facets), 3, dtype=np.int32))
    ds = g.Measure("ds", domain=mesh, subdomain_data=ft)  # restricted to the tagged face by the Stimulus' marker
    time = g.Constant(mesh, 0.0)
    x = g.SpatialCoordinate(mesh)
    t = g.variable(time)
    spatial = 1.0 + g.sin(3.0 * x[1]) * (x[dim - 1] + 0.25)
    expr = spatial * (2.0 + g.cos(t))
    M = 0.02 * np.eye(dim) + 0.01 * np.ones((dim, dim))
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=beat.stimulation.Stimulus(expr=expr, dZ=ds, marker=3), C_m=1.0,
                               params={"petsc_options": {"ksp_rtol": 1e-13}})
    omesh = fem.BoxMesh(cells, L)
    on_face = np.abs(omesh.x[:, 0]) < 1e-10
    f = lambda X: 1.0 + np.sin(3.0 * X[1]) * (X[dim - 1] + 0.25)  # noqa: E731
    w_ref = fem.exterior_facet_load(omesh, f, node_ok=on_face)
    w = pde._stimuli[0].field.numpy()
    assert np.abs(w_ref).max() > 0 and np.all(w[~on_face] == 0.0)
    np.testing.assert_allclose(w, w_ref, rtol=1e-11, atol=1e-15)
    model = fem.OracleMonodomainModel(omesh, M, [fem.OracleStimulus(lambda tt: 2.0 + np.cos(tt), w_ref)], C_m=1.0, theta=0.5)
    dt = 0.05
    for i in range(3):
        pde.step((i * dt, (i + 1) * dt))
        pde.assign_previous()
        model.step((i * dt, (i + 1) * dt))
        model.assign_previous()
    v = np.asarray(pde.state.x.array)
    assert np.abs(v).max() > 1e-3
    np.testing.assert_allclose(v, model.state, rtol=0, atol=1e-11 * np.abs(model.state).max())

    # x and t inside one factor: no separation possible, the facet integral is taken again at every step
    time2 = g.Constant(mesh, 0.0)
    expr2 = g.sin(2.0 * x[1] + g.variable(time2))
    pde2 = beat.MonodomainModel(time=time2, mesh=mesh, M=M, I_s=beat.stimulation.Stimulus(expr=expr2, dZ=ds, marker=3), C_m=1.0,
                                params={"petsc_options": {"ksp_rtol": 1e-13}})
    model2 = fem.OracleMonodomainModel(omesh, M, [], C_m=1.0, theta=0.5)
    for i in range(3):
        tm = (i + 0.5) * dt  # theta = 0.5: the form is evaluated at t0 + theta dt (base_model.py:216-223)
        model2.stimuli = [fem.OracleStimulus(lambda tt: 1.0, fem.exterior_facet_load(omesh, lambda X, tm=tm: np.sin(2.0 * X[1] + tm),
                                                                                     node_ok=on_face))]
        pde2.step((i * dt, (i + 1) * dt))
        pde2.assign_previous()
        model2.step((i * dt, (i + 1) * dt))
        model2.assign_previous()
    np.testing.assert_allclose(np.asarray(pde2.state.x.array), model2.state, rtol=0, atol=1e-11 * np.abs(model2.state).max())


@pytest.mark.parametrize("dim", [2, 3])
def test_nodal_fibre_function_conductivity_matches_quadrature_assembly(hip_ctx, dim):
    """conductivities.py:101-118 with ``f0`` a vector P1 FUNCTION (the reference's ``geo.f0``): M(x) = s_l f f^T +
    s_t (I - f f^T) varies inside every simplex.  The operators (per-node rows from the closed-form cell average of
    the tensor) equal the oracle's stiffness matrix integrated by quadrature at the points of every simplex, and a
    theta-step matches the assembled sparse solve."""
    import beat
    from beat import grid as g
    from oracle import fem

    ctx = hip_ctx
    if dim == 2:
        cells, L = (12, 10), (1.2, 1.0)
        mesh = g.create_rectangle(g.COMM_WORLD, [(0.0, 0.0), L], cells)
    else:
        cells, L = (8, 7, 6), (0.8, 0.7, 0.6)
        mesh = g.create_box(g.COMM_WORLD, [(0.0, 0.0, 0.0), L], cells)
    V = g.functionspace(mesh, ("P", 1, (dim,)))
    f0 = g.Function(V)

    def fibres(x):  # rotating, normalised at the nodes
        ang = 0.4 + 2.0 * x[0] + 1.5 * x[1] * x[1] + (0.7 * x[2] if dim == 3 else 0.0)
        comp = [np.cos(ang), np.sin(ang)] + ([0.3 * np.sin(3.0 * x[0])] if dim == 3 else [])
        F = np.array(comp)
        return F / np.linalg.norm(F, axis=0)

    f0.interpolate(fibres)
    assert f0.x.array.shape == (mesh.num_nodes_global * dim,)
    cond = beat.conductivities.default_conductivities("Niederer")
    s_l, s_t = beat.conductivities.get_harmonic_mean_conductivity(**cond)
    M = beat.conductivities.define_conductivity_tensor(f0=f0, **cond)
    time = g.Constant(mesh, 0.0)
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, C_m=0.01, params={"petsc_options": {"ksp_rtol": 1e-13}})
    omesh = fem.BoxMesh(cells, L)
    np.testing.assert_allclose(omesh.x, mesh.node_coordinates(pad3=False), atol=1e-14)
    K = fem.assemble_stiffness_nodal_fibres(omesh, fibres(omesh.x.T).T, s_l, s_t, m=4)
    Mass = fem.assemble_mass(omesh)
    ops = pde._ops
    n = omesh.num_nodes
    rng = np.random.default_rng(8)
    x = rng.standard_normal(n)
    fx, fy = ops.new_field(), ops.new_field()
    fx.set(x)
    ops.apply(3, fx, fy)
    ctx.synchronize()
    ref = K @ x
    assert np.abs(fy.numpy() - ref).max() <= 1e-12 * (abs(K) @ np.ones(n)).max() * np.abs(x).max()
    # a constant fibre field would give a different operator: the variation inside the cells matters
    K0 = fem.assemble_stiffness(omesh, s_l * np.outer(fibres(np.zeros((3, 1)))[:, 0], fibres(np.zeros((3, 1)))[:, 0])
                                + s_t * (np.eye(dim) - np.outer(fibres(np.zeros((3, 1)))[:, 0], fibres(np.zeros((3, 1)))[:, 0])))
    assert np.abs((K0 - K) @ x).max() > 1e-3 * np.abs(ref).max()
    # one theta-step against the sparse direct solve
    import scipy.sparse.linalg as spla

    v0 = -80.0 + 30.0 * np.exp(-((omesh.x - 0.4) ** 2).sum(axis=1) / 0.05)
    pde.v_.x.array[:] = v0
    dt, theta, C_m = 0.05, 0.5, 0.01
    pde.step((0.0, dt))
    A = (C_m * Mass + theta * dt * K).tocsc()
    b = (C_m * Mass - (1 - theta) * dt * K) @ v0
    np.testing.assert_allclose(np.asarray(pde.state.x.array), spla.spsolve(A, b), rtol=0, atol=1e-9 * np.abs(v0).max())
