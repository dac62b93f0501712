"""CPU, world_size 2 and 3 on gloo: the product's slab decomposition, halo exchange and distributed
PCG orchestration (beat/_engine.py) with oracle-backed kernels, against a single-process sparse-LU
solve of the undivided problem."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    """A port for a rendezvous on this host, from BELOW the kernel's ephemeral range (32768-60999): a port handed out by
    bind(0) can be taken by any process's outgoing connection between this probe and the launcher's own bind -- seen once as
    EADDRINUSE from torchrun's TCPStore in the GPU suite -- while nothing but another listener takes one of these."""
    import random
    import socket

    for _ in range(64):
        port = random.randint(20000, 32000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


CELLS, L = (9, 6, 10), (0.9, 0.6, 1.0)
C_M, THETA, DT, AMP = 0.01, 0.5, 0.05, 0.357


def _problem():
    from oracle import fem

    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M = 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
    mesh = fem.BoxMesh(CELLS, L)
    rng = np.random.default_rng(5)
    v_prev = -85.0 + 100.0 * np.exp(-((mesh.x - 0.3) ** 2).sum(axis=1) / 0.05) + 0.01 * rng.standard_normal(mesh.num_nodes)
    w = fem.stimulus_weights(mesh, mesh.locate_cells(lambda x: x[2] <= 0.55))  # straddles the slab cut
    return mesh, M, v_prev, w


def _worker(rank, world, port, out_dir, degree):
    for p in (str(ROOT), str(ROOT / "fenicsx-beat_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _oracle_ops import CpuField, OracleOps
        from beat import _stencil
        from beat._engine import DiffusionSolver, Slab

        mesh, M, v_prev, w = _problem()
        nx, ny, nz = mesh.shape_nodes
        plane = nx * ny
        slab = Slab(nz, rank, world)
        mt, kt = _stencil.stencil_tables(3, tuple(l / c for l, c in zip(L, CELLS)), M)
        ops = OracleOps((nx, ny, slab.nz), slab.lo_phys, slab.hi_phys, mt, kt)
        ops.set_preconditioner(degree)
        ops.set_timestep(C_M, THETA, DT)
        solver = DiffusionSolver(ops, slab)
        sl = slice(slab.z0 * plane, slab.z1 * plane)
        fv, fx, fw = (CpuField(ops.n, plane) for _ in range(3))
        fv.data.copy_(torch.from_numpy(v_prev[sl].copy()))
        fw.data.copy_(torch.from_numpy(w[sl].copy()))
        res = solver.solve(fv, [fw], [AMP], fx, rtol=1e-12, atol=1e-50, max_it=300)
        # in-place variant (v_prev and the unknown share storage), as the fused split step uses it
        res2 = solver.solve(fv, [fw], [AMP], fv, rtol=1e-12, atol=1e-50, max_it=300)
        np.savez(Path(out_dir) / f"rank{rank}.npz", x=fx.numpy(), x_inplace=fv.numpy(), its=res.iterations,
                 its2=res2.iterations, reason=res.converged_reason, z0=slab.z0, z1=slab.z1, rnorm=res.residual_norm,
                 bnorm=res.rhs_norm)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,degree", [(2, 1), (3, 1), (2, 3)])
def test_slab_decomposed_pcg_matches_undivided_solve(world, degree, tmp_path):
    from oracle import fem

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), degree), nprocs=world, join=True)
    mesh, M, v_prev, w = _problem()
    model = fem.OracleMonodomainModel(mesh, M, [fem.OracleStimulus(lambda t: AMP, w)], C_m=C_M, theta=THETA,
                                      default_timestep=DT)
    model.state[:] = v_prev
    model.assign_previous()
    model.step((0.0, DT))
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert [int(p["z0"]) for p in parts] == [0] + [int(p["z1"]) for p in parts[:-1]]
    assert int(parts[-1]["z1"]) == mesh.shape_nodes[2]
    x = np.concatenate([p["x"] for p in parts])
    assert np.abs(x - model.state).max() <= 1e-9 * np.abs(model.state).max()
    np.testing.assert_allclose(np.concatenate([p["x_inplace"] for p in parts]), x, rtol=1e-12)
    its = {int(p["its"]) for p in parts} | {int(p["its2"]) for p in parts}
    assert len(its) == 1 and all(int(p["reason"]) > 0 for p in parts)  # every rank latched at the same iteration
    # same iteration count and norms as the undivided PCG
    A = (C_M * fem.assemble_mass(mesh) + THETA * DT * fem.assemble_stiffness(mesh, M)).tocsr()
    b = model.rhs(THETA * DT, DT)
    if degree == 1:
        _, its_ref, rn = fem.pcg_jacobi(A, b, v_prev, rtol=1e-12)
    else:
        from beat import _stencil
        from beat._engine import chebyshev_coefficients, spectrum_bounds

        mt, kt = _stencil.stencil_tables(3, tuple(l / c for l, c in zip(L, CELLS)), M)
        coef = chebyshev_coefficients(degree, *spectrum_bounds(C_M * mt + THETA * DT * kt))
        _, its_ref, rn = fem.pcg_polynomial(A, b, v_prev, coef, rtol=1e-12)
    assert abs(its.pop() - its_ref) <= 1
    assert np.isclose(float(parts[0]["bnorm"]), np.linalg.norm(b), rtol=1e-12)


def _parity_worker(rank, world, port, out_dir, break_rank):
    """bench.py's multi-rank parity block (bench_parity.compare) on gloo: three theta-steps of the decomposed diffusion solve
    with oracle-backed operators as the decomposed run, the oracle's undivided model on rank 0 as the reference."""
    for p in (str(ROOT), str(ROOT / "fenicsx-beat_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if break_rank is not None:
        os.environ["BEAT_BENCH_TEST_PARITY_BREAK"] = str(break_rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json

        import bench_parity
        from _oracle_ops import CpuField, OracleOps
        from beat import _stencil
        from beat._engine import DiffusionSolver, Slab
        from oracle import fem

        mesh, M, v_prev, w = _problem()
        nx, ny, nz = mesh.shape_nodes
        plane = nx * ny
        slab = Slab(nz, rank, world)
        mt, kt = _stencil.stencil_tables(3, tuple(l / c for l, c in zip(L, CELLS)), M)
        ops = OracleOps((nx, ny, slab.nz), slab.lo_phys, slab.hi_phys, mt, kt)
        ops.set_timestep(C_M, THETA, DT)
        solver = DiffusionSolver(ops, slab)
        sl = slice(slab.z0 * plane, slab.z1 * plane)
        fv, fw = CpuField(ops.n, plane), CpuField(ops.n, plane)
        fv.data.copy_(torch.from_numpy(v_prev[sl].copy()))
        fw.data.copy_(torch.from_numpy(w[sl].copy()))
        its = []
        for _ in range(3):
            its.append(int(solver.solve(fv, [fw], [AMP], fv, rtol=1e-12, atol=1e-50, max_it=300).iterations))

        def undivided():
            model = fem.OracleMonodomainModel(mesh, M, [fem.OracleStimulus(lambda t: AMP, w)], C_m=C_M, theta=THETA, default_timestep=DT)
            A = (C_M * fem.assemble_mass(mesh) + THETA * DT * fem.assemble_stiffness(mesh, M)).tocsr()
            model.state[:] = v_prev
            ref_its = []
            for k in range(3):
                model.assign_previous()
                ref_its.append(int(fem.pcg_jacobi(A, model.rhs(THETA * DT, DT), model.state.copy(), rtol=1e-12)[1]))
                model.step((k * DT, (k + 1) * DT))
            return model.state.copy(), ref_its

        verdict = bench_parity.compare(dist, rank, world, fv.numpy(), its, undivided)
        (Path(out_dir) / f"verdict{rank}.json").write_text(json.dumps(verdict))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("break_rank", [None, 1])
def test_bench_parity_block_compares_decomposed_with_undivided(break_rank, tmp_path):
    """What ``bench.py --gpus N`` reports as ``multi_rank_parity`` (bench_parity.compare): the slabs gathered on rank 0, the
    undivided run there, max |v_N - v_1| / max |v_1| against bench_parity.TOLERANCE, the same verdict on every rank -- and a
    slab that is off (BEAT_BENCH_TEST_PARITY_BREAK) is caught."""
    import json

    import bench_parity

    world = 2
    mp.spawn(_parity_worker, args=(world, _free_port(), str(tmp_path), break_rank), nprocs=world, join=True)
    verdicts = [json.loads((tmp_path / f"verdict{r}.json").read_text()) for r in range(world)]
    assert verdicts[0] == verdicts[1]
    v = verdicts[0]
    assert v["nodes"] == int(np.prod([c + 1 for c in CELLS])) and v["steps"] == 3 and v["finite"]
    assert v["iterations_equal_across_ranks"] and abs(v["k"] - v["k_undivided"]) <= 1.0
    if break_rank is None:
        assert v["ok"] and v["max_rel_diff"] <= bench_parity.TOLERANCE
    else:
        assert not v["ok"] and v["max_rel_diff"] > 1e-4


def test_slab_partition():
    from beat._engine import Slab

    for nz, world in ((512, 8), (7, 3), (10, 4)):
        slabs = [Slab(nz, r, world) for r in range(world)]
        assert slabs[0].z0 == 0 and slabs[-1].z1 == nz
        assert all(a.z1 == b.z0 for a, b in zip(slabs[:-1], slabs[1:]))
        assert max(s.nz for s in slabs) - min(s.nz for s in slabs) <= 1
        assert slabs[0].lo_phys and slabs[-1].hi_phys and not slabs[0].hi_phys
    with pytest.raises(ValueError):
        Slab(2, 0, 3)


# ---- voxel-masked domain with per-voxel tensors, slabs cut by tissue weight --------------------------------------
VCELLS, VH = (10, 8, 14), (0.1, 0.1, 0.1)


def _voxel_problem():
    from oracle import fem

    rng = np.random.default_rng(11)
    nbox = int(np.prod(VCELLS))
    cc = np.stack(np.meshgrid(*[np.arange(c) + 0.5 for c in reversed(VCELLS)], indexing="ij"), axis=-1)[..., ::-1].reshape(-1, 3)
    r = np.sqrt((((cc - np.array(VCELLS) / 2.0) / (np.array(VCELLS) / 2.0)) ** 2).sum(axis=1))
    active = (r < 0.95) & (r > 0.35) & (cc[:, 2] < 11)
    ang = 0.4 * cc[:, 2]
    f = np.stack([np.cos(ang), np.sin(ang), 0.0 * ang], axis=1)
    M = 1.25e-4 * np.eye(3)[None] + (9.5e-4 - 1.25e-4) * f[:, :, None] * f[:, None, :]
    mesh = fem.BoxMesh(VCELLS, tuple(c * h for c, h in zip(VCELLS, VH)))
    act_s = np.repeat(active, 6)
    tissue = fem.assemble_mass(mesh, np.nonzero(act_s)[0]).diagonal() > 0
    v_prev = np.where(tissue, -85.0 + 100.0 * np.exp(-((mesh.x - 0.4) ** 2).sum(axis=1) / 0.05) + 0.01 * rng.standard_normal(mesh.num_nodes), 0.0)
    w = fem.stimulus_weights(mesh, np.nonzero(act_s & (mesh.x[mesh.cells].mean(axis=1)[:, 2] < 0.75))[0])
    return mesh, active, M, v_prev, w


def _voxel_worker(rank, world, port, out_dir):
    for p in (str(ROOT), str(ROOT / "fenicsx-beat_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _oracle_ops import CpuField, OracleOps
        from beat import _stencil
        from beat import grid as g
        from beat._engine import DiffusionSolver

        mesh_o, active, M, v_prev, w = _voxel_problem()
        mesh = g.create_voxel_mesh(g.COMM_WORLD, active.reshape(tuple(reversed(VCELLS))), VH)  # slabs by tissue weight
        slab = mesh.slab
        assert (slab.rank, slab.world) == (rank, world)
        nx, ny, nz = mesh.shape_global
        plane = nx * ny
        mf, kf = _stencil.stencil_fields(3, VCELLS, VH, M, active, z_range=(slab.z0, slab.z1))
        ops = OracleOps(mesh.shape_local, slab.lo_phys, slab.hi_phys, mf, kf, per_node=True)
        ops.set_timestep(C_M, THETA, DT)
        solver = DiffusionSolver(ops, slab, group=mesh.comm.group)
        sl = slice(slab.z0 * plane, slab.z1 * plane)
        fv, fx, fw = (CpuField(ops.n, plane) for _ in range(3))
        fv.data.copy_(torch.from_numpy(v_prev[sl].copy()))
        fw.data.copy_(torch.from_numpy(w[sl].copy()))
        res = solver.solve(fv, [fw], [AMP], fx, rtol=1e-12, atol=1e-50, max_it=500)
        np.savez(Path(out_dir) / f"rank{rank}.npz", x=fx.numpy(), its=res.iterations, reason=res.converged_reason,
                 z0=slab.z0, z1=slab.z1, bnorm=res.rhs_norm, tissue=int(mesh.node_active().sum()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_voxel_mask_per_node_rows_slab_decomposed(world, tmp_path):
    """Per-node rows cut out per rank (stencil_fields z_range), slabs balanced by tissue weight, halo exchange and
    the deferred-x PCG on gloo: equals the sparse-LU solve of the undivided masked problem."""
    from oracle import fem

    port = _free_port()
    mp.spawn(_voxel_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mesh, active, M, v_prev, w = _voxel_problem()
    act_s = np.repeat(active, 6)
    model = fem.OracleMonodomainModel(mesh, np.repeat(M, 6, axis=0), [fem.OracleStimulus(lambda t: AMP, w)], C_m=C_M,
                                      theta=THETA, default_timestep=DT, active_cells=act_s)
    model.state[:] = v_prev
    model.assign_previous()
    model.step((0.0, DT))
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert [int(p["z0"]) for p in parts] == [0] + [int(p["z1"]) for p in parts[:-1]]
    assert int(parts[-1]["z1"]) == mesh.shape_nodes[2]
    x = np.concatenate([p["x"] for p in parts])
    assert np.abs(x - model.state).max() <= 1e-9 * np.abs(model.state).max()
    assert len({int(p["its"]) for p in parts}) == 1 and all(int(p["reason"]) > 0 for p in parts)
    # tissue-weighted cuts: no rank holds more than ~1.6x its share of the tissue nodes, and the planes differ
    tissue = [int(p["tissue"]) for p in parts]
    assert max(tissue) <= 1.6 * sum(tissue) / world
    assert len({int(p["z1"]) - int(p["z0"]) for p in parts}) > 1
    A = model.A.tocsr()
    _, its_ref, _ = fem.pcg_jacobi(A, model.rhs(THETA * DT, DT), v_prev, rtol=1e-12)
    assert abs(int(parts[0]["its"]) - its_ref) <= 1


def test_bench_refuses_more_ranks_than_visible_gpus():
    """``python bench.py --gpus 8`` on a host that shows fewer devices: one clear line, non-zero exit, within seconds,
    before any rank process is started (counting devices does not initialise a GPU)."""
    import subprocess
    import sys
    import time
    from pathlib import Path

    import torch

    if torch.cuda.device_count() >= 8:
        pytest.skip("this host has 8 GPUs")
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BEAT_DIST_BACKEND")}
    tic = time.perf_counter()
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=120,
                         cwd=root, env=env)
    assert res.returncode != 0 and time.perf_counter() - tic < 30
    assert "needs 8 visible GPUs" in res.stderr and not res.stdout.strip()


def _layout2d_worker(rank, world, port, out_dir):
    for p in (str(ROOT), str(ROOT / "fenicsx-beat_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from beat import _stencil
        from beat import grid as g

        mesh = g.create_unit_square(g.COMM_WORLD, 6, 9)
        out = {"z0": mesh.slab.z0, "z1": mesh.slab.z1, "plane": mesh.plane, "shape_local": np.array(mesh.shape_local),
               "xyz": mesh.node_coordinates(pad3=True), "nodes": mesh.num_nodes}
        for name, space in (("p2", ("P", 2)), ("dg1", ("DG", 1))):
            V = g.functionspace(mesh, space)
            (idx, w), to_p1 = V.layout()
            out[name + "_idx"], out[name + "_w"], out[name + "_to_p1"] = idx, w, to_p1
        np.savez(Path(out_dir) / f"rank{rank}.npz", **out)
    finally:
        dist.destroy_process_group()


def test_two_dimensional_mesh_is_cut_into_slabs_of_rows(tmp_path):
    """The reference's unit-square tests run under ``mpirun -n 2``: a 2-D mesh on several ranks is cut along y.  Host
    logic on gloo (world 3): every rank owns whole rows, ``plane`` is one row of nodes, the kernels' grid is (nx, 1,
    rows), local coordinates are those of the owned rows; the P2 / DG1 layouts of the slabs add up to the one-rank
    layouts; and the operator tables handed to the kernels (y in the place of z) describe the same matrix: applying
    them row by row on the (nx, 1, ny) grid reproduces the assembled 2-D operator of the oracle."""
    from beat import _stencil
    from beat import grid as g
    from oracle import fem

    port = _free_port()
    mp.spawn(_layout2d_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(3)]
    one = g.create_unit_square(g.Comm(), 6, 9)
    xyz = one.node_coordinates(pad3=True, local=False)
    assert [int(p["plane"]) for p in parts] == [7, 7, 7] and sum(int(p["nodes"]) for p in parts) == 70
    assert all(tuple(p["shape_local"]) == (7, 1, int(p["z1"]) - int(p["z0"])) for p in parts)
    np.testing.assert_allclose(np.concatenate([p["xyz"] for p in parts]), xyz)
    (idx1, w1), _ = g.functionspace(one, ("P", 2)).layout()
    got = []
    for p in parts:
        base = int(p["z0"]) * 7
        got += list(map(tuple, np.column_stack([p["p2_idx"] + base, p["p2_w"]]).tolist()))
        nloc = int(p["nodes"])
        idx = p["dg1_idx"][:, 0]
        np.testing.assert_array_equal(idx[p["dg1_to_p1"]], np.arange(nloc))
        assert idx.min() >= -7 and idx.max() < nloc + 7
    assert sorted(got) == sorted(map(tuple, np.column_stack([idx1, w1]).tolist()))
    # the remapped tables: y-as-z stencil applied with plain NumPy on the (nx, 1, ny) grid == the oracle's assembled matrix
    M = np.array([[1.0, 0.2], [0.2, 0.6]])
    mt, kt = _stencil.stencil_tables(2, one.h, M)
    m2, k2 = _stencil.tables_y_as_z(mt), _stencil.tables_y_as_z(kt)
    om = fem.BoxMesh((6, 9), (1.0, 1.0))
    K = fem.assemble_stiffness(om, M)
    Mass = fem.assemble_mass(om)
    nx, ny = 7, 10
    rng = np.random.default_rng(0)
    x = rng.standard_normal(nx * ny)
    for tab, mat in ((m2, Mass), (k2, K)):
        y = np.zeros_like(x)
        for iz in range(ny):  # the kernels' z = the mesh's y
            for ix in range(nx):
                tx = 0 if ix == 0 else 2 if ix == nx - 1 else 1
                tz = 0 if iz == 0 else 2 if iz == ny - 1 else 1
                row = tab[tx + 3 + 9 * tz]
                acc = 0.0
                for k, (dx, dy, dz) in enumerate(_stencil.OFFSETS):
                    if row[k] != 0.0:
                        acc += row[k] * x[(ix + dx) + nx * (iz + dz)]
                        assert dy == 0
                y[ix + nx * iz] = acc
        np.testing.assert_allclose(y, mat @ x, rtol=0, atol=1e-12 * np.abs(mat @ x).max())


def _percell2d_problem():
    from oracle import fem

    cells, L = (14, 11), (1.4, 1.1)
    mesh = fem.BoxMesh(cells, L)
    rng = np.random.default_rng(4)
    cc = np.stack(np.meshgrid((np.arange(cells[1]) + 0.5) / cells[1], (np.arange(cells[0]) + 0.5) / cells[0], indexing="ij"), -1)[..., ::-1].reshape(-1, 2)
    ang = 2.0 * cc[:, 0] + 0.7 * cc[:, 1] + 0.2 * rng.standard_normal(len(cc))
    f = np.stack([np.cos(ang), np.sin(ang)], axis=-1)
    M = 2e-4 * np.eye(2)[None] + 8e-4 * f[:, :, None] * f[:, None, :]
    active = ((cc - 0.5) ** 2).sum(axis=1) < 0.46**2  # a disc of active cells: rows of nodes without any tissue at the rim
    v_prev = -80.0 + 50.0 * np.exp(-((mesh.x - np.array([0.5, 0.6])) ** 2).sum(axis=1) / 0.03)
    return mesh, cells, tuple(l / c for l, c in zip(L, cells)), M, active, v_prev


def _percell2d_worker(rank, world, port, out_dir):
    for p in (str(ROOT), str(ROOT / "fenicsx-beat_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _oracle_ops import CpuField, OracleOps
        from beat import _stencil
        from beat import grid as g
        from beat._engine import DiffusionSolver

        mesh_o, cells, h, M, active, v_prev = _percell2d_problem()
        mesh = g.create_rectangle(g.COMM_WORLD, [np.zeros(2), np.array([1.4, 1.1])], list(cells))
        slab = mesh.slab
        assert mesh.kernel_y_as_z and (slab.rank, slab.world) == (rank, world)
        nxn = mesh.shape_local[0]
        # what beat._engine.build_ops does for such a mesh (there with the HIP operators): the 2-D rows, this rank's rows of
        # nodes, coefficients moved to the slots of the (nx, 1, rows) grid
        mass, stiff = _stencil.stencil_fields(2, cells, h, M, active)
        cut = slice(slab.z0 * nxn, slab.z1 * nxn)
        ops = OracleOps(mesh.shape_local, slab.lo_phys, slab.hi_phys, _stencil.fields_y_as_z(mass[:, cut]),
                        _stencil.fields_y_as_z(stiff[:, cut]), per_node=True)
        ops.set_timestep(C_M, THETA, DT)
        solver = DiffusionSolver(ops, slab, group=mesh.comm.group)
        fv, fx = CpuField(ops.n, nxn), CpuField(ops.n, nxn)
        tissue = mass[0, cut] != 0.0
        fv.data.copy_(torch.from_numpy(np.where(tissue, v_prev[cut], 0.0)))
        res = solver.solve(fv, [], [], fx, rtol=1e-12, atol=1e-50, max_it=500)
        np.savez(Path(out_dir) / f"rank{rank}.npz", x=fx.numpy(), its=res.iterations, reason=res.converged_reason, z0=slab.z0, z1=slab.z1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_per_cell_tensors_and_mask_on_a_two_dimensional_mesh_cut_into_rows(world, tmp_path):
    """The last NotImplementedError on a path the reference's parallel CI reaches (tests/test_monodomain_solver.py:33-216
    under `mpirun -n 2`, .github/workflows/main-mpi.yml:33): a 2-D mesh on several ranks with a conductivity tensor per cell
    and a cell mask.  Every rank cuts its rows of nodes out of the 2-D per-node operator and hands them to the kernels'
    (nx, 1, rows) grid with the coefficients moved to the y-as-z slots (_stencil.fields_y_as_z); the theta-step solved by the
    decomposed PCG on gloo equals the sparse-LU solve of the undivided masked problem assembled by the oracle."""
    from oracle import fem

    port = _free_port()
    mp.spawn(_percell2d_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mesh, cells, h, M, active, v_prev = _percell2d_problem()
    act_s = np.repeat(active, 2)
    model = fem.OracleMonodomainModel(mesh, np.repeat(M, 2, axis=0), [], C_m=C_M, theta=THETA, default_timestep=DT, active_cells=act_s)
    tissue = fem.assemble_mass(mesh, np.nonzero(act_s)[0]).diagonal() > 0
    model.state[:] = np.where(tissue, v_prev, 0.0)
    model.assign_previous()
    model.step((0.0, DT))
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert [int(p["z0"]) for p in parts] == [0] + [int(p["z1"]) for p in parts[:-1]] and int(parts[-1]["z1"]) == cells[1] + 1
    x = np.concatenate([p["x"] for p in parts])
    assert np.abs(x - model.state)[tissue].max() <= 1e-9 * np.abs(model.state).max()
    assert len({int(p["its"]) for p in parts}) == 1 and all(int(p["reason"]) > 0 for p in parts)


def _layout_worker(rank, world, port, out_dir):
    for p in (str(ROOT), str(ROOT / "fenicsx-beat_amd"), str(ROOT / "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from beat import grid as g

        mesh = g.create_box(g.COMM_WORLD, [np.zeros(3), np.array([1.0, 0.8, 1.2])], [5, 4, 7])
        assert mesh.comm.size == world and mesh.slab.rank == rank
        out = {"z0": mesh.slab.z0, "z1": mesh.slab.z1, "plane": mesh.plane}
        for name, space in (("p2", ("P", 2)), ("dg1", ("DG", 1))):
            V = g.functionspace(mesh, space)
            (idx, w), to_p1 = V.layout()
            out[name + "_idx"], out[name + "_w"], out[name + "_to_p1"] = idx, w, to_p1
            out[name + "_xyz"] = V.tabulate_dof_coordinates()
            assert V.num_dofs == len(idx) == V.dofmap.index_map.size_local
        # nodal vector fields (fibres): the whole field on every rank, reduced per simplex of the slab at assembly
        W = g.functionspace(mesh, ("P", 1, (3,)))
        f = g.Function(W)
        f.interpolate(lambda x: np.stack([np.cos(x[2]), np.sin(x[2]), 0.0 * x[0]]))
        out["fibre"] = np.asarray(f.x.array).copy()
        np.savez(Path(out_dir) / f"rank{rank}.npz", **out)
    finally:
        dist.destroy_process_group()


def test_p2_and_dg1_layouts_and_nodal_fibres_on_two_slabs(tmp_path):
    """The ODE spaces the reference's splitting tests parametrise over (tests/test_monodomain_solver.py:33-216, run under
    ``mpirun -n 2`` by .github/workflows/main-mpi.yml:33) and the nodal fibre field of its geometries, on a mesh cut into
    two z-slabs (gloo, world 2; layouts are host logic, no GPU): every P2 dof -- a vertex of the slab or an edge starting
    at one -- is the same two-term combination of the same GLOBAL vertices as on one rank, every edge and vertex is held
    exactly once; every DG1 dof sits on a vertex of one of its cells, the layer of cells between the slabs is held by both
    ranks, the others by one; an edge or cell that reaches across the cut refers to the neighbour's plane through the ghost
    plane (local index >= n_local or < 0); ``to_p1`` picks, for every vertex of the slab, a dof located there."""
    from beat import grid as g

    port = _free_port()
    mp.spawn(_layout_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    one = g.create_box(g.Comm(), [np.zeros(3), np.array([1.0, 0.8, 1.2])], [5, 4, 7])
    assert one.comm.size == 1
    xyz = one.node_coordinates(pad3=True, local=False)
    plane = int(parts[0]["plane"])
    n_glob = one.num_nodes_global
    assert int(parts[0]["z0"]) == 0 and int(parts[0]["z1"]) == int(parts[1]["z0"]) and int(parts[1]["z1"]) == 8
    # P2: union over the ranks == the one-rank layout, as sets of (global a, global b, weights)
    (idx1, w1), _ = g.functionspace(one, ("P", 2)).layout()
    ref = sorted(map(tuple, np.column_stack([idx1, w1]).tolist()))
    got = []
    for p in parts:
        base, nloc = int(p["z0"]) * plane, (int(p["z1"]) - int(p["z0"])) * plane
        idx = p["p2_idx"]
        assert idx.min() >= 0 and idx.max() < nloc + plane  # an upward edge of the top plane ends in the upper ghost plane
        got += list(map(tuple, np.column_stack([idx + base, p["p2_w"]]).tolist()))
        np.testing.assert_array_equal(p["p2_to_p1"], np.arange(nloc))
        np.testing.assert_allclose(p["p2_xyz"], p["p2_w"][:, :1] * xyz[idx[:, 0] + base] + p["p2_w"][:, 1:] * xyz[idx[:, 1] + base])
    assert sorted(got) == ref
    assert any((p["p2_idx"].max() >= (int(p["z1"]) - int(p["z0"])) * plane) for p in parts[:1])  # rank 0 does reach across
    # DG1: cells as vertex tuples
    (idxd, _), _ = g.functionspace(one, ("DG", 1)).layout()
    cells_ref = sorted(map(tuple, idxd[:, 0].reshape(-1, 4).tolist()))
    seen = {}
    for r, p in enumerate(parts):
        base, nloc = int(p["z0"]) * plane, (int(p["z1"]) - int(p["z0"])) * plane
        idx = p["dg1_idx"][:, 0]
        assert idx.min() >= -plane and idx.max() < nloc + plane
        for c in map(tuple, (idx + base).reshape(-1, 4).tolist()):
            seen.setdefault(c, []).append(r)
        to_p1 = p["dg1_to_p1"]
        np.testing.assert_array_equal(idx[to_p1], np.arange(nloc))  # the chosen dof sits on that vertex
        np.testing.assert_allclose(p["dg1_xyz"], xyz[idx + base])
    assert sorted(seen) == cells_ref
    cut = int(parts[0]["z1"])  # cells between node planes cut-1 and cut are held by both ranks
    for c, ranks in seen.items():
        zs = {v // plane for v in c}
        assert ranks == ([0, 1] if zs == {cut - 1, cut} else [0] if max(zs) < cut else [1]), (c, ranks)
    # the nodal fibre field is the global one on both ranks
    for p in parts:
        assert p["fibre"].shape == (3 * n_glob,)
        np.testing.assert_allclose(p["fibre"].reshape(-1, 3), np.stack([np.cos(xyz[:, 2]), np.sin(xyz[:, 2]), 0 * xyz[:, 0]], axis=1))
