"""GPU: the slab-decomposed diffusion solve -- the in-library loop (beat_pde_solve_dist: RCCL ghost-plane exchange on
the library's side stream, RCCL all-reduces, split SpMV; one C call per solve) and the stage-driven Python loop it
replaced on the product path -- as far as a single-GPU box allows: one-rank RCCL communicators (also with the rank as
its own z-neighbour, so that the library's ncclSend/ncclRecv pairs really run), several ranks played by threads or by
processes sharing the GPU over host-staged callbacks, against the fused single-slab solve."""

import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("per_node", [False, True])
def test_collective_path_on_one_rank_matches_single_slab_solve(hip_ctx, per_node):
    import torch.distributed as dist

    from beat import _stencil
    from beat._engine import DiffusionSolver, HipOps, Slab

    ctx = hip_ctx
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=ctx.device)
        created = True
    try:
        nx, ny, nz = 40, 33, 17
        f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
        M = 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
        if per_node:  # masked grid + per-node rows (beat_pde_create_var)
            cc = np.stack(np.meshgrid(np.arange(nz - 1), np.arange(ny - 1), np.arange(nx - 1), indexing="ij"), -1).reshape(-1, 3)
            active = ((cc - np.array([8, 16, 20])) ** 2).sum(axis=1) < 14**2
            mt, kt = _stencil.stencil_fields(3, (nx - 1, ny - 1, nz - 1), (0.1, 0.1, 0.1), M, active)
        else:
            mt, kt = _stencil.stencil_tables(3, (0.1, 0.1, 0.1), M)
        rng = np.random.default_rng(3)
        v = -85.0 + 30.0 * rng.random(nx * ny * nz)
        results = []
        for mode in ("fused", "stage", "lib"):  # "lib" last: the deferred-update checks below run through it
            ops = HipOps(ctx, (nx, ny, nz), True, True, mt, kt, per_node=per_node)
            ops.set_timestep(0.01, 0.5, 0.05)
            solver = DiffusionSolver(ops, Slab(nz), force_distributed=mode != "fused", stage_driven=mode == "stage")
            assert (solver.libcomm is not None) == (mode == "lib")
            fv, fx = ops.new_field(), ops.new_field()
            fv.set(v)
            res = solver.solve(fv, [], [], fx, rtol=1e-11, atol=1e-50, max_it=200)
            results.append((fx.numpy(), res))
        (x1, r1), (xs, rs), (x2, r2) = results
        assert r2.converged_reason > 0 and abs(r1.iterations - r2.iterations) <= 1
        np.testing.assert_allclose(x2, x1, rtol=0, atol=1e-9 * np.abs(x1).max())
        # one rank: per-node rows -- library loop, Python stage loop and fused solve run the same kernels in the same
        # order; constant coefficients -- the fused solve and the library loop run the register-row kernels (same
        # bits), the stage loop the classic three-kernel iteration (same values up to the summation order)
        if per_node:
            np.testing.assert_array_equal(xs, x2)
            assert rs.iterations == r2.iterations
        else:
            np.testing.assert_array_equal(x1, x2)
            np.testing.assert_allclose(xs, x2, rtol=0, atol=1e-11 * np.abs(x2).max())
            assert abs(rs.iterations - r2.iterations) <= 1 and r1.iterations == r2.iterations
        # deferred last update through the collective path: pending until flushed, then the same bits
        fx3 = ops.new_field()
        res3 = solver.solve(fv, [], [], fx3, rtol=1e-11, atol=1e-50, max_it=200, defer_flush=True)
        assert res3.iterations == r2.iterations
        if res3.iterations % 6:
            assert ops.pending is not None and not np.array_equal(fx3.numpy(), x2)
        ops.flush_pending()
        assert ops.pending is None
        np.testing.assert_array_equal(fx3.numpy(), x2)
    finally:
        if created:
            dist.destroy_process_group()


class _ThreadWorld:
    """A torch.distributed look-alike for several threads of one process, each playing one rank: the pieces
    DiffusionSolver uses (all_reduce on device tensors, P2POp / batch_isend_irecv of ghost planes).  A one-GPU box
    cannot host a multi-rank RCCL group (one rank per device), so this is how the N > 1 orchestration is run against
    the real HIP kernels -- slabs with live ghost planes, split SpMV, boundary-plane launches -- before the
    driver's multi-GPU run; RCCL itself is covered by the one-rank group above and by torch."""

    def __init__(self, world):
        import queue
        import threading

        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.mail = {(a, b): queue.Queue() for a in range(world) for b in range(world)}

    def rank_view(self, rank):
        return _ThreadRank(self, rank)


class _ThreadRank:
    class ReduceOp:
        SUM = "sum"

    isend, irecv = "isend", "irecv"

    def __init__(self, world, rank):
        self.w, self.rank = world, rank

    def all_reduce(self, t, op=None, group=None):
        w = self.w
        self.reduces = getattr(self, "reduces", 0) + 1
        w.slots[self.rank] = t.clone()
        w.barrier.wait(timeout=120)
        total = w.slots[0].clone()
        for other in w.slots[1:]:
            total += other
        w.barrier.wait(timeout=120)  # everyone has read the slots before they are reused
        t.copy_(total)

    def P2POp(self, fn, tensor, peer, group=None):
        return (fn, tensor, peer)

    def get_global_rank(self, group, rank):
        return rank

    def batch_isend_irecv(self, ops):
        reqs = []
        for fn, tensor, peer in ops:
            if fn == self.isend:
                self.w.mail[(self.rank, peer)].put(tensor.clone())
            else:
                reqs.append(_ThreadRecv(self.w.mail[(peer, self.rank)], tensor))
        return reqs


class _ThreadRecv:
    def __init__(self, box, tensor):
        self.box, self.tensor = box, tensor

    def wait(self):
        self.tensor.copy_(self.box.get(timeout=120))


def _thread_solver(ops, slab, dist_view, loop):
    """DiffusionSolver of one thread-rank: ``loop="lib"`` runs beat_pde_solve_dist over a callback communicator whose
    transport is the thread world; ``loop="stage"`` the Python stage loop over the same world."""
    from beat._engine import DiffusionSolver, LibComm

    if dist_view is None:
        return DiffusionSolver(ops, slab)
    if loop == "lib":
        return DiffusionSolver(ops, slab, force_distributed=True, libcomm=LibComm(ops.ctx, slab, dist_view, None, "callbacks"))
    solver = DiffusionSolver(ops, slab, force_distributed=True, stage_driven=True)
    solver.dist = dist_view
    return solver


@pytest.mark.parametrize("loop", ["lib", "stage"])
@pytest.mark.parametrize("per_node,world,nz", [(False, 2, 19), (False, 3, 19), (True, 3, 19), (False, 4, 5), (True, 4, 6),
                                               (False, 5, 5), (False, 8, 19), (True, 8, 21)])
def test_slabs_in_threads_match_single_slab_solve(hip_ctx, per_node, world, nz, loop):
    """world ranks as threads, each with its own context, slab operators and DiffusionSolver: the assembled solution
    equals the undivided solve, iteration counts agree, and the deferred last update flushes to the same values --
    through the in-library loop (callback transport) and through the stage-driven Python loop."""
    import threading

    from beat import _stencil
    from beat._device import Context
    from beat._engine import DiffusionSolver, HipOps, Slab

    nx, ny = 40, 33  # nz = 5 on 4 or 5 ranks: slabs of one or two planes, both ghost planes live
    cells = (nx - 1, ny - 1, nz - 1)
    h = (0.1, 0.1, 0.1)
    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M = 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
    active = None
    if per_node:
        cc = np.stack(np.meshgrid(np.arange(nz - 1), np.arange(ny - 1), np.arange(nx - 1), indexing="ij"), -1).reshape(-1, 3)
        active = ((cc - np.array([min(9, nz // 2), 16, 20])) ** 2).sum(axis=1) < 15**2

    def operators(z_range):
        if per_node:
            return _stencil.stencil_fields(3, cells, h, M, active, z_range=z_range)
        return _stencil.stencil_tables(3, h, M)

    plane = nx * ny
    rng = np.random.default_rng(5)
    v = -85.0 + 30.0 * rng.random(nx * ny * nz)
    w_stim = np.zeros_like(v)
    w_stim[: 2 * plane] = rng.random(2 * plane) * 1e-3
    if per_node:  # only tissue nodes carry a stimulus weight
        tissue = operators(None)[0][0] != 0.0
        w_stim *= tissue

    def solve(ctx, slab, dist_view, out, key):
        ops = HipOps(ctx, (nx, ny, slab.nz), slab.lo_phys, slab.hi_phys, *operators((slab.z0, slab.z1)), per_node=per_node)
        ops.set_timestep(0.01, 0.5, 0.05)
        solver = _thread_solver(ops, slab, dist_view, loop)
        fv, fx, fw, fx2 = ops.new_field(), ops.new_field(), ops.new_field(), ops.new_field()
        sl = slice(slab.z0 * plane, slab.z1 * plane)
        fv.set(v[sl])
        fw.set(w_stim[sl])
        res = solver.solve(fv, [fw], [0.7], fx, rtol=1e-11, atol=1e-50, max_it=200)
        res2 = solver.solve(fv, [fw], [0.7], fx2, rtol=1e-11, atol=1e-50, max_it=200, defer_flush=True)
        ops.flush_pending()
        ctx.synchronize()
        out[key] = (fx.numpy().copy(), res, fx2.numpy().copy(), res2)
        fused[key] = bool(ctx.lib.beat_pde_fused_dist_pass(ops.handle))

    out, fused = {}, {}
    solve(hip_ctx, Slab(nz), None, out, "whole")
    tw = _ThreadWorld(world)
    errors = []

    def run(rank):
        try:
            solve(Context(), Slab(nz, rank, world), tw.rank_view(rank), out, rank)
        except Exception as exc:  # noqa: BLE001
            errors.append((rank, exc))
            tw.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    x_whole, r_whole, _, _ = out["whole"]
    x_parts = np.concatenate([out[r][0] for r in range(world)])
    x_defer = np.concatenate([out[r][2] for r in range(world)])
    its = {out[r][1].iterations for r in range(world)}
    assert len(its) == 1 and abs(its.pop() - r_whole.iterations) <= 1
    assert all(out[r][1].converged_reason > 0 for r in range(world))
    np.testing.assert_allclose(x_parts, x_whole, rtol=0, atol=1e-9 * np.abs(x_whole).max())
    np.testing.assert_array_equal(x_defer, x_parts)
    # per-node rows through the library loop: the fused tile pass on every rank or on none (the ranks agree: a slab without a tissue
    # node has no tile list) -- on all of them where every slab holds tissue (19+ planes over 2-8 ranks around a sphere of radius 15)
    agreed = {fused[r] for r in range(world)}
    assert len(agreed) == 1 and not fused["whole"]
    if per_node and loop == "lib" and nz >= 19:
        assert agreed == {True}
    if not per_node or loop == "stage":
        assert agreed == {False}


@pytest.mark.parametrize("rtol", [1e-8, 1e-12])
@pytest.mark.parametrize("world,nz", [(2, 19), (3, 19), (4, 5), (8, 19)])
def test_single_reduction_iteration_on_slabs_in_threads(hip_ctx, monkeypatch, world, nz, rtol):
    """BEAT_DIST_MERGED=1: the decomposed solve with ONE all-reduce per iteration (Chronopoulos-Gear recurrences on the
    register-row kernels, csrc/beat_pde_rr.hip) on 2-8 thread-ranks against the classic two-reduction solve of the undivided
    grid: same solution, iteration counts within one, the TRUE residual b - A x of the assembled solution meets the stopping
    test, the deferred update flushes to the same bits, and the ranks call the all-reduce once per enqueued pass instead of
    twice."""
    import threading

    from beat import _stencil
    from beat._device import Context
    from beat._engine import HipOps, Slab

    nx, ny = 40, 33
    h = (0.1, 0.1, 0.1)
    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M = 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
    mt, kt = _stencil.stencil_tables(3, h, M)
    plane = nx * ny
    rng = np.random.default_rng(17)
    v = -85.0 + 30.0 * rng.random(plane * nz)

    def solve(ctx, slab, dist_view, out, key):
        ops = HipOps(ctx, (nx, ny, slab.nz), slab.lo_phys, slab.hi_phys, mt, kt)
        ops.set_timestep(0.01, 0.5, 0.05)
        solver = _thread_solver(ops, slab, dist_view, "lib")
        fv, fx, fx2 = ops.new_field(), ops.new_field(), ops.new_field()
        fv.set(v[slab.z0 * plane : slab.z1 * plane])
        res = solver.solve(fv, [], [], fx, rtol=rtol, atol=1e-50, max_it=200)
        ctx.synchronize()
        before = getattr(dist_view, "reduces", 0)
        res2 = solver.solve(fv, [], [], fx2, rtol=rtol, atol=1e-50, max_it=200, defer_flush=True)
        ops.flush_pending()
        ctx.synchronize()
        calls = getattr(dist_view, "reduces", 0) - before  # of the second solve, whose first batch is sized from the first
        merged = int(ctx.lib.beat_comm_merged_solves(solver.libcomm.handle)) if dist_view is not None else 0
        out[key] = (fx.numpy().copy(), res, fx2.numpy().copy(), res2, calls, merged)
        return ops

    def run_world(out):
        tw = _ThreadWorld(world)
        errors = []

        def run(rank):
            try:
                solve(Context(), Slab(nz, rank, world), tw.rank_view(rank), out, rank)
            except Exception as exc:  # noqa: BLE001
                errors.append((rank, exc))
                tw.barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors

    out, two = {}, {}
    ops = solve(hip_ctx, Slab(nz), None, out, "whole")
    monkeypatch.delenv("BEAT_DIST_MERGED", raising=False)
    run_world(two)
    monkeypatch.setenv("BEAT_DIST_MERGED", "1")
    run_world(out)
    x_whole, r_whole = out["whole"][:2]
    x_parts = np.concatenate([out[r][0] for r in range(world)])
    its = {out[r][1].iterations for r in range(world)}
    assert len(its) == 1
    k = its.pop()
    assert abs(k - r_whole.iterations) <= 1 and abs(k - two[0][1].iterations) <= 1 and k >= 3
    assert all(out[r][1].converged_reason > 0 and out[r][5] == 2 for r in range(world))
    assert all(two[r][5] == 0 for r in range(world))
    np.testing.assert_allclose(x_parts, x_whole, rtol=0, atol=max(10 * rtol, 1e-10) * np.abs(x_whole).max())
    np.testing.assert_array_equal(np.concatenate([out[r][2] for r in range(world)]), x_parts)
    # the true residual of the assembled solution, with the undivided operator
    fv, fx, b, ax = (ops.new_field() for _ in range(4))
    fv.set(v)
    ops.apply(1, fv, b)
    bnorm = float(np.linalg.norm(b.numpy()))

    def true_residual(x):
        fx.set(x)
        ops.apply(0, fx, ax)
        return float(np.linalg.norm(b.numpy() - ax.numpy()))

    assert np.isclose(bnorm, out[0][1].rhs_norm, rtol=1e-12)
    floor = 2e-15 * bnorm  # what evaluating b - A x in double precision leaves
    assert true_residual(x_parts) <= 1.1 * rtol * bnorm + floor
    assert true_residual(x_parts) <= 3.0 * true_residual(x_whole) + floor
    # all-reduce calls of a solve that enqueues one iteration more than the previous one needed (beat_pde_first_chunk):
    # the right-hand side's + one per enqueued pass (k + 1 and the pass that finds r_k converged) against two per iteration
    assert {out[r][4] for r in range(world)} == {1 + (k + 2)}
    k2 = two[0][1].iterations
    assert {two[r][4] for r in range(world)} == {1 + 2 * (k2 + 1)}


@pytest.mark.parametrize("loop,order,per_node", [("lib", 0, False), ("lib", 2, False), ("lib", 1, False), ("stage", 0, False),
                                                 ("lib", 2, True), ("lib", 0, True), ("lib", 3, False), ("lib", 4, True), ("lib", "auto", False),
                                                 ("lib/merged", 0, False), ("lib/merged", 2, False), ("lib/merged", "auto", False)])
def test_split_steps_on_slabs_in_threads_match_one_rank(hip_ctx, loop, order, per_node, monkeypatch):
    """bench.py's N > 1 step (TP06 ionic kernel applying the previous solve's pending directions on the slab's V row,
    then the slab-decomposed diffusion solve with the deferred last update) on 3 ranks played by threads: after 25 steps
    the assembled state array equals the one-rank run to 1e-9 (the reductions are summed in a different order).
    order > 0: the solves start from the extrapolated guess (beat_pde_set_guess_order) -- the ionic kernel then also
    records the step's diffusion increment, whose ghost planes the next decomposed solve exchanges.  per_node: the
    same on a voxel-masked domain with per-node operator rows (the guess increment is materialised and exchanged)."""
    import ctypes as C
    import threading

    from beat import _hip, _stencil
    from beat._device import Context, StateArray
    from beat._engine import DiffusionSolver, HipOps, Slab
    from beat.models import tp06

    if loop.endswith("/merged"):  # the single-reduction iteration (BEAT_DIST_MERGED) under the fused split step
        monkeypatch.setenv("BEAT_DIST_MERGED", "1")
        loop = "lib"
    else:
        monkeypatch.delenv("BEAT_DIST_MERGED", raising=False)
    nx, ny, nz, world, nsteps = 24, 17, 11, 3, 25
    plane = nx * ny
    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M = 9.5301e-4 * np.outer(f0, f0) + 1.2576e-4 * (np.eye(3) - np.outer(f0, f0))
    cells = (nx - 1, ny - 1, nz - 1)
    cc = np.stack(np.meshgrid(np.arange(nz - 1), np.arange(ny - 1), np.arange(nx - 1), indexing="ij"), -1).reshape(-1, 3)
    active = ((cc - np.array([5, 8, 9])) ** 2).sum(axis=1) < 9**2

    def operators(z_range):
        if per_node:
            return _stencil.stencil_fields(3, cells, (0.1, 0.1, 0.1), M, active, z_range=z_range)
        return _stencil.stencil_tables(3, (0.1, 0.1, 0.1), M)

    ic = tp06.init_state_values()
    p_host = np.ascontiguousarray(tp06.init_parameter_values(stim_amplitude=0.0))
    vi = tp06.state_index("V")
    rng = np.random.default_rng(2)
    S0 = np.repeat(ic[:, None], plane * nz, axis=1) * (1.0 + 0.01 * rng.uniform(-1, 1, (len(ic), plane * nz)))
    zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    S0[vi] = ic[vi] + 70.0 * np.exp(-((xx - 6) ** 2 + (yy - 8) ** 2 + (zz - 5) ** 2).ravel() / 18.0)

    def run(ctx, slab, dist_view, out, key):
        n_local = plane * slab.nz
        ops = HipOps(ctx, (nx, ny, slab.nz), slab.lo_phys, slab.hi_phys, *operators((slab.z0, slab.z1)), per_node=per_node)
        ops.set_guess_order(order)
        ops.set_timestep(0.01, 0.5, 0.05)
        solver = _thread_solver(ops, slab, dist_view, loop)
        states = StateArray(ctx, len(ic), n_local, plane)
        states.set(S0[:, slab.z0 * plane : slab.z1 * plane])
        v_field = states.row_field(vi)
        its = []
        for step in range(nsteps):
            pend = ops.pending
            ops.pending = None
            _hip.check(ctx.lib.beat_ode_step_pending(ctx.handle, _hip.MODEL_TP06_GRL1, states.ptr, n_local, states.ld,
                                                     p_host.ctypes.data_as(C.c_void_p), len(p_host), None, 0, 0.05 * step, 0.05,
                                                     vi, None, ops.handle, ops.ring[0].ptr, ops.fld, pend[2] if pend else 0))
            its.append(solver.solve(v_field, [], [], v_field, rtol=1e-10, atol=1e-50, max_it=200, defer_flush=True).iterations)
        ops.flush_pending()
        ctx.synchronize()
        out[key] = (states.numpy().copy(), its)

    out, errors = {}, []
    run(hip_ctx, Slab(nz), None, out, "whole")
    tw = _ThreadWorld(world)

    def worker(rank):
        try:
            run(Context(), Slab(nz, rank, world), tw.rank_view(rank), out, rank)
        except Exception as exc:  # noqa: BLE001
            errors.append((rank, exc))
            tw.barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    whole, its_whole = out["whole"]
    parts = np.concatenate([out[r][0] for r in range(world)], axis=1)
    assert all(out[r][1] == out[0][1] for r in range(world))
    assert max(abs(a - b) for a, b in zip(out[0][1], its_whole)) <= 1
    assert whole[vi].max() > 0.0  # the stimulated region fired
    scale = np.maximum(np.abs(whole), 1e-6 * np.abs(ic)[:, None] + 1e-12)
    assert (np.abs(parts - whole) / scale).max() < 1e-9


def test_bench_multi_process_rehearsal_on_one_gpu():
    """bench.py launched the way the driver launches it for N = 2 (torch.distributed.run, one process per rank), with the
    two ranks sharing this box's GPU over the host-staged gloo transport (BEAT_DIST_BACKEND=gloo; RCCL needs one
    device per rank): stdout is exactly one JSON line, it reports both ranks' work, the state is finite, and the PCG
    iteration count equals the one-process run's."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    common = ["--steps", "4", "--warmup", "1", "--size", "48", "--cpu-sample", "0"]
    one = subprocess.run([sys.executable, str(root / "bench.py"), *common], capture_output=True, text=True, timeout=300, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    env = dict(os.environ, BEAT_DIST_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", *common],
                         capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, two.stdout[-2000:]
    r1, r2 = json.loads(one.stdout.strip().splitlines()[-1]), json.loads(lines[0])
    assert r2["n_gpus"] == 2 and r2["config"]["nodes"] == 48**3 and r2["config"]["finite"] and r2["cpu_baseline"] is None
    assert abs(r2["config"]["pcg_iterations_per_step"] - r1["config"]["pcg_iterations_per_step"]) <= 0.5
    assert abs(r2["config"]["v_max"] - r1["config"]["v_max"]) < 1.0  # same bump; the gates' 1 % noise is seeded per rank
    # one rank fails to create the library's communicator: both ranks fall back, together, to the stage-driven loop
    # over torch.distributed (no initial guess there: more iterations, same answer) instead of hanging or crashing
    env_fail = dict(env, BEAT_TEST_FAIL_LIBCOMM_RANK="1")
    fb = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", *common],
                        capture_output=True, text=True, timeout=300, cwd=root, env=env_fail)
    assert fb.returncode == 0, fb.stderr[-2000:]
    assert "stage-driven loop" in fb.stderr
    r3 = json.loads([ln for ln in fb.stdout.splitlines() if ln.strip()][0])
    assert r3["config"]["finite"] and abs(r3["config"]["v_max"] - r2["config"]["v_max"]) < 1e-4
    assert r3["config"]["pcg_iterations_per_step"] >= r2["config"]["pcg_iterations_per_step"]
    # the same on the RCCL transport's set-up path (forced here on gloo; its collective create is never reached): whether
    # rank 0 fails before it has a unique id to broadcast or rank 1 fails after receiving it, every rank learns of it in
    # the agreement that follows the broadcast and none enters ncclCommInitRank alone
    for failing in ("0", "1"):
        fr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                             "127.0.0.1", "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", *common],
                            capture_output=True, text=True, timeout=300, cwd=root,
                            env=dict(env, BEAT_TEST_FAIL_LIBCOMM_RANK=failing, BEAT_DIST_TRANSPORT="rccl", BEAT_BENCH_ALT="0"))
        assert fr.returncode == 0, fr.stderr[-2000:]
        assert fr.stderr.count("stage-driven loop") >= 2, fr.stderr[-2000:]  # both ranks fell back
        r4 = json.loads([ln for ln in fr.stdout.splitlines() if ln.strip()][0])
        assert r4["config"]["finite"] and abs(r4["config"]["v_max"] - r3["config"]["v_max"]) < 1e-9


def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` with no launcher around it (how the driver calls it): the parent starts two fresh
    rank processes, relays rank 0's ONE JSON line and returns 0; the line reports both ranks (per-rank stage times, the
    transport in use) and the same steps over the ipc transport -- two processes exchanging ghost planes on the device
    through each other's mailboxes (hipIpc*).  A job that stops printing is killed by the parent's watchdog and tried once
    more in fresh children with BEAT_DIST_SERIAL=1; when that fails too the return code is non-zero."""
    import json
    import os
    import subprocess
    import sys
    import time
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["BEAT_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--size", "64", "--steps", "4", "--warmup", "1"]
    two = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, two.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["nodes"] == 64**3 and r["config"]["finite"]
    assert r["config"]["launch"]["attempts"][-1]["rc"] == 0 and len(r["config"]["launch"]["attempts"]) == 1
    assert [x["rank"] for x in r["ranks"]] == [0, 1] and sum(x["nodes"] for x in r["ranks"]) == 64**3
    assert all(x["ode_ms"] > 0 and x["pde_ms"] > 0 for x in r["ranks"])
    choice = r["config"]["transport_choice"]
    assert choice["headline_measured_on"] == "callbacks" and choice["adopted"] in (None, "ipc")
    # (an alternative that is > 3 % faster in the regime both were timed in gets the headline regime re-measured on it, and
    # that measurement becomes `value`: host-staged callbacks against device-to-device copies -- ipc usually wins here)
    assert r["config"]["comm"]["transport"] == ("ipc" if choice["adopted"] else "callbacks")
    # the ordering `value` was measured on is spelt out at the top of config, whichever branch produced the line
    assert r["config"]["ordering"] == ("overlapped" if choice["adopted"] else "serial")
    if choice["adopted"]:
        assert r["transports"]["ipc (headline)"]["ms_per_step"] == r["ms_per_step"] < r["transports"]["callbacks"]["ms_per_step"]
    ipc = r["transports"]["ipc"]
    assert "error" not in ipc and ipc["comm"]["transport"] == "ipc" and ipc["ms_per_step"] > 0
    assert abs(ipc["pcg_iterations_per_step"] - r["config"]["pcg_iterations_per_step"]) <= 1.5
    # a job that never prints: the watchdog kills it, the serial retry is made, the parent reports failure
    tic = time.perf_counter()
    hung = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root,
                          env=dict(env, BEAT_BENCH_TEST_HANG="1", BEAT_BENCH_WATCHDOG_START_S="20", BEAT_BENCH_WATCHDOG_S="20"))
    assert hung.returncode != 0 and time.perf_counter() - tic < 200
    assert hung.stderr.count("killing the ranks") == 2 and "BEAT_DIST_SERIAL" in hung.stderr and not hung.stdout.strip()


def test_bench_with_four_ranks_sharing_one_gpu(tmp_path):
    """``BEAT_DIST_BACKEND=gloo python bench.py --gpus 4 --size 80``: the launcher, four rank processes on this box's one GPU
    (the pool's process guard admits six processes with the GPU open: the test runner and a launcher are two), slabs of 20
    planes, the headline over the host-staged transport and the same steps over the mailboxes with four real processes: ONE
    line, n_gpus 4, four `ranks` entries that add up to the grid, the ordering spelt out, finite potentials."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["BEAT_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "4", "--size", "80", "--steps", "4", "--warmup", "1"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, run.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 4 and r["config"]["nodes"] == 80**3 and r["config"]["finite"]
    assert [x["rank"] for x in r["ranks"]] == [0, 1, 2, 3] and sum(x["nodes"] for x in r["ranks"]) == 80**3
    assert r["config"]["ordering"] in ("serial", "overlapped") and r["config"]["comm"]["world"] == 4
    ipc = r["transports"]["ipc"]
    assert "error" not in ipc and ipc["comm"]["transport"] == "ipc" and ipc["comm"]["allreduce"] == "ipc" and ipc["ms_per_step"] > 0
    clocks = r["config"]["clocks"]  # hwmon samples over the timed steps (None where sysfs is not readable)
    assert clocks is None or (clocks["samples"] >= 1 and clocks["sclk_mhz"]["mean"] > 100 and clocks["power_w"]["mean"] > 10)
    # the same steps with one all-reduce per PCG iteration (never adopted: reported beside the transports)
    one = r["single_reduction"]
    assert one["solves_on_the_single_reduction_iteration"] >= 4 and one["ms_per_step"] > 0
    assert one["pcg_iterations_per_step"] > 0 and one["transport"] in r["transports"]  # (the one the first headline was measured on)
    # correctness before speed (round 5): a TP06 slab and a voxel shell with per-node rows, decomposed on the four ranks against
    # undivided on rank 0 through the public API, on the headline's transport and on every alternative that was timed
    par = r["multi_rank_parity"]
    assert par["ok"] and par["tolerance"] == 1e-9 and {"callbacks", "ipc"} <= set(par)
    for transport in ("callbacks", "ipc"):
        for case in ("slab", "shell"):
            v = par[transport][case]
            assert v["ok"] and v["finite"] and v["max_rel_diff"] <= 1e-9, (transport, case, v)
            assert v["iterations_equal_across_ranks"] and abs(v["k"] - v["k_undivided"]) <= 1.0 and v["steps"] == 25, (transport, case, v)
    assert par["callbacks"]["slab"]["nodes"] == 96 * 96 * 48 and par["callbacks"]["shell"]["nodes"] == 65 * 65 * 65
    assert par["callbacks"]["seconds"] < 60.0  # (host-staged rehearsal transport; on RCCL / ipc the block takes a few seconds)


def test_bench_exits_nonzero_when_the_decomposed_run_differs_from_the_undivided_one():
    """``multi_rank_parity`` is a gate, not a decoration: with one rank's slab perturbed (BEAT_BENCH_TEST_PARITY_BREAK) the line is
    still printed -- with ``ok: false`` and the difference -- and the job's exit code is non-zero, through the launcher too."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(BEAT_DIST_BACKEND="gloo", BEAT_BENCH_TEST_PARITY_BREAK="1", BEAT_BENCH_ALT="0")
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--size", "48", "--steps", "2", "--warmup", "1", "--no-front"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert run.returncode != 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, (run.stdout[-2000:], run.stderr[-3000:])
    r = json.loads(lines[0])
    par = r["multi_rank_parity"]
    assert par["ok"] is False and r["value"] > 0
    bad = [v for t in par.values() if isinstance(t, dict) for v in t.values() if isinstance(v, dict)]
    assert bad and all(not v["ok"] and v["max_rel_diff"] > 1e-4 for v in bad)
    assert "multi-rank parity FAILED" in run.stderr and len(r["config"]["launch"]["attempts"]) == 1


@pytest.mark.parametrize("failing_rank", [0, 1])
def test_bench_alternative_transport_failing_on_one_rank_keeps_the_headline(failing_rank):
    """The alternative transports are timed after the headline (bench.py, `transports`).  One of them raising on ONE rank
    -- a transfer that timed out, a failed HIP call: the ranks are out of step from there on -- must not cost the measurement:
    that rank leaves with exit code 0 (rank 0 printing the headline line first), the others leave at the deadline, and the
    job's line carries the headline and says what happened."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(BEAT_DIST_BACKEND="gloo", BEAT_BENCH_TEST_ALT_RAISE=str(failing_rank), BEAT_BENCH_ALT_DEADLINE_S="12")
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--size", "48", "--steps", "3", "--warmup", "1"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=root, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, run.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["config"]["finite"]
    # (rank 0 reports its own failure, or -- when the other rank left -- the broken connection it then meets, or the deadline)
    assert "error" in r["transports"] and ("raised on rank" in r["transports"]["error"] or "no progress" in r["transports"]["error"])
    assert r["alt_failed"] is True and r["config"]["ordering"] == "serial"  # (the headline's transport: host-staged callbacks)
    assert "abandoned" in run.stderr


def _ipc_ranks(tmp_path, *args, timeout=240):
    """tests/_ipc_ranks_script.py in a fresh interpreter with enough hardware queues for 2 x world streams whose kernels wait
    for each other (the runtime's default of four would put a waiting kernel in front of the one it waits for) and a short
    leash on every wait (BEAT_IPC_TIMEOUT_S: a rank that gives up raises an error word, the script fails, nothing hangs)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    out = tmp_path / "out.json"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(GPU_MAX_HW_QUEUES="48", BEAT_IPC_TIMEOUT_S="20")
    run = subprocess.run([sys.executable, str(root / "tests" / "_ipc_ranks_script.py"), *[str(a) for a in args], str(out)],
                         capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    return json.loads(out.read_text())


@pytest.mark.parametrize("world", [8, 16])
def test_mailbox_allreduce_with_8_and_16_ranks_wraps_its_slot_ring_under_skew(world, tmp_path):
    """The all-reduce of the ipc transport (csrc/beat_dist.hip: ipc_allreduce_kernel -- every rank stores its 1-3 values and a
    sequence flag into EVERY rank's mailbox and adds what arrives in its own in rank order; four slots) with the rank counts
    it is written for and had never run with: 8 (the target decomposition: 512^3 in 8 slabs of 64 planes) and 16
    (BEAT_IPC_MAX_RANKS).  240 consecutive reductions of 1, 2, 3 values spanning 16 orders of magnitude, the ranks taking
    turns at being held back on the host so that the others run ahead until the slot ring stops them: every result on every
    rank equals the rank-ordered sum bit for bit (the convergence latch of the PCG relies on exactly that).  Ranks are
    threads of one process, each with its own context and streams (beat_comm_ipc_connect_local): the boxes of this pool
    admit at most six GPU processes.  Reference analogue: the reductions inside KSP.solve, src/beat/base_model.py:236, under
    the reference's `mpirun -n 2` CI (.github/workflows/main-mpi.yml:33)."""
    r = _ipc_ranks(tmp_path, "allreduce", world, 240)
    assert r["world"] == world and r["rounds"] == 240 and r["bitwise_equal_on_every_rank"] is True


def test_mailbox_ghost_plane_exchange_between_8_ranks(tmp_path):
    """60 ghost-plane exchanges between 8 ranks (threads, see above) over the mailboxes -- transfer kernels on every rank's
    side stream, sequence flags, four slots per direction, ranks held back in turn: every ghost plane holds the
    neighbour's boundary plane of THAT round (src/beat/base_model.py:242: scatter_forward)."""
    r = _ipc_ranks(tmp_path, "exchange", 8, 60)
    assert r["world"] == 8 and r["wrong_ghost_planes"] == 0


def test_decomposed_solve_on_8_ranks_over_the_mailboxes_matches_the_undivided_solve(tmp_path):
    """beat_pde_solve_dist on 8 slabs of 2-3 planes (threads, mailboxes for ghost planes AND dot products; no RCCL, no
    host in the loop) against the undivided solve: constant-coefficient rows (ghost planes of r, p formed on the ghost
    planes) and per-node rows (split SpMV, exchange of p), three solves each so that the extrapolated guess and its ghost
    planes take part; same values to 1e-9 of the scale, iteration counts within one of the undivided solve's and alike on
    all eight ranks."""
    r = _ipc_ranks(tmp_path, "solve", 8, timeout=400)
    for kind in ("constant", "per_node"):
        c = r[kind]
        assert c["max_abs_diff"] <= 1e-9 * c["scale"], (kind, c)
        assert all(its == c["iterations_ranks"][0] for its in c["iterations_ranks"]), (kind, c)
        assert all(abs(a - b) <= 1 for a, b in zip(c["iterations_ranks"][0], c["iterations_whole"])), (kind, c)


@pytest.mark.parametrize("world,transport", [(2, "callbacks"), (3, "callbacks"), (3, "ipc"), (4, "ipc"), (3, "ipc/single")])
def test_public_api_on_several_ranks_matches_one_process(world, transport, tmp_path):
    """("/single": petsc_options["ksp_cg_single_reduction"] = True, PETSc's KSPCGUseSingleReduction -- every decomposed solve
    runs the one-all-reduce iteration; iteration counts within one of the one-process run's.)
    (world 4: what the boxes of this pool admit next to the test runner and the launcher -- six processes with the GPU open
    in all, a run with five ranks was killed by the guard; 8 and 16 ranks run as threads, test_mailbox_* above.)
    (transport "ipc": the processes exchange their ghost planes ON THE DEVICE -- each maps its neighbours' mailboxes
    with hipIpcOpenMemHandle and copies its boundary planes into them on the library's side stream, ordered by
    sequence flags in device memory, and the dot products are summed through all three mailboxes (rank order, the same
    bits on every rank): nothing of the solve goes through gloo or RCCL.)
    The reference's own call sequence (geometry, stimulus, MonodomainModel, DolfinODESolver, splitting solver,
    evaluate_function, x.array) run by `world` processes, the communicator cutting the mesh into z-slabs
    (tests/_api_ranks_script.py; ranks share this box's GPU over the host-staged gloo transport): after 60 TP06 steps
    the concatenated potential, every state row and the probe values equal the one-process run to 1e-10, with the
    same PCG iteration count."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    script = str(root / "tests" / "_api_ranks_script.py")
    d1, dn = tmp_path / "one", tmp_path / "many"
    d1.mkdir()
    dn.mkdir()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BEAT_DIST_MERGED")}
    single = transport.endswith("/single")
    transport = transport.split("/")[0]
    if single:
        env["BEAT_TEST_SINGLE_REDUCTION"] = "1"
    one = subprocess.run([sys.executable, script, str(d1)], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script, str(dn)],
                          capture_output=True, text=True, timeout=300, cwd=root,
                          env=dict(env, BEAT_DIST_BACKEND="gloo", BEAT_DIST_TRANSPORT=transport))
    assert many.returncode == 0, many.stderr[-3000:]
    a = np.load(d1 / "rank0.npz")
    parts = [np.load(dn / f"rank{r}.npz") for r in range(world)]
    assert [int(p["z0"]) for p in parts] == sorted(int(p["z0"]) for p in parts) and int(parts[-1]["z1"]) == 9
    assert sum(int(p["nodes"]) for p in parts) == int(a["nodes"])
    v = np.concatenate([p["v"] for p in parts])
    S = np.concatenate([p["states"] for p in parts], axis=1)
    assert a["v"].max() > 0.0  # the stimulated corner fired
    np.testing.assert_allclose(v, a["v"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(S, a["states"], rtol=1e-10, atol=1e-12)
    for p in parts:
        np.testing.assert_allclose(p["probes"], a["probes"], rtol=0, atol=1e-10)
        assert abs(int(p["its"]) - int(a["its"])) <= (1 if single else 0)
        assert (int(p["merged_solves"]) > 0) == single
        # ECG leads (distributed mass solve + all-reduced lead integrals) and the per-rank checkpoint files
        np.testing.assert_allclose(p["leads"], a["leads"], rtol=1e-9, atol=1e-14)
        assert bool(p["roundtrip_ok"]) and abs(float(a["leads"][0])) > 0.0


def test_open_decomposed_solve_equals_the_waiting_one_bit_for_bit(tmp_path):
    """Round 5: on a decomposed grid too ``step()`` leaves its solve open (beat_pde_solve_dist_begin) and the next ionic launch
    goes behind it on every rank -- the ranks see the same all-reduced scalars on the device as on the host, so they act alike.
    Three processes over the mailbox transport, 60 TP06 steps through the public API (tests/_api_ranks_script.py, whose
    stimulated corner fires: iteration counts jump, the relaunch path is taken): every step of the default run leaves its solve
    open, none with BEAT_LAZY_KSP_DIST=0, and potentials, states, probes, ECG leads and the last iteration count are THE SAME
    BITS.  (That either equals the one-process run is test_public_api_on_several_ranks_matches_one_process.)"""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    script = str(root / "tests" / "_api_ranks_script.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BEAT_DIST_MERGED", "BEAT_LAZY_KSP", "BEAT_LAZY_KSP_DIST")}
    world, out = 3, {}
    for lazy in ("1", "0"):
        d = tmp_path / f"lazy{lazy}"
        d.mkdir()
        run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                              "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script, str(d)],
                             capture_output=True, text=True, timeout=300, cwd=root,
                             env=dict(env, BEAT_DIST_BACKEND="gloo", BEAT_DIST_TRANSPORT="ipc", BEAT_LAZY_KSP_DIST=lazy))
        assert run.returncode == 0, run.stderr[-3000:]
        out[lazy] = [np.load(d / f"rank{r}.npz") for r in range(world)]
    for a, b in zip(out["1"], out["0"]):
        assert int(a["opens"]) == 60 and int(b["opens"]) == 0
        for key in ("v", "states", "probes", "leads"):
            np.testing.assert_array_equal(a[key], b[key])
        assert int(a["its"]) == int(b["its"])
    assert max(float(p["v"].max()) for p in out["1"]) > 0.0  # the stimulated corner fired


@pytest.mark.parametrize("odespace,dim,world,percell", [("CG_2", 3, 2, False), ("DG_1", 3, 2, False), ("CG_1", 2, 2, False), ("CG_2", 2, 3, False),
                                                        ("DG_1", 2, 2, False), ("CG_1", 2, 2, True), ("CG_2", 2, 3, True)])
def test_p2_and_dg1_ode_spaces_and_nodal_fibres_on_two_ranks_match_one_process(odespace, dim, world, percell, tmp_path):
    """tests/_ode_space_ranks_script.py -- the reference's split test system (tests/test_monodomain_solver.py:33-216, which
    its CI also runs under ``mpirun -n 2``) in 3-D with the ODE on a P2 / DG1 space and the conductivity from a nodal
    fibre function -- on two processes (z-slabs; dofs on the cut interpolate across it through the exchanged ghost plane)
    against one: potential and second state at the vertices equal to 1e-11, same PCG iteration count.  ``percell``: the 2-D
    mesh with a conductivity tensor PER CELL (per-node rows cut out of the 2-D operator per rank and re-expressed for the
    kernels' (nx, 1, rows) grid, _stencil.fields_y_as_z; round 3 raised NotImplementedError here)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    script = str(root / "tests" / "_ode_space_ranks_script.py")
    d1, d2 = tmp_path / "one", tmp_path / "two"
    d1.mkdir()
    d2.mkdir()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # (dim = 2: the reference's own unit-square mesh cut into slabs of ROWS -- the kernels see it as the grid (nx, 1, ny_local))
    extra = ["percell"] if percell else []
    one = subprocess.run([sys.executable, script, str(d1), odespace, str(dim), *extra], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), script, str(d2), odespace, str(dim), *extra],
                         capture_output=True, text=True, timeout=300, cwd=root, env=dict(env, BEAT_DIST_BACKEND="gloo"))
    assert two.returncode == 0, two.stderr[-3000:]
    a = np.load(d1 / "rank0.npz")
    parts = [np.load(d2 / f"rank{r}.npz") for r in range(world)]
    assert all(int(parts[r]["z1"]) == int(parts[r + 1]["z0"]) for r in range(world - 1)) and int(parts[-1]["z1"]) == (10 if dim == 3 else 21)
    assert abs(a["v"]).max() > 0.05
    np.testing.assert_allclose(np.concatenate([p["v"] for p in parts]), a["v"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(np.concatenate([p["s"] for p in parts]), a["s"], rtol=0, atol=1e-11)
    assert sum(int(p["dofs"]) for p in parts) >= int(a["dofs"])  # DG1: the layer of cells on the cut is held twice
    assert all(int(p["its"]) == int(a["its"]) for p in parts)


def test_voxel_shell_pipeline_on_three_ranks_matches_one_process(tmp_path):
    """BASELINE configs[4] in small (tools/bench_biv.py: voxelised shell, per-voxel fibres, expand_layer markers from the
    Laplace solve, ToR-ORd endo/mid/epi through DolfinMultiODESolver, endocardial surface stimulus) on 3 processes --
    slabs cut by tissue weight, per-node operators per slab, the Laplace set-up solve replicated on every rank -- against
    one process: identical layer markers, potentials equal to 1e-9 after 20 steps, same PCG iteration counts."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    script = str(root / "tools" / "bench_biv.py")
    d1, dn = tmp_path / "one", tmp_path / "many"
    d1.mkdir()
    dn.mkdir()
    args = ["--size", "40", "--steps", "20", "--warmup", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, script, *args, "--save", str(d1)], capture_output=True, text=True, timeout=300,
                         cwd=root, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
                           "127.0.0.1", "--master-port", str(_free_port()), script, *args, "--save", str(dn)],
                          capture_output=True, text=True, timeout=300, cwd=root, env=dict(env, BEAT_DIST_BACKEND="gloo"))
    assert many.returncode == 0, many.stderr[-3000:]
    a = np.load(d1 / "rank0.npz")
    parts = [np.load(dn / f"rank{r}.npz") for r in range(3)]
    counts = [int(p["z1"]) - int(p["z0"]) for p in parts]
    assert sum(counts) == 41 and len(set(counts)) > 1  # cut by tissue weight, not evenly
    np.testing.assert_array_equal(np.concatenate([p["markers"] for p in parts]), a["markers"])
    assert a["v"].max() > 0.0
    np.testing.assert_allclose(np.concatenate([p["v"] for p in parts]), a["v"], rtol=0, atol=1e-9)
    for p in parts:
        np.testing.assert_array_equal(p["its"], a["its"])


def test_rccl_ghost_plane_exchange_on_field_slices(hip_ctx):
    """The product's halo exchange (`DiffusionSolver.start_halo / finish_halo`: torch.distributed.batch_isend_irecv over
    RCCL on slices of a field) with real RCCL point-to-point operations -- on a one-rank group a rank can only talk to
    itself, so both neighbours are rank 0: the first owned plane must arrive in one ghost plane and the last in the
    other, for a plain field and for a row of a state array (padded leading dimension), in stream order with the kernels
    that wrote the planes."""
    import torch.distributed as dist

    from beat import _stencil
    from beat._device import StateArray
    from beat._engine import DiffusionSolver, HipOps

    ctx = hip_ctx
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=ctx.device)
        created = True
    try:
        nx, ny, nz = 37, 21, 6
        plane = nx * ny

        class Interior:  # a slab with live neighbours on both sides
            rank, world, nz, lo_phys, hi_phys = 0, 2, 6, False, False

        class SelfPeer(DiffusionSolver):
            def _peer(self, group_rank):
                return 0

        ops = HipOps(ctx, (nx, ny, nz), False, False, *_stencil.stencil_tables(3, (0.1, 0.1, 0.1), np.eye(3) * 1e-3))
        solver = SelfPeer(ops, Interior(), stage_driven=True)
        rng = np.random.default_rng(4)
        states = StateArray(ctx, 3, plane * nz, plane)
        for field in (ops.new_field(), states.row_field(1)):
            for rep in range(3):
                vals = rng.standard_normal(plane * nz)
                field.set(vals)                      # written on the stream right before the exchange
                field.ghost_lo.fill_(float("nan"))
                field.ghost_hi.fill_(float("nan"))
                solver.exchange_halo(field)
                ctx.synchronize()
                np.testing.assert_array_equal(field.ghost_lo.cpu().numpy(), vals[:plane])
                np.testing.assert_array_equal(field.ghost_hi.cpu().numpy(), vals[-plane:])
                np.testing.assert_array_equal(field.numpy(), vals)
    finally:
        if created:
            dist.destroy_process_group()


class _PeriodicSelf:
    """torch.distributed look-alike of ONE rank that is its own lower and upper z-neighbour (a periodic stack of one
    slab): what the library's one-rank RCCL communicator with peers (0, 0) does, with plain device copies."""

    class ReduceOp:
        SUM = "sum"

    isend, irecv = "isend", "irecv"

    def all_reduce(self, t, op=None, group=None):
        pass  # one rank: the sum is the value

    def P2POp(self, fn, tensor, peer, group=None):
        return (fn, tensor)

    def batch_isend_irecv(self, ops):
        # DiffusionSolver.start_halo posts [send first, recv ghost_lo, send last, recv ghost_hi]
        (_, first), (_, ghost_lo), (_, last), (_, ghost_hi) = ops
        ghost_hi.copy_(first)
        ghost_lo.copy_(last)
        return []


@pytest.mark.parametrize("per_node,transport", [(pn, t) for pn in (False, True) for t in ("rccl", "rccl-serial", "ipc", "ipc+rccl")] +
                         [(False, "rccl/merged"), (False, "ipc/merged")])
def test_library_loop_with_rccl_self_neighbours_matches_stage_loop(hip_ctx, per_node, transport, monkeypatch):
    """("/merged": BEAT_DIST_MERGED=1, one all-reduce of three values per iteration; transport "rccl-serial": the same exchange on the all-reduce communicator and the compute stream, BEAT_COMM_SERIAL;
    "ipc": the ghost planes as device copies through the rank's own mailbox, ordered by its sequence flags -- every
    slot of the ring reused many times over the solves below -- and the dot products summed through the mailbox too,
    no RCCL anywhere; "ipc+rccl": those planes with RCCL's all-reduce, BEAT_IPC_ALLREDUCE=rccl.)
    The in-library solve with REAL RCCL point-to-point traffic on its side stream: a one-rank communicator whose
    lower and upper peers are the rank itself (the only multi-message topology a one-GPU box can host) makes the slab
    periodic in z -- ghost planes live on both faces, ncclSend/ncclRecv pairs in a group per SpMV, events between the
    side and the compute stream, boundary planes computed after the receive.  The stage-driven Python loop with the
    same periodic exchange done by device copies must give the same values (the same bits where both run the same
    kernels); the exchange on its own is checked too
    (plain field and a padded state-array row)."""
    from beat import _stencil
    from beat._device import StateArray
    from beat._engine import DiffusionSolver, HipOps, LibComm

    ctx = hip_ctx
    nx, ny, nz = 37, 21, 9
    plane = nx * ny
    M = np.diag([9.5e-4, 1.25e-4, 4.0e-4])
    if per_node:  # rows of the middle third of a three times taller grid: interior-type rows on both slab faces
        mt, kt = _stencil.stencil_fields(3, (nx - 1, ny - 1, 3 * nz - 1), (0.1, 0.1, 0.1), M, None, z_range=(nz, 2 * nz))
    else:
        mt, kt = _stencil.stencil_tables(3, (0.1, 0.1, 0.1), M)

    class Interior:  # a slab with live neighbours on both sides
        rank, world, nz, lo_phys, hi_phys, z0, z1 = 0, 1, 9, False, False, 0, 9

    rng = np.random.default_rng(11)
    v = -85.0 + 30.0 * rng.random(plane * nz)
    w = rng.random(plane * nz) * 1e-3
    merged = transport.endswith("/merged")
    transport = transport.split("/")[0]
    if merged:
        monkeypatch.setenv("BEAT_DIST_MERGED", "1")
    else:
        monkeypatch.delenv("BEAT_DIST_MERGED", raising=False)
    summed_by = "ipc" if transport == "ipc" else "rccl"
    if transport.startswith("ipc"):
        monkeypatch.setenv("BEAT_IPC_ALLREDUCE", summed_by)
        transport = "ipc"
    comm = LibComm(ctx, Interior(), transport=transport.split("-")[0], peers=(0, 0), serial=transport.endswith("serial"),
                   plane_doubles=plane)
    try:
        info = comm.info()
        assert info["transport"] == transport and info["rccl_ranks"] == (0 if summed_by == "ipc" else 1) and info["allreduce"] == summed_by
        # the all-reduce on its own: 1..3 values, many times over (every slot of the mailbox's ring reused)
        import torch

        for k in range(11):
            t = torch.arange(1.0, 2.0 + k % 3, dtype=torch.float64, device=ctx.device) * (k + 1)
            want = t.cpu().numpy().copy()
            comm.allreduce_sum(t)
            np.testing.assert_array_equal(t.cpu().numpy(), want)
        # the exchange on its own
        states = StateArray(ctx, 3, plane * nz, plane)
        for field in (ctx.field(plane * nz, plane), states.row_field(1)):
            vals = rng.standard_normal(plane * nz)
            field.set(vals)
            field.ghost_lo.fill_(float("nan"))
            field.ghost_hi.fill_(float("nan"))
            comm.exchange_halo(field)
            ctx.synchronize()
            np.testing.assert_array_equal(field.ghost_lo.cpu().numpy(), vals[-plane:])
            np.testing.assert_array_equal(field.ghost_hi.cpu().numpy(), vals[:plane])
            np.testing.assert_array_equal(field.numpy(), vals)
        out = {}
        for mode in ("stage", "lib"):
            ops = HipOps(ctx, (nx, ny, nz), False, False, mt, kt, per_node=per_node)
            ops.set_timestep(0.01, 0.5, 0.05)
            if mode == "lib":
                solver = DiffusionSolver(ops, Interior(), force_distributed=True, libcomm=comm)
            else:
                solver = DiffusionSolver(ops, Interior(), force_distributed=True, stage_driven=True)
                solver.dist = _PeriodicSelf()
                solver.slab = type("S", (), dict(rank=0, world=2, lo_phys=False, hi_phys=False))()  # start_halo posts both sides
            fv, fw, fx = ops.new_field(), ops.new_field(), ops.new_field()
            fv.set(v)
            fw.set(w)
            res = solver.solve(fv, [fw], [0.7], fx, rtol=1e-11, atol=1e-50, max_it=200)
            fx2 = ops.new_field()
            res2 = solver.solve(fv, [fw], [0.7], fx2, rtol=1e-11, atol=1e-50, max_it=200, defer_flush=True)
            ops.flush_pending()
            ctx.synchronize()
            out[mode] = (fx.numpy().copy(), res, fx2.numpy().copy(), res2)
        (xs, rs, xs2, _), (xl, rl, xl2, rl2) = out["stage"], out["lib"]
        assert rl.converged_reason > 0 and 3 < rl.iterations < 60 and rl.iterations == rl2.iterations
        assert abs(rl.iterations - rs.iterations) <= 1
        # the library loop runs the fused passes -- register-row kernels on constant coefficients, the tile pass on per-node rows
        # (round 4) --: ghost planes of r travel, p is formed on them; the stage loop the classic iteration with an exchange of p
        np.testing.assert_allclose(xl, xs, rtol=0, atol=1e-9 * np.abs(xs).max())
        np.testing.assert_array_equal(xl2, xl)
        np.testing.assert_array_equal(xs2, xs)
        # the periodic operator really couples the two faces: the solution differs from the Neumann-faced one
        ops_n = HipOps(ctx, (nx, ny, nz), True, True, mt, kt, per_node=per_node)
        ops_n.set_timestep(0.01, 0.5, 0.05)
        fv, fw, fx = ops_n.new_field(), ops_n.new_field(), ops_n.new_field()
        fv.set(v)
        fw.set(w)
        DiffusionSolver(ops_n, type("S", (), dict(rank=0, world=1, lo_phys=True, hi_phys=True, nz=nz))()).solve(
            fv, [fw], [0.7], fx, rtol=1e-11, atol=1e-50, max_it=200)
        assert np.abs(fx.numpy() - xl).max() > 1e-3
        # A sequence of solves from the extrapolated guess over the same RCCL communicator: the ghost planes of the guess
        # increment travel in the same group as those of v_ (two sends and two receives per face).  The periodic
        # problem has no one-rank reference, so the check is the solver's own: every solution satisfies the periodic
        # system (residual through the stage loop's exchange + beat_pde_apply) and equals the x0 = v_ solution of the same
        # right-hand side within the tolerance, in fewer iterations.
        its = {}
        for order in (0, 3):
            ops = HipOps(ctx, (nx, ny, nz), False, False, mt, kt, per_node=per_node)
            ops.set_guess_order(order)
            ops.set_timestep(0.01, 0.5, 0.05)
            solver = DiffusionSolver(ops, Interior(), force_distributed=True, libcomm=comm)
            fv, fx, fb, fa = (ops.new_field() for _ in range(4))
            its[order], sols = [], []
            for step in range(6):
                zz = np.repeat(np.arange(nz), plane)
                fv.set(v + 20.0 * np.sin(2 * np.pi * (zz + 0.3 * step) / nz) * np.cos(0.05 * np.arange(plane * nz) % 7))
                res = solver.solve(fv, [], [], fx, rtol=1e-10, atol=1e-50, max_it=200, defer_flush=bool(step % 2))
                ops.flush_pending()
                assert res.converged_reason > 0
                its[order].append(res.iterations)
                sols.append(fx.numpy().copy())
                for f in (fv, fx):
                    comm.exchange_halo(f)
                ops.apply(1, fv, fb)
                ops.apply(0, fx, fa)
                ctx.synchronize()
                r = fb.numpy() - fa.numpy()
                assert np.linalg.norm(r) <= 2e-10 * np.linalg.norm(fb.numpy())
            its[order] = (its[order], sols)
        for a, b in zip(its[0][1], its[3][1]):
            np.testing.assert_allclose(b, a, rtol=0, atol=1e-8 * np.abs(a).max())
        assert sum(its[3][0][2:]) < sum(its[0][0][2:])
        # event timing of the communication inside the solve (beat_comm_profile): one exchange per iteration + the one
        # before the right-hand side, two all-reduces per iteration + one; every span has a non-negative duration
        comm.profile(True)
        res = solver.solve(fv, [], [], fx, rtol=1e-10, atol=1e-50, max_it=200)
        comm.profile(False)
        prof = comm.profile_read()
        if merged:  # one per enqueued pass (the iterations, the pass that finds the residual converged, one spare) + the right-hand side's
            assert res.iterations + 2 <= prof["allreduce_count"] <= res.iterations + 4
            assert int(ctx.lib.beat_comm_merged_solves(comm.handle)) > 0
        assert prof["allreduce_count"] >= (1 if merged else 2) * res.iterations + 1 and prof["halo_count"] >= res.iterations + 1
        assert prof["halo_ms"] > 0.0 and prof["allreduce_ms"] > 0.0 and prof["halo_stall_ms"] >= 0.0
        assert (prof["halo_stall_count"] > 0) == (transport != "rccl-serial")
    finally:
        comm.close()


def test_non_convergence_is_reported_not_raised(hip_ctx):
    """A solve that runs out of iterations returns converged_reason = -3 (KSP_DIVERGED_ITS) with the last iterate in x
    -- fused solve and in-library decomposed solve -- instead of raising (src/beat/base_model.py:23-30,
    telemetry.py:67-76 read the reason from the KSP)."""
    from beat import _stencil
    from beat._engine import DiffusionSolver, HipOps, Slab

    ctx = hip_ctx
    nx, ny, nz = 24, 20, 12
    mt, kt = _stencil.stencil_tables(3, (0.1, 0.1, 0.1), np.eye(3) * 1e-3)
    rng = np.random.default_rng(1)
    v = rng.standard_normal(nx * ny * nz)
    for force in (False, True):
        ops = HipOps(ctx, (nx, ny, nz), True, True, mt, kt)
        ops.set_timestep(0.01, 0.5, 0.05)
        solver = DiffusionSolver(ops, Slab(nz), force_distributed=force)
        fv, fx = ops.new_field(), ops.new_field()
        fv.set(v)
        res = solver.solve(fv, [], [], fx, rtol=1e-14, atol=1e-50, max_it=2)
        assert res.converged_reason == -3 and res.iterations == 2 and res.residual_norm > 0.0
        x2 = fx.numpy().copy()
        assert np.isfinite(x2).all() and not np.array_equal(x2, v)
        res = solver.solve(fv, [], [], fx, rtol=1e-11, atol=1e-50, max_it=200)
        assert res.converged_reason > 0


def test_ipc_communicator_set_up_errors_are_reported(hip_ctx):
    """The two-step set-up of the ipc transport through the C ABI (include/beat_hip.h): a communicator that sums through the
    mailboxes refuses to reduce before beat_comm_ipc_connect_all has been called; connect_all wants one handle per rank, each
    from the rank at its position; connecting twice, an all-reduce of more than four values and both RCCL id and callback
    at creation are errors -- reported through beat_last_error, nothing left half-built."""
    import ctypes as C

    import torch

    from beat import _hip

    ctx = hip_ctx
    lib = ctx.lib
    plane = 64
    mine = C.create_string_buffer(_hip.IPC_HANDLE_BYTES)
    h = C.c_void_p()
    _hip.check(lib.beat_comm_create_ipc(ctx.handle, 0, 1, 0, 0, plane, None, None, None, mine, C.byref(h)))
    try:
        t = torch.ones(3, dtype=torch.float64, device=ctx.device)
        with pytest.raises(RuntimeError, match="connect_all"):
            _hip.check(lib.beat_comm_allreduce_sum(h, C.c_void_p(t.data_ptr()), 3))
        with pytest.raises(RuntimeError, match="handles for 1 ranks"):
            _hip.check(lib.beat_comm_ipc_connect_all(h, mine.raw + mine.raw, 2))
        _hip.check(lib.beat_comm_ipc_connect_all(h, mine.raw, 1))
        with pytest.raises(RuntimeError, match="already connected"):
            _hip.check(lib.beat_comm_ipc_connect_all(h, mine.raw, 1))
        with pytest.raises(RuntimeError, match="already connected"):
            _hip.check(lib.beat_comm_ipc_connect(h, None, None))
        _hip.check(lib.beat_comm_allreduce_sum(h, C.c_void_p(t.data_ptr()), 3))
        ctx.synchronize()
        assert t.cpu().tolist() == [1.0, 1.0, 1.0]  # a world of one rank
        five = torch.ones(5, dtype=torch.float64, device=ctx.device)
        with pytest.raises(RuntimeError, match=r"1\.\.4 values"):
            _hip.check(lib.beat_comm_allreduce_sum(h, C.c_void_p(five.data_ptr()), 5))
    finally:
        _hip.check(lib.beat_comm_destroy(h))
    # a handle that is not the rank's own at its position
    other = C.create_string_buffer(_hip.IPC_HANDLE_BYTES)
    h2 = C.c_void_p()
    _hip.check(lib.beat_comm_create_ipc(ctx.handle, 1, 2, 0, -1, plane, None, None, None, other, C.byref(h2)))
    try:
        with pytest.raises(RuntimeError, match="not rank 0's ipc handle"):
            _hip.check(lib.beat_comm_ipc_connect_all(h2, other.raw + other.raw, 2))
    finally:
        _hip.check(lib.beat_comm_destroy(h2))
    ids = C.create_string_buffer(2 * _hip.UNIQUE_ID_BYTES)
    cb = _hip.ALLREDUCE_FN(lambda user, values, count: 0)
    h3 = C.c_void_p()
    with pytest.raises(RuntimeError, match="at most one of"):
        _hip.check(lib.beat_comm_create_ipc(ctx.handle, 0, 1, -1, -1, plane, ids, C.cast(cb, C.c_void_p), None, mine, C.byref(h3)))
