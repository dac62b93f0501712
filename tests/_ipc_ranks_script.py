"""The mailbox transport of csrc/beat_dist.hip with MANY ranks on one GPU: `world` ranks as threads of this process, each
with its own context, compute stream and side stream, connected with beat_comm_ipc_connect_local (the boxes of this pool
admit six GPU processes; the target shape is 8 ranks, the all-reduce is written for 16).

  python tests/_ipc_ranks_script.py allreduce <world> <rounds> <out.json>   -- `rounds` consecutive all-reduces of 1-3 values,
        one rank after the other held back on the host (so that the others run ahead as far as the four slots let them and the
        slot ring wraps under skew); every rank's every result compared bit for bit with the rank-ordered sum
  python tests/_ipc_ranks_script.py exchange <world> <rounds> <out.json>    -- ghost-plane exchanges of a (nx ny) plane per face
  python tests/_ipc_ranks_script.py solve <world> <out.json>               -- beat_pde_solve_dist on `world` slabs against the
        undivided solve (constant and per-node rows), iteration counts and values

Run by tests/test_distributed_gpu.py with GPU_MAX_HW_QUEUES raised: spinning kernels of 2 x world streams must not be
multiplexed onto four hardware queues.  (Mode "time" reports the library's event timing per operation; with threads it measures
the interpreter lock -- 117 us per all-reduce with 8 threads --, which is why profiles/r04_dist_ranks.md is made with processes,
tools/dist_ranks.py.)"""
import ctypes as C
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd"), str(ROOT)]


TIMED_PLANE = False


def run_ranks(world, body):
    """body(rank, ctx, libcomm, barrier) in `world` threads; returns the list of results or raises the first error."""
    from beat._device import Context
    from beat._engine import LibComm, Slab

    barrier = threading.Barrier(world)
    registry, results, errors = {}, [None] * world, []
    plane = 512 * 512 if TIMED_PLANE else 64 * 48

    def rank_main(rank):
        import torch

        try:
            # a stream of its own per rank (a Context adopts the thread's current stream): on the shared default stream a
            # kernel that waits for another rank's flag would sit in front of the kernel that raises it
            with torch.cuda.stream(torch.cuda.Stream()):
                ctx = Context()
                slab = Slab(4 * world, rank, world)
                comm = LibComm.ipc_in_process(ctx, slab, plane, registry, barrier)
                results[rank] = body(rank, ctx, comm, barrier, plane)
                ctx.synchronize()
                barrier.wait(timeout=300)  # nobody frees a mailbox another rank may still write to
                comm.close()
        except Exception as exc:  # noqa: BLE001
            errors.append((rank, repr(exc)))
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    if errors:
        raise RuntimeError(f"ranks failed: {errors[:4]}")
    return results


def mode_allreduce(world, rounds, timed=False):
    rng = np.random.default_rng(17)
    vals = rng.standard_normal((rounds, world, 3)) * 10.0 ** rng.integers(-8, 9, (rounds, world, 1))

    def body(rank, ctx, comm, barrier, plane):
        torch = ctx.torch
        got = np.zeros((rounds, 3))
        bufs = [ctx.zeros(3) for _ in range(8)]
        t0 = time.perf_counter()
        for k in range(rounds):
            cnt = 1 + k % 3
            if not timed and k % 7 == 3 and rank == (k // 7) % world:
                time.sleep(0.004)  # this rank falls behind: the others run ahead until the slot ring stops them
            b = bufs[k % 8]
            b[:cnt].copy_(ctx.from_numpy(vals[k, rank, :cnt].copy()))
            comm.allreduce_sum(b[:cnt])
            if timed:
                continue
            got[k, :cnt] = b[:cnt].cpu().numpy()
        ctx.synchronize()
        return got, time.perf_counter() - t0

    if timed:
        def body(rank, ctx, comm, barrier, plane):  # noqa: F811 -- library event timing around every operation
            b = ctx.zeros(3)
            for _ in range(20):
                comm.allreduce_sum(b[:2])
            ctx.synchronize()
            barrier.wait(timeout=120)
            comm.profile(True)
            for k in range(rounds):
                comm.allreduce_sum(b[: 1 + k % 3])
            p = comm.profile_read()
            comm.profile(False)
            return None, p["allreduce_ms"] / max(1, p["allreduce_count"]) * 1e-3

    res = run_ranks(world, body)
    if timed:
        return {"us_per_allreduce": max(r[1] for r in res) * 1e6}
    ok = True
    for k in range(rounds):
        cnt = 1 + k % 3
        ref = np.zeros(3)
        for r in range(world):  # rank order, as the kernel adds them
            ref[:cnt] = ref[:cnt] + vals[k, r, :cnt]
        for r in range(world):
            ok = ok and np.array_equal(res[r][0][k, :cnt], ref[:cnt])
    return {"world": world, "rounds": rounds, "bitwise_equal_on_every_rank": bool(ok)}


def mode_exchange(world, rounds, timed=False):
    def body(rank, ctx, comm, barrier, plane):
        from beat._device import Field

        nzl = 4
        f = Field(ctx, nzl * plane, plane)
        bad = 0
        t0 = time.perf_counter()
        for k in range(rounds):
            if not timed:
                f.data.copy_(ctx.from_numpy(np.repeat(1000.0 * k + 10.0 * rank + np.arange(nzl), plane).astype(np.float64)))
                f.ghost_lo.fill_(float("nan"))
                f.ghost_hi.fill_(float("nan"))
                if k % 5 == 2 and rank == (k // 5) % world:
                    time.sleep(0.003)
            comm.exchange_halo(f)
            if timed:
                continue
            ctx.synchronize()
            lo, hi = f.ghost_lo.cpu().numpy(), f.ghost_hi.cpu().numpy()
            if rank > 0:
                bad += int(not np.all(lo == 1000.0 * k + 10.0 * (rank - 1) + nzl - 1))
            if rank < world - 1:
                bad += int(not np.all(hi == 1000.0 * k + 10.0 * (rank + 1)))
        ctx.synchronize()
        return bad, time.perf_counter() - t0

    if timed:
        def body(rank, ctx, comm, barrier, plane):  # noqa: F811
            from beat._device import Field

            f = Field(ctx, 4 * plane, plane)
            for _ in range(20):
                comm.exchange_halo(f)
            ctx.synchronize()
            barrier.wait(timeout=120)
            comm.profile(True)
            for _ in range(rounds):
                comm.exchange_halo(f)
            p = comm.profile_read()
            comm.profile(False)
            return 0, p["halo_ms"] / max(1, p["halo_count"]) * 1e-3

    res = run_ranks(world, body)
    if timed:
        return {"us_per_exchange": max(r[1] for r in res) * 1e6}
    return {"world": world, "rounds": rounds, "wrong_ghost_planes": int(sum(r[0] for r in res))}


def mode_solve(world):
    from beat import _stencil
    from beat._device import Context
    from beat._engine import DiffusionSolver, HipOps, LibComm, Slab

    nx, ny, nz = 40, 33, max(19, 2 * world + 3)
    cells, h = (nx - 1, ny - 1, nz - 1), (0.1, 0.1, 0.1)
    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M = 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
    plane = nx * ny
    rng = np.random.default_rng(5)
    v = -85.0 + 30.0 * rng.random(nx * ny * nz)
    out = {}
    for per_node in (False, True):
        active = None
        if per_node:
            cc = np.stack(np.meshgrid(np.arange(nz - 1), np.arange(ny - 1), np.arange(nx - 1), indexing="ij"), -1).reshape(-1, 3)
            active = ((cc - np.array([nz // 2, 16, 20])) ** 2).sum(axis=1) < 15**2

        def operators(z_range):
            return _stencil.stencil_fields(3, cells, h, M, active, z_range=z_range) if per_node else _stencil.stencil_tables(3, h, M)

        def solve(ctx, slab, libcomm, barrier=None):
            ops = HipOps(ctx, (nx, ny, slab.nz), slab.lo_phys, slab.hi_phys, *operators((slab.z0, slab.z1)), per_node=per_node)
            ops.set_timestep(0.01, 0.5, 0.05)
            solver = DiffusionSolver(ops, slab) if libcomm is None else DiffusionSolver(ops, slab, force_distributed=True, libcomm=libcomm)
            fv, fx = ops.new_field(), ops.new_field()
            fv.set(v[slab.z0 * plane : slab.z1 * plane])
            ctx.synchronize()
            # (threads of one process: the set-up above uses synchronous copies, which wait for every stream of the process --
            # also for another rank's kernel that is waiting for THIS rank.  Between processes no such coupling exists; here
            # nobody starts to communicate before everybody is set up, and nobody reads back before everybody is done.)
            if barrier is not None:
                barrier.wait(timeout=300)
            its = []
            for _ in range(3):  # (the second and third start from the extrapolated guess: ghost planes of e travel too)
                res = solver.solve(fv, [], [], fx, rtol=1e-11, atol=1e-50, max_it=200)
                its.append(res.iterations)
            ops.flush_pending()
            ctx.synchronize()
            if barrier is not None:
                barrier.wait(timeout=300)
            return fx.numpy().copy(), its

        whole, its_whole = solve(Context(), Slab(nz), None)
        barrier = threading.Barrier(world)
        registry, parts, errors = {}, [None] * world, []

        def rank_main(rank):
            import torch

            try:
                with torch.cuda.stream(torch.cuda.Stream()):
                    ctx = Context()
                    slab = Slab(nz, rank, world)
                    comm = LibComm.ipc_in_process(ctx, slab, plane, registry, barrier)
                    parts[rank] = solve(ctx, slab, comm, barrier)
                    barrier.wait(timeout=300)
                    comm.close()
            except Exception as exc:  # noqa: BLE001
                errors.append((rank, repr(exc)))
                barrier.abort()

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        if errors:
            raise RuntimeError(f"ranks failed: {errors[:4]}")
        x = np.concatenate([p[0] for p in parts])
        out["per_node" if per_node else "constant"] = {
            "max_abs_diff": float(np.abs(x - whole).max()), "scale": float(np.abs(whole).max()), "iterations_whole": its_whole,
            "iterations_ranks": [p[1] for p in parts]}
    return out


def main():
    mode, world = sys.argv[1], int(sys.argv[2])
    if mode == "allreduce":
        res = mode_allreduce(world, int(sys.argv[3]))
    elif mode == "exchange":
        res = mode_exchange(world, int(sys.argv[3]))
    elif mode == "solve":
        res = mode_solve(world)
    elif mode == "time":
        global TIMED_PLANE
        TIMED_PLANE = True
        rounds = int(sys.argv[3])
        res = {"world": world, **mode_allreduce(world, rounds, timed=True), **mode_exchange(world, rounds, timed=True)}
    else:
        raise SystemExit(f"unknown mode {mode}")
    Path(sys.argv[-1]).write_text(json.dumps(res))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
