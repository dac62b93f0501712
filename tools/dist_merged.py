#!/usr/bin/env python3
"""One all-reduce per PCG iteration (BEAT_DIST_MERGED / beat_pde_set_single_reduction, csrc/beat_pde_rr.hip) against two, on the
slab ONE of 8 ranks owns at 512^3 (512 x 512 x 64 planes, constant anisotropic coefficients, rtol 1e-8), measured on ONE GPU:

  A  one process whose lower and upper neighbour are the rank itself (the mailbox transport, peers = (0, 0)): the compute
     side alone -- what the two extra/fewer stencil passes and the different vector traffic cost with reductions that are
     almost free (one rank: ~9 us each);
  B  W = 2..4 processes sharing the GPU, each with such a slab, over the mailboxes: the same comparison with W real ranks in
     every all-reduce (the processes time-share the GPU, so a solve takes ~W times as long as on a GPU of its own; what the
     line shows is the all-reduce count per solve and what the library's event timing says they took).

Per case: ms per solve (HIP events around 30 solves of the same right-hand side, enqueued back to back), PCG iterations,
all-reduces per solve and the time they took (beat_comm_profile, waiting for the slowest rank included).

    python3 tools/dist_merged.py [--json out.json] [--planes 64] [--worlds 2,3,4]
    (internal) torchrun ... tools/dist_merged.py --rank-mode out.json
"""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd"), str(ROOT), str(ROOT / "tools")]
NX = 512
SOLVES = 30


def measure(ctx, comm, slab_like, planes, rank, world, amplitude):
    """classic and single-reduction solves of one right-hand side on this rank's slab; returns {mode: figures}."""
    import torch

    from beat import _stencil
    from beat._engine import DiffusionSolver, HipOps

    plane = NX * NX
    f0 = np.array([np.cos(np.pi / 6.0), np.sin(np.pi / 6.0), 0.0])
    M = 9.5301e-4 * np.outer(f0, f0) + 1.2576e-4 * (np.eye(3) - np.outer(f0, f0))  # bench.py's operator: h 0.1 mm, C_m 0.01, theta 0.5, dt 0.01
    mt, kt = _stencil.stencil_tables(3, (0.1, 0.1, 0.1), M)
    ops = HipOps(ctx, (NX, NX, planes), slab_like.lo_phys, slab_like.hi_phys, mt, kt)
    ops.set_timestep(0.01, 0.5, 0.01)
    solver = DiffusionSolver(ops, slab_like, force_distributed=True, libcomm=comm)
    fv, fx = ops.new_field(), ops.new_field()
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(100 + rank)
    idx = torch.arange(plane * planes, device=ctx.device, dtype=torch.float64)
    xs, ys = idx % NX, torch.div(idx, NX, rounding_mode="floor") % NX
    zs = torch.div(idx, plane, rounding_mode="floor") + rank * planes
    zc = 0.5 * planes * world
    v = -85.0 + 100.0 * torch.exp(-((xs - 200.0) ** 2 + (ys - 256.0) ** 2 + (zs - zc) ** 2) * 0.01 / 0.18)
    v += amplitude * (torch.rand(plane * planes, generator=gen, device=ctx.device, dtype=torch.float64) - 0.5)
    fv.data.copy_(v)
    del idx, xs, ys, zs, v
    out = {}
    for mode in ("two_reductions", "single_reduction"):
        ops.set_single_reduction(mode == "single_reduction")
        for _ in range(3):
            res = solver.solve(fv, [], [], fx, rtol=1e-8, atol=1e-50, max_it=200)
        ctx.synchronize()
        comm.profile(True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(SOLVES):
            res = solver.solve(fv, [], [], fx, rtol=1e-8, atol=1e-50, max_it=200)
        b.record()
        ctx.synchronize()
        p = comm.profile_read()
        comm.profile(False)
        out[mode] = {"ms_per_solve": a.elapsed_time(b) / SOLVES, "iterations": int(res.iterations), "reason": int(res.converged_reason),
                     "allreduces_per_solve": p["allreduce_count"] / SOLVES, "allreduce_us_each": p["allreduce_ms"] / max(1, p["allreduce_count"]) * 1e3,
                     "allreduce_ms_per_solve": p["allreduce_ms"] / SOLVES, "exchanges_per_solve": p["halo_count"] / SOLVES}
    ops.set_single_reduction(None)
    return out


def self_neighbour(planes, amplitude):
    from beat._device import Context
    from beat._engine import LibComm

    ctx = Context(0)

    class Interior:
        rank, world, nz, lo_phys, hi_phys, z0, z1 = 0, 1, planes, False, False, 0, planes

    comm = LibComm(ctx, Interior(), transport="ipc", peers=(0, 0), plane_doubles=NX * NX)
    try:
        return measure(ctx, comm, Interior(), planes, 0, 1, amplitude)
    finally:
        comm.close()


def rank_main(out_path, planes, amplitude):
    import torch.distributed as dist

    from beat._device import Context
    from beat._engine import LibComm, Slab

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = Context(0)
    slab = Slab(planes * world, rank, world)
    comm = LibComm(ctx, slab, dist, None, "ipc", plane_doubles=NX * NX)
    res = measure(ctx, comm, slab, planes, rank, world, amplitude)
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        worst = {m: {k: max(g[m][k] for g in gathered) for k in gathered[0][m]} for m in gathered[0]}
        Path(out_path).write_text(json.dumps({"world": world, **worst}))
    ctx.synchronize()
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank-mode", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--planes", type=int, default=64)
    ap.add_argument("--worlds", default="2,3,4")
    ap.add_argument("--noise", type=float, default=1.0, help="mV of white noise on the potential (more noise, more iterations)")
    args = ap.parse_args()
    if args.rank_mode:
        rank_main(args.rank_mode, args.planes, args.noise)
        return
    from dist_ranks import free_port

    rows = []

    def show(row):
        rows.append(row)
        print(json.dumps(row), flush=True)
        if args.json:
            Path(args.json).write_text(json.dumps(rows, indent=1))

    for noise in (args.noise, 20.0 * args.noise):
        show({"ranks": "1 process, its own neighbour on both faces", "planes": args.planes, "noise_mV": noise, **self_neighbour(args.planes, noise)})
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    tmp = ROOT / "gpurun_out" / "dist_merged_tmp.json"
    tmp.parent.mkdir(exist_ok=True)
    for world in [int(w) for w in args.worlds.split(",") if w]:
        run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                              "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve()), "--rank-mode", str(tmp),
                              "--planes", str(args.planes), "--noise", str(args.noise)],
                             capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, BEAT_DIST_BACKEND="gloo"))
        if run.returncode != 0:
            print(run.stderr[-2000:], file=sys.stderr)
            show({"ranks": f"{world} processes on one GPU", "error": run.stderr[-300:]})
            continue
        show({"ranks": f"{world} processes on one GPU", "planes": args.planes, "noise_mV": args.noise, **json.loads(tmp.read_text())})
    tmp.unlink(missing_ok=True)


if __name__ == "__main__":
    main()
