#!/usr/bin/env python3
"""What do the collectives of the decomposed solve cost as the rank count grows?  Measured on ONE GPU, where RCCL cannot host
more than one rank: the library's own mailbox transport (csrc/beat_dist.hip, `ipc`) between W PROCESSES sharing the GPU
(W <= 4 next to this launcher and torchrun: the pool admits six processes with the GPU open; with five ranks the figures are
those of an oversubscribed GPU: 81 us per all-reduce, 250 us per exchange).  (8 and 16 ranks run as THREADS of one process in
tests/_ipc_ranks_script.py -- for correctness only: the threads share one interpreter lock, and what a timing of them shows
is that lock: 117 us per all-reduce and 1.5 ms per exchange with 8 threads.)

Per rank count: us per mailbox all-reduce of 1-3 doubles and us per ghost-plane exchange (one 512 x 512 plane per face),
from the library's event timing around every operation (beat_comm_profile: on the stream the operation runs on, waiting
for the slowest rank included), 400 operations each, enqueued back to back.

    python3 tools/dist_ranks.py [--json out.json]            # launcher: runs W = 2, 3, 4 processes
    (internal) torchrun ... tools/dist_ranks.py --rank-mode   # one rank of a process run

Between GPUs the stores and flags travel over xGMI instead of through one GPU's L2 / fabric; what this table pins is the
part that does not depend on the link: the kernels, their launch, the flag protocol and its growth with W."""
import argparse
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd"), str(ROOT)]
PLANE = 512 * 512
OPS = 400


def rank_main(out_path):
    import torch
    import torch.distributed as dist

    from beat._device import Context, Field
    from beat._engine import LibComm, Slab

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = Context(0)
    slab = Slab(4 * world, rank, world)
    comm = LibComm(ctx, slab, dist, None, "ipc", plane_doubles=PLANE)
    buf = ctx.zeros(3)
    f = Field(ctx, 4 * PLANE, PLANE)
    res = {}
    for what in ("allreduce", "exchange"):
        for _ in range(20):  # warm-up
            comm.allreduce_sum(buf[:2]) if what == "allreduce" else comm.exchange_halo(f)
        ctx.synchronize()
        dist.barrier()
        comm.profile(True)
        for k in range(OPS):
            comm.allreduce_sum(buf[: 1 + k % 3]) if what == "allreduce" else comm.exchange_halo(f)
        p = comm.profile_read()
        comm.profile(False)
        if what == "allreduce":
            res["us_per_allreduce"] = p["allreduce_ms"] / max(1, p["allreduce_count"]) * 1e3
        else:
            res["us_per_exchange"] = p["halo_ms"] / max(1, p["halo_count"]) * 1e3
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        Path(out_path).write_text(json.dumps({"world": world, "us_per_allreduce": max(g["us_per_allreduce"] for g in gathered),
                                              "us_per_exchange": max(g["us_per_exchange"] for g in gathered),
                                              "per_rank": gathered}))
    ctx.synchronize()
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def free_port():
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank-mode", default=None)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    if args.rank_mode:
        rank_main(args.rank_mode)
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    rows = []
    tmp = ROOT / "gpurun_out" / "dist_ranks_tmp.json"
    tmp.parent.mkdir(exist_ok=True)
    for world in (2, 3, 4):
        run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                              "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve()), "--rank-mode", str(tmp)],
                             capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(env, BEAT_DIST_BACKEND="gloo"))
        if run.returncode != 0:
            print(run.stderr[-2000:], file=sys.stderr)
            rows.append({"world": world, "ranks_are": "processes", "error": run.stderr[-300:]})
            continue
        r = json.loads(tmp.read_text())
        rows.append({"world": world, "ranks_are": "processes", "us_per_allreduce": r["us_per_allreduce"], "us_per_exchange": r["us_per_exchange"]})
        print(rows[-1], flush=True)
        if args.json:
            Path(args.json).write_text(json.dumps(rows, indent=1))
    tmp.unlink(missing_ok=True)
    if args.json:
        Path(args.json).write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
