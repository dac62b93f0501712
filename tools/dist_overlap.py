#!/usr/bin/env python3
"""Does the ghost-plane exchange of the decomposed solve travel BEHIND the interior stencil?  Measured on one GPU.

One process that is its own lower and upper z-neighbour (a periodic stack of one slab -- the topology a one-GPU box can
host with real transports, tests/test_distributed_gpu.py) runs the in-library decomposed solve (beat_pde_solve_dist) on
the slab one of 8 ranks owns at 512^3 (512 x 512 x 64 nodes) over each transport:

  rccl         ncclSend/ncclRecv groups on the library's side stream (two communicators)
  rccl-serial  the same exchange on the compute stream, one communicator (BEAT_COMM_SERIAL)
  ipc          transfer kernels through the rank's mailbox, ordered by sequence flags in device memory

and reports, per solve, from the library's own event timing (beat_comm_profile): ms the transfers took on their stream,
ms the compute stream stood waiting for ghost planes (= what the interior kernels did NOT hide), ms in all-reduces.

    python3 tools/dist_overlap.py [--transport ipc] [--nz 64] [--solves 20] [--json out.json]
    rocprofv3 --kernel-trace -d DIR -o trace --output-format csv -- python3 tools/dist_overlap.py --transport ipc
    python3 tools/dist_overlap.py --trace DIR/trace_kernel_trace.csv      # overlap of transfer and stencil kernels
"""
import argparse
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd"), str(ROOT)]


def analyse_trace(path):
    """Per transfer kernel of the trace: how much of its duration lies inside a stencil kernel of the compute stream."""
    rows = list(csv.DictReader(open(path)))
    xfer = [r for r in rows if "ipc_xfer_kernel" in r["Kernel_Name"] or "ncclDevKernel" in r["Kernel_Name"]
            or "rccl" in r["Kernel_Name"].lower()]
    sten = [r for r in rows if "rr_kernel" in r["Kernel_Name"] or "var_spmv_kernel" in r["Kernel_Name"]]
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sten)
    tot = cov = 0
    for r in xfer:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        tot += b - a
        for s, e in iv:
            if e <= a:
                continue
            if s >= b:
                break
            cov += min(b, e) - max(a, s)
    out = {"transfer_kernels": len(xfer), "stencil_kernels": len(sten), "transfer_total_us": tot / 1e3,
           "transfer_inside_stencil_us": cov / 1e3, "fraction_hidden": (cov / tot) if tot else None,
           "mean_transfer_us": (tot / len(xfer) / 1e3) if xfer else None}
    print(json.dumps(out))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--transport", default="all", choices=["all", "rccl", "rccl-serial", "ipc", "ipc-coarse"])
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--nz", type=int, default=64)
    ap.add_argument("--solves", type=int, default=20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--trace", default=None, help="analyse a rocprofv3 kernel trace instead of running")
    args = ap.parse_args()
    if args.trace:
        analyse_trace(args.trace)
        return
    import numpy as np
    import torch

    from beat import _stencil
    from beat._device import Context
    from beat._engine import DiffusionSolver, HipOps, LibComm

    ctx = Context(0)
    nx = ny = args.n
    nz = args.nz
    plane = nx * ny
    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M = 9.5301e-4 * np.outer(f0, f0) + 1.2576e-4 * (np.eye(3) - np.outer(f0, f0))
    mt, kt = _stencil.stencil_tables(3, (0.1, 0.1, 0.1), M)

    class Interior:  # a slab with live neighbours on both sides
        rank, world, lo_phys, hi_phys, z0 = 0, 1, False, False, 0
    Interior.nz = nz
    Interior.z1 = nz

    results = {}
    names = ["rccl", "rccl-serial", "ipc", "ipc-coarse"] if args.transport == "all" else [args.transport]
    import os

    for name in names:
        os.environ["BEAT_IPC_COARSE"] = "1" if name == "ipc-coarse" else "0"  # mailbox in plain hipMalloc memory (one GPU: coherent anyway)
        comm = LibComm(ctx, Interior(), transport=name.split("-")[0], peers=(0, 0), serial=name.endswith("serial"), plane_doubles=plane)
        ops = HipOps(ctx, (nx, ny, nz), False, False, mt, kt)
        ops.set_guess_order(0)  # every solve starts from x0 = v_: the same 5-6 iterations each time
        ops.set_timestep(0.01, 0.5, 0.01)
        solver = DiffusionSolver(ops, Interior(), force_distributed=True, libcomm=comm)
        v, x = ops.new_field(), ops.new_field()
        zc = torch.arange(nz, device=ctx.device, dtype=torch.float64)
        yc = torch.arange(ny, device=ctx.device, dtype=torch.float64)
        xc = torch.arange(nx, device=ctx.device, dtype=torch.float64)
        r2 = ((zc[:, None, None] - nz / 2) ** 2 + (yc[None, :, None] - ny / 2) ** 2 + (xc[None, None, :] - nx / 2) ** 2) * 0.01
        v.data.view(nz, ny, nx).copy_(-85.0 + 60.0 * torch.exp(-r2 / 8.0))
        del r2
        # the exchange on its own, nothing else on the GPU: 50 exchanges of one field back to back
        comm.exchange_halo(v)
        ctx.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(50):
            comm.exchange_halo(v)
        ev[1].record()
        torch.cuda.synchronize()
        alone_us = 1e3 * ev[0].elapsed_time(ev[1]) / 50
        for _ in range(3):
            res = solver.solve(v, [], [], x, rtol=1e-8, atol=1e-50, max_it=200)
        ctx.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(args.solves):
            res = solver.solve(v, [], [], x, rtol=1e-8, atol=1e-50, max_it=200)
        ev[1].record()
        torch.cuda.synchronize()
        plain_ms = ev[0].elapsed_time(ev[1]) / args.solves
        comm.profile(True)
        for _ in range(args.solves):
            res = solver.solve(v, [], [], x, rtol=1e-8, atol=1e-50, max_it=200)
        comm.profile(False)
        p = comm.profile_read()
        k = args.solves
        results[name] = {"ms_per_solve": plain_ms, "iterations": res.iterations, "exchange_alone_us": alone_us,
                         "transfer_ms_per_solve": p["halo_ms"] / k, "exchanges_per_solve": p["halo_count"] / k,
                         "mean_transfer_us": 1e3 * p["halo_ms"] / max(1, p["halo_count"]),
                         "stall_ms_per_solve": p["halo_stall_ms"] / k,
                         "allreduce_ms_per_solve": p["allreduce_ms"] / k, "allreduces_per_solve": p["allreduce_count"] / k}
        print(name, json.dumps(results[name]), flush=True)
        comm.close()
        del solver, ops, v, x
    # the same slab as ONE rank with physical faces: what the solve costs without any exchange
    ops = HipOps(ctx, (nx, ny, nz), True, True, mt, kt)
    ops.set_guess_order(0)
    ops.set_timestep(0.01, 0.5, 0.01)
    v, x = ops.new_field(), ops.new_field()
    v.data.fill_(-85.0)
    v.data[: plane * 8].fill_(-20.0)
    for _ in range(3):
        res = ops.solve_single(v, [], [], x, 1e-8, 1e-50, 200)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.solves):
        res = ops.solve_single(v, [], [], x, 1e-8, 1e-50, 200)
    ev[1].record()
    torch.cuda.synchronize()
    results["single_slab_fused"] = {"ms_per_solve": ev[0].elapsed_time(ev[1]) / args.solves, "iterations": res.iterations}
    print("single", json.dumps(results["single_slab_fused"]), flush=True)
    if args.json:
        Path(args.json).write_text(json.dumps({"grid": [nx, ny, nz], "solves": args.solves, "results": results}, indent=1))


if __name__ == "__main__":
    main()
