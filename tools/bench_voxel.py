#!/usr/bin/env python3
"""Per-node-coefficient diffusion path on a voxelised ellipsoidal shell (BASELINE.json configs[4] in small):
python tools/bench_voxel.py [--n 256]   -- kernel times (HIP events) and algorithmic GB/s."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256, help="voxels per axis of the bounding box")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dense", action="store_true", help="every voxel active (a box with a fibre field): per-node rows without a mask")
    ap.add_argument("--slab", action="store_true", help="both z faces belong to neighbouring ranks: time the SpMV of the decomposed solve "
                    "(the planes that need no ghost data, then the two slab-boundary planes + the reduction)")
    args = ap.parse_args()
    import torch

    from beat._device import Context
    from beat._engine import HipOps

    n, h = args.n, 0.25
    ax = (np.arange(n) + 0.5) / n - 0.5
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
    ro = np.sqrt((X / 0.48) ** 2 + (Y / 0.44) ** 2 + (Z / 0.48) ** 2)
    ri = np.sqrt((X / 0.30) ** 2 + (Y / 0.27) ** 2 + (Z / 0.30) ** 2)
    mask = (ro < 1.0) & (ri > 1.0) & (Z < 0.3)
    if args.dense:
        mask = np.ones_like(mask)
    ang = np.pi * (ri - 1.0)
    f0 = np.stack([np.cos(ang), np.sin(ang), 0.0 * ang], axis=-1).reshape(-1, 3)
    M = 1.25e-4 * np.eye(3)[None] + (9.5e-4 - 1.25e-4) * f0[:, :, None] * f0[:, None, :]
    del X, Y, Z, ro, ri, ang, f0
    nn = n + 1
    N = nn**3
    ctx = Context(0)
    HipOps.from_voxels(ctx, 3, (8, 8, 8), (h, h, h), np.eye(3), None, (9, 9, 9), 0, True, True)  # warm-up
    tic = time.perf_counter()
    phys = not args.slab
    ops = HipOps.from_voxels(ctx, 3, (n, n, n), (h, h, h), M, mask.ravel(), (nn, nn, nn), 0, phys, phys)
    ctx.synchronize()
    t_asm = time.perf_counter() - tic
    active_nodes = int((ops._mass_dev[0] > 0).sum())
    print(f"box {nn}^3 = {N/1e6:.1f} M nodes, {mask.mean()*100:.1f} % of the voxels active, {active_nodes/1e6:.2f} M tissue nodes; "
          f"upload + device assembly of the rows {t_asm:.2f} s", flush=True)
    del M
    ops.set_timestep(0.01, 0.5, 0.05)
    rng = torch.Generator(device=ctx.device)
    rng.manual_seed(1)
    v = ops.new_field()
    # smooth depolarised region + 1 % noise on the tissue nodes, 0 outside (as the split step leaves it)
    idx = torch.arange(N, device=ctx.device, dtype=torch.float64)
    xs, zs = (idx % nn) / nn, torch.div(idx, nn * nn, rounding_mode="floor") / nn
    bump = 100.0 * torch.exp(-((xs - 0.2) ** 2 + (zs - 0.4) ** 2) / 0.02)
    noise = 0.01 * torch.rand(N, generator=rng, device=ctx.device, dtype=torch.float64)
    v.data.copy_(torch.where(ops._mass_dev[0] > 0, -85.0 + bump + noise, torch.zeros_like(idx)))
    del idx, xs, zs, bump, noise
    ops.p.data.copy_(v.data)
    y = ops.new_field()

    def timeit(name, fn, bytes_per_node):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        med = ts[len(ts) // 2]
        print(f"{name:34s} {med:8.3f} ms  {bytes_per_node * N / med / 1e6:8.1f} GB/s algorithmic", flush=True)

    timeit("apply A (per-node rows)", lambda: ops.apply(0, v, y), 136)
    ops.st.zero_()
    timeit("spmv_dot (+reduce)", lambda: ops.spmv_dot(), 136)
    if args.slab:
        def two_parts():
            ops.spmv_interior(ops.p)
            ops.spmv_boundary(ops.p)

        timeit("spmv in two parts (+reduce)", two_parts, 136)
        # the decomposed solve itself: the rank is its own neighbour on both faces (mailbox transport), so the slab is periodic in z;
        # BEAT_VTL_PDOT_DIST=0 runs the three-kernel iteration with an exchange of p, the default the fused tile pass
        from beat._engine import DiffusionSolver, LibComm

        class Interior:
            rank, world, lo_phys, hi_phys, z0, z1, nz = 0, 1, False, False, 0, nn, nn

        comm = LibComm(ctx, Interior(), transport="ipc", peers=(0, 0), plane_doubles=nn * nn)
        solver = DiffusionSolver(ops, Interior(), force_distributed=True, libcomm=comm)
        x = ops.new_field()
        res = None

        def solve_dist():
            nonlocal res
            res = solver.solve(v, [], [], x, rtol=1e-8, atol=1e-50, max_it=500)

        timeit("decomposed theta-step solve rtol 1e-8", solve_dist, 0)
        print("iterations", res.iterations, "reason", res.converged_reason, "fused tile pass:", bool(ctx.lib.beat_pde_fused_dist_pass(ops.handle)))
        comm.close()
        return
    timeit("rhs (A and K rows, +reduce)", lambda: ops.rhs(v, [], [], v), 264)
    x = ops.new_field()
    res = None

    def solve():
        nonlocal res
        res = ops.solve_single(v, [], [], x, 1e-8, 1e-50, 500)

    timeit("theta-step solve rtol 1e-8", solve, 0)
    print("iterations", res.iterations, "reason", res.converged_reason)


if __name__ == "__main__":
    main()
