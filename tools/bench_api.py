#!/usr/bin/env python3
"""The benchmark's workload driven through the PUBLIC API (the reference's classes: MonodomainModel, DolfinODESolver,
MonodomainSplittingSolver.step) instead of bench.py's direct C-ABI calls: same grid, tensor, cell model, dt, tolerance and
initial state; prints ms/step for both regimes of bench.py's headline (bump) so that the cost of the Python layer at
benchmark size is a measured number.

    python tools/bench_api.py [--size 512] [--steps 20] [--warmup 5]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "fenicsx-beat_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    args = ap.parse_args()
    import torch

    import beat
    import bench
    from beat import grid as g
    from beat.models import tp06

    n = args.size
    tic = time.perf_counter()
    L = (n - 1) * bench.H
    mesh = g.create_box(g.COMM_WORLD, [np.zeros(3), np.full(3, L)], [n - 1] * 3)
    time_c = g.Constant(mesh, 0.0)
    pde = beat.MonodomainModel(time=time_c, mesh=mesh, M=bench.conductivity(), C_m=bench.C_M,
                               params={"theta": bench.THETA, "petsc_options": {"ksp_rtol": 1e-8}})
    ic, params, vi = bench.tp06_defaults()
    V = g.functionspace(mesh, ("P", 1))
    ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(V), v_pde=pde.state, fun=tp06.generalized_rush_larsen, init_states=ic,
                                         parameters=params, num_states=len(ic), v_index=vi)
    # the benchmark's initial state, written into the solver's device array
    bench.init_states(pde._ctx, ode._dev.states, ic, vi, n, mesh.slab, 1234, n)
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    torch.cuda.synchronize()
    print(f"set-up through the public API: {time.perf_counter() - tic:.1f} s", flush=True)
    t = 0.0
    for _ in range(args.warmup):
        solver.step((t, t + bench.DT))
        t += bench.DT
    torch.cuda.synchronize()
    tic = time.perf_counter()
    its = []
    for _ in range(args.steps):
        solver.step((t, t + bench.DT))
        its.append(pde.ksp.iterations)
        t += bench.DT
    pde.state.field  # applies a pending update of the potential, as bench.py's timed region does
    torch.cuda.synchronize()
    wall = time.perf_counter() - tic
    vmin, vmax = pde.state.field.minmax()
    print(f"public API: {wall / args.steps * 1e3:.3f} ms/step, {n**3 * args.steps / wall / 1e9:.3f} G node-updates/s, "
          f"PCG {np.mean(its):.2f} its/step, v in [{vmin:.2f}, {vmax:.2f}] mV")


if __name__ == "__main__":
    main()
