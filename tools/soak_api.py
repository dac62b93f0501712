#!/usr/bin/env python3
"""A long run through the PUBLIC API with the step that leaves its solve open (round 5): a slab with a stimulated corner, TP06, thousands
of steps -- depolarisation front across the whole slab, plateau, repolarisation -- with the potential checked every few hundred
steps (which finishes the open solve) and the PCG record of every solve kept.  What it looks for: non-finite values, a solve that
does not converge, a hang; what it prints: time per step, iteration statistics per window, how often the launch behind the open
solve had to be repeated (the solve needed more iterations than were enqueued on spec).
    python tools/soak_api.py [--n 160] [--steps 6000] [--dt 0.05]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=160)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--dt", type=float, default=0.05)
    ap.add_argument("--window", type=int, default=500)
    args = ap.parse_args()
    import torch

    import beat
    from beat import grid as g
    from beat.models import tp06

    n, h = args.n, 0.2
    mesh = g.create_box(g.COMM_WORLD, [np.zeros(3), np.array([n * h, n * h / 2, n * h / 4])], [n, n // 2, n // 4])
    time_c = g.Constant(mesh, 0.0)
    cond = beat.conductivities.default_conductivities("Niederer")
    cells = g.locate_entities(mesh, 3, lambda x: (x[0] <= 1.5 + 1e-10) & (x[1] <= 1.5 + 1e-10) & (x[2] <= 1.5 + 1e-10))
    tags = g.meshtags(mesh, 3, cells, np.full(len(cells), 1, dtype=np.int32))
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time_c, subdomain_data=tags, marker=1, mesh_unit="mm",
                                           amplitude=50_000.0, duration=2.0)
    f0 = np.array([1.0, 0.0, 0.0])
    M = beat.conductivities.define_conductivity_tensor(f0=f0, **cond)
    pde = beat.MonodomainModel(time=time_c, mesh=mesh, M=M, I_s=I_s, C_m=0.01, dx=I_s.dZ, params={"petsc_options": {"ksp_rtol": 1e-8}})
    ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(g.functionspace(mesh, ("P", 1))), v_pde=pde.state,
                                         fun=tp06.generalized_rush_larsen, init_states=tp06.init_state_values(),
                                         parameters=tp06.init_parameter_values(stim_amplitude=0.0), num_states=19, v_index=tp06.state_index("V"))
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    ops = pde._ops
    print(f"{mesh.num_nodes / 1e6:.2f} M nodes, dt {args.dt} ms, {args.steps} steps; lazy: {pde.can_solve_lazily()}", flush=True)
    log = []
    ops.ksp_log = log
    t, opens = 0.0, 0
    tic = tw = time.perf_counter()
    for i in range(args.steps):
        solver.step((t, t + args.dt))
        opens += int(ops.open_x is not None)
        t = t + args.dt
        if (i + 1) % args.window == 0:
            v = np.asarray(pde.state.x.array)  # finishes the open solve, applies what is pending
            torch.cuda.synchronize()
            now = time.perf_counter()
            its = np.array([r.iterations for r in log[-args.window:]])
            jumps = int((np.diff(its) >= 2).sum())
            bad = [r for r in log[-args.window:] if r.converged_reason <= 0]
            print(f"  t = {t:7.1f} ms: v in [{v.min():7.2f}, {v.max():6.2f}], finite {bool(np.isfinite(v).all())}; its/step {its.mean():5.2f} "
                  f"(max {its.max()}), solves needing >= 2 more than their predecessor {jumps}, failed {len(bad)}; {(now - tw) / args.window * 1e3:.3f} ms/step",
                  flush=True)
            if not np.isfinite(v).all() or bad:
                raise SystemExit("soak FAILED")
            tw = now
    S = np.asarray(ode.values)
    print(f"done: {(time.perf_counter() - tic):.1f} s, steps that left their solve open {opens} of {args.steps}, all states finite "
          f"{bool(np.isfinite(S).all())}, {len(log)} KSP records", flush=True)
    if not np.isfinite(S).all() or len(log) != args.steps:
        raise SystemExit("soak FAILED")


if __name__ == "__main__":
    main()
