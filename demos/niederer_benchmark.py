#!/usr/bin/env python3
"""Niederer et al. 2011 benchmark (20 x 7 x 3 mm slab, TP06, S1 stimulus in a 1.5 mm corner cube) -- the
set-up of the reference's demos/niederer_benchmark.py written against this package: only the imports differ
(beat.grid instead of dolfinx / ufl / scifem, beat.models.tp06 instead of the gotranx-generated module).

    python demos/niederer_benchmark.py [--dx 0.5] [--dt 0.05] [--T 70]

Prints the activation times (first time v > 0) at the nine probe points next to the table committed in the
reference demo for dx = 0.5 mm."""
import argparse
import time as wallclock

import _path  # noqa: F401
import numpy as np

import beat
from beat import grid as g
from beat.models import tp06

REFERENCE_TABLE = {  # (dx mm, dt ms) -> activation times at P1..P9 in ms (the table at the end of the reference's demo)
    (0.5, 0.05): (1.25, 51.1, 34.9, 58.9, 14.1, 49.5, 34.0, 56.65, 26.05),
    (0.5, 0.01): (1.22, 50.85, 33.96, 58.05, 13.98, 49.36, 33.07, 55.91, 25.64),
    (0.5, 0.005): (1.215, 50.775, 33.825, 57.96, 13.97, 49.345, 32.945, 55.825, 25.595),
    (0.2, 0.05): (1.25, 29.7, 32.9, 40.2, 9.55, 30.0, 32.95, 39.9, 18.9),
    (0.2, 0.01): (1.24, 29.09, 31.25, 38.66, 9.34, 29.4, 31.29, 38.42, 18.14),
    (0.2, 0.005): (1.235, 29.015, 31.05, 38.475, 9.315, 29.32, 31.08, 38.235, 18.045),
    (0.1, 0.05): (1.25, 26.85, 33.3, 40.35, 8.4, 27.5, 33.85, 40.55, 18.95),
    (0.1, 0.01): (1.23, 25.64, 31.46, 38.08, 8.03, 26.24, 31.94, 38.21, 17.95),
    (0.1, 0.005): (1.225, 25.5, 31.26, 37.81, 7.99, 26.09, 31.72, 37.93, 17.835),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dx", type=float, default=0.5)
    ap.add_argument("--dt", type=float, default=0.05)
    ap.add_argument("--T", type=float, default=70.0)
    ap.add_argument("--probe-every-step", action="store_true", help="read the nine probes back after every step, as the "
                    "reference's loop does (one host synchronisation per step), instead of recording them on the device")
    args = ap.parse_args()
    Lx, Ly, Lz = 20.0, 7.0, 3.0
    geo = beat.geometry.get_3D_slab_geometry(comm=g.COMM_WORLD, Lx=Lx, Ly=Ly, Lz=Lz, dx=args.dx)
    mesh = geo.mesh
    cond = beat.conductivities.default_conductivities("Niederer")
    C_m = (1.0 * beat.units.ureg("uF/cm**2")).to("uF/mm**2").magnitude
    time = g.Constant(mesh, 0.0)
    L, tol = 1.5, 1e-10
    cells = g.locate_entities(mesh, 3, lambda x: (x[0] <= L + tol) & (x[1] <= L + tol) & (x[2] <= L + tol))
    tags = g.meshtags(mesh, 3, cells, np.full(len(cells), 1, dtype=np.int32))
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=tags, marker=1,
                                           mesh_unit="mm", amplitude=50_000.0)
    M = beat.conductivities.define_conductivity_tensor(f0=geo.f0, **cond)
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=I_s, C_m=C_m, dx=I_s.dZ)
    ic = tp06.init_state_values(V=-85.23, Xr1=0.00621, Xr2=0.4712, Xs=0.0095, m=0.00172, h=0.7444, j=0.7045,
                                d=3.373e-05, f=0.7888, f2=0.9755, fCass=0.9953, s=0.999998, r=2.42e-08,
                                Ca_i=0.000126, R_prime=0.9073, Ca_SR=3.64, Ca_ss=0.00036, Na_i=8.604, K_i=136.89)
    ode = beat.odesolver.DolfinODESolver(
        v_ode=g.Function(g.functionspace(mesh, ("Lagrange", 1))), v_pde=pde.state, fun=tp06.generalized_rush_larsen,
        init_states=ic, parameters=tp06.init_parameter_values(stim_amplitude=0.0), num_states=len(ic),
        v_index=tp06.state_index("V"))
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    points = {"P1": (0, 0, 0), "P2": (0, Ly, 0), "P3": (Lx, 0, 0), "P4": (Lx, Ly, 0), "P5": (0, 0, Lz), "P6": (0, Ly, Lz),
              "P7": (Lx, 0, Lz), "P8": (Lx, Ly, Lz), "P9": (Lx / 2, Ly / 2, Lz / 2)}
    plist = np.array(list(points.values()), dtype=float)
    activation = {p: None for p in points}
    t, dt, nsteps = 0.0, args.dt, 0
    # The reference evaluates the probes after every step (scifem.evaluate_function, demos/niederer_benchmark.py:285-291).
    # Here the values are recorded on the device, one row per step, and looked at every 64 steps: the same values, the
    # same activation times (start of the first step after which v > 0), without a host round trip per step.
    rec = None if args.probe_every_step else g.ProbeRecorder(solver.pde.state, plist)
    seen, times = 0, []
    tic = wallclock.perf_counter()
    while t < args.T + 1e-12 and any(a is None for a in activation.values()):
        if rec is None:
            solver.step((t, t + dt))
            nsteps += 1
            times.append(t)
            t += dt
            rows = g.evaluate_function(solver.pde.state, plist).reshape(1, -1)
        else:
            # 64 steps per call: on this small grid MonodomainSplittingSolver.solve hands them to the library in one
            # piece (beat_split_steps), the recorder gets a row per step
            chunk = []
            while len(chunk) < 64 and t < args.T + 1e-12:
                chunk.append(t)
                t += dt
            solver.solve((chunk[0], chunk[-1] + dt), dt, recorder=rec)
            nsteps = len(rec)
            times = (times + chunk)[:nsteps]
            rows = rec.values()[seen:]
        for vals in rows:
            for p, value in zip(points, vals):
                if activation[p] is None and value > 0.0:
                    activation[p] = times[seen]
            seen += 1
    wall = wallclock.perf_counter() - tic
    if mesh.comm.rank != 0:  # several ranks (python -m torch.distributed.run --nproc-per-node N ...): one report
        return
    print(f"{mesh.num_nodes_global} nodes on {mesh.comm.size} rank(s), {nsteps} steps of {dt} ms in {wall:.2f} s "
          f"({wall / nsteps * 1e3:.2f} ms/step)")
    row = REFERENCE_TABLE.get((round(args.dx, 6), round(dt, 6)))
    ref = dict(zip(points, row)) if row else None
    for p in points:
        line = f"  {p}: {activation[p] if activation[p] is not None else float('nan'):8.2f} ms"
        if ref:
            line += f"   (reference table: {ref[p]:.2f})"
        print(line)


if __name__ == "__main__":
    main()
