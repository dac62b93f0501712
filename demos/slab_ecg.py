#!/usr/bin/env python3
"""A slab of tissue with a pseudo-ECG: TP06 cells pre-paced to their limit cycle (``single_cell.get_steady_state``),
an S1 stimulus in one corner, and every millisecond the extracellular potential at two electrodes outside the slab
recovered from the transmembrane current (``beat.ecg.ECGRecovery``: one mass-matrix solve, one weighted integral per
electrode) -- the workflow of the reference's demos/slab.py, with a checkpoint of the last potential written through
``beat.io``.

    python demos/slab_ecg.py [--dx 0.5] [--T 40] [--beats 2] [--out results_slab]"""
import argparse
from pathlib import Path

import _path  # noqa: F401
import numpy as np

import beat
from beat import grid as g
from beat.models import tp06


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dx", type=float, default=0.5)
    ap.add_argument("--T", type=float, default=40.0)
    ap.add_argument("--dt", type=float, default=0.05)
    ap.add_argument("--beats", type=int, default=2, help="single-cell pre-pacing beats (the reference's demos use 200)")
    ap.add_argument("--out", default="results_slab")
    args = ap.parse_args(argv)
    out = Path(args.out)
    Lx, Ly, Lz = 10.0, 5.0, 2.0
    geo = beat.geometry.get_3D_slab_geometry(comm=g.COMM_WORLD, Lx=Lx, Ly=Ly, Lz=Lz, dx=args.dx)
    mesh = geo.mesh
    cond = beat.conductivities.default_conductivities("Niederer")
    M = beat.conductivities.define_conductivity_tensor(f0=geo.f0, **cond)
    C_m = (1.0 * beat.units.ureg("uF/cm**2")).to("uF/mm**2").magnitude
    time = g.Constant(mesh, 0.0)
    cells = g.locate_entities(mesh, 3, lambda x: (x[0] <= 1.5 + 1e-10) & (x[1] <= 1.5 + 1e-10))
    tags = g.meshtags(mesh, 3, cells, np.full(len(cells), 1, dtype=np.int32))
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=tags, marker=1,
                                           mesh_unit="mm", amplitude=50_000.0, duration=2.0)
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=I_s, C_m=C_m, dx=I_s.dZ)
    fun = tp06.generalized_rush_larsen
    y0 = beat.single_cell.get_steady_state(fun, tp06.init_state_values(), tp06.init_parameter_values(), outdir=out / "prepacing",
                                           nbeats=args.beats, BCL=1000, dt=0.05)
    ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(g.functionspace(mesh, ("Lagrange", 1))), v_pde=pde.state, fun=fun,
                                         init_states=y0, parameters=tp06.init_parameter_values(stim_amplitude=0.0),
                                         num_states=len(y0), v_index=tp06.state_index("V"))
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    ecg = beat.ecg.ECGRecovery(v=pde.state, sigma_b=1.0, C_m=C_m, M=M)
    electrodes = {"left": (-5.0, Ly / 2, Lz / 2), "right": (Lx + 5.0, Ly / 2, Lz / 2)}
    forms = {k: ecg.eval(p) for k, p in electrodes.items()}
    every = int(round(1.0 / args.dt))
    trace = {k: [] for k in forms}
    t, i = 0.0, 0
    while t < args.T - 1e-12:
        solver.step((t, t + args.dt))
        t += args.dt
        i += 1
        if i % every == 0:
            ecg.solve()
            for k, f in forms.items():
                trace[k].append(mesh.comm.allreduce(beat.ecg.assemble_scalar(f)))
    v = np.asarray(pde.state.x.array)
    lead = np.array(trace["right"]) - np.array(trace["left"])
    beat.io.write_mesh(out / "slab.bp", mesh)
    beat.io.write_function(out / "slab.bp", pde.state, time=t, name="v")
    if mesh.comm.rank == 0:
        print(f"{mesh.num_nodes} nodes, {i} steps of {args.dt} ms; v in [{v.min():.2f}, {v.max():.2f}] mV; "
              f"{100.0 * float((v > 0).mean()):.0f} % of the nodes depolarised")
        print("pseudo-ECG (right - left electrode), one value per ms:")
        print("  " + " ".join(f"{x:8.4f}" for x in lead))
    return lead, v


if __name__ == "__main__":
    main()
