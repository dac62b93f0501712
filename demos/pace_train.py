#!/usr/bin/env python3
"""A pacing train on a slab whose cells are not all alike: the parameters of the ionic model are an array with one
column per node (the reference's demos/pace_train.py builds such a (P, N) array for its S1-S2 protocols), here a
gradient of the L-type calcium conductance along x, resident on the device (``DeviceParameters``).  Three S1 stimuli
200 ms apart in the corner; the demo reports when each wave reaches the far end and how long the action potential lasts
at both ends.

    python demos/pace_train.py [--dx 0.5] [--s1 3] [--bcl 200]"""
import argparse

import _path  # noqa: F401
import numpy as np

import beat
from beat import grid as g
from beat.models import tp06
from beat.models._base import DeviceParameters


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dx", type=float, default=0.5)
    ap.add_argument("--dt", type=float, default=0.05)
    ap.add_argument("--s1", type=int, default=3)
    ap.add_argument("--bcl", type=float, default=200.0)
    ap.add_argument("--block", action="store_true", help="the reference demo's own heterogeneity (demos/pace_train.py:133-167): "
                    "g_Kr and g_Ks set to zero in the right half of the cable instead of the g_CaL gradient -- piecewise "
                    "constant parameters, which run as two parameter classes at the uniform kernel's speed")
    args = ap.parse_args(argv)
    Lx, Ly, Lz = 12.0, 3.0, 1.0
    geo = beat.geometry.get_3D_slab_geometry(comm=g.COMM_WORLD, Lx=Lx, Ly=Ly, Lz=Lz, dx=args.dx)
    mesh = geo.mesh
    cond = beat.conductivities.default_conductivities("Niederer")
    M = beat.conductivities.define_conductivity_tensor(f0=geo.f0, **cond)
    C_m = (1.0 * beat.units.ureg("uF/cm**2")).to("uF/mm**2").magnitude
    time = g.Constant(mesh, 0.0)
    cells = g.locate_entities(mesh, 3, lambda x: x[0] <= 1.0 + 1e-10)
    tags = g.meshtags(mesh, 3, cells, np.full(len(cells), 1, dtype=np.int32))
    # S1 train: 2 ms of current every BCL -- one Stimulus per pulse, all on the same cells
    train = [beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=tags, marker=1, mesh_unit="mm",
                                              amplitude=50_000.0, duration=2.0, start=k * args.bcl) for k in range(args.s1)]
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=train, C_m=C_m, dx=train[0].dZ)
    V = g.functionspace(mesh, ("Lagrange", 1))
    x = V.tabulate_dof_coordinates()[:, 0]
    P = np.repeat(tp06.init_parameter_values(stim_amplitude=0.0)[:, None], len(x), axis=1)
    if args.block:
        for name in ("g_Kr", "g_Ks"):
            P[tp06.parameter_index(name)] = np.where(x >= Lx / 2, 0.0, P[tp06.parameter_index(name)])
    else:
        P[tp06.parameter_index("g_CaL")] *= 1.0 - 0.5 * x / Lx      # action potentials shorten towards the far end
    params = DeviceParameters(P)
    y0 = tp06.init_state_values()
    ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(V), v_pde=pde.state, fun=tp06.generalized_rush_larsen, init_states=y0,
                                         parameters=params, num_states=len(y0), v_index=tp06.state_index("V"))
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    probes = np.array([[0.5, Ly / 2, Lz / 2], [Lx - 0.5, Ly / 2, Lz / 2]])
    rec = g.ProbeRecorder(pde.state, probes)
    T = args.s1 * args.bcl
    solver.solve((0.0, T), dt=args.dt, recorder=rec)
    tr = rec.values()                                              # (steps, 2)
    t = args.dt * (1 + np.arange(len(tr)))
    route = "parameter classes: %d uniform sets" % ode._dev.classes[2] if ode._dev.classes is not None else "per-node parameter rows"
    what = "g_Kr = g_Ks = 0 for x >= %g mm" % (Lx / 2) if args.block else f"g_CaL x 1.0 at x = 0 .. x 0.5 at x = {Lx} mm"
    print(f"{mesh.num_nodes} nodes, {len(tr)} steps of {args.dt} ms, {what} ({route})")
    report = []
    for b in range(args.s1):
        win = (t >= b * args.bcl) & (t < (b + 1) * args.bcl)
        row = []
        for p, name in enumerate(("near", "far")):
            v = tr[win, p]
            tt = t[win]
            if v.max() < 0.0:
                row.append((np.nan, np.nan))
                continue
            up = tt[np.argmax(v > 0.0)]
            level = v.min() + 0.1 * (v.max() - v.min())
            after = np.nonzero((tt > up) & (v < level))[0]
            apd = (tt[after[0]] - up) if len(after) else np.nan
            row.append((up - b * args.bcl, apd))
        report.append(row)
        print(f"  S1 #{b + 1}: activation near / far end {row[0][0]:6.2f} / {row[1][0]:6.2f} ms after the stimulus, "
              f"APD90 {row[0][1]:6.1f} / {row[1][1]:6.1f} ms")
    return report


if __name__ == "__main__":
    main()
