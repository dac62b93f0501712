#!/usr/bin/env python3
"""FitzHugh-Nagumo on the unit square (the README example of the reference, 32 x 32 cells): stimulus in the lower
left quadrant for 0.5 ms, forward-Euler ionic step on the device, Godunov splitting.

    python demos/fitzhughnagumo.py [--n 32] [--T 10]"""
import argparse

import _path  # noqa: F401
import numpy as np

import beat
from beat import grid as g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--T", type=float, default=10.0)
    ap.add_argument("--dt", type=float, default=0.01)
    args = ap.parse_args()
    mesh = g.create_unit_square(g.COMM_WORLD, args.n, args.n, g.CellType.triangle)
    time = g.Constant(mesh, g.default_scalar_type(0.0))
    a, b, c1, c2, c3, v_peak, v_rest = 0.13, 0.013, 0.26, 0.1, 1.0, 40.0, -85.0
    # (c_1, c_2, c_3, a, b, v_amp, v_rest, v_peak, stim amplitude, stim duration, stim start): the model's own
    # stimulus is switched off, the PDE carries it
    parameters = np.array([c1, c2, c3, a, b, v_peak - v_rest, v_rest, v_peak, 0.0, 1.0, 0.0])
    stim = g.conditional(g.And(g.ge(time, 0.0), g.le(time, 0.5)), 600.0, 0.0)
    cells = g.locate_entities(mesh, mesh.topology.dim, lambda x: (x[0] <= 0.5) & (x[1] <= 0.5))
    tags = g.meshtags(mesh, mesh.topology.dim, cells, np.full(len(cells), 1, dtype=np.int32))
    dx = g.Measure("dx", domain=mesh, subdomain_data=tags)
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=0.001, I_s=beat.Stimulus(expr=stim, dZ=dx, marker=1), dx=dx)
    ode = beat.odesolver.DolfinODESolver(
        v_ode=g.Function(g.functionspace(mesh, ("P", 1))), v_pde=pde.state, fun=beat.models.fhn.forward_euler_readme,
        init_states=np.array([0.0, v_rest]), parameters=parameters, num_states=2, v_index=1)
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    t, dt, i = 0.0, args.dt, 0
    while t < args.T + 1e-12:
        if i % 100 == 0:
            v = solver.pde.state.x.array
            print(f"t = {t:6.2f} ms: v in [{v.min():8.3f}, {v.max():8.3f}] mV")
        solver.step((t, t + dt))
        t += dt
        i += 1


if __name__ == "__main__":
    main()
