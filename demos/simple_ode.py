#!/usr/bin/env python3
"""Free-running cell models, no tissue: ``beat.odesolver.solve`` advances many independent cells and records the
potential of each (the reference's demos/simple_ode.py pattern, there with a gotranx-generated function).  Here the
cells are ten Tusscher-Panfilov cells whose slow delayed-rectifier conductance g_Ks varies from cell to cell -- per-node
parameters, one kernel launch per step for all of them -- and the run reports each cell's action-potential duration.

    python demos/simple_ode.py [--cells 64] [--T 450] [--dt 0.05]"""
import argparse

import _path  # noqa: F401
import numpy as np

import beat
from beat.models import tp06


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=64)
    ap.add_argument("--T", type=float, default=450.0)
    ap.add_argument("--dt", type=float, default=0.05)
    args = ap.parse_args(argv)
    n = args.cells
    states = np.repeat(tp06.init_state_values()[:, None], n, axis=1)
    # the model's own stimulus (52 uA/uF for 1 ms) fires at t = 10 ms in every cell
    parameters = np.repeat(tp06.init_parameter_values(stim_start=10.0, stim_period=1000.0)[:, None], n, axis=1)
    scale = np.linspace(0.5, 2.0, n)
    parameters[tp06.parameter_index("g_Ks")] *= scale
    nsteps = int(np.ceil(args.T / args.dt)) - 1
    V = np.zeros((nsteps, n))
    vi = tp06.state_index("V")
    beat.odesolver.solve(fun=tp06.generalized_rush_larsen, t_bound=args.T, states=states, V=V, V_index=vi, dt=args.dt,
                         parameters=parameters)
    t = args.dt * (1 + np.arange(nsteps))
    rest, peak = V[0], V.max(axis=0)
    level = rest + 0.1 * (peak - rest)                       # 90 % repolarisation
    up = np.array([t[np.argmax(V[:, c] > 0.0)] for c in range(n)])
    down = np.array([t[np.nonzero(V[:, c] > level[c])[0][-1]] for c in range(n)])
    apd90 = down - up
    print(f"{n} TP06 cells, {nsteps} steps of {args.dt} ms; g_Ks scaled by {scale[0]:.2f} .. {scale[-1]:.2f}")
    for c in sorted(set([0, n // 4, n // 2, 3 * n // 4, n - 1])):
        print(f"  cell {c:4d}: g_Ks x {scale[c]:.2f}  peak {peak[c]:6.2f} mV  APD90 {apd90[c]:7.2f} ms")
    return scale, apd90


if __name__ == "__main__":
    main()
