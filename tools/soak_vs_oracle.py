#!/usr/bin/env python3
"""A paced beat and the start of the next one of ToR-ORd cells with every parameter scaled by 0.9 .. 1.1 per cell
(tools/soak_cells.py's population, cell types cycled): the device kernel's in-kernel time loop against the NumPy oracle
stepped on the host, state by state at the end and along the potential.

    python3 tools/soak_vs_oracle.py [--cells 16] [--ms 1200] [--dt 0.02] [--land]
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
    ap.add_argument("--cells", type=int, default=16)
    ap.add_argument("--ms", type=float, default=1200.0)
    ap.add_argument("--dt", type=float, default=0.02)
    ap.add_argument("--land", action="store_true")
    ap.add_argument("--tp06", action="store_true")
    args = ap.parse_args()
    from beat.models import torord, torord_land

    from oracle import torord as otor

    m = torord_land if args.land else torord
    step = otor.torord_land_generalized_rush_larsen if args.land else otor.torord_generalized_rush_larsen
    if args.tp06:
        from beat.models import tp06

        from oracle import ionic

        m, step = tp06, ionic.tp06_generalized_rush_larsen
    rng = np.random.default_rng(2)
    n = args.cells
    P0 = m.init_parameter_values()
    P = np.repeat(P0[:, None], n, axis=1) * rng.uniform(0.9, 1.1, (len(P0), n))
    if args.tp06:
        fixed = ("stim_start", "stim_period", "stim_duration")
    else:
        P[m.parameter_index("celltype")] = np.arange(n) % 3
        fixed = ("i_Stim_Start", "i_Stim_End", "i_Stim_Period", "i_Stim_PulseDuration") + (("mode", "isacs") if args.land else ())
    for k in fixed:
        P[m.parameter_index(k)] = P0[m.parameter_index(k)]
    y0 = np.repeat(m.init_state_values()[:, None], n, axis=1)
    vi = m.state_index("V" if args.tp06 else "v")
    nsteps = int(round(args.ms / args.dt))
    every = int(round(1.0 / args.dt))
    y, tr = m.generalized_rush_larsen.run(y0, P, dt=args.dt, nsteps=nsteps, nbeats=1, track_indices=[vi], save_freq=every)
    yo = y0.copy()
    vo = []
    tic = time.perf_counter()
    for j in range(nsteps):
        if j % every == 0:
            vo.append(yo[vi].copy())
        yo = step(yo, j * args.dt, args.dt, P)
        if j % 10000 == 0:
            print(f"oracle step {j} of {nsteps} ({time.perf_counter() - tic:.0f} s)", flush=True)
    vo = np.array(vo)
    v = tr[:, 0]
    scale = np.maximum(np.abs(yo), 1e-6 * np.abs(y0) + 1e-12)
    err = np.abs(y - yo) / scale
    k = np.unravel_index(err.argmax(), err.shape)
    print(f"{nsteps} steps of {args.dt} ms, {n} cells: largest relative difference of a final state {err.max():.3e} "
          f"({m.state_names[k[0]] if hasattr(m, 'state_names') else k[0]}, cell {k[1]}); potential along the run: "
          f"max |dV| {np.abs(v - vo).max():.3e} mV; V in [{vo.min():.1f}, {vo.max():.1f}]")
    sys.exit(0 if err.max() < 1e-5 and np.abs(v - vo).max() < 1e-4 else 1)


if __name__ == "__main__":
    main()
