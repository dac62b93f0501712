#!/usr/bin/env python3
"""Diffusion alone: ``MonodomainModel.solve`` without an ionic model (what the reference's demos/diffusion.py does on a
20 x 20 unit square): a constant source on the lower-left corner patch, theta-rule steps of 0.1, and the closed form the
run has to reproduce -- with no-flux boundaries the integral of v grows by (source x patch area) per unit time, whatever
the conductivity does to its shape (up to the tolerance of the linear solves).

    python demos/diffusion.py [--n 20] [--T 2.5] [--dt 0.1]"""
import argparse

import _path  # noqa: F401
import numpy as np

import beat
from beat import grid as g


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20)
    ap.add_argument("--T", type=float, default=2.5)
    ap.add_argument("--dt", type=float, default=0.1)
    ap.add_argument("--patch", type=float, default=0.3, help="side of the stimulated corner patch")
    args = ap.parse_args(argv)
    mesh = g.create_unit_square(g.COMM_WORLD, args.n, args.n, g.CellType.triangle)
    corner = g.locate_entities(mesh, mesh.topology.dim, lambda x: (x[0] <= args.patch + 1e-10) & (x[1] <= args.patch + 1e-10))
    tags = g.meshtags(mesh, mesh.topology.dim, corner, np.full(len(corner), 1, dtype=np.int32))
    dx = g.Measure("dx", domain=mesh, subdomain_data=tags)
    source = g.Constant(mesh, 1.0)
    time = g.Constant(mesh, g.default_scalar_type(0.0))
    model = beat.MonodomainModel(time=time, mesh=mesh, M=1.0, I_s=beat.base_model.Stimulus(expr=source, dZ=dx, marker=1), dx=dx)
    res = model.solve((0.0, args.T), dt=args.dt)
    v = np.asarray(res.state.x.array)
    # integral of v over the square (P1 on a uniform triangulation: nodal weights h^2, halved on edges, quartered at corners)
    w1 = np.full(args.n + 1, 1.0 / args.n)
    w1[[0, -1]] *= 0.5
    total = float((np.outer(w1, w1).ravel() * v).sum())
    cells_per_side = int(np.floor(args.patch * args.n + 1e-9))
    area = (cells_per_side / args.n) ** 2
    nsteps = int(round(args.T / args.dt))
    print(f"{mesh.num_nodes} nodes, {nsteps} steps of {args.dt}: status {res.status.name}, v in [{v.min():.5f}, {v.max():.5f}]")
    print(f"integral of v = {total:.10f}; source x patch area x T = {area * nsteps * args.dt:.10f}")
    return total, area * nsteps * args.dt


if __name__ == "__main__":
    main()
