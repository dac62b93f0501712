#!/usr/bin/env python3
"""How many iterations would a REDUCTION-FREE Chebyshev iteration need on the benchmark's diffusion solves, next to
Jacobi-PCG?  (Chebyshev removes both scalar collectives of a PCG iteration on a decomposed grid;
what it costs is iterations.)  Runs bench.py's 512^3 (or --size) TP06 problem for a few steps and, at each step, solves
the same theta-step system  A x = b  from x0 = v_  twice: with the product's PCG and with the Chebyshev semi-iteration
on D^-1 A with the spectrum bounds [lmax/5, lmax] the polynomial preconditioner already uses (lmax = Gershgorin bound;
kappa(D^-1 Mass) <= 5 for P1 tetrahedra).  The operator is applied by the library (beat_pde_apply); the vector updates
of the Chebyshev loop are torch expressions -- this script measures ITERATION COUNTS (and checks the residual with a
norm every iteration, which a production loop would do every few), not time."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--front", action="store_true", help="start from the developed planar front of bench.py")
    args = ap.parse_args()
    import torch

    import bench
    from beat import _hip, _stencil
    from beat._device import Context, StateArray
    from beat._engine import DiffusionSolver, HipOps, Slab, spectrum_bounds

    ctx = Context(0)
    n = args.size
    slab = Slab(n)
    mt, kt = _stencil.stencil_tables(3, (bench.H,) * 3, bench.conductivity())
    ops = HipOps(ctx, (n, n, n), True, True, mt, kt)
    ops.set_timestep(bench.C_M, bench.THETA, bench.DT)
    solver = DiffusionSolver(ops, slab)
    ic, params, vi = bench.tp06_defaults()
    states = StateArray(ctx, len(ic), n**3, n * n)
    if args.front:
        prof, _ = bench.developed_front_profile(ctx, n, ic, params, vi, args.rtol)
        for k in range(states.S):
            states.rows[k].view(-1, n).copy_(prof[k][None, :].expand(n * n, n))
    else:
        bench.init_states(ctx, states, ic, vi, n, slab, 1234, n)
    v = states.row_field(vi)
    A_tab = bench.C_M * mt + bench.THETA * bench.DT * kt
    lmin, lmax = spectrum_bounds(A_tab)
    dinv_tab = 1.0 / A_tab[:, 0]
    # per-node 1/diag from the node types (27 types): build once with the library: D^-1 = (A e_i)_i^-1 via apply on ones? use tables
    tx = np.ones(n, dtype=np.int64); tx[0] = 0; tx[-1] = 2
    types = (tx[None, None, :] + 3 * tx[None, :, None] + 9 * tx[:, None, None]).reshape(-1)
    dinv = ctx.from_numpy(dinv_tab[types])
    p_host = np.ascontiguousarray(params)
    b, ax, x, xv = ops.new_field(), ops.new_field(), ops.new_field(), ops.new_field()
    theta_c, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta_c / delta
    its_cg, its_ch = [], []
    t = 0.0
    for step in range(args.steps):
        _hip.check(ctx.lib.beat_ode_step(ctx.handle, _hip.MODEL_TP06_GRL1, states.ptr, n**3, states.ld,
                                         p_host.ctypes.data_as(C.c_void_p), len(p_host), None, 0, t, bench.DT, vi, None))
        ops.apply(1, v, b)  # b = B v_
        bnorm = float(torch.linalg.vector_norm(b.data))
        # Chebyshev semi-iteration on D^-1 A x = D^-1 b from x0 = v_
        xv.data.copy_(v.data)
        ops.apply(0, xv, ax)
        r = b.data - ax.data
        rho = 1.0 / sigma
        d = (dinv * r) / theta_c
        k = 0
        while float(torch.linalg.vector_norm(r)) > args.rtol * bnorm and k < 200:
            xv.data.add_(d)
            ops.apply(0, xv, ax)
            r = b.data - ax.data
            rho_new = 1.0 / (2.0 * sigma - rho)
            d = rho_new * rho * d + (2.0 * rho_new / delta) * (dinv * r)
            rho = rho_new
            k += 1
        its_ch.append(k)
        res = solver.solve(v, [], [], v, rtol=args.rtol, atol=1e-50, max_it=500)
        its_cg.append(res.iterations)
        assert float(torch.linalg.vector_norm(xv.data - v.data)) <= 1e-5 * float(torch.linalg.vector_norm(v.data))
        t += bench.DT
    print(f"{n}^3 {'front' if args.front else 'bump'}: PCG iterations {np.mean(its_cg):.2f}/step {its_cg}; "
          f"Chebyshev iterations {np.mean(its_ch):.2f}/step {its_ch}; spectrum bounds [{lmin:.3f}, {lmax:.3f}]")


if __name__ == "__main__":
    main()
