#!/usr/bin/env python3
"""Does the spread of the headline's ionic kernel (9.7 ... 10.1 ms between bench PROCESSES on one box) follow the ALLOCATION?
One process, the bench's direct path (beat_ode_step_pending + DiffusionSolver.solve at 512^3), several trials: before each the
operator, its work / guess fields and the state array are released (torch's caching allocator emptied) and allocated again, behind
a dummy allocation of a different size, so that the arrays land somewhere else.  Prints ode_ms / pde_ms per trial and the device
addresses.  Same addresses, same time / other addresses, other time => placement; no pattern => not placement.

    python3 tools/place_probe2.py [--trials 6] [--steps 20]
"""
import argparse
import ctypes as C
import gc
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd"), str(ROOT)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=6)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--vary", default="all", choices=["all", "ops", "states"], help="which of the two is released and allocated again between "
                    "trials: the operator with its work / guess fields, the state array, or both")
    args = ap.parse_args()
    import torch

    import bench
    from beat import _hip, _stencil
    from beat._device import Context, StateArray
    from beat._engine import DiffusionSolver, HipOps, Slab

    ctx = Context(0)
    n = args.n
    plane = n * n
    ic, params, v_index = bench.tp06_defaults()
    p_host = np.ascontiguousarray(params)
    p_ptr = p_host.ctypes.data_as(C.c_void_p)
    slab = Slab(n, 0, 1)
    mass_tab, stiff_tab = _stencil.stencil_tables(3, (bench.H,) * 3, bench.conductivity())
    dummy_gib = [0, 1, 3, 0, 7, 2, 5, 0]
    ops = states = solver = None
    for trial in range(args.trials):
        gib = dummy_gib[trial % len(dummy_gib)]
        dummy = torch.empty(gib * (1 << 27) + 4096 * trial, dtype=torch.float64, device=ctx.device) if gib else None
        if ops is None or args.vary in ("all", "ops"):
            del solver, ops
            gc.collect()
            torch.cuda.empty_cache()
            ops = HipOps(ctx, (n, n, n), True, True, mass_tab, stiff_tab)
            ops.set_guess_order(-1)
            ops.set_timestep(bench.C_M, bench.THETA, bench.DT)
            solver = DiffusionSolver(ops, slab)
        else:
            ops.flush_pending()
            ops.guess_reset()
        if states is None or args.vary in ("all", "states"):
            del states
            gc.collect()
            torch.cuda.empty_cache()
            states = StateArray(ctx, len(ic), plane * n, plane)
        bench.init_states(ctx, states, ic, v_index, n, slab, 1234, n)
        v_field = states.row_field(v_index)
        t = 0.0
        ode, pde = [], []
        for i in range(5 + args.steps):
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            pend = ops.pending
            ops.pending = None
            a.record()
            _hip.check(ctx.lib.beat_ode_step_pending(ctx.handle, _hip.MODEL_TP06_GRL1, states.ptr, plane * n, states.ld, p_ptr, len(p_host),
                                                     None, 0, t, bench.DT, v_index, None, ops.handle, ops.ring[0].ptr, ops.fld,
                                                     pend[2] if pend else 0))
            b.record()
            solver.solve(v_field, [], [], v_field, rtol=1e-8, atol=1e-50, max_it=500, defer_flush=True)
            c.record()
            torch.cuda.synchronize()
            if i >= 5:
                ode.append(a.elapsed_time(b))
                pde.append(b.elapsed_time(c))
            t += bench.DT
        ops.flush_pending()
        torch.cuda.synchronize()
        print(f"trial {trial}: dummy {gib} GiB  states @ {states.ptr.value:#x}  work @ {ops.ring[0].ptr.value:#x}  "
              f"ode {np.mean(ode):.3f} ms (min {np.min(ode):.3f} max {np.max(ode):.3f})  pde {np.mean(pde):.3f} ms", flush=True)
        del v_field, dummy
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
