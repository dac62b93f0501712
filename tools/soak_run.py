"""Long run of the bench workload (default 256^3): steps the split scheme for thousands of steps, checks the state
array for non-finite values every 100 steps and, if one appears, prints the node, the step and the node's states before
and after the ionic kernel.  This is how the V = 15 mV singularity of the TP06 L-type current was found.

    python tools/soak_run.py [n] [first_checked_step] [steps] [defer|nodefer] [guess order, default 3]
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "fenicsx-beat_amd")]
import bench  # noqa: E402
import torch  # noqa: E402
from beat import _hip, _stencil  # noqa: E402
from beat._device import Context, StateArray  # noqa: E402
from beat._engine import DiffusionSolver, HipOps, Slab  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
start_check = int(sys.argv[2]) if len(sys.argv) > 2 else 2400
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3100
defer = (sys.argv[4] != "nodefer") if len(sys.argv) > 4 else True
guess_order = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ctx = Context(0)
lib = ctx.lib
slab = Slab(n, 0, 1)
plane = n * n
N = plane * n
ops = HipOps(ctx, (n, n, n), True, True, *_stencil.stencil_tables(3, (bench.H,) * 3, bench.conductivity()))
ops.set_preconditioner(1)
ops.set_guess_order(guess_order)
ops.set_timestep(bench.C_M, bench.THETA, bench.DT)
solver = DiffusionSolver(ops, slab)
ic, params, v_index = bench.tp06_defaults()
states = StateArray(ctx, len(ic), N, plane)
bench.init_states(ctx, states, ic, v_index, n, slab, 1234, n)
v_field = states.row_field(v_index)
p_host = np.ascontiguousarray(params)
p_ptr = p_host.ctypes.data_as(C.c_void_p)
t = 0.0
prev = None
all_its = []
for step in range(nsteps):
    check = step >= start_check and step % 100 == 0
    if step % 1000 == 0 and not check:
        print(step, "(not checked yet)", flush=True)  # heartbeat: a silent run looks hung to the job runner
    if check:
        ops.flush_pending()
        prev = [r.clone() for r in states.rows]
    pend = ops.pending
    ops.pending = None
    _hip.check(lib.beat_ode_step_pending(ctx.handle, _hip.MODEL_TP06_GRL1, states.ptr, N, states.ld, p_ptr, len(p_host), None, 0,
                                         t, bench.DT, v_index, None, ops.handle, ops.ring[0].ptr, ops.fld, pend[2] if pend else 0))
    if check:
        bad = [k for k, r in enumerate(states.rows) if not bool(torch.isfinite(r).all())]
        if bad:
            k = bad[0]
            idx = int(torch.nonzero(~torch.isfinite(states.rows[k]))[0])
            print("step", step, "non-finite after the ionic step in states", bad, "node", idx, (idx % n, idx // n % n, idx // plane))
            print("before:", [float(r[idx]) for r in prev])
            print("after: ", [float(r[idx]) for r in states.rows])
            break
    try:
        res = solver.solve(v_field, [], [], v_field, rtol=1e-8, atol=1e-50, max_it=500, defer_flush=defer)
    except Exception as exc:
        print("step", step, "solve failed:", exc)
        bad = [k for k, r in enumerate(states.rows) if not bool(torch.isfinite(r).all())]
        print("non-finite rows before the solve:", bad)
        break
    all_its.append(res.iterations)
    if check and step % 1000 == 0:
        ops.flush_pending()
        print(step, res.iterations, float(v_field.data.min()), float(v_field.data.max()), flush=True)
    t += bench.DT
ops.flush_pending()
its = np.array(all_its)
print(f"guess order {guess_order}: {len(its)} steps, PCG iterations mean {its.mean():.2f} max {its.max()}, per 1000 steps "
      + " ".join(f"{its[k:k + 1000].mean():.2f}" for k in range(0, len(its), 1000)))
print("final v in [%.6f, %.6f], sum %.9e" % (float(v_field.data.min()), float(v_field.data.max()), float(v_field.data.sum())))
