#!/usr/bin/env python3
"""Per-kernel timing on one GPU (HIP events on the launch stream): python tools/bench_kernels.py [--n 512]
Prints ms and algorithmic GB/s for each hot kernel of the split step."""
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
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch

    from beat import _hip, _stencil
    from beat._device import Context, StateArray
    from beat._engine import HipOps
    from beat.models import tp06
    import bench

    ctx = Context(0)
    lib = ctx.lib
    n = args.n
    plane, N = n * n, n**3
    mt, kt = _stencil.stencil_tables(3, (0.1,) * 3, bench.conductivity())
    ops = HipOps(ctx, (n, n, n), True, True, mt, kt)
    ops.set_timestep(0.01, 0.5, 0.01)
    x, y = ops.new_field(), ops.new_field()
    x.data.copy_(torch.randn(N, dtype=torch.float64, device=ctx.device))
    ops.p.data.copy_(x.data)
    ops.r.data.copy_(torch.randn(N, dtype=torch.float64, device=ctx.device))
    st = ops.st
    stp = C.c_void_p(st.data_ptr())

    def timeit(name, fn, bytes_per_node, reps=args.reps):
        if args.only and args.only not in name:
            return
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        med = ts[len(ts) // 2]
        print(f"{name:28s} {med:9.3f} ms   {bytes_per_node * N / med / 1e6:8.1f} GB/s algorithmic   (min {ts[0]:.3f})", flush=True)

    def reset_st():
        st.zero_()
        st[1] = 1.0
        st[3] = 2.0
        st[7] = 0.5

    reset_st()
    timeit("apply A (stencil)", lambda: _hip.check(lib.beat_pde_apply(ops.handle, 0, x.ptr, y.ptr)), 16)
    timeit("spmv_dot (+reduce)", lambda: _hip.check(lib.beat_pde_spmv_dot(ops.handle, ops.p.ptr, ops.q.ptr, stp)), 16)
    empty_p = (C.c_void_p * 1)()
    empty_a = (C.c_double * 1)()
    timeit("rhs (+reduce)", lambda: _hip.check(lib.beat_pde_rhs(ops.handle, x.ptr, empty_p, empty_a, 0, x.ptr, ops.r.ptr, ops.p.ptr, stp)), 16)
    reset_st()
    timeit("cg_update (+reduce)", lambda: (reset_st(), _hip.check(lib.beat_pde_cg_update(ops.handle, stp, y.ptr, ops.r.ptr, ops.p.ptr, ops.q.ptr)))[1], 48)
    reset_st()
    timeit("cg_next (pupdate)", lambda: (reset_st(), _hip.check(lib.beat_pde_cg_next(ops.handle, stp, ops.r.ptr, ops.p.ptr)))[1], 24)
    timeit("copy", lambda: _hip.check(lib.beat_copy(ctx.handle, y.ptr, x.ptr, N)), 16)

    ic = tp06.init_state_values()
    P = np.ascontiguousarray(tp06.init_parameter_values(stim_amplitude=0.0))
    sa = StateArray(ctx, 19, N, plane)
    for k in range(19):
        sa.rows[k].fill_(float(ic[k]))
    sa.rows[17].add_(torch.rand(N, dtype=torch.float64, device=ctx.device) * 100.0)
    snap = sa.rows.clone() if N <= 300**3 else None

    def ode():
        _hip.check(lib.beat_ode_step(ctx.handle, _hip.MODEL_TP06_GRL1, sa.ptr, N, sa.ld, P.ctypes.data_as(C.c_void_p), 53,
                                     None, 0, 0.0, 0.01, 17, None))

    timeit("ode_step tp06", ode, 304, reps=6)
    del sa, snap

    from beat.models import torord

    ic = torord.init_state_values()
    P2 = np.ascontiguousarray(torord.init_parameter_values())
    vi = torord.state_index("v")
    sb = StateArray(ctx, len(ic), N, plane)
    for k in range(len(ic)):
        sb.rows[k].fill_(float(ic[k]))
    sb.rows[vi].add_(torch.rand(N, dtype=torch.float64, device=ctx.device) * 100.0)

    def ode2():
        _hip.check(lib.beat_ode_step(ctx.handle, _hip.MODEL_TORORD_DYNCL_GRL1, sb.ptr, N, sb.ld, P2.ctypes.data_as(C.c_void_p),
                                     len(P2), None, 0, 0.0, 0.01, vi, None))

    timeit("ode_step torord", ode2, 16 * len(ic), reps=4)
    del sb

    from beat.models import torord_land

    ic = torord_land.init_state_values()
    P3 = np.ascontiguousarray(torord_land.init_parameter_values())
    vi3 = torord_land.state_index("v")
    sc = StateArray(ctx, len(ic), N, plane)
    for k in range(len(ic)):
        sc.rows[k].fill_(float(ic[k]))
    sc.rows[vi3].add_(torch.rand(N, dtype=torch.float64, device=ctx.device) * 100.0)
    sc.rows[torord_land.state_index("CaTrpn")].fill_(0.05)

    def ode3():
        _hip.check(lib.beat_ode_step(ctx.handle, _hip.MODEL_TORORD_LAND_GRL1, sc.ptr, N, sc.ld, P3.ctypes.data_as(C.c_void_p),
                                     len(P3), None, 0, 0.0, 0.01, vi3, None))

    timeit("ode_step torord_land", ode3, 16 * len(ic), reps=4)


if __name__ == "__main__":
    main()
