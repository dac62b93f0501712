#!/usr/bin/env python3
"""Where does the level of the ionic kernel come from (consecutive processes alternate between two levels ~5 % apart, rounds 3 - 6)?
The 19-row streaming probe (beat_stream_probe mode 4: the kernel's access pattern without its arithmetic) shows the same two levels,
so it is the memory system, not the kernel.  This script allocates the 512^3 TP06 state array several times IN ONE PROCESS -- through
torch's allocator, and directly with hipMalloc (beat_malloc) -- keeps all of them, and measures the probe on each: does the level
belong to the allocation (then a process can pick a good one) or to the process?

    python tools/place_probe3.py [--allocs 4] [--reps 5]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--allocs", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--n", type=int, default=512)
    args = ap.parse_args()
    import torch

    from beat import _hip
    from beat._device import Context, StateArray

    ctx = Context(0)
    lib = ctx.lib
    N, plane, S = args.n ** 3, args.n ** 2, 19

    def probe(ptr, ld):
        def once():
            _hip.check(lib.beat_stream_probe(ctx.handle, ptr, N, 4, 3, 1, 0, S, ld))
        once()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            once()
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        ms = ts[len(ts) // 2]
        return 2 * S * N * 8 / ms / 1e6  # GB/s

    import numpy as np

    from beat.models import tp06
    ic = tp06.init_state_values()
    prm = np.ascontiguousarray(tp06.init_parameter_values(stim_amplitude=0.0))
    vi = tp06.state_index("V")

    def kernel_ms(sa):
        g = torch.Generator(device="cuda")
        g.manual_seed(7)
        for k in range(S):
            if k == vi:
                sa.rows[k].copy_(torch.rand(N, generator=g, device="cuda", dtype=torch.float64) * 120.0 - 90.0)
            else:
                sa.rows[k].fill_(float(ic[k]))
        ts = []
        for _ in range(args.reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _hip.check(lib.beat_ode_step(ctx.handle, _hip.MODEL_TP06_GRL1, sa.ptr, N, sa.ld, prm.ctypes.data_as(C.c_void_p), len(prm), None, 0, 0.0, 0.01, vi, None))
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        ts = sorted(ts[1:])
        return ts[len(ts) // 2]

    keep = []
    for a in range(args.allocs):
        sa = StateArray(ctx, S, N, plane)
        sa.buf.fill_(1.25)
        keep.append(sa)
        rate = probe(sa.ptr, sa.ld)
        print(f"torch alloc {a}: ptr {sa.buf.data_ptr():#x}  rows pattern {rate:7.0f} GB/s   TP06 kernel {kernel_ms(sa):6.3f} ms", flush=True)
    ld = keep[0].ld
    nbytes = (plane + S * ld + 32) * 8
    raws = []
    for a in range(args.allocs):
        p = C.c_void_p()
        _hip.check(lib.beat_malloc(ctx.handle, C.c_size_t(nbytes), C.byref(p)))
        raws.append(p)
        base = C.c_void_p(p.value + 8 * plane)
        # (beat_malloc zero-fills; the probe multiplies by 1.0)
        rate = probe(base, ld)
        # the kernel on this allocation: the states of the first torch array copied over
        _hip.check(lib.beat_copy(ctx.handle, p, C.c_void_p(keep[0].buf.data_ptr()), plane + S * ld))
        ts = []
        for _ in range(args.reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _hip.check(lib.beat_ode_step(ctx.handle, _hip.MODEL_TP06_GRL1, base, N, ld, prm.ctypes.data_as(C.c_void_p), len(prm), None, 0, 0.0, 0.01, vi, None))
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts = sorted(ts[1:])
        print(f"hipMalloc   {a}: ptr {p.value:#x}  rows pattern {rate:7.0f} GB/s   TP06 kernel {ts[len(ts) // 2]:6.3f} ms", flush=True)
    # the first ones again (has anything changed while the others were allocated?)
    for a in range(min(2, args.allocs)):
        sa = keep[a]
        print(f"torch alloc {a} again: rows pattern {probe(sa.ptr, sa.ld):7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
