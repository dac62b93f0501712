"""Is the two-valued time of the TP06 ionic kernel at 512^3 (9.7 or 10.2 ms, process to process) a property of WHERE the state
array lies?  One process: allocate the array several times (keeping or freeing the previous ones), time the kernel on each."""
import ctypes as C, sys, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd")]
import torch
from beat import _hip
from beat._device import Context, StateArray
from beat.models import tp06
ctx = Context(0)
n1 = 512
N, plane = n1**3, n1 * n1
ic = tp06.init_state_values()
P = tp06.init_parameter_values(stim_amplitude=0.0)
vi = tp06.state_index("V")

def time_on(sa, reps=6):
    for k in range(19):
        sa.rows[k].fill_(float(ic[k]))
    sa.rows[vi].add_(torch.rand(N, dtype=torch.float64, device=ctx.device) * 60.0)
    ts = []
    for r in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _hip.check(ctx.lib.beat_ode_step(ctx.handle, _hip.MODEL_TP06_GRL1, sa.ptr, N, sa.ld, P.ctypes.data_as(C.c_void_p), 53, None, 0, 0.0, 0.01, vi, None))
        b.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]

keep = []
mode = sys.argv[1] if len(sys.argv) > 1 else "keep"
for trial in range(6):
    sa = StateArray(ctx, 19, N, plane)
    t = time_on(sa)
    print(f"trial {trial}: data_ptr {sa.buf.data_ptr():#x}  (mod 2 MiB {sa.buf.data_ptr() % (1<<21):#x}, mod 1 GiB {sa.buf.data_ptr() % (1<<30):#x})  ld {sa.ld}  kernel {t:.3f} ms", flush=True)
    if mode == "keep":
        keep.append(sa)  # the next array lands elsewhere
    else:
        del sa
        torch.cuda.empty_cache()
