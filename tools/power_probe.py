#!/usr/bin/env python3
"""Power / clock probe: launches one kernel of the split step back to back for a few seconds while a thread polls
`rocm-smi` (socket power, shader clock), to tell an issue-bound kernel from a power-capped one.
usage: python tools/power_probe.py [--what tp06|copy|rr] [--seconds 8] [--n 512]"""
import argparse
import ctypes as C
import json
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "fenicsx-beat_amd"))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout)
            out.append((time.time(), d))
        except Exception as e:  # noqa: BLE001
            out.append((time.time(), {"error": str(e)}))
        time.sleep(0.05)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="tp06")
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--n", type=int, default=512)
    args = ap.parse_args()
    import torch

    from beat import _hip
    from beat._device import Context, StateArray
    from beat.models import tp06

    ctx = Context(0)
    lib = ctx.lib
    n = args.n
    N = n**3
    if args.what == "tp06":
        ic = tp06.init_state_values()
        P = np.ascontiguousarray(tp06.init_parameter_values(stim_amplitude=0.0))
        sa = StateArray(ctx, 19, N, n * n)
        for k in range(19):
            sa.rows[k].fill_(float(ic[k]))
        sa.rows[17].add_(torch.rand(N, dtype=torch.float64, device=ctx.device) * 100.0)
        snap = sa.rows[17].clone()

        def fn():
            _hip.check(lib.beat_ode_step(ctx.handle, _hip.MODEL_TP06_GRL1, sa.ptr, N, sa.ld, P.ctypes.data_as(C.c_void_p), 53,
                                         None, 0, 0.0, 0.01, 17, None))
    else:
        a = torch.empty(19 * N, dtype=torch.float64, device=ctx.device).normal_()
        b = torch.empty_like(a)

        def fn():
            b.copy_(a)

    samples, stop = [], threading.Event()
    th = threading.Thread(target=poll, args=(stop, samples))
    idle = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout
    print("idle:", idle.strip()[:600])
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th.start()
    t0 = time.time()
    reps, times = 0, []
    while time.time() - t0 < args.seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 20)
        reps += 20
    stop.set()
    th.join()
    print(f"{args.what}: {reps} launches, per-launch ms first {times[0]:.3f} median {sorted(times)[len(times)//2]:.3f} last {times[-1]:.3f}")
    for ts, d in samples:
        if "error" in d:
            print("  err", d["error"])
            continue
        for card, v in d.items():
            keys = {k: v[k] for k in v if any(s in k.lower() for s in ("power", "sclk", "mclk", "temperature (sensor junction)", "fclk"))}
            print(f"  t={ts - t0:6.2f}s {card}: {keys}")


if __name__ == "__main__":
    main()
