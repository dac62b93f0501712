#!/usr/bin/env python3
"""TP06 ionic step at 256^3 with (i) uniform parameters, (ii) the reference's pace_train heterogeneity -- a (P, N) array in
which g_Kr and g_Ks are zero in half of the domain (demos/pace_train.py:133-167), recognised as two parameter classes --,
(iii) the same array forced through the per-node kernel (BEAT_PARAM_CLASSES=0), (iv) a smooth per-node field in ONE parameter: the varying row alone on the device
(beat_ode_step_rows) and, for comparison, all 53 rows (BEAT_PARAM_SPARSE=0).  HIP events around the kernel, median of --reps launches.
    python tools/bench_param_classes.py [--n 256]"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--reps", type=int, default=8)
    args = ap.parse_args()
    import torch

    from beat._device import Context
    from beat.models import tp06
    from beat.models._base import DeviceParameters
    from beat.odesolver import _DeviceODE
    from beat.telemetry import NullMonitor

    ctx = Context(0)
    n = args.n
    N = n**3
    P0 = tp06.init_parameter_values(stim_amplitude=0.0)
    ic = tp06.init_state_values()
    vi = tp06.state_index("V")
    x = (torch.arange(N, device=ctx.device) % n).to(torch.float64) / n

    def per_node(kind):
        t = torch.from_numpy(P0).to(ctx.device)[:, None].repeat(1, N)
        if kind == "block":
            for name in ("g_Kr", "g_Ks"):
                t[tp06.parameter_index(name)] = torch.where(x >= 0.5, torch.zeros_like(x), t[tp06.parameter_index(name)])
        else:
            t[tp06.parameter_index("g_CaL")] *= 1.0 - 0.5 * x
        dp = DeviceParameters.__new__(DeviceParameters)
        dp.ctx, dp.version, dp._dev = ctx, 1, t
        return dp

    def run(label, params, env=None):
        os.environ.pop("BEAT_PARAM_CLASSES", None)
        os.environ.pop("BEAT_PARAM_SPARSE", None)
        if env:
            os.environ.update(env)
        dev = _DeviceODE(ctx, tp06.generalized_rush_larsen, 19, N, n * n, params, NullMonitor())
        for k in range(19):
            dev.states.rows[k].fill_(float(ic[k]))
        dev.states.rows[vi].add_(torch.rand(N, dtype=torch.float64, device=ctx.device) * 100.0)
        for _ in range(2):
            dev.step(0.0, 0.01, v_index=vi)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in ev:
            a.record()
            dev.step(0.0, 0.01, v_index=vi)
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
        route = (f"{dev.classes[2]} classes" if dev.classes is not None else f"{len(dev._sparse[1])} varying row(s) + uniform vector"
                 if dev._sparse is not None else ("all per-node rows" if getattr(params, "ndim", 1) == 2 else "uniform"))
        print(f"{label:52s} {ms:7.3f} ms   ({route})", flush=True)
        del dev
        torch.cuda.empty_cache()
        return ms

    u = run("uniform parameters", P0)
    blk = per_node("block")
    c = run("g_Kr = g_Ks = 0 in half of the domain (P, N)", blk)
    p = run("  the same through the per-node kernel", blk, {"BEAT_PARAM_CLASSES": "0"})
    del blk
    sm = per_node("smooth")
    g = run("smooth g_CaL gradient (P, N): sparse rows", sm)
    gd = run("  the same with all 53 rows on the device", sm, {"BEAT_PARAM_SPARSE": "0"})
    print(f"classes / uniform = {c / u:.3f}; per-node / uniform = {p / u:.3f}; smooth field: sparse rows {g / u:.3f}, all rows {gd / u:.3f}")


if __name__ == "__main__":
    main()
