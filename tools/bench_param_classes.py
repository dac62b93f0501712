#!/usr/bin/env python3
"""TP06 (or --model torord) ionic step at 256^3 with (i) uniform parameters, (ii) the reference's pace_train heterogeneity -- a (P, N) array in
which g_Kr and g_Ks are zero in half of the domain (demos/pace_train.py:133-167), recognised as two parameter classes --,
(iii) the same array forced through the per-node kernel (BEAT_PARAM_CLASSES=0), (iv) a smooth per-node field in ONE parameter: the varying row alone on the device
(beat_ode_step_rows) -- on the kernel instance compiled for that index at first use (csrc/beat_ode_jit.h) and on the run-time-index kernel
(BEAT_JIT=0) -- and, for comparison, all rows (BEAT_PARAM_SPARSE=0).  HIP events around the kernel, median of --reps launches.
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
    ap.add_argument("--model", default="tp06", choices=["tp06", "torord"], help="torord: ToR-ORd-dynCl, 45 states, 112 parameters")
    args = ap.parse_args()
    import torch

    from beat._device import Context
    from beat.models import torord
    from beat.models import tp06 as tp06_module
    from beat.models._base import DeviceParameters
    from beat.odesolver import _DeviceODE
    from beat.telemetry import NullMonitor

    ctx = Context(0)
    n = args.n
    N = n**3
    tor = args.model == "torord"
    tp06 = torord if tor else tp06_module  # (the model module the rows below address)
    P0 = tp06.init_parameter_values() if tor else tp06.init_parameter_values(stim_amplitude=0.0)
    ic = tp06.init_state_values()
    vi = tp06.state_index("v" if tor else "V")
    blocked, smooth_name = (("GKr_b", "GKs_b"), "PCa_b") if tor else (("g_Kr", "g_Ks"), "g_CaL")
    x = (torch.arange(N, device=ctx.device) % n).to(torch.float64) / n

    def per_node(kind):
        t = torch.from_numpy(P0).to(ctx.device)[:, None].repeat(1, N)
        if kind == "block":
            for name in blocked:
                t[tp06.parameter_index(name)] = torch.where(x >= 0.5, torch.zeros_like(x), t[tp06.parameter_index(name)])
        elif kind == "smooth8":  # eight smooth rows (round 5: compiled instances take up to 16 varying rows)
            names8 = (("GKr_b", "GKs_b", "GNa", "Gto_b", "GK1_b", "PCa_b", "GNaL_b", "Gncx_b") if tor
                      else ("g_CaL", "g_Kr", "g_Ks", "g_Na", "g_to", "g_K1", "g_bca", "g_pCa"))
            for k, name in enumerate(names8):
                t[tp06.parameter_index(name)] *= 1.0 - (0.1 + 0.04 * k) * x
        else:
            t[tp06.parameter_index(smooth_name)] *= 1.0 - 0.5 * x
        dp = DeviceParameters.__new__(DeviceParameters)
        dp.ctx, dp.version, dp._dev = ctx, 1, t
        return dp

    def run(label, params, env=None):
        os.environ.pop("BEAT_PARAM_CLASSES", None)
        os.environ.pop("BEAT_PARAM_SPARSE", None)
        os.environ.pop("BEAT_JIT", None)
        if env:
            os.environ.update(env)
        dev = _DeviceODE(ctx, tp06.generalized_rush_larsen, len(ic), N, n * n, params, NullMonitor())
        for k in range(len(ic)):
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
    c = run(f"{blocked[0]} = {blocked[1]} = 0 in half of the domain (P, N)", blk)
    p = run("  the same through the per-node kernel", blk, {"BEAT_PARAM_CLASSES": "0"})
    del blk
    sm = per_node("smooth")
    run(f"smooth {smooth_name} gradient: compiling", sm)  # (the first call compiles the instance: not a timing)
    gj = run(f"smooth {smooth_name} gradient (P, N): sparse rows, compiled instance", sm)
    g = run("  the same with the run-time-index kernel (BEAT_JIT=0)", sm, {"BEAT_JIT": "0"})
    gd = run(f"  the same with all {len(P0)} rows on the device", sm, {"BEAT_PARAM_SPARSE": "0"})
    print(f"classes / uniform = {c / u:.3f}; per-node / uniform = {p / u:.3f}; smooth field: compiled instance {gj / u:.3f}, run-time index {g / u:.3f}, all rows {gd / u:.3f}")
    del sm
    sm8 = per_node("smooth8")
    run("eight smooth rows: compiling", sm8)
    g8 = run("eight smooth rows (P, N): sparse rows, compiled instance", sm8)
    g8d = run(f"  the same with all {len(P0)} rows on the device", sm8, {"BEAT_PARAM_SPARSE": "0"})
    print(f"eight smooth rows: compiled instance {g8 / u:.3f} x the uniform kernel, all rows {g8d / u:.3f}")


if __name__ == "__main__":
    main()
