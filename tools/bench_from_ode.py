#!/usr/bin/env python3
"""A generated cell model's kernel on its own: ms per step and TB/s of its algorithmic bytes (16 B per state and node), with the
library's table-driven exp (the default of beat.models.from_ode) and with libm's.    python tools/bench_from_ode.py [file.ode] [--n 16777216]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ode", nargs="?", default=str(ROOT / "tests" / "data" / "small_cell.ode"))
    ap.add_argument("--n", type=int, default=256**3)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch

    from beat import telemetry
    from beat._device import Context
    from beat.models import from_ode
    from beat.odesolver import _DeviceODE

    ctx = Context.default()
    for fast in (True, False, True, False):
        model = from_ode(args.ode, fast_exp=fast)
        p = model.init_parameter_values()
        ode = _DeviceODE(ctx, model, model.num_states, args.n, 0, p, telemetry.NullMonitor())
        ode.set_initial(model.init_state_values())
        rows = ode.states.rows
        if model.v_name:
            rows[model.state_index(model.v_name)] += 40.0 * torch.rand(args.n, dtype=torch.float64, device=rows.device)
        t = 0.0
        for _ in range(3):
            ode.step(t, 0.01)
            t += 0.01
        torch.cuda.synchronize()
        tic = time.perf_counter()
        for _ in range(args.steps):
            ode.step(t, 0.01)
            t += 0.01
        torch.cuda.synchronize()
        ms = (time.perf_counter() - tic) / args.steps * 1e3
        byts = 16.0 * model.num_states * args.n
        print(f"{Path(args.ode).name}: {model.num_states} states, {args.n} nodes, exp = {'table-driven (FastMath)' if fast else 'libm'}: "
              f"{ms:.3f} ms/step = {byts / ms / 1e9:.2f} TB/s of {16 * model.num_states} B/node, finite {bool(torch.isfinite(rows).all())}", flush=True)
        del ode


if __name__ == "__main__":
    main()
