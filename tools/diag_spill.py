#!/usr/bin/env python3
"""Diagnosis of a heavily spilled generated kernel giving wrong values (tests/data/big_cell.ode with BEAT_ODE_EMIT=global, per-node
parameters): which states, which nodes, deterministic or not, and which compiler flags make it go away (BEAT_JIT_EXTRA_FLAGS).
    python tools/diag_spill.py"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%r, "fenicsx-beat_amd"))
from beat.models import from_ode
model = from_ode(os.path.join(%r, "tests", "data", "big_cell.ode"))
rng = np.random.default_rng(17)
n = 8000
y = np.repeat(model.init_state_values()[:, None], n, axis=1)
y[0] = rng.uniform(-95.0, 40.0, n)
for k, name in enumerate(model.state_names):
    if name.startswith("x"):
        y[k] = rng.uniform(0.0, 1.0, n)
    elif name.startswith("c_"):
        y[k] *= rng.uniform(0.7, 1.4, n)
p = model.init_parameter_values(stim_amplitude=30.0)
pn = np.repeat(p[:, None], n, axis=1)
pn[model.parameter_index("g_3")] *= rng.uniform(0.5, 1.5, n)
pn[model.parameter_index("k_5")] *= rng.uniform(0.5, 1.5, n)
for tag, prm in (("uniform", p), ("per-node", pn), ("per-node again", pn), ("per-node, all columns equal", np.repeat(p[:, None], n, axis=1))):
    dev = model(states=y, t=1.4, parameters=prm, dt=0.02)
    ref = model.numpy_step(y, 1.4, prm, 0.02)
    err = np.abs(dev - ref) / np.maximum(np.maximum(np.abs(ref), np.abs(y)), 1e-12)
    bad = err > 1e-9
    rows = np.nonzero(bad.any(axis=1))[0]
    cols = np.nonzero(bad.any(axis=0))[0]
    print("   %%-28s max rel err %%.3e; wrong: %%d states %%s, %%d of %%d nodes, first nodes %%s, lanes (mod 64) %%s" %% (
        tag, err.max(), len(rows), [model.state_names[k] for k in rows[:6]], len(cols), n, cols[:6].tolist(), sorted(set((cols %% 64).tolist()))[:10]), flush=True)
''' % (str(ROOT), str(ROOT))


def main():
    for emit in ("global", "by_state"):
        for flags in ("", "-mllvm -amdgpu-spill-vgpr-to-agpr=0", "-O1", "-mllvm -amdgpu-use-divergent-register-indexing"):
            env = dict(os.environ, BEAT_JIT_EXTRA_FLAGS=flags)
            if emit == "global":
                env["BEAT_ODE_EMIT"] = "global"
            else:
                env.pop("BEAT_ODE_EMIT", None)
            print(f"== emit {emit}, extra flags [{flags}]", flush=True)
            r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
            print(r.stdout.rstrip() or r.stderr[-1500:], flush=True)
            if emit == "by_state":
                break


if __name__ == "__main__":
    main()
