#!/usr/bin/env python3
"""Where does the HOST spend its time in MonodomainSplittingSolver.step?  A 64^3 grid (multi-launch diffusion path, GPU time per step
well under the host's), 3000 steps under cProfile: the table is the Python side of the per-step critical path."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd"), str(ROOT)]
import beat  # noqa: E402
from beat import grid as g  # noqa: E402
from beat.models import tp06  # noqa: E402

n1 = 64
mesh = g.create_box(g.COMM_WORLD, [np.zeros(3), np.full(3, (n1 - 1) * 0.1)], [n1 - 1] * 3)
pde = beat.MonodomainModel(time=g.Constant(mesh, 0.0), mesh=mesh, M=9.5301e-4 * np.eye(3), C_m=0.01,
                           params={"theta": 0.5, "petsc_options": {"ksp_rtol": 1e-8}})
ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(g.functionspace(mesh, ("P", 1))), v_pde=pde.state, fun=tp06.generalized_rush_larsen,
                                     init_states=tp06.init_state_values(), parameters=tp06.init_parameter_values(stim_amplitude=0.0),
                                     num_states=19, v_index=tp06.state_index("V"))
solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
dt = 0.01
for i in range(50):
    solver.step((i * dt, (i + 1) * dt))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
tic = time.perf_counter()
for i in range(50, 50 + steps):
    solver.step((i * dt, (i + 1) * dt))
wall = time.perf_counter() - tic
print(f"{wall / steps * 1e6:.1f} us per step without the profiler")
pr = cProfile.Profile()
pr.enable()
for i in range(50 + steps, 50 + 2 * steps):
    solver.step((i * dt, (i + 1) * dt))
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
