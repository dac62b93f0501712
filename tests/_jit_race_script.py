"""Run three at a time by tests/test_api_gpu.py with ONE empty cache directory: every process needs the same kernel instance at the
same moment (as the ranks of a decomposed run do at their first step), compiles it under a name of its own and renames it into place;
each must end up with a loaded instance and the all-rows kernel's values."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd"), str(ROOT)]
import beat  # noqa: E402
from beat import _hip  # noqa: E402
from beat import grid as g  # noqa: E402
from beat.models import tp06  # noqa: E402

mesh = g.create_box(g.COMM_WORLD, [np.zeros(3), np.array([2.0, 1.0, 0.6])], [20, 10, 6])
V = g.functionspace(mesh, ("P", 1))
n = V.dofmap.index_map.size_local
xs = mesh.node_coordinates(pad3=True)
P = np.repeat(tp06.init_parameter_values(stim_amplitude=0.0)[:, None], n, axis=1)
P[tp06.parameter_index("g_to")] *= 1.0 - 0.3 * xs[:, 0] + 0.1 * xs[:, 2]
S0 = np.repeat(tp06.init_state_values()[:, None], n, axis=1)
S0[tp06.state_index("V")] = np.random.default_rng(4).uniform(-90.0, 30.0, n)
out = {}
for sparse in ("1", "0"):
    os.environ["BEAT_PARAM_SPARSE"] = sparse
    ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(V), v_pde=g.Function(V), fun=tp06.generalized_rush_larsen, init_states=S0,
                                         parameters=P, num_states=19, v_index=tp06.state_index("V"))
    for i in range(5):
        ode.step(0.02 * i, 0.02)
    out[sparse] = np.asarray(ode.values).copy()
np.testing.assert_allclose(out["1"], out["0"], rtol=1e-11, atol=1e-300)
stats = (C.c_longlong * 4)()
usable = _hip.load().beat_ode_jit_stats(stats)
assert usable == 1 and stats[0] == 1 and stats[3] == 0, list(stats)
print("jit-race ok", list(stats))
