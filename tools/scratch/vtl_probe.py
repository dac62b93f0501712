"""Bounded probe of the tile SpMV: one configuration per process, so a hang costs one timeout."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

def main():
    case = sys.argv[1]
    from beat import _stencil
    from beat._device import Context
    from beat._engine import HipOps
    import test_var_gpu as T
    ctx = Context(0)
    if case.startswith("slab"):
        cells, L = (30, 12, 9), (3.0, 1.2, 0.9)
        z0, z1 = 3, 8
    else:
        cells, L = (22, 17, 13), (2.2, 1.7, 1.3)
        z0, z1 = 0, 14
    h = tuple(l / c for l, c in zip(L, cells))
    active, M = T._shell_case(cells, L, 4)
    nx, ny, nz = (c + 1 for c in cells)
    mf, kf = _stencil.stencil_fields(3, cells, h, M, active, z_range=(z0, z1))
    ops = HipOps(ctx, (nx, ny, z1 - z0), z0 == 0, z1 == nz, mf, kf, per_node=True)
    ops.set_timestep(0.01, 0.5, 0.05)
    rng = np.random.default_rng(1)
    ops.ring[0].set(rng.standard_normal(ops.n))
    ops.st.zero_()
    for k in range(3):
        ops.spmv_dot()
        ctx.synchronize()
        print(case, "launch", k, "pq", float(ops.st[3]), flush=True)

main()
