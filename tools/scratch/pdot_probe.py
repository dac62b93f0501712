import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd"), str(ROOT), str(ROOT / "tests")]
from beat import _stencil
from beat._device import Context
from beat._engine import HipOps
import test_var_gpu as T
ctx = Context(0)
cells, L = (70, 9, 40), (7.0, 0.9, 4.0)
h = tuple(l / c for l, c in zip(L, cells))
active, M = T._shell_case(cells, L, 11)
nn = tuple(c + 1 for c in cells)
mf, kf = _stencil.stencil_fields(3, cells, h, M, active)
rng = np.random.default_rng(5)
n = int(np.prod(nn))
x = rng.standard_normal(n)
tissue = mf[0] != 0.0
res = {}
for mode in ("0", "1"):
    os.environ.update(BEAT_VTL="1", BEAT_VTL_RY="8", BEAT_VTL_RUN="16", BEAT_VTL_PDOT=mode, BEAT_VRR="0")
    for k in (5, 6, 7, 8, 12, 13, 30):
        ops = HipOps(ctx, nn, True, True, mf, kf, per_node=True)
        ops.set_timestep(0.01, 0.5, 0.05)
        fv, fx = ops.new_field(), ops.new_field()
        fv.set(np.where(tissue, -80.0 + 20.0 * x, 0.0))
        try:
            ops.solve_single(fv, [], [], fx, 1e-30, 1e-300, k)
        except Exception as e:
            pass
        ctx.synchronize()
        res[(mode, k)] = (fx.numpy().copy(), ops.ring[0].numpy().copy(), ops.ring[1].numpy().copy(), ops.q.numpy().copy(), ops.r.numpy().copy() if hasattr(ops, "r") else None)
for k in (5, 6, 7, 8, 12, 13, 30):
    a, b = res[("0", k)], res[("1", k)]
    for name, u, v in (("x", a[0], b[0]), ("ring0", a[1], b[1]), ("ring1", a[2], b[2]), ("q", a[3], b[3])):
        d = np.nonzero(u != v)[0]
        d = d[tissue[d]]
        msg = ""
        if len(d):
            i = d[0]
            ix, iy, iz = i % nn[0], (i // nn[0]) % nn[1], i // (nn[0] * nn[1])
            msg = f" first at node {i} = ({ix},{iy},{iz}): {u[i]!r} vs {v[i]!r}; x-positions of differing nodes: {sorted(set((d % nn[0]).tolist()))[:20]}"
        print(f"max_it {k}: {name}: {len(d)} tissue nodes differ{msg}")
