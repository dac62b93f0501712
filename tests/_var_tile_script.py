"""Per-node SpMV and a theta-step solve on a voxel-masked shell large enough for the tile-ordered segment list
(>= 4096 segments of 64 nodes; csrc/beat_pde_var.hip: var_range / var_walk): writes q = A p on the tissue nodes, p.q and
the solution of one solve to an .npz.  Run twice by tests/test_var_gpu.py -- BEAT_VAR_TILE=0 (node order) and the default
(tiles of 8 rows x 8 planes, every XCD walking one contiguous eighth) -- in fresh interpreters: the switch is read once per
process."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))
sys.path.insert(0, str(ROOT))

from beat import _stencil  # noqa: E402
from beat._device import Context  # noqa: E402
from beat._engine import HipOps  # noqa: E402


def main():
    out = sys.argv[1]
    rng = np.random.default_rng(21)
    cells, L = (110, 74, 66), (5.5, 3.7, 3.3)
    h = tuple(l / c for l, c in zip(L, cells))
    cc = np.stack(np.meshgrid(*(np.arange(c) + 0.5 for c in cells[::-1]), indexing="ij"), -1).reshape(-1, 3)[:, ::-1] * np.array(h)
    rr = np.sqrt((((cc - 0.5 * np.array(L)) / (0.5 * np.array(L))) ** 2).sum(axis=1))
    active = (rr < 0.97) & (rr > 0.55)
    ang = rng.uniform(0, np.pi, len(active))
    f0 = np.stack([np.cos(ang), np.sin(ang), 0 * ang], axis=-1)
    M = 1.2e-4 * np.eye(3)[None] + 8e-4 * f0[:, :, None] * f0[:, None, :]
    ctx = Context(0)
    nn = [c + 1 for c in cells]
    ops = HipOps.from_voxels(ctx, 3, cells, h, M, active, nn, 0, True, True)
    ops.set_guess_order(0)
    ops.set_timestep(0.01, 0.5, 0.05)
    n = ops.n
    x = rng.standard_normal(n)
    ops.ring[0].set(x)
    ops.q.fill(float("nan"))
    ops.st.zero_()
    ops.spmv_dot()
    ctx.synchronize()
    q = ops.q.numpy().copy()
    pq = float(ops.st[3])
    fv, fx = ops.new_field(), ops.new_field()
    fv.set(-85.0 + 60.0 * np.exp(-((np.arange(n) % nn[0]) * h[0] - 1.4) ** 2 / 0.08))
    res = ops.solve_single(fv, [], [], fx, 1e-10, 1e-50, 400)
    np.savez(out, q=q, pq=pq, x=fx.numpy(), its=res.iterations, nseg_nodes=int(np.isfinite(q).sum()))


if __name__ == "__main__":
    main()
