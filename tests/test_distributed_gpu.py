"""GPU: the collective code path of beat._engine.DiffusionSolver (stage kernels driven from Python,
RCCL all-reduces on slices of the device-side solver state, split SpMV) on a ONE-rank NCCL group --
what can be exercised of the N > 1 path on a single-GPU box -- against the fused single-slab solve."""

import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("per_node", [False, True])
def test_collective_path_on_one_rank_matches_single_slab_solve(hip_ctx, per_node):
    import torch.distributed as dist

    from beat import _stencil
    from beat._engine import DiffusionSolver, HipOps, Slab

    ctx = hip_ctx
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=ctx.device)
        created = True
    try:
        nx, ny, nz = 40, 33, 17
        f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
        M = 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
        if per_node:  # masked grid + per-node rows (beat_pde_create_var)
            cc = np.stack(np.meshgrid(np.arange(nz - 1), np.arange(ny - 1), np.arange(nx - 1), indexing="ij"), -1).reshape(-1, 3)
            active = ((cc - np.array([8, 16, 20])) ** 2).sum(axis=1) < 14**2
            mt, kt = _stencil.stencil_fields(3, (nx - 1, ny - 1, nz - 1), (0.1, 0.1, 0.1), M, active)
        else:
            mt, kt = _stencil.stencil_tables(3, (0.1, 0.1, 0.1), M)
        rng = np.random.default_rng(3)
        v = -85.0 + 30.0 * rng.random(nx * ny * nz)
        results = []
        for force in (False, True):
            ops = HipOps(ctx, (nx, ny, nz), True, True, mt, kt, per_node=per_node)
            ops.set_timestep(0.01, 0.5, 0.05)
            solver = DiffusionSolver(ops, Slab(nz), force_distributed=force)
            fv, fx = ops.new_field(), ops.new_field()
            fv.set(v)
            res = solver.solve(fv, [], [], fx, rtol=1e-11, atol=1e-50, max_it=200)
            results.append((fx.numpy(), res))
        (x1, r1), (x2, r2) = results
        assert r2.converged_reason > 0 and abs(r1.iterations - r2.iterations) <= 1
        np.testing.assert_allclose(x2, x1, rtol=0, atol=1e-9 * np.abs(x1).max())
        # deferred last update through the collective path: pending until flushed, then the same bits
        fx3 = ops.new_field()
        res3 = solver.solve(fv, [], [], fx3, rtol=1e-11, atol=1e-50, max_it=200, defer_flush=True)
        assert res3.iterations == r2.iterations
        if res3.iterations % 6:
            assert ops.pending is not None and not np.array_equal(fx3.numpy(), x2)
        ops.flush_pending()
        assert ops.pending is None
        np.testing.assert_array_equal(fx3.numpy(), x2)
    finally:
        if created:
            dist.destroy_process_group()
