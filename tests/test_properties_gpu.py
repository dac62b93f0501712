"""GPU: size-independent properties of the hot path at the benchmark's full size (512^3 nodes, anisotropic fibre
tensor, BASELINE.json configs[3]) -- where no CPU oracle finishes in seconds:

  * the stencil operators are symmetric, the stiffness rows sum to zero (constants are in its kernel), the mass
    matrix integrates constants to the volume of the box;
  * a theta-step solved by the PCG satisfies its own stopping test when the residual is recomputed from scratch
    with the operator kernels (b - A x from beat_pde_apply, not the recurrence);
  * deferring the last update of the potential to the ionic kernel changes nothing;
  * one TP06 step keeps every gate in [0, 1] and leaves a resting cell at rest.
"""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 512
H = 0.1


def _ops(ctx):
    from beat import _stencil
    from beat._engine import HipOps

    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M = 9.5301e-4 * np.outer(f0, f0) + 1.2576e-4 * (np.eye(3) - np.outer(f0, f0))
    ops = HipOps(ctx, (N, N, N), True, True, *_stencil.stencil_tables(3, (H, H, H), M))
    ops.set_timestep(0.01, 0.5, 0.01)
    return ops


def _dot(ctx, a, b):
    from beat import _hip

    out = C.c_double()
    _hip.check(ctx.lib.beat_field_dot(ctx.handle, a.ptr, b.ptr, a.n, C.byref(out)))
    return out.value


def test_operator_properties_at_full_size(hip_ctx):
    import torch

    ctx = hip_ctx
    ops = _ops(ctx)
    n = N**3
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(7)
    x, y, ax, ay, ones = (ops.new_field() for _ in range(5))
    x.data.copy_(torch.randn(n, generator=gen, device=ctx.device, dtype=torch.float64))
    y.data.copy_(torch.randn(n, generator=gen, device=ctx.device, dtype=torch.float64))
    ones.fill(1.0)
    for which in (0, 2, 3):  # A, Mass, K
        ops.apply(which, x, ax)
        ops.apply(which, y, ay)
        lhs, rhs = _dot(ctx, y, ax), _dot(ctx, x, ay)
        scale = np.sqrt(_dot(ctx, ax, ax) * _dot(ctx, y, y))
        assert abs(lhs - rhs) <= 1e-12 * scale, (which, lhs, rhs)
    ops.apply(3, ones, ax)  # K 1 = 0
    kmax = max(abs(v) for v in ax.minmax())
    ops.apply(3, x, ay)
    assert kmax <= 1e-13 * max(abs(v) for v in ay.minmax())
    ops.apply(2, ones, ax)  # 1^T Mass 1 = |Omega|
    assert np.isclose(_dot(ctx, ones, ax), ((N - 1) * H) ** 3, rtol=1e-12)


def test_pcg_meets_its_stopping_test_at_full_size_and_deferral_is_exact(hip_ctx):
    import torch

    from beat import _hip

    ctx = hip_ctx
    ops = _ops(ctx)
    n = N**3
    idx = torch.arange(n, device=ctx.device, dtype=torch.float64)
    xs, ys, zs = idx % N, torch.div(idx, N, rounding_mode="floor") % N, torch.div(idx, N * N, rounding_mode="floor")
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(11)
    # a sharp depolarised blob (sigma 0.3 mm) plus 0.5 mV of noise: several PCG iterations at rtol 1e-8
    v0 = -85.0 + 100.0 * torch.exp(-(((xs - 200.0) ** 2 + (ys - 256.0) ** 2 + (zs - 300.0) ** 2) * H * H) / 0.18)
    v0 += torch.rand(n, generator=gen, device=ctx.device, dtype=torch.float64) - 0.5
    del idx, xs, ys, zs
    v, x1, x2, b, ax = (ops.new_field() for _ in range(5))
    v.data.copy_(v0)
    rtol = 1e-8
    res = ops.solve_single(v, [], [], x1, rtol, 1e-50, 200)
    assert res.converged_reason > 0 and 2 <= res.iterations <= 20
    ops.apply(1, v, b)      # b = B v (no stimulus)
    ops.apply(0, x1, ax)    # A x
    r = b.data - ax.data
    rnorm, bnorm = float(torch.linalg.vector_norm(r)), float(torch.linalg.vector_norm(b.data))
    assert np.isclose(bnorm, res.rhs_norm, rtol=1e-12)
    assert rnorm <= 1.05 * rtol * bnorm  # recurrence and true residual agree to a few percent at this tolerance
    # the same solve with the last update deferred, then applied by the flush pass: identical bits
    res2 = ops.solve_single(v, [], [], x2, rtol, 1e-50, 200, defer_flush=True)
    assert res2.iterations == res.iterations and ops.pending is not None
    assert not torch.equal(x1.data, x2.data)
    ops.flush_pending()
    assert torch.equal(x1.data, x2.data)


def test_tp06_step_invariants_at_full_size(hip_ctx):
    import torch

    from beat import _hip
    from beat._device import StateArray
    from beat.models import tp06

    ctx = hip_ctx
    n = N**3
    ic = tp06.init_state_values()
    P = np.ascontiguousarray(tp06.init_parameter_values(stim_amplitude=0.0))
    vi = tp06.state_index("V")
    sa = StateArray(ctx, 19, n, N * N)
    for k in range(19):
        sa.rows[k].fill_(float(ic[k]))
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(3)
    half = n // 2  # first half: resting cells; second half: random potentials between -95 and +45 mV
    sa.rows[vi][half:].copy_(-95.0 + 140.0 * torch.rand(n - half, generator=gen, device=ctx.device, dtype=torch.float64))
    for step in range(3):
        _hip.check(ctx.lib.beat_ode_step(ctx.handle, _hip.MODEL_TP06_GRL1, sa.ptr, n, sa.ld, P.ctypes.data_as(C.c_void_p),
                                         len(P), None, 0, step * 0.01, 0.01, vi, None))
    ctx.synchronize()
    assert bool(torch.isfinite(sa.rows).all())
    gates = ["Xr1", "Xr2", "Xs", "m", "h", "j", "d", "f", "f2", "fCass", "s", "r", "R_prime"]
    for g_name in gates:
        row = sa.rows[tp06.state_index(g_name)]
        assert float(row.min()) >= 0.0 and float(row.max()) <= 1.0 + 1e-12, g_name
    rest = sa.rows[vi][:half]
    assert float((rest - ic[vi]).abs().max()) < 0.05  # 0.03 ms of a resting cell: drift well below 0.05 mV
    assert float(sa.rows[tp06.state_index("Ca_i")].min()) > 0.0 and float(sa.rows[tp06.state_index("K_i")].min()) > 100.0


def test_per_node_operator_properties_on_a_large_voxel_shell(hip_ctx):
    """Per-node rows assembled on the device for a 257^3 box with an ellipsoidal shell of tissue and a rotating
    fibre field: A is symmetric (also through the SpMV that reads backward coefficients as the neighbours' forward
    ones), K annihilates constants on the tissue, Mass integrates 1 to the tissue volume, rows outside are identity."""
    import torch

    from beat._engine import HipOps

    ctx = hip_ctx
    n, h = 256, 0.25
    ax = (np.arange(n) + 0.5) / n - 0.5
    Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
    ro = np.sqrt((X / 0.48) ** 2 + (Y / 0.44) ** 2 + (Z / 0.48) ** 2)
    ri = np.sqrt((X / 0.30) ** 2 + (Y / 0.27) ** 2 + (Z / 0.30) ** 2)
    mask = (ro < 1.0) & (ri > 1.0) & (Z < 0.3)
    ang = np.pi * (ri - 1.0)
    f0 = np.stack([np.cos(ang), np.sin(ang), 0.0 * ang], axis=-1).reshape(-1, 3)
    M = 1.25e-4 * np.eye(3)[None] + (9.5e-4 - 1.25e-4) * f0[:, :, None] * f0[:, None, :]
    nn = n + 1
    N3 = nn**3
    ops = HipOps.from_voxels(ctx, 3, (n, n, n), (h, h, h), M, mask.ravel(), (nn, nn, nn), 0, True, True)
    ops.set_timestep(0.01, 0.5, 0.05)
    tissue = ops._mass_dev[0] > 0
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(5)
    x, y, ax_, ay_, ones = (ops.new_field() for _ in range(5))
    x.data.copy_(torch.randn(N3, generator=gen, device=ctx.device, dtype=torch.float64))
    y.data.copy_(torch.randn(N3, generator=gen, device=ctx.device, dtype=torch.float64))
    ones.fill(1.0)
    # symmetry through the dense apply
    ops.apply(0, x, ax_)
    ops.apply(0, y, ay_)
    lhs, rhs = _dot(ctx, y, ax_), _dot(ctx, x, ay_)
    assert abs(lhs - rhs) <= 1e-12 * np.sqrt(_dot(ctx, ax_, ax_) * _dot(ctx, y, y))
    # the solver's SpMV (segment list + symmetric coefficient reads) equals the dense apply on the tissue segments
    ops.ring[0].data.copy_(x.data)
    ops.st.zero_()
    ops.spmv_dot()
    ctx.synchronize()
    q = ops.q.data
    touched = q != 0  # segments without tissue are never written
    assert bool(touched[tissue].all())
    assert float((q - ax_.data)[touched].abs().max()) <= 1e-13 * float(ax_.data.abs().max())
    # identity rows outside, K 1 = 0, 1^T Mass 1 = tissue volume
    assert torch.equal(ax_.data[~tissue], x.data[~tissue])
    ops.apply(3, ones, ay_)
    ops.apply(3, x, ax_)
    assert float(ay_.data.abs().max()) <= 1e-13 * float(ax_.data.abs().max())
    ops.apply(2, ones, ax_)
    assert np.isclose(_dot(ctx, ones, ax_), mask.sum() * h**3, rtol=1e-12)
