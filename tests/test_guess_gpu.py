"""Extrapolated initial guess of the diffusion solve (beat_pde_set_guess_order, include/beat_hip.h).

The reference leaves PETSc's initial guess off (src/beat/base_model.py:141-151: no ksp_initial_guess_nonzero, no
KSPGuess), so the solution of A x = b it returns does not depend on where the iteration starts -- which is what is
checked here: with the guess on, every solve still ends at ||b - A x|| <= rtol ||b||, the values agree with those of
the x0 = v_ iteration within that tolerance and with a sparse direct solve, the guess never costs a pass of its own
(it rides with the deferred update), and all it changes is the iteration count.
"""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["one-launch", "multi-launch"])
def small_grid_path(request):
    """Grids of a few thousand nodes are solved by one workgroup in one launch (csrc/beat_pde_small.hip) unless the
    operator is told otherwise: every test of this module runs on both paths."""
    from beat._engine import HipOps

    old = HipOps.default_small
    HipOps.default_small = request.param == "one-launch"
    yield request.param
    HipOps.default_small = old


def _system(cells, C_m=0.01, theta=0.5, dt=0.05):
    from oracle import fem

    mesh = fem.BoxMesh(cells, tuple(0.1 * c for c in cells))
    M = np.array([[2.0, 0.3, 0.0], [0.3, 1.0, 0.1], [0.0, 0.1, 0.5]]) * 1e-3
    Mass, K = fem.assemble_mass(mesh), fem.assemble_stiffness(mesh, M)
    A = (C_m * Mass + theta * dt * K).tocsc()
    B = (C_m * Mass - (1 - theta) * dt * K).tocsr()
    return mesh, M, A, B


def _ops(ctx, cells, M, order, C_m=0.01, theta=0.5, dt=0.05, active=None):
    from beat import _stencil
    from beat._engine import HipOps

    nn = tuple(c + 1 for c in cells)
    if active is not None:  # per-node operator rows on a voxel-masked domain
        ops = HipOps(ctx, nn, True, True, *_stencil.stencil_fields(3, cells, (0.1,) * 3, M, active), per_node=True)
    else:
        ops = HipOps(ctx, nn, True, True, *_stencil.stencil_tables(3, (0.1,) * 3, M))
    ops.set_guess_order(order)
    ops.set_timestep(C_m, theta, dt)
    return ops


def _history(ops):
    h0, h1, cnt = C.c_void_p(), C.c_void_p(), C.c_int()
    from beat import _hip

    _hip.check(ops.lib.beat_pde_guess_history(ops.handle, C.byref(h0), C.byref(h1), C.byref(cnt)))
    return h0.value, h1.value, cnt.value


def _moving_bump(mesh, t):
    x = mesh.x
    c = np.array([0.4 + 0.6 * t, 0.5, 0.3])
    return -85.0 + 100.0 * np.exp(-((x - c) ** 2).sum(axis=1) / (2 * 0.25**2))


@pytest.mark.parametrize("defer", [False, True])
@pytest.mark.parametrize("rtol,few", [(1e-8, True), (1e-13, False)])
def test_guess_changes_the_iteration_count_not_the_solution(hip_ctx, defer, rtol, few):
    """A bump that moves a little every step (what a depolarisation front does to the right-hand side).  Orders 0 to 4
    on the same sequence of right-hand sides: each solution equals the sparse direct solve within the tolerance the
    stopping test promises, the recorded increment is x - v_, and the iteration totals fall with the order.  With
    rtol = 1e-13 the solves take more iterations than the ring of search directions holds, so the update of x happens
    in several cycles (the first carries the guess, the later ones add to the recorded increment)."""
    cells = (24, 20, 12)
    mesh, M, A, B = _system(cells)
    lu = spla.splu(A)
    n = mesh.num_nodes
    totals = {}
    for order in (0, 1, 2, 3, 4, "auto"):
        ops = _ops(hip_ctx, cells, M, order)
        fv, fx = ops.new_field(), ops.new_field()
        its = []
        for step in range(10):
            v = _moving_bump(mesh, 0.02 * step)
            fv.set(v)
            res = ops.solve_single(fv, [], [], fx, rtol, 1e-50, 500, defer_flush=defer)
            assert res.converged_reason > 0
            if defer:
                ops.flush_pending()
            x = fx.numpy()
            exact = lu.solve(B @ v)
            # ||r|| <= rtol ||b||  =>  ||x - x*|| <= rtol ||A^-1|| ||b||; cond(A) is ~4 here
            assert np.linalg.norm(A @ x - B @ v) <= 1.5 * rtol * np.linalg.norm(B @ v) + 1e-13 * np.linalg.norm(B @ v)
            np.testing.assert_allclose(x, exact, rtol=0, atol=max(20 * rtol, 1e-12) * np.abs(exact).max())
            its.append(res.iterations)
            if order != 0 and res.iterations > 0:
                h0, _, cnt = _history(ops)
                assert cnt == min(step + 1, 4)
                d = hip_ctx.torch.empty(n, dtype=hip_ctx.torch.float64, device=hip_ctx.device)
                from beat import _hip

                _hip.check(ops.lib.beat_copy(hip_ctx.handle, C.c_void_p(d.data_ptr()), C.c_void_p(h0), n))
                np.testing.assert_allclose(d.cpu().numpy(), x - v, rtol=0, atol=1e-12 * np.abs(v).max())
        totals[order] = sum(its[4:])
        if not few:
            assert max(its) > 6  # several ring cycles per solve
    assert totals[4] <= totals[3] < totals[2] < totals[1] < totals[0], totals
    # "auto" switches between the quadratic and the cubic by the iteration counts it sees: never worse than the worse
    # of the two (plus the one solve in which it tries the other)
    assert totals["auto"] <= max(totals[3], totals[4]) + 2, totals


@pytest.mark.parametrize("defer", [False, True])
def test_guess_on_a_masked_domain_with_per_node_rows(hip_ctx, defer):
    """The per-node-row operators (voxel mask, beat_pde_create_var): the guess increment is materialised on the tissue
    nodes and gathered by the right-hand side.  Orders 1 and 2 end at the solution of the x0 = v_ iteration (tight
    tolerance, so several ring cycles per solve), nodes outside the tissue keep their value, iterations fall."""
    cells = (24, 20, 12)
    mesh, M, _, _ = _system(cells)
    cc = np.stack(np.meshgrid(*(np.arange(c) for c in cells[::-1]), indexing="ij"), -1).reshape(-1, 3)
    active = ((cc - np.array([4, 8, 8])) ** 2).sum(axis=1) < 10**2
    sols, totals = {}, {}
    for order in (0, 1, 2, 3, 4):
        ops = _ops(hip_ctx, cells, M, order, active=active)
        fv, fx = ops.new_field(), ops.new_field()
        its, xs = [], []
        for step in range(9):
            v = _moving_bump(mesh, 0.02 * step)
            fv.set(v)
            res = ops.solve_single(fv, [], [], fx, 1e-13, 1e-50, 500, defer_flush=defer)
            assert res.converged_reason > 0
            if defer:
                ops.flush_pending()
            xs.append(fx.numpy())
            its.append(res.iterations)
        sols[order], totals[order] = np.array(xs), sum(its[4:])
        assert max(its) > 6
    tissue = np.abs(sols[0][0] - _moving_bump(mesh, 0.0)) > 0
    assert 0.2 < tissue.mean() < 0.9
    for order in (1, 2, 3, 4):
        np.testing.assert_allclose(sols[order], sols[0], rtol=0, atol=1e-10 * np.abs(sols[0]).max())
        outside = ~tissue
        for step in range(9):
            np.testing.assert_array_equal(sols[order][step][outside], _moving_bump(mesh, 0.02 * step)[outside])
    assert totals[4] <= totals[3] < totals[2] < totals[1] < totals[0], totals


def test_guess_that_already_solves_the_system(hip_ctx, small_grid_path):
    """Second solve of the same system: x0 = v_ + d1 is the previous solution, the stopping test holds before the first
    iteration, and the answer must still be v_ + d1 (the increment is applied although no search direction exists)."""
    cells = (16, 12, 8)
    mesh, M, A, B = _system(cells)
    v = _moving_bump(mesh, 0.3)
    for defer in (False, True):
        ops = _ops(hip_ctx, cells, M, 1)
        fv, fx = ops.new_field(), ops.new_field()
        fv.set(v)
        first = ops.solve_single(fv, [], [], fx, 1e-10, 1e-50, 500)
        x1 = fx.numpy()
        assert first.iterations > 2
        fx.fill(0.0)
        again = ops.solve_single(fv, [], [], fx, 1e-9, 1e-50, 500, defer_flush=defer)
        assert again.iterations == 0 and again.converged_reason > 0
        if defer and small_grid_path == "multi-launch":
            assert ops.pending is not None and ops.pending[2] == 0
            ops.flush_pending()
        else:  # the one-launch solve leaves nothing pending
            assert ops.pending is None
        np.testing.assert_allclose(fx.numpy(), x1, rtol=0, atol=1e-13 * np.abs(x1).max())
        fv.set(np.full(mesh.num_nodes, -80.0))
        # a third solve whose answer is its v_ (K v = 0): the stale increment is a poor guess, not a wrong answer
        rest = ops.solve_single(fv, [], [], fx, 1e-9, 1e-50, 500)
        assert rest.converged_reason > 0
        np.testing.assert_allclose(fx.numpy(), -80.0, rtol=0, atol=1e-6)


def test_time_step_change_and_reset_drop_the_history(hip_ctx):
    cells = (12, 10, 8)
    mesh, M, A, B = _system(cells)
    ops = _ops(hip_ctx, cells, M, 2)
    fv, fx = ops.new_field(), ops.new_field()
    for step in range(5):
        fv.set(_moving_bump(mesh, 0.02 * step))
        ops.solve_single(fv, [], [], fx, 1e-9, 1e-50, 500)
        assert _history(ops)[2] == min(step + 1, 4)
    ops.guess_reset()
    assert _history(ops)[2] == 0
    ops.solve_single(fv, [], [], fx, 1e-9, 1e-50, 500)
    assert _history(ops)[2] == 1
    ops.set_timestep(0.01, 0.5, 0.025)
    assert _history(ops)[2] == 0
    # the polynomial preconditioner (classic loop) starts from x0 = v_ and leaves no history behind
    ops.solve_single(fv, [], [], fx, 1e-9, 1e-50, 500)
    ops.set_preconditioner(3)
    res = ops.solve_single(fv, [], [], fx, 1e-9, 1e-50, 500)
    assert res.converged_reason > 0 and _history(ops)[2] == 0
    from beat._hip import BeatHipError

    with pytest.raises(BeatHipError):
        ops.set_guess_order(5)


def test_split_step_with_guess_matches_split_step_without(hip_ctx):
    """The public-API split step (TP06 + diffusion, fused route: the guess and the last search directions are added to
    the potential by the next ionic kernel) with ksp_guess_order 0 to 4 at a tight tolerance: same trajectories,
    fewer iterations; reading the potential between steps (flush pass instead of the ionic kernel) changes nothing."""
    import beat
    from beat import grid as g
    from beat.models import tp06

    def run(order, peek):
        geo = beat.geometry.get_3D_slab_geometry(comm=g.COMM_WORLD, Lx=4.0, Ly=2.0, Lz=1.0, dx=0.25)
        mesh = geo.mesh
        time = g.Constant(mesh, 0.0)
        cond = beat.conductivities.default_conductivities("Niederer")
        cells = g.locate_entities(mesh, 3, lambda x: np.logical_and(x[0] <= 1.0 + 1e-10, x[1] <= 1.0 + 1e-10))
        tags = g.meshtags(mesh, 3, cells, np.full(len(cells), 1, dtype=np.int32))
        I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=tags, marker=1,
                                               mesh_unit="mm", amplitude=50_000.0)
        Mt = beat.conductivities.define_conductivity_tensor(f0=geo.f0, **cond)
        C_m = (1.0 * beat.units.ureg("uF/cm**2")).to("uF/mm**2").magnitude
        pde = beat.MonodomainModel(time=time, mesh=mesh, M=Mt, I_s=I_s, C_m=C_m, dx=I_s.dZ,
                                   params={"petsc_options": {"ksp_type": "cg", "ksp_rtol": 1e-12, "ksp_guess_order": order}})
        assert pde._ops.guess_order == order
        ode = beat.odesolver.DolfinODESolver(
            v_ode=g.Function(g.functionspace(mesh, ("Lagrange", 1))), v_pde=pde.state, fun=tp06.generalized_rush_larsen,
            init_states=tp06.init_state_values(), parameters=tp06.init_parameter_values(stim_amplitude=0.0),
            num_states=19, v_index=tp06.state_index("V"))
        solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode, theta=1.0)
        its = []
        for i in range(40):
            solver.step((i * 0.05, (i + 1) * 0.05))
            its.append(pde.ksp.iterations)
            if peek:
                np.asarray(pde.state.x.array)
        return ode.values.copy(), its

    base, its0 = run(0, False)
    assert base[17].max() > 0.0
    for order in (1, 2, 3, 4):
        vals, its = run(order, False)
        np.testing.assert_allclose(vals, base, rtol=5e-8, atol=1e-9)
        assert sum(its) < sum(its0)
        peeked, its_p = run(order, True)
        assert its_p == its
        np.testing.assert_array_equal(peeked, vals)
