"""Correctness evidence for ``bench.py --gpus N``: before anything is timed, a small problem is run DECOMPOSED on the N ranks and
UNDIVIDED on rank 0, through the public API, and the two potentials are compared -- so that the first run on a real multi-GPU
node yields a parity figure and not only a speed.  The reference's parallel CI does the same thing with its accuracy tests
(/root/reference/.github/workflows/main-mpi.yml:33 runs tests/test_monodomain_solver.py:33-216 under ``mpirun -n 2``).

Two cases, both 25 Godunov split steps (src/beat/monodomain_solver.py:53-116):
  slab   96 x 96 x 12 N nodes, anisotropic constant tensor (fibre at 30 degrees in xy), TP06, corner stimulus + a bump in V:
         the constant-coefficient register-row kernels, ghost planes of r, the direction formed on the ghost planes
  shell  a voxelised ellipsoidal shell in a 64 x 64 x 16 N box, fibre rotation through the wall (per-node rows), two parameter
         classes of TP06 through DolfinMultiODESolver (class kernel, compact layout, node map), endocardial surface stimulus,
         slabs cut by tissue weight: the per-node-row tile kernels on a decomposed grid (BASELINE.json configs[4]'s machinery)

``compare`` is transport- and backend-agnostic (the CPU suite runs it on gloo with oracle-backed operators,
tests/test_distributed_cpu.py); ``slab_case`` / ``shell_case`` need the HIP library.  Exit code: bench.py leaves with a non-zero
status when ``max_rel_diff`` exceeds TOLERANCE on any transport it reports.
"""
from __future__ import annotations

import numpy as np

TOLERANCE = 1e-9  # max |v_decomposed - v_undivided| / max |v_undivided| after the steps (solves at rtol 1e-10, fp64)
STEPS = 25
RTOL = 1e-10


def compare(dist, rank: int, world: int, v_local: np.ndarray, iters: list, reference_fn, group=None) -> dict:
    """Gather the ranks' slabs of the potential on rank 0 (rank order = z order), run ``reference_fn() -> (v, iters)`` there
    (the undivided problem), compare; every rank returns the same verdict dict:
    {max_rel_diff, max_abs_diff, iterations_equal, k (mean PCG iterations per step, decomposed), k_undivided, nodes, ok}."""
    import os

    if os.environ.get("BEAT_BENCH_TEST_PARITY_BREAK") == str(rank):  # tests of the failure path: one rank's slab is off by 1e-3
        v_local = np.array(v_local, dtype=np.float64, copy=True)
        v_local[v_local.size // 2] += 1e-3 * max(1.0, float(np.abs(v_local).max()))
    parts = [None] * world if rank == 0 else None
    if world > 1:
        dist.gather_object((np.ascontiguousarray(v_local), list(iters)), parts, dst=0, group=group)
    else:
        parts = [(np.ascontiguousarray(v_local), list(iters))]
    verdict = [None]
    if rank == 0:
        v_dec = np.concatenate([p[0] for p in parts])
        v_ref, it_ref = reference_fn()
        v_ref = np.asarray(v_ref)
        same_its = all(list(p[1]) == list(parts[0][1]) for p in parts)  # every rank latched at the same iteration, every step
        if v_ref.shape != v_dec.shape:
            verdict[0] = {"error": f"shape {v_dec.shape} decomposed against {v_ref.shape} undivided", "ok": False}
        else:
            scale = float(np.abs(v_ref).max())
            diff = float(np.abs(v_dec - v_ref).max())
            rel = diff / scale if scale > 0 else diff
            verdict[0] = {
                "max_rel_diff": rel, "max_abs_diff": diff, "scale": scale,
                "iterations_equal": bool(same_its and list(parts[0][1]) == list(it_ref)),
                "iterations_equal_across_ranks": bool(same_its),
                "k": float(np.mean(parts[0][1])) if len(parts[0][1]) else 0.0,
                "k_undivided": float(np.mean(it_ref)) if len(it_ref) else 0.0,
                "nodes": int(v_ref.size), "steps": len(it_ref), "finite": bool(np.isfinite(v_dec).all()),
                "ok": bool(np.isfinite(v_dec).all() and rel <= TOLERANCE),
            }
    if world > 1:
        dist.broadcast_object_list(verdict, src=0, group=group)
    return verdict[0]


# ---------------------------------------------------------------------------------------------------------------------
# the two cases through the public API (need the HIP library and a GPU)
# ---------------------------------------------------------------------------------------------------------------------
def _tp06_ic(tp06):
    return tp06.init_state_values(V=-85.23, Xr1=0.00621, Xr2=0.4712, Xs=0.0095, m=0.00172, h=0.7444, j=0.7045, d=3.373e-05,
                                  f=0.7888, f2=0.9755, fCass=0.9953, s=0.999998, r=2.42e-08, Ca_i=0.000126, R_prime=0.9073,
                                  Ca_SR=3.64, Ca_ss=0.00036, Na_i=8.604, K_i=136.89)


def _petsc(guess_order):
    opts = {"ksp_rtol": RTOL, "ksp_atol": 1e-50, "ksp_max_it": 500}
    if guess_order is not None:
        opts["ksp_guess_order"] = guess_order
    return opts


def _run(solver, pde, steps, dt):
    its, t = [], 0.0
    for _ in range(steps):
        solver.step((t, t + dt))
        its.append(int(pde.ksp.iterations))
        t += dt
    return np.asarray(pde.state.x.array, dtype=np.float64).copy(), its


def slab_case(comm, nz_nodes: int, steps: int = STEPS, dt: float = 0.05, nxy: int = 96, hook=None, guess_order=None, info=None):
    """(v_local, iterations per step) of the slab case on ``comm`` (COMM_WORLD: this rank's slab; COMM_SELF: the whole grid).
    ``hook(pde)`` may replace the transport of ``pde._diffusion`` before the first step."""
    import beat
    from beat import grid as g
    from beat.models import tp06

    h = 0.1
    mesh = g.create_box(comm, [np.zeros(3), np.array([(nxy - 1) * h, (nxy - 1) * h, (nz_nodes - 1) * h])],
                        [nxy - 1, nxy - 1, nz_nodes - 1])
    f0 = np.array([np.cos(np.pi / 6.0), np.sin(np.pi / 6.0), 0.0])
    M = 9.5301e-4 * np.outer(f0, f0) + 1.2576e-4 * (np.eye(3) - np.outer(f0, f0))
    time_c = g.Constant(mesh, 0.0)
    # S1-like stimulus in a corner block that reaches through every slab cut (all z), so that every rank carries part of it
    L, tol = 1.5, 1e-10
    cells = g.locate_entities(mesh, 3, lambda x: (x[0] <= L + tol) & (x[1] <= L + tol))
    tags = g.meshtags(mesh, 3, cells, np.full(len(cells), 1, dtype=np.int32))
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=1400.0 * beat.units.ureg("cm**-1"), time=time_c, subdomain_data=tags,
                                           marker=1, mesh_unit="mm", amplitude=50_000.0, start=0.0, duration=2.0)
    pde = beat.MonodomainModel(time=time_c, mesh=mesh, M=M, I_s=I_s, C_m=0.01, dx=I_s.dZ,
                               params={"theta": 0.5, "petsc_options": _petsc(guess_order)})
    ic = _tp06_ic(tp06)
    X = mesh.node_coordinates(pad3=True, local=True)
    c = 0.5 * np.array([(nxy - 1) * h, (nxy - 1) * h, (nz_nodes - 1) * h])
    init = np.repeat(np.asarray(ic, dtype=np.float64)[:, None], X.shape[0], axis=1)
    init[tp06.state_index("V")] += 70.0 * np.exp(-((X - c) ** 2).sum(axis=1) / (2.0 * 1.2**2))  # a bump across the cuts
    ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(g.functionspace(mesh, ("P", 1))), v_pde=pde.state,
                                         fun=tp06.generalized_rush_larsen, init_states=init,
                                         parameters=tp06.init_parameter_values(stim_amplitude=0.0), num_states=len(ic),
                                         v_index=tp06.state_index("V"))
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    if hook is not None:
        hook(pde)
    if info is not None:  # does the decomposed solve run inside the library (initial guess and all), or stage by stage from Python?
        info["in_library"] = pde._mesh.comm.size == 1 or getattr(pde._diffusion, "libcomm", None) is not None
    try:
        return _run(solver, pde, steps, dt)
    finally:
        if hook is not None and hasattr(hook, "done"):
            hook.done(pde)


def _shell_mask(n_xy: int, n_z: int, h: float):
    ax_xy = (np.arange(n_xy) + 0.5) / n_xy - 0.5
    ax_z = (np.arange(n_z) + 0.5) / n_z - 0.5
    Z, Y, X = np.meshgrid(ax_z, ax_xy, ax_xy, indexing="ij")
    ro = np.sqrt((X / 0.47) ** 2 + (Y / 0.43) ** 2 + (Z / 0.49) ** 2)
    ri = np.sqrt((X / 0.29) ** 2 + (Y / 0.27) ** 2 + (Z / 0.33) ** 2)
    mask = (ro < 1.0) & (ri > 1.0) & (Z < 0.38)
    depth = np.clip((ri - 1.0) / np.maximum(ri - ro, 1e-12), 0.0, 1.0)
    P = np.stack([X, Y, Z], axis=-1)
    rad = P / np.maximum(np.linalg.norm(P, axis=-1, keepdims=True), 1e-12)
    circ = np.cross(np.array([0.0, 0.0, 1.0]), rad)
    circ /= np.maximum(np.linalg.norm(circ, axis=-1, keepdims=True), 1e-12)
    longi = np.cross(rad, circ)
    ang = np.deg2rad(60.0 - 120.0 * depth)[..., None]
    f0 = (np.cos(ang) * circ + np.sin(ang) * longi).reshape(-1, 3)
    return mask, f0, depth.ravel()


def shell_case(comm, n_z: int, steps: int = STEPS, dt: float = 0.05, n_xy: int = 64, hook=None, guess_order=None, info=None):
    """(v_local, iterations per step) of the voxel-shell case: per-node rows, two TP06 parameter classes, surface stimulus."""
    import beat
    from beat import grid as g
    from beat.models import tp06

    h = 0.25
    mask, f0, _ = _shell_mask(n_xy, n_z, h)
    mesh = g.create_voxel_mesh(comm, mask, h)
    cond = beat.conductivities.default_conductivities("Bishop")
    M = beat.conductivities.define_conductivity_tensor(f0=g.CellField(mesh, f0), **cond)
    time_c = g.Constant(mesh, 0.0)
    # endocardial surface = exterior facets nearer to the inner ellipsoid (as tools/bench_biv.py tags them)
    facets = mesh.exterior_facets()
    xyz = g._node_xyz(mesh, mesh.facet_vertices(facets).ravel()).reshape(len(facets), 4, 3)
    box = np.array([n_xy * h, n_xy * h, n_z * h])
    ctr = xyz.mean(axis=1) / box - 0.5
    ro = np.sqrt((ctr[:, 0] / 0.47) ** 2 + (ctr[:, 1] / 0.43) ** 2 + (ctr[:, 2] / 0.49) ** 2)
    ri = np.sqrt((ctr[:, 0] / 0.29) ** 2 + (ctr[:, 1] / 0.27) ** 2 + (ctr[:, 2] / 0.33) ** 2)
    endo = np.abs(ri - 1.0) < np.abs(ro - 1.0)
    ft = g.meshtags(mesh, 2, facets[endo], np.full(int(endo.sum()), 10, dtype=np.int32))
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time_c, subdomain_data=ft, marker=10,
                                           mesh_unit="mm", amplitude=2000.0, start=0.0, duration=1.0)
    pde = beat.MonodomainModel(time=time_c, mesh=mesh, M=M, I_s=I_s, C_m=0.01,
                               params={"petsc_options": _petsc(guess_order)})
    # two parameter classes: the inner and the outer half of the wall (by distance of the node from the box centre line), -1 outside
    V = g.functionspace(mesh, ("P", 1))
    X = mesh.node_coordinates(pad3=True, local=True) / box - 0.5
    rn_o = np.sqrt((X[:, 0] / 0.47) ** 2 + (X[:, 1] / 0.43) ** 2 + (X[:, 2] / 0.49) ** 2)
    rn_i = np.sqrt((X[:, 0] / 0.29) ** 2 + (X[:, 1] / 0.27) ** 2 + (X[:, 2] / 0.33) ** 2)
    inner = np.abs(rn_i - 1.0) < np.abs(rn_o - 1.0)
    tissue = np.asarray(mesh.node_active())
    markers = g.Function(V)
    markers.x.array[:] = np.where(tissue, np.where(inner, 1.0, 2.0), -1.0)
    ic = _tp06_ic(tp06)
    keys = (1, 2)
    par = {1: tp06.init_parameter_values(stim_amplitude=0.0), 2: tp06.init_parameter_values(stim_amplitude=0.0, g_Ks=0.196)}
    ode = beat.odesolver.DolfinMultiODESolver(
        v_ode=g.Function(V), v_pde=pde.state, markers=markers, num_states={k: len(ic) for k in keys},
        fun={k: tp06.generalized_rush_larsen for k in keys}, init_states={k: ic for k in keys}, parameters=par,
        v_index={k: tp06.state_index("V") for k in keys})
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    if hook is not None:
        hook(pde)
    if info is not None:  # does the decomposed solve run inside the library (initial guess and all), or stage by stage from Python?
        info["in_library"] = pde._mesh.comm.size == 1 or getattr(pde._diffusion, "libcomm", None) is not None
    try:
        return _run(solver, pde, steps, dt)
    finally:
        if hook is not None and hasattr(hook, "done"):
            hook.done(pde)


def run_cases(dist, rank: int, world: int, hook=None, cases=("slab", "shell"), group=None) -> dict:
    """Both cases on the transport ``hook`` installs (None: whatever the package's DiffusionSolver chose): the decomposed run on
    every rank, the undivided one on rank 0 (grid.COMM_SELF), the comparison.  Returns {case: verdict}."""
    from beat import grid as g

    out = {}
    for case in cases:
        fn, nz = (slab_case, 12 * world) if case == "slab" else (shell_case, 16 * world)
        info = {}
        v, its = fn(g.COMM_WORLD, nz, hook=hook, info=info)
        # The undivided run takes the same iteration as the decomposed one: the library's loop with its extrapolated initial
        # guess, or -- when no communicator of the library's could be made and every rank fell back to the stage-driven loop
        # over torch.distributed, which starts from x0 = v_ -- order 0.  (Different starting points give different iterates,
        # each within rtol of the exact solve: a difference of the solver's tolerance, not of the decomposition.)
        order = None if info.get("in_library", True) else 0
        out[case] = compare(dist, rank, world, v, its, lambda: fn(g.COMM_SELF, nz, guess_order=order), group=group)
        out[case]["loop"] = "library" if order is None else "stage-driven"
    return out
