"""CPU: pins the oracle (oracle/*.py) to the reference.

* golden vectors generated from the reference's own code / model specification
  (tests/golden/*.npz, script tests/golden/make_golden.py);
* the analytic thresholds of the reference's tests (tests/test_monodomain.py,
  tests/test_stimulation.py, tests/test_odesolver.py);
* the Niederer activation-time table (demos/niederer_benchmark.py:315-319), which also decides
  between the two readings of gotranx's GRL1 linearisation."""

import json
from pathlib import Path

import numpy as np
import pytest

from oracle import fem, ionic, splitting, torord

GOLD = Path(__file__).resolve().parent / "golden"


# ---- TP06 against the reference's .ode specification -------------------------------------------
def test_tp06_rhs_jacobian_and_step_match_the_ode_spec():
    g = np.load(GOLD / "tp06_spec.npz")
    assert tuple(g["state_names"]) == ionic.TP06_STATES
    assert tuple(g["parameter_names"]) == ionic.TP06_PARAMETERS
    np.testing.assert_array_equal(g["state_defaults"], ionic.tp06_init_state_values())
    np.testing.assert_array_equal(g["parameter_defaults"], ionic.tp06_init_parameter_values())
    S, t, dt = g["states"], float(g["t"]), float(g["dt"])
    P = ionic.tp06_init_parameter_values()
    f, J = ionic.tp06_rhs_and_linearized(S, t, P)
    for i in range(19):
        np.testing.assert_allclose(f[i], g["rhs"][i], rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(J[i], g["jac_total"][i], rtol=1e-11, atol=1e-300)
    out = ionic.tp06_generalized_rush_larsen(S, t, dt, P)
    np.testing.assert_allclose(out, g["grl1_total"], rtol=1e-13, atol=1e-300)
    # with derivatives of the expressions as written only the 12 gates + R_prime would be Rush-Larsen
    assert list(g["explicit_jac_is_zero"]) == [n in ("Ca_i", "Ca_SR", "Ca_ss", "Na_i", "V", "K_i") for n in ionic.TP06_STATES]


def test_tp06_single_cell_action_potential_is_physiological():
    S = ionic.tp06_init_state_values()[:, None].copy()
    P = ionic.tp06_init_parameter_values()
    dt, t, vs = 0.05, 0.0, []
    for _ in range(int(400 / dt)):
        S = ionic.tp06_generalized_rush_larsen(S, t, dt, P)
        t += dt
        vs.append(S[17, 0])
    vs = np.array(vs)
    assert 30.0 < vs.max() < 45.0 and abs(vs[-1] + 85.0) < 2.0
    up = np.nonzero(vs > 0)[0][0] * dt
    assert 10.0 < up < 12.5  # fires right after the model's own stimulus (t = 10 ms)
    apd90 = (np.nonzero(vs > vs.min() + 0.1 * (vs.max() - vs.min()))[0][-1]) * dt - up
    assert 250.0 < apd90 < 330.0


# ---- ToR-ORd-dynCl: the hand restatement (oracle/torord.py) against the reference's .ode specification ----------
def test_torord_hand_restatement_matches_the_ode_spec():
    """oracle/torord.py shares nothing with the fixture's generator chain (no .ode parser, no SymPy: hand-ordered
    expressions + forward-mode dual numbers); the fixture is the reference's .ode text evaluated by
    tests/golden/ode_spec.py.  Names, defaults, RHS (1e-13), total self-derivatives (1e-11) and one GRL1 step for
    endo / epi / mid."""
    g = np.load(GOLD / "torord_spec.npz")
    assert tuple(g["state_names"]) == torord.TORORD_STATES
    assert tuple(g["parameter_names"]) == torord.TORORD_PARAMETERS
    np.testing.assert_array_equal(g["state_defaults"], torord.torord_init_state_values())
    np.testing.assert_array_equal(g["parameter_defaults"], torord.torord_init_parameter_values())
    S, t, dt = g["states"], float(g["t"]), float(g["dt"])
    for celltype in (0, 1, 2):
        P = torord.torord_init_parameter_values(celltype=float(celltype))
        f, J = torord.torord_rhs_and_linearized(S, t, P)
        np.testing.assert_allclose(f, g[f"rhs_celltype{celltype}"], rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(J, g[f"jac_celltype{celltype}"], rtol=1e-11, atol=1e-300)
        assert (np.abs(J) > 0).all()  # no state is advanced by forward Euler for structural reasons
        out = torord.torord_generalized_rush_larsen(S, t, dt, P)
        np.testing.assert_allclose(out, g[f"grl1_celltype{celltype}"], rtol=1e-12, atol=1e-300)
    # the cell types really differ
    assert not np.allclose(g["grl1_celltype0"], g["grl1_celltype1"], rtol=1e-9)


def test_torord_hand_restatement_along_an_action_potential():
    """One GRL1 step from the 60 states the fixture sampled along a paced action potential of the specification
    (upstroke, plateau, repolarisation), and per-node parameters: a (112, N) array with a different cell type per
    node gives what the uniform calls give."""
    g = np.load(GOLD / "torord_spec.npz")
    P = torord.torord_init_parameter_values()
    S = g["traj_states"]
    out = torord.torord_generalized_rush_larsen(S, float(g["traj_step_t"]), float(g["traj_dt"]), P)
    np.testing.assert_allclose(out, g["traj_grl1"], rtol=1e-11, atol=1e-300)
    n = S.shape[1]
    ct = np.arange(n) % 3
    Pn = np.repeat(P[:, None], n, axis=1)
    Pn[torord.torord_parameter_index("celltype")] = ct
    per_node = torord.torord_generalized_rush_larsen(S, 5.0, 0.02, Pn)
    for c in (0, 1, 2):
        ref = torord.torord_generalized_rush_larsen(S[:, ct == c], 5.0, 0.02, torord.torord_init_parameter_values(celltype=float(c)))
        np.testing.assert_array_equal(per_node[:, ct == c], ref)


def test_torord_land_hand_restatement_matches_the_ode_spec():
    """The Land variant (odes/torord/ToRORd_dynCl_endo_Land.ode, 52 states / 140 parameters) of oracle/torord.py against
    tests/golden/torord_land_spec.npz: names, defaults, RHS, total self-derivatives and one GRL1 step for six parameter
    sets (three cell types; stretched / lengthening / shortening cells with non-default troponin and tropomyosin
    exponents) at states that take every branch of the mechanics part, and one step from each of 60 states along a
    paced action potential."""
    g = np.load(GOLD / "torord_land_spec.npz")
    assert tuple(g["state_names"]) == torord.TORORD_LAND_STATES
    assert tuple(g["parameter_names"]) == torord.TORORD_LAND_PARAMETERS
    np.testing.assert_array_equal(g["state_defaults"], torord.torord_land_init_state_values())
    np.testing.assert_array_equal(g["parameter_defaults"], torord.torord_land_init_parameter_values())
    S, t, dt = g["states"], float(g["t"]), float(g["dt"])
    names = list(g["state_names"])
    zs, cd, ct = S[names.index("Zetas")], S[names.index("Cd")], S[names.index("CaTrpn")]
    assert (zs > 0).any() and (zs < -1).any() and ((zs > -1) & (zs < 0)).any()   # the three branches of gammasu
    assert (ct ** -1.2 < 100).any() and (ct ** -1.2 > 100).any() and (cd > 0).any() and (cd < -0.2 + 0.25).any()
    for k, P in enumerate(g["parameter_sets"]):
        f, J = torord.torord_land_rhs_and_linearized(S, t, P)
        np.testing.assert_allclose(f, g["rhs"][k], rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(J, g["jac"][k], rtol=1e-11, atol=1e-300)
        assert (np.abs(J) > 0).all()
        out = torord.torord_land_generalized_rush_larsen(S, t, dt, P)
        np.testing.assert_allclose(out, g["grl1"][k], rtol=1e-12, atol=1e-300)
    assert not np.allclose(g["grl1"][0], g["grl1"][3], rtol=1e-9)  # stretch matters
    out = torord.torord_land_generalized_rush_larsen(g["traj_states"], float(g["traj_step_t"]), float(g["traj_dt"]),
                                                     g["parameter_defaults"])
    np.testing.assert_allclose(out, g["traj_grl1"], rtol=1e-11, atol=1e-300)
    # the electrophysiology states follow the 45-state model except through the calcium equation: with the same
    # states, every derivative but cai's (and the mechanics rows) equals the base model's
    base_names = list(torord.TORORD_STATES)
    Sb = np.array([S[names.index(n)] for n in base_names])
    fb, _ = torord.torord_rhs_and_linearized(Sb, t, g["parameter_sets"][1][:112])
    fl, _ = torord.torord_land_rhs_and_linearized(S, t, g["parameter_sets"][1])
    for i, n in enumerate(base_names):
        if n != "cai":
            np.testing.assert_array_equal(fb[i], fl[names.index(n)])


def test_torord_single_cell_action_potential_is_physiological():
    """The model's own stimulus (-53 A/F for 1 ms at t = 0) fires one endocardial cell: overshoot 20-60 mV, APD90
    in the human ventricular range, back at rest after 450 ms (dt = 0.1 ms keeps the CPU suite short;
    demos/biv_endocardial.py:134 uses 0.05)."""
    S = torord.torord_init_state_values()[:, None].copy()
    P = torord.torord_init_parameter_values()
    vi = torord.torord_state_index("v")
    dt, t, vs = 0.1, 0.0, []
    for _ in range(int(450 / dt)):
        S = torord.torord_generalized_rush_larsen(S, t, dt, P)
        t += dt
        vs.append(S[vi, 0])
    vs = np.array(vs)
    assert np.isfinite(S).all() and 20.0 < vs.max() < 60.0 and vs[-1] < -85.0
    up = np.nonzero(vs > 0)[0][0]
    apd90 = (np.nonzero(vs > vs.min() + 0.1 * (vs.max() - vs.min()))[0][-1] - up) * dt
    assert 220.0 < apd90 < 330.0, apd90


def test_grl1_converges_to_the_ode_solution():
    """GRL1 -> exact solution as dt -> 0 (first order), checked against scipy's LSODA on the same RHS."""
    from scipy.integrate import solve_ivp

    P = ionic.tp06_init_parameter_values(stim_start=1.0)
    y0 = ionic.tp06_init_state_values()
    T = 5.0
    sol = solve_ivp(lambda t, y: np.array([float(v) for v in ionic.tp06_rhs(y, t, P)]), (0, T), y0, method="LSODA",
                    rtol=1e-10, atol=1e-12, max_step=0.01)
    ref = sol.y[:, -1]
    errs = []
    for dt in (0.02, 0.01, 0.005):
        S = y0[:, None].copy()
        for k in range(int(round(T / dt))):
            S = ionic.tp06_generalized_rush_larsen(S, k * dt, dt, P)
        errs.append(abs(S[17, 0] - ref[17]))
    assert errs[0] > errs[1] > errs[2] and errs[2] < 0.5
    assert 0.7 < np.log2(errs[0] / errs[1]) < 1.6


# ---- splitting / data movement against the reference's own modules ----------------------------------
def _simple(states, t, dt, parameters):
    return ionic.simple_ode_forward_euler(states, t, dt, parameters)


def test_dolfin_ode_solver_data_movement_matches_reference():
    g = np.load(GOLD / "splitting_reference.npz")
    n = 7
    ode = splitting.OracleODE(n, np.array([1.0, 2.0]), np.array([1.5, 0.5]), _simple, 2, 0)
    np.testing.assert_array_equal(ode.values, g["dolfin_values_init"])
    ode.step(0.0, 0.1)
    np.testing.assert_array_equal(ode.values, g["dolfin_values_after_step"])
    np.testing.assert_array_equal(ode.v_ode, g["dolfin_v_ode_before_to_dolfin"])
    ode.to_dolfin()
    np.testing.assert_array_equal(ode.v_ode, g["dolfin_v_ode_after_to_dolfin"])
    ode.ode_to_pde()
    np.testing.assert_array_equal(ode.v_pde, g["dolfin_v_pde_after_ode_to_pde"])
    ode.v_pde[:] = np.linspace(-1.0, 1.0, n)
    ode.pde_to_ode()
    ode.from_dolfin()
    np.testing.assert_array_equal(ode.values, g["dolfin_values_after_from_dolfin"])


def test_multi_ode_solver_matches_reference():
    g = np.load(GOLD / "splitting_reference.npz")
    multi = splitting.OracleMultiODE(
        g["multi_markers"],
        {0: np.array([1.0, 2.0]), 1: np.array([3.0, 4.0]), 2: np.array([5.0, 6.0])},
        {0: np.array([1.0, 1.0]), 1: np.array([2.0, 0.5]), 2: np.array([0.25, 4.0])},
        {0: _simple, 1: _simple, 2: _simple}, {0: 2, 1: 2, 2: 2}, {0: 0, 1: 0, 2: 0})
    multi.step(0.0, 0.1)
    multi.to_dolfin()
    np.testing.assert_array_equal(multi.v_ode, g["multi_v_ode_after_to_dolfin"])
    np.testing.assert_array_equal(multi.full_values(), g["multi_full_values_after_step"])
    multi.v_ode[:] = np.arange(10.0)
    multi.from_dolfin()
    np.testing.assert_array_equal(multi.full_values(), g["multi_full_values_after_from_dolfin"])
    for mk in (0, 1, 2):
        np.testing.assert_array_equal(multi.values[mk], g[f"multi_values_marker{mk}"])


class _RecordingPDE:
    def __init__(self, n, log):
        self.state, self.v_, self.log = np.zeros(n), np.zeros(n), log

    def assign_previous(self):
        self.log.append("pde.assign_previous")
        self.v_[:] = self.state

    def step(self, interval):
        self.log.append(f"pde.step({interval[0]:.6f},{interval[1]:.6f})")
        self.state[:] = 0.5 * self.v_ + 1.0


@pytest.mark.parametrize("theta", [1.0, 0.5])
def test_splitting_call_order_and_values_match_reference(theta):
    g = np.load(GOLD / "splitting_reference.npz")
    meta = json.loads((GOLD / "splitting_reference.json").read_text())
    tag = f"split_theta{theta:g}".replace(".", "p")
    log = []
    pde = _RecordingPDE(5, log)

    def fun(states, t, dt, parameters):
        log.append(f"ode.fun(t={t:.6f},dt={dt:.6f})")
        return _simple(states, t, dt, parameters)

    ode = splitting.OracleODE(5, g[f"{tag}_init_states"], np.array([1.0, 1.0]), fun, 2, 0)
    solver = splitting.OracleSplitting(pde, ode, theta)
    assert log == meta[f"{tag}_calls_init"]
    del log[:]
    solver.step((0.0, 0.1))
    assert log == meta[f"{tag}_calls_step"]
    np.testing.assert_array_equal(ode.values, g[f"{tag}_values_after_step"])
    np.testing.assert_array_equal(pde.state, g[f"{tag}_pde_state_after_step"])
    np.testing.assert_array_equal(pde.v_, g[f"{tag}_pde_prev_after_step"])
    del log[:]
    solver.solve((0.1, 0.4), dt=0.1)
    assert log == meta[f"{tag}_calls_solve"]
    np.testing.assert_array_equal(ode.values, g[f"{tag}_values_after_solve"])


def test_known_answer_forward_euler_step():
    """tests/test_odesolver.py:86-90: v = v0 - s0 dt, s = s0 + v0 dt."""
    ode = splitting.OracleODE(4, np.array([1.0, 2.0]), np.array([1, 1]), _simple, 2, 0)
    ode.step(0.0, 0.1)
    assert np.allclose(ode.values[0], 0.8) and np.allclose(ode.values[1], 2.1)


# ---- FEM restatement against the reference's analytic tests ------------------------------------------
@pytest.mark.parametrize("Mv,amp,thr", [
    (0.0, lambda t: np.cos(t), 1e-4),
    (1.0, lambda t: np.cos(t) + 8 * np.pi**2 * np.sin(t), 2e-4),
    (2.0, lambda t: np.cos(t) + 16 * np.pi**2 * np.sin(t), 2e-4),
])
def test_fem_manufactured_solution_thresholds(Mv, amp, thr):
    """tests/test_monodomain.py:11-64."""
    N, dt = 15, 1e-3
    T = 10 * dt
    mesh = fem.BoxMesh((N, N), (1.0, 1.0))
    w = fem.load_vector(mesh, lambda x: np.cos(2 * np.pi * x[0]) * np.cos(2 * np.pi * x[1]))
    model = fem.OracleMonodomainModel(mesh, Mv, [fem.OracleStimulus(amp, w)], theta=0.5)
    v = model.solve((0, T), dt=dt)
    err = fem.l2_error(mesh, v, lambda x: np.cos(2 * np.pi * x[0]) * np.cos(2 * np.pi * x[1]) * np.sin(T))
    assert err < thr


def test_fem_spatial_rate():
    """tests/test_monodomain.py:67-104: rate >= 2."""
    errors, dt = [], 1e-3
    T = 10 * dt
    for N in (4, 8, 16, 32):
        mesh = fem.BoxMesh((N, N), (1.0, 1.0))
        w = fem.load_vector(mesh, lambda x: np.cos(2 * np.pi * x[0]) * np.cos(2 * np.pi * x[1]))
        model = fem.OracleMonodomainModel(mesh, 1.0, [fem.OracleStimulus(lambda t: np.cos(t) + 8 * np.pi**2 * np.sin(t), w)])
        v = model.solve((0, T), dt=dt)
        errors.append(fem.l2_error(mesh, v, lambda x: np.cos(2 * np.pi * x[0]) * np.cos(2 * np.pi * x[1]) * np.sin(T)))
    rates = [np.log2(a / b) for a, b in zip(errors[:-1], errors[1:])]
    assert all(r >= 2.0 for r in rates), rates


def test_fem_stimulus_accumulation_and_solve_quirk():
    """tests/test_stimulation.py:12-48 (M = 0, 1-D): closed-form accumulation incl. the missing
    assign_previous() after the last step of solve()."""
    mesh = fem.BoxMesh((10,), (1.0,))
    value, end, start, dt = 2.0, 1.0, 0.5, 0.01
    w = fem.stimulus_weights(mesh)
    model = fem.OracleMonodomainModel(mesh, 0.0, [fem.OracleStimulus(fem.window(start, end - start, value), w)])
    model.step((0.0, 0.4))
    assert np.allclose(model.state, 0.0)
    model.solve((0.4, 0.9), dt=dt)
    assert np.allclose(model.state, value * (0.9 - start))
    model.solve((0.9, end + dt), dt=dt)
    assert np.allclose(model.state, (end - start - dt) * value)
    model.solve((end + dt, 2 * end), dt=dt)
    assert np.allclose(model.state, (end - start - dt) * value)


def test_stencil_is_the_assembled_operator():
    rng = np.random.default_rng(0)
    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M3 = 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
    mesh = fem.BoxMesh((5, 4, 3), (2.5, 2.0, 1.5))
    mt, kt = fem.stencil_table(3, (0.5, 0.5, 0.5), M3, 1.0, 1.0)
    x = rng.standard_normal(mesh.num_nodes)
    assert np.abs(fem.apply_stencil(mt, mesh.shape_nodes, x) - fem.assemble_mass(mesh) @ x).max() < 1e-15
    assert np.abs(fem.apply_stencil(kt, mesh.shape_nodes, x) - fem.assemble_stiffness(mesh, M3) @ x).max() < 1e-17


# ---- Niederer table ---------------------------------------------------------------------------------
NIEDERER_DX05_DT005 = dict(P1=1.25, P2=51.1, P3=34.9, P4=58.9, P5=14.1, P6=49.5, P7=34.0, P8=56.65, P9=26.05)


def _niederer(ode_step, dt=0.05, T=62.0):
    Lx, Ly, Lz, dx = 20.0, 7.0, 3.0, 0.5
    mesh = fem.BoxMesh((40, 14, 6), (Lx, Ly, Lz))
    M = np.diag([0.0009529837251356239, 0.00012575841147269718, 0.00012575841147269718])
    cells = mesh.locate_cells(lambda x: (x[0] <= 1.5 + 1e-10) & (x[1] <= 1.5 + 1e-10) & (x[2] <= 1.5 + 1e-10))
    w = fem.stimulus_weights(mesh, cells)
    amp = 50000.0 / 1400.0 / 100.0  # (50000 uA/cm^3) / (1400 /cm) in uA/mm^2
    model = fem.OracleMonodomainModel(mesh, M, [fem.OracleStimulus(fem.window(0.0, 2.0, amp), w)], C_m=0.01,
                                      theta=0.5, default_timestep=dt)
    S = np.repeat(ionic.tp06_init_state_values()[:, None], mesh.num_nodes, axis=1)
    P = ionic.tp06_init_parameter_values(stim_amplitude=0.0)
    pts = {"P1": (0, 0, 0), "P2": (0, Ly, 0), "P3": (Lx, 0, 0), "P4": (Lx, Ly, 0), "P5": (0, 0, Lz), "P6": (0, Ly, Lz),
           "P7": (Lx, 0, Lz), "P8": (Lx, Ly, Lz), "P9": (Lx / 2, Ly / 2, Lz / 2)}
    ids = {k: int(round(p[0] / dx)) + 41 * (int(round(p[1] / dx)) + 15 * int(round(p[2] / dx))) for k, p in pts.items()}
    at = {k: -1.0 for k in pts}
    t = 0.0
    while t < T and any(a < 0 for a in at.values()):
        S = ode_step(S, t, dt, P)
        model.state[:] = S[17]
        model.assign_previous()
        model.step((t, t + dt))
        S[17] = model.state
        for k, i in ids.items():
            if model.state[i] > 0 and at[k] < 0:
                at[k] = t
        t += dt
    return at


def test_niederer_table_pins_the_grl1_variant():
    """With total self-derivatives every point is within 2 dt of the reference's row
    (dx = 0.5, dt = 0.05); with derivatives of the expressions as written (V, Ca_i, Ca_SR, Ca_ss,
    Na_i, K_i on forward Euler) the far corners are 8-13 dt early."""
    dt = 0.05
    at = _niederer(ionic.tp06_generalized_rush_larsen, dt)
    for k, ref in NIEDERER_DX05_DT005.items():
        assert abs(at[k] - ref) <= 2 * dt + 1e-9, (k, at[k], ref)

    fe_states = [ionic.tp06_state_index(n) for n in ("Ca_i", "Ca_SR", "Ca_ss", "Na_i", "V", "K_i")]

    def explicit_variant(S, t, h, P):
        new = ionic.tp06_generalized_rush_larsen(S, t, h, P)
        fe = ionic.tp06_forward_euler(S, t, h, P)
        new[fe_states] = fe[fe_states]
        return new

    at2 = _niederer(explicit_variant, dt)
    miss = {k: at2[k] - ref for k, ref in NIEDERER_DX05_DT005.items()}
    assert miss["P3"] < -8 * dt and miss["P4"] < -8 * dt and miss["P8"] < -8 * dt, miss
