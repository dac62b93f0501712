"""GPU: the HIP path against the committed golden fixtures (tests/golden/, generated from the
reference's own modules and model specification by tests/golden/make_golden.py)."""

import json
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.fixture(autouse=True)
def _ctx(hip_ctx):
    return hip_ctx


def test_tp06_kernel_matches_the_ode_spec_golden(hip_ctx):
    """One GRL1 step of the HIP kernel vs the reference's .ode specification evaluated independently
    (SymPy total self-derivatives): 1e-11 relative to the state scale."""
    from beat.models import tp06

    g = np.load(GOLD / "tp06_spec.npz")
    assert tuple(g["state_names"]) == tp06.generalized_rush_larsen.state_names
    assert tuple(g["parameter_names"]) == tp06.generalized_rush_larsen.parameter_names
    np.testing.assert_array_equal(g["state_defaults"], tp06.init_state_values())
    np.testing.assert_array_equal(g["parameter_defaults"], tp06.init_parameter_values())
    out = tp06.generalized_rush_larsen(states=g["states"], t=float(g["t"]), parameters=tp06.init_parameter_values(),
                                       dt=float(g["dt"]))
    ref = g["grl1_total"]
    err = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)
    assert err.max() < 1e-11, err.max()
    # and a 1-D state vector (single cell), as single_cell-style callers pass it
    one = tp06.generalized_rush_larsen(states=g["states"][:, 0], t=float(g["t"]), parameters=tp06.init_parameter_values(),
                                       dt=float(g["dt"]))
    assert one.shape == (19,) and np.allclose(one, ref[:, 0], rtol=1e-11)


def test_dolfin_ode_solver_matches_reference_golden():
    import beat
    from beat import grid as g

    gold = np.load(GOLD / "splitting_reference.npz")
    mesh = g.create_unit_interval(g.COMM_WORLD, 6)  # 7 nodes
    V = g.functionspace(mesh, ("P", 1))
    v_ode, v_pde = g.Function(V), g.Function(V)
    ode = beat.odesolver.DolfinODESolver(v_ode=v_ode, v_pde=v_pde, init_states=np.array([1.0, 2.0]),
                                         parameters=np.array([1.5, 0.5]), fun=beat.models.simple.forward_euler,
                                         num_states=2, v_index=0)
    np.testing.assert_array_equal(ode.values, gold["dolfin_values_init"])
    ode.step(0.0, 0.1)
    np.testing.assert_allclose(ode.values, gold["dolfin_values_after_step"], rtol=1e-15)
    np.testing.assert_array_equal(np.asarray(v_ode.x.array), gold["dolfin_v_ode_before_to_dolfin"])
    ode.to_dolfin()
    np.testing.assert_allclose(np.asarray(v_ode.x.array), gold["dolfin_v_ode_after_to_dolfin"], rtol=1e-15)
    ode.ode_to_pde()
    np.testing.assert_allclose(np.asarray(v_pde.x.array), gold["dolfin_v_pde_after_ode_to_pde"], rtol=1e-15)
    v_pde.x.array[:] = np.linspace(-1.0, 1.0, 7)
    ode.pde_to_ode()
    ode.from_dolfin()
    np.testing.assert_allclose(ode.values, gold["dolfin_values_after_from_dolfin"], rtol=1e-15)


def test_multi_ode_solver_matches_reference_golden():
    import beat
    from beat import grid as g

    gold = np.load(GOLD / "splitting_reference.npz")
    mesh = g.create_unit_interval(g.COMM_WORLD, 9)
    V = g.functionspace(mesh, ("P", 1))
    v_ode, v_pde, markers = g.Function(V), g.Function(V), g.Function(V)
    markers.x.array[:] = gold["multi_markers"]
    fe = beat.models.simple.forward_euler
    multi = beat.odesolver.DolfinMultiODESolver(
        v_ode=v_ode, v_pde=v_pde, markers=markers,
        init_states={0: np.array([1.0, 2.0]), 1: np.array([3.0, 4.0]), 2: np.array([5.0, 6.0])},
        parameters={0: np.array([1.0, 1.0]), 1: np.array([2.0, 0.5]), 2: np.array([0.25, 4.0])},
        fun={0: fe, 1: fe, 2: fe}, num_states={0: 2, 1: 2, 2: 2}, v_index={0: 0, 1: 0, 2: 0})
    multi.step(0.0, 0.1)
    multi.to_dolfin()
    np.testing.assert_allclose(np.asarray(v_ode.x.array), gold["multi_v_ode_after_to_dolfin"], rtol=1e-15)
    np.testing.assert_allclose(multi.full_values, gold["multi_full_values_after_step"], rtol=1e-15)
    v_ode.x.array[:] = np.arange(10.0)
    multi.from_dolfin()
    np.testing.assert_allclose(multi.full_values, gold["multi_full_values_after_from_dolfin"], rtol=1e-15)
    for mk in (0, 1, 2):
        np.testing.assert_allclose(multi.values(mk), gold[f"multi_values_marker{mk}"], rtol=1e-15)


class _RecordingPDE:
    """Same affine fake diffusion step as the fixture generator: state <- 0.5 v_ + 1."""

    def __init__(self, state, log):
        self.state, self.log = state, log
        self.v_ = np.zeros(state.x.array.size)

    def assign_previous(self):
        self.log.append("pde.assign_previous")
        self.v_[:] = np.asarray(self.state.x.array)

    def step(self, interval):
        self.log.append(f"pde.step({interval[0]:.6f},{interval[1]:.6f})")
        self.state.x.array[:] = 0.5 * self.v_ + 1.0


@pytest.mark.parametrize("theta", [1.0, 0.5])
def test_splitting_solver_call_order_and_values_match_reference_golden(theta):
    import beat
    from beat import grid as g

    gold = np.load(GOLD / "splitting_reference.npz")
    meta = json.loads((GOLD / "splitting_reference.json").read_text())
    tag = f"split_theta{theta:g}".replace(".", "p")
    mesh = g.create_unit_interval(g.COMM_WORLD, 4)
    V = g.functionspace(mesh, ("P", 1))
    log = []
    pde = _RecordingPDE(g.Function(V), log)
    ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(V), v_pde=pde.state, init_states=gold[f"{tag}_init_states"],
                                         parameters=np.array([1.0, 1.0]), fun=beat.models.simple.forward_euler,
                                         num_states=2, v_index=0)
    real_step = ode.step

    def logged_step(t0, dt):
        log.append(f"ode.fun(t={t0:.6f},dt={dt:.6f})")
        real_step(t0, dt)

    ode.step = logged_step
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode, theta=theta)
    assert log == meta[f"{tag}_calls_init"]
    del log[:]
    solver.step((0.0, 0.1))
    assert log == meta[f"{tag}_calls_step"]
    np.testing.assert_allclose(ode.values, gold[f"{tag}_values_after_step"], rtol=1e-15)
    np.testing.assert_allclose(np.asarray(pde.state.x.array), gold[f"{tag}_pde_state_after_step"], rtol=1e-15)
    np.testing.assert_allclose(pde.v_, gold[f"{tag}_pde_prev_after_step"], rtol=1e-15)
    del log[:]
    solver.solve((0.1, 0.4), dt=0.1)
    assert log == meta[f"{tag}_calls_solve"]
    np.testing.assert_allclose(ode.values, gold[f"{tag}_values_after_solve"], rtol=1e-15)


def test_free_running_ode_solve_matches_reference_golden():
    import beat

    gold = np.load(GOLD / "splitting_reference.npz")
    states = np.zeros((2, 3))
    states.T[:] = [1.0, 0.0]
    trace = np.zeros((12, 3))
    beat.odesolver.solve(fun=beat.models.simple.forward_euler, t_bound=1.0, states=states, V=trace, V_index=0, dt=0.1,
                         parameters=np.array([1.0, 1.0]))
    np.testing.assert_allclose(trace, gold["solve_trace"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(states, gold["solve_final_states"], rtol=1e-14)


def test_single_cell_steady_state_device_loop(tmp_path):
    """beat.single_cell.get_steady_state (src/beat/single_cell.py:86-156): the in-kernel nbeats x arange(0, BCL, dt)
    loop gives what stepping the oracle one call at a time gives, and the tracked trace has the reference's
    shape (ceil(len(times)/save_freq)*nbeats, len(track_indices))."""
    import beat
    from beat.models import tp06
    from oracle import ionic

    P = tp06.init_parameter_values(stim_start=1.0)
    y0 = tp06.init_state_values()
    dt, BCL, nbeats = 0.05, 40, 2
    y = beat.single_cell.get_steady_state(tp06.generalized_rush_larsen, y0, P, tmp_path, nbeats=nbeats, BCL=BCL, dt=dt,
                                          track_indices=[tp06.state_index("V"), tp06.state_index("Ca_i")])
    ref = y0[:, None].copy()
    times = np.arange(0.0, BCL, dt)
    for _ in range(nbeats):
        for t in times:
            ref = ionic.tp06_generalized_rush_larsen(ref, t, dt, P)
    assert np.abs(y - ref[:, 0]).max() / np.abs(ref).max() < 1e-7 and ref[17, 0] > -60.0  # mid action potential
    tracked = np.load(next(tmp_path.glob("tracked_values_*.npy")))
    assert tracked.shape == (int(np.ceil(len(times) / 20) * nbeats), 2)
    assert tracked[0, 0] == y0[17] and tracked[:, 0].max() > 0.0
    # cached on the second call
    y2 = beat.single_cell.get_steady_state(tp06.generalized_rush_larsen, y0, P, tmp_path, nbeats=nbeats, BCL=BCL, dt=dt,
                                           track_indices=[17, 13])
    np.testing.assert_array_equal(y, y2)


@pytest.mark.parametrize("celltype", [0, 1, 2])
def test_torord_kernel_matches_the_ode_spec_golden(celltype):
    """ToR-ORd-dynCl (45 states): one GRL1 step of the generated HIP kernel vs the reference's .ode
    specification evaluated independently (NumPy RHS, SymPy total self-derivatives of the fully resolved
    expressions), endo / epi / mid: 1e-10 relative to the state scale."""
    from beat.models import torord

    g = np.load(GOLD / "torord_spec.npz")
    assert tuple(g["state_names"]) == torord.generalized_rush_larsen.state_names
    assert tuple(g["parameter_names"]) == torord.generalized_rush_larsen.parameter_names
    np.testing.assert_array_equal(g["state_defaults"], torord.init_state_values())
    np.testing.assert_array_equal(g["parameter_defaults"], torord.init_parameter_values())
    P = torord.init_parameter_values(celltype=float(celltype))
    out = torord.generalized_rush_larsen(states=g["states"], t=float(g["t"]), parameters=P, dt=float(g["dt"]))
    ref = g[f"grl1_celltype{celltype}"]
    assert np.isfinite(out).all()
    scale = np.maximum(np.abs(ref), 1e-6 * np.abs(g["state_defaults"])[:, None] + 1e-12)
    err = np.abs(out - ref) / scale
    assert err.max() < 1e-10, (err.max(), np.unravel_index(err.argmax(), err.shape))


def test_torord_action_potential():
    """Pacing one ToR-ORd cell for one beat with the in-kernel loop: physiological upstroke and repolarisation."""
    from beat.models import torord

    y0 = torord.init_state_values()
    P = torord.init_parameter_values()
    vi = torord.state_index("v")
    y, tr = torord.generalized_rush_larsen.run(y0, P, dt=0.02, nsteps=25000, nbeats=1, track_indices=[vi], save_freq=50)
    v = tr[:, 0]
    # sampled once per ms, so the sub-millisecond overshoot peak itself is usually missed
    assert np.isfinite(y).all() and 5.0 < v.max() < 60.0 and v[-1] < -80.0
    apd90 = (np.nonzero(v > v.min() + 0.1 * (v.max() - v.min()))[0][-1] - np.nonzero(v > 0)[0][0]) * 1.0
    assert 200.0 < apd90 < 400.0, apd90


def test_torord_kernel_along_an_action_potential():
    """One GRL1 step from 60 states sampled along a paced action potential of the reference specification
    (upstroke, plateau, repolarisation): 1e-9 relative to the state scale."""
    from beat.models import torord

    g = np.load(GOLD / "torord_spec.npz")
    P = torord.init_parameter_values()
    out = torord.generalized_rush_larsen(states=g["traj_states"], t=float(g["traj_step_t"]), parameters=P,
                                         dt=float(g["traj_dt"]))
    ref = g["traj_grl1"]
    scale = np.maximum(np.abs(ref), 1e-6 * np.abs(g["state_defaults"])[:, None] + 1e-12)
    err = np.abs(out - ref) / scale
    bad = np.argwhere(err > 1e-9)
    assert len(bad) == 0, [(g["state_names"][i], float(g["traj_times"][j]), float(err[i, j])) for i, j in bad[:12]]


@pytest.mark.parametrize("model", ["torord", "torord_land"])
def test_torord_update_forms_agree_across_their_switch_points(model):
    """The ToR-ORd step takes the GRL1 increment of a non-gate state as f dt phi(J dt) by polynomial when |J dt| <= 1/16 (per
    lane) and updates the gates whose rate has a bound B by polynomial when dt B <= 1/32 (one uniform branch; B = 1/0.6 for
    the default parameters, i.e. dt <= 0.01875 ms), otherwise through exp() (csrc/torord_dyncl.h: advance, gate_b).  From
    the action-potential samples, with every parameter perturbed per node, one step at time steps on both sides of the gate
    switch and spanning the per-lane window (dt = 0.002 .. 0.2 ms) equals the NumPy oracle -- which knows only the
    exp() form -- to 1e-9 of the state scale."""
    from beat.models import torord, torord_land

    from oracle import torord as otor

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_torord_host import _perturbed_parameters

    if model == "torord":
        m, g = torord, np.load(GOLD / "torord_spec.npz")
        names, P0, ref_fn = list(otor.TORORD_PARAMETERS), otor.torord_init_parameter_values(), otor.torord_generalized_rush_larsen
    else:
        m, g = torord_land, np.load(GOLD / "torord_land_spec.npz")
        names, P0 = list(otor.TORORD_LAND_PARAMETERS), otor.torord_land_init_parameter_values()
        ref_fn = otor.torord_land_generalized_rush_larsen
    S = g["traj_states"]
    P = _perturbed_parameters(P0, names, S.shape[1], 9)
    scale0 = 1e-6 * np.abs(g["state_defaults"])[:, None] + 1e-12
    for dt in (0.002, 0.01, 0.0187, 0.0188, 0.05, 0.2):
        out = m.generalized_rush_larsen(states=S, t=0.4, parameters=P, dt=dt)
        ref = ref_fn(S, 0.4, dt, P)
        err = np.abs(out - ref) / np.maximum(np.abs(ref), scale0)
        assert err.max() < 1e-9, (dt, float(err.max()), g["state_names"][int(np.unravel_index(err.argmax(), err.shape)[0])])


def test_torord_kernel_on_unphysiological_states_agrees_wherever_the_specification_is_finite():
    """The differential run of tests/test_torord_host.py (6000 states nobody should reach: gates in -0.1 .. 1.1, every
    concentration, load and flux scaled by -1 .. 10) on the device build: wherever the NumPy oracle returns finite values,
    the kernel returns the same ones to 1e-7 of the state scale.  (A negative SR load is what perturbed parameter sets do
    reach for a while: tools/soak_cells.py.)"""
    import warnings

    from beat.models import torord

    from oracle import torord as otor

    g = np.load(GOLD / "torord_spec.npz")
    names = list(g["state_names"])
    rng = np.random.default_rng(5)
    n = 6000
    base = g["traj_states"]
    S = base[:, rng.integers(0, base.shape[1], n)].copy()
    scaled = ("nai", "nass", "ki", "kss", "cai", "cass", "cansr", "cajsr", "cli", "clss", "CaMKt", "Jrel_np", "Jrel_p")
    for k, name in enumerate(names):
        if name == "v":
            S[k] = rng.uniform(-135, 100, n)
        elif name in scaled:
            S[k] *= rng.choice([-1.0, -0.1, 0.05, 0.5, 1.0, 2.0, 10.0], n, p=[0.05, 0.05, 0.1, 0.2, 0.3, 0.2, 0.1])
        else:
            S[k] = rng.uniform(-0.1, 1.1, n)
    P = otor.torord_init_parameter_values()
    scale0 = 1e-6 * np.abs(g["state_defaults"])[:, None] + 1e-12
    for dt in (0.01, 0.05):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = otor.torord_generalized_rush_larsen(S, 0.3, dt, P)
        out = torord.generalized_rush_larsen(states=S, t=0.3, parameters=P, dt=dt)
        cols = np.isfinite(ref).all(axis=0)
        assert cols.sum() > n // 3 and np.isfinite(out[:, cols]).all()
        err = np.abs(out[:, cols] - ref[:, cols]) / np.maximum(np.abs(ref[:, cols]), scale0)
        assert err.max() < 1e-7, (dt, float(err.max()), names[int(np.unravel_index(err.argmax(), err.shape)[0])])


def test_torord_land_kernel_matches_the_ode_spec_golden_and_the_oracle():
    """The Land instance of the ToR-ORd kernel (beat.models.torord_land: odes/torord/ToRORd_dynCl_endo_Land.ode, 52
    states, 140 parameters): one GRL1 step vs the specification fixture for six parameter sets (cell types, stretch,
    stretch rate, troponin / tropomyosin exponents) at states on every branch of the mechanics part and along a paced
    action potential (1e-10 / 1e-9 relative to the state scale), and vs oracle/torord.py with per-node parameters."""
    from beat.models import torord_land as tl

    from oracle import torord as otor

    g = np.load(GOLD / "torord_land_spec.npz")
    m = tl.generalized_rush_larsen
    assert tuple(g["state_names"]) == m.state_names and tuple(g["parameter_names"]) == m.parameter_names
    np.testing.assert_array_equal(g["state_defaults"], tl.init_state_values())
    np.testing.assert_array_equal(g["parameter_defaults"], tl.init_parameter_values())
    assert m.num_states == 52 and m.num_parameters == 140 and tl.state_index("v") == 41 and tl.state_index("cai") == 44

    def err(out, ref):
        return np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6 * np.abs(g["state_defaults"])[:, None] + 1e-12)

    for k, P in enumerate(g["parameter_sets"]):
        out = m(states=g["states"], t=float(g["t"]), parameters=P, dt=float(g["dt"]))
        assert np.isfinite(out).all()
        e = err(out, g["grl1"][k])
        assert e.max() < 1e-10, (k, e.max(), g["state_names"][np.unravel_index(e.argmax(), e.shape)[0]])
    out = m(states=g["traj_states"], t=float(g["traj_step_t"]), parameters=g["parameter_defaults"], dt=float(g["traj_dt"]))
    assert err(out, g["traj_grl1"]).max() < 1e-9
    # per-node parameters
    S = g["traj_states"]
    n = S.shape[1]
    P = np.repeat(tl.init_parameter_values()[:, None], n, axis=1)
    P[tl.parameter_index("celltype")] = np.arange(n) % 3
    P[tl.parameter_index("i_Stim_Amplitude")] = np.where(np.arange(n) % 2, -53.0, 0.0)
    P[tl.parameter_index("lmbda")] = 0.8 + 0.5 * np.arange(n) / n
    P[tl.parameter_index("dLambda")] = 0.001 * ((np.arange(n) % 5) - 2)
    for t in (0.5, 7.0):
        out = m(states=S, t=t, parameters=P, dt=0.02)
        assert err(out, otor.torord_land_generalized_rush_larsen(S, t, 0.02, P)).max() < 1e-9


def test_torord_land_beat_follows_the_oracle_and_develops_tension():
    """One paced beat of the Land cell with the in-kernel loop: the electrophysiology is an ordinary ToR-ORd action
    potential, troponin binds calcium and the cross-bridge states rise and fall again (active tension
    Ta = h(lambda) Tref / rs (XS (Zetas + 1) + XW Zetaw), .ode:706, at lambda = 1); the first 300 steps equal the NumPy
    oracle's within 1e-8."""
    from beat.models import torord_land as tl

    from oracle import torord as otor

    m = tl.generalized_rush_larsen
    y0, P = tl.init_state_values(), tl.init_parameter_values()
    ix = [tl.state_index(k) for k in ("v", "cai", "CaTrpn", "XS", "XW", "TmB")]
    y, tr = m.run(y0, P, dt=0.02, nsteps=25000, nbeats=1, track_indices=ix, save_freq=50)
    v, cai, catrpn, xs = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
    assert np.isfinite(y).all() and 5.0 < v.max() < 60.0 and v[-1] < -80.0
    assert 2e-4 < cai.max() < 2e-3 and cai[-1] < 1.5e-4
    assert 0.05 < catrpn.max() < 0.9 and catrpn.argmax() > cai.argmax()      # troponin follows the calcium transient
    ta = 120.0 / 0.25 * xs                                                     # Zetas = Zetaw = 0 at dLambda = 0
    assert 5.0 < ta.max() < 150.0 and xs.argmax() > catrpn.argmax() and ta[-1] < 0.2 * ta.max()
    ys, tr2 = m.run(y0, P, dt=0.02, nsteps=300, nbeats=1, track_indices=ix, save_freq=300)
    yo = y0[:, None].copy()
    for i in range(300):
        yo = otor.torord_land_generalized_rush_larsen(yo, i * 0.02, 0.02, P)
    scale = np.maximum(np.abs(yo[:, 0]), 1e-6 * np.abs(y0) + 1e-12)
    assert (np.abs(ys.reshape(-1) - yo[:, 0]) / scale).max() < 1e-8


@pytest.mark.parametrize("model", ["tp06", "torord", "torord_land", "fhn"])
def test_run_kernel_equals_repeated_steps(model):
    """beat_ode_run (in-kernel time loop) gives what repeated beat_ode_step launches give: bit for bit for the models
    compiled without contraction; the TP06 and ToR-ORd steps are compiled with a * b + c contracted to fma (round 3: -5 %
    VALU instructions on kernels bound by them), which the compiler decides per kernel instantiation -- the register loop
    and the step kernel may then round an expression differently (seen: 3 of 3640 values one ulp apart after 25 steps)."""
    from beat.models import fhn, torord, torord_land, tp06

    m = {"tp06": tp06.generalized_rush_larsen, "torord": torord.generalized_rush_larsen,
         "torord_land": torord_land.generalized_rush_larsen, "fhn": fhn.forward_euler_readme}[model]
    rng = np.random.default_rng(4)
    y0 = np.repeat(m.init_state_values()[:, None], 70, axis=1)
    y0[m.state_index(m.v_name)] += rng.uniform(0, 20, 70)
    P = m.init_parameter_values()
    ys = y0.copy()
    for j in range(25):
        ys = m(states=ys, t=j * 0.02, parameters=P, dt=0.02)
    yr, tr = m.run(y0, P, dt=0.02, nsteps=25, track_indices=[m.state_index(m.v_name)], save_freq=5)
    if model == "fhn":
        np.testing.assert_array_equal(yr, ys)
    else:
        np.testing.assert_allclose(yr, ys, rtol=1e-13, atol=1e-300)
    assert tr.shape == (5, 1, 70)
    np.testing.assert_array_equal(tr[0, 0], y0[m.state_index(m.v_name)])


def test_torord_step_is_regular_through_the_ghk_singularity():
    """The Goldman-Hodgkin-Katz fluxes of the specification (ICaL, ICaNa, ICaK, ICab, INab) are 0/0 at v = 0 and their
    v-derivative loses all accuracy next to it; the generated kernel evaluates them at a potential kept 1e-4 mV away
    (csrc/ionic_models.h: beat_guard).  From plateau states with v within 1e-15 .. 1e-3 mV of 0 every state stays
    finite, and each increment lies on the smooth curve through the values at +-0.01 .. +-0.04 mV (literal
    evaluation) up to the guard's own footprint: inside the window the rates are evaluated up to 1e-4 mV off, i.e. an
    increment moves by at most |d increment / dv| * 1e-4 mV."""
    from beat.models import torord

    g = np.load(GOLD / "torord_spec.npz")
    vi = torord.state_index("v")
    traj = g["traj_states"]
    col = int(np.argmin(np.abs(traj[vi])))  # the sample of the paced action potential closest to 0 mV
    offsets = np.array([-0.04, -0.03, -0.02, -0.01, -1e-3, -0.9e-4, -1e-6, -1e-10, -1e-15, 0.0, 1e-15, 1e-10, 1e-6,
                        0.9e-4, 1e-3, 0.01, 0.02, 0.03, 0.04])
    S = np.repeat(traj[:, col : col + 1], len(offsets), axis=1)
    S[vi] = offsets
    P = torord.init_parameter_values()
    out = torord.generalized_rush_larsen(states=S, t=float(g["traj_step_t"]), parameters=P, dt=float(g["traj_dt"]))
    assert np.isfinite(out).all()
    inc = out - S
    far = np.abs(offsets) >= 0.01
    for k in range(S.shape[0]):
        coef = np.polyfit(offsets[far], inc[k, far], 3)
        fit = np.polyval(coef, offsets)
        scale = np.abs(inc[k]).max()
        slope = abs(coef[-2])  # d increment / dv at v = 0
        # (only the guarded inputs move, so it is their partial slope that counts: allow 3x the total slope)
        bound = 3.0 * slope * 1.0e-4 + 1e-6 * scale
        assert np.abs(inc[k] - fit).max() <= bound, (g["state_names"][k], np.abs(inc[k] - fit).max(), bound)


def test_torord_step_on_edge_case_states():
    """40 000 states from edge values -- potentials at and next to 0 mV (the singular point of the GHK fluxes), far
    hyper- and depolarised, gates at exactly 0 / 1 / 1e-300, concentrations scaled by 0.05 .. 10 -- for the three cell
    types and three step sizes: every output is finite.  The one exception is the specification's own: it divides by
    km2n = jca (ToRORd_dynCl_endo.ode, `anca_*`), so a jca gate of exactly 0 -- unreachable by its relaxation dynamics --
    is not generated here."""
    from beat.models import torord

    g = np.load(GOLD / "torord_spec.npz")
    names = list(g["state_names"])
    rng = np.random.default_rng(7)
    n = 40000
    base = g["traj_states"]
    S = base[:, rng.integers(0, base.shape[1], n)].copy()
    vi = torord.state_index("v")
    S[vi] = rng.choice([-120.0, -95.0, -88.0, -40.0, 0.0, 1e-13, -1e-13, 5e-5, 1e-3, 20.0, 60.0, 90.0], n) + rng.choice([0.0, 1e-9, 0.25], n)
    scaled = ("nai", "nass", "ki", "kss", "cai", "cass", "cansr", "cajsr", "cli", "clss", "CaMKt", "Jrel_np", "Jrel_p")
    for k, name in enumerate(names):
        if name == "v":
            continue
        if name in scaled:
            S[k] *= rng.choice([0.05, 0.5, 1.0, 2.0, 10.0], n)
        else:
            S[k] = rng.choice([1.0, 1e-300, 1e-9, 0.5, 1 - 1e-16] if name == "jca" else [0.0, 1.0, 1e-300, 1e-9, 0.5, 1 - 1e-16], n)
    for celltype in (0, 1, 2):
        P = torord.init_parameter_values(celltype=float(celltype))
        for dt in (0.05, 0.01, 0.5):
            out = torord.generalized_rush_larsen(states=S, t=0.0, parameters=P, dt=dt)
            bad = ~np.isfinite(out).all(axis=0)
            assert not bad.any(), (celltype, dt, int(bad.sum()), [names[r] for r in np.flatnonzero(~np.isfinite(out[:, np.flatnonzero(bad)[0]]))])


@pytest.mark.parametrize("model_name", ["torord", "tp06"])
def test_steps_stay_finite_and_right_at_unphysiological_potentials(model_name):
    """Potentials from -135 to +360 mV (a 2000 uA/cm^2 surface stimulus drives staircase corners of a voxel mesh to
    +230 mV, BASELINE configs[4]): gate rates reach 1e21 / ms there, so rate * dt leaves the range the kernels' exp()
    handles by itself -- those arguments are clamped (the result is 0 either way).  Every state finite and equal to the
    NumPy oracle (libm arithmetic), 1e-7 relative to the state scale (the Markov occupancies lose digits to cancellation
    up there)."""
    from beat.models import torord, tp06
    from oracle import ionic
    from oracle import torord as otor

    n = 400
    rng = np.random.default_rng(4)
    if model_name == "torord":
        model, vname = torord, "v"
        step = otor.torord_generalized_rush_larsen
        S = np.repeat(otor.torord_init_state_values()[:, None], n, axis=1)
        names = otor.TORORD_STATES
        gates = ["m", "h", "j", "hp", "jp", "d", "ff_", "fs", "a", "iF", "iS", "xs1", "xs2", "mL", "hL", "O_", "C1", "C2", "I_"]
        params = [otor.torord_init_parameter_values(celltype=float(c), i_Stim_Amplitude=0.0) for c in (0, 1, 2)]
    else:
        model, vname = tp06, "V"
        step = ionic.tp06_generalized_rush_larsen
        S = np.repeat(ionic.tp06_init_state_values()[:, None], n, axis=1)
        names = ionic.TP06_STATES
        gates = ["Xr1", "Xr2", "Xs", "m", "h", "j", "d", "f", "f2", "fCass", "s", "r"]
        params = [ionic.tp06_init_parameter_values(stim_amplitude=0.0)]
    S[names.index(vname)] = np.linspace(-135.0, 360.0, n)
    for gname in gates:
        S[names.index(gname)] = rng.uniform(0.0, 1.0, n)
    scale = np.abs(S[:, :1]) * 1e-6 + 1e-12
    for P in params:
        for dt in (0.05, 0.01):
            out = model.generalized_rush_larsen(states=S, t=5.0, parameters=P, dt=dt)
            assert np.isfinite(out).all(), [names[i] for i in np.unique(np.argwhere(~np.isfinite(out))[:, 0])]
            ref = step(S, 5.0, dt, P)
            ok = np.isfinite(ref)
            err = np.abs(out - ref) / np.maximum(np.abs(ref), scale)
            assert err[ok].max() < 1e-7, (model_name, dt, names[np.unravel_index(np.where(ok, err, 0).argmax(), err.shape)[0]])
