"""CPU: the hand-organised ToR-ORd-dynCl step (fenicsx-beat_amd/csrc/torord_dyncl.h -- the very source the HIP kernel
compiles) built for the host with g++ (tests/torord_host_harness.cpp: libm exp / log / division in place of the device's
table-driven ones) and checked against

* the golden fixture generated from the reference's ``.ode`` specification (tests/golden/torord_spec.npz): one GRL1 step
  for endo / epi / mid and along a paced action potential, 1e-11 relative to the state scale;
* the independent NumPy oracle (oracle/torord.py) on per-node parameters and over a 400-step trajectory.

The Land instance of the same source (52 states: ToRORd_dynCl_endo_Land.ode) is checked the same way against
tests/golden/torord_land_spec.npz and the oracle.

The arithmetic organisation of the kernel (blocks, running sums, sparse dual numbers) is therefore verified without a
GPU; the GPU suite then checks the device build of the same source."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import torord

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"


def _host_step(tmp_path_factory, flags):
    if shutil.which("g++") is None:
        pytest.skip("no g++ on this machine")
    d = tmp_path_factory.mktemp("torord_host")
    exe = d / "torord_host"
    subprocess.run(["g++", "-O2", "-std=c++17", *flags, "-o", str(exe), str(ROOT / "tests" / "torord_host_harness.cpp")],
                   check=True)

    def run(S, P, t, dt):
        S = np.ascontiguousarray(S, dtype=np.float64)
        np.ascontiguousarray(S).tofile(d / "s.bin")
        np.ascontiguousarray(P, dtype=np.float64).tofile(d / "p.bin")
        subprocess.run([str(exe), str(d / "s.bin"), str(d / "p.bin"), str(d / "o.bin"), str(S.shape[1]), repr(float(t)),
                        repr(float(dt))], check=True)
        return np.fromfile(d / "o.bin").reshape(S.shape)

    return run


@pytest.fixture(scope="module")
def host_step(tmp_path_factory):
    return _host_step(tmp_path_factory, [])


@pytest.fixture(scope="module")
def host_step_land(tmp_path_factory):
    return _host_step(tmp_path_factory, ["-DBEAT_HOST_LAND=1"])


def _err(out, ref, defaults):
    return np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6 * np.abs(defaults)[:, None] + 1e-12)


def test_hand_kernel_source_matches_the_ode_spec_fixture(host_step):
    g = np.load(GOLD / "torord_spec.npz")
    for celltype in (0, 1, 2):
        P = torord.torord_init_parameter_values(celltype=float(celltype))
        out = host_step(g["states"], P, float(g["t"]), float(g["dt"]))
        assert _err(out, g[f"grl1_celltype{celltype}"], g["state_defaults"]).max() < 1e-11
    out = host_step(g["traj_states"], torord.torord_init_parameter_values(), float(g["traj_step_t"]), float(g["traj_dt"]))
    assert _err(out, g["traj_grl1"], g["state_defaults"]).max() < 1e-11


def test_hand_kernel_source_matches_the_numpy_oracle(host_step):
    """Per-node parameters (a different cell type and stimulus per node, inside and outside the model's own stimulus
    window) and a 400-step trajectory through the upstroke, against oracle/torord.py."""
    g = np.load(GOLD / "torord_spec.npz")
    S = g["traj_states"]
    n = S.shape[1]
    P = np.repeat(torord.torord_init_parameter_values()[:, None], n, axis=1)
    P[torord.torord_parameter_index("celltype")] = np.arange(n) % 3
    P[torord.torord_parameter_index("i_Stim_Amplitude")] = np.where(np.arange(n) % 2, -53.0, 0.0)
    for t in (0.5, 7.0):  # inside / outside the 1 ms stimulus window
        out = host_step(S, P, t, 0.02)
        ref = torord.torord_generalized_rush_larsen(S, t, 0.02, P)
        assert _err(out, ref, g["state_defaults"]).max() < 1e-11
    y = torord.torord_init_state_values()[:, None].copy()
    yo = y.copy()
    P1 = torord.torord_init_parameter_values()
    for i in range(400):
        y = host_step(y, P1, i * 0.01, 0.01)
        yo = torord.torord_generalized_rush_larsen(yo, i * 0.01, 0.01, P1)
    assert yo[torord.torord_state_index("v"), 0] > 0.0  # fired
    assert _err(y, yo, g["state_defaults"]).max() < 1e-9


def test_land_instance_matches_the_ode_spec_fixture(host_step_land):
    """Six parameter sets (three cell types; stretched / lengthening / shortening cells with other troponin and
    tropomyosin exponents) at states that take every branch of the mechanics part, and one step from every state of a
    paced action potential."""
    g = np.load(GOLD / "torord_land_spec.npz")
    for k, P in enumerate(g["parameter_sets"]):
        out = host_step_land(g["states"], P, float(g["t"]), float(g["dt"]))
        assert _err(out, g["grl1"][k], g["state_defaults"]).max() < 1e-11, k
    out = host_step_land(g["traj_states"], g["parameter_defaults"], float(g["traj_step_t"]), float(g["traj_dt"]))
    assert _err(out, g["traj_grl1"], g["state_defaults"]).max() < 1e-11


def test_land_instance_matches_the_numpy_oracle(host_step_land):
    """Per-node parameters (cell type, stimulus and stretch differ from node to node) and a 400-step trajectory
    through the upstroke and the start of the calcium transient, against oracle/torord.py."""
    g = np.load(GOLD / "torord_land_spec.npz")
    S = g["traj_states"]
    n = S.shape[1]
    names = list(torord.TORORD_LAND_PARAMETERS)
    P = np.repeat(torord.torord_land_init_parameter_values()[:, None], n, axis=1)
    P[names.index("celltype")] = np.arange(n) % 3
    P[names.index("i_Stim_Amplitude")] = np.where(np.arange(n) % 2, -53.0, 0.0)
    P[names.index("lmbda")] = 0.8 + 0.5 * np.arange(n) / n
    P[names.index("dLambda")] = 0.001 * ((np.arange(n) % 5) - 2)
    for t in (0.5, 7.0):
        out = host_step_land(S, P, t, 0.02)
        ref = torord.torord_land_generalized_rush_larsen(S, t, 0.02, P)
        assert _err(out, ref, g["state_defaults"]).max() < 1e-11
    y = torord.torord_land_init_state_values()[:, None].copy()
    yo = y.copy()
    P1 = torord.torord_land_init_parameter_values()
    for i in range(400):
        y = host_step_land(y, P1, i * 0.01, 0.01)
        yo = torord.torord_land_generalized_rush_larsen(yo, i * 0.01, 0.01, P1)
    assert yo[torord.TORORD_LAND_STATES.index("v"), 0] > 0.0
    assert _err(y, yo, g["state_defaults"]).max() < 1e-9


def _perturbed_parameters(P0, names, n, seed):
    """Every parameter moved per node: non-zero ones scaled by 0.9 .. 1.1, the voltage shifts and offsets that default to
    zero (EKshift, vShift, offset) set to a few mV / ms; switches (celltype, mode, isacs) and the stimulus protocol cycled."""
    rng = np.random.default_rng(seed)
    P = np.repeat(P0[:, None], n, axis=1) * rng.uniform(0.9, 1.1, (len(P0), n))
    for name, lo, hi in (("EKshift", -4.0, 4.0), ("vShift", -3.0, 3.0), ("offset", 0.0, 0.5)):
        P[names.index(name)] = rng.uniform(lo, hi, n)
    for name in ("celltype", "mode", "isacs"):
        if name in names:
            P[names.index(name)] = P0[names.index(name)]
    P[names.index("celltype")] = np.arange(n) % 3
    for name in ("i_Stim_Start", "i_Stim_End", "i_Stim_Period", "i_Stim_PulseDuration"):
        P[names.index(name)] = P0[names.index(name)]
    return P


def test_every_parameter_reaches_the_step_as_the_specification_has_it(host_step, host_step_land):
    """The kernel forms products, quotients and reciprocals of parameters once per parameter set (Derived) and shares
    exponentials whose arguments differ by a parameter-dependent constant (exp((v + EKshift + 70)/20) = exp(v/20) *
    const): with ALL parameters perturbed per node -- the shifts that default to zero included -- one step from the
    action-potential samples still equals the NumPy oracle, which evaluates the specification literally."""
    g = np.load(GOLD / "torord_spec.npz")
    S = g["traj_states"]
    n = S.shape[1]
    P = _perturbed_parameters(torord.torord_init_parameter_values(), list(torord.TORORD_PARAMETERS), n, 5)
    out = host_step(S, P, 0.4, 0.01)
    ref = torord.torord_generalized_rush_larsen(S, 0.4, 0.01, P)
    assert _err(out, ref, g["state_defaults"]).max() < 1e-10
    gl = np.load(GOLD / "torord_land_spec.npz")
    Sl = gl["traj_states"]
    Pl = _perturbed_parameters(torord.torord_land_init_parameter_values(), list(torord.TORORD_LAND_PARAMETERS), Sl.shape[1], 6)
    outl = host_step_land(Sl, Pl, 0.4, 0.01)
    refl = torord.torord_land_generalized_rush_larsen(Sl, 0.4, 0.01, Pl)
    assert _err(outl, refl, gl["state_defaults"]).max() < 1e-10


def test_sarcoplasmic_load_at_zero_and_below(host_step, host_step_land):
    """cajsr enters the release equations through 1/cajsr (ToRORd_dynCl_endo.ode: Jrel_inf, tau_rel = max(bt/(1 + 0.0123/cajsr),
    0.001)).  Perturbed parameter sets drive it through zero and below for a while (tools/soak_cells.py); the
    specification stays finite there -- its max() returns the floor when the quotient turns negative -- and the kernel,
    which forms 1/tau_rel from one shared reciprocal, must do what the specification does: one step from the
    action-potential samples with cajsr set to values around zero equals the NumPy oracle."""
    import warnings

    g = np.load(GOLD / "torord_spec.npz")
    S = g["traj_states"].copy()
    n = S.shape[1]
    vals = np.array([-0.5, -0.1, -0.02, -0.012, -1e-3, -3.7e-4, -1e-9, 1e-12, 1e-9, 1.8e-5, 1e-3, 0.05])
    S[torord.torord_state_index("cajsr")] = vals[np.arange(n) % len(vals)]
    P = torord.torord_init_parameter_values()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = torord.torord_generalized_rush_larsen(S, 0.4, 0.01, P)
    out = host_step(S, P, 0.4, 0.01)
    assert np.isfinite(out).all() and np.isfinite(ref).all()
    assert _err(out, ref, g["state_defaults"]).max() < 1e-9
    gl = np.load(GOLD / "torord_land_spec.npz")
    Sl = gl["traj_states"].copy()
    Sl[list(torord.TORORD_LAND_STATES).index("cajsr")] = vals[np.arange(Sl.shape[1]) % len(vals)]
    Pl = torord.torord_land_init_parameter_values()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        refl = torord.torord_land_generalized_rush_larsen(Sl, 0.4, 0.01, Pl)
    outl = host_step_land(Sl, Pl, 0.4, 0.01)
    assert _err(outl, refl, gl["state_defaults"]).max() < 1e-9


def test_unphysiological_states_agree_wherever_the_specification_is_finite(host_step):
    """Differential run on 6000 states nobody should reach -- potentials in -135 .. 100 mV, gates in -0.1 .. 1.1, every
    concentration, load and release flux scaled by -1 .. 10 (negative ones included) -- wherever the NumPy oracle, which
    evaluates the specification literally, returns finite values (logarithms of negative quotients do not), the kernel
    source returns the same ones: its rewrites (shared reciprocals, rates s/(c s + 1), cajsr^8/(half^8 + cajsr^8), ...)
    assume no sign."""
    import warnings

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
    P = torord.torord_init_parameter_values()
    for dt in (0.01, 0.05):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = torord.torord_generalized_rush_larsen(S, 0.3, dt, P)
        out = host_step(S, P, 0.3, dt)
        cols = np.isfinite(ref).all(axis=0)
        assert cols.sum() > n // 3
        assert np.isfinite(out[:, cols]).all()
        assert _err(out[:, cols], ref[:, cols], g["state_defaults"]).max() < 1e-7
