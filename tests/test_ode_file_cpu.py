"""CPU: ``beat.models.from_ode`` -- a device cell model generated from a gotran ``.ode`` file (the reference takes any
gotranx-generated ``fun``, /root/reference/demos/niederer_benchmark.py:82-99).  Without a GPU: the generated NumPy evaluation
against an INDEPENDENT evaluation of the same file (tests/golden/ode_spec.py: assignments evaluated numerically, total
self-derivatives by SymPy without common-subexpression elimination), the generated C++ compiled for gfx950 (hipcc cross-compiles
here), and -- in the build container only, where the reference's files are -- the generated TP06 against the committed fixture
tests/golden/tp06_spec.npz."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
CSRC = ROOT / "fenicsx-beat_amd" / "csrc"
SMALL = ROOT / "tests" / "data" / "small_cell.ode"
HIPCC = os.environ.get("BEAT_HIPCC") or ("/opt/rocm/bin/hipcc" if Path("/opt/rocm/bin/hipcc").exists() else shutil.which("hipcc"))
REF_TP06 = Path("/root/reference/odes/tentusscher_panfilov_2006/tentusscher_panfilov_2006_epi_cell.ode")


def _states(model, n, seed):
    rng = np.random.default_rng(seed)
    y = np.repeat(model.init_state_values()[:, None], n, axis=1)
    y[model.state_index("V")] = rng.uniform(-95.0, 45.0, n)
    for g in ("m", "h", "n"):
        y[model.state_index(g)] = rng.uniform(0.0, 1.0, n)
    y[model.state_index("ca")] = 10.0 ** rng.uniform(-4.5, -2.5, n)
    return y


def test_generated_numpy_step_equals_an_independent_evaluation_of_the_file():
    from ode_spec import OdeSpec

    from beat.models import from_ode

    model = from_ode(SMALL)
    assert model.state_names == ("V", "m", "h", "n", "ca") and model.num_parameters == 14 and model.v_name == "V"
    assert model.state_index("ca") == 4 and model.parameter_index("g_out") == 2
    np.testing.assert_array_equal(model.init_state_values(V=-80.0), [-80.0, 0.002, 0.98, 0.01, 0.0001])
    spec = OdeSpec(SMALL)
    y = _states(model, 400, 3)
    p = model.init_parameter_values(stim_amplitude=30.0)
    for t in (0.2, 0.9):  # outside and inside the stimulus window
        new = model.numpy_step(y, t, p, 0.02)
        _, _, ref = spec.grl1(dict(zip(model.state_names, y)), dict(zip(model.parameter_names, p)), t, 0.02)
        ref = np.array([np.broadcast_to(ref[s], y[0].shape) for s in model.state_names])
        assert np.isfinite(new).all()
        assert (np.abs(new - ref) / np.maximum(np.abs(ref), 1e-12)).max() < 1e-11
    fe = from_ode(SMALL, scheme="forward_euler")
    vals = spec.evaluate(dict(zip(model.state_names, y)), dict(zip(model.parameter_names, p)), 0.9)
    rhs = np.array([np.broadcast_to(vals[f"d{s}_dt"], y[0].shape) for s in model.state_names])
    np.testing.assert_allclose(fe.numpy_step(y, 0.9, p, 0.02), y + 0.02 * rhs, rtol=1e-13, atol=1e-300)
    with pytest.raises(ValueError):
        from_ode(SMALL, scheme="rk4")
    with pytest.raises(KeyError):
        from_ode(SMALL, v_name="Vm")


@pytest.mark.skipif(HIPCC is None, reason="no hipcc")
def test_generated_model_compiles_for_gfx950(tmp_path):
    from beat.models import from_ode

    model = from_ode(SMALL)
    assert f"struct {model.cxx_name}" in model.source and "io.store(4," in model.source
    # (uniform parameters without / with a pending update; all per-node rows + pending; parameter classes + pending: the
    # instantiations csrc/beat_ode_jit.hip writes for a registered model -- and the in-kernel time loop, checked below)
    for pend in ("false", "true", "true, per_node", "true, classes"):
        flags = {"false": "false, false", "true": "false, true", "true, per_node": "true, true", "true, classes": "false, true, true"}[pend]
        tag = pend.replace(", ", "_")
        unit = tmp_path / f"unit_{tag}.hip"
        unit.write_text('#include "beat_ode_kernel.h"\n' + model.source +
                        f"\ntemplate __global__ void ode_step_kernel<{model.cxx_name}, {flags}>(\n    double*, int64_t, int64_t, "
                        f"ParamPack<{model.cxx_name}::NP>, typename {model.cxx_name}::Derived, const double*, int64_t, double, double, int, "
                        "double*, PendingV, MarkedArgs, SparseRows);\n")
        out = tmp_path / f"unit_{tag}.hsaco"
        run = subprocess.run([HIPCC, "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DBEAT_ODE_WAVES=3",
                              "-mllvm", "-disable-machine-licm", "-w", f"-I{CSRC}", str(unit), "-o", str(out)],
                             capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stderr[-3000:]
        names = {s for s in out.read_bytes().split(b"\0") if s.startswith(b"_Z") and b"ode_step_kernel" in s and b"." not in s}
        assert len(names) == 1, names
    unit = tmp_path / "unit_run.hip"
    unit.write_text('#include "beat_ode_kernel.h"\n' + model.source +
                    f"\ntemplate __global__ void ode_run_kernel<{model.cxx_name}, true>(\n    double*, int64_t, int64_t, ParamPack<{model.cxx_name}::NP>, "
                    f"typename {model.cxx_name}::Derived, const double*, int64_t, double, double, int64_t, int, int, TrackSpec, double*);\n")
    out = tmp_path / "unit_run.hsaco"
    run = subprocess.run([HIPCC, "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DBEAT_ODE_WAVES=3",
                          "-mllvm", "-disable-machine-licm", "-w", f"-I{CSRC}", str(unit), "-o", str(out)],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    names = {s for s in out.read_bytes().split(b"\0") if s.startswith(b"_Z") and b"ode_run_kernel" in s and b"." not in s}
    assert len(names) == 1, names


def test_gotranx_scheme_names_are_understood():
    from beat.models import from_ode

    assert from_ode(SMALL, scheme="forward_explicit_euler").scheme == "forward_euler"
    assert from_ode(SMALL, scheme="forward_generalized_rush_larsen").scheme == "generalized_rush_larsen"
    with pytest.raises(ValueError):
        from_ode(SMALL, scheme="hybrid_rush_larsen")


def test_generated_numpy_step_takes_per_node_parameters():
    """(P, N) parameters through the NumPy twin = the (P,) evaluation column by column (the checker of the GPU suite's per-node
    and class routes must itself be right)."""
    from beat.models import from_ode

    model = from_ode(SMALL)
    rng = np.random.default_rng(2)
    n = 40
    y = np.repeat(model.init_state_values()[:, None], n, axis=1)
    y[model.state_index("V")] = rng.uniform(-90.0, 40.0, n)
    p = np.repeat(model.init_parameter_values(stim_amplitude=20.0)[:, None], n, axis=1)
    p[model.parameter_index("g_in")] *= rng.uniform(0.5, 1.5, n)
    p[model.parameter_index("E_out")] += rng.uniform(-5.0, 5.0, n)
    whole = model.numpy_step(y, 0.7, p, 0.02)
    for j in range(n):
        np.testing.assert_allclose(whole[:, j], model.numpy_step(y[:, j], 0.7, p[:, j], 0.02), rtol=1e-14, atol=0.0)
    with pytest.raises(ValueError):
        model.numpy_step(y, 0.7, p[:, :5], 0.02)


@pytest.mark.skipif(not REF_TP06.is_file(), reason="the reference's .ode files are only in the build container")
def test_generated_tp06_reproduces_the_committed_fixture():
    """The reference's own TP06 file through the generator: right-hand sides, total self-derivatives and GRL1 steps of
    tests/golden/tp06_spec.npz (96 states along an action potential) from the generated NumPy evaluation."""
    from beat.models import from_ode

    g = np.load(ROOT / "tests" / "golden" / "tp06_spec.npz")
    model = from_ode(REF_TP06)
    assert list(model.state_names) == list(g["state_names"]) and list(model.parameter_names) == list(g["parameter_names"])
    new = model.numpy_step(g["states"], float(g["t"]), g["parameter_defaults"], float(g["dt"]))
    ref = g["grl1_total"]
    assert (np.abs(new - ref) / np.maximum(np.abs(ref), 1e-9)).max() < 1e-10


REF_TORORD = Path("/root/reference/odes/torord/ToRORd_dynCl_endo.ode")
REF_LAND = Path("/root/reference/odes/torord/ToRORd_dynCl_endo_Land.ode")


@pytest.mark.skipif(not REF_TORORD.is_file(), reason="the reference's .ode files are only in the build container")
def test_generated_torord_reproduces_the_committed_fixtures():
    """The reference's two ToR-ORd files (45 and 52 states; demos/biv_endocardial.py:124-173 generates its ``fun`` from the first)
    through the generator: GRL1 steps of tests/golden/torord_spec.npz (48 states along an action potential, the three cell types)
    and torord_land_spec.npz (six parameter sets) from the generated NumPy evaluation to 1e-11.  The Land file multiplies
    comparisons into its arithmetic (``Gt(Zetas, 0)*Zetas``) and has conditions on parameters alone next to conditions on states:
    both are what _ComparisonsAsNumbers and the broadcast parameters in numpy_step are for."""
    from beat.models import from_ode

    g = np.load(ROOT / "tests" / "golden" / "torord_spec.npz")
    model = from_ode(REF_TORORD)
    assert list(model.state_names) == list(g["state_names"]) and list(model.parameter_names) == list(g["parameter_names"])
    for ct in (0, 1, 2):
        p = g["parameter_defaults"].copy()
        p[model.parameter_index("celltype")] = ct
        new = model.numpy_step(g["states"], float(g["t"]), p, float(g["dt"]))
        ref = g[f"grl1_celltype{ct}"]
        assert (np.abs(new - ref) / np.maximum(np.abs(ref), 1e-9)).max() < 1e-11, ct
    g = np.load(ROOT / "tests" / "golden" / "torord_land_spec.npz")
    model = from_ode(REF_LAND)
    assert list(model.state_names) == list(g["state_names"]) and list(model.parameter_names) == list(g["parameter_names"])
    for k, p in enumerate(g["parameter_sets"]):
        new = model.numpy_step(g["states"], float(g["t"]), p, float(g["dt"]))
        ref = g["grl1"][k]
        assert (np.abs(new - ref) / np.maximum(np.abs(ref), 1e-9)).max() < 1e-11, k


def test_ode_expressions_are_data_not_code(tmp_path):
    """An .ode file's right-hand sides are evaluated to build expression trees: only arithmetic, comparisons and calls of the
    functions such a file may use get that far -- attribute access, subscripts, lambdas, unknown calls, keyword arguments and
    strings are refused with the line they stand on."""
    from beat.models import from_ode

    text = SMALL.read_text()
    for bad in ("x_bad = ().__class__", "x_bad = [1, 2][0]", "x_bad = (lambda: 1)()", "x_bad = open('/etc/passwd')",
                "x_bad = exp(V, evaluate=False)", "x_bad = 'a'"):
        f = tmp_path / "bad.ode"
        f.write_text(text + "\n" + bad + "\n")
        with pytest.raises(ValueError, match="bad.ode"):
            from_ode(f)
    # a comparison standing in arithmetic is a number (gotran): an infix one and a function-style one
    f = tmp_path / "ind.ode"
    f.write_text(text.replace("dn_dt =", "gate_on = (V > -60)*1.0 + Lt(V, -100)*2.0\ndn_dt = 0*gate_on +"))
    m = from_ode(f)
    assert m.num_states == 5 and np.isfinite(m.numpy_step(m.init_state_values(), 0.0, m.init_parameter_values(), 0.01)).all()
