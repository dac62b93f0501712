"""GPU: the demo scripts under demos/ run end to end at small sizes and produce what their docstrings promise (they are
the callers either side of the split step: stand-alone diffusion, free-running cells with per-node parameters, a slab
with single-cell pre-pacing, pseudo-ECG and checkpoint, a pacing train over heterogeneous tissue)."""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DEMOS = Path(__file__).resolve().parents[1] / "demos"


def _demo(name):
    sys.path.insert(0, str(DEMOS))
    try:
        spec = importlib.util.spec_from_file_location(f"demo_{name}", DEMOS / f"{name}.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(str(DEMOS))
    return mod


def test_diffusion_demo_conserves_what_the_source_puts_in(capsys):
    total, expected = _demo("diffusion").main(["--n", "20", "--T", "2.5", "--dt", "0.1"])
    assert np.isclose(total, expected, rtol=2e-4) and expected > 0.0   # to the default tolerance of the linear solves
    assert "status OK" in capsys.readouterr().out


def test_simple_ode_demo_apd_shortens_with_g_ks():
    scale, apd = _demo("simple_ode").main(["--cells", "16", "--T", "420", "--dt", "0.05"])
    assert np.all(np.isfinite(apd)) and np.all(np.diff(apd) < 0.0)   # more I_Ks: earlier repolarisation
    assert 200.0 < apd[-1] < apd[0] < 400.0


def test_slab_ecg_demo(tmp_path):
    lead, v = _demo("slab_ecg").main(["--dx", "0.5", "--T", "12", "--beats", "1", "--out", str(tmp_path / "out")])
    assert len(lead) == 12 and np.all(np.isfinite(lead)) and np.abs(lead).max() > 1e-4
    assert v.max() > 0.0 and v.min() < -80.0                      # a front inside the slab
    assert (tmp_path / "out" / "prepacing").is_dir() and any((tmp_path / "out").glob("slab.bp*"))


def test_pace_train_demo():
    report = _demo("pace_train").main(["--dx", "0.5", "--s1", "2", "--bcl", "320"])
    for near, far in report:
        assert 0.0 < near[0] < 3.0 and 10.0 < far[0] < 40.0        # every S1 propagates to the far end
        assert far[1] < near[1]                                    # less I_CaL there: shorter action potential
    assert report[0][0][1] - report[0][1][1] > 10.0                # by tens of ms on the first, rested beat
    assert report[1][0][1] < report[0][0][1] - 30.0                # restitution: the beat 320 ms later is shorter


def test_pace_train_demo_with_the_reference_heterogeneity(capsys):
    """demos/pace_train.py --block: g_Kr = g_Ks = 0 in the right half of the cable, as the reference's demo sets them
    (demos/pace_train.py:133-167) -- the (P, N) parameters are recognised as two classes; without the two repolarising
    currents the far end stays depolarised far longer than the near end."""
    report = _demo("pace_train").main(["--dx", "0.5", "--s1", "1", "--bcl", "450", "--block"])
    assert "parameter classes: 2 uniform sets" in capsys.readouterr().out
    near, far = report[0]
    assert 0.0 < near[0] < 3.0 and 10.0 < far[0] < 40.0
    assert np.isfinite(near[1]) and (np.isnan(far[1]) or far[1] > near[1] + 50.0)


def test_ode_file_demo_runs_a_generated_model_on_a_slab(capsys):
    """demos/ode_file_slab.py: a cell model straight from an .ode file (beat.models.from_ode) -- pre-paced as a single cell in one
    launch (single_cell.get_steady_state -> beat_ode_run on the generated kernel), then on a slab through the fused split step: the
    wave started in one corner reaches the opposite one.  (Run in a fresh process too: the library used to be loaded ahead of
    PyTorch there -- two HIP runtimes in one process, "no ROCm-capable device" -- which no test that gets its context from the
    fixture could see.)"""
    import subprocess

    t_act, v_far, solver = _demo("ode_file_slab").main(["--dx", "0.25", "--T", "30", "--beats", "2"])
    out = capsys.readouterr().out
    assert "5 states, 14 parameters" in out and "status OK" in out
    assert t_act is not None and 10.0 < t_act < 30.0 and v_far.max() > 0.0
    assert solver.ode.on_device and solver.ode._dev.model.model_id >= 100
    fresh = subprocess.run([sys.executable, str(DEMOS / "ode_file_slab.py"), "--dx", "0.5", "--T", "5", "--beats", "1"],
                           capture_output=True, text=True, timeout=300)
    assert fresh.returncode == 0 and "status OK" in fresh.stdout, fresh.stderr[-2000:]
