"""CPU: the soak of tests/test_soak_gpu.py on the HOST builds of the kernel sources (tests/torord_host_harness.cpp,
tests/tp06_host_harness.cpp: the very csrc/torord_dyncl.h and csrc/ionic_models.h the HIP kernels compile), plain and under
AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on this pool): 24 cells of the 512-cell
population -- every parameter +-10 %, cell types cycled, the seven cells the pre-fix ToR-ORd kernel lost included -- paced
for three beats of 1000 ms at dt = 0.05 in the harness's time loop; all states finite, gates in [0, 1], concentrations
positive; and 100 further steps of 8 cells equal the NumPy oracle's to 1e-8."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

import _soak
from oracle import ionic, torord

ROOT = Path(__file__).resolve().parents[1]
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


def _build(tmp, source, flags, name):
    if shutil.which("g++") is None:
        pytest.skip("no g++ on this machine")
    exe = tmp / name
    inc = ["-I/opt/rocm/include"] if source.startswith("tp06") else []
    subprocess.run(["g++", "-std=c++17", *flags, *inc, "-o", str(exe), str(ROOT / "tests" / source)], check=True)

    def run(S, P, t, dt, nsteps=1, per_beat=None):
        S = np.ascontiguousarray(S, dtype=np.float64)
        S.tofile(tmp / "s.bin")
        np.ascontiguousarray(P, dtype=np.float64).tofile(tmp / "p.bin")
        res = subprocess.run([str(exe), str(tmp / "s.bin"), str(tmp / "p.bin"), str(tmp / "o.bin"), str(S.shape[1]), repr(float(t)),
                              repr(float(dt)), str(int(nsteps)), str(int(per_beat or nsteps))], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-3000:]
        return np.fromfile(tmp / "o.bin").reshape(S.shape)

    return run


MODELS = {
    "tp06": ("tp06_host_harness.cpp", [], ionic.tp06_init_state_values, ionic.tp06_init_parameter_values,
             lambda k: list(ionic.TP06_STATES).index(k), lambda k: list(ionic.TP06_PARAMETERS).index(k), ionic.tp06_generalized_rush_larsen),
    "torord": ("torord_host_harness.cpp", [], torord.torord_init_state_values, torord.torord_init_parameter_values,
               torord.torord_state_index, lambda k: list(torord.TORORD_PARAMETERS).index(k), torord.torord_generalized_rush_larsen),
    "torord_land": ("torord_host_harness.cpp", ["-DBEAT_HOST_LAND=1"], torord.torord_land_init_state_values,
                    torord.torord_land_init_parameter_values, lambda k: list(torord.TORORD_LAND_STATES).index(k),
                    lambda k: list(torord.TORORD_LAND_PARAMETERS).index(k), torord.torord_land_generalized_rush_larsen),
}


@pytest.mark.parametrize("name", ["tp06", "torord", "torord_land"])
def test_host_build_of_the_kernel_source_survives_the_soak(tmp_path, name):
    source, flags, init_states, init_params, sidx, pidx, oracle_step = MODELS[name]
    plain = _build(tmp_path, source, ["-O2", *flags], name + "_plain")
    san = _build(tmp_path, source, [*SAN, *flags], name + "_san")
    dt, per_beat = 0.05, 20000
    cells = _soak.subset(512, 24)
    P = _soak.population(name, init_params(), pidx, 512)[:, cells]
    y0 = np.repeat(init_states()[:, None], len(cells), axis=1)
    y3 = plain(y0, P, 0.0, dt, nsteps=3 * per_beat, per_beat=per_beat)
    _soak.check_physical(name, y3, sidx, " after three beats (host build)")
    # the sanitizer build over the part of the run where the pre-fix kernel went wrong (the third beat), from the plain
    # build's state after two: same bits, no report
    y2 = plain(y0, P, 0.0, dt, nsteps=2 * per_beat, per_beat=per_beat)
    y3s = san(y2[:, :8], P[:, :8], 0.0, dt, nsteps=per_beat, per_beat=per_beat)
    np.testing.assert_allclose(y3s, y3[:, :8], rtol=1e-9, atol=1e-300)
    # ... and the oracle from there
    yh = plain(y3[:, :8], P[:, :8], 0.0, dt, nsteps=100, per_beat=100)
    yo = y3[:, :8].copy()
    for j in range(100):
        yo = oracle_step(yo, j * dt, dt, P[:, :8])
    assert _soak.relative_difference(yh, yo, init_states()).max() < 1e-8
