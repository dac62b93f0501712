"""GPU: the C ABI used the way a foreign host would use it -- bare ctypes, device memory from beat_malloc /
beat_memcpy_*, no torch and no beat package in the process (tests/_ctypes_only_script.py)."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def test_tp06_split_steps_from_bare_ctypes_match_oracle():
    root = Path(__file__).resolve().parents[1]
    run = subprocess.run([sys.executable, str(root / "tests" / "_ctypes_only_script.py")], capture_output=True, text=True,
                         timeout=300, cwd=root)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    assert "ctypes-only ok" in run.stdout


def test_padded_work_fields_change_nothing(tmp_path):
    """beat_pde_field_stride pads the work fields and the guess's history on big grids only; BEAT_FIELD_SKEW forces a padding on
    a small one: the bare-ctypes host (which sizes its work area from beat_pde_field_stride) still matches the oracle, and the
    splitting solver of the public API gives the same bits with and without."""
    import os

    import numpy as np

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, BEAT_FIELD_SKEW="1056")
    run = subprocess.run([sys.executable, str(root / "tests" / "_ctypes_only_script.py")], capture_output=True, text=True,
                         timeout=300, cwd=root, env=env)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    assert "ctypes-only ok" in run.stdout
    got = {}
    for skew in ("0", "1056"):
        out = tmp_path / skew
        out.mkdir()
        env = dict(os.environ, BEAT_FIELD_SKEW=skew, WORLD_SIZE="1")
        run = subprocess.run([sys.executable, str(root / "tests" / "_ode_space_ranks_script.py"), str(out), "P_1"],
                             capture_output=True, text=True, timeout=300, cwd=root, env=env)
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
        got[skew] = np.load(out / "rank0.npz")
    assert np.array_equal(got["0"]["v"], got["1056"]["v"]) and np.array_equal(got["0"]["s"], got["1056"]["s"])
    assert int(got["0"]["its"]) == int(got["1056"]["its"]) > 0
