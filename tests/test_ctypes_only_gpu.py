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
