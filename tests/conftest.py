"""pytest configuration: registers the ``gpu`` marker and puts the product package
(``fenicsx-beat_amd/beat``) and the repo root (for ``oracle``) on sys.path."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "fenicsx-beat_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ctx():
    """A device context; only usable in tests marked ``gpu``."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (run with -m 'not gpu' on CPU)")
    from beat._device import Context

    return Context.default()
