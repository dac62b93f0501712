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


def pytest_sessionstart(session):
    """A checkout without the built artefacts (libbeat_hip.so is not tracked): build them once with the driver's
    own entry point.  The product itself never does this -- beat._hip.load() raises when the library is missing."""
    from beat import _hip

    if not _hip.library_path().is_file():
        import __graft_entry__ as entry

        try:
            entry.build()
        except Exception as exc:  # the tests that need the library will say so
            print(f"could not build libbeat_hip.so: {exc}", file=sys.stderr)


@pytest.fixture(scope="session")
def hip_ctx():
    """A device context; only usable in tests marked ``gpu``."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (run with -m 'not gpu' on CPU)")
    from beat._device import Context

    return Context.default()
