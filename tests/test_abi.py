"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
that include/beat_hip.h declares (no compute without a GPU)."""

import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / "include" / "beat_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(beat_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from beat import _hip

    assert _declared_symbols() == sorted(_hip.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from beat import _hip

    lib = _hip.load()  # raises if the .so is missing: there is no CPU fallback
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.beat_abi_version() == 1
    assert _hip.stencil_offsets()[0] == (0, 0, 0)


def test_stencil_offsets_match_host_tables():
    from beat import _hip, _stencil

    assert tuple(_hip.stencil_offsets()) == _stencil.OFFSETS


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from beat import _hip

    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "_LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_hip.BeatHipError):
        _hip.load()


def test_no_gpu_context_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from beat import _hip
    from beat._device import Context

    with pytest.raises(_hip.BeatHipError):
        Context()
