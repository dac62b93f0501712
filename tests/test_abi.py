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


def test_model_ids_and_sizes_agree_between_header_binding_library_and_model_modules():
    """include/beat_hip.h's BEAT_MODEL_* constants = beat._hip's MODEL_* = what beat_ode_model_info (host-only, no
    GPU) reports for each id = the state / parameter counts of the model modules' defaults."""
    import ctypes as C

    from beat import _hip
    from beat.models import fhn, simple, torord, torord_land, tp06

    text = (ROOT / "include" / "beat_hip.h").read_text()
    header = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+BEAT_MODEL_(\w+)\s+(\d+)", text)}
    binding = {k[len("MODEL_"):]: v for k, v in vars(_hip).items() if k.startswith("MODEL_") and isinstance(v, int)}
    assert header == binding and len(set(header.values())) == len(header)
    lib = _hip.load()
    models = {"TP06_GRL1": tp06.generalized_rush_larsen, "TORORD_DYNCL_GRL1": torord.generalized_rush_larsen,
              "TORORD_LAND_GRL1": torord_land.generalized_rush_larsen, "FHN_README": fhn.forward_euler_readme,
              "FHN_DEMO": fhn.forward_euler_demo, "SIMPLE_ODE": simple.forward_euler}
    assert set(models) == set(header)
    for name, model in models.items():
        ns, npar = C.c_int(), C.c_int()
        _hip.check(lib.beat_ode_model_info(header[name], C.byref(ns), C.byref(npar)))
        assert model.model_id == header[name], name
        assert (ns.value, npar.value) == (model.num_states, model.num_parameters), name
    with pytest.raises(_hip.BeatHipError):
        _hip.check(lib.beat_ode_model_info(99, None, None))
