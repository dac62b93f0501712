"""CPU: the run-time compilation behind beat_ode_step_rows (csrc/beat_ode_jit.h) as far as it goes without a GPU -- the library finds
its kernel sources, a compiler and a cache directory, and the translation unit it would write for an instance compiles for gfx950
(hipcc cross-compiles here) into a code object that holds exactly that kernel."""
import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "fenicsx-beat_amd" / "csrc"
HIPCC = os.environ.get("BEAT_HIPCC") or ("/opt/rocm/bin/hipcc" if Path("/opt/rocm/bin/hipcc").exists() else shutil.which("hipcc"))


def test_library_finds_sources_compiler_and_cache_directory(tmp_path, monkeypatch):
    import sys

    sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))
    from beat import _hip

    monkeypatch.setenv("BEAT_JIT_CACHE", str(tmp_path / "cache"))
    lib = _hip.load()
    out = (C.c_longlong * 4)()
    assert lib.beat_ode_jit_stats(out) == 1  # csrc/ and include/ beside the package, hipcc, a writable cache directory
    assert list(out) == [0, 0, 0, 0]


@pytest.mark.skipif(HIPCC is None, reason="no hipcc")
@pytest.mark.parametrize("model,pend,idx,mask", [("Tp06Grl1", "true", (6, 2), (0xA0, 0)), ("TorordLandGrl1", "false", (100,), (0, 0)),
                                                 ("Tp06Grl1", "true", (6, 2, 3, 4, 8, 1, 7, 18), (0xA0, 0))])
def test_generated_unit_compiles_for_gfx950(tmp_path, model, pend, idx, mask):
    # (the indices of the varying rows as a pack: up to 16 of them since round 5)
    inst = f"ode_step_kernel<{model}, true, {pend}, false, true, IdxPack<{', '.join(map(str, idx))}>, {mask[0]:#x}ull, {mask[1]:#x}ull>"
    src = tmp_path / "unit.hip"
    src.write_text('#include "beat_ode_kernel.h"\n'
                   f"template __global__ void {inst}(\n    double*, int64_t, int64_t, ParamPack<{model}::NP>, typename {model}::Derived, "
                   "const double*, int64_t, double, double, int, double*, PendingV, MarkedArgs, SparseRows);\n")
    out = tmp_path / "unit.hsaco"
    # the flags of csrc/beat_ode_jit.hip (kFlags) = those of the library's own build of beat_ode.hip
    run = subprocess.run([HIPCC, "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DBEAT_ODE_WAVES=3",
                          "-mllvm", "-disable-machine-licm", "-w", f"-I{CSRC}", str(src), "-o", str(out)],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    blob = out.read_bytes()
    names = {s for s in blob.split(b"\0") if s.startswith(b"_Z") and b"ode_step_kernel" in s and b"." not in s}
    assert len(names) == 1, names  # one kernel, found the way the library finds it (kernel_symbol)
