"""Drives libbeat_hip.so from bare ctypes -- no torch, no beat package: device memory from beat_malloc, transfers by
beat_memcpy_*, five TP06 split steps (ionic kernel + in-place diffusion solve on the V row) checked against the oracle.
This is the binding INTEGRATION.md shows, executed.  Run by tests/test_ctypes_only_gpu.py in a fresh interpreter."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import fem, ionic  # noqa: E402  (the checker)

lib = C.CDLL(str(ROOT / "fenicsx-beat_amd" / "beat" / "lib" / "libbeat_hip.so"))
vp, i64, dbl, cint = C.c_void_p, C.c_int64, C.c_double, C.c_int


class KspInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged_reason", C.c_int32), ("residual_norm", dbl), ("rhs_norm", dbl)]


lib.beat_last_error.restype = C.c_char_p
lib.beat_ctx_create.argtypes = [cint, vp, C.POINTER(vp)]
lib.beat_ctx_destroy.argtypes = [vp]
lib.beat_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
lib.beat_free.argtypes = [vp, vp]
lib.beat_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
lib.beat_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
lib.beat_ode_model_info.argtypes = [cint, C.POINTER(cint), C.POINTER(cint)]
lib.beat_ode_step.argtypes = [vp, cint, vp, i64, i64, vp, cint, vp, i64, dbl, dbl, cint, vp]
lib.beat_pde_create.argtypes = [vp, C.POINTER(i64), cint, cint, vp, vp, C.POINTER(vp)]
lib.beat_pde_destroy.argtypes = [vp]
lib.beat_pde_set_timestep.argtypes = [vp, dbl, dbl, dbl]
lib.beat_pde_work_fields.argtypes = [vp]
lib.beat_pde_solve.argtypes = [vp, vp, vp, vp, cint, vp, vp, dbl, dbl, cint, C.POINTER(KspInfo)]
lib.beat_field_minmax.argtypes = [vp, vp, i64, C.POINTER(dbl), C.POINTER(dbl)]


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libbeat_hip error {rc}: {lib.beat_last_error().decode()}")


def main():
    ctx = vp()
    check(lib.beat_ctx_create(0, None, C.byref(ctx)))  # NULL stream: the library's own default stream
    ns, npar = cint(), cint()
    check(lib.beat_ode_model_info(3, C.byref(ns), C.byref(npar)))
    assert (ns.value, npar.value) == (19, 53)

    cells, L = (12, 9, 5), (1.2, 0.9, 0.5)
    mesh = fem.BoxMesh(cells, L)
    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M = 9.5301e-4 * np.outer(f0, f0) + 1.2576e-4 * (np.eye(3) - np.outer(f0, f0))
    C_m, theta, dt = 0.01, 0.5, 0.01
    nn = tuple(c + 1 for c in cells)
    n, plane = mesh.num_nodes, nn[0] * nn[1]
    mt, kt = fem.stencil_table(3, tuple(l / c for l, c in zip(L, cells)), M, 1.0, 1.0)  # Mass and K tables (27, 15)
    mt, kt = np.ascontiguousarray(mt), np.ascontiguousarray(kt)
    pde = vp()
    check(lib.beat_pde_create(ctx, (i64 * 3)(*nn), 1, 1, mt.ctypes.data_as(vp), kt.ctypes.data_as(vp), C.byref(pde)))
    check(lib.beat_pde_set_timestep(pde, C_m, theta, dt))

    # (19, ld) state-major array with a ghost plane before row 0 and after the last row
    S = 19
    ld = (n + 2 * plane + 31) // 32 * 32
    raw = vp()
    check(lib.beat_malloc(ctx, 8 * (plane + S * ld + plane), C.byref(raw)))
    states = vp(raw.value + 8 * plane)
    nwork = lib.beat_pde_work_fields(pde)
    lib.beat_pde_field_stride.restype = i64
    lib.beat_pde_field_stride.argtypes = [vp]
    fld = lib.beat_pde_field_stride(pde)  # a field with its ghost planes (+ padding on big grids)
    assert fld >= n + 2 * plane
    work = vp()
    check(lib.beat_malloc(ctx, 8 * nwork * fld, C.byref(work)))

    vi = ionic.tp06_state_index("V")
    S0 = np.repeat(ionic.tp06_init_state_values()[:, None], n, axis=1)
    S0[vi] += 60.0 * np.exp(-((mesh.x - 0.5 * np.array(L)) ** 2).sum(axis=1) / 0.05)
    S0 = np.ascontiguousarray(S0)
    for k in range(S):
        check(lib.beat_memcpy_h2d(ctx, vp(states.value + 8 * k * ld), S0[k].ctypes.data_as(vp), 8 * n))
    P = np.ascontiguousarray(ionic.tp06_init_parameter_values(stim_amplitude=0.0))
    v_row = vp(states.value + 8 * vi * ld)

    Sr = S0.copy()
    model = fem.OracleMonodomainModel(mesh, M, [], C_m=C_m, theta=theta, default_timestep=dt)
    info = KspInfo()
    t = 0.0
    for _ in range(5):
        check(lib.beat_ode_step(ctx, 3, states, n, ld, P.ctypes.data_as(vp), 53, None, 0, t, dt, vi, None))
        check(lib.beat_pde_solve(pde, v_row, None, None, 0, v_row, work, 1e-12, 1e-50, 500, C.byref(info)))
        assert info.converged_reason > 0 and info.iterations > 0
        Sr = ionic.tp06_generalized_rush_larsen(Sr, t, dt, P)
        model.state[:] = Sr[vi]
        model.assign_previous()
        model.step((t, t + dt))
        Sr[vi] = model.state
        t += dt
    out = np.empty((S, n))
    for k in range(S):
        check(lib.beat_memcpy_d2h(ctx, out[k].ctypes.data_as(vp), vp(states.value + 8 * k * ld), 8 * n))
    lo, hi = dbl(), dbl()
    check(lib.beat_field_minmax(ctx, v_row, n, C.byref(lo), C.byref(hi)))
    assert lo.value == out[vi].min() and hi.value == out[vi].max()
    err = np.abs(out - Sr) / np.maximum(np.abs(Sr), 1e-3)
    assert np.isfinite(out).all() and err.max() < 1e-9, err.max()
    check(lib.beat_free(ctx, work))
    check(lib.beat_free(ctx, raw))
    check(lib.beat_pde_destroy(pde))
    check(lib.beat_ctx_destroy(ctx))
    assert "torch" not in sys.modules and "beat" not in sys.modules
    print(f"ctypes-only ok: max rel diff vs oracle {err.max():.2e}, last PCG its {info.iterations}")


if __name__ == "__main__":
    main()
