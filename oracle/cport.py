"""ORACLE (test infrastructure only): ctypes loader of the C restatement oracle/beat_oracle.c."""

import ctypes as C
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_lib = None


def load():
    global _lib
    if _lib is None:
        path = _HERE / "liboracle.so"
        if not path.is_file():
            raise FileNotFoundError(f"{path} not built: run `make -C oracle`")
        lib = C.CDLL(str(path))
        lib.oracle_num_threads.restype = C.c_int
        lib.oracle_set_num_threads.argtypes = [C.c_int]
        lib.oracle_tp06_grl1.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_double, C.c_double]
        lib.oracle_stencil_apply.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p]
        lib.oracle_theta_step.restype = C.c_int
        lib.oracle_theta_step.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p,
                                          C.c_double, C.c_double, C.c_int, C.c_void_p]
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def tp06_grl1(states: np.ndarray, t: float, dt: float, parameters: np.ndarray) -> None:
    """In place on a C-contiguous (19, n) float64 array."""
    assert states.flags.c_contiguous and states.dtype == np.float64 and states.shape[0] == 19
    p = np.ascontiguousarray(parameters, dtype=np.float64)
    load().oracle_tp06_grl1(_p(states), states.shape[1], states.shape[1], _p(p), t, dt)


def stencil_apply(tab: np.ndarray, shape, x: np.ndarray) -> np.ndarray:
    nx, ny, nz = shape
    tab = np.ascontiguousarray(tab, dtype=np.float64)
    y = np.empty_like(x)
    load().oracle_stencil_apply(_p(tab), nx, ny, nz, _p(np.ascontiguousarray(x)), _p(y))
    return y


def theta_step(A_tab, B_tab, shape, v: np.ndarray, w, amp_dt, rtol, max_it=1000, work=None) -> int:
    nx, ny, nz = shape
    A_tab = np.ascontiguousarray(A_tab, dtype=np.float64)
    B_tab = np.ascontiguousarray(B_tab, dtype=np.float64)
    if work is None:
        work = np.empty(5 * v.size)
    return load().oracle_theta_step(_p(A_tab), _p(B_tab), nx, ny, nz, _p(v), None if w is None else _p(w), amp_dt,
                                    rtol, max_it, _p(work))
