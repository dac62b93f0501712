"""ctypes binding of ``libbeat_hip.so`` (C ABI declared in ``include/beat_hip.h``).

There is deliberately no CPU fallback: if the shared library is missing or a call fails the
product path raises.  Loading the library does not need a GPU (tests check the exported
symbols on CPU-only machines); creating a context does.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_LIB_PATH = Path(os.environ.get("BEAT_HIP_LIBRARY", _HERE / "lib" / "libbeat_hip.so"))

MODEL_SIMPLE_ODE = 0
MODEL_FHN_DEMO = 1
MODEL_FHN_README = 2
MODEL_TP06_GRL1 = 3
MODEL_TORORD_DYNCL_GRL1 = 4
MODEL_TORORD_LAND_GRL1 = 5
MAX_STIM = 8
MAX_CLASSES = 32

# slots of the PCG scalar state (see include/beat_hip.h)
ST_BB, ST_RZ, ST_RR, ST_PQ, ST_RZN, ST_RRN, ST_TOL2, ST_BETA, ST_STOP, ST_ITERS, ST_REASON = range(11)
ST_NUPD = 14
ST_SIZE = 16
MAX_BATCH = 1024  # BEAT_MAX_BATCH: steps per beat_split_steps call


class BeatHipError(RuntimeError):
    pass


class KspInfo(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("converged_reason", C.c_int32),
        ("residual_norm", C.c_double),
        ("rhs_norm", C.c_double),
    ]


_vp = C.c_void_p
_i64 = C.c_int64
_dbl = C.c_double
_int = C.c_int

# name -> (restype, argtypes); every symbol declared in include/beat_hip.h
SIGNATURES = {
    "beat_abi_version": (_int, []),
    "beat_last_error": (C.c_char_p, []),
    "beat_ctx_create": (_int, [_int, _vp, C.POINTER(_vp)]),
    "beat_ctx_destroy": (_int, [_vp]),
    "beat_ctx_set_stream": (_int, [_vp, _vp]),
    "beat_ctx_synchronize": (_int, [_vp]),
    "beat_malloc": (_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "beat_free": (_int, [_vp, _vp]),
    "beat_memcpy_h2d": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "beat_memcpy_d2h": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "beat_ode_model_info": (_int, [_int, C.POINTER(_int), C.POINTER(_int)]),
    "beat_ode_model_register": (_int, [C.c_char_p, C.c_char_p, _int, _int, _int, C.POINTER(_int)]),
    "beat_ode_step": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _int, _vp, _i64, _dbl, _dbl, _int, _vp]),
    "beat_ode_step_pending": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _int, _vp, _i64, _dbl, _dbl, _int, _vp, _vp, _vp, _i64, _int]),
    "beat_ode_step_rows": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _int, _vp, _int, _vp, _i64, _dbl, _dbl, _int, _vp, _vp, _vp, _i64, _int]),
    "beat_ode_jit_stats": (_int, [C.POINTER(C.c_longlong)]),
    "beat_ode_class_table_doubles": (_int, [_int, C.POINTER(_int)]),
    "beat_ode_class_table_fill": (_int, [_vp, _int, _vp, _int, _int, _vp]),
    "beat_ode_step_classes": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _int, _vp, _dbl, _dbl, _int, _vp, _vp, _vp, _vp, _vp, _i64, _int]),
    "beat_ode_run": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _int, _vp, _i64, _dbl, _dbl, _i64, _int, _int, _vp, _int, _vp]),
    "beat_copy": (_int, [_vp, _vp, _vp, _i64]),
    "beat_fill": (_int, [_vp, _vp, _dbl, _i64]),
    "beat_stream_probe": (_int, [_vp, _vp, _i64, _int, _int, _int, _int, _int, _i64]),
    "beat_gather": (_int, [_vp, _vp, _vp, _vp, _i64]),
    "beat_scatter": (_int, [_vp, _vp, _vp, _vp, _i64]),
    "beat_interp2": (_int, [_vp, _vp, _vp, _vp, _vp, _i64]),
    "beat_pde_create": (_int, [_vp, C.POINTER(_i64), _int, _int, _vp, _vp, C.POINTER(_vp)]),
    "beat_pde_assemble_rows": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "beat_rows_apply_dirichlet": (_int, [_vp, C.POINTER(_i64), _vp, _i64, _vp, _vp, _vp]),
    "beat_pde_create_var": (_int, [_vp, C.POINTER(_i64), _int, _int, _vp, _vp, _i64, C.POINTER(_vp)]),
    "beat_pde_destroy": (_int, [_vp]),
    "beat_stencil_offsets": (C.POINTER(_int), []),
    "beat_pde_set_timestep": (_int, [_vp, _dbl, _dbl, _dbl]),
    "beat_pde_apply": (_int, [_vp, _int, _vp, _vp]),
    "beat_pde_rhs": (_int, [_vp, _vp, C.POINTER(_vp), C.POINTER(_dbl), _int, _vp, _vp, _vp, _vp]),
    "beat_pde_cg_begin": (_int, [_vp, _vp, _dbl, _dbl, _int]),
    "beat_pde_spmv_dot": (_int, [_vp, _vp, _vp, _vp]),
    "beat_pde_spmv_dot_part": (_int, [_vp, _vp, _vp, _vp, _int]),
    "beat_pde_cg_update": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "beat_pde_cg_next": (_int, [_vp, _vp, _vp, _vp]),
    "beat_pde_ring_size": (_int, []),
    "beat_pde_cg_update_r": (_int, [_vp, _vp, _vp, _vp, _int]),
    "beat_pde_cg_next_oop": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "beat_pde_x_flush": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int]),
    "beat_pde_set_small_grid_solve": (_int, [_vp, _int]),
    "beat_pde_small_grid_solve_active": (_int, [_vp]),
    "beat_split_steps": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _int, _int, _vp, _int, _vp, _vp, _vp, _vp, _int, _dbl, _dbl,
                                _int, _vp, _vp, _int, _vp, _vp]),
    "beat_split_steps_big": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _int, _int, _vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _dbl, _dbl, _int, _int,
                             C.POINTER(KspInfo), C.POINTER(_int), C.POINTER(C.c_float)]),  # host_pending: int[3]
    "beat_pde_set_guess_order": (_int, [_vp, _int]),
    "beat_pde_set_single_reduction": (_int, [_vp, _int]),
    "beat_pde_fused_dist_pass": (_int, [_vp]),
    "beat_pde_tile_route": (_int, [_vp]),
    "beat_pde_guess_reset": (_int, [_vp]),
    "beat_pde_guess_pending": (_int, [_vp]),
    "beat_pde_guess_history": (_int, [_vp, _vp, _vp, _vp]),
    "beat_pde_guess_traffic": (_int, [_vp, C.POINTER(_int)]),
    "beat_pde_set_preconditioner": (_int, [_vp, _int, _vp]),
    "beat_pde_pc_num_passes": (_int, [_vp]),
    "beat_pde_pc_pass": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp]),
    "beat_pde_cg_first_z": (_int, [_vp, _vp, _vp, _vp]),
    "beat_pde_cg_next_z": (_int, [_vp, _vp, _vp, _vp]),
    "beat_pde_work_fields": (_int, [_vp]),
    "beat_pde_field_stride": (_i64, [_vp]),
    "beat_pde_solve": (
        _int,
        [_vp, _vp, C.POINTER(_vp), C.POINTER(_dbl), _int, _vp, _vp, _dbl, _dbl, _int, C.POINTER(KspInfo)],
    ),
    "beat_pde_solve_begin": (_int, [_vp, _vp, C.POINTER(_vp), C.POINTER(_dbl), _int, _vp, _vp, _dbl, _dbl, _int]),
    "beat_pde_solve_end": (_int, [_vp, C.POINTER(KspInfo), C.POINTER(_int)]),
    "beat_pde_solve_is_open": (_int, [_vp]),
    "beat_pde_solve_can_open": (_int, [_vp]),
    "beat_pde_solve_ex": (
        _int,
        [_vp, _vp, C.POINTER(_vp), C.POINTER(_dbl), _int, _vp, _vp, _dbl, _dbl, _int, _int, C.POINTER(KspInfo), C.POINTER(_int)],
    ),
    "beat_field_probe": (_int, [_vp, _vp, _vp, _vp, _int, _vp]),
    "beat_field_probe_record": (_int, [_vp, _vp, _vp, _vp, _int, _vp]),
    "beat_field_dot": (_int, [_vp, _vp, _vp, _i64, C.POINTER(_dbl)]),
    "beat_field_minmax": (_int, [_vp, _vp, _i64, C.POINTER(_dbl), C.POINTER(_dbl)]),
}

# callbacks of a beat_comm whose transport is supplied by the caller (include/beat_hip.h)
HALO_FN = C.CFUNCTYPE(_int, _vp, _vp, _vp, _vp, _vp, _i64)
ALLREDUCE_FN = C.CFUNCTYPE(_int, _vp, _vp, _int)
ALLREDUCE_FN_OR_NULL = _vp  # an ALLREDUCE_FN cast to void* (ctypes function types do not accept None)
UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 2048
IPC_MAX_RANKS = 16
COMM_SERIAL = 1
MAX_SPARSE_ROWS = 16  # BEAT_MAX_SPARSE_ROWS of csrc/beat_ode_kernel.h: on an instance compiled for the rows (run-time compilation)
MAX_SPARSE_ROWS_RT = 4  # BEAT_MAX_SPARSE_ROWS_RT: on the shipped kernel
CUSTOM_MODEL_BASE = 100  # BEAT_MODEL_CUSTOM_BASE: ids of models registered as source (beat_ode_model_register)
TRANSPORT_NAMES = {0: "callbacks", 1: "rccl", 2: "rccl-serial", 3: "ipc"}
E_NOT_CONVERGED = -3

SIGNATURES.update({
    "beat_pde_set_ghost_types": (_int, [_vp, _int, _int]),
    "beat_comm_unique_id": (_int, [_vp]),
    "beat_comm_create_rccl": (_int, [_vp, _int, _int, _int, _int, _vp, C.POINTER(_vp)]),
    "beat_comm_create_rccl_ex": (_int, [_vp, _int, _int, _int, _int, _vp, _int, C.POINTER(_vp)]),
    "beat_comm_create_ipc": (_int, [_vp, _int, _int, _int, _int, _i64, _vp, ALLREDUCE_FN_OR_NULL, _vp, _vp, C.POINTER(_vp)]),
    "beat_comm_ipc_connect": (_int, [_vp, _vp, _vp]),
    "beat_comm_ipc_connect_all": (_int, [_vp, _vp, _int]),
    "beat_comm_ipc_connect_local": (_int, [_vp, _vp, _int]),
    "beat_comm_info": (_int, [_vp, C.POINTER(_int)]),
    "beat_comm_merged_solves": (_i64, [_vp]),
    "beat_comm_profile": (_int, [_vp, _int]),
    "beat_comm_profile_read": (_int, [_vp, C.POINTER(_dbl)]),
    "beat_comm_create_callbacks": (_int, [_vp, _int, _int, _int, _int, HALO_FN, ALLREDUCE_FN, _vp, C.POINTER(_vp)]),
    "beat_comm_destroy": (_int, [_vp]),
    "beat_comm_halo_exchange": (_int, [_vp, _vp, _i64, _i64]),
    "beat_comm_allreduce_sum": (_int, [_vp, _vp, _int]),
    "beat_pde_solve_dist": (
        _int,
        [_vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_dbl), _int, _vp, _vp, _dbl, _dbl, _int, _int, C.POINTER(KspInfo), C.POINTER(_int)],
    ),
    "beat_pde_solve_dist_begin": (_int, [_vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_dbl), _int, _vp, _vp, _dbl, _dbl, _int]),
})

_lib = None


def library_path() -> Path:
    return _LIB_PATH


def load():
    """Load libbeat_hip.so and declare every prototype.  Raises BeatHipError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.is_file():
        raise BeatHipError(
            f"HIP extension not built: {_LIB_PATH} is missing. Run `python -c 'import __graft_entry__ "
            "as g; g.build()'` (or `make -C fenicsx-beat_amd/csrc`). There is no CPU fallback."
        )
    # PyTorch FIRST: it brings its own copy of the HIP runtime (torch/lib/libamdhip64.so); loaded after this library's (the system's
    # copy) the process holds two runtimes, and the one that initialises second finds no device ("no ROCm-capable device is
    # detected" from beat_ctx_create although torch.cuda.is_available() -- seen when a generated model was registered before
    # anything had imported torch).  With torch's copy in the process this library's dependency resolves to it.
    try:
        import torch  # noqa: F401
    except ImportError:  # (a host without PyTorch: bare-ctypes use of the library, tests/_ctypes_only_script.py does not come through here)
        pass
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, allow_not_converged: bool = False) -> None:
    """Raise on a failed call.  ``allow_not_converged``: a solve that hit max_it is reported through its
    beat_ksp_info (converged_reason < 0), as PETSc's KSP does, not raised."""
    if rc == E_NOT_CONVERGED and allow_not_converged:
        return
    if rc != 0:
        msg = load().beat_last_error().decode(errors="replace")
        raise BeatHipError(f"libbeat_hip error {rc}: {msg}")


def stencil_offsets() -> list[tuple[int, int, int]]:
    p = load().beat_stencil_offsets()
    return [(p[3 * k], p[3 * k + 1], p[3 * k + 2]) for k in range(15)]
