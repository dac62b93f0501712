"""Pseudo-ECG recovery (interface of src/beat/ecg.py:229-298 and the lead algebra of :301-397).

``ECGRecovery.solve`` recovers the transmembrane current density  Im  from  -C_m (Im, w) = (M grad v, grad w)
-- one consistent-mass solve with right-hand side K v, done with the diffusion step's PCG on the device --
and ``eval(point)`` gives the lead integral  1/(4 pi sigma_b) int Im / |x - p| dx  as a dot product with nodal
weights integrated once per electrode."""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Any, NamedTuple

import numpy as np

from . import _hip, grid
from ._engine import DiffusionSolver, build_ops, conductivity_array
from .stimulation import assemble_weights


class LeadForm:
    """What ``dolfinx.fem.form(... * dx)`` is to the caller: something ``assemble_scalar`` turns into a number."""

    def __init__(self, sol: grid.Function, weights):
        self.sol = sol
        self.weights = weights

    def assemble(self) -> float:
        ctx = self.sol._ctx
        out = C.c_double()
        f = self.sol.field
        _hip.check(ctx.lib.beat_field_dot(ctx.handle, f.ptr, self.weights.ptr, f.n, C.byref(out)))
        return out.value


def assemble_scalar(form: LeadForm) -> float:
    """``dolfinx.fem.assemble_scalar`` for lead forms (local part; all-reduce over ranks as the reference does)."""
    return form.assemble()


@dataclass
class ECGRecovery:
    v: grid.Function
    sigma_b: float | grid.Constant = 1.0
    C_m: float | grid.Constant = 1.0
    dx: grid.Measure | None = None
    M: Any = 1.0
    petsc_options: dict[str, Any] = field(default_factory=lambda: {"ksp_type": "cg", "pc_type": "sor",
                                                                   "ksp_rtol": 1.0e-8, "ksp_atol": 1.0e-8})

    def __post_init__(self):
        mesh = self.mesh
        self._ctx = self.v._ctx
        self._ops = build_ops(self._ctx, mesh, conductivity_array(self.M, mesh))
        self._ops.set_timestep(1.0, 0.0, 1.0)  # A = Mass
        self._solver = DiffusionSolver(self._ops, mesh.slab, group=mesh.comm.group)
        self.sol = grid.Function(self.V, name="Im")
        self._kv = self._ops.new_field()
        self._zero = self._ops.new_field()
        self.ksp = None

    @property
    def V(self) -> grid.FunctionSpace:
        return self.v.function_space

    @property
    def mesh(self) -> grid.Mesh:
        return self.v.function_space.mesh

    def solve(self):
        """Mass Im = -(1/C_m) K v."""
        ops = self._ops
        src = self.v.field
        if self.mesh.comm.size > 1:
            self._solver.exchange_halo(src)
        ops.apply(3, src, self._kv)
        x = self.sol.writable_field()
        x.fill(0.0)
        opts = self.petsc_options or {}
        self.ksp = self._solver.solve(self._zero, [self._kv], [-1.0 / float(self.C_m)], x,
                                      rtol=float(opts.get("ksp_rtol", 1e-8)), atol=float(opts.get("ksp_atol", 1e-8)),
                                      max_it=int(opts.get("ksp_max_it", 10_000)))
        self.sol._touch()

    def eval(self, point) -> LeadForm:
        mesh = self.mesh
        p = np.zeros(3)
        p[: len(point)] = np.asarray(point, dtype=np.float64)
        X = grid.SpatialCoordinate(mesh)
        r2 = sum((X[a] - float(p[a])) ** 2 for a in range(mesh.dim))
        kernel = (1.0 / (4.0 * np.pi * float(self.sigma_b))) / grid.sqrt(r2)
        cells = None if self.dx is None else self.dx.cells()
        w = self._ctx.field(mesh.num_nodes, mesh.plane)
        w.set(assemble_weights(mesh, cells, kernel))
        return LeadForm(self.sol, w)


# ---- lead algebra (standard 12-lead definitions, ecg.py:307-397) -------------------------------------------------
class Leads12(NamedTuple):
    RA: np.ndarray
    LA: np.ndarray
    LL: np.ndarray
    RL: np.ndarray | None = None
    V1: np.ndarray | None = None
    V2: np.ndarray | None = None
    V3: np.ndarray | None = None
    V4: np.ndarray | None = None
    V5: np.ndarray | None = None
    V6: np.ndarray | None = None

    @property
    def I(self):  # noqa: E743
        return self.LA - self.RA

    @property
    def II(self):
        return self.LL - self.RA

    @property
    def III(self):
        return self.LL - self.LA

    @property
    def Vw(self):
        """Wilson's central terminal."""
        return (self.RA + self.LA + self.LL) / 3.0

    @property
    def aVR(self):
        return 1.5 * (self.RA - self.Vw)

    @property
    def aVL(self):
        return 1.5 * (self.LA - self.Vw)

    @property
    def aVF(self):
        return 1.5 * (self.LL - self.Vw)

    def _precordial(self, name):
        val = getattr(self, name)
        if val is None:
            raise AttributeError(f"Missing attribute {name}")
        return val - self.Vw

    @property
    def V1_(self):
        return self._precordial("V1")

    @property
    def V2_(self):
        return self._precordial("V2")

    @property
    def V3_(self):
        return self._precordial("V3")

    @property
    def V4_(self):
        return self._precordial("V4")

    @property
    def V5_(self):
        return self._precordial("V5")

    @property
    def V6_(self):
        return self._precordial("V6")


# ---- QT-interval helpers on a lead signal (ecg.py:20-227), post-processing on the host ------------------------------
class QTIntervalResult(NamedTuple):
    qt_interval: float
    start_index: int
    end_index: int


def detect_r_peaks(ecg_signal: np.ndarray, min_distance: float = 20) -> np.ndarray:
    """Indices of the R peaks: local maxima at least ``min_distance`` samples apart and at least half as high as
    the largest sample."""
    from scipy.signal import find_peaks

    top = np.max(ecg_signal)
    peaks, _ = find_peaks(ecg_signal, distance=min_distance, height=0.5 * top if top > 0 else None)
    return peaks


def detect_t_end(averaged_rr: np.ndarray, r_peak_index: int, window_start_offset: int = 50,
                 window_end_offset: int = 400) -> int:
    """End of the T wave: inside the window [R + start, R + end) find the T peak (largest |signal|) and return the
    sample of steepest descent after it."""
    if averaged_rr is None or len(averaged_rr) == 0:
        raise RuntimeError("Error: Cannot detect T-end on empty or None averaged RR interval.")
    lo = max(0, r_peak_index + window_start_offset)
    hi = min(len(averaged_rr), r_peak_index + window_end_offset)
    segment = averaged_rr[lo:hi]
    slope = np.diff(segment)
    t_peak = int(np.argmax(np.abs(segment)))
    return int(lo + t_peak + np.argmin(slope[t_peak:]))


def qt_interval(t: np.ndarray, ecg_signal: np.ndarray, min_distance: float = 20.0, window_start_offset: int = 50,
                window_end_offset: int = 400) -> QTIntervalResult:
    r_peaks = detect_r_peaks(ecg_signal=ecg_signal, min_distance=min_distance)
    assert len(r_peaks) > 0, "No R-peaks detected. Check signal quality and detection parameters."
    start = int(r_peaks[0])
    end = detect_t_end(ecg_signal, start, window_start_offset=window_start_offset, window_end_offset=window_end_offset)
    return QTIntervalResult(start_index=start, end_index=end, qt_interval=t[end] - t[start])


def example(sampling_rate_hz: int = 1000, duration_s: float = 10, heart_rate_bpm: float = 60, q_offset_ms: float = 40,
            s_offset_ms: float = 40, t_peak_offset_ms: float = 200, r_width_ms: float = 20, q_width_ms: float = 20,
            s_width_ms: float = 30, t_width_ms: float = 60, qrs_peak_time: float = 200, noise_amplitude: float = 0.0,
            wander_freq_hz: float = 0.2, wander_amplitude: float = 0.1):
    """Synthetic lead signal: per beat four Gaussians (R +1.0, Q -0.2, S -0.3, T +0.4) plus optional noise and
    baseline wander; returns (time in ms, signal)."""
    rr_ms = 60_000.0 / heart_rate_bpm
    t_ms = np.linspace(0, duration_s * 1000, int(duration_s * sampling_rate_hz), endpoint=False)
    sig = np.zeros_like(t_ms)
    bump = lambda centre, width: np.exp(-(((t_ms - centre) / width) ** 2))  # noqa: E731
    for beat_no in range(int(duration_s / (rr_ms / 1000.0))):
        r_time = (beat_no + qrs_peak_time / 1000) * rr_ms
        sig += bump(r_time, r_width_ms) - 0.2 * bump(r_time - q_offset_ms, q_width_ms)
        sig += -0.3 * bump(r_time + s_offset_ms, s_width_ms) + 0.4 * bump(r_time + t_peak_offset_ms, t_width_ms)
    if noise_amplitude > 0:
        sig += noise_amplitude * np.random.randn(len(t_ms))
    sig += wander_amplitude * np.sin(2 * np.pi * (wander_freq_hz / 1000.0) * t_ms)
    return t_ms, sig
