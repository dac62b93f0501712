"""beat (MI355X) -- HIP-native operator-split monodomain solver with the fenicsx-beat API surface.

``beat.MonodomainModel``, ``beat.odesolver.DolfinODESolver`` and ``beat.MonodomainSplittingSolver``
keep the reference's constructor signatures, methods and step semantics; meshes are structured
boxes (``beat.grid`` provides the handful of dolfinx/ufl names the callers use) and all arithmetic
runs in ``libbeat_hip.so``.  There is no CPU fallback."""

from . import (  # noqa: F401
    base_model,
    conductivities,
    ecg,
    geometry,
    grid,
    io,
    models,
    monodomain_model,
    monodomain_solver,
    odesolver,
    single_cell,
    stimulation,
    telemetry,
    units,
    utils,
)
from .ecg import ECGRecovery
from .monodomain_model import MonodomainModel
from .monodomain_solver import MonodomainSplittingSolver
from .stimulation import Stimulus
from .telemetry import BaseMonitor, NullMonitor, PerformanceMonitor

__version__ = "0.1.0"
__program_name__ = "fenicsx-beat-amd"

__all__ = [
    "monodomain_model", "odesolver", "base_model", "MonodomainModel", "monodomain_solver",
    "MonodomainSplittingSolver", "utils", "single_cell", "conductivities", "stimulation", "geometry", "grid", "models",
    "Stimulus", "io", "ecg", "ECGRecovery", "telemetry", "BaseMonitor", "NullMonitor", "PerformanceMonitor", "units",
]
