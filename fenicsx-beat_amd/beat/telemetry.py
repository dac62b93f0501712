"""Monitors threaded through the PDE / ODE / splitting solvers.

Public contract of the reference's telemetry module (src/beat/telemetry.py:15-136): a monitor offers
``track_time(name)`` (context manager), ``record_ksp(ksp)`` and ``advance_step(t0, t1)``; ``PerformanceMonitor``
exposes ``timings``, ``step_counter``, the ``ksp_*`` statistics, ``display_summary()`` and ``save_summary(path)``
with the keys ``total_steps`` / ``ksp`` / ``timings``.

Kernels are launched asynchronously here, so a timed region is closed only after the device stream has drained
(``synchronize=True``); the do-nothing monitor never synchronises and is what the solvers use by default.
"""

from __future__ import annotations

import json
import logging
import time
from pathlib import Path

logger = logging.getLogger(__name__)


class _Region:
    """Context manager object handed out by track_time: adds the elapsed wall time to ``sink[name]`` on exit."""

    __slots__ = ("sink", "name", "sync", "start")

    def __init__(self, sink, name, sync):
        self.sink, self.name, self.sync = sink, name, sync

    def __enter__(self):
        self.start = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.sync is not None:
            self.sync()
        self.sink[self.name] = self.sink.get(self.name, 0.0) + (time.perf_counter() - self.start)
        return False


class _NoRegion:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_REGION = _NoRegion()


class BaseMonitor:
    """Interface; the defaults do nothing, so a subclass overrides only the hooks it cares about."""

    def track_time(self, name: str):
        return _NO_REGION

    def record_ksp(self, ksp) -> None:
        return None

    def advance_step(self, t0: float, t1: float) -> None:
        return None


class NullMonitor(BaseMonitor):
    """Explicit do-nothing monitor (the solvers' default)."""


def _drain_device():
    from ._device import Context

    ctx = Context._default
    if ctx is not None:
        ctx.synchronize()


class PerformanceMonitor(BaseMonitor):
    """Wall time per named region, linear-solver statistics, a log line every ``log_frequency`` steps."""

    _KSP_GETTERS = (("ksp_last_iterations", "getIterationNumber", int),
                    ("ksp_last_residual_norm", "getResidualNorm", float),
                    ("ksp_last_converged_reason", "getConvergedReason", int))

    def __init__(self, log_frequency: int = 1, comm=None, synchronize: bool = True):
        if comm is None:
            from .grid import COMM_WORLD as comm
        self.comm = comm
        self.log_frequency = log_frequency
        self.synchronize = synchronize
        self.timings: dict[str, float] = {}
        self.step_counter = 0
        self.ksp_total_iterations = 0
        self.ksp_max_iterations = 0
        self.ksp_last_iterations = 0
        self.ksp_last_residual_norm = 0.0
        self.ksp_last_converged_reason = 0

    # -- hooks ------------------------------------------------------------------------------------------------
    def track_time(self, name: str):
        return _Region(self.timings, name, _drain_device if self.synchronize else None)

    def record_ksp(self, ksp) -> None:
        """Reads iteration count, residual norm and converged reason off a KSP-like object (anything else, e.g.
        ``None`` before the first solve, is ignored)."""
        if not all(hasattr(ksp, getter) for _, getter, _ in self._KSP_GETTERS):
            return
        for attr, getter, cast in self._KSP_GETTERS:
            setattr(self, attr, cast(getattr(ksp, getter)()))
        self.ksp_total_iterations += self.ksp_last_iterations
        if self.ksp_last_iterations > self.ksp_max_iterations:
            self.ksp_max_iterations = self.ksp_last_iterations

    def advance_step(self, t0: float, t1: float) -> None:
        self.step_counter += 1
        if self.log_frequency > 0 and self.step_counter % self.log_frequency == 0:
            parts = [f"PDE step timing step={self.step_counter}", f"t=({t0:.5f}, {t1:.5f})",
                     f"ksp_iterations={self.ksp_last_iterations}",
                     f"ksp_residual_norm={self.ksp_last_residual_norm:.6e}",
                     f"ksp_converged_reason={self.ksp_last_converged_reason}"]
            parts += [f"{key}={seconds:.6f}s" for key, seconds in self.timings.items()]
            logger.info(", ".join(parts))

    # -- reports (rank 0 only) -----------------------------------------------------------------------------------
    def _report(self) -> dict:
        return {"total_steps": self.step_counter,
                "ksp": {"total_iterations": self.ksp_total_iterations, "max_iterations": self.ksp_max_iterations},
                "timings": self.timings}

    def display_summary(self) -> None:
        if self.comm.rank != 0:
            return
        rule = "=" * 50
        rows = sorted(self.timings.items(), key=lambda kv: -kv[1])
        text = [rule, "PERFORMANCE SUMMARY".center(50), rule,
                f"Total Steps:           {self.step_counter}",
                f"KSP Total Iterations:  {self.ksp_total_iterations}",
                f"KSP Max Iterations:    {self.ksp_max_iterations}",
                "-" * 50, f"{'Metric':<35} | {'Time (s)':>10}", "-" * 50]
        text += [f"{key:<35} | {seconds:>10.4f}" for key, seconds in rows]
        text.append(rule)
        logger.info("\n" + "\n".join(text) + "\n")

    def save_summary(self, filepath) -> None:
        if self.comm.rank != 0:
            return
        target = Path(filepath)
        target.parent.mkdir(parents=True, exist_ok=True)
        target.write_text(json.dumps(self._report(), indent=4))
        logger.info("Performance summary saved to %s", target)
