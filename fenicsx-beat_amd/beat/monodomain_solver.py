"""Operator-splitting orchestrator -- interface, step order and monitor keys of
src/beat/monodomain_solver.py:14-116 (theta = 1 Godunov, theta != 1 adds a corrective ODE step).

When the ODE side is a built-in device model on the PDE's own P1 space, ``step`` takes a fused route
(theta == 1: one ionic kernel; theta < 1: the corrective ionic kernel follows the solve): the ionic kernel updates the state array, the diffusion solve runs in place
on its V row (previous and new potential share the storage), and ``pde.state`` / ``pde.v_`` /
``ode.v_ode`` are left as aliases of that row -- same values as the reference's six N-vector
copies per step (steps 2-4, 6, 7 of the reference sequence), none of the traffic."""

from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import Protocol

import numpy as np

from .monodomain_model import MonodomainModel
from .telemetry import BaseMonitor, NullMonitor

logger = logging.getLogger(__name__)
EPS = 1e-12


class ODESolver(Protocol):
    def to_dolfin(self) -> None: ...

    def from_dolfin(self) -> None: ...

    def ode_to_pde(self) -> None: ...

    def pde_to_ode(self) -> None: ...

    def step(self, t0: float, dt: float) -> None: ...


@dataclass
class MonodomainSplittingSolver:
    pde: MonodomainModel
    ode: ODESolver
    theta: float = 1.0
    monitor: BaseMonitor = field(default_factory=NullMonitor)
    fused: bool = True

    def __post_init__(self) -> None:
        self.ode.to_dolfin()
        self.ode.ode_to_pde()
        self.pde.assign_previous()

    def solve(self, interval, dt):
        T0, T = interval
        if dt is None:
            dt = T - T0
        t0 = T0
        t1 = T0 + dt
        while t1 < T + EPS:
            self.step((t0, t1))
            t0 = t1
            t1 = t0 + dt

    # ---------------------------------------------------------------------------------------
    def _can_fuse(self) -> bool:
        from .odesolver import DolfinODESolver

        ode, pde = self.ode, self.pde
        return (
            self.fused
            and 0.0 < self.theta <= 1.0
            and isinstance(ode, DolfinODESolver)
            and ode.on_device
            and isinstance(pde, MonodomainModel)
            and ode.v_pde is pde.state
            and ode.num_points == pde.state.x.array.size
        )

    def _fused_step(self, t0, t1):
        ode, pde, mon = self.ode, self.pde, self.monitor
        dt = t1 - t0
        row = ode._v_row
        with mon.track_time("total_step"):
            with mon.track_time("ode_step"):
                # functions aliasing the row are re-aliased below, so no materialisation here
                ode._dev.parameters = ode.parameters
                # a deferred x += sum alpha_j p_j of the previous solve is applied by this kernel
                ode._dev.step(t0, self.theta * dt, v_index=ode.v_index, pending_ops=ode._pending_ops, v_row=row)
            with mon.track_time("pde_step"):
                theta_pde = pde.parameters["theta"]
                with pde.monitor.track_time("pde_total_step"):
                    pde.time.value = t0 + theta_pde * dt
                    if not abs(dt - float(pde._timestep)) < 1.0e-12:
                        pde._timestep.value = dt
                        with pde.monitor.track_time("pde_update_matrices"):
                            pde._update_matrices()
                    stim_w, stim_amp = [], []
                    for s in pde._stimuli:
                        a = s.amplitude()
                        if a != 0.0 and s.field is not None:
                            stim_w.append(s.field)
                            stim_amp.append(a)
                    with pde.monitor.track_time("pde_linear_solve"):
                        pde.solve_in_place(row, stim_w, stim_amp, defer_flush=True)
                    ode._pending_ops = pde._ops
                    pde.monitor.record_ksp(pde.ksp)
                pde.monitor.advance_step(t0, t1)
            if not np.isclose(self.theta, 1.0):
                # corrective ionic step of length (1 - theta) dt from t0 + theta dt (monodomain_solver.py:98-113)
                with mon.track_time("corrective_ode_step"):
                    ode._dev.step(t0 + self.theta * dt, (1.0 - self.theta) * dt, v_index=ode.v_index,
                                  pending_ops=ode._pending_ops, v_row=row)
            with mon.track_time("pde_assign_previous_after"):
                for f in (pde.state, pde.v_, ode.v_ode):
                    f.alias_to(row, sync=pde._ops.flush_pending)
                ode._aliases = [pde.state, pde.v_, ode.v_ode]
        mon.advance_step(t0, t1)

    def _can_fuse_multi(self) -> bool:
        from .odesolver import DolfinMultiODESolver

        ode, pde = self.ode, self.pde
        return (
            self.fused
            and 0.0 < self.theta <= 1.0
            and isinstance(ode, DolfinMultiODESolver)
            and ode.on_device
            and isinstance(pde, MonodomainModel)
            and ode.v_pde is pde.state
            and ode.v_ode.x.array.size == pde.state.x.array.size
        )

    def _fused_multi_step(self, t0, t1):
        """Per-marker cell models: the potentials are scattered straight into the PDE unknown, the solve runs in
        place there, the new potentials are gathered back; ``pde.v_`` and ``ode.v_ode`` become aliases of
        ``pde.state`` -- the values of the literal sequence without its five full-vector copies per step."""
        ode, pde, mon = self.ode, self.pde, self.monitor
        dt = t1 - t0
        with mon.track_time("total_step"):
            with mon.track_time("ode_step"):
                ode.step(t0=t0, dt=self.theta * dt)
            x = pde.state.writable_field(overwrite_all=False)
            with mon.track_time("ode_to_dolfin"):
                ode.scatter_v(x)
            with mon.track_time("pde_step"):
                theta_pde = pde.parameters["theta"]
                with pde.monitor.track_time("pde_total_step"):
                    pde.time.value = t0 + theta_pde * dt
                    if not abs(dt - float(pde._timestep)) < 1.0e-12:
                        pde._timestep.value = dt
                        with pde.monitor.track_time("pde_update_matrices"):
                            pde._update_matrices()
                    stim_w, stim_amp = [], []
                    for s in pde._stimuli:
                        a = s.amplitude()
                        if a != 0.0 and s.field is not None:
                            stim_w.append(s.field)
                            stim_amp.append(a)
                    with pde.monitor.track_time("pde_linear_solve"):
                        pde.solve_in_place(x, stim_w, stim_amp)
                    pde.monitor.record_ksp(pde.ksp)
                pde.monitor.advance_step(t0, t1)
            with mon.track_time("ode_from_dolfin"):
                ode.gather_v(x)
            if not np.isclose(self.theta, 1.0):
                with mon.track_time("corrective_ode_step"):
                    ode.step(t0 + self.theta * dt, (1.0 - self.theta) * dt)
                with mon.track_time("corrective_ode_to_dolfin"):
                    ode.scatter_v(x)
            pde.state._touch()
            with mon.track_time("pde_assign_previous_after"):
                for f in (pde.v_, ode.v_ode):
                    if f is not pde.state:
                        f.alias_to(x)
        mon.advance_step(t0, t1)

    def step(self, interval):
        theta = self.theta
        t0, t1 = interval
        dt = t1 - t0
        t = t0 + theta * dt

        if self._can_fuse():
            return self._fused_step(t0, t1)
        if self._can_fuse_multi():
            return self._fused_multi_step(t0, t1)

        with self.monitor.track_time("total_step"):
            with self.monitor.track_time("ode_step"):
                self.ode.step(t0=t0, dt=theta * dt)
            with self.monitor.track_time("ode_to_dolfin"):
                self.ode.to_dolfin()
            with self.monitor.track_time("ode_to_pde"):
                self.ode.ode_to_pde()
            with self.monitor.track_time("pde_assign_previous_before"):
                self.pde.assign_previous()
            with self.monitor.track_time("pde_step"):
                self.pde.step((t0, t1))
            with self.monitor.track_time("pde_to_ode"):
                self.ode.pde_to_ode()
            with self.monitor.track_time("ode_from_dolfin"):
                self.ode.from_dolfin()
            if np.isclose(theta, 1.0):
                with self.monitor.track_time("pde_assign_previous_after"):
                    self.pde.assign_previous()
            else:
                with self.monitor.track_time("corrective_ode_step"):
                    self.ode.step(t, (1.0 - theta) * dt)
                with self.monitor.track_time("corrective_ode_to_dolfin"):
                    self.ode.to_dolfin()
                with self.monitor.track_time("corrective_ode_to_pde"):
                    self.ode.ode_to_pde()
                with self.monitor.track_time("corrective_pde_assign_previous"):
                    self.pde.assign_previous()
        self.monitor.advance_step(t0, t1)
