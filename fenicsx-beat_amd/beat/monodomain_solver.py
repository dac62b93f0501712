"""Operator-splitting orchestrator -- interface, step order and monitor keys of
src/beat/monodomain_solver.py:14-116 (theta = 1 Godunov, theta != 1 adds a corrective ODE step).

When the ODE side is a built-in device model on the PDE's own P1 space, ``step`` takes a fused route
(theta == 1: one ionic kernel; theta < 1: the corrective ionic kernel follows the solve): the ionic kernel updates the state array, the diffusion solve runs in place
on its V row (previous and new potential share the storage), and ``pde.state`` / ``pde.v_`` /
``ode.v_ode`` are left as aliases of that row -- same values as the reference's six N-vector
copies per step (steps 2-4, 6, 7 of the reference sequence), none of the traffic."""

from __future__ import annotations

import logging
import os
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

    def solve(self, interval, dt, recorder=None):
        """The reference's loop (monodomain_solver.py:53-66).  ``recorder`` (a ``grid.ProbeRecorder`` on ``pde.state``;
        not part of the reference's signature) gets one row per step.  On a grid small enough for the one-launch
        diffusion solve, with a device cell model, theta = 1, stimuli whose time dependence is a scalar factor and no
        monitor attached, the steps are handed to the library in batches (beat_split_steps): same kernels, same
        values, no host round trip between steps."""
        T0, T = interval
        if dt is None:
            dt = T - T0
        steps = []
        t0 = T0
        t1 = T0 + dt
        while t1 < T + EPS:
            steps.append((t0, t1))
            t0 = t1
            t1 = t0 + dt
        if steps and self._can_batch(recorder):
            self._batched_steps(steps, recorder)
            return
        for iv in steps:
            self.step(iv)
            if recorder is not None:
                recorder.record()

    def _can_batch(self, recorder) -> bool:
        from .odesolver import DolfinODESolver

        if not (isinstance(self.ode, DolfinODESolver) and self._can_fuse() and np.isclose(self.theta, 1.0)):
            return False
        ode, pde = self.ode, self.pde
        ops = getattr(pde, "_ops", None)
        if ops is None or not hasattr(ops, "small_active"):
            return False
        if not ops.small_active():
            # a grid of any size on one rank: the step loop inside the library (beat_split_steps_big); no probe rows there
            # (the potential is complete only after the next ionic launch or a flush)
            diff = getattr(pde, "_diffusion", None)
            one_rank = diff is not None and getattr(diff, "dist", None) is None and getattr(diff, "libcomm", None) is None
            if not (one_rank and recorder is None and hasattr(ops, "work") and os.environ.get("BEAT_BATCH_BIG", "1") != "0"):
                return False
        if not all(type(m) is NullMonitor for m in (self.monitor, pde.monitor, getattr(ode._dev, "monitor", NullMonitor()))):
            return False  # someone wants per-step timings / KSP records
        if any(getattr(s, "general", None) is not None or hasattr(s, "cellfun") for s in pde._stimuli):
            return False  # weights that change from step to step
        if recorder is not None and recorder._f is not pde.state:
            return False
        ode._dev.parameters = ode.parameters
        return ode._dev._param_args()[0] is not None  # uniform parameters (one host vector)

    def _batched_steps(self, steps, recorder) -> None:
        import ctypes as C

        from . import _hip
        from ._engine import KspResult

        ode, pde = self.ode, self.pde
        ops, dev, row = pde._ops, ode._dev, ode._v_row
        dt = steps[0][1] - steps[0][0]
        # What an earlier step or solve() left for the next ionic launch (the x update of its last solve, deferred): the first launch
        # of the library's loop applies it, as step() would have (pending_in) -- a call of solve() per output interval does not pay
        # a pass over the potential per call (2 ms at 512^3).  Anything else that is pending (another row, another time step, the
        # one-launch path) is flushed.
        carry = 0
        if getattr(ops, "open_x", None) is not None:
            ops.solve_finish()
        same_dt = abs(dt - float(pde._timestep)) < 1.0e-12
        pend = getattr(ops, "pending", None)
        if (same_dt and pend is not None and not ops.small_active() and pend[0].ptr.value == row.ptr.value
                and ode._pending_ops is ops and os.environ.get("BEAT_BATCH_CARRY", "1") != "0"):
            carry = int(pend[2])
            ops.pending = None  # (the guess increment, if one is due, is known to the operator)
        else:
            ops.flush_pending()
        self._last_carry = carry
        if not same_dt:
            pde._timestep.value = dt
            pde._update_matrices()
        theta_pde = pde.parameters["theta"]
        stims = [s for s in pde._stimuli if s.field is not None]
        rtol, atol, max_it = pde._solver_tolerances()
        hp, npar, _, _ = dev._param_args()
        w_ptrs = (C.c_void_p * max(1, len(stims)))(*[s.field.ptr for s in stims])
        done = 0
        if not ops.small_active():
            self._batched_steps_big(steps, stims, w_ptrs, hp, npar, (rtol, atol, max_it), pending_in=carry)
            done = len(steps)
        while done < len(steps):
            nb = min(len(steps) - done, _hip.MAX_BATCH)
            probe = (None, None, 0, None)
            if recorder is not None:
                ptr, nb = recorder._reserve(nb)
                probe = (recorder._idx.ctypes.data_as(C.c_void_p), recorder._wts.ctypes.data_as(C.c_void_p), recorder.npts,
                         C.c_void_p(ptr))
            chunk = steps[done : done + nb]
            t_start = np.ascontiguousarray([a for a, _ in chunk], dtype=np.float64)
            dts = np.ascontiguousarray([self.theta * (b - a) for a, b in chunk], dtype=np.float64)
            amps = np.zeros((nb, max(1, len(stims))))
            for k, (a, _) in enumerate(chunk):  # the stimulus expressions are evaluated at t0 + theta dt, as step() does
                pde.time.value = a + theta_pde * (chunk[k][1] - a)
                for j, s in enumerate(stims):
                    amps[k, j] = s.amplitude()
            infos = (_hip.KspInfo * nb)()
            rc = dev.ctx.lib.beat_split_steps(
                dev.ctx.handle, dev.model.model_id, dev.states.ptr, dev.n, dev.states.ld, hp, npar, int(ode.v_index), ops.handle,
                nb, t_start.ctypes.data_as(C.c_void_p), dts.ctypes.data_as(C.c_void_p), w_ptrs, amps.ctypes.data_as(C.c_void_p), len(stims),
                rtol, atol, max_it, *probe, infos)
            _hip.check(rc, allow_not_converged=True)
            if recorder is not None:
                recorder._commit(nb)
            bad = [i for i in range(nb) if infos[i].converged_reason < 0]
            last = infos[bad[0]] if bad else infos[nb - 1]
            pde.ksp = KspResult(last.iterations, last.residual_norm, last.converged_reason, last.rhs_norm)
            pde._check_converged()
            done += nb
        ode._pending_ops = ops
        for f in (pde.state, pde.v_, ode.v_ode):
            f.alias_to(row, sync=ops.flush_pending)
        ode._aliases = [pde.state, pde.v_, ode.v_ode]

    def _batched_steps_big(self, steps, stims, w_ptrs, hp, npar, tol, pending_in: int = 0) -> None:
        """The steps of a grid too big for the one-launch solve, run by the library's own loop (beat_split_steps_big): per step the
        ionic launch that applies what the previous solve deferred and the solve in place on the potential row -- what
        ``_fused_step`` does, without Python between the steps.  ``self.batch_ode_ms`` (a list, if the caller sets one) collects
        the duration of every ionic launch."""
        import ctypes as C

        from . import _hip
        from ._engine import KspResult

        ode, pde = self.ode, self.pde
        ops, dev, row = pde._ops, ode._dev, ode._v_row
        theta_pde = pde.parameters["theta"]
        rtol, atol, max_it = tol
        # pending_in: what an earlier deferred solve left for the next ionic launch (directions of its last ring cycle): applied by
        # the first launch of the batch (_batched_steps hands it over when it belongs to this row and flushes it otherwise)
        times = getattr(self, "batch_ode_ms", None)
        done = 0
        # a call into the library cannot be interrupted: its length is kept near one second of steps (the first call makes 16 and
        # times them; 14 ms per step at 512^3, ten times that at 1024^3), and it returns early at a solve that ran out of iterations
        import time as _time

        cap = min(16, _hip.MAX_BATCH)
        try:
            while done < len(steps):
                nb = min(len(steps) - done, cap)
                chunk = steps[done : done + nb]
                t_start = np.ascontiguousarray([a for a, _ in chunk], dtype=np.float64)
                dts = np.ascontiguousarray([self.theta * (b - a) for a, b in chunk], dtype=np.float64)
                amps = np.zeros((nb, max(1, len(stims))))
                for k, (a, b) in enumerate(chunk):  # the stimulus expressions are evaluated at t0 + theta dt, as step() does
                    pde.time.value = a + theta_pde * (b - a)
                    for j, s in enumerate(stims):
                        amps[k, j] = s.amplitude()
                infos = (_hip.KspInfo * nb)()
                pend = (C.c_int * 3)()
                ode_ms = (C.c_float * nb)() if times is not None else None
                ops.st_ptr_for_flush = None
                tic = _time.perf_counter()
                rc = dev.ctx.lib.beat_split_steps_big(
                    dev.ctx.handle, dev.model.model_id, dev.states.ptr, dev.n, dev.states.ld, hp, npar, int(ode.v_index), ops.handle,
                    C.c_void_p(ops.work.data_ptr()), nb, t_start.ctypes.data_as(C.c_void_p), dts.ctypes.data_as(C.c_void_p), w_ptrs,
                    amps.ctypes.data_as(C.c_void_p), len(stims), rtol, atol, max_it, pending_in, infos, pend, ode_ms)
                ran = int(pend[2])  # steps done: all of them, or up to and including a solve that did not converge
                per_step = (_time.perf_counter() - tic) / max(1, ran)
                cap = int(min(_hip.MAX_BATCH, max(1, 1.0 / max(per_step, 1e-6))))
                pending_in = int(pend[1])
                ops.pending = (row, int(pend[0]), int(pend[1])) if (pend[1] > 0 or dev.ctx.lib.beat_pde_guess_pending(ops.handle)) else None
                if times is not None:
                    times.extend(float(v) for v in ode_ms[:ran])
                if ran > 0:
                    last = infos[ran - 1]
                    pde.ksp = KspResult(last.iterations, last.residual_norm, last.converged_reason, last.rhs_norm)
                    pde.time.value = chunk[ran - 1][0] + theta_pde * (chunk[ran - 1][1] - chunk[ran - 1][0])
                _hip.check(rc, allow_not_converged=True)
                # (with ksp_error_if_not_converged this raises HERE, with the state as the failing step left it; without, the
                # reference's loop goes on from that iterate and so does the next call)
                pde._check_converged()
                done += max(ran, 0)
                if ran == 0:
                    raise _hip.BeatHipError("beat_split_steps_big made no progress")
                if done < len(steps):
                    ops.pending = None  # the next batch's first ionic launch applies it (pending_in)
        finally:  # (also when ksp_error_if_not_converged raised: the row, its aliases and what is pending stay consistent)
            ode._pending_ops = ops
            for f in (pde.state, pde.v_, ode.v_ode):
                f.alias_to(row, sync=ops.flush_pending)
            ode._aliases = [pde.state, pde.v_, ode.v_ode]

    # ---------------------------------------------------------------------------------------
    def _can_fuse(self) -> bool:
        from .odesolver import DolfinMultiODESolver, DolfinODESolver

        ode, pde = self.ode, self.pde
        if isinstance(ode, DolfinMultiODESolver):
            # markers that share one device model live in one state array advanced by one launch: the single-model route
            return (self.fused and 0.0 < self.theta <= 1.0 and getattr(ode, "_marked", False) and isinstance(pde, MonodomainModel)
                    and ode.v_pde is pde.state and ode.v_ode.x.array.size == pde.state.x.array.size)
        return (
            self.fused
            and 0.0 < self.theta <= 1.0
            and isinstance(ode, DolfinODESolver)
            and ode.on_device
            and isinstance(pde, MonodomainModel)
            and ode.v_pde is pde.state
            and ode.num_points == pde.state.x.array.size
        )

    def _fused_prepare(self) -> int:
        """Parameters as they are now (the reference hands the live arrays to ``fun`` every step); returns the row of
        the potential."""
        ode = self.ode
        if hasattr(ode, "_fused_prepare"):
            return ode._fused_prepare()
        ode._dev.parameters = ode.parameters
        return ode.v_index

    def _fused_step(self, t0, t1):
        ode, pde, mon = self.ode, self.pde, self.monitor
        dt = t1 - t0
        row = ode._v_row
        with mon.track_time("total_step"):
            with mon.track_time("ode_step"):
                # functions aliasing the row are re-aliased below, so no materialisation here
                v_index = self._fused_prepare()
                # a deferred x += sum alpha_j p_j of the previous solve is applied by this kernel
                ode._dev.step(t0, self.theta * dt, v_index=v_index, pending_ops=ode._pending_ops, v_row=row)
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
                    # (with nobody asking for the KSP record step by step the solve is left open: the next step's ionic launch
                    # goes into the queue behind it before the host waits, and the device never idles in between)
                    lazy = pde.can_solve_lazily()
                    with pde.monitor.track_time("pde_linear_solve"):
                        pde.solve_in_place(row, stim_w, stim_amp, defer_flush=True, lazy=lazy)
                    ode._pending_ops = pde._ops
                    if not lazy:
                        pde.monitor.record_ksp(pde.ksp)
                pde.monitor.advance_step(t0, t1)
            if not np.isclose(self.theta, 1.0):
                # corrective ionic step of length (1 - theta) dt from t0 + theta dt (monodomain_solver.py:98-113)
                with mon.track_time("corrective_ode_step"):
                    ode._dev.step(t0 + self.theta * dt, (1.0 - self.theta) * dt, v_index=v_index,
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
