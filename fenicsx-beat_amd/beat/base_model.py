"""PDE (diffusion) step -- interface and step/solve semantics of src/beat/base_model.py:23-297,
with the DOLFINx assembly + PETSc KSP replaced by the matrix-free HIP operators of libbeat_hip
(right-hand-side build + Jacobi-PCG, see _engine.DiffusionSolver)."""

from __future__ import annotations

import abc
import os
import logging
from enum import Enum, auto
from typing import Any, Literal, NamedTuple

import numpy as np

from . import grid
from .stimulation import Stimulus, assemble_facet_weights, assemble_weights
from .telemetry import BaseMonitor, NullMonitor

logger = logging.getLogger(__name__)


class Status(str, Enum):
    OK = auto()
    NOT_CONVERGING = auto()


class Results(NamedTuple):
    state: grid.Function
    status: Status


def _transform_I_s(I_s, dZ: grid.Measure) -> list[Stimulus]:
    if I_s is None:
        return [Stimulus(expr=grid.zero(), dZ=dZ)]
    if isinstance(I_s, Stimulus):
        return [I_s]
    if isinstance(I_s, (grid.Expr, grid.CellFunction)):
        return [Stimulus(expr=I_s, dZ=dZ)]
    return list(I_s)


class _CompiledStimulus:
    """amplitude(t) (host scalar) x nodal weight field (device)."""

    def __init__(self, model, stim: Stimulus):
        self.model = model
        self.stim = stim
        mesh = model._mesh
        if isinstance(stim.expr, grid.CellFunction):
            # piecewise-constant current updated by the caller (demos/ukb_atlas.py:340-356): nodal weights are
            # rebuilt from the non-zero cells whenever the values change
            self.cellfun = stim.expr
            self.cells = stim.dz.cells()
            self.facets = None
            self.zero = False
            self.general = None
            self.field = model._ctx.field(mesh.num_nodes, mesh.plane)
            self._seen = -1
            self.amplitude = self._refresh_cells
            return
        expr = grid.as_expr(stim.expr)
        measure = stim.dz
        self.facets = measure.facets() if measure.integral_type == "ds" else None
        self.cells = None if self.facets is not None else measure.cells()
        self.zero = isinstance(expr, grid.Literal) and expr.value == 0.0
        sep = grid.separate(expr)
        self.field = None
        self.general = None
        if self.zero:
            self.amplitude = lambda: 0.0
        elif self.facets is not None and sep is not None:
            # any UFL expression on a surface measure (stimulation.py:14-24, base_model.py:247-248): the
            # coordinate-dependent factor goes into the nodal weights by quadrature on the facets
            spatial, temporal = sep
            self.field = model._ctx.field(mesh.num_nodes, mesh.plane)
            self.field.set(assemble_facet_weights(mesh, self.facets, spatial))
            self.amplitude = (lambda: 1.0) if temporal is None else (lambda: float(temporal.evaluate()))
        elif sep is not None:
            spatial, temporal = sep
            self.field = model._ctx.field(mesh.num_nodes, mesh.plane)
            self.field.set(assemble_weights(mesh, self.cells, spatial))
            self.amplitude = (lambda: 1.0) if temporal is None else (lambda: float(temporal.evaluate()))
        else:
            # coordinate and time mixed inside one factor: re-integrate every step (host; rare)
            self.general = expr
            self.field = model._ctx.field(mesh.num_nodes, mesh.plane)
            self.amplitude = self._refresh

    def _refresh_cells(self) -> float:
        f = self.cellfun
        if f._version != self._seen:
            mesh = self.model._mesh
            vals = np.asarray(f.x.array)
            if self.cells is not None:
                sel = np.zeros(vals.shape, dtype=bool)
                sel[self.cells] = True
                vals = np.where(sel, vals, 0.0)
            ids = np.flatnonzero(vals)
            if mesh.active is not None:
                ids = ids[mesh.active[ids]]
            d = mesh.dim
            vol = float(np.prod(mesh.h)) / {1: 1, 2: 2, 3: 6}[d]
            w = np.zeros(mesh.num_nodes)
            lo, hi = mesh.slab.z0 * mesh.plane, mesh.slab.z1 * mesh.plane
            if len(ids):
                verts = mesh.cell_vertices(ids)
                contrib = np.repeat(vals[ids] * vol / (d + 1), d + 1)
                v = verts.ravel()
                keep = (v >= lo) & (v < hi)
                np.add.at(w, v[keep] - lo, contrib[keep])
            self.field.set(w)
            self._nonzero = bool(len(ids))
            self._seen = f._version
        return 1.0 if self._nonzero else 0.0

    def _refresh(self) -> float:
        if self.facets is not None:
            self.field.set(assemble_facet_weights(self.model._mesh, self.facets, self.general))
        else:
            self.field.set(assemble_weights(self.model._mesh, self.cells, self.general))
        return 1.0


class BaseModel:
    def __init__(self, time: grid.Constant, mesh: grid.Mesh, dx: grid.Measure | None = None,
                 params: dict[str, Any] | None = None, I_s=None, monitor: BaseMonitor | None = None,
                 **kwargs: Any) -> None:
        if kwargs:
            logger.warning("Unused keyword arguments: %s", ", ".join(f"{k}={v}" for k, v in kwargs.items()))
        from ._device import Context

        self._mesh = mesh
        self._ctx = Context.default()
        self.time = time
        self.dx = dx or grid.dx(domain=mesh)
        self.monitor = monitor or NullMonitor()

        self.parameters = type(self).default_parameters()
        if params is not None:
            self.parameters.update(params)

        self._I_s = _transform_I_s(I_s, dZ=self.dx)
        self._setup_state_space()
        self._timestep = grid.Constant(mesh, self.parameters["default_timestep"])
        self._setup_operators()
        # initial guess of the linear solves from the previous steps' increments (PETSc's KSPGuess; the reference
        # leaves it off): petsc_options["ksp_guess_order"] in 0..4 or "auto", default BEAT_GUESS_ORDER or "auto" (quadratic
        # or cubic extrapolation in time, whichever has been costing fewer iterations: the quadratic on the benchmark's
        # steps, the cubic on a developed front)
        order = (self.parameters.get("petsc_options") or {}).get("ksp_guess_order", os.environ.get("BEAT_GUESS_ORDER", "auto"))
        if hasattr(self._ops, "set_guess_order"):
            from ._hip import BeatHipError

            try:
                self._ops.set_guess_order("auto" if order in ("auto", "-1", -1) else int(order))
            except BeatHipError as exc:  # e.g. no memory for the history fields on a grid that fills the GPU
                logger.warning("initial guess from previous steps disabled (%s): solves start from x0 = v_", exc)
                self._ops.set_guess_order(0)
        # petsc_options["ksp_cg_single_reduction"] (PETSc's KSPCGUseSingleReduction): one all-reduce per PCG iteration on a
        # decomposed grid instead of two; unset: what BEAT_DIST_MERGED says
        single = (self.parameters.get("petsc_options") or {}).get("ksp_cg_single_reduction")
        if single is not None and hasattr(self._ops, "set_single_reduction"):
            self._ops.set_single_reduction(str(single).lower() not in ("0", "false", "no", ""))
        self._stimuli = [_CompiledStimulus(self, s) for s in self._I_s]
        self._update_matrices()
        self._ksp = None  # KSP-like record of the last solve (see the ``ksp`` property)
        self.status = Status.OK  # NOT_CONVERGING once a linear solve has run out of iterations

    @property
    def ksp(self):
        """KSP-like record of the last solve (what telemetry.py:67-76 reads from PETSc's KSP).  A solve the fused step left
        open is finished here."""
        ops = getattr(self, "_ops", None)
        if ops is not None and getattr(ops, "open_x", None) is not None:
            ops.solve_finish()
        return self._ksp

    @ksp.setter
    def ksp(self, value) -> None:
        self._ksp = value

    @property
    def status(self) -> Status:
        """``Status.NOT_CONVERGING`` once a linear solve has run out of iterations (base_model.py:23-30).  Like ``ksp``, reading it
        finishes a solve the fused step left open: the record of the LAST step is in it too."""
        ops = getattr(self, "_ops", None)
        if ops is not None and getattr(ops, "open_x", None) is not None:
            ops.solve_finish()
        return self._status

    @status.setter
    def status(self, value: Status) -> None:
        self._status = value

    @abc.abstractmethod
    def _setup_state_space(self) -> None: ...

    @abc.abstractmethod
    def _setup_operators(self) -> None: ...

    @property
    @abc.abstractmethod
    def state(self) -> grid.Function: ...

    @abc.abstractmethod
    def assign_previous(self) -> None: ...

    @staticmethod
    def default_parameters(solver_type: Literal["iterative", "direct"] = "direct") -> dict[str, Any]:
        if solver_type == "iterative":
            petsc_options = {"ksp_type": "cg", "pc_type": "hypre", "pc_hypre_type": "boomeramg"}
        else:
            petsc_options = {"ksp_type": "preonly", "pc_type": "lu", "pc_factor_mat_solver_type": "mumps"}
        return {
            "theta": 0.5,
            "degree": 1,
            "family": "Lagrange",
            "default_timestep": 1.0,
            "jit_options": {},
            "form_compiler_options": {},
            "petsc_options": petsc_options,
            "log_timings": False,
            "timing_log_frequency": 1,
        }

    # -- linear-solver controls: the recognised subset of the reference's petsc_options -----------
    def _solver_tolerances(self):
        opts = self.parameters.get("petsc_options") or {}
        # The reference's default is an exact solve (LU); PCG is therefore run to a tight relative
        # residual unless the caller asks for a looser one through ksp_rtol.
        rtol = float(opts.get("ksp_rtol", 1e-10))
        atol = float(opts.get("ksp_atol", 1e-50))
        max_it = int(opts.get("ksp_max_it", 10_000))
        return rtol, atol, max_it

    def _check_converged(self) -> None:
        """A solve that hit ksp_max_it is reported the way PETSc reports it -- ``ksp.getConvergedReason() < 0``, seen
        by the monitor (telemetry.py:67-76) -- and makes ``solve()`` return ``Status.NOT_CONVERGING``
        (base_model.py:23-30); it only raises with ``petsc_options["ksp_error_if_not_converged"]``."""
        ksp = self.ksp
        if ksp is None or ksp.converged_reason >= 0:
            return
        self.status = Status.NOT_CONVERGING
        msg = (f"linear solve did not converge: reason {ksp.converged_reason} after {ksp.iterations} iterations, "
               f"||r|| = {ksp.residual_norm:.3e}, ||b|| = {ksp.rhs_norm:.3e}")
        if (self.parameters.get("petsc_options") or {}).get("ksp_error_if_not_converged"):
            raise RuntimeError(msg)
        logger.warning(msg)

    def _update_matrices(self):
        """A = C_m Mass + theta dt K, B = C_m Mass - (1 - theta) dt K for the current dt
        (base_model.py:188-194)."""
        self._ops.set_timestep(float(self.C_m), float(self.parameters["theta"]), float(self._timestep))

    def step(self, interval):
        t0, t1 = interval
        dt = t1 - t0
        theta = self.parameters["theta"]
        t = t0 + theta * dt

        with self.monitor.track_time("pde_total_step"):
            with self.monitor.track_time("pde_set_time"):
                self.time.value = t

            timestep_unchanged = abs(dt - float(self._timestep)) < 1.0e-12
            if not timestep_unchanged:
                self._timestep.value = dt
                with self.monitor.track_time("pde_update_matrices"):
                    self._update_matrices()

            with self.monitor.track_time("pde_update_rhs"):
                stim_w, stim_amp = [], []
                for s in self._stimuli:
                    a = s.amplitude()
                    if a != 0.0 and s.field is not None:
                        stim_w.append(s.field)
                        stim_amp.append(a)

            with self.monitor.track_time("pde_linear_solve"):
                self._solve_linear(stim_w, stim_amp)

            self.monitor.record_ksp(self.ksp)
            self._check_converged()

            with self.monitor.track_time("pde_scatter_forward"):
                pass  # ghost planes are refreshed by the halo exchange inside the solve

        self.monitor.advance_step(t0, t1)

    @abc.abstractmethod
    def _solve_linear(self, stim_w, stim_amp) -> None: ...

    def _G_stim(self, w):  # kept for interface parity; forms are not symbolic here
        raise NotImplementedError

    def solve(self, interval: tuple[float, float], dt: float | None = None) -> Results:
        """Time loop of base_model.py:250-297 -- note: no assign_previous() after the final step."""
        T0, T = interval
        if dt is None:
            dt = T - T0
        t0 = T0
        t1 = T0 + dt
        self.status = Status.OK
        while True:
            logger.info("Solving on t = (%g, %g)" % (t0, t1))
            self.step((t0, t1))
            if (t1 + dt) > (T + 1e-12):
                break
            self.assign_previous()
            t0 = t1
            t1 = t0 + dt
        return Results(state=self.state, status=self.status)
