"""``MonodomainModel`` -- interface of src/beat/monodomain_model.py:18-98.

Solves  C_m dv/dt = div(M grad v) + I_stim  with the theta-rule weak form of :68-98 on P1
elements.  ``M`` is a scalar, a ``Constant``, a constant (dim, dim) tensor or a per-cell tensor field
(``grid.CellField`` / array of shape (ncells, dim, dim), see beat.conductivities.define_conductivity_tensor).
Constant tensors on an unmasked box use the 27-type stencil tables; per-cell tensors and voxel-masked
meshes use per-node stencil rows."""

from __future__ import annotations

import logging

from . import grid
from ._engine import DiffusionSolver, build_ops, conductivity_array
from .base_model import BaseModel

logger = logging.getLogger(__name__)


class MonodomainModel(BaseModel):
    def __init__(self, time, mesh, M, I_s=None, params=None, C_m: float = 1.0, dx=None, **kwargs) -> None:
        self._M = M
        self.C_m = grid.Constant(mesh, C_m)
        super().__init__(mesh=mesh, time=time, params=params, I_s=I_s, dx=dx, **kwargs)

    def _setup_state_space(self) -> None:
        k = self.parameters["degree"]
        family = self.parameters["family"]
        self.V = grid.FunctionSpace(self._mesh, family, k)
        self.v_ = grid.Function(self.V, name="v_")
        self._state = grid.Function(self.V, name="v")

    def _setup_operators(self) -> None:
        mesh = self._mesh
        self._ops = build_ops(self._ctx, mesh, conductivity_array(self._M, mesh))
        self._diffusion = DiffusionSolver(self._ops, mesh.slab, group=mesh.comm.group)

    @property
    def state(self) -> grid.Function:
        return self._state

    def assign_previous(self):
        self.v_.x.array[:] = self.state.x.array

    @staticmethod
    def default_parameters():
        params = super(MonodomainModel, MonodomainModel).default_parameters()
        params["use_custom_preconditioner"] = True
        return params

    def _solve_linear(self, stim_w, stim_amp) -> None:
        rtol, atol, max_it = self._solver_tolerances()
        x = self._state.writable_field()
        self.ksp = self._diffusion.solve(self.v_.field, stim_w, stim_amp, x, rtol=rtol, atol=atol, max_it=max_it)
        self._state._touch()

    def can_solve_lazily(self) -> bool:
        """May the fused step leave its solve open (enqueued, not waited for) until the next step's ionic launch is in the
        queue behind it?  Jacobi on one rank, or on a decomposed grid whose solve the library runs itself (a LibComm: every rank
        sees the same all-reduced scalars on the device, so every rank's launch behind the solve does the same); nobody who
        wants the KSP record step by step (a monitor; PETSc's ``ksp_error_if_not_converged``, whose exception belongs to the
        failing step); BEAT_LAZY_KSP=0 switches it off, BEAT_LAZY_KSP_DIST=0 on decomposed grids only."""
        import os

        from .telemetry import NullMonitor

        d, ops = self._diffusion, self._ops
        if os.environ.get("BEAT_LAZY_KSP", "1") == "0" or not hasattr(ops, "can_open"):
            return False
        if type(self.monitor) is not NullMonitor:
            return False
        if (self.parameters.get("petsc_options") or {}).get("ksp_error_if_not_converged"):
            return False
        if d.libcomm is not None:  # the in-library decomposed solve (Jacobi: the polynomial preconditioner is stage-driven)
            return os.environ.get("BEAT_LAZY_KSP_DIST", "1") != "0" and ops.pc_num_passes == 0
        if d.dist is not None:
            return False
        return ops.can_open()

    def solve_in_place(self, field, stim_w, stim_amp, defer_flush: bool = False, lazy: bool = False):
        """Fused-step entry: v_ and the unknown share ``field`` (the V row of the ODE state array).  With
        ``defer_flush`` the last update of the potential may stay pending in ``self._ops`` (see HipOps.solve_single).
        ``lazy`` (with defer_flush, when ``can_solve_lazily``): the solve is enqueued and NOT waited for; ``self.ksp`` --
        read when somebody asks, or when the next step has put its ionic kernel behind the solve -- finishes it."""
        rtol, atol, max_it = self._solver_tolerances()
        if lazy and defer_flush:
            ops = self._ops
            ops.on_finish = self._solve_finished
            ops.solve_begin(field, stim_w, stim_amp, field, rtol, atol, max_it, comm=self._diffusion.libcomm)
            return None
        self.ksp = self._diffusion.solve(field, stim_w, stim_amp, field, rtol=rtol, atol=atol, max_it=max_it,
                                         defer_flush=defer_flush)
        self._check_converged()
        return self.ksp

    def _solve_finished(self, res) -> None:
        self._ksp = res
        self._check_converged()
