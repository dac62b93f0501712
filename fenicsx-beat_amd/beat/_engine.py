"""Slab-decomposed diffusion solve: halo exchange + Jacobi-PCG orchestration.

One process per GPU.  The global structured grid is cut into z-slabs (slowest index); rank g
owns planes ``[z0, z1)`` plus one ghost plane on either side of every field.  Per SpMV the
two boundary planes of ``p`` travel to the neighbouring ranks (``torch.distributed``
point-to-point = RCCL send/recv over xGMI on the GPU box, gloo in the CPU tests); per PCG
iteration two small all-reduces combine the dot products.  All arithmetic is done by an
``ops`` object: :class:`HipOps` (libbeat_hip.so through the C ABI) in the product; the CPU
tests substitute an oracle-backed implementation to exercise exactly this orchestration code
with world_size 2 on gloo.

Replaces, for the decomposed case, ``b.ghostUpdate`` / ``KSP.solve`` / ``scatter_forward`` of
src/beat/base_model.py:203-242 (MPI neighbour exchange + PETSc-internal all-reduces).
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Any

import numpy as np

from . import _hip


@dataclass
class Slab:
    """z-range owned by one rank.  Planes are dealt evenly, or -- with per-plane ``weights`` (tissue nodes per
    plane of a voxelised geometry) -- so that every rank gets about the same total weight."""

    nz_global: int
    rank: int = 0
    world: int = 1
    weights: Any = None

    def __post_init__(self):
        if self.world > self.nz_global:
            raise ValueError(f"{self.world} ranks for only {self.nz_global} z-planes")
        if self.weights is not None and self.world > 1:
            w = np.asarray(self.weights, dtype=np.float64) + 1e-9  # every plane costs something
            if len(w) != self.nz_global:
                raise ValueError("one weight per z-plane expected")
            cum = np.cumsum(w)
            cuts = [0]
            for r in range(1, self.world):
                k = int(np.argmin(np.abs(cum - cum[-1] * r / self.world))) + 1  # planes [0, k) closest to the share
                k = max(k, cuts[-1] + 1)                      # at least one plane per rank ...
                k = min(k, self.nz_global - (self.world - r))  # ... also for the ranks still to come
                cuts.append(k)
            cuts.append(self.nz_global)
            counts = [cuts[r + 1] - cuts[r] for r in range(self.world)]
        else:
            base, extra = divmod(self.nz_global, self.world)
            counts = [base + (1 if r < extra else 0) for r in range(self.world)]
        self.counts = counts
        self.z0 = sum(counts[: self.rank])
        self.z1 = self.z0 + counts[self.rank]

    @property
    def nz(self) -> int:
        return self.z1 - self.z0

    @property
    def lo_phys(self) -> bool:
        return self.rank == 0

    @property
    def hi_phys(self) -> bool:
        return self.rank == self.world - 1


@dataclass
class KspResult:
    """What PerformanceMonitor.record_ksp reads from a PETSc KSP (telemetry.py:67-76)."""

    iterations: int = 0
    residual_norm: float = 0.0
    converged_reason: int = 0
    rhs_norm: float = 0.0

    def getIterationNumber(self):
        return self.iterations

    def getResidualNorm(self):
        return self.residual_norm

    def getConvergedReason(self):
        return self.converged_reason


def chebyshev_coefficients(m: int, lmin: float, lmax: float) -> np.ndarray:
    """Coefficients c_0..c_{m-1} of the polynomial preconditioner z = sum_k c_k B^k rhat (B = D^-1 A,
    rhat = D^-1 r) given by m steps of the Chebyshev iteration for B z = rhat on [lmin, lmax] started
    from zero.  m = 1 is Jacobi (scaled by 2/(lmin+lmax), irrelevant for CG).  The polynomial is positive
    on (0, lmax], so the preconditioner stays SPD as long as lmax bounds the spectrum from above."""
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    rho = 1.0 / sigma

    def pad(a, n):
        return np.pad(a, (0, n - len(a)))

    def add(a, b):
        n = max(len(a), len(b))
        return pad(a, n) + pad(b, n)

    d = np.array([1.0 / theta])
    z = d.copy()
    res = add(np.array([1.0]), -np.concatenate([[0.0], z]))
    for _ in range(1, m):
        rho_new = 1.0 / (2.0 * sigma - rho)
        d = add(rho_new * rho * d, (2.0 * rho_new / delta) * res)
        z = add(z, d)
        res = add(np.array([1.0]), -np.concatenate([[0.0], z]))
        rho = rho_new
    return pad(z, m)


def spectrum_bounds(A_tab: np.ndarray, ratio: float = 5.0) -> tuple[float, float]:
    """(lmin, lmax) estimates for D^-1 A from the 27x15 coefficient table: lmax = largest Gershgorin
    row bound sum|a_k|/a_0 (a true upper bound); lmin = lmax/ratio (kappa(D^-1 Mass) <= 5 for P1
    tetrahedra, and the stiffness part only adds to the top of the spectrum)."""
    rows = A_tab[np.abs(A_tab[:, 0]) > 0]
    lmax = float((np.abs(rows).sum(axis=1) / rows[:, 0]).max())
    return lmax / ratio, lmax


class HipOps:
    """The product compute backend: every method is one C-ABI call into libbeat_hip.so."""

    default_small = True  # small grids: whole solve in one launch (tests flip this to exercise the multi-launch kernels)

    def __init__(self, ctx, shape_local, lo_phys, hi_phys, mass_tab, stiff_tab, per_node: bool = False):
        """mass_tab / stiff_tab: the (27, 15) tables of _stencil.stencil_tables, or with ``per_node`` the
        (15, n) rows of _stencil.stencil_fields (voxel masks, spatially varying conductivity)."""
        self.ctx = ctx
        self.lib = ctx.lib
        nx, ny, nz = (int(v) for v in shape_local)
        self.shape = (nx, ny, nz)
        self.plane = nx * ny
        self.n = nx * ny * nz
        n3 = (C.c_int64 * 3)(nx, ny, nz)
        host = isinstance(mass_tab, np.ndarray) or not per_node
        mt = np.ascontiguousarray(mass_tab, dtype=np.float64) if host else None
        kt = np.ascontiguousarray(stiff_tab, dtype=np.float64) if host else None
        handle = C.c_void_p()
        self.per_node = bool(per_node)
        if self.per_node:
            if tuple(mass_tab.shape) != (15, self.n) or tuple(stiff_tab.shape) != (15, self.n):
                raise ValueError(f"per-node rows must have shape (15, {self.n}), got {mass_tab.shape} / {stiff_tab.shape}")
            # device copies stay alive with this object: the handle borrows them
            if isinstance(mass_tab, np.ndarray):
                self._mass_dev, self._stiff_dev = ctx.from_numpy(mt), ctx.from_numpy(kt)
            else:  # already on the device (from_voxels)
                self._mass_dev, self._stiff_dev = mass_tab, stiff_tab
            _hip.check(
                self.lib.beat_pde_create_var(ctx.handle, n3, int(lo_phys), int(hi_phys),
                                             C.c_void_p(self._mass_dev.data_ptr()),
                                             C.c_void_p(self._stiff_dev.data_ptr()), self.n, C.byref(handle))
            )
            mt = kt = None  # the host copies are not needed any more
        else:
            _hip.check(
                self.lib.beat_pde_create(ctx.handle, n3, int(lo_phys), int(hi_phys), mt.ctypes.data_as(C.c_void_p),
                                         kt.ctypes.data_as(C.c_void_p), C.byref(handle))
            )
        self.handle = handle
        self.mass_tab, self.stiff_tab = mt, kt
        nfields = max(4, int(self.lib.beat_pde_work_fields(handle)))  # see beat_pde_solve
        fld = int(self.lib.beat_pde_field_stride(handle))  # a field with its ghost planes (+ padding against channel aliasing)
        self.work_placement = None
        self.work = self._place_work(ctx, nfields, fld)
        from ._device import Field

        # same layout as beat_pde_solve: r, q, z, ring[...]; p is ring[0] for the in-place recurrences
        self.fld = fld
        self.r = Field(ctx, self.n, self.plane, buf=self.work, offset=self.plane)
        self.q = Field(ctx, self.n, self.plane, buf=self.work, offset=self.plane + fld)
        self.z = Field(ctx, self.n, self.plane, buf=self.work, offset=self.plane + 2 * fld)
        self.ring = [Field(ctx, self.n, self.plane, buf=self.work, offset=self.plane + (3 + j) * fld)
                     for j in range(nfields - 3)]
        self.p = self.ring[0]
        self.st = ctx.zeros(_hip.ST_SIZE)
        self.pending = None            # (field, ring_base, count) of a deferred potential update
        self.st_ptr_for_flush = None   # scalar state the pending update belongs to (None: the handle's own)
        # a solve that is enqueued and not yet looked at (solve_begin): the field it works on; its record arrives with solve_finish,
        # which `on_finish` (the PDE model: its .ksp, its status) and `ksp_log` (a list, when a caller wants every record) receive
        self.open_x = None
        self.on_finish = None
        self.ksp_log = None
        self.pc_degree = 1
        self._coeffs = (1.0, 0.5, 0.0)
        self.guess_order = 0  # x0 = v_ unless asked for (set_guess_order; BaseModel asks for "auto" by default)
        if not type(self).default_small:
            self.set_small(False)

    @classmethod
    def from_voxels(cls, ctx, dim, cells, h, M, active, shape_local, z0, lo_phys, hi_phys):
        """Per-node operators assembled ON THE DEVICE from per-voxel data (beat_pde_assemble_rows).
        cells: global voxels per axis (length dim); M: (dim, dim) or (nvoxels, dim, dim); active: bool
        (nvoxels,) or None; shape_local / z0: this rank's slab of nodes."""
        from . import _stencil

        nvox = int(np.prod(cells))
        c3 = [int(v) for v in cells] + [1] * (3 - dim)
        T, Me = _stencil.voxel_element_tensors(dim, h)
        M = np.asarray(M, dtype=np.float64)
        if M.ndim == 3 and M.shape != (nvox, dim, dim):
            raise ValueError(f"per-voxel conductivity has shape {M.shape}, expected ({nvox}, {dim}, {dim})")
        if active is not None:
            active = np.asarray(active, dtype=bool).ravel()
            if active.size != nvox:
                raise ValueError(f"active mask has {active.size} entries for {nvox} voxels")
        # a slab only needs the voxel layers that touch its node planes: upload those, and let the kernel see a voxel
        # grid that starts at the first of them
        z0 = int(z0)
        if dim == 3:
            lo, hi = max(z0 - 1, 0), min(z0 + int(shape_local[2]), c3[2])
            if (lo, hi) != (0, c3[2]):
                per_layer = c3[0] * c3[1]
                if M.ndim == 3:
                    M = M[lo * per_layer : hi * per_layer]
                if active is not None:
                    active = active[lo * per_layer : hi * per_layer]
                c3[2], z0 = hi - lo, z0 - lo
                nvox = per_layer * c3[2]
        m_dev, m_const = None, None
        if M.ndim == 3:
            if dim == 3:
                M9 = M  # already (nvox, 3, 3): no padded copy
            else:
                M9 = np.zeros((nvox, 3, 3))
                M9[:, :dim, :dim] = M
            m_dev = ctx.from_numpy(M9.reshape(nvox, 9))
        else:
            m_const = np.zeros((3, 3))
            m_const[:dim, :dim] = _stencil.conductivity_matrix(M, dim)
        a_dev = None
        if active is not None:
            a_dev = ctx.from_numpy(active.astype(np.uint8))
        nx, ny, nz = (int(v) for v in shape_local)
        n = nx * ny * nz
        mass = ctx.torch.empty((15, n), dtype=ctx.torch.float64, device=ctx.device)
        stiff = ctx.torch.empty((15, n), dtype=ctx.torch.float64, device=ctx.device)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        _hip.check(ctx.lib.beat_pde_assemble_rows(
            ctx.handle, (C.c_int64 * 3)(nx, ny, nz), (C.c_int64 * 3)(*c3), int(z0),
            T.ctypes.data_as(C.c_void_p), Me.ctypes.data_as(C.c_void_p), ptr(m_dev),
            None if m_const is None else m_const.ctypes.data_as(C.c_void_p), ptr(a_dev), ptr(mass), ptr(stiff), n))
        return cls(ctx, shape_local, lo_phys, hi_phys, mass, stiff, per_node=True)

    # -- field helpers ----------------------------------------------------------------------
    def new_field(self):
        return self.ctx.field(self.n, self.plane)

    def read_state(self):
        """Synchronising read of the PCG scalar state."""
        return self.st.cpu().numpy()

    # -- stages -----------------------------------------------------------------------------
    def set_timestep(self, C_m, theta, dt):
        self.flush_pending()
        _hip.check(self.lib.beat_pde_set_timestep(self.handle, float(C_m), float(theta), float(dt)))
        self._coeffs = (float(C_m), float(theta), float(dt))
        self._update_preconditioner()

    def set_preconditioner(self, degree: int):
        """degree 1: Jacobi; m >= 2: Chebyshev polynomial preconditioner with m terms (m-1 stencil passes)."""
        if not 1 <= int(degree) <= 8:
            raise ValueError("preconditioner degree must be in 1..8")
        if self.per_node and int(degree) > 1:
            raise ValueError("the polynomial preconditioner is not available with per-node coefficients")
        self.pc_degree = int(degree)
        self._update_preconditioner()

    def _update_preconditioner(self):
        m = self.pc_degree
        if m <= 1:
            _hip.check(self.lib.beat_pde_set_preconditioner(self.handle, 1, None))
            return
        C_m, theta, dt = self._coeffs
        lmin, lmax = spectrum_bounds(C_m * self.mass_tab + theta * dt * self.stiff_tab)
        coef = np.ascontiguousarray(chebyshev_coefficients(m, lmin, lmax))
        _hip.check(self.lib.beat_pde_set_preconditioner(self.handle, m, coef.ctypes.data_as(C.c_void_p)))

    # polynomial preconditioner stages (see include/beat_hip.h)
    @property
    def pc_num_passes(self) -> int:
        return self.pc_degree - 1

    def pc_io(self, j):
        """(input field, output field) of Horner pass j: outputs alternate q / z and end in z."""
        n = self.pc_num_passes
        out = lambda jj: self.z if (n - 1 - jj) % 2 == 0 else self.q  # noqa: E731
        return (self.r if j == 0 else out(j - 1)), out(j)

    def pc_pass(self, j, slot):
        stp = self.st.data_ptr()
        _hip.check(self.lib.beat_pde_pc_pass(self.handle, j, self.r.ptr, self.z.ptr, self.q.ptr, C.c_void_p(stp),
                                             C.c_void_p(stp + 8 * slot)))

    def cg_first_z(self):
        _hip.check(self.lib.beat_pde_cg_first_z(self.handle, C.c_void_p(self.st.data_ptr()), self.z.ptr, self.p.ptr))

    def cg_next_z(self):
        _hip.check(self.lib.beat_pde_cg_next_z(self.handle, C.c_void_p(self.st.data_ptr()), self.z.ptr, self.p.ptr))

    @staticmethod
    def _stim_args(stim_w, stim_amp):
        k = len(stim_w)
        ptrs = (C.c_void_p * max(1, k))()
        amps = (C.c_double * max(1, k))()
        for i, (w, a) in enumerate(zip(stim_w, stim_amp)):
            ptrs[i] = w.ptr
            amps[i] = float(a)
        return ptrs, amps, k

    def rhs(self, v_prev, stim_w, stim_amp, x):
        ptrs, amps, k = self._stim_args(stim_w, stim_amp)
        _hip.check(self.lib.beat_pde_rhs(self.handle, v_prev.ptr, ptrs, amps, k, x.ptr, self.r.ptr, self.p.ptr,
                                         C.c_void_p(self.st.data_ptr())))

    def cg_begin(self, rtol, atol, max_it):
        _hip.check(self.lib.beat_pde_cg_begin(self.handle, C.c_void_p(self.st.data_ptr()), rtol, atol, max_it))

    def spmv_dot(self):
        _hip.check(self.lib.beat_pde_spmv_dot(self.handle, self.p.ptr, self.q.ptr, C.c_void_p(self.st.data_ptr())))

    def spmv_interior(self, p=None):
        """q = A p on the planes that need no ghost data (runs while the halo exchange is in flight)."""
        p = p or self.p
        _hip.check(self.lib.beat_pde_spmv_dot_part(self.handle, p.ptr, self.q.ptr, C.c_void_p(self.st.data_ptr()), 0))

    def spmv_boundary(self, p=None):
        """q = A p on the slab-boundary planes, then the local p.q."""
        p = p or self.p
        _hip.check(self.lib.beat_pde_spmv_dot_part(self.handle, p.ptr, self.q.ptr, C.c_void_p(self.st.data_ptr()), 1))

    # deferred-x stages (see include/beat_hip.h)
    def cg_update_r(self, slot):
        _hip.check(self.lib.beat_pde_cg_update_r(self.handle, C.c_void_p(self.st.data_ptr()), self.r.ptr, self.q.ptr, slot))

    def cg_next_oop(self, p_cur, p_next):
        _hip.check(self.lib.beat_pde_cg_next_oop(self.handle, C.c_void_p(self.st.data_ptr()), self.r.ptr, p_cur.ptr,
                                                 p_next.ptr))

    def x_flush(self, x, ring_base, only_if_full):
        _hip.check(self.lib.beat_pde_x_flush(self.handle, C.c_void_p(self.st.data_ptr()), x.ptr, self.ring[0].ptr,
                                             self.fld, int(ring_base), int(only_if_full)))

    def cg_update(self, x):
        _hip.check(self.lib.beat_pde_cg_update(self.handle, C.c_void_p(self.st.data_ptr()), x.ptr, self.r.ptr,
                                               self.p.ptr, self.q.ptr))

    def cg_next(self):
        _hip.check(self.lib.beat_pde_cg_next(self.handle, C.c_void_p(self.st.data_ptr()), self.r.ptr, self.p.ptr))

    def solve_single(self, v_prev, stim_w, stim_amp, x, rtol, atol, max_it, defer_flush: bool = False) -> KspResult:
        """With ``defer_flush`` the last partially filled cycle of search directions is left unapplied and recorded
        in ``self.pending`` = (field, ring_base, count): the next ionic kernel adds it (beat_ode_step_pending) or
        ``flush_pending`` does."""
        self.flush_pending()
        self.st_ptr_for_flush = None  # a pending update of this solve belongs to the handle's own scalar state
        ptrs, amps, k = self._stim_args(stim_w, stim_amp)
        info = _hip.KspInfo()
        pend = (C.c_int * 2)()
        _hip.check(self.lib.beat_pde_solve_ex(self.handle, v_prev.ptr, ptrs, amps, k, x.ptr,
                                              C.c_void_p(self.work.data_ptr()), rtol, atol, max_it, int(defer_flush),
                                              C.byref(info), pend), allow_not_converged=True)
        if pend[1] > 0 or self.lib.beat_pde_guess_pending(self.handle):  # (the guess increment alone may be due)
            self.pending = (x, int(pend[0]), int(pend[1]))
        res = KspResult(info.iterations, info.residual_norm, info.converged_reason, info.rhs_norm)
        if self.ksp_log is not None:
            self.ksp_log.append(res)
        return res

    # -- the solve in two halves (beat_pde_solve_begin / _end): see include/beat_hip.h ------------------------------------
    def can_open(self) -> bool:
        return bool(self.lib.beat_pde_solve_can_open(self.handle))

    def solve_begin(self, v_prev, stim_w, stim_amp, x, rtol, atol, max_it, comm: "LibComm | None" = None) -> None:
        """Enqueue the solve and return without waiting.  Until ``solve_finish`` -- which everything that needs the result or
        the field calls: ``flush_pending``, the model's ``ksp``, the next ionic step -- the device works and the host is free.
        ``comm``: the slab-decomposed solve (beat_pde_solve_dist_begin), its exchanges and all-reduces enqueued with it."""
        self.flush_pending()
        self.st_ptr_for_flush = None
        ptrs, amps, k = self._stim_args(stim_w, stim_amp)
        if comm is not None:
            _hip.check(self.lib.beat_pde_solve_dist_begin(self.handle, comm.handle, v_prev.ptr, ptrs, amps, k, x.ptr,
                                                          C.c_void_p(self.work.data_ptr()), rtol, atol, max_it))
        else:
            _hip.check(self.lib.beat_pde_solve_begin(self.handle, v_prev.ptr, ptrs, amps, k, x.ptr, C.c_void_p(self.work.data_ptr()),
                                                     rtol, atol, max_it))
        self.open_x = x

    def _record(self, info) -> KspResult:
        res = KspResult(info.iterations, info.residual_norm, info.converged_reason, info.rhs_norm)
        if self.ksp_log is not None:
            self.ksp_log.append(res)
        if self.on_finish is not None:
            self.on_finish(res)
        return res

    def _place_work(self, ctx, nfields: int, fld: int):
        """The PCG's work fields (r, q, z, the ring): like the state array (``StateArray._place_buffer``, profiles/r06_placement.md) a
        buffer of 1 GB or more is the best of BEAT_WORK_PLACE (default 3; 1: off) allocations by the library's streaming probe over
        four of the fields -- the ring's directions are read by the ionic kernel as well (the pending update): ten bench processes,
        alternating, 12.10 - 12.25 -> 12.03 - 12.11 ms per 512^3 step, developed front 14.19 - 14.49 -> 13.96 - 14.26
        (profiles/r06_ab_work_place.txt)."""
        import os

        tries = int(os.environ.get("BEAT_WORK_PLACE", "3"))
        numel = nfields * fld
        if tries <= 1 or 8 * numel < int(os.environ.get("BEAT_STATE_PLACE_MIN_BYTES", str(1 << 30))) or (fld & 1) or (self.plane & 1):
            return ctx.zeros(numel)
        torch = ctx.torch

        def rate(buf):
            ptr = C.c_void_p(buf.data_ptr() + 8 * self.plane)
            ts = []
            for it in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(ctx.stream)
                _hip.check(self.lib.beat_stream_probe(ctx.handle, ptr, self.n & ~1, 4, 3, 1, 0, 4, fld))
                b.record(ctx.stream)
                b.synchronize()
                if it:
                    ts.append(a.elapsed_time(b))
            return 2.0 * 4 * (self.n & ~1) * 8 / (sorted(ts)[len(ts) // 2] * 1e6)

        cands, rates = [], []
        for _ in range(tries):
            free_b, _t = torch.cuda.mem_get_info(ctx.device)
            if cands and free_b < 2 * 8 * numel:
                break
            cands.append(ctx.zeros(numel))
            rates.append(rate(cands[-1]))
        best = max(range(len(cands)), key=lambda j: rates[j])
        buf = cands[best]
        self.work_placement = {"candidates": [round(r, 1) for r in rates], "chosen": best, "unit": "GB/s"}
        del cands
        torch.cuda.empty_cache()
        return buf

    def solve_finish(self):
        """Wait for the open solve (more iterations are enqueued if it needs them), take its record; what it leaves pending
        goes to ``self.pending`` as after ``solve_single(defer_flush=True)``.  None when no solve is open."""
        if self.open_x is None:
            return None
        x, self.open_x = self.open_x, None
        info = _hip.KspInfo()
        pend = (C.c_int * 2)()
        _hip.check(self.lib.beat_pde_solve_end(self.handle, C.byref(info), pend), allow_not_converged=True)
        if pend[1] > 0 or self.lib.beat_pde_guess_pending(self.handle):
            self.pending = (x, int(pend[0]), int(pend[1]))
        return self._record(info)

    def finished_behind(self):
        """The open solve was finished inside the ionic call that was enqueued behind it (beat_ode_step_* with pending = -1):
        collect its record; nothing is pending."""
        self.open_x = None
        self.pending = None
        info = _hip.KspInfo()
        pend = (C.c_int * 2)()
        _hip.check(self.lib.beat_pde_solve_end(self.handle, C.byref(info), pend), allow_not_converged=True)  # (no solve open: the last record)
        return self._record(info)

    def solve_dist(self, comm: "LibComm", v_prev, stim_w, stim_amp, x, rtol, atol, max_it, defer_flush: bool = False) -> KspResult:
        """The slab-decomposed solve as ONE C call (beat_pde_solve_dist): halo exchange and all-reduces are issued
        by the library on ``comm``; same deferred-update contract as solve_single."""
        self.flush_pending()
        self.st_ptr_for_flush = None
        ptrs, amps, k = self._stim_args(stim_w, stim_amp)
        info = _hip.KspInfo()
        pend = (C.c_int * 2)()
        _hip.check(self.lib.beat_pde_solve_dist(self.handle, comm.handle, v_prev.ptr, ptrs, amps, k, x.ptr,
                                                C.c_void_p(self.work.data_ptr()), rtol, atol, max_it, int(defer_flush),
                                                C.byref(info), pend), allow_not_converged=True)
        if pend[1] > 0 or self.lib.beat_pde_guess_pending(self.handle):
            self.pending = (x, int(pend[0]), int(pend[1]))
        res = KspResult(info.iterations, info.residual_norm, info.converged_reason, info.rhs_norm)
        if self.ksp_log is not None:
            self.ksp_log.append(res)
        return res

    def set_guess_order(self, order: int) -> None:
        """0: every solve starts from x0 = v_; m = 1..4: from v_ plus the degree-(m-1) extrapolation in time of the last
        m diffusion increments; "auto" (-1): quadratic or cubic, whichever has been leaving the smaller initial residual
        (beat_pde_set_guess_order).  Drops the history."""
        if isinstance(order, str):
            if order != "auto":
                raise ValueError(f"guess order must be an integer 0..4 or 'auto', got {order!r}")
            order = -1
        self.flush_pending()
        _hip.check(self.lib.beat_pde_set_guess_order(self.handle, int(order)))
        self.guess_order = int(order)

    def set_single_reduction(self, on) -> None:
        """PETSc's ``-ksp_cg_single_reduction`` for the slab-decomposed solve: True one all-reduce per PCG iteration, False two,
        None what BEAT_DIST_MERGED says (beat_pde_set_single_reduction)."""
        _hip.check(self.lib.beat_pde_set_single_reduction(self.handle, -1 if on is None else int(bool(on))))

    def set_small(self, enable: bool) -> None:
        """Grids of a few thousand nodes are solved in one launch of one workgroup (beat_pde_small.hip); False keeps
        this operator on the multi-launch kernels (tests of those kernels on small grids, deferral semantics)."""
        _hip.check(self.lib.beat_pde_set_small_grid_solve(self.handle, int(bool(enable))))

    def small_active(self) -> bool:
        """True when solves of this operator run as one launch of one workgroup (and beat_split_steps accepts it)."""
        return bool(self.lib.beat_pde_small_grid_solve_active(self.handle))

    def guess_traffic(self) -> dict:
        """Fields the last solve's x update reads / writes for the initial guess (beat_pde_guess_traffic)."""
        out = (C.c_int * 4)()
        _hip.check(self.lib.beat_pde_guess_traffic(self.handle, out))
        return {"reads": int(out[0]), "writes": int(out[1]), "order": int(out[2]), "pending": bool(out[3])}

    def guess_reset(self) -> None:
        """Forget the recorded increments (the potential was overwritten: the next solve starts from x0 = v_)."""
        self.flush_pending()
        _hip.check(self.lib.beat_pde_guess_reset(self.handle))

    def flush_pending(self) -> None:
        """Apply a deferred update of the potential (no-op when nothing is pending).  A solve that is still open is finished first."""
        if self.open_x is not None:
            self.solve_finish()
        if self.pending is not None:
            x, ring_base, _ = self.pending
            self.pending = None
            self.flushes = getattr(self, "flushes", 0) + 1  # separate passes taken (the fused step should take none)
            _hip.check(self.lib.beat_pde_x_flush(self.handle, C.c_void_p(self.st_ptr_for_flush), x.ptr, self.ring[0].ptr,
                                                 self.fld, ring_base, 0))

    def apply(self, which, x, y):
        _hip.check(self.lib.beat_pde_apply(self.handle, which, x.ptr, y.ptr))


def conductivity_array(M, mesh) -> np.ndarray:
    """(dim, dim) for a constant tensor, (ncells, dim, dim) for a per-cell field (grid.CellField or array)."""
    from . import _stencil, grid

    if isinstance(M, (grid.Function, grid.VectorFunction)):
        raise TypeError("the conductivity is a tensor: build it from a fibre field with "
                        "beat.conductivities.define_conductivity_tensor (vector P1 functions and per-cell fields are accepted)")
    if isinstance(M, grid.CellField):
        M = M.values
    if isinstance(M, grid.Constant):
        M = M.value
    M = np.asarray(M, dtype=np.float64)
    if M.ndim == 3:
        d = mesh.dim
        if M.shape[1:] != (d, d):
            raise ValueError(f"per-cell conductivity has shape {M.shape}, expected (ncells, {d}, {d})")
        return M
    return _stencil.conductivity_matrix(M, mesh.dim)


def build_ops(ctx, mesh, M: np.ndarray) -> "HipOps":
    """Mass / stiffness operators of ``mesh`` with conductivity ``M`` (see conductivity_array): the 27-type
    stencil tables for a constant tensor on an unmasked box, per-node rows otherwise (assembled on the device
    when the data are per box cell, on the host when they are per simplex)."""
    from . import _stencil

    slab = mesh.slab
    if M.ndim == 2 and mesh.active is None:
        mass_tab, stiff_tab = _stencil.stencil_tables(mesh.dim, mesh.h, M)
        if getattr(mesh, "kernel_y_as_z", False):  # a 2-D mesh cut into slabs of rows: the kernels see (nx, 1, ny_local)
            mass_tab, stiff_tab = _stencil.tables_y_as_z(mass_tab), _stencil.tables_y_as_z(stiff_tab)
        return HipOps(ctx, mesh.shape_local, slab.lo_phys, slab.hi_phys, mass_tab, stiff_tab)
    if getattr(mesh, "kernel_y_as_z", False):
        # a 2-D mesh cut into slabs of ROWS (the reference's unit-square tests under `mpirun -n 2`,
        # .github/workflows/main-mpi.yml:33, tests/test_monodomain_solver.py:33-216) with per-cell tensors or a cell mask:
        # the rows of the whole (small) 2-D operator, this rank's rows of nodes cut out of them, the coefficients moved to
        # the slots the kernels' (nx, 1, ny_local) grid reads them from
        mass, stiff = _stencil.stencil_fields(2, mesh.n, mesh.h, M, mesh.active)
        nxn = mesh.shape_local[0]
        cut = slice(slab.z0 * nxn, slab.z1 * nxn)
        return HipOps(ctx, mesh.shape_local, slab.lo_phys, slab.hi_phys, _stencil.fields_y_as_z(mass[:, cut]),
                      _stencil.fields_y_as_z(stiff[:, cut]), per_node=True)
    per_voxel = (M.ndim == 2 or M.shape[0] == mesh.num_box_cells) and (mesh.active is None or mesh.active_box is not None)
    if per_voxel:
        return HipOps.from_voxels(ctx, mesh.dim, mesh.n, mesh.h, M, mesh.active_box, mesh.shape_local, slab.z0,
                                  slab.lo_phys, slab.hi_phys)
    z_range = (slab.z0, slab.z1) if mesh.dim == 3 else None
    mass, stiff = _stencil.stencil_fields(mesh.dim, mesh.n, mesh.h, M, mesh.active, z_range=z_range)
    return HipOps(ctx, mesh.shape_local, slab.lo_phys, slab.hi_phys, mass, stiff, per_node=True)


class _HostStagedDist:
    """``torch.distributed`` on a backend without device-tensor point-to-point (gloo) for device fields: messages and
    all-reduces go through host copies.  A rehearsal transport only -- it lets several ranks share ONE GPU, where RCCL
    (one rank per device) cannot run, so that the multi-process path can be exercised end to end on a one-GPU box
    (``BEAT_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2``).  The product
    transport is RCCL."""

    class _Request:
        def __init__(self, work, device_dst=None, host=None):
            self.work, self.device_dst, self.host = work, device_dst, host

        def wait(self):
            self.work.wait()
            if self.device_dst is not None:
                self.device_dst.copy_(self.host)

    def __init__(self, dist):
        self._dist = dist
        self.ReduceOp = dist.ReduceOp
        self.isend, self.irecv = dist.isend, dist.irecv
        self.get_global_rank = dist.get_global_rank

    def P2POp(self, fn, tensor, peer, group=None):
        return (fn, tensor, peer, group)

    def batch_isend_irecv(self, ops):
        import torch

        reqs = []
        for fn, tensor, peer, group in ops:
            if fn is self.isend:
                host = tensor.cpu()
                reqs.append(self._Request(self._dist.isend(host, peer, group=group), host=host))
            else:
                host = torch.empty(tensor.shape, dtype=tensor.dtype)
                reqs.append(self._Request(self._dist.irecv(host, peer, group=group), tensor, host))
        return reqs

    def all_reduce(self, t, op=None, group=None):
        host = t.cpu()
        self._dist.all_reduce(host, op=op if op is not None else self.ReduceOp.SUM, group=group)
        t.copy_(host)


def exchange_ghost_planes(field, slab: Slab, group=None) -> None:
    """Fill the ghost planes of ``field`` with the neighbouring slabs' boundary planes over ``torch.distributed`` (device
    tensors on nccl = RCCL, host-staged on any other backend): the reference's ``scatter_forward`` for a function that
    something other than the diffusion solve is about to read across the slab boundary (the interpolation of the
    potential onto a P2 / DG1 ODE space, utils.local_project)."""
    if slab.world <= 1:
        return
    import torch.distributed as dist

    d = dist if dist.get_backend(group) == "nccl" else _HostStagedDist(dist)
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    plane, ops = field.plane, []
    if not slab.lo_phys:
        ops.append(d.P2POp(d.isend, field.data[:plane], peer(slab.rank - 1), group))
        ops.append(d.P2POp(d.irecv, field.ghost_lo, peer(slab.rank - 1), group))
    if not slab.hi_phys:
        ops.append(d.P2POp(d.isend, field.data[field.n - plane:], peer(slab.rank + 1), group))
        ops.append(d.P2POp(d.irecv, field.ghost_hi, peer(slab.rank + 1), group))
    for req in (d.batch_isend_irecv(ops) if ops else []):
        req.wait()


class LibCommUnavailable(RuntimeError):
    """Raised by LibComm on EVERY rank when any rank could not set its side up (agreed collectively)."""


class LibComm:
    """``beat_comm`` of this rank (include/beat_hip.h): the transport the in-library decomposed solve uses.

    ``transport="rccl"``: RCCL communicators created by the library (ghost planes on its side stream, all-reduces
    on the compute stream; ``serial=True`` / env ``BEAT_DIST_SERIAL=1``: one communicator, one stream); rank 0's
    unique id reaches the other ranks through ``torch.distributed`` (any backend) -- that broadcast at set-up is all
    PyTorch contributes.  ``transport="ipc"``: ghost planes as interprocess device-to-device copies (mailboxes mapped
    with hipIpc*, ordered by sequence flags in device memory) and all-reduces through the same mailboxes (up to
    ``IPC_MAX_RANKS`` ranks; ``BEAT_IPC_ALLREDUCE=rccl|caller``: by RCCL / handed back to Python) -- no RCCL at all,
    and the transport that lets several processes sharing ONE GPU exchange planes on the device.
    ``transport="callbacks"``: both operations are handed back to Python and staged through the host over ``dist``
    (gloo) -- the rehearsal transport.  ``peers``: override of (peer_lo, peer_hi), used by the one-rank periodic
    self-exchange tests.

    Set-up is collective and fails collectively: every step that can fail on one rank (rank 0's unique id, the
    library calls, ``BEAT_TEST_FAIL_LIBCOMM_RANK``) is followed by an agreement over ``dist``, and no rank enters a
    collective create (``ncclCommInitRank``) unless all are ready to -- otherwise every rank raises
    :class:`LibCommUnavailable` (a rank failing alone would leave the others blocked in the next collective)."""

    def __init__(self, ctx, slab: Slab, dist=None, group=None, transport: str = "rccl", peers=None, serial: bool | None = None,
                 plane_doubles: int | None = None):
        self.ctx, self.slab, self.dist, self.group = ctx, slab, dist, group
        lib = ctx.lib
        rank, world = slab.rank, slab.world
        peer_lo, peer_hi = peers if peers is not None else (-1 if slab.lo_phys else rank - 1, -1 if slab.hi_phys else rank + 1)
        self.peer_lo, self.peer_hi = int(peer_lo), int(peer_hi)
        handle = C.c_void_p()
        self.handle = None
        self.transport = transport
        if serial is None:
            serial = os.environ.get("BEAT_DIST_SERIAL", "0") == "1"
        self.serial = bool(serial) and transport == "rccl"
        if transport not in ("rccl", "ipc", "callbacks"):
            raise ValueError(f"unknown transport {transport!r}")
        collective = world > 1 and dist is not None and getattr(dist, "is_initialized", lambda: False)()
        nccl = collective and dist.get_backend(group) == "nccl"
        err = None
        if os.environ.get("BEAT_TEST_FAIL_LIBCOMM_RANK") == str(rank):  # tests: a rank whose communicator cannot be made
            err = RuntimeError("simulated failure to create the library communicator (BEAT_TEST_FAIL_LIBCOMM_RANK)")

        def agree(stage):
            """Collective: raise on every rank if any rank holds an error."""
            nonlocal err
            failed = err is not None
            if collective:
                import torch

                flag = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64, device=ctx.device if nccl else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
                failed = float(flag.item()) > 0.0
            if failed:
                if handle:
                    lib.beat_comm_destroy(handle)
                if err is not None and not collective:
                    raise err
                raise LibCommUnavailable(f"{transport} transport, {stage}: " + (str(err) if err is not None else "failed on another rank")) from err

        ids = None
        # who sums the dot products of the ipc transport: "ipc" (the mailboxes, no RCCL at all), "rccl", "caller" (dist)
        ipc_sum = os.environ.get("BEAT_IPC_ALLREDUCE", "ipc") if transport == "ipc" else None
        if ipc_sum == "ipc" and world > _hip.IPC_MAX_RANKS:
            ipc_sum = "rccl"
        if ipc_sum == "rccl" and collective and not nccl:
            ipc_sum = "caller"  # several processes on one GPU (gloo rehearsal): RCCL cannot run there
        if ipc_sum not in (None, "ipc", "rccl", "caller"):
            raise ValueError(f"BEAT_IPC_ALLREDUCE={ipc_sum!r}: ipc, rccl or caller")
        use_rccl = transport == "rccl" or ipc_sum == "rccl"
        if use_rccl:
            ids = C.create_string_buffer(2 * _hip.UNIQUE_ID_BYTES)
            box = [None]
            if rank == 0 or not collective:
                try:
                    if err is None:
                        _hip.check(lib.beat_comm_unique_id(ids))
                        box[0] = ids.raw
                except Exception as exc:  # noqa: BLE001
                    err = exc
            if collective:  # always runs, whatever happened on rank 0: the others are waiting in it
                src = 0 if group is None else dist.get_global_rank(group, 0)
                dist.broadcast_object_list(box, src=src, group=group)
                if box[0] is None and err is None:
                    err = RuntimeError("rank 0 could not produce the RCCL unique id")
                elif box[0] is not None:
                    ids = C.create_string_buffer(box[0], 2 * _hip.UNIQUE_ID_BYTES)
            agree("unique id")
        if transport == "rccl":
            try:  # collective inside RCCL: entered only after every rank agreed it can
                _hip.check(lib.beat_comm_create_rccl_ex(ctx.handle, rank, world, self.peer_lo, self.peer_hi, ids,
                                                        _hip.COMM_SERIAL if self.serial else 0, C.byref(handle)))
            except Exception as exc:  # noqa: BLE001
                err = exc
            agree("communicator")
        elif transport == "ipc":
            if plane_doubles is None:
                raise ValueError("the ipc transport needs plane_doubles (size of one ghost plane)")
            mine = C.create_string_buffer(_hip.IPC_HANDLE_BYTES)
            self._allreduce_cb = None
            cb = None
            if ipc_sum == "caller":
                self._allreduce_cb = _hip.ALLREDUCE_FN(self._allreduce)
                cb = C.cast(self._allreduce_cb, C.c_void_p)
            try:
                if err is None:
                    _hip.check(lib.beat_comm_create_ipc(ctx.handle, rank, world, self.peer_lo, self.peer_hi, int(plane_doubles),
                                                        ids if use_rccl else None, cb, None, mine, C.byref(handle)))
            except Exception as exc:  # noqa: BLE001
                err = exc
            agree("mailbox")
            handles = {rank: mine.raw}
            if collective:
                gathered = [None] * world
                dist.all_gather_object(gathered, mine.raw, group=group)
                handles = dict(enumerate(gathered))
            try:
                if ipc_sum == "ipc":  # every rank maps every mailbox: the all-reduce writes into all of them
                    _hip.check(lib.beat_comm_ipc_connect_all(handle, b"".join(handles[r] for r in range(world)), world))
                else:
                    lo = handles.get(self.peer_lo) if self.peer_lo >= 0 and self.peer_lo != rank else None
                    hi = handles.get(self.peer_hi) if self.peer_hi >= 0 and self.peer_hi != rank else None
                    _hip.check(lib.beat_comm_ipc_connect(handle, lo, hi))
            except Exception as exc:  # noqa: BLE001
                err = exc
            agree("connect")
        else:
            self._halo_cb = _hip.HALO_FN(self._halo)          # keep the thunks alive with the object
            self._allreduce_cb = _hip.ALLREDUCE_FN(self._allreduce)
            try:
                if err is None:
                    _hip.check(lib.beat_comm_create_callbacks(ctx.handle, rank, world, self.peer_lo, self.peer_hi, self._halo_cb,
                                                              self._allreduce_cb, None, C.byref(handle)))
            except Exception as exc:  # noqa: BLE001
                err = exc
            agree("callbacks")
        self.handle = handle

    @classmethod
    def ipc_in_process(cls, ctx, slab: Slab, plane_doubles: int, registry: dict, barrier) -> "LibComm":
        """The ipc transport between ranks that are THREADS of this process (each with its own Context): every rank creates
        its mailbox, `registry[rank]` collects the communicators, `barrier` (threading.Barrier(world)) separates creating
        from connecting (beat_comm_ipc_connect_local: nothing is exported or opened).  The rehearsal of the target shape --
        8 ranks, 16 for the all-reduce -- on a box that admits six GPU processes (tests/test_distributed_gpu.py)."""
        self = cls.__new__(cls)
        self.ctx, self.slab, self.dist, self.group = ctx, slab, None, None
        rank, world = slab.rank, slab.world
        self.peer_lo, self.peer_hi = (-1 if slab.lo_phys else rank - 1), (-1 if slab.hi_phys else rank + 1)
        self.transport, self.serial, self._allreduce_cb, self.handle = "ipc", False, None, None
        handle, mine = C.c_void_p(), C.create_string_buffer(_hip.IPC_HANDLE_BYTES)
        _hip.check(ctx.lib.beat_comm_create_ipc(ctx.handle, rank, world, self.peer_lo, self.peer_hi, int(plane_doubles), None, None,
                                                None, mine, C.byref(handle)))
        registry[rank] = handle
        barrier.wait(timeout=120)
        arr = (C.c_void_p * world)(*[registry[r] for r in range(world)])
        _hip.check(ctx.lib.beat_comm_ipc_connect_local(handle, arr, world))
        barrier.wait(timeout=120)
        self.handle = handle
        return self

    def info(self) -> dict:
        out = (C.c_int * 4)()
        _hip.check(self.ctx.lib.beat_comm_info(self.handle, out))
        return {"transport": _hip.TRANSPORT_NAMES.get(out[0], str(out[0])), "rccl_ranks": int(out[1]), "world": int(out[2]),
                "allreduce": {0: "caller", 1: "rccl", 2: "ipc"}[int(out[3])]}

    def profile(self, enable: bool) -> None:
        _hip.check(self.ctx.lib.beat_comm_profile(self.handle, int(bool(enable))))

    def profile_read(self) -> dict:
        out = (C.c_double * 6)()
        _hip.check(self.ctx.lib.beat_comm_profile_read(self.handle, out))
        return {"halo_ms": out[0], "halo_count": int(out[1]), "allreduce_ms": out[2], "allreduce_count": int(out[3]),
                "halo_stall_ms": out[4], "halo_stall_count": int(out[5])}

    # -- host-staged callbacks (rehearsal transport) ------------------------------------------------------------
    def _d2h(self, ptr, count):
        out = np.empty(int(count))
        _hip.check(self.ctx.lib.beat_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), 8 * int(count)))
        return out

    def _h2d(self, ptr, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        _hip.check(self.ctx.lib.beat_memcpy_h2d(self.ctx.handle, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), 8 * arr.size))

    def _global(self, group_rank: int) -> int:
        return group_rank if self.group is None else self.dist.get_global_rank(self.group, group_rank)

    def _halo(self, user, first, ghost_lo, last, ghost_hi, plane):
        try:
            import torch

            d, ops, recvs = self.dist, [], []
            for send_ptr, recv_ptr, peer in ((first, ghost_lo, self.peer_lo), (last, ghost_hi, self.peer_hi)):
                if peer < 0:
                    continue
                out = torch.from_numpy(self._d2h(send_ptr, plane))
                inn = torch.empty(int(plane), dtype=torch.float64)
                ops.append(d.P2POp(d.isend, out, self._global(peer), self.group))
                ops.append(d.P2POp(d.irecv, inn, self._global(peer), self.group))
                recvs.append((recv_ptr, inn))
            for req in (d.batch_isend_irecv(ops) if ops else []):
                req.wait()
            for recv_ptr, inn in recvs:
                self._h2d(recv_ptr, inn.numpy())
            return 0
        except Exception:  # an exception must not unwind through the C frames
            import traceback

            traceback.print_exc()
            return 1

    def _allreduce(self, user, values, count):
        try:
            import torch

            t = torch.from_numpy(self._d2h(values, count))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            self._h2d(values, t.numpy())
            return 0
        except Exception:
            import traceback

            traceback.print_exc()
            return 1

    # -- the two operations on their own ---------------------------------------------------------------------------
    def exchange_halo(self, field) -> None:
        _hip.check(self.ctx.lib.beat_comm_halo_exchange(self.handle, field.ptr, field.n, field.plane))

    def allreduce_sum(self, tensor) -> None:
        _hip.check(self.ctx.lib.beat_comm_allreduce_sum(self.handle, C.c_void_p(tensor.data_ptr()), int(tensor.numel())))

    def close(self, collective: bool | None = None) -> None:
        """Destroy the communicator.  For the ipc transport this is COLLECTIVE by default when the ranks are processes of a
        torch.distributed group: a rank frees its mailbox only after every rank has drained its streams -- a neighbour's
        last receive kernel still stores its `freed` count into this mailbox, and its all-reduce stores may target it too
        (beat_comm_destroy itself synchronises this rank's streams only).  ``collective=False``: local teardown (the
        finaliser; a rank leaving alone after a failure)."""
        if self.handle:
            if collective is None:
                collective = (self.transport == "ipc" and self.slab.world > 1 and self.dist is not None
                              and getattr(self.dist, "is_initialized", lambda: False)())
            if collective:
                self.ctx.synchronize()
                self.dist.barrier(group=self.group)
            self.ctx.lib.beat_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close(collective=False)
        except Exception:
            pass


class DiffusionSolver:
    """theta-rule diffusion step on one slab of a (possibly) decomposed grid."""

    def __init__(self, ops, slab: Slab, group=None, force_distributed: bool = False, stage_driven: bool | None = None,
                 libcomm: "LibComm | None" = None):
        """Decomposed grids (``slab.world > 1``, or ``force_distributed`` on one rank for tests): with the HIP
        backend the whole solve is one library call over a :class:`LibComm` (RCCL when ``torch.distributed`` runs on
        nccl, host-staged callbacks on any other backend -- the rehearsal transport).  ``stage_driven=True`` (env
        ``BEAT_STAGE_DRIVEN=1``) keeps the iteration in Python, stage by stage over ``torch.distributed``: that is
        the orchestration the CPU tests run with oracle-backed ``ops`` on gloo, and the only route for the
        polynomial preconditioner."""
        import os

        self.ops = ops
        self.slab = slab
        self.group = group
        self._last_its = 8
        self.libcomm = libcomm
        if isinstance(ops, HipOps) and slab.world > 1 and hasattr(slab, "nz_global"):
            # z type of the ghost planes: a face of the whole grid when the neighbour owns just that one plane
            lo = 0 if (not slab.lo_phys and slab.z0 - 1 == 0) else 1
            hi = 2 if (not slab.hi_phys and slab.z1 == slab.nz_global - 1) else 1
            _hip.check(ops.lib.beat_pde_set_ghost_types(ops.handle, lo, hi))
        if stage_driven is None:
            stage_driven = os.environ.get("BEAT_STAGE_DRIVEN", "0") == "1"
        force_distributed = force_distributed or os.environ.get("BEAT_FORCE_DISTRIBUTED", "0") == "1"
        if slab.world > 1 or force_distributed:  # force_distributed: run the collective path on 1 rank (tests)
            import torch.distributed as dist

            self.dist = dist
            on_device = getattr(getattr(ops, "st", None), "is_cuda", False)
            nccl = dist.is_initialized() and dist.get_backend(group) == "nccl"
            if dist.is_initialized() and not nccl and on_device:
                self.dist = _HostStagedDist(dist)  # device fields on a host-only backend: rehearsal transport
            if self.libcomm is None and on_device and not stage_driven and isinstance(ops, HipOps):
                # The library's own communicator.  Should creating it fail on any rank (librccl missing, a refused
                # ncclCommInitRank), ALL ranks fall back together to the stage-driven loop over torch.distributed -- slower
                # (Python per iteration, no initial guess) but the same numbers -- instead of leaving the job half-connected.
                # LibComm's set-up fails on all ranks or on none (LibCommUnavailable).
                transport = os.environ.get("BEAT_DIST_TRANSPORT") or ("rccl" if (nccl or slab.world == 1) else "callbacks")
                try:
                    self.libcomm = LibComm(ops.ctx, slab, dist, group, transport, plane_doubles=ops.plane)
                except LibCommUnavailable as exc:
                    import warnings

                    warnings.warn(f"in-library communicator unavailable ({exc}): using the stage-driven loop over torch.distributed",
                                  RuntimeWarning, stacklevel=2)
        else:
            self.dist = None

    # -- communication ------------------------------------------------------------------------
    def exchange_halo(self, field) -> None:
        """Send the first/last owned plane to the z-neighbours, receive into the ghost planes."""
        if self.libcomm is not None:
            if self.slab.world > 1:
                self.libcomm.exchange_halo(field)
            return
        self.finish_halo(self.start_halo(field))

    def finish_halo(self, reqs) -> None:
        for req in reqs:
            req.wait()

    def start_halo(self, field):
        """Enqueue the ghost-plane exchange (RCCL send/recv on the communication stream); returns the
        requests to wait on before anything reads the ghost planes."""
        if self.dist is None or self.slab.world == 1:
            return []
        dist, slab, plane = self.dist, self.slab, field.plane
        ops = []
        first = field.data[:plane]
        last = field.data[field.n - plane :]
        if not slab.lo_phys:
            ops.append(dist.P2POp(dist.isend, first, self._peer(slab.rank - 1), self.group))
            ops.append(dist.P2POp(dist.irecv, field.ghost_lo, self._peer(slab.rank - 1), self.group))
        if not slab.hi_phys:
            ops.append(dist.P2POp(dist.isend, last, self._peer(slab.rank + 1), self.group))
            ops.append(dist.P2POp(dist.irecv, field.ghost_hi, self._peer(slab.rank + 1), self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def _peer(self, group_rank: int) -> int:
        if self.group is None:
            return group_rank
        return self.dist.get_global_rank(self.group, group_rank)

    def _allreduce(self, t) -> None:
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def _precondition(self, slot: int) -> None:
        ops = self.ops
        for j in range(ops.pc_num_passes):
            self.exchange_halo(ops.pc_io(j)[0])
            ops.pc_pass(j, slot)

    # -- solve ----------------------------------------------------------------------------------
    def solve(self, v_prev, stim_w, stim_amp, x, rtol=1e-8, atol=1e-50, max_it=1000, defer_flush: bool = False) -> KspResult:
        """x <- solution of A x = B v_prev + dt*sum amp_k w_k, started from x0 = v_prev.  ``defer_flush``: see
        HipOps.solve_single (backends without that support simply apply the update)."""
        ops = self.ops
        can_defer = defer_flush and hasattr(ops, "flush_pending")
        if hasattr(ops, "flush_pending"):
            ops.flush_pending()
        if self.dist is None and self.libcomm is None:
            if can_defer:
                ops.st_ptr_for_flush = None
                return ops.solve_single(v_prev, stim_w, stim_amp, x, rtol, atol, max_it, defer_flush=True)
            return ops.solve_single(v_prev, stim_w, stim_amp, x, rtol, atol, max_it)
        if self.libcomm is not None and ops.pc_num_passes == 0:
            return ops.solve_dist(self.libcomm, v_prev, stim_w, stim_amp, x, rtol, atol, max_it, defer_flush=bool(can_defer))
        self.exchange_halo(v_prev)
        ops.rhs(v_prev, stim_w, stim_amp, x)
        self._allreduce(ops.st[0:3])
        ops.cg_begin(rtol, atol, max_it)
        npass = ops.pc_num_passes
        if npass:  # z = M^-1 r by the polynomial preconditioner, p = z
            self._precondition(_hip.ST_RZ)
            self._allreduce(ops.st[1:2])
            ops.cg_first_z()
        launched = 0
        chunk = max(1, self._last_its)
        ring = ops.ring
        K = len(ring)
        while True:
            chunk = min(chunk, max_it - launched)
            for it in range(chunk):
                i = launched + it
                p_cur = ops.p if npass else ring[i % K]
                reqs = self.start_halo(p_cur)   # ghost planes of p travel ...
                ops.spmv_interior(p_cur)        # ... while the interior planes are computed
                self.finish_halo(reqs)
                ops.spmv_boundary(p_cur)
                self._allreduce(ops.st[3:4])
                if npass:
                    ops.cg_update(x)
                    self._precondition(_hip.ST_RZN)  # replaces the Jacobi r.z written by cg_update
                    self._allreduce(ops.st[4:6])
                    ops.cg_next_z()
                else:  # Jacobi with deferred x (see beat_pde_solve)
                    ops.cg_update_r(i % K)
                    self._allreduce(ops.st[4:6])
                    if i % K == K - 1:
                        ops.x_flush(x, i + 1 - K, True)
                    ops.cg_next_oop(p_cur, ring[(i + 1) % K])
            launched += chunk
            st = ops.read_state()
            if st[_hip.ST_STOP] != 0.0 or launched >= max_it:
                break
            chunk = 2
        if not npass:
            nupd = int(st[_hip.ST_NUPD])
            if nupd % K:
                if can_defer:
                    ops.pending = (x, (nupd // K) * K, nupd % K)
                    ops.st_ptr_for_flush = ops.st.data_ptr()
                else:
                    ops.x_flush(x, (nupd // K) * K, False)
        its = int(st[_hip.ST_ITERS])
        self._last_its = max(its, 1)
        reason = int(st[_hip.ST_REASON]) if st[_hip.ST_STOP] != 0.0 else -3
        # a solve that ran out of iterations is REPORTED (converged_reason < 0, as PETSc's KSP), not raised: the
        # caller decides (BaseModel.step -> Status.NOT_CONVERGING / ksp_error_if_not_converged)
        res = KspResult(its, float(np.sqrt(st[_hip.ST_RR])), reason, float(np.sqrt(st[_hip.ST_BB])))
        if getattr(ops, "ksp_log", None) is not None:
            ops.ksp_log.append(res)
        return res
