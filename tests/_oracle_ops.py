"""Test-only compute backend for beat._engine.DiffusionSolver: the same stage interface as
``HipOps`` but every stage is the CPU oracle (NumPy) on CPU torch tensors, so that the product's
slab decomposition / halo exchange / all-reduce orchestration can run under gloo without a GPU."""

import numpy as np
import torch

from oracle import fem

ST_BB, ST_RZ, ST_RR, ST_PQ, ST_RZN, ST_RRN, ST_TOL2, ST_BETA, ST_STOP, ST_ITERS, ST_REASON, ST_RTOL, ST_ATOL, ST_MAXIT, ST_NUPD = range(15)


class CpuField:
    def __init__(self, n, plane):
        self.n, self.plane = n, plane
        self.buf = torch.zeros(n + 2 * plane, dtype=torch.float64)
        self.data = self.buf[plane : plane + n]

    @property
    def ghost_lo(self):
        return self.buf[: self.plane]

    @property
    def ghost_hi(self):
        return self.buf[self.plane + self.n :]

    def numpy(self):
        return self.data.numpy()


class OracleOps:
    def __init__(self, shape_local, lo_phys, hi_phys, mass_tab, stiff_tab, per_node=False):
        """mass_tab / stiff_tab: (27, 15) tables, or with ``per_node`` the (15, n) rows of this slab."""
        self.shape = tuple(int(v) for v in shape_local)
        nx, ny, nz = self.shape
        self.plane, self.n = nx * ny, nx * ny * nz
        self.lo_phys, self.hi_phys = bool(lo_phys), bool(hi_phys)
        self.per_node = bool(per_node)
        self.mass_tab, self.stiff_tab = np.asarray(mass_tab), np.asarray(stiff_tab)
        if self.per_node:
            assert self.mass_tab.shape == (15, self.n) and self.stiff_tab.shape == (15, self.n)
            self.tissue = self.mass_tab[0] != 0.0
        self.r, self.q, self.z = (CpuField(self.n, self.plane) for _ in range(3))
        self.ring = [CpuField(self.n, self.plane) for _ in range(6)]
        self.p = self.ring[0]
        self.alphas = np.zeros(6)
        self.pc_degree, self.pc_coef = 1, np.array([1.0])
        self.st = torch.zeros(16, dtype=torch.float64)
        # node types of the slab: z type decided by the PHYSICAL position of the plane
        def t(n, lo, hi):
            a = np.ones(n, dtype=np.int64)
            if n == 1 and lo and hi:
                return a
            if lo:
                a[0] = 0
            if hi:
                a[-1] = 2
            return a
        tx, ty, tz = t(nx, True, True), t(ny, True, True), t(nz, self.lo_phys, self.hi_phys)
        self.typ = (tx[None, None, :] + 3 * ty[None, :, None] + 9 * tz[:, None, None])

    def new_field(self):
        return CpuField(self.n, self.plane)

    def read_state(self):
        return self.st.numpy().copy()

    def set_timestep(self, C_m, theta, dt):
        self.C_m, self.theta, self.dt = C_m, theta, dt
        self.A = C_m * self.mass_tab + theta * dt * self.stiff_tab
        if self.per_node:  # nodes touched by no element: identity rows
            self.A[:, ~self.tissue] = 0.0
            self.A[0, ~self.tissue] = 1.0
            self.dinv = 1.0 / self.A[0]
            return
        self.dinv = (1.0 / self.A[:, 0])[self.typ].ravel()
        self._update_pc()

    # ---- polynomial preconditioner (same stage interface as HipOps) -------------------------------
    @property
    def pc_num_passes(self):
        return self.pc_degree - 1

    def set_preconditioner(self, degree):
        self.pc_degree = int(degree)
        if hasattr(self, "A"):
            self._update_pc()

    def _update_pc(self):
        from beat._engine import chebyshev_coefficients, spectrum_bounds

        if self.pc_degree > 1:
            self.pc_coef = chebyshev_coefficients(self.pc_degree, *spectrum_bounds(self.A))

    def pc_io(self, j):
        n = self.pc_num_passes
        out = lambda jj: self.z if (n - 1 - jj) % 2 == 0 else self.q  # noqa: E731
        return (self.r if j == 0 else out(j - 1)), out(j)

    def pc_pass(self, j, slot):
        if self.st[ST_STOP] != 0:
            return
        n = self.pc_num_passes
        fin, fout = self.pc_io(j)
        if j == 0:  # stage c_in * D^-1 r, ghost planes included
            tmp = CpuField(self.n, self.plane)
            tmp.buf.copy_(fin.buf)
            tmp.data.mul_(torch.from_numpy(self.pc_coef[n] * self.dinv))
            nx, ny, nz = self.shape
            dl = (1.0 / self.A[:, 0])[self._ghost_types(lo=True)].ravel()
            dh = (1.0 / self.A[:, 0])[self._ghost_types(lo=False)].ravel()
            tmp.ghost_lo.mul_(torch.from_numpy(self.pc_coef[n] * dl))
            tmp.ghost_hi.mul_(torch.from_numpy(self.pc_coef[n] * dh))
            fin = tmp
        s = self._apply(self.A, fin)
        r = self.r.data.numpy()
        out = self.dinv * (self.pc_coef[n - 1 - j] * r + s)
        fout.data.copy_(torch.from_numpy(out))
        if j == n - 1:
            self.st[slot] = float(r @ out)

    def _ghost_types(self, lo):
        """node types of the neighbouring slab's plane adjacent to this slab (interior in z there unless
        that neighbour plane is itself a physical face, which cannot happen for a ghost plane)."""
        nx, ny, _ = self.shape
        def t(n):
            a = np.ones(n, dtype=np.int64)
            if n > 1:
                a[0], a[-1] = 0, 2
            return a
        return t(nx)[None, :] + 3 * t(ny)[:, None] + 9

    def cg_first_z(self):
        if self.st[ST_STOP] != 0:
            return
        self.p.data.copy_(self.z.data)

    def cg_next_z(self):
        st = self.st
        if st[ST_STOP] != 0:
            return
        self._roll_scalars()
        if st[ST_STOP] != 0:
            return
        self.p.data.mul_(float(st[ST_BETA])).add_(self.z.data)

    def _apply(self, tab, f: CpuField, poison_ghosts=False):
        nx, ny, nz = self.shape
        X = np.zeros((nz + 2, ny + 2, nx + 2))
        X[1:-1, 1:-1, 1:-1] = f.data.numpy().reshape(nz, ny, nx)
        if not self.lo_phys:
            X[0, 1:-1, 1:-1] = np.nan if poison_ghosts else f.ghost_lo.numpy().reshape(ny, nx)
        if not self.hi_phys:
            X[-1, 1:-1, 1:-1] = np.nan if poison_ghosts else f.ghost_hi.numpy().reshape(ny, nx)
        y = np.zeros((nz, ny, nx))
        for k, (ox, oy, oz) in enumerate(fem.STENCIL_OFFSETS):
            coef = tab[k].reshape(nz, ny, nx) if self.per_node else tab[:, k][self.typ]
            xs = X[1 + oz : 1 + oz + nz, 1 + oy : 1 + oy + ny, 1 + ox : 1 + ox + nx]
            # like the kernels: a neighbour is only read where its coefficient is non-zero (poisoned ghosts)
            y += np.where(coef != 0.0, coef * np.where(coef != 0.0, xs, 0.0), 0.0)
        return y.ravel()

    def rhs(self, v_prev, stim_w, stim_amp, x):
        Mv, Kv = self._apply(self.mass_tab, v_prev), self._apply(self.stiff_tab, v_prev)
        stim = np.zeros(self.n)
        for w, a in zip(stim_w, stim_amp):
            stim += a * w.data.numpy()
        b = self.C_m * Mv - (1 - self.theta) * self.dt * Kv + self.dt * stim
        if self.per_node:
            b = np.where(self.tissue, b, 0.0)  # nodes outside the tissue are not part of the system
        r = self.dt * (stim - Kv)
        z = self.dinv * r
        if x is not v_prev:
            x.data.copy_(v_prev.data)
        self.r.data.copy_(torch.from_numpy(r))
        self.p.data.copy_(torch.from_numpy(z))
        self.st[ST_BB], self.st[ST_RZ], self.st[ST_RR] = float(b @ b), float(r @ z), float(r @ r)

    def cg_begin(self, rtol, atol, max_it):
        st = self.st
        tr, ta = rtol * rtol * float(st[ST_BB]), atol * atol
        st[ST_TOL2] = max(tr, ta)
        st[ST_ITERS], st[ST_RTOL], st[ST_ATOL], st[ST_MAXIT], st[ST_BETA], st[ST_NUPD] = 0.0, rtol, atol, float(max_it), 0.0, 0.0
        done = float(st[ST_RR]) <= float(st[ST_TOL2])
        st[ST_STOP] = 1.0 if done else 0.0
        st[ST_REASON] = (2.0 if float(st[ST_RR]) <= tr else 3.0) if done else 0.0

    def spmv_dot(self):
        if self.st[ST_STOP] != 0:
            return
        q = self._apply(self.A, self.p)
        self.q.data.copy_(torch.from_numpy(q))
        self.st[ST_PQ] = float(self.p.data.numpy() @ q)

    # ---- deferred-x stages ------------------------------------------------------------------------------
    def cg_update_r(self, slot):
        if self.st[ST_STOP] != 0:
            return
        alpha = float(self.st[ST_RZ]) / float(self.st[ST_PQ])
        self.alphas[slot] = alpha
        self.r.data.add_(self.q.data, alpha=-alpha)
        r = self.r.data.numpy()
        self.st[ST_RZN], self.st[ST_RRN] = float(r @ (self.dinv * r)), float(r @ r)
        self.st[ST_NUPD] += 1.0

    def cg_next_oop(self, p_cur, p_next):
        if self.st[ST_STOP] != 0:
            return
        self._roll_scalars()
        if self.st[ST_STOP] != 0:
            return
        p_next.data.copy_(torch.from_numpy(self.dinv) * self.r.data + float(self.st[ST_BETA]) * p_cur.data)

    def x_flush(self, x, ring_base, only_if_full):
        nvalid = int(min(6, max(0, int(self.st[ST_NUPD]) - ring_base)))
        if nvalid == 0 or (only_if_full and nvalid < 6):
            return
        for j in range(nvalid):
            x.data.add_(self.ring[j].data, alpha=float(self.alphas[j]))

    def spmv_interior(self, p=None):
        """Planes that need no ghost data, computed with the ghost planes POISONED to prove it."""
        p = p or self.p
        if self.st[ST_STOP] != 0:
            return
        nx, ny, nz = self.shape
        lo, hi = (0 if self.lo_phys else 1), nz - (0 if self.hi_phys else 1)
        # the real ghost planes may be mid-receive here: they are neither read nor written
        q = self._apply(self.A, p, poison_ghosts=True).reshape(nz, ny * nx)
        if hi > lo:
            assert np.isfinite(q[lo:hi]).all()
            self.q.data.view(nz, ny * nx)[lo:hi] = torch.from_numpy(q[lo:hi].copy())

    def spmv_boundary(self, p=None):
        p = p or self.p
        if self.st[ST_STOP] != 0:
            return
        nx, ny, nz = self.shape
        q = self._apply(self.A, p).reshape(nz, ny * nx)
        qv = self.q.data.view(nz, ny * nx)
        if not self.lo_phys:
            qv[0] = torch.from_numpy(q[0].copy())
        if not self.hi_phys:
            qv[nz - 1] = torch.from_numpy(q[nz - 1].copy())
        self.st[ST_PQ] = float(p.data.numpy() @ self.q.data.numpy())

    def cg_update(self, x):
        if self.st[ST_STOP] != 0:
            return
        alpha = float(self.st[ST_RZ]) / float(self.st[ST_PQ])
        x.data.add_(self.p.data, alpha=alpha)
        self.r.data.add_(self.q.data, alpha=-alpha)
        r = self.r.data.numpy()
        self.st[ST_RZN], self.st[ST_RRN] = float(r @ (self.dinv * r)), float(r @ r)

    def cg_next(self):
        st = self.st
        if st[ST_STOP] != 0:
            return
        self._roll_scalars()
        z = torch.from_numpy(self.dinv) * self.r.data
        self.p.data.mul_(float(st[ST_BETA])).add_(z)

    def _roll_scalars(self):
        st = self.st
        st[ST_BETA] = float(st[ST_RZN]) / float(st[ST_RZ])
        st[ST_RZ], st[ST_RR] = float(st[ST_RZN]), float(st[ST_RRN])
        st[ST_ITERS] += 1.0
        tr = float(st[ST_RTOL]) ** 2 * float(st[ST_BB])
        if float(st[ST_RR]) <= float(st[ST_TOL2]):
            st[ST_STOP], st[ST_REASON] = 1.0, (2.0 if float(st[ST_RR]) <= tr else 3.0)
        elif float(st[ST_ITERS]) >= float(st[ST_MAXIT]):
            st[ST_STOP], st[ST_REASON] = 1.0, -3.0

    def solve_single(self, v_prev, stim_w, stim_amp, x, rtol, atol, max_it):
        from beat._engine import KspResult

        self.rhs(v_prev, stim_w, stim_amp, x)
        self.cg_begin(rtol, atol, max_it)
        npass = self.pc_num_passes
        if npass:
            for j in range(npass):
                self.pc_pass(j, ST_RZ)
            self.cg_first_z()
        while self.st[ST_STOP] == 0:
            self.spmv_dot()
            self.cg_update(x)
            if npass:
                for j in range(npass):
                    self.pc_pass(j, ST_RZN)
                self.cg_next_z()
            else:
                self.cg_next()
        st = self.st.numpy()
        return KspResult(int(st[ST_ITERS]), float(np.sqrt(st[ST_RR])), int(st[ST_REASON]), float(np.sqrt(st[ST_BB])))
