"""How many Jacobi-PCG iterations does the diffusion solve need from different initial guesses?

    x0 = v                       (what the library does: the ionic step's output)
    x0 = v + d1                  (d1 = previous step's diffusion increment x - v)
    x0 = v + 2 d1 - d2           (linear extrapolation of the increment)
    x0 = v + 3 d1 - 3 d2 + d3    (quadratic extrapolation)
    x0 = v + 4 d1 - 6 d2 + 4 d3 - d4   (cubic)
    x0 = v + 5 d1 - 10 d2 + 10 d3 - 5 d4 + d5   (quartic)

Same stopping test as the library (||r|| <= rtol ||b||).  Runs the bench workload (bump, then developed front) on
one GPU at a reduced size with a PCG written in torch over beat_pde_apply, so no kernel has to exist before the
numbers say whether it is worth writing.  Usage: python tools/guess_probe.py [n] [steps] [dt] [h]
(round 5: dt and h as arguments -- the slab at the shell's dt = 0.05 ms / h = 0.25 mm, tools/shell_guess_probe.py is the
shell's counterpart -- and, beside the polynomial extrapolations, the A-norm optimal combination of the last 2 / 4 / 6
increments: what the best linear guess from those increments would give)
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import bench  # noqa: E402
from beat import _hip, _stencil  # noqa: E402
from beat._device import Context, Field, StateArray  # noqa: E402
from beat._engine import HipOps, Slab  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    if len(sys.argv) > 3:
        bench.DT = float(sys.argv[3])
    if len(sys.argv) > 4:
        bench.H = float(sys.argv[4])
    print(f"slab {n}^3, TP06, dt = {bench.DT} ms, h = {bench.H} mm, D_l dt / h^2 = {bench.S_L / bench.C_M * bench.DT / bench.H ** 2:.3f}", flush=True)
    rtol = 1e-8
    ctx = Context(0)
    slab = Slab(n, 0, 1)
    plane = n * n
    N = plane * n
    ic, params, v_index = bench.tp06_defaults()
    mass_tab, stiff_tab = _stencil.stencil_tables(3, (bench.H,) * 3, bench.conductivity())
    ops = HipOps(ctx, (n, n, n), slab.lo_phys, slab.hi_phys, mass_tab, stiff_tab)
    ops.set_timestep(bench.C_M, bench.THETA, bench.DT)
    states = StateArray(ctx, len(ic), N, plane)
    lib = ctx.lib
    p_host = np.ascontiguousarray(params)
    p_ptr = p_host.ctypes.data_as(C.c_void_p)
    v = states.row_field(v_index)
    fx, fy = Field(ctx, N, plane), Field(ctx, N, plane)

    def A(x):
        fx.data.copy_(x)
        ops.apply(0, fx, fy)
        return fy.data.clone()

    def B(x):
        fx.data.copy_(x)
        ops.apply(1, fx, fy)
        return fy.data.clone()

    # diagonal of A: nodes of equal parity in x, y, z do not couple (stencil reach 1)
    diag = torch.zeros(N, dtype=torch.float64, device=ctx.device)
    idx = torch.arange(n, device=ctx.device)
    for c in range(8):
        m = (((idx % 2) == (c & 1))[None, None, :] & ((idx % 2) == ((c >> 1) & 1))[None, :, None]
             & ((idx % 2) == ((c >> 2) & 1))[:, None, None]).reshape(-1).to(torch.float64)
        diag += A(m) * m
    dinv = 1.0 / diag

    def pcg(b, x0, bb):
        x = x0.clone()
        r = b - A(x)
        tol2 = rtol * rtol * bb
        k = 0
        rr = float(r @ r)
        r0 = rr
        if rr <= tol2:
            return x, 0, r0
        z = dinv * r
        p = z.clone()
        rz = float(r @ z)
        while True:
            q = A(p)
            alpha = rz / float(p @ q)
            x += alpha * p
            r -= alpha * q
            k += 1
            rr = float(r @ r)
            if rr <= tol2 or k >= 200:
                return x, k, r0
            z = dinv * r
            rzn = float(r @ z)
            p = z + (rzn / rz) * p
            rz = rzn

    def optimal(b, v0, bb, D):
        """x0 = v0 + the A-norm optimal combination of the increments D (Galerkin projection on their span)"""
        rv = b - A(v0)
        AD = [A(d) for d in D]
        m = len(D)
        G = torch.tensor([[float(D[a] @ AD[c]) for c in range(m)] for a in range(m)], dtype=torch.float64)
        g = torch.tensor([float(D[a] @ rv) for a in range(m)], dtype=torch.float64)
        c = torch.linalg.lstsq(G, g[:, None]).solution[:, 0]
        return pcg(b, v0 + sum(float(c[j]) * D[j] for j in range(m)), bb)[1:]

    def run(label, nsteps):
        d1 = d2 = d3 = d4 = d5 = d6 = None
        v_last = None
        t = 0.0
        rows = []
        for i in range(nsteps):
            _hip.check(lib.beat_ode_step_pending(ctx.handle, _hip.MODEL_TP06_GRL1, states.ptr, N, states.ld, p_ptr, len(p_host),
                                                 None, 0, t, bench.DT, v_index, None, ops.handle, ops.ring[0].ptr, ops.fld, 0))
            v0 = v.data.clone()
            b = B(v0)
            bb = float(b @ b)
            x, k0, r0 = pcg(b, v0, bb)
            ks, rs = [k0], [r0]
            for g in ((v0 + d1) if d1 is not None else None, (v0 + 2 * d1 - d2) if d2 is not None else None,
                      (v0 + 3 * d1 - 3 * d2 + d3) if d3 is not None else None,
                      (v0 + 4 * d1 - 6 * d2 + 4 * d3 - d4) if d4 is not None else None,
                      (v0 + 5 * d1 - 10 * d2 + 10 * d3 - 5 * d4 + d5) if d5 is not None else None):
                if g is None:
                    ks.append(-1), rs.append(float("nan"))
                    continue
                xg, kg, rg = pcg(b, g, bb)
                ks.append(kg), rs.append(rg)
            for m, D in ((2, [d1, d2]), (4, [d1, d2, d3, d4]), (6, [d1, d2, d3, d4, d5, d6])):
                if D[-1] is None:
                    ks.append(-1), rs.append(float("nan"))
                else:
                    ko, ro = optimal(b, v0, bb, D)
                    ks.append(ko), rs.append(ro)
            d6, d5, d4, d3, d2, d1 = d5, d4, d3, d2, d1, x - v0
            dv = float((v0 - v_last).abs().max()) if v_last is not None else float("nan")
            v_last = v0
            v.data.copy_(x)
            rows.append(ks + [np.sqrt(r / bb) for r in rs])
            t += bench.DT
            if i % 5 == 4 or i < 8:
                print(f"{label} step {i:3d} max|dv_|/step {dv:6.2f} mV: k(v)={ks[0]} poly1..5={ks[1:6]} optimal2/4/6={ks[6:9]}  r0/b: "
                      + " ".join(f"{np.sqrt(r / bb):.2e}" for r in rs), flush=True)
        rows = np.array(rows[8:], dtype=float)
        names = ["v_", "poly1", "poly2", "poly3", "poly4", "poly5", "opt2", "opt4", "opt6"]
        print(f"{label}: mean over steps >= 8 (k arithmetic, r0/b geometric): "
              + ", ".join(f"{nm} {rows[:, j].mean():.2f} / {np.exp(np.log(rows[:, 9 + j]).mean()):.1e}" for j, nm in enumerate(names)), flush=True)

    bench.init_states(ctx, states, ic, v_index, n, slab, 1234, n)
    run("bump", steps)
    prof, pre = bench.developed_front_profile(ctx, n, ic, params, v_index, rtol)
    for k in range(states.S):
        states.rows[k].view(-1, n).copy_(prof[k][None, :].expand(N // n, n))
    run("front", steps)


if __name__ == "__main__":
    main()
