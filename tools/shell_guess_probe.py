#!/usr/bin/env python3
"""Why does the extrapolated initial guess take the PCG from 8 to 3 iterations on the slab and only from 11 to 9 on the
voxel shell (BASELINE.json configs[4])?  Instruments the shell run step by step:

  per step     : the library's own solve (ksp_guess_order as given): iterations, ||r0||/||b|| is what the candidates below say
  probe steps  : for the SAME system (b, A, v_ of that step) the initial residual ||b - A x0|| / ||b|| and the Jacobi-PCG
                 iteration count (a PCG in torch over beat_pde_apply, the library's stopping test) from
                   x0 = v_                                   ("order 0")
                   x0 = v_ + polynomial extrapolation of the last m diffusion increments, m = 1..4 (what the library does)
                   x0 = v_ + the A-norm optimal combination of the last m increments, m = 2, 4, 6 (Galerkin projection of the
                        new system on their span -- the best ANY linear guess built from those increments can do; PETSc's
                        KSPGuessFischer)
  and the time scale: max over the tissue of |v_(n) - v_(n-1)| per step (how many steps an upstroke takes).

    python tools/shell_guess_probe.py --size 240 --steps 400 --every 20 [--dt 0.05] [--order auto]

Reference workload: /root/reference/demos/biv_endocardial.py:187-282 (dt = 0.05 ms, 2000 uA/cm^2 on the endocardium for 1 ms).
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))
sys.path.insert(0, str(ROOT / "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=240)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--every", type=int, default=20)
    ap.add_argument("--dt", type=float, default=0.05)
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--order", default="auto")
    ap.add_argument("--probe-from", type=int, default=8)
    args = ap.parse_args()
    import os

    os.environ["BEAT_GUESS_ORDER"] = args.order
    import torch

    import bench_biv

    P = bench_biv.build(args.size, rtol=args.rtol)
    pde, ode, solver, tissue = P["pde"], P["ode"], P["solver"], P["tissue"]
    ops = pde._ops
    ctx = ops.ctx
    dev = ctx.device
    row = ode._v_row
    N = ops.n
    dt = args.dt
    rtol = args.rtol
    fx, fy = ops.new_field(), ops.new_field()
    tis = torch.from_numpy(np.asarray(tissue)).to(dev)

    def apply(which, x):
        fx.data.copy_(x)
        ops.apply(which, fx, fy)
        return fy.data.clone()

    dinv = None

    def pcg(b, x0, bb, max_it=60):
        x = x0.clone()
        r = b - apply(0, x)
        tol2 = rtol * rtol * bb
        rr = float(r @ r)
        r0 = rr
        if rr <= tol2:
            return 0, r0
        z = dinv * r
        p = z.clone()
        rz = float(r @ z)
        k = 0
        while True:
            q = apply(0, p)
            alpha = rz / float(p @ q)
            x += alpha * p
            r -= alpha * q
            k += 1
            rr = float(r @ r)
            if rr <= tol2 or k >= max_it:
                return k, r0
            z = dinv * r
            rzn = float(r @ z)
            p = z + (rzn / rz) * p
            rz = rzn

    binom = {1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}
    hist = []  # newest first: d_{n-1}, d_{n-2}, ...
    v_last = None
    t = 0.0
    lib_its = []
    print("# step  t[ms]  lib_k  max|dv_|/step[mV]  | r0/b and k from: v_ | poly 1 2 3 4 | optimal 2 4 6", flush=True)
    rows = []
    for i in range(args.steps):
        t0, t1 = t, t + dt
        v_index = solver._fused_prepare()
        ode._dev.step(t0, dt, v_index=v_index, pending_ops=ode._pending_ops, v_row=row)
        pde.time.value = t0 + pde.parameters["theta"] * dt
        if not abs(dt - float(pde._timestep)) < 1.0e-12:
            pde._timestep.value = dt
            pde._update_matrices()
            dinv = None
        if dinv is None:
            C_m, theta, _ = ops._coeffs
            diag = C_m * ops._mass_dev[0] + theta * dt * ops._stiff_dev[0]
            dinv = torch.where(ops._mass_dev[0] > 0, 1.0 / torch.where(diag != 0, diag, torch.ones_like(diag)), torch.ones_like(diag))
        stim_w, stim_amp = [], []
        for s in pde._stimuli:
            a = s.amplitude()
            if a != 0.0 and s.field is not None:
                stim_w.append(s.field)
                stim_amp.append(a)
        v0 = row.data.clone()
        probe = i >= args.probe_from and (i % args.every == 0) and len(hist) >= 6
        if probe:
            b = apply(1, v0)
            for w, a in zip(stim_w, stim_amp):
                b += dt * a * w.data
            b = torch.where(tis, b, torch.zeros_like(b))  # nodes outside the tissue: identity rows, not part of ||b||
            bb = float(b @ b)
            res = []
            k0, r0 = pcg(b, v0, bb)
            res.append((np.sqrt(r0 / bb), k0))
            for m in (1, 2, 3, 4):
                e = sum(c * hist[j] for j, c in enumerate(binom[m]))
                k, r = pcg(b, v0 + e, bb)
                res.append((np.sqrt(r / bb), k))
            rv = b - apply(0, v0)
            for m in (2, 4, 6):
                D = hist[:m]
                AD = [apply(0, d) for d in D]
                G = torch.tensor([[float(D[a] @ AD[c]) for c in range(m)] for a in range(m)], dtype=torch.float64)
                g = torch.tensor([float(D[a] @ rv) for a in range(m)], dtype=torch.float64)
                c = torch.linalg.lstsq(G, g[:, None]).solution[:, 0]
                e = sum(float(c[j]) * D[j] for j in range(m))
                k, r = pcg(b, v0 + e, bb)
                res.append((np.sqrt(r / bb), k))
                del AD
        ksp = pde.solve_in_place(row, stim_w, stim_amp, defer_flush=False)
        ode._pending_ops = pde._ops
        x = row.data
        d = x - v0
        hist.insert(0, d)
        del hist[6:]
        dv = float((v0 - v_last).abs().max()) if v_last is not None else float("nan")
        v_last = v0
        lib_its.append(ksp.iterations)
        if probe:
            rows.append([i, t1, ksp.iterations, dv] + [x for pair in res for x in pair])
            print(f"{i:5d} {t1:7.2f} {ksp.iterations:3d} {dv:8.2f} | " + " ".join(f"{a:.1e}/{k:d}" for a, k in res), flush=True)
        t = t1
    lib_its = np.array(lib_its)
    print(f"library ({args.order}): mean k over all steps {lib_its.mean():.2f}; per 50 steps: "
          + " ".join(f"{lib_its[j:j + 50].mean():.1f}" for j in range(0, len(lib_its), 50)))
    if rows:
        R = np.array(rows)
        names = ["v_", "poly1", "poly2", "poly3", "poly4", "opt2", "opt4", "opt6"]
        print("mean over the probe steps (r0/b geometric, k arithmetic):")
        for j, nm in enumerate(names):
            print(f"  {nm:6s} r0/b {np.exp(np.log(R[:, 4 + 2 * j]).mean()):.2e}   k {R[:, 5 + 2 * j].mean():.2f}")


if __name__ == "__main__":
    main()
