#!/usr/bin/env python3
"""BASELINE.json configs[4] in synthetic form, end to end through the public API on ONE GPU: a truncated
ellipsoidal shell voxelised on a box, transmural fibre rotation, endo / mid / epi layers from the Laplace solve of
utils.expand_layer, ToR-ORd-dynCl with one parameter set per layer (DolfinMultiODESolver), endocardial surface
stimulus, Godunov split steps.

    python tools/bench_biv.py [--n 256] [--steps 10]      # --n 560 gives ~40 M tissue nodes (needs ~150 GB of HBM)
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))


def shell(n, h, chunk=32):
    """mask (n, n, n) and fibre directions (n^3, 3) of the shell, generated in z-chunks to keep the host
    footprint at the size of the outputs"""
    c = n * h / 2.0
    so = 0.48 * n * h * np.array([1.0, 0.9, 1.0])
    si = 0.66 * so
    mask = np.zeros((n, n, n), dtype=bool)
    f0 = np.zeros((n, n, n, 3))
    ax = (np.arange(n) + 0.5) * h
    for z0 in range(0, n, chunk):
        z1 = min(n, z0 + chunk)
        Z, Y, X = np.meshgrid(ax[z0:z1], ax, ax, indexing="ij")
        P = np.stack([X - c, Y - c, Z - c], axis=-1)
        ro = np.sqrt(((P / so) ** 2).sum(-1))
        ri = np.sqrt(((P / si) ** 2).sum(-1))
        mask[z0:z1] = (ro < 1.0) & (ri > 1.0) & (Z < 0.8 * n * h)
        depth = np.clip((ri - 1.0) / np.maximum(ri - ro, 1e-12), 0.0, 1.0)
        rad = P / np.maximum(np.linalg.norm(P, axis=-1, keepdims=True), 1e-12)
        circ = np.cross(np.array([0.0, 0.0, 1.0]), rad)
        circ /= np.maximum(np.linalg.norm(circ, axis=-1, keepdims=True), 1e-12)
        longi = np.cross(rad, circ)
        ang = np.deg2rad(60.0 - 120.0 * depth)[..., None]
        f0[z0:z1] = np.cos(ang) * circ + np.sin(ang) * longi
    return mask, f0.reshape(-1, 3), (so, si, c)


def build(n, h=0.25, rtol=1e-8, verbose=True):
    """The whole pipeline through the public API; returns its pieces (mesh, tissue mask of the nodes, facet tags,
    layer markers, pde, ode, solver, shell semi-axes)."""
    import beat
    from beat import grid as g
    from beat.models import torord

    say = (lambda *a: print(*a, flush=True)) if verbose else (lambda *a: None)
    tic = time.perf_counter()
    mask, f0, (so, si, c) = shell(n, h)
    mesh = g.create_voxel_mesh(g.COMM_WORLD, mask, h)
    tissue = mesh.node_active()
    say(f"box {n + 1}^3 = {(n + 1) ** 3 / 1e6:.1f} M nodes, {tissue.sum() / 1e6:.2f} M tissue nodes "
        f"({mask.mean() * 100:.0f} % of the voxels); geometry {time.perf_counter() - tic:.1f} s")
    tic = time.perf_counter()
    facets = mesh.exterior_facets()
    xyz = g._node_xyz(mesh, mesh.facet_vertices(facets).ravel()).reshape(len(facets), 4, 3)
    ctr = xyz.mean(axis=1) - c
    ro = np.sqrt(((ctr / so) ** 2).sum(axis=1))
    ri = np.sqrt(((ctr / si) ** 2).sum(axis=1))
    base = (np.ptp(xyz[:, :, 2], axis=1) < 1e-12) & (ri > 1.0) & (ro < 1.0)
    values = np.where(base, 0, np.where(np.abs(ri - 1.0) < np.abs(ro - 1.0) * (1.0 / 0.66), 10, 20)).astype(np.int32)
    ft = g.meshtags(mesh, 2, facets[values > 0], values[values > 0])
    del xyz, ctr, ro, ri
    V = g.functionspace(mesh, ("P", 1))
    layers = beat.utils.expand_layer(V, ft, 10, 20, endo_size=0.3, epi_size=0.3)
    marker_arr = np.where(tissue, np.asarray(layers.x.array), -1.0)
    markers = g.Function(V)
    markers.x.array[:] = marker_arr
    say(f"facet tags + expand_layer (Laplace PCG on the device): {time.perf_counter() - tic:.1f} s; "
        f"endo/mid/epi nodes = {[(marker_arr == k).sum() for k in (1, 0, 2)]}")
    tic = time.perf_counter()
    cond = beat.conductivities.default_conductivities("Bishop")
    M = beat.conductivities.define_conductivity_tensor(f0=g.CellField(mesh, f0), **cond)
    del f0
    time_c = g.Constant(mesh, 0.0)
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time_c, subdomain_data=ft, marker=10,
                                           mesh_unit="mm", amplitude=2000.0, start=0.0, duration=1.0)
    pde = beat.MonodomainModel(time=time_c, mesh=mesh, M=M, I_s=I_s, C_m=0.01,
                               params={"petsc_options": {"ksp_rtol": rtol}})
    keys = (0, 1, 2)  # celltype parameter of the model: 0 endo, 1 epi, 2 mid; layer markers: 1 endo, 0 mid, 2 epi
    celltype = {1: 0, 2: 1, 0: 2}
    ic = torord.init_state_values()
    ode = beat.odesolver.DolfinMultiODESolver(
        v_ode=g.Function(V), v_pde=pde.state, markers=markers, num_states={k: len(ic) for k in keys},
        fun={k: torord.generalized_rush_larsen for k in keys}, init_states={k: ic for k in keys},
        parameters={k: torord.init_parameter_values(i_Stim_Amplitude=0.0, celltype=celltype[k]) for k in keys},
        v_index={k: torord.state_index("v") for k in keys})
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    say(f"operators (device assembly), states: {time.perf_counter() - tic:.1f} s")
    return dict(mesh=mesh, tissue=tissue, ft=ft, marker_arr=marker_arr, pde=pde, ode=ode, solver=solver, h=h,
                voxels=int(mask.sum()), semi_axes=(so, si, c))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", "--n", dest="n", type=int, default=256, help="voxels per axis (--n clashes with torchrun's own options)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dt", type=float, default=0.05)
    ap.add_argument("--trace", type=int, default=0, help="print mean PCG iterations and ms/step per window of this many steps")
    ap.add_argument("--save", default="", help="directory for rank<r>.npz (potential, layer markers, slab) after the run")
    args = ap.parse_args()
    import os

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:  # one process per rank (torch.distributed.run); BEAT_DIST_BACKEND=gloo: ranks share one GPU (rehearsal)
        import torch.distributed as dist

        backend = os.environ.get("BEAT_DIST_BACKEND", "nccl")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local if backend == "nccl" else local % torch.cuda.device_count())
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))

    P = build(args.n)
    mesh, tissue, marker_arr, pde, solver = P["mesh"], P["tissue"], P["marker_arr"], P["pde"], P["solver"]
    t, dt = 0.0, args.dt
    its = []
    for _ in range(args.warmup):
        solver.step((t, t + dt))
        t += dt
    torch.cuda.synchronize()
    # (the KSP record of a step is taken from the operator's log: asking pde.ksp after every step would make the host wait for a
    # solve that step() leaves open until the next ionic launch is queued behind it)
    pde._ops.flush_pending()
    pde._ops.ksp_log = log = []
    tic = time.perf_counter()
    tw = tic
    for i in range(args.steps):
        solver.step((t, t + dt))
        t += dt
        if args.trace and (i + 1) % args.trace == 0 and mesh.comm.rank == 0:
            pde._ops.flush_pending()
            its[:] = [r.iterations for r in log]
            torch.cuda.synchronize()
            now = time.perf_counter()
            print(f"  steps {i + 1 - args.trace:5d}-{i + 1:5d} (t = {t:7.2f} ms): {np.mean(its[-args.trace:]):5.2f} its/step, "
                  f"{(now - tw) / args.trace * 1e3:6.2f} ms/step", flush=True)
            tw = now
    pde._ops.flush_pending()
    torch.cuda.synchronize()
    wall = time.perf_counter() - tic
    its[:] = [r.iterations for r in log]
    vmin, vmax = pde.state.field.minmax()
    nt = int(tissue.sum())
    if args.save:
        np.savez(os.path.join(args.save, f"rank{mesh.comm.rank}.npz"), v=np.asarray(pde.state.x.array), markers=marker_arr,
                 z0=mesh.slab.z0, z1=mesh.slab.z1, its=np.array(its))
    if world > 1:
        nt = int(mesh.comm.allreduce(float(nt)))
        vmin, vmax = -mesh.comm.allreduce_max(-vmin), mesh.comm.allreduce_max(vmax)
    if mesh.comm.rank == 0:
        print(f"{wall / args.steps * 1e3:.2f} ms/step, {nt * args.steps / wall / 1e9:.3f} G tissue-node-updates/s, "
          f"PCG {np.mean(its):.1f} its/step, v in [{vmin:.1f}, {vmax:.1f}] mV (0 = outside the tissue)"
          + (f"  [{world} ranks]" if world > 1 else ""), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
