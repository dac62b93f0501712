"""Checkpoints of nodal functions (stand-in for the io4dolfinx calls of the reference's demos:
``write_function_on_input_mesh`` / ``write_function`` / ``read_function`` / ``read_timestamps`` /
``write_mesh``, demos/biv_endocardial.py:328-345, 399-417).

A checkpoint is a directory: ``meta.json`` (grid shape, spacing, list of time stamps per function name) and
one ``<name>_<k>_r<rank>.npy`` per stored time and rank holding that rank's z-slab of nodal values (x fastest).
Reading on a different number of ranks re-slices the stored slabs."""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np

from . import grid


def _meta_path(filename) -> Path:
    return Path(filename) / "meta.json"


def _load_meta(filename) -> dict:
    p = _meta_path(filename)
    if p.is_file():
        return json.loads(p.read_text())
    return {"functions": {}, "mesh": None}


def _mesh_meta(mesh: grid.Mesh) -> dict:
    return {"cells": list(mesh.n), "lower": list(mesh.lower), "upper": list(mesh.upper),
            "shape_global": list(mesh.shape_global)}


def write_mesh(filename, mesh: grid.Mesh, **kw) -> None:
    """Starts a NEW checkpoint (io4dolfinx.write_mesh opens its file in write mode): whatever an earlier run left
    under ``filename`` -- time stamps, function slabs -- is removed, so that ``read_timestamps`` / ``read_function``
    never mix two runs.  ``write_function`` then appends to it."""
    if mesh.comm.rank == 0:
        root = Path(filename)
        root.mkdir(parents=True, exist_ok=True)
        # only the files the previous checkpoint itself recorded (its meta.json): <name>_<k>_r<rank>.npy per function,
        # time stamp and slab -- anything else the caller keeps in this directory is not ours to delete
        for name, entry in _load_meta(filename).get("functions", {}).items():
            for k, slabs in enumerate(entry.get("slabs", [])):
                for r in range(len(slabs)):
                    (root / f"{name}_{k}_r{r}.npy").unlink(missing_ok=True)
        meta = {"mesh": _mesh_meta(mesh), "functions": {}}
        _meta_path(filename).write_text(json.dumps(meta))
    mesh.comm.Barrier()


def write_function(filename, u: grid.Function, time: float = 0.0, name: str | None = None, **kw) -> None:
    mesh = u.function_space.mesh
    name = name or u.name
    comm = mesh.comm
    Path(filename).mkdir(parents=True, exist_ok=True)
    meta = _load_meta(filename)
    entry = meta["functions"].setdefault(name, {"times": [], "slabs": []})
    k = len(entry["times"])
    np.save(Path(filename) / f"{name}_{k}_r{comm.rank}.npy", np.asarray(u.x.array))
    comm.Barrier()
    if comm.rank == 0:
        entry["times"].append(float(time))
        entry["slabs"].append([[int(a), int(b)] for a, b in _slab_ranges(mesh)])
        if meta.get("mesh") is None:
            meta["mesh"] = _mesh_meta(mesh)
        _meta_path(filename).write_text(json.dumps(meta))
    comm.Barrier()


write_function_on_input_mesh = write_function


def _slab_ranges(mesh: grid.Mesh):
    cuts = np.concatenate([[0], np.cumsum(mesh.slab.counts)])
    return [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:])]


def read_timestamps(comm=None, filename=None, function_name: str = "v", **kw) -> np.ndarray:
    meta = _load_meta(filename)
    if function_name not in meta["functions"]:
        raise KeyError(f"no function {function_name!r} in {filename}")
    return np.asarray(meta["functions"][function_name]["times"], dtype=np.float64)


def read_function(filename, u: grid.Function, time: float = 0.0, name: str | None = None, **kw) -> None:
    mesh = u.function_space.mesh
    name = name or u.name
    meta = _load_meta(filename)
    entry = meta["functions"].get(name)
    if entry is None:
        raise KeyError(f"no function {name!r} in {filename}")
    if meta["mesh"] is not None and list(meta["mesh"]["shape_global"]) != list(mesh.shape_global):
        raise ValueError(f"checkpoint grid {meta['mesh']['shape_global']} does not match the mesh {mesh.shape_global}")
    times = np.asarray(entry["times"])
    hits = np.nonzero(np.isclose(times, time, rtol=0, atol=1e-9))[0]
    if len(hits) == 0:
        raise KeyError(f"no time stamp {time} for {name!r} in {filename} (have {times.tolist()})")
    k = int(hits[-1])  # a time stamp written twice: the later write wins
    plane = mesh.plane
    z0, z1 = mesh.slab.z0, mesh.slab.z1
    out = np.empty((z1 - z0) * plane)
    for r, (a, b) in enumerate(entry["slabs"][k]):
        lo, hi = max(a, z0), min(b, z1)
        if lo >= hi:
            continue
        data = np.load(Path(filename) / f"{name}_{k}_r{r}.npy", mmap_mode="r")
        out[(lo - z0) * plane : (hi - z0) * plane] = data[(lo - a) * plane : (hi - a) * plane]
    u.x.array[:] = out
