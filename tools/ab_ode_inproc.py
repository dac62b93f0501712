#!/usr/bin/env python3
"""A/B of the ionic step kernel of several builds of the library INSIDE ONE PROCESS, on the same state array at the same device
addresses (and so the same physical pages): separate processes alternate between two levels of this kernel's time (HISTORY.md,
the "bimodal" TP06 kernel; rounds 3 - 5), which aliases with an A B A B schedule of processes.  Each build is loaded with its
own ``ctypes.CDLL`` handle (RTLD_LOCAL: the symbols of the builds do not meet) and gets its own ``beat_ctx`` on torch's current
stream; launches alternate build by build, HIP events around each launch.

    python tools/ab_ode_inproc.py [--n 512] [--model tp06|torord] [--reps 12] [--allocs 2] lib_a.so lib_b.so ...

``--allocs K``: the whole measurement is repeated on K successive allocations of the state array (the earlier ones kept
alive, so that the addresses differ): does the level depend on where the array lies?
Kernels: ``beat_ode_step`` (the plain instance), ``beat_ode_step_pending`` with nothing pending (the instance the split step
launches, without the 8 k + 40 B/node of a pending update) and ``beat_ode_step_classes`` (three classes in runs of whole tiles)."""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--model", default="tp06")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--allocs", type=int, default=2)
    ap.add_argument("--dt", type=float, default=0.01)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()

    import torch

    from beat import _hip
    from beat._device import Context, StateArray

    if args.model == "tp06":
        from beat.models import tp06 as model
        mid = _hip.MODEL_TP06_GRL1
        params = np.ascontiguousarray(model.init_parameter_values(stim_amplitude=0.0))
    else:
        from beat.models import torord as model
        mid = _hip.MODEL_TORORD_DYNCL_GRL1
        params = np.ascontiguousarray(model.init_parameter_values())
    ic = model.init_state_values()
    vi = model.state_index("V" if args.model == "tp06" else "v")
    S = len(ic)
    n = args.n ** 3
    plane = args.n ** 2
    ctx0 = Context.default()  # (torch's device and stream; the package's own library for StateArray)
    stream = torch.cuda.current_stream()
    builds = []
    import os
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="ab_ode_")
    envs = {}
    for spec in args.libs:
        # lib.so@NAME=VALUE[,NAME=VALUE]: a private copy of the library whose first launch (which reads the library's environment
        # switches into function-local statics, e.g. BEAT_ODE_GRID) runs with these variables set
        path, _, env = spec.partition("@")
        if env:
            copy = Path(tmp) / (Path(path).stem + "_" + env.replace("=", "").replace(",", "_") + ".so")
            shutil.copy(Path(path).resolve(), copy)
            envs[copy.name] = dict(kv.split("=", 1) for kv in env.split(","))
            path = str(copy)
        lib = C.CDLL(str(Path(path).resolve()), mode=C.RTLD_LOCAL)
        for name in ("beat_ctx_create", "beat_ode_step", "beat_ode_step_pending", "beat_last_error", "beat_ode_step_classes",
                     "beat_ode_class_table_doubles", "beat_ode_class_table_fill"):
            rt, at = _hip.SIGNATURES[name]
            getattr(lib, name).restype = rt
            getattr(lib, name).argtypes = at
        h = C.c_void_p()
        rc = lib.beat_ctx_create(ctx0.device_index, C.c_void_p(stream.cuda_stream), C.byref(h))
        assert rc == 0, lib.beat_last_error()
        builds.append((Path(path).name, lib, h))

    def fill(sa, seed):
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        for k in range(S):
            row = sa.rows[k]
            if k == vi:  # a travelling-front-like mix of both branches of the V < -40 conditionals
                row.copy_(torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 120.0 - 90.0)
            else:
                row.copy_(ic[k] * (1.0 + 0.01 * (2.0 * torch.rand(n, generator=g, device="cuda", dtype=torch.float64) - 1.0)))

    pp = params.ctypes.data_as(C.c_void_p)
    out = {"n": n, "model": args.model, "allocs": []}
    keep = []
    for a in range(args.allocs):
        sa = StateArray(ctx0, S, n, plane)
        keep.append(sa)
        fill(sa, 1234 + a)
        kinds = ("plain", "pending0", "classes")
        res = {name: {k: [] for k in kinds} for name, _, _ in builds}
        # three classes (with the same parameter values: the kernel's work does not depend on them) in runs of whole tiles: every
        # wavefront meets one class, as on the class-sorted compact layout of a voxelised wall
        markers = ((torch.arange(n, device="cuda") // (256 * 977)) % 3).to(torch.uint8)
        tables = {}
        for name, lib, h in builds:
            stride = C.c_int()
            assert lib.beat_ode_class_table_doubles(mid, C.byref(stride)) == 0
            P3 = np.ascontiguousarray(np.stack([params, params, params]))
            tab = torch.zeros(3 * stride.value, dtype=torch.float64, device="cuda")
            rc = lib.beat_ode_class_table_fill(h, mid, P3.ctypes.data_as(C.c_void_p), len(params), 3, C.c_void_p(tab.data_ptr()))
            assert rc == 0, lib.beat_last_error()
            tables[name] = tab

        def launch(lib, h, which, name=None):
            if which == "plain":
                rc = lib.beat_ode_step(h, mid, sa.ptr, n, sa.ld, pp, len(params), None, 0, 0.0, args.dt, vi, None)
            elif which == "pending0":
                rc = lib.beat_ode_step_pending(h, mid, sa.ptr, n, sa.ld, pp, len(params), None, 0, 0.0, args.dt, vi, None, None, None, 0, 0)
            else:
                rc = lib.beat_ode_step_classes(h, mid, sa.ptr, n, sa.ld, C.c_void_p(tables[name].data_ptr()), 3, C.c_void_p(markers.data_ptr()),
                                               0.0, args.dt, vi, None, None, None, None, None, 0, 0)
            assert rc == 0, lib.beat_last_error()

        for name, lib, h in builds:  # warm-up (module load, clocks; the library's environment switches are read here)
            for k, v in envs.get(name, {}).items():
                os.environ[k] = v
            for which in kinds:
                launch(lib, h, which, name)
            for k in envs.get(name, {}):
                os.environ.pop(k, None)
        torch.cuda.synchronize()
        for rep in range(args.reps):
            for which in kinds:
                for name, lib, h in builds:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    launch(lib, h, which, name)
                    e1.record(stream)
                    e1.synchronize()
                    res[name][which].append(e0.elapsed_time(e1))
            if rep % 4 == 3:  # keep the states physiological over many steps of dt: refill
                fill(sa, 99 + rep)
        finite = bool(torch.isfinite(sa.rows[vi]).all())
        line = {"ptr": hex(sa.rows.data_ptr()), "finite": finite}
        for name, _, _ in builds:
            for which in kinds:
                v = np.array(res[name][which][1:])
                line[f"{name}:{which}"] = {"median": float(np.median(v)), "min": float(v.min()), "max": float(v.max())}
                print(f"alloc {a} {name:28s} {which:9s} median {np.median(v):7.3f} ms  min {v.min():7.3f}  max {v.max():7.3f}", flush=True)
        out["allocs"].append(line)
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
