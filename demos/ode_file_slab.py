#!/usr/bin/env python3
"""A cell model the package does not ship, straight from a gotran ``.ode`` file -- what the reference does with
``gotranx.load_ode`` + ``gotran2py`` (demos/niederer_benchmark.py:82-99), here with ``beat.models.from_ode``: the file is turned
into a device kernel (compiled at first use), pre-paced as a single cell on the device (``single_cell.get_steady_state``: the
whole pacing loop in one launch), then put on a slab whose corner is stimulated; the script reports when the far corner
activates.  The default file is the small excitable-cell model of the test suite (tests/data/small_cell.ode).

    python demos/ode_file_slab.py [--ode my_model.ode] [--dx 0.25] [--T 30] [--dt 0.05] [--beats 3]"""
import argparse
import tempfile
from pathlib import Path

import _path  # noqa: F401
import numpy as np

import beat
from beat import grid as g
from beat.models import from_ode


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ode", default=str(_path.ROOT / "tests" / "data" / "small_cell.ode"))
    ap.add_argument("--dx", type=float, default=0.25)
    ap.add_argument("--T", type=float, default=30.0)
    ap.add_argument("--dt", type=float, default=0.05)
    ap.add_argument("--beats", type=int, default=3)
    ap.add_argument("--scheme", default="generalized_rush_larsen")
    args = ap.parse_args(argv)

    model = from_ode(args.ode, scheme=args.scheme)
    v_name = model.v_name or model.state_names[0]
    print(f"{Path(args.ode).name}: {model.num_states} states, {model.num_parameters} parameters, potential `{v_name}`, scheme {model.scheme}")
    # single-cell pre-pacing with the model's own stimulus current, on the device (one launch for all beats)
    with tempfile.TemporaryDirectory() as tmp:
        paced = beat.single_cell.get_steady_state(fun=model, init_states=model.init_state_values(),
                                                  parameters=model.init_parameter_values(), outdir=Path(tmp), nbeats=args.beats,
                                                  BCL=400, dt=args.dt)
    print(f"after {args.beats} paced beats: {v_name} = {paced[model.state_index(v_name)]:.3f}")

    geo = beat.geometry.get_3D_slab_geometry(comm=g.COMM_WORLD, Lx=10.0, Ly=4.0, Lz=2.0, dx=args.dx)
    mesh = geo.mesh
    time = g.Constant(mesh, 0.0)
    cond = beat.conductivities.default_conductivities("Niederer")
    cells = g.locate_entities(mesh, 3, lambda x: (x[0] <= 1.5 + 1e-10) & (x[1] <= 1.5 + 1e-10) & (x[2] <= 1.5 + 1e-10))
    tags = g.meshtags(mesh, 3, cells, np.full(len(cells), 1, dtype=np.int32))
    I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=tags, marker=1, mesh_unit="mm",
                                           amplitude=50_000.0, duration=2.0)
    M = beat.conductivities.define_conductivity_tensor(f0=geo.f0, **cond)
    pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=I_s, C_m=0.01, dx=I_s.dZ)
    no_stim = {k: 0.0 for k in model.parameter_names if k in ("stim_amplitude", "i_Stim_Amplitude")}
    ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(g.functionspace(mesh, ("P", 1))), v_pde=pde.state, fun=model,
                                         init_states=paced, parameters=model.init_parameter_values(**no_stim),
                                         num_states=model.num_states, v_index=model.state_index(v_name))
    solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
    far = np.array([[10.0, 4.0, 2.0]])
    t, t_act, v_far = 0.0, None, []
    rest = float(paced[model.state_index(v_name)])
    while t < args.T - 1e-9:
        solver.step((t, t + args.dt))
        t += args.dt
        if round(t / args.dt) % 10 == 0:
            v = float(np.asarray(g.evaluate_function(pde.state, far)).ravel()[0])
            v_far.append(v)
            if t_act is None and v > rest + 0.5 * 60.0:
                t_act = t
    v = np.asarray(pde.state.x.array)
    print(f"{mesh.num_nodes} nodes, {round(args.T / args.dt)} steps on the device (fused split step, no host round trip): "
          f"far corner activated at {t_act if t_act is None else round(t_act, 2)} ms; V in [{v.min():.1f}, {v.max():.1f}]; "
          f"last solve {pde.ksp.iterations} PCG iterations, status {pde.status.name}")
    return t_act, np.array(v_far), solver


if __name__ == "__main__":
    main()
