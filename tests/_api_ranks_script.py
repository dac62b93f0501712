"""Run through torch.distributed.run by tests/test_distributed_gpu.py: a small Niederer-type problem through the PUBLIC
API on every rank (the mesh is cut into z-slabs by the communicator), results saved per rank.  argv: out_dir [dx]"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "fenicsx-beat_amd")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(os.environ.get("BEAT_DIST_BACKEND", "nccl"))

import beat  # noqa: E402
from beat import grid as g  # noqa: E402
from beat.models import tp06  # noqa: E402

out_dir = Path(sys.argv[1])
dx = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
comm = g.COMM_WORLD
geo = beat.geometry.get_3D_slab_geometry(comm=comm, Lx=6.0, Ly=3.0, Lz=4.0, dx=dx)
mesh = geo.mesh
cond = beat.conductivities.default_conductivities("Niederer")
time = g.Constant(mesh, 0.0)
cells = g.locate_entities(mesh, 3, lambda x: (x[0] <= 1.5 + 1e-10) & (x[1] <= 1.5 + 1e-10) & (x[2] <= 2.5 + 1e-10))
tags = g.meshtags(mesh, 3, cells, np.full(len(cells), 1, dtype=np.int32))
I_s = beat.stimulation.define_stimulus(mesh=mesh, chi=cond["chi"], time=time, subdomain_data=tags, marker=1, mesh_unit="mm",
                                       amplitude=50_000.0)
M = beat.conductivities.define_conductivity_tensor(f0=geo.f0, **cond)
petsc_options = {"ksp_rtol": 1e-10}
if os.environ.get("BEAT_TEST_SINGLE_REDUCTION") == "1":  # PETSc's -ksp_cg_single_reduction, honoured on decomposed grids
    petsc_options["ksp_cg_single_reduction"] = True
pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=I_s, C_m=0.01, dx=I_s.dZ, params={"petsc_options": petsc_options})
ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(g.functionspace(mesh, ("P", 1))), v_pde=pde.state,
                                     fun=tp06.generalized_rush_larsen, init_states=tp06.init_state_values(),
                                     parameters=tp06.init_parameter_values(stim_amplitude=0.0), num_states=19,
                                     v_index=tp06.state_index("V"))
solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode)
points = np.array([[0.0, 0.0, 0.0], [6.0, 3.0, 4.0], [3.0, 1.5, 2.0], [1.0, 1.0, 2.25], [5.5, 0.5, 3.9]])
dt, t, probes, opens = 0.05, 0.0, [], 0
for step in range(60):
    solver.step((t, t + dt))
    opens += int(getattr(pde._ops, "open_x", None) is not None)  # the step left its solve open (round 5: decomposed solves too)
    t += dt
    if step % 10 == 9:
        probes.append(g.evaluate_function(pde.state, points).ravel().copy())
v = np.asarray(pde.state.x.array).copy()
# ECG recovery (mass solve of K v + lead integrals: all-reduced dot products) and a checkpoint round trip
# (one file per rank; read back on this communicator and compared with what was written)
ecg = beat.ECGRecovery(v=pde.state, M=M, C_m=0.01, sigma_b=1.0, petsc_options={"ksp_rtol": 1e-12, "ksp_atol": 1e-30})
forms = [ecg.eval(p) for p in ((8.0, 1.0, 2.0), (-2.0, 4.0, 5.0))]
ecg.solve()
leads = np.array([comm.allreduce(beat.ecg.assemble_scalar(f)) for f in forms])  # callers reduce, as with dolfinx
chk = out_dir / "chk.bp"
beat.io.write_mesh(chk, mesh)
beat.io.write_function(chk, pde.state, time=t, name="v")
back = g.Function(pde.state.function_space)
beat.io.read_function(chk, back, time=t, name="v")
roundtrip_ok = bool(np.array_equal(np.asarray(back.x.array), v))
full = ode.full_values if hasattr(ode, "full_values") else None
np.savez(out_dir / f"rank{comm.rank}.npz", v=v, probes=np.array(probes), z0=mesh.slab.z0, z1=mesh.slab.z1,
         states=np.asarray(ode.values), its=pde.ksp.getIterationNumber(), nodes=mesh.num_nodes, leads=leads,
         roundtrip_ok=roundtrip_ok, opens=opens,
         merged_solves=(int(pde._ops.lib.beat_comm_merged_solves(pde._diffusion.libcomm.handle))
                        if getattr(pde._diffusion, "libcomm", None) is not None else 0))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
