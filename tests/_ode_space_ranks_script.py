"""Run through torch.distributed.run by tests/test_distributed_gpu.py: the split system of the reference's
tests/test_monodomain_solver.py (forward-Euler test ODE, source term in the PDE) on a 3-D box, with the ODE on a P2 or
DG1 space and the conductivity built from a NODAL fibre function -- on every rank of a z-slab decomposition.
argv: out_dir odespace [dim [percell]]   (dim 2: the reference's unit square, cut into slabs of rows; percell: a tensor per cell)"""
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

out_dir, odespace = Path(sys.argv[1]), sys.argv[2]
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 3
percell = len(sys.argv) > 4 and sys.argv[4] == "percell"  # 2-D: a tensor per cell (beat.grid.CellField) instead of a constant one
comm = g.COMM_WORLD
if dim == 3:
    mesh = g.create_box(comm, [np.zeros(3), np.array([1.0, 1.0, 0.75])], [12, 10, 9])
else:  # the reference's own test mesh (tests/test_monodomain_solver.py: unit square), cut into slabs of rows
    mesh = g.create_unit_square(comm, 24, 20)
time = g.Constant(mesh, 0.0)
x = g.SpatialCoordinate(mesh)
I_s = 8 * g.pi**2 * g.cos(2 * g.pi * x[0]) * g.cos(2 * g.pi * x[1]) * g.sin(time)
if dim == 3:
    # fibres as a vector P1 function (src/beat/conductivities.py:101-118): rotating with height
    W = g.functionspace(mesh, ("P", 1, (3,)))
    f0 = g.Function(W)
    f0.interpolate(lambda p: np.stack([np.cos(1.3 * p[2]), np.sin(1.3 * p[2]), 0.0 * p[0]]))
    M = beat.conductivities.define_conductivity_tensor(f0=f0, chi=1.0, g_il=1.0, g_it=0.4, g_el=1.0, g_et=0.4)
elif percell:
    # a conductivity tensor per box cell (what define_conductivity_tensor gives for per-cell fibres; per-cell M is how the
    # reference's ventricular demos build theirs, src/beat/conductivities.py:101-118): rotating with x, stiffer towards y = 1
    cc = g.cell_centers(mesh)
    ang = 0.9 * cc[:, 0] + 0.4 * cc[:, 1]
    f = np.stack([np.cos(ang), np.sin(ang)], axis=-1)
    M = g.CellField(mesh, (0.4 + 0.3 * cc[:, 1])[:, None, None] * np.eye(2)[None] + 0.6 * f[:, :, None] * f[:, None, :])
else:
    M = np.array([[1.0, 0.2], [0.2, 0.6]])
pde = beat.MonodomainModel(time=time, mesh=mesh, M=M, I_s=I_s, params={"petsc_options": {"ksp_rtol": 1e-12}})
V_ode = beat.utils.space_from_string(odespace, mesh, dim=1)
s = g.Function(V_ode)
s.interpolate(lambda p: -np.cos(2 * np.pi * p[0]) * np.cos(2 * np.pi * p[1]) * (1.0 + 0.3 * p[dim - 1]))
init_states = np.zeros((2, s.x.array.size))
init_states[1, :] = np.asarray(s.x.array)
ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(V_ode), v_pde=pde.state, fun=beat.models.simple.forward_euler,
                                     init_states=init_states, parameters=None, num_states=2, v_index=0)
assert ode.num_points == V_ode.num_dofs
solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode, theta=0.5 if odespace == "DG_1" else 1.0)
solver.solve((0.0, 0.3), dt=0.01)
v = np.asarray(pde.state.x.array).copy()
# the second state at the vertices of the slab (through the space's own to_p1 map), to compare across decompositions
vals = np.asarray(ode.values)
if V_ode.is_p1:
    s_vert = vals[1]
else:
    s_vert = vals[1][V_ode.layout()[1]]
np.savez(out_dir / f"rank{comm.rank}.npz", v=v, s=s_vert, z0=mesh.slab.z0, z1=mesh.slab.z1, dofs=V_ode.num_dofs,
         its=pde.ksp.getIterationNumber())
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
