"""Structured-grid stand-ins for the few dolfinx / ufl / scifem names the hot-path callers touch
(demos/fitzhughnagumo.py, demos/niederer_benchmark.py, README.md, tests/test_monodomain*.py,
tests/test_stimulation.py, tests/test_odesolver.py).

The reference discretises on DOLFINx meshes of boxes (src/beat/geometry.py:78-139).  Here a box
mesh *is* a structured grid with the same simplicial subdivision (see beat/_stencil.py), nodes
numbered x-fastest, optionally cut into z-slabs (one per rank).  Nothing here computes on the hot
path: functions hold their values in HBM (``Field``) and expose them through a lazy ``x.array``.
"""

from __future__ import annotations

import math
import operator
from dataclasses import dataclass

import numpy as np

from . import _stencil

default_scalar_type = np.float64


# ------------------------------------------------------------------------------------------------
# communicator stand-in
# ------------------------------------------------------------------------------------------------
class Comm:
    """Minimal ``MPI.Intracomm`` look-alike over torch.distributed (rank/size/allreduce/Barrier)."""

    def __init__(self, group=None):
        self.group = group

    @staticmethod
    def _dist():
        """torch.distributed if a process group exists.  Under a launcher that exports RANK / WORLD_SIZE > 1
        (``python -m torch.distributed.run --nproc-per-node N script.py`` -- what ``mpirun -n N`` is to the reference)
        the default group is created on first use: one process per GPU, RCCL; ``BEAT_DIST_BACKEND=gloo`` for a
        rehearsal with ranks sharing a device."""
        try:
            import torch.distributed as dist
        except Exception:  # pragma: no cover
            return None
        if not dist.is_available():
            return None
        if not dist.is_initialized():
            import os

            if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or "RANK" not in os.environ:
                return None
            import torch

            backend = os.environ.get("BEAT_DIST_BACKEND", "nccl")
            local = int(os.environ.get("LOCAL_RANK", "0"))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if torch.cuda.is_available():
                torch.cuda.set_device(local if backend == "nccl" else local % torch.cuda.device_count())
                kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
            else:
                backend, kw = "gloo", {}
            dist.init_process_group(backend, **kw)
        return dist

    @property
    def rank(self) -> int:
        d = self._dist()
        return d.get_rank(self.group) if d else 0

    @property
    def size(self) -> int:
        d = self._dist()
        return d.get_world_size(self.group) if d else 1

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def Barrier(self):
        d = self._dist()
        if d:
            d.barrier(self.group)

    def allreduce_array(self, values):
        """Element-wise sum of a small float array over the ranks (returns a NumPy array)."""
        values = np.asarray(values, dtype=np.float64)
        d = self._dist()
        if not d:
            return values
        import torch

        t = torch.from_numpy(values.copy())
        if d.get_backend(self.group) == "nccl":
            t = t.cuda()
        d.all_reduce(t, group=self.group)
        return t.cpu().numpy()

    def allreduce(self, value, op=None):
        """mpi4py's ``comm.allreduce(value, op=MPI.SUM | MPI.MAX | MPI.MIN | MPI.PROD)`` for one scalar."""
        key = "sum" if op is None else str(op).lower()
        if key not in ("sum", "max", "min", "prod"):
            raise ValueError(f"unsupported reduction {op!r}: use MPI.SUM, MPI.MAX, MPI.MIN or MPI.PROD")
        d = self._dist()
        if not d:
            return value
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        if d.get_backend(self.group) == "nccl":
            t = t.cuda()
        rop = {"sum": d.ReduceOp.SUM, "max": d.ReduceOp.MAX, "min": d.ReduceOp.MIN, "prod": d.ReduceOp.PRODUCT}[key]
        d.all_reduce(t, op=rop, group=self.group)
        return float(t.item())


    def allreduce_max(self, value) -> float:
        d = self._dist()
        if not d:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        if d.get_backend(self.group) == "nccl":
            t = t.cuda()
        d.all_reduce(t, op=d.ReduceOp.MAX, group=self.group)
        return float(t.item())


COMM_WORLD = Comm()


class _SelfComm(Comm):
    """``MPI.COMM_SELF``: this process alone, whatever launcher started it -- a mesh on it is never decomposed (the undivided
    reference run of bench.py's multi-rank parity block lives on it, on rank 0, beside the decomposed run on COMM_WORLD)."""

    @staticmethod
    def _dist():
        return None


COMM_SELF = _SelfComm()


# ------------------------------------------------------------------------------------------------
# tiny expression language (the subset of UFL used to describe stimuli and exact solutions)
# ------------------------------------------------------------------------------------------------
class Expr:
    """Scalar expression of the spatial coordinate and of mutable ``Constant`` objects (time)."""

    def evaluate(self, x=None):
        raise NotImplementedError

    def depends_on_x(self) -> bool:
        return False

    def depends_on_constants(self) -> bool:
        return False

    # arithmetic ---------------------------------------------------------------------------------
    def __add__(self, o):
        return BinOp(operator.add, self, as_expr(o))

    def __radd__(self, o):
        return BinOp(operator.add, as_expr(o), self)

    def __sub__(self, o):
        return BinOp(operator.sub, self, as_expr(o))

    def __rsub__(self, o):
        return BinOp(operator.sub, as_expr(o), self)

    def __mul__(self, o):
        return BinOp(operator.mul, self, as_expr(o))

    def __rmul__(self, o):
        return BinOp(operator.mul, as_expr(o), self)

    def __truediv__(self, o):
        return BinOp(operator.truediv, self, as_expr(o))

    def __rtruediv__(self, o):
        return BinOp(operator.truediv, as_expr(o), self)

    def __pow__(self, o):
        return BinOp(operator.pow, self, as_expr(o))

    def __rpow__(self, o):
        return BinOp(operator.pow, as_expr(o), self)

    def __neg__(self):
        return BinOp(operator.mul, Literal(-1.0), self)


class Literal(Expr):
    def __init__(self, value):
        self.value = float(value)

    def evaluate(self, x=None):
        return self.value


def as_expr(v) -> Expr:
    if isinstance(v, Expr):
        return v
    return Literal(v)


class BinOp(Expr):
    def __init__(self, op, a: Expr, b: Expr):
        self.op, self.a, self.b = op, a, b

    def evaluate(self, x=None):
        return self.op(self.a.evaluate(x), self.b.evaluate(x))

    def depends_on_x(self):
        return self.a.depends_on_x() or self.b.depends_on_x()

    def depends_on_constants(self):
        return self.a.depends_on_constants() or self.b.depends_on_constants()


class Func(Expr):
    def __init__(self, fn, a: Expr):
        self.fn, self.a = fn, a

    def evaluate(self, x=None):
        return self.fn(self.a.evaluate(x))

    def depends_on_x(self):
        return self.a.depends_on_x()

    def depends_on_constants(self):
        return self.a.depends_on_constants()


class Coordinate(Expr):
    def __init__(self, axis: int):
        self.axis = axis

    def evaluate(self, x=None):
        if x is None:
            raise ValueError("expression depends on the spatial coordinate")
        return x[self.axis]

    def depends_on_x(self):
        return True


class SpatialCoordinate:
    """``ufl.SpatialCoordinate(mesh)``: indexable, ``x[0]``, ``x[1]``, ..."""

    def __init__(self, mesh=None):
        self.mesh = mesh

    def __getitem__(self, i: int) -> Coordinate:
        return Coordinate(int(i))


class Condition(Expr):
    def __init__(self, op, a, b):
        self.op, self.a, self.b = op, as_expr(a), as_expr(b)

    def evaluate(self, x=None):
        return self.op(self.a.evaluate(x), self.b.evaluate(x))

    def depends_on_x(self):
        return self.a.depends_on_x() or self.b.depends_on_x()

    def depends_on_constants(self):
        return self.a.depends_on_constants() or self.b.depends_on_constants()


class Conditional(Expr):
    def __init__(self, cond, a, b):
        self.cond, self.a, self.b = cond, as_expr(a), as_expr(b)

    def evaluate(self, x=None):
        return np.where(self.cond.evaluate(x), self.a.evaluate(x), self.b.evaluate(x))

    def depends_on_x(self):
        return self.cond.depends_on_x() or self.a.depends_on_x() or self.b.depends_on_x()

    def depends_on_constants(self):
        return self.cond.depends_on_constants() or self.a.depends_on_constants() or self.b.depends_on_constants()


pi = math.pi


def cos(a):
    return Func(np.cos, as_expr(a))


def sin(a):
    return Func(np.sin, as_expr(a))


def exp(a):
    return Func(np.exp, as_expr(a))


def sqrt(a):
    return Func(np.sqrt, as_expr(a))


def ge(a, b):
    return Condition(operator.ge, a, b)


def le(a, b):
    return Condition(operator.le, a, b)


def gt(a, b):
    return Condition(operator.gt, a, b)


def lt(a, b):
    return Condition(operator.lt, a, b)


def And(a, b):
    return Condition(np.logical_and, a, b)


def Or(a, b):
    return Condition(np.logical_or, a, b)


def conditional(cond, a, b):
    return Conditional(cond, a, b)


def variable(e):
    return e


def zero():
    return Literal(0.0)


def product_factors(e: Expr) -> list[Expr]:
    if isinstance(e, BinOp) and e.op is operator.mul:
        return product_factors(e.a) + product_factors(e.b)
    return [e]


def separate(e: Expr):
    """Split ``e`` into (spatial Expr | None, temporal Expr | None) with e = spatial * temporal, or
    return None when some factor mixes the coordinate and a mutable Constant."""
    spatial, temporal = [], []
    for f in product_factors(as_expr(e)):
        dx, dc = f.depends_on_x(), f.depends_on_constants()
        if dx and dc:
            return None
        (spatial if dx else temporal).append(f)

    def prod(fs):
        out = None
        for f in fs:
            out = f if out is None else BinOp(operator.mul, out, f)
        return out

    return prod(spatial), prod(temporal)


# ------------------------------------------------------------------------------------------------
# mesh
# ------------------------------------------------------------------------------------------------
class CellType:
    interval = "interval"
    triangle = "triangle"
    tetrahedron = "tetrahedron"


class _Topology:
    def __init__(self, dim):
        self.dim = dim


class _Geometry:
    def __init__(self, mesh):
        self._mesh = mesh

    @property
    def x(self) -> np.ndarray:
        """(num_local_nodes, 3) coordinates, x fastest."""
        return self._mesh.node_coordinates(pad3=True)

    @property
    def dim(self):
        return self._mesh.dim


class Mesh:
    """Uniform box mesh with the reference's simplicial subdivision, optionally z-slab decomposed and
    optionally restricted to a subset of ACTIVE cells (a voxelised geometry: the domain is the union of the
    active simplices; nodes that touch no active cell carry identity rows in every operator)."""

    def __init__(self, cells, lower, upper, comm: Comm | None = None, active=None):
        self.n = tuple(int(c) for c in cells)
        self.dim = len(self.n)
        if self.dim not in (1, 2, 3) or min(self.n) < 1:
            raise ValueError(f"invalid cell counts {cells}")
        self.lower = tuple(float(v) for v in lower)
        self.upper = tuple(float(v) for v in upper)
        self.h = tuple((u - l) / c for l, u, c in zip(self.lower, self.upper, self.n))
        self.comm = comm or COMM_WORLD
        self.topology = _Topology(self.dim)
        self.geometry = _Geometry(self)
        nodes = [c + 1 for c in self.n] + [1] * (3 - self.dim)
        self.shape_global = tuple(nodes)  # (nx, ny, nz)
        from ._engine import Slab

        world = self.comm.size
        if world > 1 and self.dim < 2:
            raise NotImplementedError("slab decomposition is implemented for 2-D and 3-D boxes")
        # The mesh is cut along its slowest axis: z-slabs in 3-D, slabs of rows in 2-D (the reference's own tests run their
        # unit squares under ``mpirun -n 2``: .github/workflows/main-mpi.yml:33).  The kernels decompose along THEIR
        # slowest axis, so a decomposed 2-D mesh is handed to them as the grid (nx, 1, ny_local) -- same node numbering,
        # a "plane" is one row of nodes -- with its operator tables re-expressed accordingly (_stencil.tables_y_as_z).
        self.split_axis = self.dim - 1 if world > 1 else 2
        self.kernel_y_as_z = world > 1 and self.dim == 2
        weights = None
        if active is not None and world > 1 and np.asarray(active).size == int(np.prod(self.n)):
            # voxel mask: balance the slabs by tissue per node plane (a plane touches the voxel layers on both sides)
            per_layer = np.asarray(active, dtype=bool).reshape(tuple(reversed(self.n))).sum(axis=(1, 2)).astype(np.float64)
            weights = np.zeros(nodes[2])
            weights[:-1] += per_layer
            weights[1:] += per_layer
        self.slab = Slab(nodes[self.split_axis], self.comm.rank, world, weights)
        if self.kernel_y_as_z:
            self.shape_local = (nodes[0], 1, self.slab.nz)
            self.plane = nodes[0]
        else:
            self.shape_local = (nodes[0], nodes[1], self.slab.nz)
            self.plane = nodes[0] * nodes[1]
        self.num_nodes = self.plane * self.slab.nz
        self.num_nodes_global = nodes[0] * nodes[1] * nodes[2]
        self.simplices_per_cell = {1: 1, 2: 2, 3: 6}[self.dim]
        self.num_box_cells = int(np.prod(self.n))
        self.active = None  # bool per simplex (global numbering) or None = every cell
        self.active_box = None  # the same per box cell when the mask was given per box cell (voxels)
        if active is not None:
            active = np.asarray(active, dtype=bool).ravel()
            if active.size == self.num_box_cells:
                self.active_box = active
                active = np.repeat(active, self.simplices_per_cell)
            if active.size != self.num_box_cells * self.simplices_per_cell:
                raise ValueError(f"active mask has {active.size} entries for {self.num_box_cells} box cells")
            self.active = active
        self.num_cells_global = (int(self.active.sum()) if self.active is not None
                                 else self.num_box_cells * self.simplices_per_cell)

    def basix_cell(self):
        return {1: CellType.interval, 2: CellType.triangle, 3: CellType.tetrahedron}[self.dim]

    # ---- geometry ------------------------------------------------------------------------------
    def axis_coordinates(self, axis: int, local: bool = True) -> np.ndarray:
        if axis >= self.dim:
            return np.zeros(1)
        c = self.lower[axis] + self.h[axis] * np.arange(self.n[axis] + 1)
        if axis == self.split_axis and local:
            c = c[self.slab.z0 : self.slab.z1]
        return c

    def node_coordinates(self, pad3=False, local=True) -> np.ndarray:
        ax = [self.axis_coordinates(a, local) for a in range(3)]
        Z, Y, X = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        cols = [X.ravel(), Y.ravel(), Z.ravel()]
        return np.stack(cols if pad3 else cols[: self.dim], axis=1)

    # ---- cells ---------------------------------------------------------------------------------
    def _cell_vertex_mask(self, node_ok: np.ndarray) -> np.ndarray:
        """node_ok: bool (nz, ny, nx) on the GLOBAL grid -> bool (num_cells_global,) 'all vertices ok',
        cell id = box_cell_id * simplices_per_cell + k, box cells numbered x fastest."""
        d = self.dim
        cx = self.n[0]
        cy = self.n[1] if d >= 2 else 1
        cz = self.n[2] if d == 3 else 1
        out = np.zeros((cz, cy, cx, self.simplices_per_cell), dtype=bool)
        simp = _stencil._SIMPLICES[d]
        for k, s in enumerate(simp):
            ok = np.ones((cz, cy, cx), dtype=bool)
            for corner in s:
                ox, oy, oz = corner & 1, (corner >> 1) & 1, (corner >> 2) & 1
                ok &= node_ok[oz : oz + cz, oy : oy + cy, ox : ox + cx]
            out[..., k] = ok
        return out.reshape(-1)

    def cell_vertices(self, cell_ids: np.ndarray) -> np.ndarray:
        """(len, dim+1) GLOBAL node ids of the given cells."""
        cell_ids = np.asarray(cell_ids, dtype=np.int64)
        d = self.dim
        spc = self.simplices_per_cell
        box, k = np.divmod(cell_ids, spc)
        cx = self.n[0]
        cy = self.n[1] if d >= 2 else 1
        ix = box % cx
        iy = (box // cx) % cy
        iz = box // (cx * cy)
        nx, ny = self.shape_global[0], self.shape_global[1]
        simp = np.array(_stencil._SIMPLICES[d], dtype=np.int64)[k]  # (len, d+1) corner ids
        ox, oy, oz = simp & 1, (simp >> 1) & 1, (simp >> 2) & 1
        return (ix[:, None] + ox) + nx * ((iy[:, None] + oy) + ny * (iz[:, None] + oz))

    def all_cells(self) -> np.ndarray:
        if self.active is not None:
            return np.nonzero(self.active)[0].astype(np.int64)
        return np.arange(self.num_cells_global, dtype=np.int64)

    # ---- facets (faces of box cells; only the exterior ones are ever needed) ------------------------
    # facet id = box_cell * 2*dim + 2*axis + side: the face of that box cell normal to `axis` on its low
    # (side 0) or high (side 1) end.  On the simplicial mesh such a face is 1 edge (2-D) or 2 triangles
    # (3-D) split along the diagonal from its lowest to its highest corner (geometry.py:121-139).
    def _box_active(self) -> np.ndarray:
        d = self.dim
        shape = tuple(reversed(self.n))
        if self.active is None:
            return np.ones(shape, dtype=bool)
        if self.active_box is not None:
            return self.active_box.reshape(shape)
        per = self.active.reshape(shape + (self.simplices_per_cell,))
        if not (per.all(axis=-1) | ~per.any(axis=-1)).all():
            raise NotImplementedError("facets of a mesh whose mask cuts through box cells")
        return per.all(axis=-1)

    def exterior_facets(self) -> np.ndarray:
        """ids of the faces of active box cells whose other side is outside the (active) domain."""
        d = self.dim
        act = self._box_active()
        pad = np.pad(act, 1, constant_values=False)
        ids = []
        box = np.arange(self.num_box_cells, dtype=np.int64).reshape(act.shape)
        for axis in range(d):
            npax = d - 1 - axis  # numpy axis of the array (z, y, x order)
            for side in (0, 1):
                sl = [slice(1, -1)] * d
                sl[npax] = slice(0, -2) if side == 0 else slice(2, None)
                ext = act & ~pad[tuple(sl)]
                ids.append(box[ext] * (2 * d) + 2 * axis + side)
        return np.sort(np.concatenate(ids))

    def facet_vertices(self, facet_ids: np.ndarray) -> np.ndarray:
        """(len, 2^(dim-1)) GLOBAL node ids of the facets' corners, lowest corner id first."""
        facet_ids = np.asarray(facet_ids, dtype=np.int64)
        d = self.dim
        box, f = np.divmod(facet_ids, 2 * d)
        axis, side = np.divmod(f, 2)
        cx = self.n[0]
        cy = self.n[1] if d >= 2 else 1
        ix, iy, iz = box % cx, (box // cx) % cy, box // (cx * cy)
        nx, ny = self.shape_global[0], self.shape_global[1]
        out = np.zeros((len(facet_ids), 2 ** (d - 1)), dtype=np.int64)
        for a in range(d):
            sel = axis == a
            if not sel.any():
                continue
            corners = [c for c in range(2**d)]
            for s in (0, 1):
                m = sel & (side == s)
                if not m.any():
                    continue
                cs = [c for c in corners if ((c >> a) & 1) == s]
                for j, c in enumerate(cs):
                    ox, oy, oz = c & 1, (c >> 1) & 1, (c >> 2) & 1
                    out[m, j] = (ix[m] + ox) + nx * ((iy[m] + oy) + ny * (iz[m] + oz))
        return out

    def facet_area(self, facet_ids: np.ndarray) -> np.ndarray:
        d = self.dim
        axis = (np.asarray(facet_ids, dtype=np.int64) % (2 * d)) // 2
        h = np.array(self.h)
        full = float(np.prod(h))
        return full / h[axis]

    def node_active(self, local: bool = True) -> np.ndarray:
        """bool per node: touched by at least one active cell (all True without a mask)."""
        nx, ny, nz = self.shape_global
        if self.active is None:
            ok = np.ones(nx * ny * nz, dtype=bool)
        elif self.active_box is not None:
            # voxel mask: a node is touched iff one of the (up to 2^dim) voxels around it is active
            d = self.dim
            act = self.active_box.reshape(tuple(reversed(self.n)))
            grid_ok = np.zeros(tuple(reversed(self.shape_global[:d])), dtype=bool)
            for corner in range(2**d):
                sl = tuple(slice((corner >> (d - 1 - a)) & 1, ((corner >> (d - 1 - a)) & 1) + act.shape[a]) for a in range(d))
                grid_ok[sl] |= act
            ok = grid_ok.ravel()
        else:
            ok = np.zeros(nx * ny * nz, dtype=bool)
            ids = np.nonzero(self.active)[0]
            for s in range(0, len(ids), 1 << 20):
                ok[self.cell_vertices(ids[s : s + (1 << 20)]).ravel()] = True
        if local:
            ok = ok[self.slab.z0 * self.plane : self.slab.z1 * self.plane]
        return ok


def _mesh(comm, lower, upper, n):
    return Mesh(n, lower, upper, comm if isinstance(comm, Comm) else COMM_WORLD)


def create_voxel_mesh(comm, mask, h, origin=None):
    """Voxelised geometry: ``mask`` is a bool array of shape (cz, cy, cx) (or (cy, cx) / (cx,)), True for
    voxels inside the tissue; every active voxel is split into simplices exactly like a box-mesh cell
    (src/beat/geometry.py:121-139), so a fully active mask reproduces ``create_box``."""
    mask = np.asarray(mask, dtype=bool)
    d = mask.ndim
    n = tuple(reversed(mask.shape))
    h = (float(h),) * d if np.ndim(h) == 0 else tuple(float(v) for v in h)
    origin = (0.0,) * d if origin is None else tuple(float(v) for v in origin)
    upper = tuple(o + c * hh for o, c, hh in zip(origin, n, h))
    return Mesh(n, origin, upper, comm if isinstance(comm, Comm) else COMM_WORLD, active=mask.ravel())


def cell_centers(mesh: "Mesh") -> np.ndarray:
    """(num_box_cells, dim) centres of the box cells, x fastest."""
    ax = [mesh.lower[a] + mesh.h[a] * (np.arange(mesh.n[a]) + 0.5) for a in range(mesh.dim)]
    grids = np.meshgrid(*reversed(ax), indexing="ij")
    return np.stack([g.ravel() for g in reversed(grids)], axis=1)


class CellField:
    """Piecewise-constant data per box cell (voxel) or per simplex: fibre directions (ncells, dim) or
    tensors (ncells, dim, dim); the stand-in for the DG0 functions the reference's geometries carry."""

    def __init__(self, mesh: "Mesh", values):
        values = np.asarray(values, dtype=np.float64)
        if values.shape[0] not in (mesh.num_box_cells, mesh.num_box_cells * mesh.simplices_per_cell):
            raise ValueError(f"cell data has leading size {values.shape[0]}; expected one entry per box cell "
                             f"({mesh.num_box_cells}) or per simplex")
        self.mesh = mesh
        self.values = values



def create_unit_interval(comm, nx, **kw):
    return _mesh(comm, (0.0,), (1.0,), (nx,))


def create_interval(comm, nx, points, **kw):
    return _mesh(comm, (points[0],), (points[1],), (nx,))


def create_unit_square(comm, nx, ny, cell_type=CellType.triangle, **kw):
    _require_simplex(cell_type, CellType.triangle)
    return _mesh(comm, (0.0, 0.0), (1.0, 1.0), (nx, ny))


def create_rectangle(comm, points, n, cell_type=CellType.triangle, **kw):
    _require_simplex(cell_type, CellType.triangle)
    return _mesh(comm, tuple(points[0]), tuple(points[1]), tuple(n))


def create_unit_cube(comm, nx, ny, nz, cell_type=CellType.tetrahedron, **kw):
    _require_simplex(cell_type, CellType.tetrahedron)
    return _mesh(comm, (0.0,) * 3, (1.0,) * 3, (nx, ny, nz))


def create_box(comm, points, n, cell_type=CellType.tetrahedron, **kw):
    _require_simplex(cell_type, CellType.tetrahedron)
    return _mesh(comm, tuple(points[0]), tuple(points[1]), tuple(n))


def _require_simplex(cell_type, expected):
    name = getattr(cell_type, "name", cell_type)
    if name != expected:
        raise NotImplementedError(f"only {expected} cells are implemented (got {cell_type})")


def locate_entities(mesh: Mesh, dim: int, marker) -> np.ndarray:
    """Cells (``dim == mesh.topology.dim``) whose vertices ALL satisfy ``marker(x)``, x of shape
    (3, num_points) -- dolfinx.mesh.locate_entities semantics."""
    if dim == mesh.topology.dim - 1:
        return locate_entities_boundary(mesh, dim, marker)
    if dim != mesh.topology.dim:
        raise NotImplementedError("only cells and exterior facets are implemented")
    x = mesh.node_coordinates(pad3=True, local=False).T
    ok = np.asarray(marker(x), dtype=bool)
    if ok.shape == ():
        ok = np.full(x.shape[1], bool(ok))
    nx, ny, nz = mesh.shape_global
    sel = mesh._cell_vertex_mask(ok.reshape(nz, ny, nx))
    if mesh.active is not None:
        sel &= mesh.active
    return np.nonzero(sel)[0].astype(np.int32)


def locate_entities_boundary(mesh: Mesh, dim: int, marker) -> np.ndarray:
    """Exterior facets (``dim == mesh.topology.dim - 1``) whose vertices ALL satisfy ``marker(x)`` --
    dolfinx.mesh.locate_entities_boundary semantics; ids as in Mesh.exterior_facets."""
    if dim != mesh.topology.dim - 1:
        raise NotImplementedError("only exterior facets (dim == topological dimension - 1) are implemented")
    x = mesh.node_coordinates(pad3=True, local=False).T
    ok = np.asarray(marker(x), dtype=bool)
    if ok.shape == ():
        ok = np.full(x.shape[1], bool(ok))
    facets = mesh.exterior_facets()
    return facets[ok[mesh.facet_vertices(facets)].all(axis=1)]


@dataclass
class MeshTags:
    mesh: Mesh
    dim: int
    indices: np.ndarray
    values: np.ndarray

    def find(self, value) -> np.ndarray:
        return self.indices[self.values == value]


def meshtags(mesh: Mesh, dim: int, entities, values) -> MeshTags:
    entities = np.asarray(entities, dtype=np.int64)
    values = np.broadcast_to(np.asarray(values), entities.shape).copy()
    order = np.argsort(entities, kind="stable")
    return MeshTags(mesh, int(dim), entities[order], values[order])


class _IntegralType(str):
    """'dx' / 'ds'; calling it gives ufl's long name ("cell" / "exterior_facet"), as ``Measure.integral_type()``."""

    def __call__(self):
        return {"dx": "cell", "ds": "exterior_facet"}[str(self)]


class Measure:
    """``ufl.Measure("dx", domain=mesh, subdomain_data=tags)``; calling it with a marker restricts
    the integration domain to the cells carrying that tag."""

    def __init__(self, integral_type="dx", domain=None, subdomain_data=None, subdomain_id=None, metadata=None):
        if integral_type not in ("dx", "ds"):
            raise NotImplementedError("only cell ('dx') and exterior-facet ('ds') measures are implemented")
        self.integral_type = _IntegralType(integral_type)
        self.domain = domain
        self.subdomain_data = subdomain_data
        self.subdomain_id = subdomain_id
        self.metadata = metadata

    def __call__(self, subdomain_id=None, domain=None, metadata=None, **kw):
        return Measure(self.integral_type, domain or self.domain, self.subdomain_data, subdomain_id,
                       metadata or self.metadata)

    def facets(self):
        """Exterior facet ids to integrate over (``ds``)."""
        if self.integral_type != "ds":
            raise ValueError("not a facet measure")
        if self.subdomain_id is None:
            return self.domain.exterior_facets()
        if self.subdomain_data is None:
            raise ValueError("measure has a subdomain id but no subdomain_data")
        return self.subdomain_data.find(self.subdomain_id)

    def cells(self):
        """Cell ids to integrate over, or None for the whole mesh."""
        if self.integral_type != "dx":
            raise ValueError("not a cell measure")
        if self.subdomain_id is None:
            return None
        if self.subdomain_data is None:
            raise ValueError("measure has a subdomain id but no subdomain_data")
        return self.subdomain_data.find(self.subdomain_id)


def dx(domain=None, **kw):
    return Measure("dx", domain=domain, **kw)


def ds(domain=None, **kw):
    return Measure("ds", domain=domain, **kw)


# ------------------------------------------------------------------------------------------------
# constants, spaces, functions
# ------------------------------------------------------------------------------------------------
class Constant(Expr):
    """``dolfinx.fem.Constant(mesh, value)``: mutable scalar (or small vector) parameter."""

    def __init__(self, mesh, value):
        self.mesh = mesh
        self._value = np.array(value, dtype=np.float64)

    @property
    def value(self):
        return self._value if self._value.ndim else float(self._value)

    @value.setter
    def value(self, v):
        self._value = np.array(v, dtype=np.float64)

    def __float__(self):
        return float(self._value)

    def __len__(self):
        return len(self._value)

    def evaluate(self, x=None):
        return float(self._value)

    def depends_on_constants(self):
        return True


class _Element:
    def __init__(self, family, degree):
        self.family_name = family
        self._degree = degree

    def degree(self):
        return self._degree


class FunctionSpace:
    """P1 (the PDE space, values on the device with ghost planes), DG0 (host-side cell data), and -- as ODE spaces
    only -- P2 and DG1, whose functions are plain device vectors exchanged with P1 by interpolation
    (``utils.local_project``)."""

    def __init__(self, mesh: Mesh, family="Lagrange", degree=1):
        fam = {"P": "Lagrange", "CG": "Lagrange", "Lagrange": "Lagrange"}.get(family)
        if family in ("DG", "Discontinuous Lagrange", "dP") and degree in (0, 1):
            fam = "DG"  # degree 0: host-side cell data (CellFunction); degree 1: ODE space
        elif fam is None or degree not in (1, 2):
            raise NotImplementedError(f"function space {family} {degree} is not implemented on the HIP backend")
        self.mesh = mesh
        self.family = fam
        self.degree = degree
        self._element = _Element(fam, degree)
        self._layout = None

    def ufl_element(self):
        return self._element

    @property
    def is_p1(self) -> bool:
        return self.family == "Lagrange" and self.degree == 1

    @property
    def dofmap(self):
        """``V.dofmap.index_map.size_local`` etc. (demos/pace_train.py:133-137): local / global dof counts; there are
        no ghost dofs in ``x.array`` (ghost planes are a device-side detail)."""
        from types import SimpleNamespace

        n = self.num_dofs
        n_global = self.mesh.num_nodes_global if self.is_p1 else n
        return SimpleNamespace(index_map=SimpleNamespace(size_local=n, num_ghosts=0, size_global=n_global),
                               index_map_bs=1)

    # -- degrees of freedom of the non-P1 spaces ------------------------------------------------------------------
    def _edges(self):
        """(first vertex, second vertex) of every edge of the simplicial mesh whose FIRST vertex this rank owns: node i
        and i + o for the forward stencil offsets o that stay inside the box (every such pair shares a box cell, hence an
        edge); GLOBAL node ids -- on a decomposed mesh the second vertex of an edge that points upwards out of the slab
        is a node of the upper neighbour's first plane (this rank's upper ghost plane)."""
        mesh = self.mesh
        nx, ny, nz = mesh.shape_global
        ranges = [np.arange(nx), np.arange(ny), np.arange(nz)]
        if mesh.comm.size > 1:
            ranges[mesh.split_axis] = np.arange(mesh.slab.z0, mesh.slab.z1)
        iz, iy, ix = np.meshgrid(ranges[2], ranges[1], ranges[0], indexing="ij")
        node = (ix + nx * (iy + ny * iz)).ravel()
        ix, iy, iz = ix.ravel(), iy.ravel(), iz.ravel()
        a, b = [], []
        for ox, oy, oz in _stencil.OFFSETS[1::2]:
            if (oy and mesh.dim < 2) or (oz and mesh.dim < 3):
                continue
            ok = (ix + ox < nx) & (iy + oy < ny) & (iz + oz < nz)
            a.append(node[ok])
            b.append(node[ok] + ox + nx * (oy + ny * oz))
        return np.concatenate(a), np.concatenate(b)

    def layout(self):
        """(from_p1, to_p1): from_p1 = (idx (n, 2), w (n, 2)) expressing every dof as a combination of P1 vertex
        values (the P1 interpolant evaluated at the dof's point); to_p1 = for every vertex one dof located there."""
        if self._layout is None:
            # Node ids are LOCAL to the rank's slab: 0 .. n_local-1 its own nodes (plane by plane), n_local .. n_local +
            # plane - 1 the upper ghost plane, -plane .. -1 the lower one -- the positions of those values in a field.  On a
            # decomposed mesh a P2 space holds the vertices of the slab and the edges that start at them; a DG1 space the
            # cells that touch one of the slab's node planes (the layer between two slabs is held by both ranks, as
            # DOLFINx holds ghost cells: the ODE is pointwise, both compute the same values there).
            mesh = self.mesh
            plane = mesh.plane
            multi = mesh.comm.size > 1
            base = mesh.slab.z0 * plane if multi else 0
            nv = mesh.num_nodes if multi else mesh.num_nodes_global
            if self.family == "Lagrange" and self.degree == 2:
                ea, eb = self._edges()
                ea, eb = ea - base, eb - base
                idx = np.concatenate([np.stack([np.arange(nv), np.arange(nv)], axis=1), np.stack([ea, eb], axis=1)])
                w = np.concatenate([np.tile([1.0, 0.0], (nv, 1)), np.full((len(ea), 2), 0.5)])
                to_p1 = np.arange(nv, dtype=np.int64)
            elif self.family == "DG" and self.degree == 1:
                cells = mesh.all_cells()
                if multi:
                    # cell id = box cell (x fastest) * simplices + k: cells per layer along the axis the mesh is cut along
                    per_layer = int(np.prod(mesh.n[: mesh.split_axis])) * mesh.simplices_per_cell
                    layer = cells // per_layer
                    cells = cells[(layer >= mesh.slab.z0 - 1) & (layer <= mesh.slab.z1 - 1)]
                verts = mesh.cell_vertices(cells).ravel() - base
                idx = np.stack([verts, verts], axis=1)
                w = np.tile([1.0, 0.0], (len(verts), 1))
                own = (verts >= 0) & (verts < nv)
                to_p1 = np.full(nv, -1, dtype=np.int64)
                pos = np.nonzero(own)[0]
                to_p1[verts[pos][::-1]] = pos[::-1]  # first dof sitting at each vertex
                if (to_p1 < 0).any():
                    if mesh.active is None:
                        raise RuntimeError("a vertex of the slab belongs to no cell of the DG1 space")
                    to_p1[to_p1 < 0] = 0  # vertices outside the tissue: any dof (their potential is not part of the system)
            else:
                raise NotImplementedError("layout is only needed for P2 / DG1 spaces")
            self._layout = ((np.ascontiguousarray(idx, dtype=np.int64), np.ascontiguousarray(w)), to_p1)
        return self._layout

    @property
    def num_dofs(self):
        if self.family == "DG" and self.degree == 0:
            return self.mesh.num_box_cells * self.mesh.simplices_per_cell
        if self.is_p1:
            return self.mesh.num_nodes
        return len(self.layout()[0][0])

    def tabulate_dof_coordinates(self):
        if self.family == "DG" and self.degree == 0:
            return cell_midpoints(self.mesh, np.arange(self.num_dofs, dtype=np.int64))
        if self.is_p1:
            return self.mesh.node_coordinates(pad3=True)
        (idx, w), _ = self.layout()
        xyz = self.mesh.node_coordinates(pad3=True, local=False)
        base = self.mesh.slab.z0 * self.mesh.plane if self.mesh.comm.size > 1 else 0
        return w[:, :1] * xyz[idx[:, 0] + base] + w[:, 1:] * xyz[idx[:, 1] + base]


class VectorFunctionSpace:
    """``functionspace(mesh, ("P", 1, (dim,)))``: vector P1 -- the space of the fibre / sheet fields the reference's
    geometries carry (``geo.f0``).  Its functions are host-side nodal data (:class:`VectorFunction`); they enter the
    hot path only through the conductivity tensor, which is reduced to one tensor per simplex when the operators are
    assembled."""

    def __init__(self, mesh: Mesh, value_size: int):
        # (on a decomposed mesh every rank holds the whole nodal field on the host -- it is input data, reduced to one
        # tensor per simplex of the rank's slab when the operators are assembled; x.array is therefore the GLOBAL vector)
        self.mesh = mesh
        self.value_size = int(value_size)
        self.family, self.degree = "Lagrange", 1
        self._element = _Element("Lagrange", 1)

    def ufl_element(self):
        return self._element

    is_p1 = False

    @property
    def num_dofs(self) -> int:
        return self.mesh.num_nodes_global * self.value_size

    def tabulate_dof_coordinates(self) -> np.ndarray:
        return self.mesh.node_coordinates(pad3=True, local=False)


class VectorFunction:
    """Vector P1 function: nodal values (num_nodes, value_size) on the host; ``x.array`` is the flat, node-major
    (xyzxyz...) view dolfinx uses for blocked spaces."""

    def __init__(self, V: VectorFunctionSpace, name: str = "f"):
        self.function_space = V
        self.name = name
        self.values = np.zeros((V.mesh.num_nodes_global, V.value_size))
        self._x = _CellVector(self.values.reshape(-1))

    @property
    def x(self):
        return self._x

    def interpolate(self, f) -> None:
        """``f(x)`` with x of shape (3, num_nodes) returning (value_size, num_nodes), as dolfinx."""
        x = self.function_space.tabulate_dof_coordinates().T
        self.values[:] = np.asarray(f(x), dtype=np.float64).reshape(self.function_space.value_size, -1).T

    def ufl_element(self):
        return self.function_space.ufl_element()


def functionspace(mesh: Mesh, element, **kw):
    if isinstance(element, _Element):
        return FunctionSpace(mesh, element.family_name, element.degree())
    family, degree = element[0], element[1]
    if len(element) > 2 and element[2] not in (None, ()):  # ("P", 1, (dim,)): vector P1
        shape = tuple(element[2])
        if len(shape) != 1 or family not in ("P", "CG", "Lagrange") or degree != 1:
            raise NotImplementedError(f"function space {element} is not implemented on the HIP backend")
        return VectorFunctionSpace(mesh, shape[0])
    return FunctionSpace(mesh, family, degree)


def cell_midpoints(mesh: Mesh, cell_ids: np.ndarray) -> np.ndarray:
    """(len, 3) centroids of the given simplices."""
    verts = mesh.cell_vertices(cell_ids)
    return _node_xyz(mesh, verts.ravel()).reshape(verts.shape + (3,)).mean(axis=1)


class _TrackedArray(np.ndarray):
    """ndarray whose item assignments bump the owner's version (so device copies can be refreshed lazily)."""

    _owner = None

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        if self._owner is not None:
            self._owner._version += 1


class _CellVector:
    def __init__(self, array):
        self.array = array

    def scatter_forward(self):
        pass


class CellFunction:
    """``dolfinx.fem.Function`` on a DG0 space: one value per simplex, kept on the host.  Used as a stimulus
    current ``I_s`` that the caller re-interpolates every step (demos/ukb_atlas.py:340-356, 440-445)."""

    def __init__(self, V: "FunctionSpace", name: str = "f"):
        self.function_space = V
        self.name = name
        arr = np.zeros(V.num_dofs).view(_TrackedArray)
        arr._owner = self
        self._version = 0
        self.x = _CellVector(arr)

    def interpolate(self, expr) -> None:
        mesh = self.function_space.mesh
        e = expr.expr if isinstance(expr, Expression) else expr
        if hasattr(e, "evaluate_cells"):
            vals = e.evaluate_cells(mesh)  # fast path for sums of local windows
        elif isinstance(e, Expr):
            ids = mesh.all_cells()
            vals = np.zeros(self.x.array.shape)
            vals[ids] = np.broadcast_to(e.evaluate(cell_midpoints(mesh, ids).T), ids.shape)
        else:
            ids = mesh.all_cells()
            vals = np.zeros(self.x.array.shape)
            vals[ids] = e(cell_midpoints(mesh, ids).T)
        self.x.array[:] = vals


class Expression:
    """``dolfinx.fem.Expression(expr, points)``: here just the expression (points are implied by the space)."""

    def __init__(self, expr, points=None, **kw):
        self.expr = expr


class LazyArray:
    """``Function.x.array``: values live in HBM; the host copy is fetched on first use and cached
    until the device data changes.  Supports the idioms the reference's callers use
    (``arr[:] = ...``, ``arr[i]``, ``arr.max()``, ``np.asarray(arr)``, ``arr.size``)."""

    __array_priority__ = 100

    def __init__(self, function: "Function"):
        self._f = function

    # numpy protocol ------------------------------------------------------------------------------
    def __array__(self, dtype=None, copy=None):
        a = self._f._host()
        return a.astype(dtype) if dtype is not None and dtype != a.dtype else a

    # size queries must not touch the values: `field` would first apply a deferred update of an aliased row (one
    # extra 1.4 ms pass per step at 512^3 when MonodomainSplittingSolver._can_fuse asked for `.size`)
    def __len__(self):
        return self._f.num_values

    @property
    def size(self):
        return self._f.num_values

    @property
    def shape(self):
        return (self._f.num_values,)

    @property
    def dtype(self):
        return np.dtype(np.float64)

    @property
    def ndim(self):
        return 1

    def copy(self):
        return self._f._host().copy()

    def max(self):
        return self._f.field.minmax()[1] if self.size else -np.inf

    def min(self):
        return self._f.field.minmax()[0] if self.size else np.inf

    def __getitem__(self, key):
        return self._f._host()[key]

    def __setitem__(self, key, value):
        f = self._f
        if isinstance(value, LazyArray):
            if isinstance(key, slice) and key == slice(None):
                src = value._f.field
                dst = f.writable_field()
                if dst is not src:
                    dst.copy_from(src)  # device-to-device, no host round trip
                f._touch()
                return
            value = np.asarray(value)
        if isinstance(key, slice) and key == slice(None):
            f.writable_field().set(value)
            f._touch()
            return
        host = f._host().copy()
        host[key] = value
        f.writable_field().set(host)
        f._touch()

    # arithmetic falls back to the host copy
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(np.asarray(i) if isinstance(i, LazyArray) else i for i in inputs)
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __add__(self, o):
        return np.asarray(self) + o

    def __sub__(self, o):
        return np.asarray(self) - o

    def __mul__(self, o):
        return np.asarray(self) * o

    def __repr__(self):
        return f"LazyArray({np.asarray(self)!r})"


class _Vector:
    def __init__(self, function):
        self.array = LazyArray(function)
        self._f = function

    def scatter_forward(self):
        pass


class Function:
    """``dolfinx.fem.Function(V)``: nodal values of a P1 function held on the device.

    A function either owns its storage or is a read-only *alias* of another field (the fused split
    step leaves ``pde.state``, ``pde.v_`` and ``ode.v_ode`` aliased to the V row of the state array
    instead of copying it three times; the alias is materialised into the function's own storage
    the moment the two would diverge)."""

    def __new__(cls, V: FunctionSpace = None, *a, **kw):
        if isinstance(V, VectorFunctionSpace):
            return VectorFunction(V, *a, **kw)
        if V is not None and getattr(V, "family", None) == "DG" and V.degree == 0:
            return CellFunction(V, *a, **kw)
        return super().__new__(cls)

    def __init__(self, V: FunctionSpace, name: str = "f", field=None):
        from ._device import Context

        self.function_space = V
        self.name = name
        mesh = V.mesh
        self._ctx = Context.default()
        if field is not None:
            self._own = field
        elif V.is_p1:
            self._own = self._ctx.field(mesh.num_nodes, mesh.plane)
        else:  # P2 / DG1 ODE spaces: a plain device vector (never an operand of the stencil kernels)
            self._own = self._ctx.field(V.num_dofs, 0)
        self._alias = None
        self._alias_sync = None
        self._version = 0
        self._cache = None
        self._x = _Vector(self)

    @property
    def x(self):
        return self._x

    def ufl_element(self):
        return self.function_space.ufl_element()

    # ---- storage ---------------------------------------------------------------------------
    @property
    def num_values(self) -> int:
        """Number of (local) degrees of freedom; does not synchronise anything."""
        return (self._alias if self._alias is not None else self._own).n

    @property
    def field(self):
        """Field to READ the current values from."""
        if self._alias is not None:
            if self._alias_sync is not None:
                self._alias_sync()  # e.g. a deferred update of the aliased row (fused split step)
            return self._alias
        return self._own

    def writable_field(self, overwrite_all: bool = True):
        """Field to WRITE into; ends an alias (copying the aliased values first unless everything is
        about to be overwritten).  Call ``_touch()`` after the write."""
        if self._alias is not None:
            if not overwrite_all:
                self._own.copy_from(self.field)
            self._alias = None
            self._alias_sync = None
        return self._own

    def alias_to(self, field, sync=None) -> None:
        """``sync``: callable run before the aliased field is read (brings it up to date)."""
        self._alias = field
        self._alias_sync = sync
        self._touch()

    def materialize(self) -> None:
        if self._alias is not None:
            self._own.copy_from(self.field)
            self._alias = None
            self._alias_sync = None

    def _touch(self):
        """Call after any device-side modification of the values."""
        self._version += 1

    def _host(self) -> np.ndarray:
        if self._cache is None or self._cache[0] != self._version:
            arr = self.field.numpy()
            arr.setflags(write=False)
            self._cache = (self._version, arr)
        return self._cache[1]

    def interpolate(self, f) -> None:
        x = self.function_space.tabulate_dof_coordinates().T
        if isinstance(f, Expression):
            f = f.expr
        if isinstance(f, Expr):
            vals = f.evaluate(x)
        elif isinstance(f, Function):
            vals = np.asarray(f.x.array)
        else:
            vals = f(x)
        fld = self.writable_field()
        fld.set(np.broadcast_to(np.asarray(vals, dtype=np.float64), (fld.n,)))
        self._touch()

    def copy(self):
        g = Function(self.function_space, self.name)
        g._own.copy_from(self.field)
        g._touch()
        return g


def _locate_points(mesh: "Mesh", points):
    """Vertices (GLOBAL ids) and barycentric weights of the simplex each point lies in, cached per mesh: callers
    evaluate the same probe points every time step."""
    pts = np.atleast_2d(np.asarray(points, dtype=np.float64))
    d = mesh.dim
    cache = mesh.__dict__.setdefault("_probe_cache", {})
    key = pts.tobytes()
    if key in cache:
        return cache[key]
    idx = np.zeros((len(pts), 4), dtype=np.int64)
    wts = np.zeros((len(pts), 4), dtype=np.float64)
    lower, h, n = np.array(mesh.lower), np.array(mesh.h), np.array(mesh.n)
    for k, p in enumerate(pts):
        rel = (p[:d] - lower) / h
        c = np.clip(np.floor(rel).astype(np.int64), 0, n - 1)
        box = c[0] + (n[0] * (c[1] + (n[1] * c[2] if d == 3 else 0)) if d >= 2 else 0)
        best = None
        for s in range(mesh.simplices_per_cell):
            verts = mesh.cell_vertices(np.array([box * mesh.simplices_per_cell + s]))[0]
            X = _node_xyz(mesh, verts)[:, :d]
            A = np.hstack([np.ones((d + 1, 1)), X])
            lam = np.linalg.solve(A.T, np.concatenate([[1.0], p[:d]]))
            if best is None or lam.min() > best[0]:
                best = (lam.min(), lam, verts)
        idx[k, : d + 1] = best[2]
        wts[k, : d + 1] = best[1]
    if len(cache) < 64:
        cache[key] = (idx, wts)
    return idx, wts


def evaluate_function(f: Function, points) -> np.ndarray:
    """``scifem.evaluate_function(f, points)`` for P1 functions on the box mesh; on a slab-decomposed mesh every
    rank returns the same values (partial sums over the vertices it owns, all-reduced)."""
    idx, wts = _locate_points(f.function_space.mesh, points)
    return _probe(f, idx, wts)


class ProbeRecorder:
    """Point values of a P1 function recorded step by step ON THE DEVICE: ``record()`` enqueues one tiny kernel and
    returns, ``values()`` reads the whole record back ((steps, points) array).  For time loops that look at probes
    every step -- the reference's Niederer demo calls ``scifem.evaluate_function`` once per step
    (demos/niederer_benchmark.py:285-291), which on a GPU is one host synchronisation per step; the recorded values
    are the ones ``evaluate_function`` returns (same kernel)."""

    def __init__(self, f: Function, points, capacity: int = 4096):
        mesh = f.function_space.mesh
        idx, wts = _locate_points(mesh, points)
        if mesh.comm.size > 1:
            idx, wts = _local_probe_args(mesh, idx, wts)
        self._f, self._mesh = f, mesh
        self._idx, self._wts = np.ascontiguousarray(idx), np.ascontiguousarray(wts)
        self.npts = len(self._idx)
        self.capacity = int(capacity)
        self._buf = f._ctx.zeros(self.capacity * self.npts)
        self._rows = 0       # rows in the device buffer
        self._done = []      # host copies of full buffers

    def record(self) -> None:
        import ctypes as C

        from . import _hip

        if self._rows == self.capacity:
            self._done.append(self._read())
            self._rows = 0
        ctx = self._f._ctx
        out = C.c_void_p(self._buf.data_ptr() + 8 * self._rows * self.npts)
        _hip.check(ctx.lib.beat_field_probe_record(ctx.handle, self._f.field.ptr, self._idx.ctypes.data_as(C.c_void_p),
                                                   self._wts.ctypes.data_as(C.c_void_p), self.npts, out))
        self._rows += 1

    def _reserve(self, nrows: int):
        """Device address of the next free row and how many rows (<= nrows) may be written there; the caller commits
        them with ``_commit`` (MonodomainSplittingSolver.solve records a batch of steps in one library call)."""
        if self._rows == self.capacity:
            self._done.append(self._read())
            self._rows = 0
        return self._buf.data_ptr() + 8 * self._rows * self.npts, min(int(nrows), self.capacity - self._rows)

    def _commit(self, nrows: int) -> None:
        self._rows += int(nrows)

    def _read(self) -> np.ndarray:
        part = self._buf[: self._rows * self.npts].cpu().numpy().reshape(self._rows, self.npts).copy()
        if self._mesh.comm.size > 1:  # every rank holds the partial sums over the vertices it owns
            part = self._mesh.comm.allreduce_array(part.ravel()).reshape(part.shape)
        return part

    def __len__(self) -> int:
        return sum(len(a) for a in self._done) + self._rows

    def values(self) -> np.ndarray:
        """All rows recorded so far, (steps, points).  Synchronises."""
        parts = self._done + ([self._read()] if self._rows else [])
        return np.concatenate(parts) if parts else np.zeros((0, self.npts))


def _local_probe_args(mesh: Mesh, idx: np.ndarray, wts: np.ndarray):
    """GLOBAL vertex ids / weights -> this rank's share: vertices of other slabs get weight 0 (and a valid dummy
    index); summing the ranks' partial values gives the interpolated value without any ghost data."""
    lo, hi = mesh.slab.z0 * mesh.plane, mesh.slab.z1 * mesh.plane
    own = (idx >= lo) & (idx < hi)
    return np.where(own, idx - lo, 0).astype(np.int64), np.where(own, wts, 0.0)


def _probe(f: "Function", idx: np.ndarray, wts: np.ndarray) -> np.ndarray:
    import ctypes as C

    from . import _hip

    mesh = f.function_space.mesh
    if mesh.comm.size > 1:
        idx, wts = _local_probe_args(mesh, idx, wts)
        idx, wts = np.ascontiguousarray(idx), np.ascontiguousarray(wts)
    out = np.zeros(len(idx))
    ctx = f._ctx
    _hip.check(ctx.lib.beat_field_probe(ctx.handle, f.field.ptr, idx.ctypes.data_as(C.c_void_p),
                                        wts.ctypes.data_as(C.c_void_p), len(idx), out.ctypes.data_as(C.c_void_p)))
    if mesh.comm.size > 1:
        out = mesh.comm.allreduce_array(out)
    return out.reshape(-1, 1)


def _node_xyz(mesh: Mesh, ids: np.ndarray) -> np.ndarray:
    nx, ny, _ = mesh.shape_global
    ix, iy, iz = ids % nx, (ids // nx) % ny, ids // (nx * ny)
    lo = list(mesh.lower) + [0.0] * (3 - mesh.dim)
    hh = list(mesh.h) + [0.0] * (3 - mesh.dim)
    return np.stack([lo[0] + hh[0] * ix, lo[1] + hh[1] * iy, lo[2] + hh[2] * iz], axis=1)


class VTXWriter:
    """No-op stand-in for dolfinx.io.VTXWriter (output is not part of the hot path)."""

    def __init__(self, *a, **kw):
        pass

    def write(self, t):
        pass

    def close(self):
        pass
