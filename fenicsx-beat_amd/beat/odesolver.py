"""ODE (reaction) solvers -- interface of src/beat/odesolver.py:24-354.

``fun`` follows the reference's convention, ``fun(states=, t=, parameters=, dt=) -> new states``.
When ``fun`` is one of the built-in device models (``beat.models``) the ``(S, N)`` state array
lives in HBM and a step is one kernel launch; any other callable is the caller's own NumPy code and
is run exactly as the reference runs it (host arrays, the transmembrane potential staged to and
from the device field)."""

from __future__ import annotations

import abc
import ctypes as C
import logging
from dataclasses import dataclass, field
from typing import Any, Callable, NamedTuple

import numpy as np

from . import _hip, grid
from .models._base import DeviceModel, DeviceParameters, host_and_device_parameters
from .telemetry import BaseMonitor, NullMonitor
from .utils import local_project

EPS = 1e-12
logger = logging.getLogger(__name__)


class ODEResults(NamedTuple):
    y: np.ndarray
    t: np.ndarray


def solve(fun, t_bound: float, states, V, V_index: int, dt: float, parameters, t0: float = 0.0, extra=None):
    """Free-running multi-point loop of odesolver.py:24-43 (``states`` is advanced in place)."""
    if extra is None:
        extra = {}
    i = 0
    t = t0
    while t + dt < t_bound:
        new = fun(states=states, t=t, parameters=parameters, dt=dt, **extra)
        if new is not None and new is not states:
            states[:] = new
        V[i, :] = states[V_index, :]
        i += 1
        t += dt


@dataclass
class ODESystemSolver:
    fun: Callable
    states: np.ndarray
    parameters: np.ndarray
    missing_variables: np.ndarray | None = None
    _kwargs: dict = field(default_factory=dict)
    monitor: BaseMonitor = field(default_factory=NullMonitor)

    def __post_init__(self):
        if self.missing_variables is not None:
            self._kwargs["missing_variables"] = self.missing_variables

    @property
    def num_points(self) -> int:
        return self.states.shape[1]

    @property
    def num_states(self) -> int:
        return self.states.shape[0]

    def step(self, t0: float, dt: float) -> None:
        with self.monitor.track_time("ode_total_step"):
            with self.monitor.track_time("ode_function_call"):
                updated_states = self.fun(states=self.states, t=t0, parameters=self.parameters, dt=dt, **self._kwargs)
            with self.monitor.track_time("ode_state_update"):
                self.states[:] = updated_states


def _initial_values(init_states, shape, on_device: bool):
    """(S, N) initial values as the reference builds them (odesolver.py:149-153): a full array is copied, a
    per-state vector is broadcast to every point.  On the device the broadcast is done row by row in HBM
    (no (S, N) host array), signalled by returning the 1-D vector."""
    if np.shape(init_states) == shape:
        return np.copy(init_states)
    vec = np.asarray(init_states, dtype=np.float64)
    if on_device and vec.shape == (shape[0],):
        return vec
    values = np.zeros(shape)
    values.T[:] = init_states
    return values


class _DeviceODE:
    """(S, N) state array in HBM advanced by a built-in model kernel."""

    def __init__(self, ctx, model: DeviceModel, num_states, n, plane, parameters, monitor):
        from ._device import StateArray

        if num_states != model.num_states:
            raise ValueError(f"{model.name} has {model.num_states} states, num_states={num_states}")
        if hasattr(model, "register"):  # a model generated from an .ode file (beat.models.from_ode): its source goes to the library
            model.register()
        self.ctx = ctx
        self.model = model
        self.n = n
        self.states = StateArray(ctx, num_states, n, plane)
        self.parameters = parameters
        self.monitor = monitor
        self._ppn = None
        self._ppn_host = None  # what the device copy was uploaded from (None: the per-node route's cache is not current)
        self._dp_key = None    # (id, version) of the DeviceParameters handle the cached route was derived from
        self._per_node_args = None
        self._sparse = None    # (uniform vector, indices of the varying rows, their (K, N) device rows) or None
        self.classes = None    # (marker bytes on the device, class table, number of classes): see set_classes
        self.explicit_classes = False  # the owner set the classes itself (DolfinMultiODESolver); else they follow the parameters
        self.node_map = None   # (int32 node of the PDE grid per entry, the field holding the potential): compact layout
        self._class_host = None

    def set_classes(self, markers_dev, param_sets) -> None:
        """One launch for several uniform parameter sets (beat_ode_step_classes): ``markers_dev`` = a byte per node naming
        its set (255: none, the node is not advanced), ``param_sets`` = the sets in class order.  Calling it again with
        the same byte row only refreshes the table, and only if a set changed (the reference hands ``fun`` the live
        parameter arrays every step, odesolver.py:70-76)."""
        P = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.float64) for p in param_sets]))
        if P.ndim != 2 or P.shape[1] != self.model.num_parameters or not 1 <= P.shape[0] <= _hip.MAX_CLASSES:
            raise ValueError(f"{self.model.name}: {P.shape[0]} parameter sets of {P.shape[1:]} values "
                             f"(expected 1..{_hip.MAX_CLASSES} sets of {self.model.num_parameters})")
        lib = self.ctx.lib
        if self.classes is None or self.classes[2] != P.shape[0]:
            stride = C.c_int()
            _hip.check(lib.beat_ode_class_table_doubles(self.model.model_id, C.byref(stride)))
            table = self.ctx.zeros(stride.value * P.shape[0])
            self._class_host = None
        else:
            table = self.classes[1]
        if self._class_host is None or not np.array_equal(self._class_host, P):
            _hip.check(lib.beat_ode_class_table_fill(self.ctx.handle, self.model.model_id, P.ctypes.data_as(C.c_void_p), P.shape[1],
                                                     P.shape[0], C.c_void_p(table.data_ptr())))
            self._class_host = P.copy()
        self.classes = (markers_dev, table, P.shape[0])

    def set_initial(self, values) -> None:
        if values.ndim == 1:
            for k, val in enumerate(values):
                self.states.row_field(k).fill(float(val))
        else:
            self.states.set(values)

    def _classify(self, t):
        """Per-node parameters (P, N) that are piecewise constant -- demos/pace_train.py:133-167 zeroes two conductances
        in half of the cable -- are a handful of uniform parameter sets: the distinct columns become classes, a byte per
        node names its class, and the step runs the class kernel (uniform parameters in scalar registers, no per-node
        rows to load: 424 B/node for TP06, 896 for ToR-ORd) instead of the per-node one.  Returns (marker bytes on the
        device, (classes, P) host array) or None when the columns are not few (a smooth gradient: per-node kernel).
        BEAT_PARAM_CLASSES=0 turns the analysis off."""
        import os

        if os.environ.get("BEAT_PARAM_CLASSES", "1") == "0":
            return None
        torch = self.ctx.torch
        varying = (t != t[:, :1]).any(dim=1)
        n = t.shape[1]
        if not bool(varying.any()):
            return torch.zeros(n, dtype=torch.uint8, device=t.device), t[:, :1].T.cpu().numpy()
        cols = t[varying]
        stride = max(1, n // 65536)  # a sample first: a smooth field shows at once
        if torch.unique(cols[:, ::stride], dim=1).shape[1] > _hip.MAX_CLASSES:
            return None
        uniq, inv = torch.unique(cols, dim=1, return_inverse=True)
        if uniq.shape[1] > _hip.MAX_CLASSES:
            return None
        first = torch.full((uniq.shape[1],), n, dtype=torch.int64, device=t.device)
        first.scatter_reduce_(0, inv, torch.arange(n, device=t.device), reduce="amin")
        return inv.to(torch.uint8), t[:, first].T.contiguous().cpu().numpy()

    def _per_node_or_classes(self, tensor, num_rows):
        """(host params, P, device rows, ld) for per-node parameters held in ``tensor``: the class route if they allow it,
        else -- when at most sixteen ROWS vary over the nodes (smooth gradients in a few conductances; four without run-time
        compilation) -- those rows alone
        next to the uniform vector (``self._sparse``: beat_ode_step_rows reads 8 B per varying row and node instead of 8 P),
        else all P rows."""
        per_node = (None, num_rows, C.c_void_p(tensor.data_ptr()), self.n)
        self._sparse = None
        found = self._classify(tensor)
        if found is not None:
            self.set_classes(found[0], list(found[1]))
            self._per_node_args = per_node  # for the caller the class kernel does not serve (a mirror of another row than V)
            return None, num_rows, None, 0
        self.classes = None
        import os

        # (a model registered as source has no compiled sparse-row instance: all its rows are read)
        if os.environ.get("BEAT_PARAM_SPARSE", "1") != "0" and tensor.shape[1] == self.n and self.model.model_id < _hip.CUSTOM_MODEL_BASE:
            varying = (tensor != tensor[:, :1]).any(dim=1)
            idx = varying.nonzero().flatten().cpu().numpy().astype(np.int32)
            # (up to 16 varying rows on a kernel instance compiled for their indices, 4 where that cannot be had)
            jit = os.environ.get("BEAT_JIT", "1") != "0" and self.ctx.lib.beat_ode_jit_stats(None) == 1
            if 1 <= len(idx) <= (_hip.MAX_SPARSE_ROWS if jit else _hip.MAX_SPARSE_ROWS_RT):
                rows = tensor[varying].contiguous()  # (K, N): the only parameter data a step reads from memory
                uniform = np.ascontiguousarray(tensor[:, 0].cpu().numpy(), dtype=np.float64)
                self._sparse = (uniform, np.ascontiguousarray(idx), rows)
        return per_node

    def _forget_routes(self) -> None:
        """The parameters are no per-node rows (any more): no classes, and neither cached route may be taken for current
        again -- a classified (P, N) array, then a vector, then the same array must classify again (the vector branch
        cleared the classes the cached arguments rely on)."""
        self.classes = None
        self._sparse = None
        self._ppn_host = None
        self._dp_key = None

    def _param_args(self):
        p = self.parameters
        if p is None:
            self._forget_routes()
            return None, 0, None, 0
        if isinstance(p, DeviceParameters):  # resident handle: nothing to check or move per step
            _, dev, ld = host_and_device_parameters(self.ctx, p, self.model.num_parameters, self.n)
            key = (id(p), p.version)
            if self._dp_key != key:  # new values: look at them once
                self._ppn_host = None  # (the classes now follow THIS handle: the NumPy route's cache is stale)
                self._dp_args = self._per_node_or_classes(dev, p.shape[0])
                self._dp_key = key
            return self._dp_args
        p = np.asarray(p, dtype=np.float64)
        if p.ndim == 1:
            hp = np.ascontiguousarray(p)
            self._keep = hp
            self._forget_routes()
            return hp.ctypes.data_as(C.c_void_p), len(hp), None, 0
        # per-node NumPy parameters: the reference hands the live array to ``fun`` every step (odesolver.py:70-76),
        # so any in-place edit must reach the kernel.  Exact comparison with the copy that was uploaded (one host
        # pass over (P, N) per step; pass a DeviceParameters handle to avoid it on large grids).
        if self._ppn is None or self._ppn_host is None or self._ppn_host.shape != p.shape or not np.array_equal(self._ppn_host, p):
            _, self._ppn, _ = host_and_device_parameters(self.ctx, p, self.model.num_parameters, self.n)
            self._ppn_host = p.copy()
            self._dp_key = None  # (see above)
            self._ppn_args = self._per_node_or_classes(self._ppn, p.shape[0])
        return self._ppn_args

    def step(self, t0, dt, v_index=0, v_copy=None, pending_ops=None, v_row=None):
        """``pending_ops``: diffusion operators whose last solve deferred its update of ``v_row`` (row v_index of
        these states): the kernel adds it while loading the potential (beat_ode_step_pending)."""
        # (classes set explicitly -- one model for several markers -- stay; otherwise the parameters decide the route)
        hp, npar, ppn, pld = (None, 0, None, 0) if self.explicit_classes else self._param_args()
        use_classes = self.classes is not None
        if use_classes and not self.explicit_classes and v_copy is not None and self.model.v_name \
                and int(v_index) != self.model.state_index(self.model.v_name):
            # the class kernel mirrors the model's own potential row only: per-node parameters that were routed to classes
            # go back to the per-node kernel, which mirrors any row (what the reference's v_index means, odesolver.py:135-146)
            use_classes = False
            hp, npar, ppn, pld = self._per_node_args
        pend = None
        behind = False  # enqueue this launch BEHIND a solve that is still open (pending = -1: beat_ode_step_pending)
        if pending_ops is not None and getattr(pending_ops, "open_x", None) is not None:
            model_v = self.model.state_index(self.model.v_name) if self.model.v_name else -1
            long_ring = len(pending_ops.ring) > 6  # (only the class kernel takes more than six pending directions)
            if (v_row is not None and int(v_index) == model_v and pending_ops.open_x.ptr.value == v_row.ptr.value
                    and (self.node_map is None or self.node_map[1].ptr.value == v_row.ptr.value)
                    and (use_classes or not long_ring)):
                behind = True
                pend = (v_row, 0, -1)
            else:
                pending_ops.solve_finish()
        if not behind and pending_ops is not None and pending_ops.pending is not None:
            model_v = self.model.state_index(self.model.v_name) if self.model.v_name else -1
            if (v_row is not None and int(v_index) == model_v
                    and pending_ops.pending[0].ptr.value == v_row.ptr.value
                    and (self.node_map is None or self.node_map[1].ptr.value == v_row.ptr.value)):
                pend = pending_ops.pending
                pending_ops.pending = None
            else:
                pending_ops.flush_pending()
        with self.monitor.track_time("ode_total_step"):
            with self.monitor.track_time("ode_function_call"):
                if use_classes:
                    mk, table, ncls = self.classes
                    if int(v_index) != self.model.state_index(self.model.v_name) and pend is not None:
                        raise ValueError("a pending update needs the model's own potential row")
                    nmap, vfield = (None, None) if self.node_map is None else (C.c_void_p(self.node_map[0].data_ptr()), self.node_map[1].ptr)
                    _hip.check(self.ctx.lib.beat_ode_step_classes(
                        self.ctx.handle, self.model.model_id, self.states.ptr, self.n, self.states.ld, C.c_void_p(table.data_ptr()),
                        ncls, C.c_void_p(mk.data_ptr()), float(t0), float(dt), int(v_index), None if v_copy is None else v_copy.ptr,
                        nmap, vfield,
                        pending_ops.handle if pend is not None else None, pending_ops.ring[0].ptr if pend is not None else None,
                        pending_ops.fld if pend is not None else 0, int(pend[2]) if pend is not None else 0))
                elif self._sparse is not None and ppn is not None and not self.explicit_classes:
                    uni, idx, rows = self._sparse
                    _hip.check(self.ctx.lib.beat_ode_step_rows(
                        self.ctx.handle, self.model.model_id, self.states.ptr, self.n, self.states.ld, uni.ctypes.data_as(C.c_void_p),
                        len(uni), idx.ctypes.data_as(C.c_void_p), len(idx), C.c_void_p(rows.data_ptr()), self.n, float(t0), float(dt),
                        int(v_index), None if v_copy is None else v_copy.ptr,
                        pending_ops.handle if pend is not None else None, pending_ops.ring[0].ptr if pend is not None else None,
                        pending_ops.fld if pend is not None else 0, int(pend[2]) if pend is not None else 0))
                elif pend is not None:
                    _hip.check(
                        self.ctx.lib.beat_ode_step_pending(
                            self.ctx.handle, self.model.model_id, self.states.ptr, self.n, self.states.ld, hp, npar, ppn,
                            pld, float(t0), float(dt), int(v_index), None if v_copy is None else v_copy.ptr,
                            pending_ops.handle, pending_ops.ring[0].ptr, pending_ops.fld, int(pend[2]))
                    )
                else:
                    _hip.check(
                        self.ctx.lib.beat_ode_step(self.ctx.handle, self.model.model_id, self.states.ptr, self.n,
                                                   self.states.ld, hp, npar, ppn, pld, float(t0), float(dt), int(v_index),
                                                   None if v_copy is None else v_copy.ptr)
                    )
            if behind:  # the call has finished the solve it was enqueued behind: take its record
                pending_ops.finished_behind()
            with self.monitor.track_time("ode_state_update"):
                pass  # updated in place by the kernel


class BaseDolfinODESolver(abc.ABC):
    v_ode: grid.Function
    v_pde: grid.Function
    _metadata: dict[str, Any] | None = None

    def _initialize_metadata(self):
        if self.v_ode.ufl_element().family_name == "Quadrature":
            self._metadata = {"quadrature_degree": self.v_ode.ufl_element().degree()}
        else:
            self._metadata = None

    @abc.abstractmethod
    def to_dolfin(self) -> None: ...

    @abc.abstractmethod
    def from_dolfin(self) -> None: ...

    def ode_to_pde(self) -> None:
        local_project(self.v_ode, self.v_pde.function_space, self.v_pde)

    def pde_to_ode(self) -> None:
        local_project(self.v_pde, self.v_ode.function_space, self.v_ode)

    @abc.abstractmethod
    def step(self, t0: float, dt: float) -> None: ...

    @property
    @abc.abstractmethod
    def full_values(self): ...


@dataclass
class DolfinODESolver(BaseDolfinODESolver):
    v_ode: grid.Function
    v_pde: grid.Function
    init_states: np.ndarray
    parameters: np.ndarray
    fun: Callable
    num_states: int
    v_index: int = 0
    missing_variables: np.ndarray | None = None
    num_missing_variables: int = 0
    monitor: BaseMonitor = field(default_factory=NullMonitor)

    def __post_init__(self):
        self._aliases: list[grid.Function] = []
        self._pending_ops = None  # diffusion operators that may hold a deferred update of the V row (fused step)
        self.on_device = isinstance(self.fun, DeviceModel)
        values = _initial_values(self.init_states, self.shape, self.on_device)
        if self.on_device:
            mesh = self.v_ode.function_space.mesh
            plane = mesh.plane if self.v_ode.function_space.is_p1 else 0  # only P1 rows are stencil operands
            self._dev = _DeviceODE(self.v_ode._ctx, self.fun, self.num_states, self.num_points, plane,
                                   self.parameters, self.monitor)
            self._dev.set_initial(values)
            self._v_row = self._dev.states.row_field(self.v_index)
            self._ode = self._dev
        else:
            self._values = values
            self._ode = ODESystemSolver(fun=self.fun, states=self._values, parameters=self.parameters,
                                        missing_variables=self.missing_variables, monitor=self.monitor)
        self._initialize_metadata()

    # ---- alias bookkeeping (see grid.Function) ------------------------------------------------
    def _sync_v(self):
        """Bring the V row up to date if the last fused diffusion solve left its final update to the next ionic
        kernel (deferred-x PCG): everything that reads or overwrites the row outside that kernel calls this."""
        if self._pending_ops is not None:
            self._pending_ops.flush_pending()

    def _release_aliases(self):
        """The V row is about to change outside the fused step: give every function that merely
        aliases it its own copy of the current values first."""
        self._sync_v()
        for f in self._aliases:
            if f._alias is self._v_row:
                f.materialize()
        self._aliases = []

    # ---- reference interface --------------------------------------------------------------------
    def to_dolfin(self) -> None:
        """values[v_index] -> v_ode  (odesolver.py:164-166)"""
        if self.on_device:
            if self.v_ode._alias is self._v_row:
                return
            self._sync_v()
            self.v_ode.writable_field().copy_from(self._v_row)
            self.v_ode._touch()
        else:
            self.v_ode.x.array[:] = self._values[self.v_index, :]

    def from_dolfin(self) -> None:
        """v_ode -> values[v_index]  (odesolver.py:168-170)"""
        if self.on_device:
            if self.v_ode._alias is self._v_row:
                return
            self._release_aliases()
            self._v_row.copy_from(self.v_ode.field)
        else:
            self._values[self.v_index, :] = np.asarray(self.v_ode.x.array)

    @property
    def values(self):
        if self.on_device:
            self._sync_v()
            return self._dev.states.numpy()
        return self._values

    @property
    def num_parameters(self) -> int:
        return len(self.parameters)

    @property
    def shape(self) -> tuple[int, int]:
        return (self.num_states, self.num_points)

    @property
    def shape_missing_values(self) -> tuple[int, int]:
        return (self.num_missing_variables, self.num_points)

    @property
    def num_points(self) -> int:
        return self.v_ode.x.array.size

    def step(self, t0: float, dt: float):
        if self.on_device:
            self._release_aliases()
            self._dev.parameters = self.parameters
            self._dev.step(t0, dt)
        else:
            self._ode.step(t0=t0, dt=dt)

    @property
    def full_values(self):
        return self.values

    def set_values(self, values) -> None:
        """Overwrite the whole (S, N) state array (e.g. with a pre-paced steady state)."""
        values = np.asarray(values, dtype=np.float64)
        if self.on_device:
            self._release_aliases()
            self._dev.states.set(np.broadcast_to(values.reshape(self.num_states, -1), self.shape))
        else:
            self._values[:] = values.reshape(self.num_states, -1)

    def assign_all_states(self, functions: list[grid.Function]) -> None:
        vals = self.values
        assert len(functions) == vals.shape[0], "Number of functions must match number of states"
        for index, f in enumerate(functions):
            f.x.array[:] = vals[index, :]

    def states_to_dolfin(self, names: list[str] | None = None) -> list[grid.Function]:
        V = self.v_ode.function_space
        num_states = self.num_states
        if names is not None:
            msg = f"Number of names must match number of states, got {len(names)} names, but number of states is {num_states}"
            assert len(names) == num_states, msg
        else:
            names = [f"state_{i}" for i in range(num_states)]
        functions = [grid.Function(V, name=name) for name in names]
        self.assign_all_states(functions)
        return functions


@dataclass
class DolfinMultiODESolver(BaseDolfinODESolver):
    """One cell model / parameter set per marker value (odesolver.py:228-354)."""

    v_ode: grid.Function
    v_pde: grid.Function
    markers: grid.Function
    init_states: dict
    parameters: dict
    fun: dict
    num_states: dict
    v_index: dict
    monitor: BaseMonitor = field(default_factory=NullMonitor)

    def __post_init__(self):
        if self.v_ode.x.array.size != self.markers.x.array.size:
            raise RuntimeError("Marker and voltage need to be in the same function space")
        self._marker_values = tuple(self.init_states.keys())
        marker_arr = np.asarray(self.markers.x.array)
        self._num_points, self._inds, self._idx_dev, self._odes, self._values = {}, {}, {}, {}, {}
        ctx = self.v_ode._ctx
        self._ctx = ctx
        self.on_device = all(isinstance(f, DeviceModel) for f in self.fun.values())
        if self.on_device:
            for f in self.fun.values():  # models generated from .ode files get their ids now (markers sharing one are compared below)
                if hasattr(f, "register"):
                    f.register()
        self._initialize_full_values()
        self._aliases: list[grid.Function] = []
        self._pending_ops = None
        self._marked = self.on_device and self._one_launch_possible()
        if self._marked:
            self._setup_one_launch(marker_arr)
            self._initialize_metadata()
            return
        for marker in self._marker_values:
            where = marker_arr == marker
            n_m = int(where.sum())
            self._num_points[marker] = n_m
            self._inds[marker] = where
            values = _initial_values(self.init_states[marker], self.shape(marker), self.on_device)
            if self.on_device:
                dev = _DeviceODE(ctx, self.fun[marker], self.num_states[marker], n_m, 0, self.parameters[marker],
                                 self.monitor)
                dev.set_initial(values)
                self._odes[marker] = dev
                self._idx_dev[marker] = ctx.from_numpy(np.nonzero(where)[0].astype(np.int64))
            else:
                self._values[marker] = values
                self._odes[marker] = ODESystemSolver(fun=self.fun[marker], states=values,
                                                     parameters=self.parameters[marker], monitor=self.monitor)
        self._initialize_metadata()

    # ---- one launch for all markers (the markers share one device model: cell types, parameter regions) ----------------
    def _one_launch_possible(self) -> bool:
        """All markers run the same built-in model with uniform (1-D) parameter sets on the PDE's P1 space: then the
        nodes live in ONE (S, N) state array with a byte per node naming the node's parameter set, and a step is one
        kernel launch (beat_ode_step_classes) instead of one per marker plus a scatter and a gather of the potential
        (BEAT_MULTI_ONE_LAUNCH=0 keeps the per-marker arrays: the reference's literal data layout)."""
        import os

        if os.environ.get("BEAT_MULTI_ONE_LAUNCH", "1") == "0":
            return False
        ms = self._marker_values
        funs = [self.fun[m] for m in ms]
        return (
            1 <= len(ms) <= _hip.MAX_CLASSES
            and all(f.model_id == funs[0].model_id for f in funs)
            and len({int(self.num_states[m]) for m in ms}) == 1
            and len({int(self.v_index[m]) for m in ms}) == 1
            and all(not isinstance(self.parameters[m], DeviceParameters) and np.ndim(self.parameters[m]) == 1 for m in ms)
            and self.v_ode.function_space.is_p1
        )

    def _setup_one_launch(self, marker_arr) -> None:
        import os

        ctx, ms = self._ctx, self._marker_values
        model, S, vi = self.fun[ms[0]], int(self.num_states[ms[0]]), int(self.v_index[ms[0]])
        n = marker_arr.size
        mesh = self.v_ode.function_space.mesh
        cls_full = np.full(n, 255, dtype=np.uint8)
        for k, marker in enumerate(ms):
            where = marker_arr == marker
            self._inds[marker] = where
            self._num_points[marker] = int(where.sum())
            cls_full[where] = k
        self._marked_any = cls_full != 255
        self._all_marked = bool(self._marked_any.all())
        self._idx_all = None if self._all_marked else ctx.from_numpy(np.nonzero(self._marked_any)[0].astype(np.int64))
        # Layout.  The ionic kernels are bound by fp64 issue: a lane without a cell costs what a busy one does.  When the
        # nodes that carry a model are a fraction of the grid (the wall of a voxelised geometry inside its box: 26 %) the
        # state array holds only those -- plus any other node of the tissue, which still diffuses -- and a node map
        # tells the kernel where each one's potential lives in the PDE's field; otherwise the array spans the grid and
        # its potential row IS that field.  BEAT_MULTI_COMPACT = 0 | 1 forces one or the other.
        tissue = mesh.node_active() if mesh.active is not None else np.ones(n, dtype=bool)
        held = self._marked_any | tissue
        # wavefronts (64 consecutive nodes) that would meet more than one class: each costs one pass per class
        seg = np.full(((n + 63) // 64) * 64, 255, dtype=np.uint8)
        seg[:n] = cls_full
        seg = seg.reshape(-1, 64)
        present = np.zeros(len(seg), dtype=np.int64)
        for k in range(len(ms)):
            present += (seg == k).any(axis=1)
        mixed = float((present > 1).sum()) / max(1, int((present > 0).sum()))
        want = os.environ.get("BEAT_MULTI_COMPACT")
        compact = (held.mean() < 0.9 or mixed > 0.05) if want is None else (want == "1")
        self._vi = vi
        if compact:
            # one run of whole 256-node tiles per class (then the tissue nodes without a model), nodes ascending inside a run
            tile = 256
            runs = [(k, np.nonzero(self._inds[m])[0]) for k, m in enumerate(ms)]
            rest = np.nonzero(held & ~self._marked_any)[0]
            if rest.size:
                runs.append((255, rest))
            node_parts, cls_parts, pos, real_pos, off = [], [], {}, [], 0
            for k, nodes in runs:
                pad = (-nodes.size) % tile
                node_parts += [nodes, np.zeros(pad, dtype=np.int64)]
                cls_parts += [np.full(nodes.size, k, dtype=np.uint8), np.full(pad, 254, dtype=np.uint8)]
                if k != 255:
                    pos[ms[k]] = off + np.arange(nodes.size)
                real_pos.append(off + np.arange(nodes.size))
                off += nodes.size + pad
            node_idx = np.concatenate(node_parts)
            cls = np.concatenate(cls_parts)
            real_pos = np.concatenate(real_pos)
            if node_idx.size == 0 or node_idx.max() >= 2**31:
                raise ValueError("the node map holds 32-bit indices")
            n_c = int(node_idx.size)
            self._dev = _DeviceODE(ctx, model, S, n_c, 0, None, self.monitor)
            self._v_row = ctx.field(n, mesh.plane)  # the potential's home: a field of the PDE grid
            self._v_row.copy_from(self.v_ode.field)  # nodes outside every marker keep the potential they have
            self._real_pos = ctx.from_numpy(real_pos.astype(np.int64))       # entries of the array that are nodes ...
            self._real_nodes = ctx.from_numpy(node_idx[real_pos].astype(np.int64))  # ... and the nodes they are
            self._node_idx = self._real_nodes
            self._dev.node_map = (ctx.from_numpy(node_idx.astype(np.int32)), self._v_row)
        else:
            self._dev = _DeviceODE(ctx, model, S, n, mesh.plane, None, self.monitor)
            self._v_row = self._dev.states.row_field(vi)
            self._v_row.copy_from(self.v_ode.field)
            self._node_idx = None
            cls = cls_full
            pos = {m: np.nonzero(self._inds[m])[0] for m in ms}
        rows = self._dev.states.rows  # (S, n) view of the device array: filled there (tens of GB at full size, not a host array)
        self._idx_dev = {}
        for marker in ms:
            idx = ctx.from_numpy(pos[marker].astype(np.int64))
            self._idx_dev[marker] = idx
            init = np.asarray(self.init_states[marker], dtype=np.float64)
            if init.ndim == 1:  # (S,) broadcast to the marker's nodes (odesolver.py:149-153)
                for r in range(S):
                    rows[r].index_fill_(0, idx, float(init[r]))
            else:
                if init.shape != self.shape(marker):
                    raise ValueError(f"init_states[{marker}] has shape {init.shape}, expected {self.shape(marker)}")
                rows.index_copy_(1, idx, ctx.from_numpy(np.ascontiguousarray(init)))
        if compact:  # the marked nodes' initial potential goes to its home (the others keep what v_ode held)
            for marker in ms:
                nodes = ctx.from_numpy(np.nonzero(self._inds[marker])[0].astype(np.int64))
                self._v_row.data.index_copy_(0, nodes, rows[vi].index_select(0, self._idx_dev[marker]))
        self._cls_dev = ctx.from_numpy(cls)
        self._dev.explicit_classes = True
        self._dev.set_classes(self._cls_dev, [self.parameters[m] for m in ms])
        self._odes = {}

    def _sync_v(self):
        if self._pending_ops is not None:
            self._pending_ops.flush_pending()

    def _refresh_compact_v(self) -> None:
        """Compact layout: the potential row of the state array mirrors the field only as of the last ionic launch."""
        if self._marked and self._node_idx is not None:
            self._sync_v()
            self._dev.states.rows[self._vi].index_copy_(0, self._real_pos, self._v_row.data.index_select(0, self._real_nodes))

    def _release_aliases(self):
        self._sync_v()
        for f in self._aliases:
            if f._alias is self._v_row:
                f.materialize()
        self._aliases = []

    def _fused_prepare(self) -> int:
        """Called by the fused split step before the ionic launch: refresh the class table if a parameter set was
        edited; returns the potential's row."""
        self._dev.set_classes(self._cls_dev, [self.parameters[m] for m in self._marker_values])
        return self._vi

    def _masked_copy(self, dst, src) -> None:
        """dst[i] = src[i] on the nodes that carry a marker (all of them: a plain copy)."""
        if self._all_marked:
            dst.copy_from(src)
            return
        lib, n_m = self._ctx.lib, int(self._idx_all.numel())
        tmp = self._ctx.zeros(n_m)
        _hip.check(lib.beat_gather(self._ctx.handle, C.c_void_p(tmp.data_ptr()), src.ptr, C.c_void_p(self._idx_all.data_ptr()), n_m))
        _hip.check(lib.beat_scatter(self._ctx.handle, dst.ptr, C.c_void_p(tmp.data_ptr()), C.c_void_p(self._idx_all.data_ptr()), n_m))

    def _initialize_full_values(self):
        ns = tuple(self.num_states.values())
        self._all_states_equal_size = bool((np.array(ns) == ns[0]).all())
        if self._all_states_equal_size:
            self._full_values = np.zeros((ns[0], self.markers.x.array.size))

    def to_dolfin(self) -> None:
        if self._marked:
            if self.v_ode._alias is self._v_row:
                return
            self._sync_v()
            self._masked_copy(self.v_ode.writable_field(overwrite_all=False), self._v_row)
            self.v_ode._touch()
        elif self.on_device:
            dst = self.v_ode.writable_field(overwrite_all=False)
            for marker in self._marker_values:
                row = self._odes[marker].states.row_field(self.v_index[marker])
                _hip.check(self._ctx.lib.beat_scatter(self._ctx.handle, dst.ptr, row.ptr,
                                                      C.c_void_p(self._idx_dev[marker].data_ptr()), row.n))
            self.v_ode._touch()
        else:
            arr = np.asarray(self.v_ode.x.array).copy()
            for marker in self._marker_values:
                arr[self._inds[marker]] = self._values[marker][self.v_index[marker], :]
            self.v_ode.x.array[:] = arr

    def scatter_v(self, dst) -> None:
        """Potentials of every marker's states -> field ``dst`` (fused split step)."""
        if self._marked:
            self._sync_v()
            return self._masked_copy(dst, self._v_row)
        for marker in self._marker_values:
            row = self._odes[marker].states.row_field(self.v_index[marker])
            _hip.check(self._ctx.lib.beat_scatter(self._ctx.handle, dst.ptr, row.ptr,
                                                  C.c_void_p(self._idx_dev[marker].data_ptr()), row.n))

    def gather_v(self, src) -> None:
        """Field ``src`` -> the potential rows of every marker's states (fused split step)."""
        if self._marked:
            self._release_aliases()
            return self._masked_copy(self._v_row, src)
        for marker in self._marker_values:
            row = self._odes[marker].states.row_field(self.v_index[marker])
            _hip.check(self._ctx.lib.beat_gather(self._ctx.handle, row.ptr, src.ptr,
                                                 C.c_void_p(self._idx_dev[marker].data_ptr()), row.n))

    def from_dolfin(self) -> None:
        if self._marked:
            if self.v_ode._alias is self._v_row:
                return
            self._release_aliases()
            self._masked_copy(self._v_row, self.v_ode.field)
        elif self.on_device:
            src = self.v_ode.field
            for marker in self._marker_values:
                row = self._odes[marker].states.row_field(self.v_index[marker])
                _hip.check(self._ctx.lib.beat_gather(self._ctx.handle, row.ptr, src.ptr,
                                                     C.c_void_p(self._idx_dev[marker].data_ptr()), row.n))
        else:
            arr = np.asarray(self.v_ode.x.array)
            for marker in self._marker_values:
                self._values[marker][self.v_index[marker], :] = arr[self._inds[marker]]

    def values(self, marker: int):
        if self._marked:  # the marker's columns of the one state array, (S, N_marker) as the reference keeps them
            self._sync_v()
            self._refresh_compact_v()
            return self._dev.states.rows.index_select(1, self._idx_dev[marker]).cpu().numpy()
        return self._odes[marker].states.numpy() if self.on_device else self._values[marker]

    def set_values(self, marker: int, values) -> None:
        """Overwrite the (S, N_marker) states of one marker (e.g. with a pre-paced steady state)."""
        values = np.broadcast_to(np.asarray(values, dtype=np.float64).reshape(self.num_states[marker], -1), self.shape(marker))
        if self._marked:
            self._release_aliases()
            self._dev.states.rows.index_copy_(1, self._idx_dev[marker], self._ctx.from_numpy(np.ascontiguousarray(values)))
            if self._node_idx is not None:  # the potential's home is the field
                nodes = self._ctx.from_numpy(np.nonzero(self._inds[marker])[0].astype(np.int64))
                self._v_row.data.index_copy_(0, nodes, self._ctx.from_numpy(np.ascontiguousarray(values[self._vi])))
        elif self.on_device:
            self._odes[marker].states.set(np.ascontiguousarray(values))
        else:
            self._values[marker][:] = values

    def num_parameters(self, marker: int) -> int:
        return len(self.parameters[marker])

    def shape(self, marker: int) -> tuple[int, int]:
        return (self.num_states[marker], self._num_points[marker])

    def num_points(self, marker: int) -> int:
        return self._num_points[marker]

    def step(self, t0: float, dt: float):
        if self._marked:
            with self.monitor.track_time("total_ode_step"):
                self._release_aliases()
                self._dev.step(t0, dt, v_index=self._fused_prepare())
            return
        with self.monitor.track_time("total_ode_step"):
            for marker, ode in self._odes.items():
                with self.monitor.track_time(f"marker_{marker}_ode_step"):
                    if self.on_device:
                        ode.parameters = self.parameters[marker]
                        ode.step(t0, dt)
                    else:
                        ode.step(t0=t0, dt=dt)

    def assign_all_states(self, functions: list[grid.Function]) -> None:
        num_states = self.num_states[self._marker_values[0]]
        assert len(functions) == num_states, "Number of functions must match number of states"
        for index, f in enumerate(functions):
            arr = np.asarray(f.x.array).copy()
            for marker in self._marker_values:
                arr[self._inds[marker]] = self.values(marker)[index, :]
            f.x.array[:] = arr

    def states_to_dolfin(self, names: list[str] | None = None) -> list[grid.Function]:
        V = self.v_ode.function_space
        num_states = self.num_states[self._marker_values[0]]
        if names is not None:
            assert len(names) == num_states, "Number of names must match number of states"
        else:
            names = [f"state_{i}" for i in range(num_states)]
        functions = [grid.Function(V, name=name) for name in names]
        self.assign_all_states(functions)
        return functions

    @property
    def full_values(self):
        if not self._all_states_equal_size:
            msg = ("Cannot get full values size states are not of equal size. "
                   f"Have {self.num_states=}, use .values(marker) instead")
            raise RuntimeError(msg)
        for marker in self._marker_values:
            self._full_values[:, self._inds[marker]] = self.values(marker)
        return self._full_values
