"""Built-in cell models: handles the HIP backend recognises.

A :class:`DeviceModel` is what the reference calls ``fun`` (src/beat/odesolver.py:70-76): it is
still callable by keyword on NumPy arrays, ``fun(states=, t=, parameters=, dt=)`` returning the new
``(S, N)`` array -- the call stages the arrays through the GPU and runs the same HIP kernel as the
solver (there is no CPU implementation in the product) -- and it exposes the helper functions a
gotranx-generated module has (``init_state_values``, ``init_parameter_values``, ``state_index``,
``parameter_index``; demos/niederer_benchmark.py:66,99,212).
"""

from __future__ import annotations

import ctypes as C

import numpy as np


class DeviceModel:
    def __init__(self, name, model_id, states: dict, parameters: dict, v_name=None):
        self.name = name
        self.model_id = int(model_id)
        self.state_names = tuple(states)
        self.state_defaults = dict(states)
        self.parameter_names = tuple(parameters)
        self.parameter_defaults = dict(parameters)
        self.v_name = v_name
        self.__name__ = name

    # ---- gotranx-module-like helpers -----------------------------------------------------
    @property
    def num_states(self) -> int:
        return len(self.state_names)

    @property
    def num_parameters(self) -> int:
        return len(self.parameter_names)

    def state_index(self, name: str) -> int:
        try:
            return self.state_names.index(name)
        except ValueError:
            raise KeyError(f"Unknown state {name}") from None

    def parameter_index(self, name: str) -> int:
        try:
            return self.parameter_names.index(name)
        except ValueError:
            raise KeyError(f"Unknown parameter {name}") from None

    def init_state_values(self, **values) -> np.ndarray:
        d = dict(self.state_defaults)
        for k, v in values.items():
            if k not in d:
                raise KeyError(f"Unknown state {k}")
            d[k] = v
        return np.array([d[k] for k in self.state_names], dtype=np.float64)

    def init_parameter_values(self, **values) -> np.ndarray:
        d = dict(self.parameter_defaults)
        for k, v in values.items():
            if k not in d:
                raise KeyError(f"Unknown parameter {k}")
            d[k] = v
        return np.array([d[k] for k in self.parameter_names], dtype=np.float64)

    # ---- the reference's `fun` calling convention ---------------------------------------------
    def __call__(self, states=None, t=0.0, parameters=None, dt=None, **kwargs):
        """Advance ``states`` ((S,) or (S, N) NumPy) by one step on the GPU; returns a new array."""
        from .. import _hip
        from .._device import Context, StateArray

        if dt is None:
            raise TypeError("dt is required")
        ctx = Context.default()
        arr = np.asarray(states, dtype=np.float64)
        one_d = arr.ndim == 1
        a2 = arr.reshape(arr.shape[0], -1)
        S, n = a2.shape
        if S != self.num_states:
            raise ValueError(f"{self.name} has {self.num_states} states, got {S}")
        sa = StateArray(ctx, S, n)
        sa.set(a2)
        hp, ppn, pld = host_and_device_parameters(ctx, parameters, self.num_parameters, n)
        _hip.check(
            ctx.lib.beat_ode_step(ctx.handle, self.model_id, sa.ptr, n, sa.ld,
                                  None if hp is None else hp.ctypes.data_as(C.c_void_p), self.num_parameters if (hp is not None or ppn is not None) else 0,
                                  None if ppn is None else C.c_void_p(ppn.data_ptr()), pld, float(t), float(dt), 0, None)
        )
        out = sa.numpy()
        return out[:, 0].copy() if one_d else out


    # ---- many steps in one launch ---------------------------------------------------------------------
    def run(self, states, parameters, dt, nsteps, nbeats=1, t0=0.0, track_indices=None, save_freq=1):
        """Advance ``states`` ((S,) or (S, N)) by nbeats x nsteps steps inside one kernel launch (t restarts at
        t0 each beat, t = t0 + j*dt within it).  Returns (new_states, track) where track has shape
        (rows, len(track_indices)[, N]) or is None."""
        from .. import _hip
        from .._device import Context, StateArray

        ctx = Context.default()
        arr = np.asarray(states, dtype=np.float64)
        one_d = arr.ndim == 1
        a2 = arr.reshape(arr.shape[0], -1)
        S, n = a2.shape
        if S != self.num_states:
            raise ValueError(f"{self.name} has {self.num_states} states, got {S}")
        sa = StateArray(ctx, S, n)
        sa.set(a2)
        hp, ppn, pld = host_and_device_parameters(ctx, parameters, self.num_parameters, n)
        ntrack = 0 if track_indices is None else len(track_indices)
        trace, tidx = None, None
        rows = 0
        if ntrack:
            rows = int(nbeats) * int(-(-int(nsteps) // int(save_freq)))
            trace = ctx.zeros(rows * ntrack * n)
            tidx = (C.c_int * ntrack)(*[int(i) for i in track_indices])
        _hip.check(
            ctx.lib.beat_ode_run(ctx.handle, self.model_id, sa.ptr, n, sa.ld,
                                 None if hp is None else hp.ctypes.data_as(C.c_void_p),
                                 self.num_parameters if (hp is not None or ppn is not None) else 0,
                                 None if ppn is None else C.c_void_p(ppn.data_ptr()), pld, float(t0), float(dt), int(nsteps),
                                 int(nbeats), int(save_freq), tidx, ntrack,
                                 None if trace is None else C.c_void_p(trace.data_ptr()))
        )
        out = sa.numpy()
        tr = None
        if ntrack:
            tr = trace.cpu().numpy().reshape(rows, ntrack, n)
            if one_d:
                tr = tr[:, :, 0]
        return (out[:, 0].copy() if one_d else out), tr


class DeviceParameters:
    """Per-node parameters ``(P, N)`` (demos/pace_train.py:133-167) kept resident in HBM.

    A plain NumPy ``(P, N)`` array handed to an ODE solver is compared with the copy that was uploaded before every
    step (the reference passes the live array to ``fun`` each step, so in-place edits must be seen: exact, but one
    O(P N) host pass per step).  This handle removes that pass: the values live on the device, ``set`` / ``set_row``
    replace them and bump ``version``; nothing is checked or moved per step."""

    ndim = 2

    def __init__(self, values, ctx=None):
        from .._device import Context

        self.ctx = ctx or Context.default()
        self.version = 0
        self._dev = None
        self.set(values)

    @property
    def shape(self):
        return tuple(self._dev.shape)

    def __len__(self) -> int:
        return int(self._dev.shape[0])

    @property
    def tensor(self):
        return self._dev

    def set(self, values) -> None:
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim != 2:
            raise ValueError(f"per-node parameters must be (P, N), got shape {v.shape}")
        if self._dev is not None and tuple(self._dev.shape) == v.shape:
            self._dev.copy_(self.ctx.torch.from_numpy(v))
        else:
            self._dev = self.ctx.from_numpy(v).reshape(v.shape)
        self.version += 1

    def set_row(self, k: int, values) -> None:
        """One parameter at every node (a scalar is broadcast)."""
        row = np.broadcast_to(np.asarray(values, dtype=np.float64), (self._dev.shape[1],))
        self._dev[int(k)].copy_(self.ctx.torch.from_numpy(np.array(row, dtype=np.float64)))
        self.version += 1

    def numpy(self) -> np.ndarray:
        return self._dev.cpu().numpy()


def host_and_device_parameters(ctx, parameters, num_parameters, n):
    """Split ``parameters`` into (host (P,) array | None, device (P, N) tensor | None, ld)."""
    if parameters is None:
        return None, None, 0
    if isinstance(parameters, DeviceParameters):
        if parameters.shape != (num_parameters, n):
            raise ValueError(f"per-node parameters must have shape ({num_parameters}, {n}), got {parameters.shape}")
        return None, parameters.tensor, n
    p = np.asarray(parameters, dtype=np.float64)
    if p.ndim == 1:
        if p.shape[0] != num_parameters:
            raise ValueError(f"expected {num_parameters} parameters, got {p.shape[0]}")
        return np.ascontiguousarray(p), None, 0
    if p.shape != (num_parameters, n):
        raise ValueError(f"per-node parameters must have shape ({num_parameters}, {n}), got {p.shape}")
    return None, ctx.from_numpy(np.ascontiguousarray(p)), n
