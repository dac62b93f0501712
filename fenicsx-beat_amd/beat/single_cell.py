"""Single-cell pre-pacing to a limit cycle -- the interface of src/beat/single_cell.py:68-156
(``get_steady_state(fun, init_states, parameters, outdir, nbeats, BCL, save_every_ms, dt, track_indices)``, results
cached in ``outdir``; the ventricular demos call it once per cell type, demos/biv_endocardial.py:152-181).

The pacing protocol is ``nbeats`` repetitions of the step times ``arange(0, BCL, dt)`` (time restarts every beat, so the
model's own periodic stimulus fires once per beat); every ``ceil(save_every_ms / dt)``-th step of a beat the tracked
states are recorded BEFORE the step.  With a built-in device model the whole protocol is one kernel launch
(``beat_ode_run``: the cell's states stay in registers, csrc/beat_ode.hip); any other callable is paced on the host by
``pace_on_host`` through the ``fun(states=, t=, parameters=, dt=)`` convention of the ODE solvers."""

from __future__ import annotations

import hashlib
import logging
from pathlib import Path
from typing import Callable

import numpy as np

from .models._base import DeviceModel

logger = logging.getLogger(__name__)


def _identity_of(fun: Callable) -> bytes:
    """What distinguishes one step function from another in the cache key."""
    if isinstance(fun, DeviceModel):
        if hasattr(fun, "cxx_name"):  # a model generated from an .ode file: its id is per process, its source digest is not
            return f"device:{fun.name}:{fun.cxx_name}".encode()
        return f"device:{fun.name}:{fun.model_id}".encode()
    code = getattr(fun, "__code__", None)
    return code.co_code if code is not None else repr(fun).encode()


def compute_hash(fun: Callable, init_states: np.ndarray, parameters: np.ndarray, nbeats: int = 200,
                 BCL: float = 1000.0, dt: float = 0.05) -> str:
    """Cache key of one pacing run: the step function, the exact bytes of the initial states and parameters, and the
    protocol."""
    key = hashlib.md5(_identity_of(fun))
    for arr in (init_states, parameters):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        key.update(str(a.shape).encode())
        key.update(a.tobytes())
    key.update(repr((int(nbeats), float(BCL), float(dt))).encode())
    return key.hexdigest()


def pace_on_host(fun: Callable, states: np.ndarray, parameters: np.ndarray, dt: float, nsteps: int, nbeats: int,
                 save_freq: int = 1, track_indices=None):
    """The pacing protocol for an arbitrary step function.  Returns (final states, tracked values or None); the tracked
    array has one row per recorded step, ``nbeats * ceil(nsteps / save_freq)`` in all."""
    y = np.array(states, dtype=np.float64)
    tracked = None
    if track_indices is not None:
        cols = np.asarray(track_indices, dtype=np.intp)
        tracked = np.zeros((int(nbeats) * -(-int(nsteps) // int(save_freq)), len(cols)))
    row = 0
    for _beat in range(int(nbeats)):
        for j in range(int(nsteps)):
            if tracked is not None and j % save_freq == 0:
                tracked[row] = y[cols]
                row += 1
            y[...] = fun(states=y, t=j * dt, parameters=parameters, dt=dt)
    return y, tracked


def _pace_over_times(fun, nbeats, times, y, p, dt, save_freq=1, track_values=None, track_indices=None):
    """The pacing loop over explicit step times (the two reference entry points below share it)."""
    row = 0
    for _beat in range(int(nbeats)):
        for j, t in enumerate(times):
            if track_values is not None and j % save_freq == 0:
                for i, index in enumerate(track_indices):
                    track_values[row, i] = y[index]
                row += 1
            y[:] = fun(states=y, t=t, parameters=p, dt=dt)
    return y


def solve_with_save(fun, nbeats, times, y, p, dt, save_freq, track_values, track_indices):
    """Reference signature (src/beat/single_cell.py:40-57): ``nbeats`` passes over ``times``, ``y`` advanced in place,
    the tracked states written to ``track_values`` every ``save_freq`` steps.  Returns (y, track_values)."""
    return _pace_over_times(fun, nbeats, times, y, p, dt, save_freq, track_values, track_indices), track_values


def solve_without_save(fun, nbeats, times, y, p, dt):
    """Reference signature (src/beat/single_cell.py:60-65)."""
    return _pace_over_times(fun, nbeats, times, y, p, dt)


def get_steady_state(fun: Callable, init_states: np.ndarray, parameters: np.ndarray, outdir: Path, nbeats: int = 200,
                     BCL: int = 1000, save_every_ms: float = 1.0, dt: float = 0.05,
                     track_indices: list[int] | None = None):
    outdir = Path(outdir)
    key = compute_hash(fun=fun, init_states=init_states, parameters=parameters, nbeats=nbeats, BCL=BCL, dt=dt)
    cached = outdir / f"steady_states_{key}.npy"
    if cached.is_file():
        return np.load(cached)
    outdir.mkdir(exist_ok=True, parents=True)
    logger.info(f"Computing steady state with {nbeats} beats.")
    nsteps = len(np.arange(0.0, BCL, dt))  # the reference's step times; t = j * dt within a beat
    save_freq = int(np.ceil(save_every_ms / dt))
    if isinstance(fun, DeviceModel):
        y, tracked = fun.run(np.array(init_states, dtype=np.float64), parameters, dt, nsteps=nsteps, nbeats=nbeats, t0=0.0,
                             track_indices=track_indices, save_freq=save_freq)
    else:
        y, tracked = pace_on_host(fun, init_states, parameters, dt, nsteps, nbeats, save_freq, track_indices)
    if tracked is not None:
        np.save(outdir / f"tracked_values_{key}.npy", tracked)
    np.save(cached, y)
    return y
