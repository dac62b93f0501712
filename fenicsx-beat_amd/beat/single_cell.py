"""Single-cell pre-pacing to a limit cycle -- interface of src/beat/single_cell.py:68-156.

With a built-in device model the ``nbeats x arange(0, BCL, dt)`` loop runs inside ONE kernel launch
(``beat_ode_run``: the cell's states stay in registers); any other callable is stepped on the host exactly
as the reference does it.  Results are cached in ``outdir`` under a hash of the inputs like the reference."""

from __future__ import annotations

import hashlib
import logging
from pathlib import Path
from typing import Callable

import numpy as np

from .models._base import DeviceModel

logger = logging.getLogger(__name__)


def compute_hash(fun: Callable, init_states: np.ndarray, parameters: np.ndarray, nbeats: int = 200,
                 BCL: float = 1000.0, dt: float = 0.05):
    hash_input = hashlib.md5()
    if isinstance(fun, DeviceModel):
        hash_input.update(f"{fun.name}:{fun.model_id}".encode())
    else:
        hash_input.update(fun.__code__.co_code)
    hash_input.update(str(init_states).encode())
    hash_input.update(str(parameters).encode())
    hash_input.update(str(nbeats).encode())
    hash_input.update(str(BCL).encode())
    hash_input.update(str(dt).encode())
    return hash_input.hexdigest()


def solve_without_save(fun, nbeats, times, y, p, dt):
    for _ in range(nbeats):
        for t in times:
            y[:] = fun(states=y, t=t, parameters=p, dt=dt)
    return y


def solve_with_save(fun, nbeats, times, y, p, dt, save_freq, track_values, track_indices):
    k = 0
    for _ in range(nbeats):
        j = 0
        for t in times:
            if j % save_freq == 0:
                for i, index in enumerate(track_indices):
                    track_values[k, i] = y[index]
                k += 1
            y[:] = fun(states=y, t=t, parameters=p, dt=dt)
            j += 1
    return y, track_values


def get_steady_state(fun: Callable, init_states: np.ndarray, parameters: np.ndarray, outdir: Path, nbeats: int = 200,
                     BCL: int = 1000, save_every_ms: float = 1.0, dt: float = 0.05,
                     track_indices: list[int] | None = None):
    outdir = Path(outdir)
    hash_input = compute_hash(fun=fun, init_states=init_states, parameters=parameters, nbeats=nbeats, BCL=BCL, dt=dt)
    fname = outdir / f"steady_states_{hash_input}.npy"
    if fname.is_file():
        return np.load(fname)
    outdir.mkdir(exist_ok=True, parents=True)
    logger.info(f"Computing steady state with {nbeats} beats.")
    times = np.arange(0.0, BCL, dt)
    y = np.array(init_states, dtype=np.float64)
    save_freq = int(np.ceil(save_every_ms / dt))
    if isinstance(fun, DeviceModel):
        y, track_values = fun.run(y, parameters, dt, nsteps=len(times), nbeats=nbeats, t0=0.0,
                                  track_indices=track_indices, save_freq=save_freq)
    elif track_indices is not None:
        M = int(np.ceil(len(times) / save_freq) * nbeats)
        track_values = np.zeros((M, len(track_indices)))
        y, track_values = solve_with_save(fun, nbeats, times, y, parameters, dt, save_freq, track_values,
                                          np.array(track_indices).astype(np.int32))
    else:
        y, track_values = solve_without_save(fun, nbeats, times, y, parameters, dt), None
    if track_values is not None:
        np.save(outdir / f"tracked_values_{hash_input}.npy", track_values)
    np.save(fname, y)
    return y
