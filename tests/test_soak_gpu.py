"""GPU: seeded, budgeted soak of the ionic kernels (tools/soak_cells.py and tools/soak_vs_oracle.py as tests; see tests/_soak.py).

* 512 cells per model, every parameter +-10 %, cell types cycled, three paced beats in the in-kernel time loop
  (beat_ode_run, what beat.single_cell.get_steady_state drives; src/beat/single_cell.py:86-156) at dt = 0.05 ms: all states
  finite, gates and occupancies in [0, 1], concentrations positive, every cell fired;
* 64 of them (the seven the pre-fix ToR-ORd kernel lost among them) against the NumPy oracle stepped on the host, from the
  device's own states at four phases of the third beat (stimulus, plateau, repolarisation, rest), 150 steps each: <= 1e-8
  relative to the state scale."""
import numpy as np
import pytest

import _soak

pytestmark = pytest.mark.gpu


def _model(name):
    from beat.models import torord, torord_land, tp06
    from oracle import ionic
    from oracle import torord as otor

    return {"tp06": (tp06, ionic.tp06_generalized_rush_larsen), "torord": (torord, otor.torord_generalized_rush_larsen),
            "torord_land": (torord_land, otor.torord_land_generalized_rush_larsen)}[name]


@pytest.mark.parametrize("name", ["tp06", "torord", "torord_land"])
def test_perturbed_cells_survive_three_paced_beats_and_follow_the_oracle(hip_ctx, name):
    m, oracle_step = _model(name)
    dt, per_beat, n = 0.05, 20000, 512
    P = _soak.population(name, m.init_parameter_values(), m.parameter_index, n)
    y0 = np.repeat(m.init_state_values()[:, None], n, axis=1)
    vi = m.state_index(_soak.SPEC[name]["v"])
    run = m.generalized_rush_larsen.run
    y2, tr = run(y0, P, dt=dt, nsteps=per_beat, nbeats=2, track_indices=[vi], save_freq=20)
    _soak.check_physical(name, y2, m.state_index, " after two beats")
    assert (tr[:, 0].max(axis=0) > 0.0).all(), "every cell fires"
    cells = _soak.subset(n, 64)
    Pc, scale = P[:, cells], m.init_state_values()
    worst = 0.0
    y, t_done = y2, 0
    for phase in (0, 4000, 8000, 16000):  # steps into the third beat at which a comparison window starts
        if phase > t_done:
            y, _ = run(y, P, dt=dt, nsteps=phase - t_done, nbeats=1, t0=t_done * dt)
            t_done = phase
        _soak.check_physical(name, y, m.state_index, f" at {phase * dt:.0f} ms of beat three")
        yd, _ = run(y[:, cells], Pc, dt=dt, nsteps=150, nbeats=1, t0=phase * dt)
        yo = y[:, cells].copy()
        for j in range(150):
            yo = oracle_step(yo, phase * dt + j * dt, dt, Pc)
        assert np.isfinite(yo).all()
        worst = max(worst, float(_soak.relative_difference(yd, yo, scale).max()))
    y3, _ = run(y, P, dt=dt, nsteps=per_beat - t_done, nbeats=1, t0=t_done * dt)
    _soak.check_physical(name, y3, m.state_index, " after three beats")
    assert worst < 1e-8, worst
