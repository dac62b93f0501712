"""CPU: the oracle's C restatement (oracle/beat_oracle.c, used for the CPU baseline) against the NumPy
oracle that is pinned to the reference."""

import numpy as np
import pytest

from oracle import fem, ionic

cport = pytest.importorskip("oracle.cport")


@pytest.fixture(scope="module", autouse=True)
def _built():
    try:
        cport.load()
    except FileNotFoundError:
        import subprocess
        from pathlib import Path

        subprocess.run(["make", "-C", str(Path(__file__).resolve().parents[1] / "oracle")], check=True)
        cport.load()


def test_c_tp06_step_matches_numpy_oracle():
    rng = np.random.default_rng(3)
    n = 4000
    S = np.repeat(ionic.tp06_init_state_values()[:, None], n, axis=1)
    S[17] = rng.uniform(-95, 50, n)
    for g in range(13):
        S[g] = rng.uniform(0, 1, n)
    S[13] = 10 ** rng.uniform(-4.2, -2.8, n)
    S[15] = 10 ** rng.uniform(-4, -2, n)
    S[14] = rng.uniform(1, 4.5, n)
    S[16] = rng.uniform(6, 12, n)
    S[18] = rng.uniform(125, 145, n)
    for t, P in ((1.0, ionic.tp06_init_parameter_values(stim_amplitude=0.0)), (10.3, ionic.tp06_init_parameter_values())):
        ref = ionic.tp06_generalized_rush_larsen(S, t, 0.02, P)
        out = np.ascontiguousarray(S.copy())
        cport.tp06_grl1(out, t, 0.02, P)
        err = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)
        near = np.abs(S[17] - 15.0) < 0.05
        assert err[:, ~near].max() < 1e-11, err[:, ~near].max()


def test_c_theta_step_matches_sparse_lu():
    cells, L = (14, 9, 7), (1.4, 0.9, 0.7)
    mesh = fem.BoxMesh(cells, L)
    f0 = np.array([np.cos(np.pi / 6), np.sin(np.pi / 6), 0.0])
    M = 9.5e-4 * np.outer(f0, f0) + 1.25e-4 * (np.eye(3) - np.outer(f0, f0))
    C_m, theta, dt = 0.01, 0.5, 0.05
    mt, kt = fem.stencil_table(3, tuple(l / c for l, c in zip(L, cells)), M, 1.0, 1.0)
    A, B = C_m * mt + theta * dt * kt, C_m * mt - (1 - theta) * dt * kt
    rng = np.random.default_rng(0)
    v0 = -85.0 + 30 * rng.random(mesh.num_nodes)
    w = fem.stimulus_weights(mesh, mesh.locate_cells(lambda x: x[0] <= 0.5))
    model = fem.OracleMonodomainModel(mesh, M, [fem.OracleStimulus(lambda t: 0.3, w)], C_m=C_m, theta=theta, default_timestep=dt)
    model.state[:] = v0
    model.assign_previous()
    model.step((0.0, dt))
    v = v0.copy()
    its = cport.theta_step(A, B, mesh.shape_nodes, v, w, 0.3 * dt, 1e-12)
    assert 0 < its < 100
    assert np.abs(v - model.state).max() <= 1e-9 * np.abs(model.state).max()
    x = rng.standard_normal(mesh.num_nodes)
    np.testing.assert_allclose(cport.stencil_apply(mt, mesh.shape_nodes, x), fem.assemble_mass(mesh) @ x, atol=1e-14)
