"""Populations and checks shared by the soak tests (tests/test_soak_gpu.py on the device, tests/test_soak_host.py on the host build
of the same kernel sources): cells with EVERY parameter scaled by 0.9 .. 1.1, cell types cycled, paced for several beats.

Why it is in the suite: the one real bug of round 3 -- a rewrite of the ToR-ORd release time constant that assumed a positive
SR load -- passed every parity test and blew up 7 of 512 such cells in their third paced beat (tools/soak_cells.py found it;
`git show 6a58722^:fenicsx-beat_amd/csrc/torord_dyncl.h` fails these tests: cells 14, 20, 29, 128, 326, 353, 479 of the
population below turn non-finite).  The reference paces single cells the same way before every realistic run
(src/beat/single_cell.py:86-156, demos/biv_endocardial.py:124-173)."""
import numpy as np

# GRL1 advances every state on its own: the occupancies of the IKr Markov model are not conserved exactly and may pass 1 by a
# few 1e-3 (C3 at rest, perturbed rate constants) -- in the specification's scheme, not only here
TOL = 5e-3
SENSITIVE = (14, 20, 29, 128, 326, 353, 479)  # of the 512-cell ToR-ORd population: the ones the pre-fix kernel lost

SPEC = {
    "tp06": dict(v="V", conc=("Ca_i", "Ca_SR", "Ca_ss", "Na_i", "K_i"),
                 gates=("Xr1", "Xr2", "Xs", "m", "h", "j", "d", "f", "f2", "fCass", "s", "r", "R_prime"),
                 fixed=("stim_start", "stim_period", "stim_duration")),
    "torord": dict(v="v", conc=("cai", "cajsr", "cansr", "cass", "cli", "clss", "ki", "kss", "nai", "nass"),
                   gates=("C1", "C2", "C3", "I_", "O_", "a", "ap", "iF", "iFp", "iS", "iSp", "d", "fcaf", "fcafp", "fcas", "ff_", "ffp",
                          "fs", "jca", "nca_i", "nca_ss", "h", "hp", "j", "jp", "m", "hL", "hLp", "mL", "xs1", "xs2"),
                   fixed=("i_Stim_Start", "i_Stim_End", "i_Stim_Period", "i_Stim_PulseDuration")),
    "torord_land": dict(v="v", conc=("cai", "cajsr", "cansr", "cass", "cli", "clss", "ki", "kss", "nai", "nass"),
                        gates=("d", "m", "h", "j", "XS", "XW", "TmB"),
                        fixed=("i_Stim_Start", "i_Stim_End", "i_Stim_Period", "i_Stim_PulseDuration", "mode", "isacs")),
}


def population(name, P0, parameter_index, n=512, seed=2):
    """(P, n) parameters of tools/soak_cells.py's population: all scaled per cell, cell types cycled, protocol untouched."""
    rng = np.random.default_rng(seed)
    P = np.repeat(np.asarray(P0, dtype=np.float64)[:, None], n, axis=1) * rng.uniform(0.9, 1.1, (len(P0), n))
    if name != "tp06":
        P[parameter_index("celltype")] = np.arange(n) % 3
    for k in SPEC[name]["fixed"]:
        P[parameter_index(k)] = P0[parameter_index(k)]
    return P


def subset(n_total, count):
    """``count`` cells of the population: the sensitive ones first, then the lowest indices (all three cell types)."""
    rest = [i for i in range(n_total) if i not in SENSITIVE]
    return np.array(sorted(list(SENSITIVE) + rest[: max(0, count - len(SENSITIVE))]))


def check_physical(name, y, state_index, label=""):
    """every state finite, gates and occupancies in [0, 1] (+- TOL), concentrations positive"""
    assert np.isfinite(y).all(), f"{name}{label}: non-finite states in cells {np.nonzero(~np.isfinite(y).all(axis=0))[0][:20]}"
    for k in SPEC[name]["gates"]:
        row = y[state_index(k)]
        assert row.min() > -TOL and row.max() < 1.0 + TOL, f"{name}{label}: {k} in [{row.min():.6g}, {row.max():.6g}]"
    for k in SPEC[name]["conc"]:
        assert (y[state_index(k)] > 0.0).all(), f"{name}{label}: {k} min {y[state_index(k)].min():.6g}"


def relative_difference(y, ref, y_scale):
    return np.abs(y - ref) / np.maximum(np.abs(ref), 1e-6 * np.abs(y_scale)[:, None] + 1e-12)
