#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run in the BUILD CONTAINER only).

Two sources, both the reference itself:

1. ``tp06_spec.npz`` / ``torord_spec.npz`` / ``torord_land_spec.npz`` -- the reference's TP06 / ToR-ORd-dynCl
   (+ Land) ``.ode`` specifications (odes/tentusscher_panfilov_2006/tentusscher_panfilov_2006_epi_cell.ode,
   odes/torord/ToRORd_dynCl_endo.ode, odes/torord/ToRORd_dynCl_endo_Land.ode) evaluated by the
   independent evaluator in ``ode_spec.py``: right-hand sides, total self-derivatives and one
   GRL1 step at seeded random states.
2. ``splitting_reference.npz`` / ``.json`` -- the reference's OWN ``src/beat/odesolver.py`` and
   ``src/beat/monodomain_solver.py`` loaded against inert stub modules for the third-party
   packages that are not installed here (dolfinx, ufl, petsc4py, mpi4py, basix, pint): state
   layouts, broadcast of initial states, the ODE<->PDE data-movement sequence, per-marker
   scatter/gather and the exact order in which ``MonodomainSplittingSolver.step`` calls its
   collaborators for theta = 1 and theta = 0.5.

The fixtures are data (inputs + expected outputs); no reference source text is stored.
"""

from __future__ import annotations

import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
sys.path.insert(0, str(HERE))


# ------------------------------------------------------------------------------------------------
def make_tp06_spec():
    from ode_spec import OdeSpec

    spec = OdeSpec(REF / "odes/tentusscher_panfilov_2006/tentusscher_panfilov_2006_epi_cell.ode")
    rng = np.random.default_rng(20261003)
    n = 96
    st = {k: np.full(n, v) for k, v in spec.states.items()}
    st["V"] = rng.uniform(-95, 50, n)
    st["V"][:4] = [-85.23, -40.0 - 1e-9, -40.0, 14.0]
    for g in ["Xr1", "Xr2", "Xs", "m", "h", "j", "d", "f", "f2", "fCass", "s", "r", "R_prime"]:
        st[g] = rng.uniform(0, 1, n)
    st["Ca_i"] = 10 ** rng.uniform(-4.2, -2.8, n)
    st["Ca_ss"] = 10 ** rng.uniform(-4, -2, n)
    st["Ca_SR"] = rng.uniform(1, 4.5, n)
    st["Na_i"] = rng.uniform(6, 12, n)
    st["K_i"] = rng.uniform(125, 145, n)
    for k in spec.states:  # first column = the file's default initial state
        st[k][0] = spec.states[k]
    par = dict(spec.parameters)
    t, dt = 10.3, 0.05  # inside the model's own stimulus window (10 <= t <= 11)
    rhs, J, new = spec.grl1(st, par, t, dt, total=True)
    _, J_explicit, new_explicit = spec.grl1(st, par, t, dt, total=False)
    names = spec.state_names
    np.savez_compressed(
        HERE / "tp06_spec.npz",
        state_names=np.array(names),
        parameter_names=np.array(spec.parameter_names),
        state_defaults=np.array([spec.states[k] for k in names]),
        parameter_defaults=np.array([spec.parameters[k] for k in spec.parameter_names]),
        states=np.array([st[k] for k in names]),
        t=t,
        dt=dt,
        rhs=np.array([rhs[k] for k in names]),
        jac_total=np.array([J[k] for k in names]),
        grl1_total=np.array([new[k] for k in names]),
        explicit_jac_is_zero=np.array([J_explicit[k] is None for k in names]),
        grl1_explicit=np.array([new_explicit[k] for k in names]),
    )
    print("tp06_spec.npz:", n, "points,", len(names), "states")


def make_torord_spec():
    """ToR-ORd-dynCl (odes/torord/ToRORd_dynCl_endo.ode): RHS, total self-derivatives and one GRL1 step at
    seeded states around the file's initial state, for the three cell types."""
    from ode_spec import OdeSpec

    spec = OdeSpec(REF / "odes/torord/ToRORd_dynCl_endo.ode")
    rng = np.random.default_rng(20261004)
    n = 48
    names = spec.state_names
    st = {k: spec.states[k] * (1.0 + 0.05 * rng.uniform(-1, 1, n)) for k in names}
    st["v"] = rng.uniform(-90.0, 40.0, n)
    for k in names:
        st[k][0] = spec.states[k]
    t, dt = 0.3, 0.05
    out = dict(state_names=np.array(names), parameter_names=np.array(spec.parameter_names),
               state_defaults=np.array([spec.states[k] for k in names]),
               parameter_defaults=np.array([spec.parameters[k] for k in spec.parameter_names]),
               states=np.array([st[k] for k in names]), t=t, dt=dt)
    for celltype in (0, 1, 2):
        par = dict(spec.parameters)
        par["celltype"] = float(celltype)
        rhs, J, new = spec.grl1(st, par, t, dt, total=True)
        out[f"rhs_celltype{celltype}"] = np.array([rhs[k] for k in names])
        out[f"jac_celltype{celltype}"] = np.array([J[k] for k in names])
        out[f"grl1_celltype{celltype}"] = np.array([new[k] for k in names])
    # states along one paced action potential (endo, dt = 0.02 ms): covers the upstroke, plateau and
    # repolarisation regimes that the perturbed resting states above do not reach
    import sympy

    lin = spec.linearized(total=True)
    fns = {}
    for s in names:
        syms = sorted(lin[s].free_symbols, key=lambda x: x.name)
        fns[s] = (syms, sympy.lambdify(syms, lin[s], "numpy", cse=True))
    par = dict(spec.parameters)
    cur = {k: np.array([v]) for k, v in spec.states.items()}
    hdt, tt = 0.02, 0.0
    keep_at = sorted(set(list(range(0, 150, 5)) + list(range(150, 15000, 500))))
    traj, traj_t = [], []
    for i in range(15000):
        if i in keep_at:
            traj.append(np.array([cur[k][0] for k in names]))
            traj_t.append(tt)
        vals = spec.evaluate(cur, par, tt)
        env = dict(par)
        env.update(cur)
        env["time"] = tt
        new = {}
        for s in names:
            f = np.asarray(vals[f"d{s}_dt"], dtype=float)
            syms, fn = fns[s]
            Jv = np.asarray(fn(*[env[x.name] for x in syms]), dtype=float)
            with np.errstate(all="ignore"):
                new[s] = cur[s] + np.where(np.abs(Jv) > 1e-8, f * (np.exp(Jv * hdt) - 1) / Jv, f * hdt)
        cur, tt = new, tt + hdt
    traj = np.array(traj).T  # (45, npts)
    out["traj_states"] = traj
    out["traj_times"] = np.array(traj_t)
    out["traj_dt"] = hdt
    # one GRL1 step from every trajectory state, all evaluated at the SAME time t (stimulus off)
    stt = {k: traj[i] for i, k in enumerate(names)}
    _, _, new = spec.grl1(stt, par, 5.0, hdt, total=True)
    out["traj_grl1"] = np.array([new[k] for k in names])
    out["traj_step_t"] = 5.0
    np.savez_compressed(HERE / "torord_spec.npz", **out)
    print("torord_spec.npz:", n, "points,", len(names), "states, 3 cell types;", traj.shape[1], "trajectory states, V range",
          traj[names.index("v")].min(), traj[names.index("v")].max())


def make_torord_land_spec():
    """ToR-ORd-dynCl with the Land contraction model (odes/torord/ToRORd_dynCl_endo_Land.ode, 52 states, 140
    parameters): RHS, total self-derivatives and one GRL1 step at seeded states, for the three cell types at the
    file's parameters and for a stretched, lengthening cell (lmbda = 1.1 / 1.25 / 0.85, dLambda != 0) so that both
    branches of every condition of the mechanics part are taken; plus states along one paced action potential."""
    from ode_spec import OdeSpec

    import sympy

    spec = OdeSpec(REF / "odes/torord/ToRORd_dynCl_endo_Land.ode")
    rng = np.random.default_rng(20261005)
    n = 48
    names = spec.state_names
    st = {k: spec.states[k] * (1.0 + 0.05 * rng.uniform(-1, 1, n)) for k in names}
    st["v"] = rng.uniform(-90.0, 40.0, n)
    st["cai"] = 10 ** rng.uniform(-4.2, -2.9, n)
    st["XS"] = rng.uniform(0.0, 0.05, n)
    st["XW"] = rng.uniform(0.0, 0.05, n)
    st["CaTrpn"] = 10 ** rng.uniform(-4.0, -0.4, n)  # both sides of CaTrpn**(-ntm/2) = 100 (CaTrpn = 0.0215)
    st["TmB"] = rng.uniform(0.2, 1.0, n)
    st["Zetas"] = rng.uniform(-1.6, 0.6, n)          # the three branches of gammasu
    st["Zetaw"] = rng.uniform(-0.3, 0.3, n)
    st["Cd"] = rng.uniform(-0.2, 0.4, n)             # both signs of C - Cd
    for k in names:
        st[k][0] = spec.states[k]
    t, dt = 0.3, 0.05
    psets = []
    for celltype in (0.0, 1.0, 2.0):
        psets.append(dict(spec.parameters, celltype=celltype))
    psets.append(dict(spec.parameters, lmbda=1.1, dLambda=0.002))
    psets.append(dict(spec.parameters, lmbda=1.25, dLambda=-0.001, celltype=1.0, ntrpn=1.7, ntm=2.2))
    psets.append(dict(spec.parameters, lmbda=0.85, dLambda=0.0005, celltype=2.0))
    out = dict(state_names=np.array(names), parameter_names=np.array(spec.parameter_names),
               state_defaults=np.array([spec.states[k] for k in names]),
               parameter_defaults=np.array([spec.parameters[k] for k in spec.parameter_names]),
               states=np.array([st[k] for k in names]), t=t, dt=dt,
               parameter_sets=np.array([[ps[k] for k in spec.parameter_names] for ps in psets]))
    rhs_all, jac_all, new_all = [], [], []
    for ps in psets:
        rhs, J, new = spec.grl1(st, ps, t, dt, total=True)
        rhs_all.append([rhs[k] for k in names])
        jac_all.append([J[k] for k in names])
        new_all.append([new[k] for k in names])
    out["rhs"], out["jac"], out["grl1"] = np.array(rhs_all), np.array(jac_all), np.array(new_all)
    # one paced action potential (endo, dt = 0.02 ms) with the calcium transient driving troponin and the cross-bridges
    lin = spec.linearized(total=True)
    fns = {}
    for s in names:
        syms = sorted(lin[s].free_symbols, key=lambda x: x.name)
        fns[s] = (syms, sympy.lambdify(syms, lin[s], "numpy", cse=True))
    par = dict(spec.parameters)
    cur = {k: np.array([v]) for k, v in spec.states.items()}
    hdt, tt = 0.02, 0.0
    keep_at = sorted(set(list(range(0, 150, 5)) + list(range(150, 15000, 500))))
    traj = []
    for i in range(15000):
        if i in keep_at:
            traj.append(np.array([cur[k][0] for k in names]))
        vals = spec.evaluate(cur, par, tt)
        env = dict(par)
        env.update(cur)
        env["time"] = tt
        new = {}
        for s in names:
            f = np.asarray(vals[f"d{s}_dt"], dtype=float) + 0.0 * cur[s]
            syms, fn = fns[s]
            args = [np.broadcast_to(np.asarray(env[x.name], dtype=float), f.shape) for x in syms]
            with np.errstate(all="ignore"):
                Jv = np.asarray(fn(*args), dtype=float) + 0.0 * f
                new[s] = cur[s] + np.where(np.abs(Jv) > 1e-8, f * (np.exp(Jv * hdt) - 1) / Jv, f * hdt)
        cur, tt = new, tt + hdt
    traj = np.array(traj).T
    out["traj_states"] = traj
    out["traj_dt"] = hdt
    stt = {k: traj[i] for i, k in enumerate(names)}
    _, _, new = spec.grl1(stt, par, 5.0, hdt, total=True)
    out["traj_grl1"] = np.array([new[k] for k in names])
    out["traj_step_t"] = 5.0
    np.savez_compressed(HERE / "torord_land_spec.npz", **out)
    print("torord_land_spec.npz:", n, "points,", len(names), "states,", len(psets), "parameter sets;", traj.shape[1],
          "trajectory states, V range", traj[names.index("v")].min(), traj[names.index("v")].max(), "max CaTrpn",
          traj[names.index("CaTrpn")].max(), "max XS", traj[names.index("XS")].max())


# ------------------------------------------------------------------------------------------------
class _Absorb(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = _Absorb(f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return _Absorb(self.__name__ + "()")


def load_reference_modules():
    """Appendix A of SURVEY.md: inert stand-ins for the un-installed third-party packages, then the
    reference's own modules loaded by path."""
    def mod(name, **attrs):
        m = _Absorb(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Comm:
        rank, size = 0, 1

    mpi = mod("mpi4py")
    mpi.MPI = mod("mpi4py.MPI", COMM_WORLD=_Comm(), Intracomm=_Comm)
    petsc = mod("petsc4py")
    petsc.PETSc = mod("petsc4py.PETSc", KSP=object, Error=type("Error", (Exception,), {}))
    dolfinx = mod("dolfinx", __version__="0.10.0")
    dolfinx.fem = mod("dolfinx.fem")
    dolfinx.fem.petsc = mod("dolfinx.fem.petsc")
    dolfinx.mesh = mod("dolfinx.mesh")
    dolfinx.fem.Function = object
    mod("basix")
    ufl = mod("ufl", Measure=object, Coefficient=object, Form=object)
    ufl.core = mod("ufl.core")
    ufl.core.expr = mod("ufl.core.expr", Expr=object)
    mod("pint", UnitRegistry=lambda: _Absorb("ureg"), Quantity=object)

    beat = types.ModuleType("beat")
    beat.__path__ = [str(REF / "src/beat")]
    sys.modules["beat"] = beat
    out = {}
    for name in ("telemetry", "utils", "odesolver", "units", "stimulation", "base_model", "monodomain_model",
                 "monodomain_solver"):
        spec = importlib.util.spec_from_file_location(f"beat.{name}", REF / "src/beat" / f"{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"beat.{name}"] = m
        spec.loader.exec_module(m)
        setattr(beat, name, m)
        out[name] = m
    return out


class FakeVector:
    def __init__(self, n):
        self.array = np.zeros(n)


class FakeElement:
    family_name = "Lagrange"

    def degree(self):
        return 1


class FakeFunction:
    def __init__(self, n, space="V"):
        self.x = FakeVector(n)
        self.function_space = space

    def ufl_element(self):
        return FakeElement()


def simple_ode_forward_euler(states, t, dt, parameters):
    v, s = states
    a, b = (1.0, 1.0) if parameters is None else parameters
    values = np.zeros_like(states)
    values[0] = v - a * s * dt
    values[1] = s + b * v * dt
    return values


def make_splitting_reference():
    mods = load_reference_modules()
    odesolver, msolver = mods["odesolver"], mods["monodomain_solver"]
    arrays, meta = {}, {}

    # --- DolfinODESolver: broadcast + data movement (tests/test_odesolver.py:52-117) ----------------
    n = 7
    v_ode, v_pde = FakeFunction(n), FakeFunction(n)
    ode = odesolver.DolfinODESolver(v_ode=v_ode, v_pde=v_pde, init_states=np.array([1.0, 2.0]),
                                    parameters=np.array([1.5, 0.5]), fun=simple_ode_forward_euler,
                                    num_states=2, v_index=0)
    arrays["dolfin_values_init"] = ode.values.copy()
    ode.step(0.0, 0.1)
    arrays["dolfin_values_after_step"] = ode.values.copy()
    arrays["dolfin_v_ode_before_to_dolfin"] = v_ode.x.array.copy()
    ode.to_dolfin()
    arrays["dolfin_v_ode_after_to_dolfin"] = v_ode.x.array.copy()
    ode.ode_to_pde()
    arrays["dolfin_v_pde_after_ode_to_pde"] = v_pde.x.array.copy()
    v_pde.x.array[:] = np.linspace(-1.0, 1.0, n)
    ode.pde_to_ode()
    ode.from_dolfin()
    arrays["dolfin_values_after_from_dolfin"] = ode.values.copy()

    # --- free-running multi-point loop (odesolver.py:24-43) ------------------------------------------
    states = np.zeros((2, 3))
    states.T[:] = [1.0, 0.0]
    trace = np.zeros((12, 3))  # the loop runs while t + dt < t_bound with accumulated t

    def inplace(states, t, dt, parameters):
        states[:] = simple_ode_forward_euler(states, t, dt, parameters)

    odesolver.solve(fun=inplace, t_bound=1.0, states=states, V=trace, V_index=0, dt=0.1, parameters=np.array([1.0, 1.0]))
    arrays["solve_trace"] = trace
    arrays["solve_final_states"] = states.copy()

    # --- DolfinMultiODESolver (odesolver.py:228-354) -------------------------------------------------
    n = 10
    markers = FakeFunction(n)
    markers.x.array[:] = np.array([0, 1, 1, 0, 2, 2, 0, 1, 0, 2], dtype=float)
    v_ode, v_pde = FakeFunction(n), FakeFunction(n)
    multi = odesolver.DolfinMultiODESolver(
        v_ode=v_ode, v_pde=v_pde, markers=markers,
        init_states={0: np.array([1.0, 2.0]), 1: np.array([3.0, 4.0]), 2: np.array([5.0, 6.0])},
        parameters={0: np.array([1.0, 1.0]), 1: np.array([2.0, 0.5]), 2: np.array([0.25, 4.0])},
        fun={0: simple_ode_forward_euler, 1: simple_ode_forward_euler, 2: simple_ode_forward_euler},
        num_states={0: 2, 1: 2, 2: 2}, v_index={0: 0, 1: 0, 2: 0})
    arrays["multi_markers"] = markers.x.array.copy()
    multi.step(0.0, 0.1)
    multi.to_dolfin()
    arrays["multi_v_ode_after_to_dolfin"] = v_ode.x.array.copy()
    arrays["multi_full_values_after_step"] = multi.full_values.copy()
    v_ode.x.array[:] = np.arange(10.0)
    multi.from_dolfin()
    arrays["multi_full_values_after_from_dolfin"] = multi.full_values.copy()
    for mk in (0, 1, 2):
        arrays[f"multi_values_marker{mk}"] = multi.values(mk).copy()

    # --- MonodomainSplittingSolver.step: call order + values with a recording fake PDE ---------------
    class FakePDE:
        """state <- 0.5 * v_ + 1  (an affine 'diffusion step' so that values are checkable)"""

        def __init__(self, n, log):
            self.state = FakeFunction(n)
            self.v_ = np.zeros(n)
            self.log = log

        def assign_previous(self):
            self.log.append("pde.assign_previous")
            self.v_[:] = self.state.x.array

        def step(self, interval):
            self.log.append(f"pde.step({interval[0]:.6f},{interval[1]:.6f})")
            self.state.x.array[:] = 0.5 * self.v_ + 1.0

    for theta in (1.0, 0.5):
        log = []
        n = 5
        pde = FakePDE(n, log)
        v_ode = FakeFunction(n)
        init = np.zeros((2, n))
        init[0] = np.linspace(1.0, 2.0, n)
        init[1] = np.linspace(-1.0, 1.0, n)

        def logged_fun(states, t, dt, parameters, _log=log):
            _log.append(f"ode.fun(t={t:.6f},dt={dt:.6f})")
            return simple_ode_forward_euler(states, t, dt, parameters)

        ode = odesolver.DolfinODESolver(v_ode=v_ode, v_pde=pde.state, init_states=init, parameters=np.array([1.0, 1.0]),
                                        fun=logged_fun, num_states=2, v_index=0)
        solver = msolver.MonodomainSplittingSolver(pde=pde, ode=ode, theta=theta)
        tag = f"split_theta{theta:g}".replace(".", "p")
        meta[f"{tag}_calls_init"] = list(log)
        del log[:]
        solver.step((0.0, 0.1))
        meta[f"{tag}_calls_step"] = list(log)
        arrays[f"{tag}_init_states"] = init
        arrays[f"{tag}_values_after_step"] = ode.values.copy()
        arrays[f"{tag}_pde_state_after_step"] = pde.state.x.array.copy()
        arrays[f"{tag}_pde_prev_after_step"] = pde.v_.copy()
        del log[:]
        solver.solve((0.1, 0.4), dt=0.1)
        meta[f"{tag}_calls_solve"] = list(log)
        arrays[f"{tag}_values_after_solve"] = ode.values.copy()

    np.savez_compressed(HERE / "splitting_reference.npz", **arrays)
    (HERE / "splitting_reference.json").write_text(json.dumps(meta, indent=1))
    print("splitting_reference.npz:", len(arrays), "arrays;", len(meta), "call logs")


if __name__ == "__main__":
    if not REF.is_dir():
        raise SystemExit("/root/reference is not present: fixtures can only be regenerated in the build container")
    which = sys.argv[1:] or ["tp06", "torord", "torord_land", "splitting"]
    if "tp06" in which:
        make_tp06_spec()
    if "torord" in which:
        make_torord_spec()
    if "torord_land" in which:
        make_torord_land_spec()
    if "splitting" in which:
        make_splitting_reference()
