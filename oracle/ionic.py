"""ORACLE (test infrastructure only -- never imported by the product path).

NumPy restatement of the pointwise ionic steps (``fun(states=, t=, parameters=, dt=)`` of
``src/beat/odesolver.py:67-79``) for the cell models on the hot path:

* ``simple_ode_forward_euler``  -- tests/test_odesolver.py:11-17, tests/test_monodomain_solver.py:25-30
* ``fhn_demo_forward_euler``    -- demos/fitzhughnagumo.py:45-80,224-225 (states [s, V], 10 parameters)
* ``fhn_readme_forward_euler``  -- README.md:58-89 (states [s, v], 11 parameters)
* ``tp06_generalized_rush_larsen`` -- ten Tusscher-Panfilov 2006 epicardial model,
  odes/tentusscher_panfilov_2006/tentusscher_panfilov_2006_epi_cell.ode:36-322, advanced with the
  first-order generalized Rush-Larsen scheme that ``gotranx`` generates for the demos
  (demos/niederer_benchmark.py:82-99).

gotranx is an un-vendored, un-pinned dependency (pyproject.toml:57-64); its GRL1 scheme is
restated from its published algorithm: for every state y_i with RHS f_i(y) (all intermediate
expressions resolved), take the self-derivative ``J_i = d f_i / d y_i``; if it is identically
zero the state is advanced by forward Euler, ``y_i += dt f_i``; otherwise
``y_i += f_i (exp(J_i dt) - 1) / J_i`` where ``|J_i| > 1e-8`` and ``dt f_i`` elsewhere.  All f_i,
J_i are evaluated at the old state.  For TP06 every J_i is non-zero.

Which derivative gotranx takes is not stated in the reference; it is PINNED HERE BY THE
REFERENCE'S NIEDERER TABLE (demos/niederer_benchmark.py:315-319): with the total self-derivative
(this file) the nine activation times at dx = 0.5 mm land within one dt of the committed rows,
whereas differentiating only the explicit occurrence of y_i in the written expression (which would
leave V, Ca_i, Ca_SR, Ca_ss, Na_i, K_i on forward Euler) is 8-13 dt early at dt = 0.05
(tests/test_oracle_pins.py::test_niederer_table_pins_the_grl1_variant).

PARITY UNPINNED for individual per-step GRL1 values: the reference holds no numerical output of a
gotranx-generated step.  What *is* pinned: (i) the RHS f and the J_i of this file against an
independent evaluation of the reference's ``.ode`` text (tests/golden/tp06_spec.npz, generated
by tests/golden/make_golden.py), (ii) the end-to-end Niederer activation times.
"""

from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------
# trivial models
# ----------------------------------------------------------------------------------------------


def simple_ode_forward_euler(states, t, dt, parameters=None):
    v, s = states
    a, b = (1.0, 1.0) if parameters is None else parameters
    values = np.zeros_like(states)
    values[0] = v - a * s * dt
    values[1] = s + b * v * dt
    return values


def fhn_demo_rhs(t, states, parameters):
    s = states[0]
    V = states[1]
    V_peak, V_rest, a, b, c_1, c_2, c_3, stim_amplitude, stim_duration, stim_start = parameters
    values = np.zeros_like(states, dtype=np.float64)
    V_amp = V_peak - V_rest
    i_Stim = np.where(t >= stim_start and t <= stim_start + stim_duration, stim_amplitude, 0)
    ds_dt = b * (-c_3 * s + (V - V_rest))
    values[0] = ds_dt
    V_th = V_amp * a + V_rest
    I = -s * (c_2 / V_amp) * (V - V_rest) + (((c_1 / V_amp**2) * (V - V_rest)) * (V - V_th)) * (
        -V + V_peak
    )
    values[1] = I + i_Stim
    return values


def fhn_demo_forward_euler(states, t, dt, parameters):
    return states + dt * fhn_demo_rhs(t, states, parameters)


def fhn_readme_forward_euler(states, t, dt, parameters):
    s, v = states
    (c_1, c_2, c_3, a, b, v_amp, v_rest, v_peak, stim_amplitude, stim_duration, stim_start) = parameters
    i_app = np.where(np.logical_and(t > stim_start, t < stim_start + stim_duration), stim_amplitude, 0)
    values = np.zeros_like(states)
    ds_dt = b * (-c_3 * s + (v - v_rest))
    values[0] = ds_dt * dt + s
    v_th = v_amp * a + v_rest
    I = -s * (c_2 / v_amp) * (v - v_rest) + (((c_1 / v_amp**2) * (v - v_rest)) * (v - v_th)) * (
        -v + v_peak
    )
    dV_dt = I + i_app
    values[1] = v + dV_dt * dt
    return values


# ----------------------------------------------------------------------------------------------
# ten Tusscher - Panfilov 2006 (epi)
# ----------------------------------------------------------------------------------------------

# order of appearance in the .ode file (lines 45-169)
TP06_STATES = (
    "Xr1", "Xr2", "Xs", "m", "h", "j", "d", "f", "f2", "fCass", "s", "r",
    "R_prime", "Ca_i", "Ca_SR", "Ca_ss", "Na_i", "V", "K_i",
)
TP06_STATE_DEFAULTS = dict(
    Xr1=0.00621, Xr2=0.4712, Xs=0.0095, m=0.00172, h=0.7444, j=0.7045, d=3.373e-05, f=0.7888,
    f2=0.9755, fCass=0.9953, s=0.999998, r=2.42e-08, R_prime=0.9073, Ca_i=0.000126, Ca_SR=3.64,
    Ca_ss=0.00036, Na_i=8.604, V=-85.23, K_i=136.89,
)
TP06_PARAMETER_DEFAULTS = dict(
    P_kna=0.03, g_K1=5.405, g_Kr=0.153, g_Ks=0.392, g_Na=14.838, g_bna=0.00029, g_CaL=0.0398,
    g_bca=0.000592, g_to=0.294, P_NaK=2.724, K_mk=1.0, K_mNa=40.0, K_NaCa=1000.0, K_sat=0.1,
    alpha=2.5, gamma=0.35, Km_Ca=1.38, Km_Nai=87.5, g_pCa=0.1238, K_pCa=0.0005, g_pK=0.0146,
    Ca_o=2.0, k1_prime=0.15, k2_prime=0.045, k3=0.06, k4=0.005, EC=1.5, max_sr=2.5, min_sr=1.0,
    V_rel=0.102, V_xfer=0.0038, K_up=0.00025, V_leak=0.00036, Vmax_up=0.006375, Buf_c=0.2,
    K_buf_c=0.001, Buf_sr=10.0, K_buf_sr=0.3, Buf_ss=0.4, K_buf_ss=0.00025, V_sr=1094.0,
    V_ss=54.68, Na_o=140.0, R=8.314, T=310.0, F=96.485, Cm=185.0, V_c=16404.0, stim_start=10.0,
    stim_period=1000.0, stim_duration=1.0, stim_amplitude=-52.0, K_o=5.4,
)
TP06_PARAMETERS = tuple(TP06_PARAMETER_DEFAULTS)


def tp06_state_index(name: str) -> int:
    return TP06_STATES.index(name)


def tp06_parameter_index(name: str) -> int:
    return TP06_PARAMETERS.index(name)


def tp06_init_state_values(**values) -> np.ndarray:
    d = dict(TP06_STATE_DEFAULTS)
    for k, v in values.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return np.array([d[k] for k in TP06_STATES], dtype=np.float64)


def tp06_init_parameter_values(**values) -> np.ndarray:
    d = dict(TP06_PARAMETER_DEFAULTS)
    for k, v in values.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return np.array([d[k] for k in TP06_PARAMETERS], dtype=np.float64)


class _NumpyNS:
    exp = staticmethod(np.exp)
    log = staticmethod(np.log)
    sqrt = staticmethod(np.sqrt)
    floor = staticmethod(np.floor)

    @staticmethod
    def where(c, a, b):
        return np.where(c, a, b)

    @staticmethod
    def lt(a, b):
        return a < b

    @staticmethod
    def ge(a, b):
        return a >= b

    @staticmethod
    def le(a, b):
        return a <= b

    @staticmethod
    def land(a, b):
        return np.logical_and(a, b)


def _sympy_ns():
    import sympy

    class NS:
        exp = staticmethod(sympy.exp)
        log = staticmethod(sympy.log)
        sqrt = staticmethod(sympy.sqrt)
        floor = staticmethod(sympy.floor)

        @staticmethod
        def where(c, a, b):
            return sympy.Piecewise((a, c), (b, True))

        lt = staticmethod(sympy.Lt)
        ge = staticmethod(sympy.Ge)
        le = staticmethod(sympy.Le)
        land = staticmethod(sympy.And)

    return NS


def tp06_rhs(states, t, parameters, ns=_NumpyNS):
    """dy/dt of the 19 states, written once for NumPy arrays and for SymPy symbols (``ns`` supplies
    exp/log/sqrt/where/...)."""
    exp, log, sqrt = ns.exp, ns.log, ns.sqrt
    (Xr1, Xr2, Xs, m, h, j, d, f, f2, fCass, s, r, R_prime, Ca_i, Ca_SR, Ca_ss, Na_i, V, K_i) = states
    (P_kna, g_K1, g_Kr, g_Ks, g_Na, g_bna, g_CaL, g_bca, g_to, P_NaK, K_mk, K_mNa, K_NaCa, K_sat,
     alpha, gamma, Km_Ca, Km_Nai, g_pCa, K_pCa, g_pK, Ca_o, k1_prime, k2_prime, k3, k4, EC, max_sr,
     min_sr, V_rel, V_xfer, K_up, V_leak, Vmax_up, Buf_c, K_buf_c, Buf_sr, K_buf_sr, Buf_ss,
     K_buf_ss, V_sr, V_ss, Na_o, R, T, F, Cm, V_c, stim_start, stim_period, stim_duration,
     stim_amplitude, K_o) = parameters

    # Reversal potentials (.ode:174-178)
    E_Na = R * T / F * log(Na_o / Na_i)
    E_K = R * T / F * log(K_o / K_i)
    E_Ks = R * T / F * log((K_o + P_kna * Na_o) / (K_i + P_kna * Na_i))
    E_Ca = 0.5 * R * T / F * log(Ca_o / Ca_i)

    # Inward rectifier (.ode:180-184)
    alpha_K1 = 0.1 / (1 + exp(0.06 * (V - E_K - 200)))
    beta_K1 = (3 * exp(0.0002 * (V - E_K + 100)) + exp(0.1 * (V - E_K - 10))) / (
        1 + exp(-0.5 * (V - E_K))
    )
    xK1_inf = alpha_K1 / (alpha_K1 + beta_K1)
    i_K1 = g_K1 * xK1_inf * sqrt(K_o / 5.4) * (V - E_K)

    # Rapid delayed rectifier (.ode:186-201)
    i_Kr = g_Kr * sqrt(K_o / 5.4) * Xr1 * Xr2 * (V - E_K)
    xr1_inf = 1 / (1 + exp((-26 - V) / 7))
    alpha_xr1 = 450 / (1 + exp((-45 - V) / 10))
    beta_xr1 = 6 / (1 + exp((V + 30) / 11.5))
    tau_xr1 = 1 * alpha_xr1 * beta_xr1
    dXr1_dt = (xr1_inf - Xr1) / tau_xr1
    xr2_inf = 1 / (1 + exp((V + 88) / 24))
    alpha_xr2 = 3 / (1 + exp((-60 - V) / 20))
    beta_xr2 = 1.12 / (1 + exp((V - 60) / 20))
    tau_xr2 = 1 * alpha_xr2 * beta_xr2
    dXr2_dt = (xr2_inf - Xr2) / tau_xr2

    # Slow delayed rectifier (.ode:203-211)
    i_Ks = g_Ks * Xs**2 * (V - E_Ks)
    xs_inf = 1 / (1 + exp((-5 - V) / 14))
    alpha_xs = 1400 / sqrt(1 + exp((5 - V) / 6))
    beta_xs = 1 / (1 + exp((V - 35) / 15))
    tau_xs = 1 * alpha_xs * beta_xs + 80
    dXs_dt = (xs_inf - Xs) / tau_xs

    # Fast sodium (.ode:213-235)
    i_Na = g_Na * m**3 * h * j * (V - E_Na)
    m_inf = 1 / (1 + exp((-56.86 - V) / 9.03)) ** 2
    alpha_m = 1 / (1 + exp((-60 - V) / 5))
    beta_m = 0.1 / (1 + exp((V + 35) / 5)) + 0.1 / (1 + exp((V - 50) / 200))
    tau_m = 1 * alpha_m * beta_m
    dm_dt = (m_inf - m) / tau_m
    h_inf = 1 / (1 + exp((V + 71.55) / 7.43)) ** 2
    lt = ns.lt(V, -40)
    alpha_h = ns.where(lt, 0.057 * exp(-(V + 80) / 6.8), 0)
    beta_h = ns.where(
        lt,
        2.7 * exp(0.079 * V) + 310000 * exp(0.3485 * V),
        0.77 / (0.13 * (1 + exp((V + 10.66) / -11.1))),
    )
    tau_h = 1 / (alpha_h + beta_h)
    dh_dt = (h_inf - h) / tau_h
    j_inf = 1 / (1 + exp((V + 71.55) / 7.43)) ** 2
    alpha_j = ns.where(
        lt,
        (-25428 * exp(0.2444 * V) - 6.948e-6 * exp(-0.04391 * V)) * (V + 37.78) / 1
        / (1 + exp(0.311 * (V + 79.23))),
        0,
    )
    beta_j = ns.where(
        lt,
        0.02424 * exp(-0.01052 * V) / (1 + exp(-0.1378 * (V + 40.14))),
        0.6 * exp(0.057 * V) / (1 + exp(-0.1 * (V + 32))),
    )
    tau_j = 1 / (alpha_j + beta_j)
    dj_dt = (j_inf - j) / tau_j

    # Sodium background (.ode:237-238)
    i_b_Na = g_bna * (V - E_Na)

    # L-type calcium (.ode:240-268)
    i_CaL = (
        g_CaL * d * f * f2 * fCass * 4 * (V - 15) * F**2 / (R * T)
        * (0.25 * Ca_ss * exp(2 * (V - 15) * F / (R * T)) - Ca_o)
        / (exp(2 * (V - 15) * F / (R * T)) - 1)
    )
    d_inf = 1 / (1 + exp((-8 - V) / 7.5))
    alpha_d = 1.4 / (1 + exp((-35 - V) / 13)) + 0.25
    beta_d = 1.4 / (1 + exp((V + 5) / 5))
    gamma_d = 1 / (1 + exp((50 - V) / 20))
    tau_d = 1 * alpha_d * beta_d + gamma_d
    dd_dt = (d_inf - d) / tau_d
    f_inf = 1 / (1 + exp((V + 20) / 7))
    tau_f = (
        1102.5 * exp(-((V + 27) ** 2) / 225) + 200 / (1 + exp((13 - V) / 10))
        + 180 / (1 + exp((V + 30) / 10)) + 20
    )
    df_dt = (f_inf - f) / tau_f
    f2_inf = 0.67 / (1 + exp((V + 35) / 7)) + 0.33
    tau_f2 = (
        562 * exp(-((V + 27) ** 2) / 240) + 31 / (1 + exp((25 - V) / 10))
        + 80 / (1 + exp((V + 30) / 10))
    )
    df2_dt = (f2_inf - f2) / tau_f2
    fCass_inf = 0.6 / (1 + (Ca_ss / 0.05) ** 2) + 0.4
    tau_fCass = 80 / (1 + (Ca_ss / 0.05) ** 2) + 2
    dfCass_dt = (fCass_inf - fCass) / tau_fCass

    # Calcium background, transient outward (.ode:270-284)
    i_b_Ca = g_bca * (V - E_Ca)
    i_to = g_to * r * s * (V - E_K)
    s_inf = 1 / (1 + exp((V + 20) / 5))
    tau_s = 85 * exp(-((V + 45) ** 2) / 320) + 5 / (1 + exp((V - 20) / 5)) + 3
    ds_dt = (s_inf - s) / tau_s
    r_inf = 1 / (1 + exp((20 - V) / 6))
    tau_r = 9.5 * exp(-((V + 40) ** 2) / 1800) + 0.8
    dr_dt = (r_inf - r) / tau_r

    # Pumps and exchanger (.ode:286-296)
    i_NaK = (
        P_NaK * K_o / (K_o + K_mk) * Na_i / (Na_i + K_mNa)
        / (1 + 0.1245 * exp(-0.1 * V * F / (R * T)) + 0.0353 * exp(-V * F / (R * T)))
    )
    i_NaCa = (
        K_NaCa
        * (exp(gamma * V * F / (R * T)) * Na_i**3 * Ca_o
           - exp((gamma - 1) * V * F / (R * T)) * Na_o**3 * Ca_i * alpha)
        / ((Km_Nai**3 + Na_o**3) * (Km_Ca + Ca_o) * (1 + K_sat * exp((gamma - 1) * V * F / (R * T))))
    )
    i_p_Ca = g_pCa * Ca_i / (Ca_i + K_pCa)
    i_p_K = g_pK * (V - E_K) / (1 + exp((25 - V) / 5.98))

    # Calcium dynamics (.ode:298-316)
    i_up = Vmax_up / (1 + K_up**2 / Ca_i**2)
    i_leak = V_leak * (Ca_SR - Ca_i)
    i_xfer = V_xfer * (Ca_ss - Ca_i)
    kcasr = max_sr - (max_sr - min_sr) / (1 + (EC / Ca_SR) ** 2)
    ddt_Ca_i_total = -(i_b_Ca + i_p_Ca - 2 * i_NaCa) * Cm / (2 * V_c * F) + (i_leak - i_up) * V_sr / V_c + i_xfer
    f_JCa_i_free = 1 / (1 + Buf_c * K_buf_c / (Ca_i + K_buf_c) ** 2)
    f_JCa_sr_free = 1 / (1 + Buf_sr * K_buf_sr / (Ca_SR + K_buf_sr) ** 2)
    f_JCa_ss_free = 1 / (1 + Buf_ss * K_buf_ss / (Ca_ss + K_buf_ss) ** 2)
    dCa_i_dt = ddt_Ca_i_total * f_JCa_i_free
    k1 = k1_prime / kcasr
    k2 = k2_prime * kcasr
    O = k1 * Ca_ss**2 * R_prime / (k3 + k1 * Ca_ss**2)
    dR_prime_dt = -k2 * Ca_ss * R_prime + k4 * (1 - R_prime)
    i_rel = V_rel * O * (Ca_SR - Ca_ss)
    ddt_Ca_sr_total = i_up - (i_rel + i_leak)
    ddt_Ca_ss_total = -i_CaL * Cm / (2 * V_ss * F) + i_rel * V_sr / V_ss - i_xfer * V_c / V_ss
    dCa_SR_dt = ddt_Ca_sr_total * f_JCa_sr_free
    dCa_ss_dt = ddt_Ca_ss_total * f_JCa_ss_free

    # Sodium, membrane, potassium (.ode:318-322)
    dNa_i_dt = -(i_Na + i_b_Na + 3 * i_NaK + 3 * i_NaCa) / (V_c * F) * Cm
    tmod = t - ns.floor(t / stim_period) * stim_period
    i_Stim = ns.where(ns.land(ns.ge(tmod, stim_start), ns.le(tmod, stim_start + stim_duration)), stim_amplitude, 0)
    dV_dt = -(i_K1 + i_to + i_Kr + i_Ks + i_CaL + i_NaK + i_Na + i_b_Na + i_NaCa + i_b_Ca + i_p_K
              + i_p_Ca + i_Stim)
    dK_i_dt = -(i_K1 + i_to + i_Kr + i_Ks + i_p_K + i_Stim - 2 * i_NaK) / (V_c * F) * Cm

    return [dXr1_dt, dXr2_dt, dXs_dt, dm_dt, dh_dt, dj_dt, dd_dt, df_dt, df2_dt, dfCass_dt, ds_dt,
            dr_dt, dR_prime_dt, dCa_i_dt, dCa_SR_dt, dCa_ss_dt, dNa_i_dt, dV_dt, dK_i_dt]


_TP06_JAC = None


def _tp06_linearized_function():
    """J_i = d f_i / d y_i with every intermediate expression resolved (total self-derivative),
    derived symbolically once and compiled with lambdify(cse=True)."""
    global _TP06_JAC
    if _TP06_JAC is None:
        import sympy

        ys = sympy.symbols(" ".join(TP06_STATES), real=True)
        ps = sympy.symbols(" ".join("p_" + n for n in TP06_PARAMETERS), real=True)
        tt = sympy.Symbol("t", real=True)
        fs = tp06_rhs(ys, tt, ps, ns=_sympy_ns())
        J = [sympy.diff(fi, yi) for fi, yi in zip(fs, ys)]
        nonzero = [not (Ji == 0) for Ji in J]
        fn = sympy.lambdify([*ys, tt, *ps], J, modules="numpy", cse=True)
        _TP06_JAC = (fn, nonzero)
    return _TP06_JAC


def tp06_rhs_and_linearized(states, t, parameters):
    """Returns (f, J): f[i] = dy_i/dt, J[i] = total d f_i / d y_i (None where identically 0)."""
    states = np.asarray(states, dtype=np.float64)
    parameters = np.asarray(parameters, dtype=np.float64)
    with np.errstate(all="ignore"):
        fvals = tp06_rhs(states, t, parameters)
        fn, nonzero = _tp06_linearized_function()
        Jraw = fn(*states, t, *parameters)
    shape = np.broadcast(*fvals).shape
    fvals = [np.broadcast_to(np.asarray(v, dtype=np.float64), shape) for v in fvals]
    J = [np.broadcast_to(np.asarray(v, dtype=np.float64), shape) if nz else None for v, nz in zip(Jraw, nonzero)]
    return fvals, J


def tp06_generalized_rush_larsen(states, t, dt, parameters, delta=1e-8):
    states = np.asarray(states, dtype=np.float64)
    fvals, J = tp06_rhs_and_linearized(states, t, parameters)
    values = np.zeros_like(states)
    for i, (y, fi, Ji) in enumerate(zip(states, fvals, J)):
        if Ji is None:
            values[i] = y + fi * dt
        else:
            with np.errstate(all="ignore"):
                values[i] = y + np.where(np.abs(Ji) > delta, fi * (np.exp(Ji * dt) - 1) / Ji, fi * dt)
    return values


def tp06_forward_euler(states, t, dt, parameters):
    states = np.asarray(states, dtype=np.float64)
    with np.errstate(all="ignore"):
        fvals = tp06_rhs(states, t, np.asarray(parameters, dtype=np.float64))
    return states + dt * np.array(np.broadcast_arrays(*fvals))


MODELS = {
    "simple_ode": (simple_ode_forward_euler, 2),
    "fhn_demo": (fhn_demo_forward_euler, 2),
    "fhn_readme": (fhn_readme_forward_euler, 2),
    "tp06_grl1": (tp06_generalized_rush_larsen, 19),
}


def _register_torord():
    from . import torord  # hand restatement of the 45-state ToR-ORd-dynCl model (own module: it is long)

    MODELS["torord_dyncl_grl1"] = (torord.torord_generalized_rush_larsen, 45)


_register_torord()
