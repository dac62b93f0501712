"""ten Tusscher & Panfilov 2006 epicardial cell model (19 states, 53 parameters).

Specification: odes/tentusscher_panfilov_2006/tentusscher_panfilov_2006_epi_cell.ode (state and
parameter order = order of appearance there; defaults :45-169).  ``generalized_rush_larsen`` is
the drop-in for the gotranx-generated function of the same name used by
demos/niederer_benchmark.py:82-99; the arithmetic lives in csrc/ionic_models.h (Tp06Grl1).
"""

from .. import _hip
from ._base import DeviceModel

_STATES = dict(
    Xr1=0.00621, Xr2=0.4712, Xs=0.0095, m=0.00172, h=0.7444, j=0.7045, d=3.373e-05, f=0.7888,
    f2=0.9755, fCass=0.9953, s=0.999998, r=2.42e-08, R_prime=0.9073, Ca_i=0.000126, Ca_SR=3.64,
    Ca_ss=0.00036, Na_i=8.604, V=-85.23, K_i=136.89,
)
_PARAMETERS = dict(
    P_kna=0.03, g_K1=5.405, g_Kr=0.153, g_Ks=0.392, g_Na=14.838, g_bna=0.00029, g_CaL=0.0398,
    g_bca=0.000592, g_to=0.294, P_NaK=2.724, K_mk=1.0, K_mNa=40.0, K_NaCa=1000.0, K_sat=0.1,
    alpha=2.5, gamma=0.35, Km_Ca=1.38, Km_Nai=87.5, g_pCa=0.1238, K_pCa=0.0005, g_pK=0.0146,
    Ca_o=2.0, k1_prime=0.15, k2_prime=0.045, k3=0.06, k4=0.005, EC=1.5, max_sr=2.5, min_sr=1.0,
    V_rel=0.102, V_xfer=0.0038, K_up=0.00025, V_leak=0.00036, Vmax_up=0.006375, Buf_c=0.2,
    K_buf_c=0.001, Buf_sr=10.0, K_buf_sr=0.3, Buf_ss=0.4, K_buf_ss=0.00025, V_sr=1094.0,
    V_ss=54.68, Na_o=140.0, R=8.314, T=310.0, F=96.485, Cm=185.0, V_c=16404.0, stim_start=10.0,
    stim_period=1000.0, stim_duration=1.0, stim_amplitude=-52.0, K_o=5.4,
)

generalized_rush_larsen = DeviceModel("tp06_generalized_rush_larsen", _hip.MODEL_TP06_GRL1, _STATES, _PARAMETERS, "V")

init_state_values = generalized_rush_larsen.init_state_values
init_parameter_values = generalized_rush_larsen.init_parameter_values
state_index = generalized_rush_larsen.state_index
parameter_index = generalized_rush_larsen.parameter_index
state = {k: i for i, k in enumerate(_STATES)}
parameter = {k: i for i, k in enumerate(_PARAMETERS)}
