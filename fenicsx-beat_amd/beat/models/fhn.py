"""FitzHugh-Nagumo (forward Euler) in the two parameterisations the reference ships:
``forward_euler_demo`` = demos/fitzhughnagumo.py:45-80,224-225 (states [s, V], 10 parameters),
``forward_euler_readme`` = README.md:58-89 (states [s, v], 11 parameters)."""

from .. import _hip
from ._base import DeviceModel

forward_euler_demo = DeviceModel(
    "fitzhughnagumo_forward_euler_demo", _hip.MODEL_FHN_DEMO, dict(s=0.0, V=-85.0),
    dict(V_peak=40.0, V_rest=-85.0, a=0.13, b=0.013, c_1=0.26, c_2=0.1, c_3=1.0, stim_amplitude=80.0,
         stim_duration=1.0, stim_start=1.0), "V",
)
forward_euler_readme = DeviceModel(
    "fitzhughnagumo_forward_euler", _hip.MODEL_FHN_README, dict(s=0.0, v=-85.0),
    dict(c_1=0.26, c_2=0.1, c_3=1.0, a=0.13, b=0.013, v_amp=125.0, v_rest=-85.0, v_peak=40.0,
         stim_amplitude=100.0, stim_duration=1.0, stim_start=0.0), "v",
)
