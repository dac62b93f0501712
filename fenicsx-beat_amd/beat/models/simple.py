"""``v' = -a s, s' = b v`` advanced by forward Euler: the ODE of the reference's analytic tests
(tests/test_odesolver.py:11-17 with parameters (a, b); tests/test_monodomain_solver.py:25-30 with
``parameters=None`` meaning a = b = 1)."""

from .. import _hip
from ._base import DeviceModel

forward_euler = DeviceModel("simple_ode_forward_euler", _hip.MODEL_SIMPLE_ODE, dict(v=0.0, s=0.0), dict(a=1.0, b=1.0), "v")
