"""Built-in device cell models (handles recognised by the HIP backend)."""

from . import fhn, simple, torord, torord_land, tp06
from ._base import DeviceModel
from .ode_file import OdeFileModel, from_ode

__all__ = ["DeviceModel", "OdeFileModel", "fhn", "from_ode", "simple", "torord", "torord_land", "tp06"]
