"""Built-in device cell models (handles recognised by the HIP backend)."""

from . import fhn, simple, torord, torord_land, tp06
from ._base import DeviceModel

__all__ = ["DeviceModel", "fhn", "simple", "torord", "torord_land", "tp06"]
