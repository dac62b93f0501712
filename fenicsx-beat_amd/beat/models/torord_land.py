"""ToR-ORd with dynamic chloride coupled to the Land contraction model (52 states, 140 parameters).

Specification: odes/torord/ToRORd_dynCl_endo_Land.ode of the reference (the electrophysiology of
odes/torord/ToRORd_dynCl_endo.ode with troponin-bound calcium, tropomyosin, the two cross-bridge states, their
distortions and a dashpot; state / parameter order = order of appearance there, ``celltype`` as in
:mod:`beat.models.torord`, stretch ``lmbda`` and stretch rate ``dLambda`` are parameters).  No demo of the reference
advances this file; it ships next to the one the ventricular demos use.  ``generalized_rush_larsen`` stands where the
gotranx-generated function of the same name would; the kernel is the LAND instance of csrc/torord_dyncl.h."""

from .. import _hip
from ._base import DeviceModel
from ._torord_land_data import PARAMETERS as _PARAMETERS
from ._torord_land_data import STATES as _STATES

generalized_rush_larsen = DeviceModel("torord_land_generalized_rush_larsen", _hip.MODEL_TORORD_LAND_GRL1, _STATES,
                                      _PARAMETERS, "v")
init_state_values = generalized_rush_larsen.init_state_values
init_parameter_values = generalized_rush_larsen.init_parameter_values
state_index = generalized_rush_larsen.state_index
parameter_index = generalized_rush_larsen.parameter_index
