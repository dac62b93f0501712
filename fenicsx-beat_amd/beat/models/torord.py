"""ToR-ORd with dynamic chloride (endocardial defaults; 45 states, 112 parameters).

Specification: odes/torord/ToRORd_dynCl_endo.ode of the reference (state / parameter order = order of
appearance there; ``celltype`` selects endo / epi / mid as in demos/biv_endocardial.py:124-173).
``generalized_rush_larsen`` is the drop-in for the gotranx-generated function of the same name; its kernel
(csrc/torord_dyncl.h) is hand-organised; the names / defaults module was written by tools/gen_model_data.py from the
model specification."""

from .. import _hip
from ._base import DeviceModel
from ._torord_dyncl_data import PARAMETERS as _PARAMETERS
from ._torord_dyncl_data import STATES as _STATES

generalized_rush_larsen = DeviceModel("torord_dyncl_generalized_rush_larsen", _hip.MODEL_TORORD_DYNCL_GRL1, _STATES,
                                      _PARAMETERS, "v")
init_state_values = generalized_rush_larsen.init_state_values
init_parameter_values = generalized_rush_larsen.init_parameter_values
state_index = generalized_rush_larsen.state_index
parameter_index = generalized_rush_larsen.parameter_index
